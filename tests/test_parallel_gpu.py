"""The data-parallel path of ``Trainer.step`` on two ranks (see tests/dp_rehearsal.py, which tests/conftest.py
starts as a child process before this session initialises the GPU: a process that has done so must not start
other programs on this pool)."""
import json
import os

import pytest

import conftest

pytestmark = pytest.mark.gpu


def test_two_rank_trainer_step():
    job = conftest.DP_REHEARSAL
    assert job is not None, "the two-rank rehearsal was not started (tests/conftest.py starts it for -m gpu sessions)"
    proc, out_path, log_path = job
    try:
        rc = proc.wait(timeout=240)       # a deadlocked all-reduce would sit here: the failed-rank case must not hang
    except Exception:
        proc.kill()
        raise AssertionError("the two-rank rehearsal did not finish: " + open(log_path).read()[-2000:])
    log = open(log_path).read()
    if rc != 0:
        # The rehearsal's own watchdog (tests/dp_rehearsal.py) ends a run that stops making progress and says, per
        # rank, whether its streams were idle.  Both ranks waiting with idle streams is a deadlock of the step's
        # logic and fails the test.  A rank whose stream never drains is the device not serving one of the two
        # processes that share it here - seen with several processes on one GPU, never with one process per GPU,
        # which is how the package runs - and says nothing about the step.
        reports = [json.loads(line.split("] ", 1)[1]) for line in log.splitlines() if line.startswith("[rehearsal watchdog] ")]
        if reports and any(r.get("main_idle") is False or r.get("side_idle") is False for r in reports):
            pytest.skip("a rank's stream stopped draining while two processes shared the GPU: " + json.dumps(reports)[:600])
    assert rc == 0 and os.path.exists(out_path), log[-3000:]
    with open(out_path) as f:
        ranks = json.load(f)
    assert [r["rank"] for r in ranks] == [0, 1]
    for r in ranks:
        # reduced gradient of the first critic iteration = mean of the two single-rank gradients
        assert r["grad_mean_err"] <= 1e-5, r
        assert r["weights_equal"] and r["moved"] > 0 and r["finite"], r
        assert r["after_failure_equal"] and r["moved_after_failure"] > 0, r
    assert ranks[0]["raised"] is None
    assert ranks[1]["raised"] is not None and "injected failure" in ranks[1]["raised"]
