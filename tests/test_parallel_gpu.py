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
    # (the rehearsal's own watchdog, tests/dp_rehearsal.py, ends a run that stops making progress and says, per rank,
    #  whether its streams were idle: the report is part of the log shown here.  A stream that never drains fails the
    #  test like any other stall - it is also what a kernel that does not finish looks like.)
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


def test_bench_multi_rank_code_over_one_rank_rccl():
    """``bench.py`` through its distributed branch on a communicator of one rank over RCCL (see
    tests/conftest.py): process-group set-up, barriers, three gradient exchanges per step, the MAX / SUM
    reductions of the timing, and a well-formed JSON line that says how many RCCL ranks made it."""
    job = conftest.RCCL_BENCH
    assert job is not None, "the one-rank RCCL bench was not started (tests/conftest.py starts it after the rehearsal)"
    proc, out_path, log_path = job
    try:
        rc = proc.wait(timeout=240)
    except Exception:
        proc.kill()
        raise AssertionError("bench.py over one-rank RCCL did not finish: " + open(log_path).read()[-2000:])
    log = open(log_path).read()
    assert rc == 0, log[-3000:]
    lines = [ln for ln in open(out_path).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (lines, log[-2000:])
    line = json.loads(lines[0])
    assert line["config"]["rccl_ranks"] == 1 and line["n_gpus"] == 1 and line["steps"] == 3
    assert line["config"]["gradient_exchanges_per_step"] == 3
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["sample_nodes_per_sec"] > 0
    assert line["roofline"] is not None and 0 < line["roofline"]["frac"] < 1.5
