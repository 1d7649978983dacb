import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


DP_REHEARSAL = None    # (process, result path, log path) of tests/dp_rehearsal.py, see pytest_configure


def _start_dp_rehearsal():
    """Two data-parallel ranks of the real training step as a child process (``torch.distributed.run``), started
    BEFORE this process makes its first HIP call: on the GPU pool a process that has initialised the GPU must
    not start other programs.  tests/test_parallel_gpu.py waits for it and checks what it wrote."""
    import socket
    import subprocess
    import tempfile
    global DP_REHEARSAL
    tmp = tempfile.mkdtemp(prefix="aig_dp_")
    out_path, log_path = os.path.join(tmp, "dp.json"), os.path.join(tmp, "dp.log")
    keep = os.path.join(ROOT, "gpurun_out")          # (kept with the run's other outputs where that directory exists)
    if os.path.isdir(keep) and os.access(keep, os.W_OK):
        log_path = os.path.join(keep, "dp_rehearsal.log")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.update(AIG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_rehearsal.py"), out_path]
    log = open(log_path, "w")
    DP_REHEARSAL = (subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT), out_path, log_path)


RCCL_BENCH = None      # (process, result path, log path) of bench.py over a one-rank RCCL communicator


def _start_rccl_bench():
    """``bench.py`` with its multi-rank code switched on over a communicator of ONE rank (``AIG_DP_FORCE=1``): the
    launcher, ``init_process_group("nccl")``, the barriers, the gradient exchange of every critic iteration and the
    MAX / SUM reductions behind the JSON line, over the real RCCL on the box's single GPU.  Started like the
    rehearsal, before this process's first HIP call, and after the rehearsal has ended (one process on the GPU at
    a time); tests/test_parallel_gpu.py checks the line it prints."""
    import socket
    import subprocess
    import tempfile
    global RCCL_BENCH
    tmp = tempfile.mkdtemp(prefix="aig_rccl_")
    out_path, log_path = os.path.join(tmp, "bench.json"), os.path.join(tmp, "bench.log")
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep) and os.access(keep, os.W_OK):
        log_path = os.path.join(keep, "rccl_one_rank_bench.log")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.update(AIG_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AIG_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--no-side-workloads", "--sample-iters", "1"]
    RCCL_BENCH = (subprocess.Popen(cmd, env=env, stdout=open(out_path, "w"), stderr=open(log_path, "w"), cwd=ROOT),
                  out_path, log_path)


WATCH = {"test": None, "since": 0.0, "trainer": None, "reported": None}


def _start_step_watchdog(limit: float = 60.0):
    """When a test has been running for ``limit`` seconds, whether the last training step's two streams are idle is
    written to stderr (and to gpurun_out/hang_report.json where that directory exists) - once per test, ahead of
    the test's own time limit, which only shows the host's stack.  With ``AIG_TEST_PROBES=1`` every step of the
    session also records its probe events (``Trainer.probes``) and the report says which of them completed; by
    default the steps run as they do in production, without the probes' marker packets on the two streams."""
    import json
    import threading
    import time
    import weakref

    import torch

    from gan_amd import train as train_mod
    inner = train_mod.Trainer.step

    with_probes = os.environ.get("AIG_TEST_PROBES") == "1"

    def step(self, *a, **k):
        WATCH["trainer"] = weakref.ref(self)
        if with_probes:
            self.probes = {}
        return inner(self, *a, **k)
    train_mod.Trainer.step = step

    def watch():
        while True:
            time.sleep(1.0)
            test = WATCH["test"]
            if test is None or WATCH["reported"] == test or time.time() - WATCH["since"] < limit:
                continue
            WATCH["reported"] = test
            tr = WATCH["trainer"]() if WATCH["trainer"] is not None else None
            out = {"test": test, "running_for_s": round(time.time() - WATCH["since"], 1)}
            try:
                if tr is not None:
                    out.update(global_step=tr.global_step, captures=tr.captures,
                               main_idle=torch.cuda.default_stream(tr.device).query(), side_idle=tr.side_stream.query(),
                               probes={k: v.query() for k, v in dict(tr.probes or {}).items()})
            except Exception as e:      # a report, not a check
                out["error"] = repr(e)
            text = json.dumps(out)
            sys.stderr.write("\n[step watchdog] " + text + "\n")
            sys.stderr.flush()
            keep = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(keep) and os.access(keep, os.W_OK):
                with open(os.path.join(keep, "hang_report.json"), "a") as f:
                    f.write(text + "\n")

    threading.Thread(target=watch, daemon=True).start()


def pytest_runtest_setup(item):
    import time
    WATCH["test"], WATCH["since"] = item.nodeid, time.time()


def pytest_runtest_teardown(item, nextitem):
    WATCH["test"] = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    import torch
    wants_gpu = "gpu" in (config.getoption("markexpr", "") or "") and "not gpu" not in (config.getoption("markexpr", "") or "")
    names = " ".join(str(a) for a in config.args)
    selected = all("test_" not in os.path.basename(a.split("::")[0]) for a in config.args) or "test_parallel_gpu" in names
    # counting devices does not start the HIP runtime
    if wants_gpu and selected and torch.cuda.device_count() > 0 and os.environ.get("RANK") is None:
        _start_dp_rehearsal()
        # ... and given the GPU to itself: it takes a few seconds, and the tests then measure and compare one
        # process on the device, which is the configuration the package runs in (one process per GPU)
        import subprocess
        try:
            if os.environ.get("AIG_TEST_OVERLAP_REHEARSAL") != "1":      # (1: the tests start next to it)
                DP_REHEARSAL[0].wait(timeout=150)
        except subprocess.TimeoutExpired:
            pass          # tests/test_parallel_gpu.py reports it
        if DP_REHEARSAL[0].poll() is not None:
            _start_rccl_bench()
            try:
                RCCL_BENCH[0].wait(timeout=240)
            except subprocess.TimeoutExpired:
                pass      # tests/test_parallel_gpu.py reports it
    if wants_gpu and torch.cuda.device_count() > 0:
        _start_step_watchdog()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_names(prefix):
    return sorted(f[len(prefix):-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))
