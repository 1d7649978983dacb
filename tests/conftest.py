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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    import torch
    wants_gpu = "gpu" in (config.getoption("markexpr", "") or "") and "not gpu" not in (config.getoption("markexpr", "") or "")
    names = " ".join(str(a) for a in config.args)
    selected = all("test_" not in os.path.basename(a.split("::")[0]) for a in config.args) or "test_parallel_gpu" in names
    # counting devices does not start the HIP runtime
    if wants_gpu and selected and torch.cuda.device_count() > 0 and os.environ.get("RANK") is None:
        _start_dp_rehearsal()


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
