"""Timing aid: one k_linear128 launch at row counts of 1, 2, 4, 6, 7, 8, 16 tiles per workgroup
(256 workgroups, 64-row tiles): the slope is the time per tile, the intercept the fixed cost of a launch."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _abi

dev = "cuda"
w = torch.randn(128, 128, device=dev)
lib = _abi.lib()
for tiles in (1, 2, 4, 6, 6.1035, 7, 8, 16, 32, 61.035):
    n = int(round(256 * 64 * tiles))
    x = torch.randn(n, 128, device=dev)
    y = torch.empty_like(x)
    ts = []
    for i in range(25):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _abi.check(lib.aig_linear128(n, _abi.ptr(x), _abi.ptr(w), 0, None, 0, _abi.ptr(y), _abi.stream()), "lin")
        b.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(a.elapsed_time(b) * 1e3)
    t = statistics.median(ts)
    print(f"tiles/wg {tiles:7.3f}  n {n:8d}  median {t:8.1f} us  min {min(ts):8.1f}  "
          f"{2 * n * 128 * 128 / t * 1e-6:7.1f} TFLOP/s")
