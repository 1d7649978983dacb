"""Timing aid: aig_wgrad (F = 128) at three row counts, checked against a float64 product; run under
tools/prof_kernels.sh for the kernel's own duration."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _abi
dev = "cuda"
lib = _abi.lib()
lib.aig_wgrad_ws_bytes.restype = __import__("ctypes").c_size_t
for n in (1, 31, 33, 64, 65, 26411, 100000, 1000000):
    g = torch.randn(n, 128, device=dev); x = torch.randn(n, 128, device=dev)
    dW = torch.empty(128, 128, device=dev)
    wsb = lib.aig_wgrad_ws_bytes(n, 128)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    for i in range(20 if n > 1000 else 2):
        _abi.check(lib.aig_wgrad(n, 128, _abi.ptr(g), _abi.ptr(x), _abi.ptr(dW), _abi.ptr(ws), wsb, _abi.stream()), "wgrad")
    torch.cuda.synchronize()
    ref = (g.double().t() @ x.double())
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    print(n, "max rel err", err)
    assert err < 1e-5
