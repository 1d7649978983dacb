#!/bin/bash
# Tuning aid: aig_wgrad (F = 128) checked against float64 and timed by the kernel's own events, both forms.
cd $GRAFT_REPO_ROOT
for f in 0 1; do
AIG_WGRAD128_F32=$f timeout -k 10 120 python3 - <<PY || exit 1
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from gan_amd import _abi
lib = _abi.lib(); dev = "cuda"
torch.manual_seed(0)
for n in (1, 31, 32, 33, 64, 65, 4097, 26411, 100000, 1000000):
    g = torch.randn(n, 128, device=dev) * (1 + 100 * torch.rand(n, 1, device=dev)); x = torch.randn(n, 128, device=dev)
    dW = torch.empty(128, 128, device=dev)
    wsb = lib.aig_wgrad_ws_bytes(n, 128); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    call = lambda: _abi.check(lib.aig_wgrad(n, 128, _abi.ptr(g), _abi.ptr(x), _abi.ptr(dW), _abi.ptr(ws), wsb, _abi.stream()), "wgrad")
    for _ in range(3): call()
    torch.cuda.synchronize()
    lib.aig_kernel_timing(1 << 5)
    for _ in range(10): call()
    torch.cuda.synchronize()
    ms, cnt = C.c_double(), C.c_int64()
    lib.aig_kernel_timing_read(5, C.byref(ms), C.byref(cnt)); lib.aig_kernel_timing(0)
    ref = g.double().t() @ x.double()
    scale = (g.double().abs().t() @ x.double().abs())
    err = float(((dW.double() - ref).abs() / scale).max())
    print(f"f32_products=$f n={n}: {ms.value * 1e3 / max(cnt.value, 1):.1f} us ({cnt.value} launches), max |err| / sum|g x| = {err:.2e}")
PY
done
