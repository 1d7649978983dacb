import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _abi
dev = "cuda"
lib = _abi.lib()
w = torch.randn(128, 128, device=dev)
for n in (26411, 100000, 1000000):
    x = torch.randn(n, 128, device=dev); y = torch.empty_like(x)
    for i in range(30):
        _abi.check(lib.aig_linear128(n, _abi.ptr(x), _abi.ptr(w), i & 1, None, 2 * (i & 1), _abi.ptr(y), _abi.stream()), "lin")
    torch.cuda.synchronize()
