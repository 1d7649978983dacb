"""aig_linear128 at the bench's sizes: the six-product bf16 form (default) against the f32 products
(AIG_LINEAR128_F32=1), time per launch from the kernel's own events (tuning aid)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gan_amd import _abi

lib = _abi.lib()
dev = torch.device("cuda")
for n in (100000, 1000000):
    x = torch.randn(n, 128, device=dev)
    w = torch.randn(128, 128, device=dev) / 11
    y = torch.empty_like(x)
    for in_major in (0, 1):
        for _ in range(3):
            _abi.check(lib.aig_linear128(n, _abi.ptr(x), _abi.ptr(w), in_major, None, 0, _abi.ptr(y), _abi.stream()))
        torch.cuda.synchronize()
        lib.aig_kernel_timing(1 << 4)
        for _ in range(20):
            _abi.check(lib.aig_linear128(n, _abi.ptr(x), _abi.ptr(w), in_major, None, 0, _abi.ptr(y), _abi.stream()))
        torch.cuda.synchronize()
        ms, cnt = C.c_double(), C.c_int64()
        lib.aig_kernel_timing_read(4, C.byref(ms), C.byref(cnt))
        lib.aig_kernel_timing(0)
        us = ms.value * 1e3 / cnt.value
        print(f"n={n} in_major={in_major} f32_products={os.environ.get('AIG_LINEAR128_F32', '0')}: {us:.1f} us, "
              f"{2 * n * 512 / us / 1e6:.2f} TB/s, {2.0 * 128 * 128 * n / us / 1e6:.1f} TFLOP/s")

# the projection that also hands over the attention layer's node terms (aig_linear128_node_terms), in turns with the
# plain one on the same buffers
def timed(call):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.aig_kernel_timing(1 << 4)
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    ms, cnt = C.c_double(), C.c_int64()
    lib.aig_kernel_timing_read(4, C.byref(ms), C.byref(cnt))
    lib.aig_kernel_timing(0)
    return ms.value * 1e3 / cnt.value


for n in (100000, 1000000):
    x = torch.randn(n, 128, device=dev)
    w = torch.randn(128, 128, device=dev) / 11
    y = torch.empty_like(x)
    vec = [torch.randn(128, device=dev) for _ in range(4)]
    a_s, a_d, ce = torch.empty(n, 4, device=dev), torch.empty(n, 4, device=dev), torch.empty(4, device=dev)
    terms = lambda: _abi.check(lib.aig_linear128_node_terms(n, _abi.ptr(x), _abi.ptr(w), 0, _abi.ptr(vec[0]), _abi.ptr(vec[1]),
                                                            _abi.ptr(vec[2]), _abi.ptr(vec[3]), _abi.ptr(y), _abi.ptr(a_s),
                                                            _abi.ptr(a_d), _abi.ptr(ce), _abi.stream()))
    plain = lambda: _abi.check(lib.aig_linear128(n, _abi.ptr(x), _abi.ptr(w), 0, None, 0, _abi.ptr(y), _abi.stream()))
    alpha = lambda: _abi.check(lib.aig_gat_node_alpha(n, _abi.ptr(y), _abi.ptr(vec[0]), _abi.ptr(vec[1]), _abi.ptr(vec[2]),
                                                      _abi.ptr(vec[3]), _abi.ptr(a_s), _abi.ptr(a_d), _abi.ptr(ce), _abi.stream()))
    print(f"n={n} in turns: plain {timed(plain):.1f} us, with node terms {timed(terms):.1f} us, plain {timed(plain):.1f} us, "
          f"with node terms {timed(terms):.1f} us")
