"""Time of aig_sort_pairs per call (events around 50 calls) at the sizes the training step sorts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _abi
lib = _abi.lib()
dev = torch.device("cuda")
for n, kb, bits in [(20000, 4, 17), (100000, 4, 8), (200000, 4, 17), (200000, 4, 18), (400000, 4, 19), (800000, 4, 20),
                    (200000, 8, 34), (400000, 8, 36), (2000000, 4, 21)]:
    dt = torch.int64 if kb == 8 else torch.int32
    kin = torch.randint(0, 1 << min(bits, 30), (n,), device=dev).to(dt)
    vin = torch.arange(n, dtype=torch.int32, device=dev)
    kout, vout = torch.empty_like(kin), torch.empty_like(vin)
    ws = _abi.workspace(lib.aig_sort_pairs_ws_bytes(n, kb, bits), dev)
    def call():
        _abi.check(lib.aig_sort_pairs(n, kb, bits, _abi.ptr(kin), _abi.ptr(kout), _abi.ptr(vin), _abi.ptr(vout), _abi.ptr(ws),
                                      ws.numel(), _abi.stream()), "aig_sort_pairs")
    for _ in range(5): call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): call()
    b.record(); torch.cuda.synchronize()
    print(f"n={n:8d} key={kb}B bits={bits:2d}  {a.elapsed_time(b) * 1e3 / 50:8.1f} us/sort")
