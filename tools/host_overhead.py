import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.discriminator import _linear128, _WGrad
dev = "cuda"
for n in (64, 100000):
    x = torch.randn(n, 128, device=dev); w = torch.randn(128, 128, device=dev)
    for name, fn in (("linear128", lambda: _linear128(x, w, False)), ("matmul", lambda: x @ w.t()), ("wgrad", lambda: _WGrad.apply(x, x))):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"n={n} {name}: host {1e6*(t1-t0)/300:.1f} us/call, total {1e6*(t2-t0)/300:.1f} us/call")
