"""Timing aid: host time of one generator forward pass on an ISCAS85-sized template (launch-bound regime)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.generator import AIGGenerator
from gan_amd.synth import synth_aig
from gan_amd.data import Data
from gan_amd.constraints import enforce_begin, enforce_finish, aig_constraint_penalties

dev = "cuda"
g = synth_aig(200, 2000, 100, seed=0, window=256, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
z = torch.randn(128, device=dev)
for _ in range(20): G(tpl, z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): G(tpl, z)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"G.forward: {(t1 - t0) / 500 * 1e6:.1f} us of host time per call ({(t2 - t0) / 500 * 1e6:.1f} us with the GPU drained)")
for name, fn in (("enforce", lambda: enforce_finish(enforce_begin(G(tpl, z)))), ("penalties", lambda: aig_constraint_penalties(out))):
    out = G(tpl, z)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {(t1 - t0) / 200 * 1e6:.1f} us host per call ({(t2 - t0) / 200 * 1e6:.1f} us drained)")
pr = cProfile.Profile()
pr.enable()
for _ in range(300): G(tpl, z)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
