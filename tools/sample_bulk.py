"""Bulk sampling of the 100k-node template for a profiler run (tuning aid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.synth import synth_aig
from gan_amd.generator import AIGGenerator
from gan_amd.sample import sample_many
from gan_amd.data import Data
dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 32
draw = lambda: torch.randn(128, device=dev)
for _ in sample_many(G, ((tpl, draw) for _ in range(4))): pass
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in sample_many(G, ((tpl, draw) for _ in range(iters))): pass
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{iters} samples, {dt / iters * 1e3:.3f} ms per sample, {100000 * iters / dt / 1e6:.1f} M nodes/s")
