"""Timing aid: where the host time of bulk sampling goes (cProfile over 64 samples, four in flight)."""
import cProfile, os, pstats, sys, time, torch
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
draw = lambda: torch.randn(128, device=dev)
for _ in sample_many(G, ((tpl, draw) for _ in range(8))): pass
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in sample_many(G, ((tpl, draw) for _ in range(64))): pass
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 64 * 1e3:.3f} ms per sample without the profiler")
pr = cProfile.Profile(); pr.enable()
for _ in sample_many(G, ((tpl, draw) for _ in range(64))): pass
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
