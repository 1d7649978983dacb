"""Tuning aid: where the host time of a bulk_k256 block goes (cProfile over 32 blocks of 64 samples, three in flight),
the loop bench.py --workload bulk_k256 times."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.data import Data
from gan_amd.generator import AIGGenerator
from gan_amd.sample import sample_bulk
from gan_amd.synth import ISCAS85_SHAPES, synth_aig
dev = "cuda"
name, n_pi, n_po, n_and = ISCAS85_SHAPES[-1]
g = synth_aig(n_pi, n_and, n_po, seed=0, window=max(8, n_and // 8 + n_pi), orientation="dataset")
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
torch.manual_seed(123)
G = AIGGenerator(candidate_k=256, score_dtype="bf16").to(dev).eval()
blocks = lambda k: [torch.randn(64, 128, device=dev) for _ in range(k)]
def run(bl):
    e = 0
    for d in sample_bulk(G, tpl, bl):
        e += int(d.edge_index.size(1))
    return e
run(blocks(4)); torch.cuda.synchronize()
bl = blocks(32); torch.cuda.synchronize()
t0 = time.perf_counter(); run(bl); torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 32 * 1e3:.3f} ms per block without the profiler")
bl = blocks(32); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(bl); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
