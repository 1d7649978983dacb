"""Tuning aid: the bulk workload (c7552-sized template, k=256, blocks of 64) with 1, 2 and 3 blocks in flight:
wall time per block and the host's enqueue time per block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.data import Data
from gan_amd.generator import AIGGenerator
from gan_amd.sample import sample_sets
from gan_amd.synth import ISCAS85_SHAPES, synth_aig
name, n_pi, n_po, n_and = ISCAS85_SHAPES[-1]
g = synth_aig(n_pi, n_and, n_po, seed=0, window=max(8, n_and // 8 + n_pi), orientation="dataset")
dev = torch.device("cuda")
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
G = AIGGenerator(candidate_k=256, score_dtype=dt).to(dev).eval()
S, steps = 64, 20
zs = [torch.randn(S, 128, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(steps)]
for fl in (1, 2, 3, 1, 2):
    for _ in sample_sets(G, tpl, zs[:4], in_flight=fl):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    it = sample_sets(G, tpl, zs, in_flight=fl)
    while True:
        h0 = time.perf_counter()
        try:
            next(it)
        except StopIteration:
            break
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{dt} in_flight {fl}: {el / steps * 1e3:.3f} ms/block, in next(): {host / steps * 1e3:.3f} ms/block", flush=True)
from gan_amd.sample import sample_bulk
for fl in (1, 3):
    for _ in sample_bulk(G, tpl, zs[:4], in_flight=fl):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = 0
    for d in sample_bulk(G, tpl, zs, in_flight=fl):
        e += int(d.edge_index.size(1))
    torch.cuda.synchronize()
    print(f"{dt} graph by graph, in_flight {fl}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/block", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for d in sample_bulk(G, tpl, zs, in_flight=3):
    pass
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
