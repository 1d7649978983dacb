"""Per-entry-point GPU time of bulk sampling on the 100k-node workload (blocks of 10 draws of one template as one
set of graphs, as gan_amd.sample runs it), from event brackets around the library's entry points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd import _abi
from gan_amd.data import Data
from gan_amd.generator import AIGGenerator
from gan_amd.sample import sample_sets, block_size
from gan_amd.synth import synth_aig

dev = torch.device("cuda", 0)
g = synth_aig(1000, int(os.environ.get("N_AND", "98000")), 1000, seed=0, window=4096, orientation="dataset")
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
G = AIGGenerator(candidate_k=64).to(dev)
gen = torch.Generator(device=dev).manual_seed(1)
S = block_size(int(tpl.x.size(0)), 10)
print("block", S)
def run(iters):
    for _ in sample_sets(G, tpl, (torch.randn(S, 128, device=dev, generator=gen) for _ in range(iters))):
        pass
run(2)
torch.cuda.synchronize()
_abi.timeline.reset(); _abi.timeline.only = None; _abi.timeline.enabled = True
import time
t0 = time.perf_counter()
iters = 6
run(iters)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
_abi.timeline.enabled = False
tot = _abi.timeline.totals_ms()
for k, (ms, calls) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:32s} {ms / iters:8.3f} ms per block  ({calls // iters} calls)")
print("sum", sum(v[0] for v in tot.values()) / iters, "wall per block (with brackets)", wall / iters * 1e3, "nodes/s", iters * S * tpl.x.size(0) / wall)
