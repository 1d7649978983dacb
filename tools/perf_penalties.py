"""Tuning aid: aig_penalties_batch on what the untrained generator wires for a synth100k template (10 calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.constraints import penalties_set
from gan_amd.data import Data
from gan_amd.generator import AIGGenerator
from gan_amd.synth import synth_aig
dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev).eval()
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
gs = G.forward_set([tpl], torch.randn(1, 128, device=dev))
for _ in range(3):
    penalties_set(gs)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    out = penalties_set(gs)
b.record(); torch.cuda.synchronize()
print("penalties_set: %.3f ms per call" % (a.elapsed_time(b) / 10), out.cpu().tolist())
