"""Timing aid: the generator's node projection chain (k_node_mlp3 + k_node_proj4) at 100k nodes."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _abi
from gan_amd.generator import AIGGenerator
from gan_amd.synth import synth_aig
from gan_amd.data import Data
dev = "cuda"
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
z = torch.randn(128, device=dev)
for _ in range(3): G(tpl, z)
_abi.timeline.reset(); _abi.timeline.enabled = True
for _ in range(20): G(tpl, z)
_abi.timeline.enabled = False; torch.cuda.synchronize()
for k, v in sorted(_abi.timeline.events.items()):
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in v)
    print(f"{k:20s} median {statistics.median(t):8.1f} us  min {t[0]:8.1f} us")
