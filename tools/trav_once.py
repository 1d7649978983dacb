import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.synth import synth_aig
from gan_amd.generator import AIGGenerator
from gan_amd.constraints import enforce_aig_constraints, aig_constraint_penalties
from gan_amd.data import Data
dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
raw = G(Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev)), torch.randn(128, device=dev))
torch.cuda.synchronize(); print("enforce:", flush=True)
f = enforce_aig_constraints(raw); torch.cuda.synchronize(); print("penalties:", flush=True)
p = aig_constraint_penalties(raw); torch.cuda.synchronize()
from gan_amd.constraints import resync_attrs
print("resync:", flush=True)
r = resync_attrs(f); torch.cuda.synchronize()
ei = f.edge_index
n = f.x.shape[0]
outdeg = torch.bincount(ei[0], minlength=n)
indeg = torch.bincount(ei[1], minlength=n)
dep = r.node_depth.long()
L = int(dep.max()) + 1
for name, deg in (("out", outdeg), ("in", indeg)):
    mx = torch.zeros(L, dtype=torch.long, device=dev).scatter_reduce(0, dep, deg, "amax")
    print(name, "deg hist>", [(k, int((deg > k).sum())) for k in (2, 6, 14, 30, 62)],
          "levels with max>", [(k, int((mx > k).sum())) for k in (2, 6, 14, 30, 62)], "of", L, flush=True)
