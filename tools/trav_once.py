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
