"""Fan-out distribution of the generator's raw and repaired graphs on the bench workload (tuning aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.synth import synth_aig
from gan_amd.generator import AIGGenerator
from gan_amd.constraints import enforce_aig_constraints
from gan_amd.data import Data
dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
raw = G(Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev)), torch.randn(128, device=dev))
def stats(name, d):
    n = d.x.shape[0]
    out = torch.bincount(d.edge_index[0], minlength=n)
    top = torch.topk(out, 8).values.tolist()
    print(name, "E", d.edge_index.shape[1], "top fan-outs", top, "nodes with fan-out >32:", int((out > 32).sum()), ">1024:", int((out > 1024).sum()),
          "edges on nodes >32:", int(out[out > 32].sum()))
stats("raw", raw)
stats("real", Data(x=g.x.to(dev), edge_index=g.edge_index.to(dev)))
stats("repaired", enforce_aig_constraints(raw))
