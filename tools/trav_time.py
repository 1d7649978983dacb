"""Timing aid: the level-by-level walks on the bench graphs - the repair's breadth-first levels (real
and generated graph) and the resync's longest-path levels (repaired generated graph), median of 20."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.synth import synth_aig
from gan_amd.generator import AIGGenerator
from gan_amd.constraints import enforce_aig_constraints, resync_attrs, aig_constraint_penalties
from gan_amd.data import Data

dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
raw = G(tpl, torch.randn(128, device=dev))
real = Data(x=g.x.to(dev), edge_index=g.edge_index.to(dev), edge_attr=g.edge_attr.to(dev), node_depth=g.node_depth.to(dev))
fixed = enforce_aig_constraints(raw.clone())


def timed(name, fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"{name:34s} median {statistics.median(ts):8.1f} us  min {min(ts):8.1f} us", flush=True)


timed("repair of a generated graph", lambda: enforce_aig_constraints(raw.clone()))
timed("repair of the dataset graph", lambda: enforce_aig_constraints(real.clone()))
timed("resync of a repaired graph", lambda: resync_attrs(fixed.clone()))
timed("resync of the dataset graph", lambda: resync_attrs(real.clone()))
timed("penalties of a generated graph", lambda: aig_constraint_penalties(raw))
