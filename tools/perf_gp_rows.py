"""Tuning aid: GpRows.rebuild (aig_gp_rows + the rows' level plan) between a synth100k template and what the
untrained generator wires on it, repaired; run under rocprofv3 --kernel-trace --stats for the kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.constraints import enforce_set
from gan_amd.data import Data
from gan_amd.discriminator import DynLevelPlan, GpRows
from gan_amd.generator import AIGGenerator
from gan_amd.synth import synth_aig
dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
n = g.x.size(0)
G = AIGGenerator(candidate_k=64).to(dev).eval()
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
fake = enforce_set(G.forward_set([tpl], torch.randn(1, 128, device=dev)))
x_fake = fake.x.reshape(n, 2).contiguous()
depth = tpl.node_depth.to(torch.int64).contiguous()
L = int(depth.max().item())
plan = DynLevelPlan(n, L, dev)
plan.rebuild(depth)
rows = GpRows(n, L, dev)
for _ in range(3):
    rows.rebuild(tpl.x, x_fake, plan)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    rows.rebuild(tpl.x, x_fake, plan)
b.record(); torch.cuda.synchronize()
print("GpRows.rebuild: %.3f ms per call, rows %d, levels %d" % (a.elapsed_time(b) / 10, int(rows.n_rows.item()), L))
