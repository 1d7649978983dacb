"""Tuning aid: is the 100k-node training step bound by the host or by the GPU?  The step has one host read (the
level-cap check of the generated side); the time the host spends blocked in it is the GPU's backlog at that point.
Prints ms/step, the blocked time per step and the host's own time (wall - blocked)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser
from gan_amd import discriminator as dmod

dev = torch.device("cuda", 0)
graphs = [synth_aig(1000, 98000, 1000, seed=s, window=4096, orientation="dataset") for s in range(8)]
mn = max(g.x.size(0) for g in graphs); me = max(g.edge_index.size(1) for g in graphs)
batches = [Batch.from_data_list([pad_data(g, mn, me)]).to(dev) for g in graphs]
for i, b in enumerate(batches): b.sample_ids = [i]
args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "64", "--save-every", "1000000000"] + sys.argv[1:])
args.level_cap = max(int(b.node_depth.max()) for b in batches)
tr = Trainer(args, mn, me, dev)
tr.global_step = 1
blocked = [0.0]
orig = dmod.DynLevelPlan.overflowed
def timed(self):
    t0 = time.perf_counter(); r = orig(self); blocked[0] += time.perf_counter() - t0; return r
dmod.DynLevelPlan.overflowed = timed
for i in range(16): tr.step(batches[i % 8])
gc.collect(); gc.freeze()
torch.cuda.synchronize()
blocked[0] = 0.0
K = 40
t0 = time.perf_counter()
for i in range(K): tr.step(batches[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"ms/step {(t2 - t0) / K * 1e3:.3f}  host loop {(t1 - t0) / K * 1e3:.3f}  blocked in the level-cap read {blocked[0] / K * 1e3:.3f}  "
      f"host's own time {(t1 - t0 - blocked[0]) / K * 1e3:.3f}  final drain {(t2 - t1) * 1e3:.3f} ms")
