"""When each probe event of the training step completes on the GPU, relative to the step's first event (ms), for the
100k-node workload: median over a few steps.  Shows what the first stream waits for at the end of the step."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import torch
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser

dev = torch.device("cuda", 0)
if os.environ.get("WORKLOAD") == "iscas85":      # minibatches of all eleven ISCAS85-sized graphs, as bench.py's side workload
    import bench
    groups, _ = bench.workload_minibatches("iscas85", 0, 4)
    batches, mn, me, _top = bench.device_minibatches(groups, dev)
else:
    n_and = int(os.environ.get("N_AND", "98000"))
    graphs = [synth_aig(1000, n_and, 1000, seed=s, window=4096, orientation="dataset") for s in range(4)]
    mn = max(g.x.size(0) for g in graphs); me = max(g.edge_index.size(1) for g in graphs)
    batches = [Batch.from_data_list([pad_data(g, mn, me)]).to(dev) for g in graphs]
    for i, b in enumerate(batches): b.sample_ids = [i]
args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "64", "--save-every", "1000000"] + sys.argv[1:])
args.level_cap = max(int(b.node_depth.max()) for b in batches)
tr = Trainer(args, mn, me, dev); tr.global_step = 1
for i in range(6): tr.step(batches[i % 4])
gc.collect(); gc.freeze()
rows = {}
for i in range(12):
    tr.probes = {}
    tr.step(batches[i % 4])
    torch.cuda.synchronize()
    t0 = tr.probes["main: step start"]
    for k, ev in tr.probes.items():
        rows.setdefault(k, []).append(t0.elapsed_time(ev))
for k, v in sorted(rows.items(), key=lambda kv: statistics.median(kv[1])):
    print(f"{statistics.median(v):8.3f} ms  {k}")
