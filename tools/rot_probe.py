"""Tuning aid: per-minibatch step time and edge counts of the rotating bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gan_amd.train import Trainer, build_parser, count_edges

dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "synth100k"
groups, desc = bench.workload_minibatches(name, 0, 8)
batches, mn, me, top = bench.device_minibatches(groups, dev)
targs = build_parser().parse_args(["--dataset-dir", "synthetic", "--candidate-k", "64", "--save-every", "1000000000"])
targs.level_cap = top
tr = Trainer(targs, mn, me, dev)
tr.global_step = 1
for b in batches:
    tr.step(b)
torch.cuda.synchronize()
for i, b in enumerate(batches):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        tr.step(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    terms = next(iter(tr._captured.values()))
    print(i, f"{dt*1e3:.2f} ms", "edges_last", tr.edges_last_step, "real", int(terms.edges_real), "direct real", int(count_edges(b.edge_index)),
          "meta", terms.plan.meta.tolist(), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(16):
    tr.step(batches[i % 8])
torch.cuda.synchronize(); print("rotating", (time.perf_counter() - t0) / 16 * 1e3, "ms", flush=True)
