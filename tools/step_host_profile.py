"""Tuning aid: where the host's time goes in the training step on the ISCAS85-sized minibatch (the step there is
bound by the host: ~190 eager launches and the Python around them).  cProfile over 40 steps, top entries by own time
and by cumulative time."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import torch
import bench
from gan_amd.train import Trainer, build_parser

dev = torch.device("cuda", 0)
wl = os.environ.get("WORKLOAD", "iscas85")
groups, _ = bench.workload_minibatches(wl, 0, 4)
batches, mn, me, top = bench.device_minibatches(groups, dev)
args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "64", "--save-every", "1000000"])
args.level_cap = top
tr = Trainer(args, mn, me, dev); tr.global_step = 1
for i in range(8): tr.step(batches[i % len(batches)])
torch.cuda.synchronize()
gc.collect(); gc.freeze()
pr = cProfile.Profile()
pr.enable()
for i in range(40): tr.step(batches[i % len(batches)])
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:44]))
