"""Tuning aid: the 100k-node training step with the caller's stream at normal and at high priority (the trainer's
other streams keep the default priority): ms per step over 40 steps each, twice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import faulthandler
import torch
faulthandler.dump_traceback_later(40, exit=True)
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser

dev = torch.device("cuda", 0)
graphs = [synth_aig(1000, 98000, 1000, seed=s, window=4096, orientation="dataset") for s in range(4)]
mn = max(g.x.size(0) for g in graphs); me = max(g.edge_index.size(1) for g in graphs)
batches = [Batch.from_data_list([pad_data(g, mn, me)]).to(dev) for g in graphs]
for i, b in enumerate(batches): b.sample_ids = [i]
args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "64", "--save-every", "1000000"])
args.level_cap = max(int(b.node_depth.max()) for b in batches)


def run(stream, label):
    print("start", label, flush=True)
    with torch.cuda.stream(stream):
        tr = Trainer(args, mn, me, dev); tr.global_step = 1
        for i in range(8): tr.step(batches[i % 4])
        gc.collect(); gc.freeze()
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40): tr.step(batches[i % 4])
            torch.cuda.synchronize()
            print(f"{label}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step", flush=True)
    del tr
    gc.unfreeze(); gc.collect()


run(torch.cuda.Stream(device=dev, priority=0), "normal priority")
run(torch.cuda.Stream(device=dev, priority=-1), "high priority")
run(torch.cuda.Stream(device=dev, priority=0), "normal priority")
