"""Where a training step of the bench workload spends its time, with and without the second stream (tuning aid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.train import Trainer, build_parser
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig

dev = torch.device("cuda")
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
n = g.x.size(0)
real = Batch.from_data_list([pad_data(g, n, 2 * n)]).to(dev); real.sample_ids = [0]
for extra in ([], ["--no-sets"], ["--no-overlap"]):
    args = build_parser().parse_args(["--dataset-dir", "x", "--out-dir", "/tmp/o", "--candidate-k", "64", "--save-every", "1000000000"] + extra)
    tr = Trainer(args, n, 2 * n, dev)
    for _ in range(4): tr.step(real)
    torch.cuda.synchronize()
    acc = {}
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        tr.marks = []
        tr.step(real)
        torch.cuda.synchronize()
        m = tr.marks
        for (la, ea, ha), (lb, eb, hb) in zip(m, m[1:]):
            acc[lb] = acc.get(lb, 0.0) + ea.elapsed_time(eb)
            acc[lb + " (host)"] = acc.get(lb + " (host)", 0.0) + (hb - ha) * 1e3
    wall = (time.perf_counter() - t0) / reps * 1e3
    print(extra or "overlap", {k: round(v / reps, 2) for k, v in acc.items()}, "wall/step incl. sync", round(wall, 2))
