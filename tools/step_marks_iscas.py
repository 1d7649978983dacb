"""Phases of the training step on the ISCAS85-like minibatch: GPU time between marks and host time (tuning aid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gan_amd.train import Trainer, build_parser
from gan_amd.data import Batch
from gan_amd.dataset import pad_data

dev = torch.device("cuda")
graphs = bench.workload_minibatches("iscas85", 0, 1)[0][0]
mn = max(g.x.size(0) for g in graphs); me = max(g.edge_index.size(1) for g in graphs)
real = Batch.from_data_list([pad_data(g, mn, me) for g in graphs]).to(dev); real.sample_ids = list(range(len(graphs)))
for extra in ([], ["--no-sets"]):
    args = build_parser().parse_args(["--dataset-dir", "x", "--out-dir", "/tmp/o", "--candidate-k", "64", "--save-every", "1000000000"] + extra)
    tr = Trainer(args, mn, me, dev)
    for _ in range(4): tr.step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): tr.step(real)
    torch.cuda.synchronize()
    print(extra, "ms/step back to back", round((time.perf_counter() - t0) * 100, 2))
    acc = {}
    reps = 10
    for _ in range(reps):
        tr.marks = []
        tr.step(real)
        torch.cuda.synchronize()
        m = tr.marks
        for (la, ea, ha), (lb, eb, hb) in zip(m, m[1:]):
            acc[lb] = acc.get(lb, 0.0) + ea.elapsed_time(eb)
            acc[lb + " (host)"] = acc.get(lb + " (host)", 0.0) + (hb - ha) * 1e3
    tr.marks = None
    print(extra, {k: round(v / reps, 2) for k, v in acc.items()})
