"""Tuning aid: per-step wall time of a long run of training steps; prints the steps that took far longer."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gan_amd.train import Trainer, build_parser
dev = torch.device("cuda")
t_start = time.perf_counter()
groups, desc = bench.workload_minibatches("synth100k", 0, 8)
batches, mn, me, top = bench.device_minibatches(groups, dev)
targs = build_parser().parse_args(["--dataset-dir", "synthetic", "--candidate-k", "64", "--save-every", "1000000000"])
targs.level_cap = top
tr = Trainer(targs, mn, me, dev)
tr.global_step = 1
mode = sys.argv[1] if len(sys.argv) > 1 else "rotating"
import gc
if len(sys.argv) > 2 and sys.argv[2] == "freeze":
    gc.collect(); gc.freeze()
print("gc counts", gc.get_count(), "objects tracked", len(gc.get_objects()))
gc.callbacks.append(lambda phase, info: print(f"  gc {phase} gen {info['generation']} at step {len(times)}") if info["generation"] == 2 else None)
times = []
for i in range(160):
    b = batches[i % 8] if (mode == "rotating" or i < 30) else batches[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(b); torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    if times[-1] > 20 and i > 8:
        print(f"step {i}: {times[-1]:.1f} ms at {time.perf_counter() - t_start:.2f} s since start, captures {tr.captures}")
print(mode, "median", round(sorted(times)[len(times) // 2], 2), "max after set-up", round(max(times[9:]), 1))
