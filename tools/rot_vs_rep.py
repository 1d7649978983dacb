"""Tuning aid: step time over rotating minibatches vs one repeated minibatch (bench workload)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gan_amd.train import Trainer, build_parser
dev = torch.device("cuda")
groups, desc = bench.workload_minibatches("synth100k", 0, 8)
batches, mn, me, top = bench.device_minibatches(groups, dev)
targs = build_parser().parse_args(["--dataset-dir", "synthetic", "--candidate-k", "64", "--save-every", "1000000000"] + sys.argv[1:])
targs.level_cap = top
tr = Trainer(targs, mn, me, dev)
tr.global_step = 1
def run(count, only=None, start=0):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(count):
        tr.step(batches[(start + i) % 8] if only is None else only)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / count
run(8); run(3)
print("rotating", round(run(16), 2))
print("rotating", round(run(16, start=5), 2))
def per_step(only, count=16):
    out = []
    for i in range(count):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.step(only); torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) * 1e3, 1))
    return out
def stats():
    ms = torch.cuda.memory_stats()
    return tr.captures, ms["num_device_alloc"], ms["num_device_free"], ms["num_alloc_retries"], ms["reserved_bytes.all.current"] >> 20
print("before", stats())
from gan_amd import discriminator as dm, _abi
orig = dm._CriticHead.backward
log = []
def timed_bwd(ctx, g):
    t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    out = orig(ctx, g)
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    log.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((t3 - t2) * 1e3, 2)))
    return out
dm._CriticHead.backward = staticmethod(timed_bwd)
tr.step(batches[0]); torch.cuda.synchronize()
print("head backward: (wait for the GPU before, host time of the call, GPU time after)", log)
dm._CriticHead.backward = orig
print("repeated b0 per step (synchronised)", per_step(batches[0]))
print("repeated b0", round(run(16, only=batches[0]), 2))
print("repeated b1", round(run(16, only=batches[1]), 2))
print("repeated b0", round(run(16, only=batches[0]), 2))
print("rotating", round(run(16), 2))
