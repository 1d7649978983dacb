"""What one small all-reduce through torch.distributed (RCCL) costs the host and the GPU on this box.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 tools/rccl_probe.py

(1) host time of the call while the GPU is busy with ~20 ms of queued work: a call that returns in microseconds
    lets the host enqueue ahead; one that takes as long as the queued work waits for the GPU;
(2) GPU time of the collective alone (events on the calling stream);
(3) the same with the collective issued from a side stream, and with async_op=True."""
import json
import os
import time

import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    dist.init_process_group(backend="nccl")
    flat = torch.zeros(59138, device=dev)
    a = torch.randn(4096, 4096, device=dev)

    def busy(n=12):
        x = a
        for _ in range(n):
            x = x @ a
            x = x / x.abs().max()
        return x

    for _ in range(3):
        dist.all_reduce(flat)
    busy(2)
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    busy()
    torch.cuda.synchronize()
    busy_ms = (time.perf_counter() - t0) * 1e3

    out = {"busy_ms": round(busy_ms, 2)}

    def probe(name, call, finish=None):
        host, total = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            busy()
            t1 = time.perf_counter()
            h = call()
            t2 = time.perf_counter()
            if finish is not None:
                finish(h)
            busy(2)
            t3 = time.perf_counter()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            host.append((t2 - t1) * 1e3)
            total.append((t4 - t0) * 1e3)
        out[name] = {"host_ms_of_call_min": round(min(host), 3), "host_ms_of_call_max": round(max(host), 3),
                     "enqueue_busy_ms": round((t1 - t0) * 1e3, 3), "wall_ms_min": round(min(total), 2)}

    probe("all_reduce_sync_op", lambda: dist.all_reduce(flat))
    probe("all_reduce_async_op", lambda: dist.all_reduce(flat, async_op=True), lambda w: w.wait())
    side = torch.cuda.Stream(device=dev)

    def on_side():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_reduce(flat)
        torch.cuda.current_stream().wait_stream(side)
    probe("all_reduce_on_side_stream", on_side)
    probe("no_collective", lambda: None)

    # GPU time of the collective alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0.record()
        dist.all_reduce(flat)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    out["gpu_ms_alone"] = {"min": round(min(ts), 4), "max": round(max(ts), 4)}
    t0 = time.perf_counter()
    for _ in range(100):
        dist.all_reduce(flat)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out["back_to_back_100"] = {"host_ms_per_call": round((t1 - t0) * 10, 4), "wall_ms_per_call": round((t2 - t0) * 10, 4)}
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
