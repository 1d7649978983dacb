"""Does a small all-reduce (torch.distributed over RCCL) wait for an unrelated long, narrow kernel that runs on another
stream?  Main stream: ~3 ms of wide work, all_reduce, ~1 ms of wide work; a second stream meanwhile spins one
thread for ~6 ms.  Wall time of the main stream's sequence with / without the collective and with / without the
spinning kernel."""
import json, os, time
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
    side = torch.cuda.Stream(device=dev)
    def busy(n):
        x = a
        for _ in range(n):
            x = x @ a
            x = x / x.abs().max()
    for _ in range(3):
        dist.all_reduce(flat)
    busy(2); torch.cuda.synchronize()
    out = {}
    for spin in (False, True):
        for coll in (False, True):
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if spin:
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(int(6e-3 * 2.0e9))
                e0.record()
                busy(3)
                if coll:
                    dist.all_reduce(flat)
                busy(1)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out[f"spin={spin} collective={coll}"] = round(min(ts), 3)
    print(json.dumps(out))
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
