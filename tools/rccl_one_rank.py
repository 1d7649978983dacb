"""One-rank rehearsal of the data-parallel training step over RCCL itself (not gloo).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 tools/rccl_one_rank.py

RCCL wants a device per rank, so a one-GPU box can only hold a communicator of one rank.  That is still the real
library: its initialisation, its stream, its event hand-over with the training step's three streams and with the
replayed recordings.  The reducer is told it has partners (``reducer.world = 2``) so every critic iteration goes
through ``all_reduce`` on the flat gradient buffer; with one rank the sum is the rank's own gradient and the
validity count is 1, so the weights must come out bit-identical to a trainer that never calls the collective.
Prints one JSON line: equality of the weights after a few steps, and ms/step with and without the collective
(the difference is the host + launch cost of three all-reduce calls per step, not wire time)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    opts = None
    if os.environ.get("NCCL_HIGH_PRIO"):
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group(backend="nccl", rank=int(os.environ.get("RANK", "0")),
                            world_size=int(os.environ.get("WORLD_SIZE", "1")), pg_options=opts)
    assert dist.get_world_size() == 1
    n_and = int(os.environ.get("N_AND", "98000"))
    graphs = [synth_aig(1000, n_and, 1000, seed=s, window=4096, orientation="dataset") for s in range(4)]
    mn = max(g.x.size(0) for g in graphs)
    me = max(g.edge_index.size(1) for g in graphs)
    batches = [Batch.from_data_list([pad_data(g, mn, me)]).to(dev) for g in graphs]
    for i, b in enumerate(batches):
        b.sample_ids = [i]
    args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "64", "--save-every", "1000000"])

    def run(collective: bool, steps: int):
        torch.manual_seed(7)
        tr = Trainer(args, mn, me, dev, 0, 2 if collective else 1)
        tr.reducer.world = 2 if collective else 1
        tr.global_step = 1
        g = torch.Generator().manual_seed(11)
        tr.draw_z = lambda: torch.randn(128, generator=g).to(dev)
        tr.draw_alpha = lambda: torch.rand(1, 1, generator=g).to(dev)
        for i in range(4):
            tr.step(batches[i % 4])
        import gc
        gc.collect()
        gc.freeze()          # (a full collection over the long-lived set-up objects is a 100 ms stall otherwise)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(batches[i % 4])
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        w = torch.cat([p.detach().reshape(-1) for p in tr.D.parameters()]).clone()
        last = {k: float(v) for k, v in tr.last.items()}
        return ms, w, last

    steps = int(os.environ.get("STEPS", "12"))
    only = os.environ.get("ONLY")          # "plain" / "coll": one configuration per process (no cross-talk)
    if only:
        variant = os.environ.get("VARIANT", "")
        if variant == "no_allreduce":          # pack / scale / unpack only
            dist.all_reduce = lambda *a, **k: None
        elif variant == "allreduce_only":      # the collective alone, on the flat buffer as it is
            from gan_amd import parallel as _par

            def only_collective(self, ok=True):
                if self.world == 1:
                    return ok
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                return True
            _par.FlatGradAllReduce.reduce = only_collective
        ms, _, _ = run(only == "coll", steps)
        print(json.dumps({"mode": only, "ms_per_step": round(ms, 3)}))
        dist.destroy_process_group()
        return
    ms_coll0, _, _ = run(True, steps)
    ms_plain, w_plain, last_plain = run(False, steps)
    ms_coll, w_coll, last_coll = run(True, steps)
    ms_plain2, _, _ = run(False, steps)
    out = {"backend": dist.get_backend(), "nodes": mn, "steps": steps, "ms_per_step_no_collective": round(ms_plain, 3),
           "ms_per_step_three_allreduce": round(ms_coll, 3),
           "ms_per_step_in_run_order": [round(v, 3) for v in (ms_coll0, ms_plain, ms_coll, ms_plain2)], "weights_bit_equal": bool(torch.equal(w_plain, w_coll)),
           "max_weight_diff": float((w_plain - w_coll).abs().max()), "finite": bool(torch.isfinite(w_coll).all()),
           "scalars_equal": last_plain == last_coll}
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
