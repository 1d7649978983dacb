"""Two data-parallel ranks of the real ``Trainer.step`` (not collected by pytest: started by tests/conftest.py
before the test session touches the GPU, checked by tests/test_parallel_gpu.py).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 ... tests/dp_rehearsal.py OUT.json

The ranks may share one GPU (``AIG_DIST_BACKEND=gloo``; RCCL wants a device per rank).  Every rank writes what it
saw; rank 0 gathers and writes OUT.json:

  grad_mean_err     first critic iteration (all trainers still hold the initial weights): the reduced gradient
                    of the 2-rank trainer against the mean of the gradients two single-rank trainers computed
                    for the two minibatches, max error relative to the largest element
  weights_equal     after two full steps every rank holds bit-identical critic weights
  raised / after_failure_equal / moved_after_failure
                    third step: rank 1's step raises before its first gradient; rank 0 must finish (the failed
                    rank still enters every all-reduce), the weights must stay identical on both ranks and move
                    by rank 0's gradient alone
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

from gan_amd import parallel
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser


def flat(params):
    return torch.cat([p.detach().reshape(-1).float().cpu() for p in params])


WATCH = {"trainer": None, "phase": "start"}


def start_watchdog(rank: int, limit: float = float(os.environ.get("AIG_REHEARSAL_LIMIT", "40"))) -> None:
    """The rehearsal takes a few seconds.  Past ``limit`` this rank says where it is - phase, which of the last
    step's probe events have completed, whether its two streams are idle, every thread's stack - and ends."""
    import faulthandler
    import threading
    import time

    def watch():
        time.sleep(limit)
        tr = WATCH["trainer"]
        out = {"rank": rank, "phase": WATCH["phase"]}
        try:
            if tr is not None:
                out.update(global_step=tr.global_step, captures=tr.captures,
                           main_idle=torch.cuda.default_stream(tr.device).query(), side_idle=tr.side_stream.query(),
                           probes={k: v.query() for k, v in dict(tr.probes or {}).items()})
        except Exception as e:
            out["error"] = repr(e)
        sys.stderr.write("\n[rehearsal watchdog] " + json.dumps(out) + "\n")
        sys.stderr.flush()
        faulthandler.dump_traceback(all_threads=True)
        os._exit(3)

    threading.Thread(target=watch, daemon=True).start()


def watched_step(tr, batch, phase):
    WATCH["trainer"], WATCH["phase"] = tr, phase
    if os.environ.get("AIG_TEST_PROBES") == "1":      # (off: the step runs as in production, without marker packets)
        tr.probes = {}
    out = tr.step(batch)
    WATCH["phase"] = phase + " (enqueued)"
    return out


def main():
    out_path = sys.argv[1]
    rank, local_rank, world = parallel.init_distributed()
    start_watchdog(rank)
    assert world == 2, "the rehearsal is written for two ranks"
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    # AIG_REHEARSAL_NODES: graphs of about that many nodes instead (100000 = what `bench.py --gpus 2` steps over)
    big = int(os.environ.get("AIG_REHEARSAL_NODES", "0"))
    if big:
        graphs = [synth_aig(big // 100, big - big // 50 - 1000 * i, big // 100, seed=50 + i, window=4096, orientation="dataset")
                  for i in range(2)]
    else:
        graphs = [synth_aig(6 + i, 60 + 20 * i, 4, seed=50 + i, window=16, orientation="dataset") for i in range(2)]
    mn = max(g.x.size(0) for g in graphs)
    me = max(g.edge_index.size(1) for g in graphs)
    batches = [Batch.from_data_list([pad_data(g, mn, me)]).to(dev) for g in graphs]
    for i, b in enumerate(batches):
        b.sample_ids = [i]
    args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "64" if big else "16", "--save-every", "1000",
                                      "--n-critic", "3" if big else "2"])
    w = np.load(os.path.join(ROOT, "tests", "golden", "gen_weights.npz"))
    gen_state = {k: torch.from_numpy(w[k]) for k in w.files}
    torch.manual_seed(1234 + rank)                       # different initial critics: the broadcast must align them
    dp = Trainer(args, mn, me, dev, rank, world)
    dp.G.load_state_dict(gen_state)
    dp.global_step = 1
    init = {k: v.detach().clone() for k, v in dp.D.state_dict().items()}

    g = torch.Generator().manual_seed(5)
    zs = [torch.randn(128, generator=g) for _ in range(64)]
    alphas = [torch.rand(1, 1, generator=g) for _ in range(64)]

    def draws(tr, who):
        zi, ai = iter(zs[who * 16:]), iter(alphas[who * 16:])    # the two minibatches get their own draws
        tr.draw_z = lambda: next(zi).to(dev)
        tr.draw_alpha = lambda: next(ai).to(dev)

    def record_first_gradient(tr, store):
        inner = tr.reducer.reduce

        def reduce(ok=True):
            res = inner(ok)
            if "g" not in store:
                store["g"] = flat([p.grad for p in tr._d_params])
            return res
        tr.reducer.reduce = reduce

    # single-rank trainers for both minibatches, on every rank (their reducer does nothing: world 1)
    singles = []
    for who in range(2):
        single = Trainer(args, mn, me, dev, 0, 1)
        single.reducer.world = 1
        single.G.load_state_dict(gen_state)
        single.D.load_state_dict(init)
        single.global_step = 1
        draws(single, who)
        store = {}
        record_first_gradient(single, store)
        watched_step(single, batches[who], f"single trainer {who}")
        singles.append(store["g"])
    mean_grad = (singles[0] + singles[1]) / 2

    draws(dp, rank)
    seen = {}
    record_first_gradient(dp, seen)
    watched_step(dp, batches[rank], "first step")
    grad_err = float((seen["g"] - mean_grad).abs().max() / mean_grad.abs().max())
    watched_step(dp, batches[rank], "second step")
    WATCH["phase"] = "gather after second step"
    weights = flat(dp.D.parameters())                   # (host tensors: the group is gloo, see gan_amd/parallel.py)
    both = [torch.empty_like(weights) for _ in range(world)]
    dist.all_gather(both, weights)
    weights_equal = bool(torch.equal(both[0], both[1]))
    moved = float((both[0] - flat(init.values())).abs().max())

    # third step: rank 1 fails before it has a gradient
    if rank == 1:
        def boom():
            raise RuntimeError("injected failure on rank 1")
        dp.draw_z = boom
    before = both[0].clone()
    raised = None
    try:
        watched_step(dp, batches[rank], "third step")
    except RuntimeError as e:
        raised = str(e)
    WATCH["phase"] = "gather after third step"
    weights = flat(dp.D.parameters())
    dist.all_gather(both, weights)
    after_failure_equal = bool(torch.equal(both[0], both[1]))
    moved_after_failure = float((both[0] - before).abs().max())     # the surviving rank's gradient still counts
    mine = {"rank": rank, "grad_mean_err": grad_err, "weights_equal": weights_equal, "moved": moved, "raised": raised,
            "after_failure_equal": after_failure_equal, "moved_after_failure": moved_after_failure,
            "finite": bool(torch.isfinite(both[0]).all())}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        with open(out_path + ".tmp", "w") as f:
            json.dump(gathered, f)
        os.replace(out_path + ".tmp", out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
