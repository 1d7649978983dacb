"""Data-parallel plumbing on two gloo ranks (CPU): flat gradient all-reduce with the validity flag,
parameter broadcast, minibatch and sample sharding.  The model here is a stand-in: the helpers
are model-agnostic and the GPU kernels are not involved."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gan_amd import parallel
from gan_amd.data import Data, GraphLoader


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    r, _, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # ranks start from different weights
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1))
    parallel.broadcast_parameters([net])
    flat0 = torch.cat([p.data.reshape(-1) for p in net.parameters()])
    gathered = [torch.zeros_like(flat0) for _ in range(world)]
    dist.all_gather(gathered, flat0)
    assert all(torch.equal(g, gathered[0]) for g in gathered), "broadcast left ranks with different weights"

    red = parallel.FlatGradAllReduce(net.parameters())
    x = torch.randn(4, 5, generator=torch.Generator().manual_seed(rank))   # rank-local minibatch
    net.zero_grad()
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    assert red.reduce(True) is True
    # reference: mean over ranks of the per-rank gradients, computed sequentially
    want = []
    for rr in range(world):
        xr = torch.randn(4, 5, generator=torch.Generator().manual_seed(rr))
        net2 = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1))
        net2.load_state_dict(net.state_dict())
        net2(xr).sum().backward()
        want.append([p.grad for p in net2.parameters()])
    for i, p in enumerate(net.parameters()):
        mean = sum(w_[i] for w_ in want) / world
        assert torch.allclose(p.grad, mean, atol=1e-6), "all-reduced gradient is not the rank mean"
    assert any(not torch.equal(a, p.grad) for a, p in zip(local, net.parameters()))

    # one rank reports a failed step: everybody returns (no hang) with the mean over the other ranks
    net.zero_grad()
    net(x).sum().backward()
    assert red.reduce(ok=(rank != 1)) is True
    for i, p in enumerate(net.parameters()):
        assert torch.allclose(p.grad, want[0][i], atol=1e-6), "gradient is not the mean over the valid ranks"
    # nobody has a gradient: zeros everywhere
    net.zero_grad()
    net(x).sum().backward()
    assert red.reduce(ok=False) is True
    assert all(float(p.grad.abs().max()) == 0.0 for p in net.parameters())

    # minibatch sharding: same shuffle everywhere, disjoint batches, equal step counts
    items = [Data(x=torch.full((1, 2), float(i))) for i in range(9)]
    mine = [b.sample_ids for b in GraphLoader(items, batch_size=2, seed=3, rank=rank, world_size=world)]
    all_ids = [None] * world
    dist.all_gather_object(all_ids, mine)
    assert len(all_ids[0]) == len(all_ids[1]) == 2
    flat = [i for per_rank in all_ids for b in per_rank for i in b]
    assert len(flat) == len(set(flat))
    assert parallel.shard_indices(7, rank, world) == list(range(rank, 7, world))
    with open(os.path.join(tmp, f"ok{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_single_process_is_a_no_op():
    net = torch.nn.Linear(3, 2)
    red = parallel.FlatGradAllReduce(net.parameters())
    net(torch.ones(1, 3)).sum().backward()
    g = net.weight.grad.clone()
    assert red.reduce(True) is True and torch.equal(net.weight.grad, g)
    assert parallel.env_world()[2] >= 1
