"""Data-parallel plumbing: one process per GPU, RCCL over xGMI through ``torch.distributed``.

The reference is single-process (train.py:191-360).  Here graph minibatches are sharded over
the ranks of one node and the only exchange is the critic's gradient: 59,137 fp32 values
(236.5 KB), once per critic iteration.  Such a message is latency-bound on xGMI, so it goes
out as ONE flat buffer (never per tensor), with a validity flag appended so that a rank whose
step raised still takes part in the collective - the reference's catch-and-continue contract
(train.py:349-352) would otherwise leave the other ranks waiting forever.  The flag is used on the
device (the gradient becomes the mean over the ranks that had one), never read by the host: a host
read per critic iteration would stop the host from enqueueing ahead of the GPU.  The generator has
no gradient (see gan_amd/generator.py), hence no second collective.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def env_world() -> tuple:
    """(rank, local_rank, world_size) from the torchrun environment, (0, 0, 1) without it."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def one_rank_exchange_requested() -> bool:
    """``AIG_DP_FORCE=1`` under a launcher that set the rendezvous variables: a communicator of ONE rank is still
    made and every gradient exchange, barrier and timing reduction goes through it - the whole multi-rank code path
    over the real library on a box with a single GPU (RCCL wants a device per rank)."""
    return os.environ.get("AIG_DP_FORCE") == "1" and "MASTER_ADDR" in os.environ


def exchanging() -> bool:
    """Do gradients go through a collective in this process?  More than one rank, or one rank made to."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or one_rank_exchange_requested()


def init_distributed(backend: Optional[str] = None) -> tuple:
    rank, local_rank, world = env_world()
    # this ROCm stack shares device memory between processes through dmabuf only; RCCL needs it said before
    # the HIP runtime starts (an environment that already says otherwise is left alone)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if (world > 1 or one_rank_exchange_requested()) and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm; AIG_DIST_BACKEND=gloo lets several ranks rehearse on one GPU
            backend = os.environ.get("AIG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _host_staged(t: torch.Tensor) -> bool:
    """Device tensors of a gloo group (several ranks rehearsing on one GPU; RCCL needs a device per rank) are
    exchanged through an explicit host copy made on the CALLER's stream.  Handed a device tensor, gloo stages it
    itself on streams of its own and makes the caller's stream wait for them; with two processes sharing one GPU's
    hardware queues the caller's stream was seen parked behind such a wait for good (the step's next replay never
    started while every stream of the process this package made was idle)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce(t: torch.Tensor, op=None) -> None:
    """``dist.all_reduce`` in place; over RCCL as it stands, over gloo through the host (see ``_host_staged``)."""
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(t):
        host = t.cpu()                      # waits for the current stream up to here, and for nothing else
        dist.all_reduce(host, op=op)
        t.copy_(host)                       # on the current stream
    else:
        dist.all_reduce(t, op=op)


def broadcast(t: torch.Tensor, src: int = 0) -> None:
    if _host_staged(t):
        host = t.cpu()
        dist.broadcast(host, src=src)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src)


def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0) -> None:
    """Every rank starts from rank ``src``'s weights (one flat broadcast per module)."""
    if not exchanging():
        return
    for m in modules:
        tensors = [p.data for p in m.parameters()] + [b.data for b in m.buffers()]
        if not tensors:
            continue
        flat = torch.cat([t.reshape(-1).float() for t in tensors])
        broadcast(flat, src=src)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


class FlatGradAllReduce:
    """Mean of the parameter gradients over ranks through one flat all-reduce.

    ``reduce(ok)``: a rank whose step failed passes ``ok=False`` and contributes nothing; every rank
    (that one included) ends up with the mean over the ranks that did report a gradient - zeros when
    none did - so all ranks take the same optimiser step and their weights stay equal.  No host
    synchronisation.  Returns True when the caller should step (always, with more than one rank).

    With more than one rank the parameters' ``.grad`` tensors ARE slices of the flat buffer: a step that zeroes
    its gradients in place and adds into them (the recorded critic terms do) leaves nothing to pack, and the
    reduced values need no unpacking - a reduce is four small launches and the collective.  (Packing and unpacking
    fifteen tensors around every collective was 0.5 ms per critic iteration at 100k nodes: three dozen small
    launches in a row on the stream everything else waits for.)  A gradient that was replaced by a tensor of its
    own - an eager backward pass after ``zero_grad(set_to_none=True)`` - is copied in, and ``.grad`` points
    back into the flat buffer afterwards."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        first = self.params[0]
        self.flat = torch.zeros(self.numel + 1, dtype=torch.float32, device=first.device)
        # (2 stands for "exchanging": a one-rank communicator made to exchange counts its single report)
        self.world = (max(dist.get_world_size(), 2) if exchanging() else 1)
        self._flags = torch.tensor([0.0, 1.0], dtype=torch.float32, device=first.device)
        self._view_cache = None
        if self.world > 1:
            for p, v in zip(self.params, self._views()):
                if p.grad is None:
                    p.grad = v

    def reduce(self, ok: bool = True) -> bool:
        if self.world == 1:
            return ok
        views = self._views()
        n = self.numel
        if ok:
            # gradients that live elsewhere (an eager pass installed tensors of its own) are copied in
            away = [(v, p.grad) for v, p in zip(views, self.params) if p.grad is not v]
            missing = [v for v, g in away if g is None]
            if missing:
                torch._foreach_zero_(missing)
            away = [(v, g) for v, g in away if g is not None]
            if away:
                torch._foreach_copy_([v for v, _ in away], [g for _, g in away])
        else:
            self.flat.zero_()
        self.flat[n:].copy_(self._flags[1:2] if ok else self._flags[0:1])
        all_reduce(self.flat)
        self.flat[:n].div_(self.flat[n].clamp(min=1.0))   # valid ranks, counted on the device
        for p, v in zip(self.params, views):
            p.grad = v
        return True

    def _views(self) -> List[torch.Tensor]:
        if self._view_cache is None:
            out, off = [], 0
            for p in self.params:
                n = p.numel()
                out.append(self.flat[off:off + n].view_as(p))
                off += n
            self._view_cache = out
        return self._view_cache


def shard_indices(total: int, rank: int, world: int) -> List[int]:
    """Samples of a bulk-sampling job owned by ``rank``: i with i % world == rank (no exchange)."""
    return list(range(rank, total, world))
