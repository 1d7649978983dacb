"""Sampling CLI: template + generator checkpoint -> repaired AIG ``.pt`` file(s).

Same flags and steps as AIGgan.py of the reference (AIGgan.py:72-117): seed, load templates,
build the generator and load its weights, draw ``z`` on the device, one forward pass without
gradients, ``enforce_aig_constraints``, ``resync_attrs``, save.  Extensions: ``--num-samples`` for
bulk sampling (sample i draws z from a device generator seeded ``seed + i``, so sample 0 is the
reference's draw and the set of samples does not depend on how many ranks produce it), and
under torchrun the samples are dealt to the ranks with no collective at all.  Bulk sampling takes the
draws in blocks of ``--batch``: a block goes through generator, repair and resync as one set of graphs
(``sample_sets``); ``--batch 1`` keeps ``--in-flight`` single samples in the air on as many streams
(``sample_many``).
"""
from __future__ import annotations

import argparse
import os
from collections import deque
from typing import Callable, Iterable, Iterator, List, Optional, Tuple

import torch

from . import parallel
from .constraints import (enforce_aig_constraints, enforce_begin, enforce_finish, repair_and_resync,
                          repair_and_resync_set, resync_attrs)
from .data import Data
from .dataset import load_pt_dataset
from .generator import AIGGenerator
from .graphset import GraphSet

PI, PO, AND = 0, 1, 2


def summarize(d: Data) -> str:
    n = d.x.size(0)
    e = d.edge_index.size(1) if getattr(d, "edge_index", None) is not None else 0
    t = d.x[:, 0].long()
    counts = [int((t == k).sum().item()) for k in (PI, PO, AND)]
    inv = int((d.edge_attr.view(-1) == 1).sum().item()) if getattr(d, "edge_attr", None) is not None else 0
    return f"N={n}, E={e}, PI={counts[0]}, PO={counts[1]}, AND={counts[2]}, inv_edges={inv}"


def sample_one(G: AIGGenerator, tpl: Data, z: torch.Tensor) -> Data:
    """AIGgan.py:104-111: forward, repair, resync."""
    with torch.no_grad():
        raw = G(tpl, z)
    return repair_and_resync(raw)   # = resync_attrs(enforce_aig_constraints(raw)), one graph walk fewer


def sample_many(G: AIGGenerator, jobs: Iterable[Tuple[Data, Callable[[], torch.Tensor]]], in_flight: int = 4,
                streams: Optional[List[torch.cuda.Stream]] = None) -> Iterator[Data]:
    """``sample_one`` over many ``(template, draw_z)`` jobs with ``in_flight`` samples in the air.

    The repair and the resync of one sample each hold a level-by-level graph walk that keeps a single
    CU busy for about a millisecond, and the repair ends in a host read (its edge count).  Samples are
    independent, so sample i is enqueued on stream ``i % in_flight`` and only finished (count read,
    resync enqueued) once ``in_flight - 1`` younger samples have been enqueued behind it: the walks of
    a few samples and the dense kernels of another run side by side.  ``draw_z`` is called under the
    sample's stream.  Results come back in job order, the same tensors ``sample_one`` returns, and
    the caller's current stream waits for each before it is handed out."""
    if in_flight < 1:
        raise ValueError("in_flight must be at least 1")
    cur = torch.cuda.current_stream()
    if streams is None:
        streams = [torch.cuda.Stream(device=cur.device) for _ in range(in_flight)]
    for st in streams:
        st.wait_stream(cur)
    pending: deque = deque()

    def finish():
        st, handle = pending.popleft()
        with torch.cuda.stream(st):
            d = resync_attrs(enforce_finish(handle))
        cur.wait_stream(st)
        for v in d.to_dict().values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return d

    for i, (tpl, draw_z) in enumerate(jobs):
        st = streams[i % len(streams)]
        with torch.cuda.stream(st), torch.no_grad():
            pending.append((st, enforce_begin(G(tpl, draw_z()), want_depth=False)))   # the resync replaces it
        if len(pending) >= len(streams):
            yield finish()
    while pending:
        yield finish()


# Scratch of a block of draws grows with draws x nodes: 2 KB of projections per node and the score matrix
# (candidates x 4 bytes per node).  A block is capped at about a million nodes - a few gigabytes.
SET_NODES = 1 << 20


def block_size(n: int, batch: int) -> int:
    """Draws per block for templates of ``n`` nodes: ``batch``, capped by the scratch a block needs."""
    return max(1, min(int(batch), SET_NODES // max(int(n), 1)))


_SET_STREAMS: dict = {}


def _set_streams(device: torch.device, count: int) -> List[torch.cuda.Stream]:
    """The streams blocks alternate over, kept per device: the allocator keeps a pool per stream, and a fresh stream
    per call would have every call start from an empty pool (a block's scratch is a few hundred megabytes)."""
    have = _SET_STREAMS.setdefault(device.index, [])
    while len(have) < count:
        have.append(torch.cuda.Stream(device=device))
    return have[:count]


def sample_sets(G: AIGGenerator, tpl: Data, z_batches: Iterable[torch.Tensor], in_flight: int = 3) -> Iterator[GraphSet]:
    """Bulk sampling from ONE template: every element of ``z_batches`` is a ``[S, z_dim]`` block of draws that
    goes through generator, repair and resync as one set of graphs (``AIGGenerator.forward_set``,
    ``repair_and_resync_set``): about forty launches per block whatever its size, and one host read (the
    repaired edge counts) that is only awaited once ``in_flight - 1`` younger blocks have been enqueued.
    Block ``i`` runs on stream ``i % in_flight`` (``in_flight`` 1: the caller's stream): the narrow kernels of one
    block's repair and selection run beside the wide scoring kernels of the next.  Blocks are independent, so the
    results do not depend on ``in_flight``.  Yields one ``GraphSet`` per block, in block order, which the caller's
    stream has been made to wait for; its graph ``i`` equals ``sample_one(G, tpl, zs[i])``."""
    if in_flight < 1:
        raise ValueError("in_flight must be at least 1")
    cur = torch.cuda.current_stream()
    streams = _set_streams(cur.device, in_flight) if in_flight > 1 else [cur]
    pending: deque = deque()
    if len(streams) > 1 and tpl.x.is_cuda and tpl.x.dtype == torch.float32 and tpl.x.is_contiguous():
        # the template's plan is shared by all blocks: built here, on the caller's stream, which every block's stream
        # waits for (built inside the first block it would be read by the second without any ordering)
        nd = getattr(tpl, "node_depth", None)
        with torch.no_grad():
            G.plan_for(tpl, tpl.x, nd.to(tpl.x.device) if nd is not None else None)

    def finish() -> GraphSet:
        gset, counts_host, ready, st = pending.popleft()
        ready.synchronize()
        gset._counts_host = [int(c) for c in counts_host.tolist()]
        if st is not cur:
            cur.wait_stream(st)
            for v in (getattr(gset, name, None) for name in type(gset).__slots__):   # allocated under st, used by the caller's stream from here on
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
        return gset

    for i, zs in enumerate(z_batches):
        st = streams[i % len(streams)]
        if st is not cur:
            st.wait_stream(cur)                # the draws (and the template) come from the caller's stream
        with torch.cuda.stream(st), torch.no_grad():
            gset = repair_and_resync_set(G.forward_set(tpl, zs))
            counts_host = torch.empty(gset.B, dtype=torch.int32, pin_memory=True)
            counts_host.copy_(gset.count, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        if st is not cur and torch.is_tensor(zs) and zs.is_cuda:
            zs.record_stream(st)
        pending.append((gset, counts_host, ready, st))
        if len(pending) >= in_flight:
            yield finish()
    while pending:
        yield finish()


def sample_bulk(G: AIGGenerator, tpl: Data, z_batches: Iterable[torch.Tensor], in_flight: int = 3) -> Iterator[Data]:
    """``sample_sets`` handed out graph by graph, in draw order: the same tensors ``sample_one(G, tpl, z)``
    returns for every z, as views of the block's tensors."""
    for gset in sample_sets(G, tpl, z_batches, in_flight):
        for g in range(gset.B):
            yield gset.graph(g, copy=False)


def out_path(base: str, i: int, total: int) -> str:
    if total == 1:
        return base
    root, ext = os.path.splitext(base)
    return f"{root}_{i:06d}{ext or '.pt'}"


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset-dir", required=True, help="directory of real AIG .pt files used as templates")
    ap.add_argument("--gen-ckpt", required=True, help="generator state dict (.pt)")
    ap.add_argument("--out", required=True, help="output path, e.g. results/ISCAS85/AIGfake/sample_0.pt")
    ap.add_argument("--g-hidden", type=int, default=128)
    ap.add_argument("--z-dim", type=int, default=128)
    ap.add_argument("--candidate-k", type=int, default=512)
    ap.add_argument("--template-idx", type=int, default=0)
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--num-samples", type=int, default=1, help="extension: bulk sampling")
    ap.add_argument("--in-flight", type=int, default=4, help="extension: samples kept in flight with --batch 1")
    ap.add_argument("--batch", type=int, default=64,
                    help="extension: draws per block during bulk sampling - a block is one set of graphs for every kernel "
                         "(the samples are the same for any value)")
    ap.add_argument("--score-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="extension: arithmetic of the node projections behind the candidate scores (bf16 = bf16 matrix "
                         "cores with f32 accumulation, for bulk sampling; the default is the reference's f32)")
    ap.add_argument("--allow-pickle", action="store_true",
                    help="extension: also read dataset files that need a full unpickle (trusted files only)")
    return ap


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    rank, local_rank, world = parallel.env_world()   # bulk sampling needs no process group
    if not torch.cuda.is_available():
        raise RuntimeError("gan_amd.sample needs a HIP device: the generator has no CPU path")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)

    dataset = load_pt_dataset(args.dataset_dir, max_files=None, allow_pickle=args.allow_pickle)
    if len(dataset) == 0:
        raise RuntimeError(f"No .pt files in {args.dataset_dir}")
    tpl = dataset[min(args.template_idx, len(dataset) - 1)].to(device)

    G = AIGGenerator(node_in_dim=2, hidden_dim=args.g_hidden, z_dim=args.z_dim, candidate_k=args.candidate_k,
                     score_dtype=args.score_dtype).to(device)
    G.load_state_dict(torch.load(args.gen_ckpt, map_location=device, weights_only=True))
    G.eval()

    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    mine = parallel.shard_indices(args.num_samples, rank, world)

    def z_of(i):   # per-sample seed: the output set does not depend on the number of ranks
        return lambda: torch.randn(args.z_dim, device=device, generator=torch.Generator(device=device).manual_seed(args.seed + i))

    block = block_size(tpl.x.size(0), args.batch)

    def z_blocks():     # this rank's draws in blocks that go through the kernels as one set of graphs
        for b in range(0, len(mine), block):
            yield torch.stack([z_of(i)() for i in mine[b:b + block]])

    def bulk():     # a block's tensors cross to the host in a few large copies; its graphs are cut out there
        for gset in sample_sets(G, tpl, z_blocks()):
            host = gset.cpu()
            for g in range(host.B):
                yield host.graph(g)

    results = bulk() if block > 1 else sample_many(G, ((tpl, z_of(i)) for i in mine), in_flight=args.in_flight)
    for i, fixed in zip(mine, results):
        if args.num_samples == 1:
            print("[INFO] FIX:", summarize(fixed))
        path = out_path(args.out, i, args.num_samples)
        torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in fixed.to_dict().items()}, path)
        if args.num_samples == 1:
            print(f"[OK] saved {path}")
    if args.num_samples > 1:
        print(f"[OK] rank {rank}: saved {len(parallel.shard_indices(args.num_samples, rank, world))} samples")


if __name__ == "__main__":
    main()
