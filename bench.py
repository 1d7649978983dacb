#!/usr/bin/env python3
"""Benchmark of the AIG-GAN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload synth100k|iscas85|synth1m]

A "step" is one full pass of the training loop body (gan_amd.train.Trainer.step) over one
minibatch of synthetic AIGs: 2 generator passes per template, repair, 3 critic iterations with
the node-only gradient penalty (10 critic evaluations, 4 backward passes), the generator-side
penalties.  Every rank holds its own minibatch (weak scaling); for N > 1 the critic gradient is
all-reduced over RCCL after each critic backward.  Rank 0 prints ONE JSON line:

  value            train edges/s = (real + fake non-sentinel edges of all ranks' minibatches) / step time
  sample_nodes_per_sec   template nodes/s of G.forward + enforce + resync (candidate-k 64), whole job, in
                   blocks of draws that go through the kernels as one set of graphs (bulk sampling);
                   ..._streams keeps single samples in flight on several streams, ..._one_at_a_time is the plain loop
  roofline         the kernel with the largest share of the step's kernel time among those with a known algorithmic cost
                   (chosen from this run's own census of a few fully eager steps): algorithmic bytes or flops of its eager
                   launches inside a timed region / their duration from events the library records around the kernel
                   itself on its launch stream; roofline_kernels lists every such kernel, per graph kind (real / fake);
                   roofline_step sets the algorithmic bytes and flops of the whole step against ms_per_step
  cpu_baseline     the CPU checker's training step on a bounded sample, timed on this host (rank 0, N=1)

Inputs are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_F32_PEAK_TF = 157.3  # dense f32 matrix peak of MI355X
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak; ~6300 GB/s is the measured copy ceiling


def workload_minibatches(name: str, rank: int, distinct: int):
    """``distinct`` different minibatches (lists of graphs) for this rank, and a description.  The timed loop
    walks through them in turn, as a shuffling loader over a dataset does (train.py:229-233): no two
    consecutive steps see the same graphs."""
    from gan_amd.synth import iscas85_like, synth_aig
    seeds = [rank * distinct + i for i in range(distinct)]
    if name == "synth100k":
        return [[synth_aig(1000, 98000, 1000, seed=s, window=4096, orientation="dataset")] for s in seeds], \
            "synthetic 100k-node AIGs (1000 PI / 98000 AND / 1000 PO, window 4096), one graph per step per GPU"
    if name == "synth1m":
        return [[synth_aig(10000, 980000, 10000, seed=s, window=4096, orientation="dataset")] for s in seeds], \
            "synthetic 1M-node AIGs (10k PI / 980k AND / 10k PO, window 4096), one graph per step per GPU"
    if name == "iscas85":
        return [iscas85_like(seed=s) for s in seeds], \
            "synthetic graphs with ISCAS85 node counts (c17..c7552), one batch of all 11 per step per GPU"
    raise SystemExit(f"unknown workload {name}")


def device_minibatches(groups, dev):
    """The minibatches padded to the workload's (max_nodes, max_edges) as train.py:200-204 does, resident in HBM."""
    from gan_amd.data import Batch
    from gan_amd.dataset import pad_data
    mn = max(g.x.size(0) for gs in groups for g in gs)
    me = max(g.edge_index.size(1) for gs in groups for g in gs)
    top = max(int(g.node_depth.max()) for gs in groups for g in gs)
    batches, next_id = [], 0
    for gs in groups:
        b = Batch.from_data_list([pad_data(g, mn, me) for g in gs])
        b._depth_top = max(int(g.node_depth.max()) for g in gs)     # noted from the host-side data, as GraphLoader does
        b.sample_ids = list(range(next_id, next_id + len(gs)))
        next_id += len(gs)
        batches.append(b.to(dev))
    return batches, mn, me, top


def measured_peaks():
    """Achievable peaks of an MI355X measured with tools/peaks.hip (committed as profiles/r01_peaks.json)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_peaks.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def pmc_traffic(workload: str, kernel: str, kind: str):
    """Bytes per launch that left the L2 towards memory for ``kernel`` on this workload's ``kind`` of graph
    ("real" / "fake"), from the committed counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, see
    profiles/pmc_traffic.json); None when not collected."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)[workload]["by_kind"][kind][kernel]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def rocprof_shares():
    """Share of GPU time per kernel (percent) from the committed rocprofv3 summary of this bench command."""
    import csv
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_bench_synth100k_kernel_stats.csv")
    out = {}
    try:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = re.sub(r"[(<].*", "", re.sub(r"^(void )?(aig::\(anonymous namespace\)::)?", "", r["Name"]))
                name = {"k_linear128_x6": "k_linear128", "k_wgrad128_x6": "k_wgrad128"}.get(name, name)     # (the kernels behind aig_linear128 / aig_wgrad)
                out[name] = round(out.get(name, 0.0) + float(r["Percentage"]), 2)
    except (OSError, KeyError, ValueError):
        pass
    return out


# Kernels whose launches the library can bracket with events of their own (aig_kernel_timing; index = kernel id) and
# what one launch on a graph of n nodes / e edges / `levels` depth levels costs by the algorithm: bytes for the
# memory-bound ones (every gathered row counts once per list entry, whether a cache serves it or not), flops for the
# two dense products (which sit at the ridge: their bytes are given too).
HBM_KERNELS = ["k_gat_aggregate", "k_gat_bwd_dst", "k_gat_bwd_src", "k_pool_walk", "k_gat1_fwd", "k_gat1_bwd", "k_linear128",
               "k_wgrad128"]
MFMA_KERNELS = []
# k_linear128 and k_wgrad128 take their products on the bf16 matrix cores (every f32 operand cut into three bf16
# pieces, six piece products: f32 accuracy at 6/16 of the f32 matrix time, csrc/dense6.hip, csrc/wgrad6.hip), which
# leaves them bound by their [n,128] operands: priced against the memory, the f32-equivalent product rate beside it.


def timed_kernel_names(lib):
    names, k = [], 0
    while lib.aig_kernel_timing_name(k):
        names.append(lib.aig_kernel_timing_name(k).decode())
        k += 1
    return names


def kernel_bytes(kernel: str, n: int, e: int, levels: int) -> int:
    """Algorithmic bytes of one launch (E' = e + n list entries with the self loops).
      k_gat_aggregate  see gat_fwd_bytes
      k_gat_bwd_dst    (second layer, fed by the readout: its incoming gradient is formed from 256 floats and a table)
                       row out read and g_pre written (2 x 512 B per node), level + weight per node (8 B), one gathered
                       source row per list entry (512 B), a_s per entry and a_d per node (16 B), row pointers, source id
                       + attribute per edge (8 B), self-loop attribute (4 B), alpha read and d_raw written per entry
                       (16 B each), d_a_d written per node (16 B), the argmax table (512 B per level)
      k_gat_bwd_src    one gathered g_pre row per list entry, the node's own xw row read and d_xw row written, alpha
                       and d_raw read per entry (16 B each), d_a_d per node, column pointers, position + target per edge
      k_pool_walk      512 B row + 12 B of plan per node, 1 KB of level table per level
      k_gat1_fwd       (first layer, no projected rows) per node: list bounds, own features, self-loop attribute (16 B),
                       the row written (512 B), alpha of the self loop (16 B), its own features as the loop's source
                       (8 B); per edge: source id, attribute, the source's features, alpha written (32 B)
      k_gat1_bwd       per node: rows g_out and out read (1024 B), bounds / features / loop attribute / loop alpha
                       (40 B); per edge: source id, attribute, source features, alpha (32 B)
      k_linear128, k_wgrad128   two [n,128] f32 arrays (1024 B per node)"""
    ep = e + n
    if kernel == "k_gat_aggregate":
        return gat_fwd_bytes(n, e)
    if kernel == "k_gat_bwd_dst":
        return 1024 * n + 8 * n + 512 * ep + 16 * ep + 16 * n + 4 * (n + 1) + 8 * e + 4 * n + 32 * ep + 16 * n + 512 * levels
    if kernel == "k_gat_bwd_src":
        return 512 * ep + 1024 * n + 32 * ep + 16 * n + 4 * (n + 1) + 8 * e
    if kernel == "k_pool_walk":
        return 524 * n + 1024 * levels
    if kernel == "k_gat1_fwd":
        return 552 * n + 32 * e
    if kernel == "k_gat1_bwd":
        return 1064 * n + 32 * e
    if kernel in ("k_linear128", "k_wgrad128"):
        return 1024 * n
    raise KeyError(kernel)


def kernel_flops(kernel: str, n: int) -> float:
    """[n,128] x [128,128] (k_linear128) and its transpose-side reduction over the nodes (k_wgrad128)."""
    if kernel in ("k_linear128", "k_wgrad128"):
        return 2.0 * 128 * 128 * n
    raise KeyError(kernel)


def step_cost_model(n: int, e_real: int, e_fake: int, levels: int, K: int, n_critic: int = 3, gp_rows: int = 0,
                    classes: int = 0):
    """Algorithmic HBM bytes and flops of ONE training step on a minibatch of n nodes, itemised: what the step's kernels
    read and write by the algorithm (arrays streamed once per kernel that takes them; gathered rows once per list
    entry), summed over the step's passes.  [n,128] f32 arrays are counted in units of A = 512 n bytes.
      edge pass (critic forward + backward over a graph with edges; 2 n_critic + 1 per step: real and fake sides of
      every critic iteration, raw graphs of the generator step): first layer forward, second layer's projection,
      attention terms (A + 32 n), aggregate, readout walk, target-side and source-side backward, projection's input
      gradient and weight gradient (k_linear128, k_wgrad128), first layer backward
      penalty term (closed-form gradient penalty, n_critic per step): 32 A (20 array reads, 12 writes over its eleven
      row passes, four products and four node-dimension reductions)
      generator pass (2 per step): node pipeline 8 n + 5 A, scoring (K + 127 staged rows of 512 B per 64 targets and
      4 K bytes of logits per target), selection (4 K per target), emission (~1.5 KB per target)
    gp_rows > 0: the penalty term runs over that many distinct rows instead of n (plus 64 n bytes to find them, once per
    step); classes > 0: the generator projects one row per class of equal feature rows and scores a slot by table
    lookups - 8 K bytes per target (candidate ids read, logits written) + selection + 60 B of emission."""
    A = 512 * n

    def edge_pass(e):
        b = (kernel_bytes("k_gat1_fwd", n, e, levels) + 2 * A + (A + 32 * n) + kernel_bytes("k_gat_aggregate", n, e, levels) +
             kernel_bytes("k_pool_walk", n, e, levels) + kernel_bytes("k_gat_bwd_dst", n, e, levels) +
             kernel_bytes("k_gat_bwd_src", n, e, levels) + 2 * A + 2 * A + kernel_bytes("k_gat1_bwd", n, e, levels))
        return b, 3 * 2.0 * 128 * 128 * n
    real_b, real_f = edge_pass(e_real)
    fake_b, fake_f = edge_pass(e_fake)
    gp_n = gp_rows if gp_rows > 0 else n
    gp_b, gp_f = 32 * 512 * gp_n, 9 * 2.0 * 128 * 128 * gp_n   # 4 products + 2 weight gradients over 128 columns (+ small ones)
    if gp_rows > 0:
        gp_b += 64 * n // n_critic
    targets = n
    if classes > 0:
        gen_b = targets * (8 * K + 4 * K + 60) + classes * 5 * 512
        gen_f = 7 * 2.0 * 128 * 128 * classes + 3.0 * 128 * classes * classes
    else:
        gen_b = 8 * n + 5 * A + targets * ((K + 127) * 512 // 64 + 4 * K) + targets * 4 * K + targets * 1536
        gen_f = 7 * 2.0 * 128 * 128 * n + 3.0 * 128 * K * targets
    items = {"critic_pass_real": (n_critic, real_b, real_f), "critic_pass_generated": (n_critic + 1, fake_b, fake_f),
             "penalty_term": (n_critic, gp_b, gp_f), "generator_pass": (2, gen_b, gen_f)}
    total_b = sum(c * b for c, b, _ in items.values())
    total_f = sum(c * f for c, _, f in items.values())
    return total_b, total_f, {k: {"per_step": c, "bytes": int(b), "flops": f} for k, (c, b, f) in items.items()}


def read_kernel_times(lib):
    """{kernel: (total ms, launches)} recorded by the library's kernel timer since it was switched on."""
    import ctypes as C
    out = {}
    for k, name in enumerate(timed_kernel_names(lib)):
        ms, cnt = C.c_double(), C.c_int64()
        if lib.aig_kernel_timing_read(k, C.byref(ms), C.byref(cnt)) == 0 and cnt.value:
            out[name] = (ms.value, cnt.value)
    return out


def gat_fwd_bytes(n: int, e: int) -> int:
    """Algorithmic bytes of one fused aggregate launch (E' = e + n edges incl. self loops):
    gather 512 B per edge, write 512 B per node, int32 source + fp32 attribute per real edge,
    row pointers, attention terms (a_s gather + alpha write per edge and head, a_d / loop attr per node)."""
    ep = e + n
    return 512 * ep + 512 * n + 8 * e + 4 * (n + 1) + 16 * ep + 16 * ep + 16 * n + 4 * n


def cpu_baseline(seconds_budget: float = 25.0):
    """The CPU checker (the oracle this repository holds its kernels to: kind "port") timed on this host on
    BASELINE.json's CPU configuration: the eleven ISCAS85-sized graphs, one epoch of ``train.py`` steps at
    batch-size 1 (templates keep the padded size, as in the reference) and one ``AIGgan.py`` sample per graph at
    candidate-k 64 - once on ONE torch thread and once on all the cores this process may use.  The reference's
    own Python path cannot run here (it imports torch_geometric); the timings measured for it on the same kind of
    graph when this work was surveyed are quoted as ``reference_python``."""
    import numpy as np
    from gan_amd.data import Batch, Data
    from gan_amd.dataset import pad_data, unpad_data
    from gan_amd.synth import iscas85_like, synth_aig
    from oracle import cpu as ocpu
    from oracle import model as omodel
    from oracle.train_oracle import CpuTrainer
    w = np.load(os.path.join(ROOT, "tests", "golden", "gen_weights.npz"))
    gs = {k: torch.from_numpy(w[k]) for k in w.files}
    graphs = iscas85_like(seed=0)
    mn = max(g.x.size(0) for g in graphs)
    me = max(g.edge_index.size(1) for g in graphs)
    padded = [pad_data(g, mn, me) for g in graphs]
    gen = torch.Generator().manual_seed(0)
    zs = [torch.randn(128, generator=gen) for _ in range(2)]
    alphas = [torch.rand(1, 1, generator=gen) for _ in range(3)]
    all_cores = torch.get_num_threads()

    def epoch_and_samples(threads):
        torch.set_num_threads(threads)
        tr = CpuTrainer(gs, omodel.init_discriminator_params(0), 64, mn, me)
        t0 = time.perf_counter()
        edges = 0
        for p in padded:
            real = Batch.from_data_list([p])
            templates = [unpad_data(d) for d in real.to_data_list()]
            edges += tr.step(real, templates, zs[:1], zs[1:], alphas, validate=False)["edges"]
        t_train = time.perf_counter() - t0
        t0 = time.perf_counter()
        nodes = 0
        for g in graphs:
            raw = omodel.generator_forward_canonical(gs, g.x, g.node_depth, zs[0], 64)
            d = Data(x=raw["x"], edge_index=raw["edge_index"], edge_attr=raw["edge_attr"], node_depth=g.node_depth.clone())
            ocpu.resync_attrs(ocpu.enforce_aig_constraints(d))
            nodes += int(g.x.size(0))
        t_sample = time.perf_counter() - t0
        return {"threads": threads, "train_edges_per_sec": round(edges / t_train, 1), "epoch_seconds": round(t_train, 3),
                "sample_nodes_per_sec": round(nodes / t_sample, 1), "sample_seconds": round(t_sample, 3)}

    one = epoch_and_samples(1)
    many = epoch_and_samples(all_cores) if all_cores > 1 else one
    torch.set_num_threads(all_cores)
    best = max((one, many), key=lambda r: r["train_edges_per_sec"])
    # secondary datum: one step on a 10k-node graph (the size at which the CPU step still takes seconds)
    g10 = synth_aig(100, 9800, 100, seed=0, window=1024, orientation="dataset")
    real10 = Batch.from_data_list([pad_data(g10, g10.x.size(0), g10.edge_index.size(1))])
    tr10 = CpuTrainer(gs, omodel.init_discriminator_params(0), 64, g10.x.size(0), g10.edge_index.size(1))
    t0 = time.perf_counter()
    m10 = tr10.step(real10, [unpad_data(d) for d in real10.to_data_list()], zs[:1], zs[1:], alphas, validate=False)
    t10 = time.perf_counter() - t0
    return {"value": best["train_edges_per_sec"], "unit": "edges/s", "cores": best["threads"], "kind": "port",
            "sample": f"one epoch of train.py steps (11 steps, batch-size 1, graphs padded to {mn} nodes / {me} edges) plus one "
                      "AIGgan.py sample per graph at candidate-k 64 on the eleven ISCAS85-sized synthetic graphs, CPU checker "
                      "(generator / constraints: single-threaded C; critic: torch CPU ops on the stated threads)",
            "one_thread": one, "all_cores": many, "host_threads_available": all_cores,
            "step_on_10k_node_graph": {"edges_per_sec": round(m10["edges"] / t10, 1), "seconds": round(t10, 2), "threads": all_cores},
            "reference_python": {"note": "the reference's unmodified Python path, measured when the work was surveyed (1 CPU thread, "
                                         "synthetic forward AIGs, candidate-k 64); it cannot be run on this box",
                                 "AIGGenerator.forward": "0.843 s at 2,300 nodes (2,728 nodes/s); 0.171 s at 486 nodes",
                                 "enforce_aig_constraints": "107 s at 2,300 nodes; 4.16 s at 486 nodes (quadratic)",
                                 "aig_constraint_penalties / validate_strict_aig": "0.061 s / 0.162 s at 2,300 nodes"}}


def side_workload(name: str, dev, steps: int, warmup: int):
    """A short measurement of another BASELINE.json configuration on the same GPU (rank 0, N=1 only):
    training step time, edges/s and bulk sampling rate, reported next to the main line."""
    from gan_amd.sample import block_size, sample_many, sample_sets
    from gan_amd.train import Trainer, build_parser
    distinct = 8
    groups, desc = workload_minibatches(name, 0, distinct)
    batches, mn, me, top = device_minibatches(groups, dev)
    targs = build_parser().parse_args(["--dataset-dir", "synthetic", "--candidate-k", "64", "--save-every", "1000000000"])
    targs.level_cap = top
    tr = Trainer(targs, mn, me, dev)
    tr.global_step = 1
    for i in range(distinct + warmup):      # one pass over the minibatches (per-sample templates), then warm-up
        tr.step(batches[i % distinct])
    import gc
    gc.collect()
    gc.freeze()        # as in the main measurement (and gan_amd.train.train): no full collection inside the timed steps
    torch.cuda.synchronize()
    edges = torch.zeros((), dtype=torch.long, device=dev)
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[i % distinct])
        edges += tr._edges_last
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    batch = batches[0]
    templates = tr.templates_of(batch)
    # bulk sampling as gan_amd.sample runs it: per template, blocks of draws as one set of graphs each
    S = min(block_size(int(t.x.size(0)), 64) for t in templates)
    blocks = lambda k: (torch.randn(S, 128, device=dev) for _ in range(k))   # noqa: E731

    def sample_all(k):
        for t in templates:
            if S > 1:
                for _ in sample_sets(tr.G, t, blocks(k)):
                    pass
            else:
                for _ in sample_many(tr.G, ((t, lambda: torch.randn(128, device=dev)) for _ in range(k))):
                    pass

    sample_all(1)
    torch.cuda.synchronize()
    s0 = time.perf_counter()
    sample_all(2)
    torch.cuda.synchronize()
    nodes = sum(int(t.x.size(0)) for t in templates) * S * 2
    return {"workload": f"{name}: {desc}; {distinct} different minibatches in rotation", "steps": steps,
            "ms_per_step": round(dt * 1e3, 3), "edges_per_sec": round(int(edges.item()) / steps / dt, 1),
            "recordings": tr.captures, "sample_block": S,
            "sample_nodes_per_sec": round(nodes / (time.perf_counter() - s0), 1)}


def bulk_sampling(a, rank, world, dev):
    """BASELINE.json's bulk-sampling configuration on this rank's share: samples from one c7552-sized template at
    candidate-k 256 with the node projections on the bf16 matrix cores, no collective (ranks only meet at the
    timing barriers).  A step = one block of --bulk-batch samples that goes through generator, repair and resync
    as one set of graphs (gan_amd.sample.sample_sets).  Rank 0 prints the JSON line."""
    from gan_amd import _abi, parallel
    from gan_amd.data import Data
    from gan_amd.generator import AIGGenerator
    from gan_amd.sample import sample_bulk
    from gan_amd.synth import ISCAS85_SHAPES, synth_aig
    name, n_pi, n_po, n_and = ISCAS85_SHAPES[-1]
    g = synth_aig(n_pi, n_and, n_po, seed=0, window=max(8, n_and // 8 + n_pi), orientation="dataset")
    tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
    n = int(g.x.size(0))
    torch.manual_seed(123)
    G = AIGGenerator(candidate_k=256, score_dtype=a.score_dtype).to(dev).eval()
    S = a.bulk_batch

    def block(step):      # z_i from the CPU generator seeded base + rank * 10^6 + i, as the configuration words it
        zs = torch.stack([torch.randn(128, generator=torch.Generator().manual_seed(7 + rank * 10**6 + step * S + i))
                          for i in range(S)])
        return zs.to(dev, non_blocking=True)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def run(blocks, **kw):
        edges = 0
        for d in sample_bulk(G, tpl, blocks, **kw):
            edges += int(d.edge_index.size(1))
        return edges

    # the draws of the timed blocks are in device memory before the clock starts (drawing 64 seeded vectors on the host
    # takes about as long as the GPU needs for the block)
    run([block(k) for k in range(max(a.warmup, 1))])
    timed = [block(a.warmup + k) for k in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    edges = run(timed)
    barrier()
    elapsed = time.perf_counter() - t0
    stat = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        parallel.all_reduce(stat, op=dist.ReduceOp.MAX)
        elapsed = float(stat.item())
    # the projection launch of a block, bracketed with events in one extra untimed step
    entry = "aig_gen_node_proj_batch"
    _abi.timeline.reset()
    _abi.timeline.only = {entry}
    _abi.timeline.enabled = True
    # (alone on the GPU, one block at a time, with one row per NODE forced: the timed blocks above project one row per
    #  class of equal feature rows, a handful per template, which says nothing about the kernel)
    limit_was, G.class_limit = G.class_limit, 0
    run([block(a.warmup + a.steps + k) for k in range(2)], in_flight=1)
    G.class_limit = limit_was
    barrier()
    _abi.timeline.enabled = False
    ms, launches = _abi.timeline.totals_ms().get(entry, (0.0, 0))
    if rank != 0:
        return
    rows = S * ((n + 63) // 64 * 64)
    out = {"metric": "AIG nodes/sec sampled (bulk, candidate-k 256)", "value": round(n * S * a.steps * world / elapsed, 1),
           "unit": "nodes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(elapsed * 1e3 / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16" if a.score_dtype == "bf16" else "f32", "data": "synthetic",
           "config": {"workload": f"bulk_k256: {S} samples per step per GPU from one {name}-sized template ({n} nodes: {n_pi} PI / "
                                  f"{n_and} AND / {n_po} PO), candidate-k 256, node projections in {a.score_dtype}, generator + repair + "
                                  "resync of a block as one set of graphs, no collectives", "samples_per_step_per_gpu": S,
                      "samples_per_gpu": S * a.steps, "nodes_per_sample": n, "edges_per_sample": round(edges / (S * a.steps), 1),
                      "parallelism": f"{world} independent rank(s)"},
           "samples_per_sec": round(S * a.steps * world / elapsed, 1)}
    if launches:
        us = ms * 1e3 / launches
        flops = 7 * 2.0 * 128 * 128 * rows
        plane_row = 256 if a.score_dtype == "bf16" else 512      # bytes of one plane's row: halfs or f32
        bytes_out = rows * (8 + 4 * plane_row)     # x read, four output rows written
        peak = MFMA_BF16_PEAK_TF if a.score_dtype == "bf16" else MFMA_F32_PEAK_TF
        plan_c = getattr(tpl.__dict__.get("_gen_plan"), "C", None)
        out["config"]["classes_of_equal_feature_rows"] = plan_c
        out["roofline_note"] = ("the timed blocks project one row per class of equal feature rows and score pairs through "
                                "(class, class) tables; roofline / roofline_mfma are the node-projection kernel's own rates from "
                                "two extra blocks with one row per node forced (class_limit = 0)")
        out["roofline_mfma"] = {"kernel": f"node projections of one block ({entry})", "bound": "mfma",
                                "achieved": round(flops / (us * 1e-6) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                                "frac": round(flops / (us * 1e-6) / 1e12 / peak, 4), "launches": launches,
                                "avg_launch_us": round(us, 2), "rows_per_launch": rows}
        out["roofline"] = {"kernel": f"node projections of one block ({entry}): 8 B read and {4 * plane_row} B of output per "
                                     "row (four planes)", "bound": "hbm",
                           "achieved": round(bytes_out / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(bytes_out / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                           "launches": launches, "avg_launch_us": round(us, 2)}
    print(json.dumps(out))


def launch_ranks(n: int) -> int:
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>` as a child
    process; one rank per GPU over RCCL.  With fewer GPUs than ranks the ranks can only rehearse the
    data-parallel path over gloo (AIG_DIST_BACKEND=gloo), sharing the devices round-robin - said loudly, because
    such a run measures nothing about scaling."""
    import socket
    import subprocess
    have = torch.cuda.device_count()       # counting devices does not start the HIP runtime
    env = dict(os.environ)
    if have < n:
        if env.get("AIG_DIST_BACKEND") != "gloo":
            print(f"[bench] --gpus {n} but only {have} GPU(s) visible: RCCL needs one device per rank "
                  "(set AIG_DIST_BACKEND=gloo to rehearse the ranks on shared devices)", file=sys.stderr)
            return 2
        print(f"[bench] note: {n} ranks share {have} GPU(s) over gloo - a rehearsal, not a scaling measurement",
              file=sys.stderr)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 1)) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synth100k", choices=["synth100k", "iscas85", "synth1m", "bulk_k256"])
    ap.add_argument("--bulk-batch", type=int, default=64, help="bulk_k256: samples per step (one projection launch each)")
    ap.add_argument("--score-dtype", default="bf16", choices=["fp32", "bf16"], help="bulk_k256: arithmetic of the node projections")
    ap.add_argument("--sample-iters", type=int, default=10)
    ap.add_argument("--sample-in-flight", type=int, default=4)
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="leave the event brackets around the critic's memory-bound kernels out of the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="skip the short ISCAS85-sized measurement reported next to the main line at N=1")
    ap.add_argument("--trainer-args", default="", help="further train.py flags for the trainer (diagnosis), space-separated")
    ap.add_argument("--no-fused-update", action="store_true",
                    help="train.py's --no-fused-update: a critic iteration's update as separate launches")
    ap.add_argument("--capture-fake", default="auto", choices=["auto", "on", "off"],
                    help="record the critic's pass over generated graphs too (train.py's option of the same name)")
    ap.add_argument("--distinct", type=int, default=None,
                    help="different minibatches per GPU the timed loop rotates through (default 8; 4 for synth1m)")
    a = ap.parse_args()
    if a.distinct is None:
        a.distinct = 4 if a.workload == "synth1m" else 8

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started as a plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes
        # (this process has not touched the GPU: torch is imported, no HIP call was made), and leave with
        # their exit code.  Rank 0 of the children prints the JSON line.
        sys.exit(launch_ranks(a.gpus))

    from gan_amd import _abi, parallel
    from gan_amd.constraints import enforce_aig_constraints, repair_and_resync, resync_attrs
    from gan_amd.data import Batch, Data
    from gan_amd.dataset import pad_data
    from gan_amd.sample import sample_many
    from gan_amd.train import Trainer, build_parser

    rank, local_rank, world = parallel.init_distributed()
    if world != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))   # (modulo: rehearsals on fewer GPUs)
    torch.cuda.set_device(dev)
    torch.manual_seed(123)

    if a.workload == "bulk_k256":
        bulk_sampling(a, rank, world, dev)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    groups, desc = workload_minibatches(a.workload, rank, a.distinct)
    batches, max_nodes, max_edges, top = device_minibatches(groups, dev)
    targs = build_parser().parse_args(["--dataset-dir", "synthetic", "--candidate-k", "64", "--save-every", "1000000000",
                                       "--capture-fake", a.capture_fake] + (["--no-fused-update"] if a.no_fused_update else []) +
                                      a.trainer_args.split())
    targs.level_cap = top        # train.py sizes the recorded critic terms by the deepest graph of the dataset
    trainer = Trainer(targs, max_nodes, max_edges, dev, rank, world)
    trainer_exchanges = trainer.exchange
    trainer.global_step = 1      # step 0 would add the save-step validation pass; steady state is measured
    nb = len(batches)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    trace = os.environ.get("AIG_BENCH_TRACE") == "1"

    def phase(name):
        if trace:
            print(f"[bench rank {rank}] {name}", file=sys.stderr, flush=True)

    if trace:     # a run that stalls says where: the step's probe events, the two streams, every thread's stack
        import faulthandler
        import threading
        trainer.probes = {}
        inner_step = trainer.step

        def probed_step(*args, **kw):
            trainer.probes = {}
            return inner_step(*args, **kw)
        trainer.step = probed_step

        def watch():
            time.sleep(float(os.environ.get("AIG_BENCH_TRACE_LIMIT", "90")))
            out = {"rank": rank, "global_step": trainer.global_step, "captures": trainer.captures,
                   "main_idle": torch.cuda.default_stream(dev).query(), "side_idle": trainer.side_stream.query(),
                   "probes": {k: v.query() for k, v in dict(trainer.probes or {}).items()}}
            print("[bench watchdog] " + json.dumps(out), file=sys.stderr, flush=True)
            faulthandler.dump_traceback(all_threads=True)
            os._exit(3)
        threading.Thread(target=watch, daemon=True).start()

    def run_steps(count, start=0, only=None):
        edges = torch.zeros((), dtype=torch.long, device=dev)
        for i in range(count):
            trainer.step(batches[(start + i) % nb] if only is None else only)
            edges += trainer._edges_last          # accumulated on the device: no host read inside the loop
        return edges

    # Set-up, not measured: one pass over the rank's minibatches, as the first epoch of a run - it builds what
    # train.py keeps per dataset sample (the template and its generator plan) and the one recording of the critic
    # terms for this minibatch shape.  Then the W warm-up steps, then the K timed steps, all in rotation.
    phase("set-up pass")
    run_steps(nb)
    phase("set-up pass enqueued")
    # what exists now lives as long as the run (dataset, templates, plans, recordings): taken out of the cyclic
    # collector's sight, as gan_amd.train.train does after its first epoch's set-up - a full collection over these
    # ~270k objects is a 100 ms stall in the middle of some step otherwise
    import gc
    gc.collect()
    gc.freeze()
    run_steps(a.warmup, start=nb)
    barrier()
    _abi.timeline.reset()
    lib = _abi.lib()
    t0 = time.perf_counter()
    edges_dev = run_steps(a.steps, start=nb + a.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    recordings = trainer.captures
    # the same step count on ONE minibatch over and over (every per-object cache hot): the figure a loop that
    # never changes its input would report, kept beside the rotating one
    phase("headline loop done; repeated-batch loop")
    run_steps(3, only=batches[0])      # (the allocator settles into the new pattern of requests first)
    barrier()
    r0 = time.perf_counter()
    run_steps(a.steps, only=batches[0])
    barrier()
    ms_repeated = (time.perf_counter() - r0) * 1e3 / a.steps
    # The same rotation with the generator step's unused backward pass left out (``Trainer.skip_unused_backward``:
    # train.py:298 back-propagates loss_G into the critic, whose gradients train.py:253 zeroes before their next
    # use).  An extension reported beside the headline, never as it: the headline step does all the reference does.
    phase("loop without the unused backward pass")
    trainer.skip_unused_backward = True
    for terms in trainer._captured.values():
        terms.release()
    trainer._captured.clear()
    run_steps(nb)
    run_steps(a.warmup, start=nb)
    barrier()
    u0 = time.perf_counter()
    run_steps(a.steps, start=nb + a.warmup)
    barrier()
    ms_skip = (time.perf_counter() - u0) * 1e3 / a.steps
    trainer.skip_unused_backward = False
    for terms in trainer._captured.values():
        terms.release()
    trainer._captured.clear()
    run_steps(nb)                      # (the full recordings again for what follows)
    # Which kernel is the roofline entry about?  The one with the largest share of the step's kernel time among the
    # kernels whose algorithmic cost is known (HBM_KERNELS, MFMA_KERNELS) - by THIS run's timings: a few untimed
    # steps with every critic pass launched eagerly (replayed launches cannot carry events) and every such kernel
    # bracketed by events of its own.
    phase("kernel census")
    census, census_steps = {}, min(3, a.steps)
    names_lib = timed_kernel_names(lib)
    all_mask = (1 << len(names_lib)) - 1
    if not a.no_kernel_timing:
        trainer.graph_capture = False
        run_steps(2)
        barrier()
        lib.aig_kernel_timing(all_mask)
        run_steps(census_steps, start=2)
        barrier()
        census = read_kernel_times(lib)
        lib.aig_kernel_timing(0)
        trainer.graph_capture = True
    modelled = [k for k in HBM_KERNELS + MFMA_KERNELS if k in census]
    roofline_kernel = max(modelled, key=lambda k: census[k][0]) if modelled else None
    # The roofline loop.  The headline loop above replays the critic's passes from recordings, and a replayed
    # kernel cannot carry events of its own.  So the same K steps are timed once more with the critic's pass over
    # the generated graphs left unrecorded (--capture-fake off: eager launches), and inside THAT timed region the
    # library brackets the roofline kernel - the kernel alone, by events its own dispatch carries
    # (aig_kernel_timing) - and notes the sizes of the aggregate's launches.  Bracketing every entry point slowed
    # the step by 10-50 %, so only this one kernel carries events here.
    phase("roofline loop")
    ms_eager_side = None
    kt_fake = {}
    if roofline_kernel is not None:
        trainer.capture_fake = "off"
        for terms in trainer._captured.values():
            terms.release()
        trainer._captured.clear()
        run_steps(nb)                   # records the real-side and penalty terms anew, without the generated side
        run_steps(a.warmup, start=nb)
        barrier()
        _abi.timeline.sizes_of = {"aig_gat_fwd"}
        lib.aig_kernel_timing(1 << names_lib.index(roofline_kernel))
        e0 = time.perf_counter()
        run_steps(a.steps, start=nb + a.warmup)
        barrier()
        ms_eager_side = (time.perf_counter() - e0) * 1e3 / a.steps
        kt_fake = read_kernel_times(lib)
        lib.aig_kernel_timing(0)
        _abi.timeline.sizes_of = None
    phase("instrumented steps")
    tl_meta = dict(_abi.timeline.meta)
    # per-entry-point GPU time from a few extra, untimed, fully instrumented steps
    inst_steps = min(3, a.steps)
    _abi.timeline.reset()
    _abi.timeline.only = None
    _abi.timeline.enabled = True
    lib.aig_kernel_timing(0 if a.no_kernel_timing else all_mask)  # here every modelled kernel carries events
    run_steps(inst_steps)
    barrier()
    kt_fake_all = read_kernel_times(lib)
    lib.aig_kernel_timing(0)
    _abi.timeline.enabled = False
    tl_all = _abi.timeline.totals_ms()
    batch = batches[0]

    edges_rank = int(edges_dev.item())
    stat = torch.tensor([elapsed, float(edges_rank)], dtype=torch.float64, device=dev)
    if dist.is_initialized():      # (also with a one-rank communicator: AIG_DP_FORCE=1, see gan_amd/parallel.py)
        tmax = stat[:1].clone()
        parallel.all_reduce(tmax, op=dist.ReduceOp.MAX)
        esum = stat[1:].clone()
        parallel.all_reduce(esum, op=dist.ReduceOp.SUM)
        elapsed, edges_total = float(tmax.item()), float(esum.item())
    else:
        edges_total = float(edges_rank)

    phase("sampling")
    # ---- sampling throughput (generator + repair + resync), same templates, fresh z per sample
    templates = trainer.templates_of(batch)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)

    draw_z = lambda: torch.randn(128, device=dev, generator=gen)   # noqa: E731
    nodes_per_iter = sum(int(t.x.size(0)) for t in templates)

    def sample_rate(run, per_iter=1):
        run(1)
        barrier()
        s0 = time.perf_counter()
        run(a.sample_iters)
        barrier()
        s_stat = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device=dev)
        if dist.is_initialized():
            parallel.all_reduce(s_stat, op=dist.ReduceOp.MAX)
        return nodes_per_iter * per_iter * a.sample_iters * world / float(s_stat.item())

    def one_at_a_time(iters):
        for _ in range(iters):
            for t in templates:
                repair_and_resync(trainer.G(t, draw_z()))

    sample_streams = [torch.cuda.Stream(device=dev) for _ in range(a.sample_in_flight)]

    def in_flight(iters):   # bulk sampling as gan_amd.sample does it: several independent samples in the air
        for _ in sample_many(trainer.G, ((t, draw_z) for _ in range(iters) for t in templates), streams=sample_streams):
            pass

    from gan_amd.sample import block_size, sample_sets
    S_set = min(block_size(int(t.x.size(0)), 64) for t in templates)

    def as_sets(iters):     # bulk sampling of small templates as gan_amd.sample does it: blocks of draws as one set of graphs
        for t in templates:
            for _ in sample_sets(trainer.G, t, (torch.stack([draw_z() for _ in range(S_set)]) for _ in range(iters))):
                pass

    # the matrix-core roofline is read here, where nothing else shares the GPU with the projection
    # kernel (inside a training step the generator pass runs next to the replayed critic term)
    _abi.timeline.reset()
    _abi.timeline.only = {"aig_gen_node_proj"}
    _abi.timeline.enabled = True
    sample_nodes_per_sec_single = sample_rate(one_at_a_time)
    _abi.timeline.enabled = False
    # ... with one row per NODE forced: the sampling and training paths project one row per class of equal feature rows
    # (a handful per template), which says nothing about the kernel
    _abi.timeline.reset()
    _abi.timeline.only = {"aig_gen_node_proj"}
    _abi.timeline.enabled = True
    limit_was, trainer.G.class_limit = trainer.G.class_limit, 0
    for _ in range(max(a.sample_iters, 2)):
        for t in templates:
            trainer.G(t, draw_z())
    trainer.G.class_limit = limit_was
    barrier()
    _abi.timeline.enabled = False
    tl_proj = _abi.timeline.totals_ms()
    sample_nodes_per_sec_streams = sample_rate(in_flight)
    sample_nodes_per_sec = sample_rate(as_sets, S_set) if S_set > 1 else sample_nodes_per_sec_streams

    # the same kernels on the REAL graphs (inside the step they only run as part of the replayed recording, which
    # carries no events): a few eager critic passes over the first minibatch, outside the timed region
    kt_real = {}
    if rank == 0 and not a.no_kernel_timing:
        real0 = trainer._captured_terms(batches[0])
        if real0 is not None:
            real0.load(batches[0])
            lib.aig_kernel_timing(all_mask)
            for _ in range(5):
                trainer.D.zero_grad()
                trainer.D(real0.real_for_D).sum().backward()
            torch.cuda.synchronize()
            kt_real = read_kernel_times(lib)
            lib.aig_kernel_timing(0)
            trainer.D.zero_grad()

    if rank == 0:
        sizes = tl_meta.get("aig_gat_fwd", [])
        n_avg = sum(n for n, _ in sizes) / max(len(sizes), 1)
        e_avg = sum(e for _, e in sizes) / max(len(sizes), 1)
        peaks = measured_peaks()
        shares = rocprof_shares()

        census_total = sum(ms for ms, _ in census.values())

        def kernel_line(kernel, kind, times, where):
            ms, launches = times[kernel]
            us = ms * 1e3 / launches
            nn, ee = int(n_avg), int(e_avg)
            line = {"kernel": kernel, "graph_kind": kind, "measured": where, "launches": launches, "avg_launch_us": round(us, 2),
                    "rocprof_share_of_gpu_time_pct": shares.get(kernel)}
            if kernel in census:      # this run's own census: ms per step of this kernel, share among the timed kernels
                line["census_ms_per_step"] = round(census[kernel][0] / census_steps, 4)
                line["census_launches_per_step"] = round(census[kernel][1] / census_steps, 1)
            bytes_alg = kernel_bytes(kernel, nn, ee, top + 1)
            if kernel in MFMA_KERNELS:
                flops = kernel_flops(kernel, nn)
                tf = flops / (us * 1e-6) / 1e12
                gbs = bytes_alg / (us * 1e-6) / 1e9
                line.update({"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                             "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": pmc_traffic(a.workload, kernel, kind),
                             "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": bytes_alg,
                             # the product sits at the ridge (32 flop per byte of an [n,128] f32 operand): its memory side
                             "hbm_achieved_GBs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)})
                return line
            if kernel == "k_pool_walk" and kind == "fake":
                # the readout only reads the rows of nodes that carry a level, and on a generated graph that is
                # the part the repair's walk reaches from the PIs (counters: a fifth of the rows): its
                # algorithmic bytes are not known on the host, so no fraction is claimed for it
                line.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                             "traffic": pmc_traffic(a.workload, kernel, kind), "algorithmic_bytes_per_launch": None})
                return line
            achieved = bytes_alg / (us * 1e-6) / 1e9
            if kernel in ("k_linear128", "k_wgrad128"):
                tf = kernel_flops(kernel, nn) / (us * 1e-6) / 1e12
                line.update({"f32_equivalent_TFLOPs": round(tf, 1), "f32_matrix_peak": MFMA_F32_PEAK_TF,
                             "arithmetic": "six bf16 piece products per f32 product on the bf16 matrix cores, f32 accumulate"})
            line.update({"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(a.workload, kernel, kind),
                         "algorithmic_bytes_per_launch": bytes_alg})
            if line["traffic"] and peaks.get("hbm_read_GBs"):
                # memory-side view: bytes that left the L2 (counters) per second of kernel time, against the read
                # bandwidth a plain streaming kernel reaches on this GPU (tools/peaks.hip)
                side = line["traffic"] / (us * 1e-6) / 1e9
                line.update({"traffic_GBs": round(side, 1), "peak_measured": peaks["hbm_read_GBs"],
                             "traffic_frac_of_measured": round(side / peaks["hbm_read_GBs"], 4)})
            return line

        roof_all = []
        every = HBM_KERNELS + MFMA_KERNELS
        if sizes:
            roof_all += [kernel_line(k, "fake", kt_fake, f"events carried by the kernel's own dispatch, eager launches inside a timed region of "
                                     f"{a.steps} steps run with --capture-fake off ({ms_eager_side:.3f} ms/step; the headline loop replays "
                                     "these launches from a recording, which cannot carry events)")
                         for k in every if k in kt_fake]
            roof_all += [kernel_line(k, "fake", kt_fake_all, f"events around the kernel, eager launches of {inst_steps} extra "
                                     "instrumented steps outside the timed region")
                         for k in every if k in kt_fake_all and k not in kt_fake]
            roof_all += [kernel_line(k, "real", kt_real, "events around the kernel, 5 eager critic passes outside the timed region")
                         for k in every if k in kt_real]
        # the headline entry: the kernel with the largest share of the step's kernel time among the kernels with a known
        # algorithmic cost, by this run's census; its figure comes from the launches inside the timed region above
        in_region = [r for r in roof_all if r["kernel"] == roofline_kernel and r["kernel"] in kt_fake and r["graph_kind"] == "fake"]
        roof = in_region[0] if in_region else None
        if roof is not None:
            roof = dict(roof, chosen_by=f"largest share of the step's kernel time among the kernels with a known algorithmic cost, "
                                        f"from {census_steps} fully eager steps of this run: " +
                                        ", ".join(f"{k} {census[k][0] / census_steps:.3f} ms/step" for k in
                                                  sorted(modelled, key=lambda k: -census[k][0])))
        # the whole step against the memory and the matrix cores: algorithmic bytes / flops of every pass of the step
        # (step_cost_model) over the headline step time
        e_real = edges_rank / max(a.steps, 1) - e_avg if sizes else 0.0
        # what the penalty term and the generator really ran over in the last step: distinct rows, classes of equal nodes
        terms_now = next(iter(trainer._captured.values()), None) if getattr(trainer, "_captured", None) else None
        gp_rows = int(terms_now.gp_rows.n_rows.item()) if terms_now is not None and terms_now.gp_rows is not None else 0
        plan_cs = [getattr(t.__dict__.get("_gen_plan"), "C", 0) for t in templates]
        classes = max(plan_cs) if plan_cs and trainer.G.class_limit and 0 < max(plan_cs) <= trainer.G.class_limit else 0
        sb, sf, items = step_cost_model(int(n_avg) if sizes else nodes_per_iter, int(max(e_real, 0)), int(e_avg), top + 1, 64,
                                        targs.n_critic, gp_rows=gp_rows, classes=classes)
        step_s = elapsed / a.steps
        roofline_step = {"algorithmic_bytes_per_step": int(sb), "flops_per_step": sf, "achieved": round(sb / step_s / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sb / step_s / 1e9 / HBM_PEAK_GBS, 4),
                         "achieved_TFLOPs": round(sf / step_s / 1e12, 2), "frac_of_f32_matrix_peak": round(sf / step_s / 1e12 / MFMA_F32_PEAK_TF, 4),
                         "items": items, "penalty_rows": gp_rows or None, "generator_classes": classes or None,
                         "note": "bytes and flops of the step's passes by the algorithm (bench.py: step_cost_model) over ms_per_step; "
                                 "the penalty term is counted over its distinct rows and the generator over its classes of equal "
                                 "feature rows (penalty_rows, generator_classes: what the last step ran over) - bytes the reference's "
                                 "formulation would move over every node are not credited; "
                                 "the step is the sum of its wide kernels' durations, each at 0.4-0.9 of its own bound"}
        # the dense contraction of the path: node projections of the generator (7 products of
        # [n,128]x[128,128] per sample) on the f32 matrix cores, flops executed / event time
        mfma = None
        ms_proj, n_proj = tl_proj.get("aig_gen_node_proj", (0.0, 0))
        if n_proj:
            flops = 7 * 2.0 * 128 * 128 * nodes_per_iter * (n_proj / max(len(templates), 1))   # one launch per template
            tf = flops / (ms_proj * 1e-3) / 1e12
            mfma = {"kernel": "k_node_mlp3 + k_node_proj4 (entry point aig_gen_node_proj) with one row per node forced "
                              "(class_limit = 0): the kernel's own rate, outside the timed region - the step and the sampling "
                              "rates project one row per class of equal feature rows", "bound": "mfma", "achieved": round(tf, 1),
                    "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
                    "launches": n_proj, "avg_launch_us": round(ms_proj * 1e3 / n_proj, 2)}
            if measured_peaks().get("mfma_f32_TFs"):
                mfma.update({"peak_measured": measured_peaks()["mfma_f32_TFs"],
                             "frac_of_measured": round(tf / measured_peaks()["mfma_f32_TFs"], 4)})
        out = {
            "metric": "train edges/sec (AIG nodes/sec sampled at k=64 in sample_nodes_per_sec)",
            "value": round(edges_total / elapsed, 1), "unit": "edges/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed * 1e3 / a.steps, 3),
            "ms_per_step_repeated_batch": round(ms_repeated, 3),
            "ms_per_step_generated_side_eager": None if ms_eager_side is None else round(ms_eager_side, 3),
            "ms_per_step_without_unused_backward": round(ms_skip, 3),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {desc}; {nb} different minibatches per GPU in rotation inside the timed "
                                   "region; candidate-k 64, n_critic 3, hidden 128, fp32",
                       "distinct_minibatches_per_gpu": nb, "critic_term_recordings": recordings,
                       "edges_per_step_per_gpu": round(edges_rank / a.steps, 1),
                       "nodes_per_step_per_gpu": nodes_per_iter, "parallelism": f"dp{world}",
                       # ranks of the RCCL communicator the gradient exchange went through (0: no communicator; a
                       # rehearsal over gloo on shared devices is named as such)
                       "rccl_ranks": (dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0),
                       "exchange_backend": dist.get_backend() if dist.is_initialized() else None,
                       "gradient_exchanges_per_step": targs.n_critic if trainer_exchanges else 0},
            "sample_nodes_per_sec": round(sample_nodes_per_sec, 1),
            "sample_nodes_per_sec_streams": round(sample_nodes_per_sec_streams, 1),
            "sample_nodes_per_sec_one_at_a_time": round(sample_nodes_per_sec_single, 1),
            "sample_note": f"G.forward + enforce + resync per sample (gan_amd.constraints.repair_and_resync: same tensors as "
                           "the two calls in a row, without the repair's own depth walk that the resync overwrites); "
                           f"bulk rate as gan_amd.sample runs it (blocks of {S_set} draws per template as one set of graphs), "
                           f"the rate with {a.sample_in_flight} independent samples in flight "
                           "on as many streams, and the rate of strictly one sample after another",
            "roofline": roof,
            "roofline_step": roofline_step,
            "roofline_kernels": roof_all,
            "kernel_census_ms_per_step": {k: round(v[0] / census_steps, 4) for k, v in sorted(census.items(), key=lambda kv: -kv[1][0])},
            "roofline_note": "bytes of the gathers count once per list entry in `achieved` (algorithmic), so a graph whose edges "
                             "hang on a few sources - what the untrained generator wires, kind 'fake' - is served by the caches and "
                             "can exceed what the memory delivers; `traffic` is what left the L2 (counter passes, per kind)",
            "roofline_mfma": mfma,
            "timeline_ms_per_step": {k: round(v[0] / inst_steps, 3) for k, v in sorted(tl_all.items())},
            "timeline_note": f"GPU ms per step by entry point, from {inst_steps} extra instrumented steps outside the timed region "
                             "(event brackets on the launching stream; entry points that run next to another stream's work include that contention)",
        }
        if world == 1 and not a.no_side_workloads and a.workload != "iscas85":
            # BASELINE.json configs[1] (ISCAS85-sized graphs, one minibatch of all of them): bound by launches
            # and Python, not by the GPU - reported for completeness, the roofline figures above are not its
            del trainer, batch, batches, templates
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["other_workloads"] = [side_workload("iscas85", dev, steps=10, warmup=5)]
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
