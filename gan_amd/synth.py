"""Seeded synthetic And-Inverter Graphs with the reference dataset's layout.

No ISCAS85 / EPFL file ships with the reference (its dataset directory is
git-ignored), so tests and benchmarks use graphs of the same shape built here:
node order PI | AND | PO, every AND with two distinct fan-ins drawn from the
``window`` preceding PI/AND nodes, every PO driven by one AND, inversion bits
~ Bernoulli(0.5).  The tensors follow data_files/generate_pt.py:45-61 of the
reference: ``x`` float32 [N,2] = (type, #inverted in-edges), ``edge_index``
int64 [2,E], ``edge_attr`` float32 [E,1], ``node_depth`` float32 [N].

``orientation="forward"`` emits fan-in -> gate edges with ``node_depth`` the
longest-path level from the PIs (what the generator emits, generator.py:199);
``orientation="dataset"`` emits gate -> fan-in edges with depth 0 at nodes
without incoming edges, i.e. the PO side (data_files/andAIG2Graphml.py:56,
generate_pt.py:9-22).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .data import Data

PI, PO, AND = 0, 1, 2

# (name, n_pi, n_po, n_and): AIG sizes shaped like the ISCAS85 suite after `strash`.
ISCAS85_SHAPES: List[Tuple[str, int, int, int]] = [
    ("c17", 5, 2, 6), ("c432", 36, 7, 208), ("c499", 41, 32, 400), ("c880", 60, 26, 327),
    ("c1355", 41, 32, 504), ("c1908", 33, 25, 414), ("c2670", 233, 140, 717),
    ("c3540", 50, 22, 1038), ("c5315", 178, 123, 1773), ("c6288", 32, 32, 2337),
    ("c7552", 207, 108, 2074),
]


def _longest_path_levels(n: int, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """Longest-path level of every node of a DAG whose edges go src -> dst."""
    depth = np.zeros(n, dtype=np.int64)
    if src.size == 0:
        return depth
    # relax until stable: at most (number of levels) sweeps, each vectorised
    while True:
        cand = depth[src] + 1
        new = depth.copy()
        np.maximum.at(new, dst, cand)
        if np.array_equal(new, depth):
            return depth
        depth = new


def synth_aig_arrays(n_pi: int, n_and: int, n_po: int, seed: int = 0, window: int = 4096,
                     orientation: str = "forward", compute_depth: bool = True) -> Dict[str, np.ndarray]:
    if n_pi < 2 and n_and > 0:
        raise ValueError("need at least 2 primary inputs to give an AND two distinct fan-ins")
    rng = np.random.default_rng(seed)
    n = n_pi + n_and + n_po
    types = np.concatenate([np.full(n_pi, PI), np.full(n_and, AND), np.full(n_po, PO)]).astype(np.int64)
    and_ids = np.arange(n_pi, n_pi + n_and, dtype=np.int64)
    lo = np.maximum(0, and_ids - window)
    span = and_ids - lo                                   # >= 2 candidates for every AND
    u1 = lo + (rng.random(n_and) * span).astype(np.int64)
    u2 = lo + (rng.random(n_and) * (span - 1)).astype(np.int64)
    u2 = np.where(u2 >= u1, u2 + 1, u2)                   # distinct from u1
    po_ids = np.arange(n_pi + n_and, n, dtype=np.int64)
    if n_and > 0:
        drv = n_pi + (rng.random(n_po) * n_and).astype(np.int64)
    else:
        drv = (rng.random(n_po) * n_pi).astype(np.int64)
    # fan-in -> gate, target-major: [u1_0, u2_0, u1_1, u2_1, ..., drv_0, ...]
    src = np.concatenate([np.stack([u1, u2], 1).reshape(-1), drv])
    dst = np.concatenate([np.repeat(and_ids, 2), po_ids])
    inv = (rng.random(src.size) < 0.5).astype(np.int64)
    inv_cnt = np.zeros(n, dtype=np.int64)
    np.add.at(inv_cnt, dst, inv)
    if orientation == "forward":
        e_src, e_dst = src, dst
    elif orientation == "dataset":
        e_src, e_dst = dst, src                            # gate -> fan-in
    else:
        raise ValueError(f"unknown orientation {orientation!r}")
    # (the level sweep is the slow part at a million nodes: a caller that takes depths from elsewhere skips it)
    depth = _longest_path_levels(n, e_src, e_dst) if compute_depth else np.zeros(n, dtype=np.int64)
    x = np.stack([types, inv_cnt], 1).astype(np.float32)
    return {
        "x": x,
        "edge_index": np.stack([e_src, e_dst], 0).astype(np.int64),
        "edge_attr": inv.astype(np.float32).reshape(-1, 1),
        "node_depth": depth.astype(np.float32),
    }


def synth_aig(n_pi: int, n_and: int, n_po: int, seed: int = 0, window: int = 4096,
              orientation: str = "forward", shuffle_ids: bool = False, compute_depth: bool = True) -> Data:
    """One synthetic AIG as a ``Data``.  ``shuffle_ids`` relabels nodes at random
    (types no longer grouped), which exercises the rank != index paths.  ``compute_depth=False`` leaves
    ``node_depth`` at zero."""
    a = synth_aig_arrays(n_pi, n_and, n_po, seed, window, orientation, compute_depth)
    if shuffle_ids:
        rng = np.random.default_rng(seed + 7919)
        n = a["x"].shape[0]
        perm = rng.permutation(n)               # new id of old node i
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        a["x"] = a["x"][inv]
        a["node_depth"] = a["node_depth"][inv]
        a["edge_index"] = perm[a["edge_index"]]
    return Data(x=torch.from_numpy(a["x"]), edge_index=torch.from_numpy(a["edge_index"]),
                edge_attr=torch.from_numpy(a["edge_attr"]), node_depth=torch.from_numpy(a["node_depth"]))


def iscas85_like(seed: int = 0, orientation: str = "dataset") -> List[Data]:
    """Eleven graphs with the node counts of c17 ... c7552."""
    out = []
    for i, (_, n_pi, n_po, n_and) in enumerate(ISCAS85_SHAPES):
        out.append(synth_aig(n_pi, n_and, n_po, seed=seed * 100 + i, window=max(8, n_and // 8 + n_pi),
                             orientation=orientation))
    return out
