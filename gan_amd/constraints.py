"""AIG structural constraints on MI355X kernels.

Drop-in for the three live functions of models/constraints.py of the reference, plus the
attribute resync of its sampling CLI:

* ``enforce_aig_constraints(data, strict=False, check_all_on_path=True) -> Data``
  (constraints.py:508-669) - repairs in place and returns the same object;
* ``aig_constraint_penalties(data, weights=None) -> Dict[str, Tensor]`` (:253-366);
* ``validate_strict_aig(data, check_all_on_path=True) -> List[str]`` (:375-505);
* ``resync_attrs(data) -> Data`` (AIGgan.py:29-62).

The reference walks Python lists node by node (its repair is quadratic in the node count);
here each function is a handful of linear-time integer kernels - radix sorts for grouping and
de-duplication, per-target selection, level-synchronous traversals - and the results are the
same integers.  The graph must live on a HIP device; there is no CPU path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import _abi
from .data import Data
from .graphset import GraphSet

PI, PO, AND = 0, 1, 2

PENALTY_KEYS = ["pi_in", "po_out", "backedge", "selfloop", "dup_fanin", "indeg_mse", "fanin_type",
                "reach_violation"]
_DEFAULT_WEIGHTS = {"pi_in": 1.0, "po_out": 1.0, "backedge": 2.0, "selfloop": 3.0, "dup_fanin": 1.0,
                    "indeg_mse": 1.0, "fanin_type": 1.0, "reach_violation": 0.5}


def _need_gpu(t: torch.Tensor, who: str) -> None:
    if t.device.type != "cuda":
        raise _abi.AigAbiError(f"{who} needs the graph on a HIP device (no CPU path exists)")


def _x32(x: torch.Tensor) -> torch.Tensor:
    return x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()


def _edges(data: Data):
    ei = getattr(data, "edge_index", None)
    if ei is None or ei.numel() == 0:
        return None, 0
    return ei.to(torch.int64).contiguous(), int(ei.size(1))


def _inv64(data: Data, E: int) -> Optional[torch.Tensor]:
    ea = getattr(data, "edge_attr", None)
    if ea is None or ea.numel() == 0:
        return None
    return ea.reshape(-1)[:E].to(torch.int64).contiguous()   # int(edge_attr[i].item())


def _depth_arg(data: Data, n: int, dev):
    nd = getattr(data, "node_depth", None) if hasattr(data, "node_depth") else None
    if nd is None or nd.numel() != n:
        return 0, None
    nd = nd.to(dev)
    if nd.dtype.is_floating_point:
        return 1, nd.to(torch.float32).contiguous()
    return 2, nd.to(torch.int64).contiguous()


# ------------------------------------------------------------------------------------------ enforce
def enforce_aig_constraints(data: Data, strict: bool = False, check_all_on_path: bool = True) -> Data:
    """Repair ``data`` into AIG shape; mutates and returns it.  ``strict`` has no effect in the
    reference either once back edges are pruned (constraints.py:630-638 re-applies the same filter)."""
    return enforce_finish(enforce_begin(data, strict, check_all_on_path))


class EnforceHandle:
    """A repair whose kernels are enqueued but whose edge count has not been read back yet."""
    __slots__ = ("data", "out_ei", "out_ea", "node_depth", "x_out", "count_host", "ready")


def enforce_begin(data: Data, strict: bool = False, check_all_on_path: bool = True,
                  want_depth: bool = True) -> EnforceHandle:
    """First half of ``enforce_aig_constraints``: enqueues the repair on the current stream and starts
    the copy of the repaired edge count to the host.  A caller that keeps several graphs in flight
    (bulk sampling) enqueues other work before it calls ``enforce_finish``.

    ``want_depth=False`` leaves out the breadth-first walk behind ``node_depth`` (the repaired graph then
    carries ``node_depth = None``): for the sampling path, where ``resync_attrs`` follows and replaces
    ``node_depth`` and ``x[:, 1]`` (AIGgan.py:105-110) - see ``repair_and_resync``."""
    lib = _abi.lib()
    h = EnforceHandle()
    h.data, h.ready = data, None
    x = data.x
    _need_gpu(x, "enforce_aig_constraints")
    dev = x.device
    n = int(x.size(0))
    ei, E = _edges(data)
    if E == 0:  # constraints.py:517-522
        data.edge_index = torch.zeros((2, 0), dtype=torch.long, device=dev)
        data.edge_attr = torch.zeros((0, 1), dtype=torch.long, device=dev)
        data.node_depth = torch.full((n,), -1, dtype=torch.long, device=dev)
        data.x[:, 1] = 0
        return h
    h.node_depth = None
    x_in = _x32(x)
    inv = _inv64(data, E)
    kind, depth = _depth_arg(data, n, dev)
    cap = 2 * n
    out_ei = torch.empty((2, cap), dtype=torch.int64, device=dev)
    out_ea = torch.empty((cap,), dtype=torch.int64, device=dev)
    out_cnt = torch.empty(1, dtype=torch.int32, device=dev)
    node_depth = torch.empty(n, dtype=torch.int64, device=dev) if want_depth else None
    x_out = torch.empty((n, 2), dtype=torch.float32, device=dev)
    ws = _abi.workspace(lib.aig_enforce_ws_bytes(n, E), dev)
    _abi.check(lib.aig_enforce(n, E, _abi.ptr(x_in), _abi.ptr(ei), _abi.ptr(inv), kind, _abi.ptr(depth),
                               _abi.ptr(out_ei), cap, _abi.ptr(out_ea), _abi.ptr(out_cnt), _abi.ptr(node_depth),
                               _abi.ptr(x_out), _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_enforce")
    h.out_ei, h.out_ea, h.node_depth, h.x_out = out_ei, out_ea, node_depth, x_out
    h.count_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    h.count_host.copy_(out_cnt, non_blocking=True)
    h.ready = torch.cuda.Event()
    h.ready.record()
    return h


def enforce_finish(h: EnforceHandle) -> Data:
    """Second half: waits for the edge count (the one host read of the repair - it sizes the returned
    tensors) and installs the repaired attributes.  Call it under the stream ``enforce_begin`` ran on."""
    data = h.data
    if h.ready is None:
        return data
    h.ready.synchronize()
    m = int(h.count_host[0])
    x = data.x
    data.edge_index = h.out_ei[:, :m].contiguous()
    data.edge_attr = h.out_ea[:m].reshape(-1, 1).contiguous()
    data.node_depth = h.node_depth
    data.x = h.x_out if x.dtype == torch.float32 else h.x_out.to(x.dtype)
    return data


# ------------------------------------------------------------------------------------------ penalties
def repair_and_resync(data: Data) -> Data:
    """``resync_attrs(enforce_aig_constraints(data))`` (AIGgan.py:105-110) without the walk whose result
    the resync overwrites: same tensors, one level-by-level traversal instead of two."""
    return resync_attrs(enforce_finish(enforce_begin(data, want_depth=False)))


def aig_constraint_penalties(data: Data, weights: Optional[Dict[str, float]] = None) -> Dict[str, torch.Tensor]:
    """Violation counts as 0-dim float32 device tensors (no host synchronisation)."""
    lib = _abi.lib()
    x = data.x
    _need_gpu(x, "aig_constraint_penalties")
    dev = x.device
    n = int(x.size(0))
    ei, E = _edges(data)
    if E == 0 or n == 0:
        zero = torch.tensor(0.0, device=dev)
        terms = {k: zero for k in PENALTY_KEYS}
        terms["total"] = zero
        return terms
    out = torch.empty(9, dtype=torch.float32, device=dev)
    ws = _abi.workspace(lib.aig_penalties_ws_bytes(n, E), dev)
    _abi.check(lib.aig_penalties(n, E, _abi.ptr(_x32(x)), _abi.ptr(ei), _abi.ptr(out), _abi.ptr(ws), ws.numel(),
                                 _abi.stream()), "aig_penalties")
    terms = {k: out[i] for i, k in enumerate(PENALTY_KEYS)}
    if weights:
        w = dict(_DEFAULT_WEIGHTS)
        w.update(weights)
        terms["total"] = sum(w[k] * terms[k] for k in PENALTY_KEYS)
    else:
        terms["total"] = out[8]
    return terms


# ------------------------------------------------------------------------------------------ validate
def _violation_messages(types, src, dst, eflag, ncode, indeg, pred2, on_path, has_cycle, check_all_on_path):
    """Message list in the reference's order and wording (constraints.py:398-503)."""
    errs: List[str] = []
    unknown = np.flatnonzero(ncode == 1).tolist()
    if unknown:
        errs.append(f"Unknown node_type indices: {unknown[:10]} ...")
    for bit, text in ((1, "Self-loops not allowed"), (2, "Duplicate edges found"), (4, "Edges into PI not allowed"),
                      (8, "Edges out of PO not allowed"), (16, "Edges with invalid src type")):
        hit = np.flatnonzero(eflag & bit)
        if hit.size:
            examples = [(int(src[e]), int(dst[e])) for e in hit[:5]]
            errs.append(f"{text}: examples {examples} ...")
    for v in np.flatnonzero((types == PO) & (ncode == 2)).tolist():
        errs.append(f"[PO] node {v} indegree={int(indeg[v])} (must be 1)")
    for v in np.flatnonzero((types == AND) & (ncode >= 3)).tolist():
        if ncode[v] == 3:
            errs.append(f"[AND] node {v} indegree={int(indeg[v])} (must be 2)")
        else:
            errs.append(f"[AND] node {v} duplicated fanin {[int(pred2[2 * v]), int(pred2[2 * v + 1])]}")
    if has_cycle:
        errs.append("Graph contains a cycle (fails DAG).")
    if check_all_on_path:
        off = np.flatnonzero(on_path == 0).tolist()
        if off:
            errs.append(f"Dangling/uncovered nodes (not on PI->PO path): {off[:10]} ... (count={len(off)})")
    return errs


def validate_strict_aig(data: Data, check_all_on_path: bool = True) -> List[str]:
    """Violation messages (empty list = valid).  The verdicts are computed on the GPU; this
    function copies them to the host once and words them."""
    lib = _abi.lib()
    x = data.x
    if x is None or x.size(1) == 0:
        return ["data.x missing or empty"]
    _need_gpu(x, "validate_strict_aig")
    dev = x.device
    n = int(x.size(0))
    ei, E = _edges(data)
    types = x[:, 0].long().cpu().numpy()
    if E == 0 or n == 0:
        # no edge: only type, indegree and coverage rules can fire
        z8 = np.zeros(0, np.uint8)
        ncode = np.where(~np.isin(types, (PI, PO, AND)), 1, np.where(types == PO, 2, np.where(types == AND, 3, 0)))
        on_path = np.zeros(n, np.uint8)
        return _violation_messages(types, z8, z8, z8, ncode.astype(np.uint8), np.zeros(n, np.int64),
                                   np.full(2 * n, -1), on_path, False, check_all_on_path)
    u8 = dict(dtype=torch.uint8, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    eflag, ncode, on_path = torch.empty(E, **u8), torch.empty(n, **u8), torch.empty(n, **u8)
    indeg, pred2, summary = torch.empty(n, **i32), torch.empty(2 * n, **i32), torch.empty(2, **i32)
    ws = _abi.workspace(lib.aig_validate_ws_bytes(n, E), dev)
    _abi.check(lib.aig_validate(n, E, _abi.ptr(_x32(x)), _abi.ptr(ei), _abi.ptr(eflag), _abi.ptr(ncode),
                                _abi.ptr(indeg), _abi.ptr(pred2), _abi.ptr(on_path), _abi.ptr(summary),
                                _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_validate")
    ei_h = ei.cpu().numpy()
    return _violation_messages(types, ei_h[0], ei_h[1], eflag.cpu().numpy(), ncode.cpu().numpy(),
                               indeg.cpu().numpy(), pred2.cpu().numpy(), on_path.cpu().numpy(),
                               bool(summary[0].item()), check_all_on_path)


# ------------------------------------------------------------------------------------------ resync
# Graphs of at least this many nodes get their resync levels from the sweep over positions (aig_resync_typed) when it
# applies; below, the level walk of a small graph (a few dozen levels, one workgroup per graph) costs less than the
# sweep's extra sort and set-up launches (bulk sampling of 2.4k-node templates: 1.46 vs 1.58 ms per block of 64).
SWEEP_MIN_NODES = 16384


def resync_attrs(d: Data) -> Data:
    """x[:,1] = number of inverted in-edges, node_depth = longest-path level (zeros on a cycle)."""
    lib = _abi.lib()
    _need_gpu(d.x, "resync_attrs")
    dev = d.x.device
    n = int(d.x.size(0))
    ei, E = _edges(d)
    if d.x.size(1) < 2:
        d.x = torch.cat([d.x, torch.zeros((n, 2 - d.x.size(1)), dtype=d.x.dtype, device=dev)], dim=1)
    if E == 0 or n == 0:
        d.x[:, 1] = 0
        d.node_depth = torch.zeros(n, dtype=torch.long, device=dev)
        return d
    inv_cnt = torch.empty(n, dtype=torch.int64, device=dev)
    depth = torch.empty(n, dtype=torch.int64, device=dev)
    x = d.x
    if n >= SWEEP_MIN_NODES and x.dtype == torch.float32 and x.size(1) == 2 and x.is_contiguous():
        # with the node types at hand a graph whose edges all run upward in PI / AND / PO order (a repaired one)
        # gets its levels from a sweep over that order; any other graph from the level walk, as without x
        ws = _abi.workspace(lib.aig_resync_typed_ws_bytes(n, E), dev)
        _abi.check(lib.aig_resync_typed(n, E, _abi.ptr(x), _abi.ptr(ei), _abi.ptr(_inv64(d, E)), _abi.ptr(inv_cnt),
                                        _abi.ptr(depth), _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_resync_typed")
    else:
        ws = _abi.workspace(lib.aig_resync_ws_bytes(n, E), dev)
        _abi.check(lib.aig_resync(n, E, _abi.ptr(ei), _abi.ptr(_inv64(d, E)), _abi.ptr(inv_cnt), _abi.ptr(depth),
                                  _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_resync")
    d.x[:, 1] = inv_cnt
    d.node_depth = depth
    return d


# ------------------------------------------------------------------------------------------ sets of graphs
def _set_depth_arg(gs: GraphSet):
    nd = gs.node_depth
    if nd is None or nd.numel() not in (gs.n, gs.B * gs.n):
        return 0, None, 0
    stride = 0 if (nd.numel() == gs.n and gs.B > 1) else gs.n
    if nd.dtype.is_floating_point:
        return 1, nd.to(torch.float32).contiguous(), stride
    return 2, nd.to(torch.int64).contiguous(), stride


def enforce_set(gs: GraphSet, want_depth: bool = True) -> GraphSet:
    """``enforce_aig_constraints`` on every graph of the set, one call: returns a new set (``cap = 2n`` edge
    slots per graph, counts on the device - no host read) whose graphs equal the single-graph repair's."""
    lib = _abi.lib()
    _need_gpu(gs.x, "enforce_set")
    dev = gs.x.device
    B, n = gs.B, gs.n
    cap = 2 * n
    kind, depth, stride = _set_depth_arg(gs)
    out_ei = torch.empty((B, 2, cap), dtype=torch.int64, device=dev)
    out_ea = torch.empty((B, cap), dtype=torch.int64, device=dev)
    out_cnt = torch.empty(B, dtype=torch.int32, device=dev)
    node_depth = torch.empty(B * n, dtype=torch.int64, device=dev) if want_depth else None
    x_out = torch.empty((B * n, 2), dtype=torch.float32, device=dev)
    ws = _abi.workspace(lib.aig_enforce_batch_ws_bytes(B, n, gs.ecap), dev)
    _abi.check(_abi.timed("aig_enforce_batch", lib.aig_enforce_batch, B, n, gs.ecap, _abi.ptr(gs.count), _abi.ptr(gs.x),
                          _abi.ptr(gs.edge_index), _abi.ptr(gs.edge_attr), kind, _abi.ptr(depth), stride,
                          _abi.ptr(out_ei), cap, _abi.ptr(out_ea), _abi.ptr(out_cnt), _abi.ptr(node_depth),
                          _abi.ptr(x_out), _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_enforce_batch")
    return GraphSet(B, n, cap, x_out, out_ei, out_ea, out_cnt, node_depth, x_dtype=gs.x_dtype)


def clean_depth_set(gs: GraphSet) -> torch.Tensor:
    """The ``node_depth`` that ``enforce_set(gs)`` reports, by itself: for a caller that repairs with
    ``want_depth=False``, hands the repaired edges on at once and fills the levels in afterwards."""
    lib = _abi.lib()
    _need_gpu(gs.x, "clean_depth_set")
    dev = gs.x.device
    depth = torch.empty(gs.B * gs.n, dtype=torch.int64, device=dev)
    ws = _abi.workspace(lib.aig_enforce_batch_ws_bytes(gs.B, gs.n, gs.ecap), dev)
    _abi.check(_abi.timed("aig_clean_depth_batch", lib.aig_clean_depth_batch, gs.B, gs.n, gs.ecap, _abi.ptr(gs.count),
                          _abi.ptr(gs.x), _abi.ptr(gs.edge_index), _abi.ptr(depth), _abi.ptr(ws), ws.numel(), _abi.stream()),
               "aig_clean_depth_batch")
    return depth


def penalties_set(gs: GraphSet) -> torch.Tensor:
    """``aig_constraint_penalties`` (default weights) of every graph: float32 ``[B, 9]`` in the order of
    ``PENALTY_KEYS`` + total, on the device."""
    lib = _abi.lib()
    _need_gpu(gs.x, "penalties_set")
    dev = gs.x.device
    out = torch.empty((gs.B, 9), dtype=torch.float32, device=dev)
    ws = _abi.workspace(lib.aig_penalties_batch_ws_bytes(gs.B, gs.n, gs.ecap), dev)
    hint = getattr(gs, "order_hint", None)
    stride = 0
    if hint is not None:
        if hint.dtype != torch.int32 or not hint.is_cuda or hint.numel() not in (gs.n, gs.B * gs.n):
            hint = None          # (only a shortcut: anything else than what the kernel takes is left out)
        else:
            hint = hint.contiguous()
            stride = gs.n if hint.numel() == gs.B * gs.n and gs.B > 1 else 0
    _abi.check(_abi.timed("aig_penalties_batch", lib.aig_penalties_batch_hinted, gs.B, gs.n, gs.ecap, _abi.ptr(gs.count),
                          _abi.ptr(gs.x), _abi.ptr(gs.edge_index), _abi.ptr(hint), stride, _abi.ptr(out), _abi.ptr(ws),
                          ws.numel(), _abi.stream()), "aig_penalties_batch_hinted")
    return out


def resync_set(gs: GraphSet) -> GraphSet:
    """``resync_attrs`` on every graph of the set (in place: ``x[:, 1]`` and ``node_depth``)."""
    lib = _abi.lib()
    _need_gpu(gs.x, "resync_set")
    dev = gs.x.device
    N = gs.B * gs.n
    inv_cnt = torch.empty(N, dtype=torch.int64, device=dev)
    depth = torch.empty(N, dtype=torch.int64, device=dev)
    if gs.n >= SWEEP_MIN_NODES and gs.x.dtype == torch.float32 and gs.x.dim() == 2 and gs.x.size(1) == 2 and gs.x.is_contiguous():
        ws = _abi.workspace(lib.aig_resync_typed_batch_ws_bytes(gs.B, gs.n, gs.ecap), dev)
        _abi.check(_abi.timed("aig_resync_batch", lib.aig_resync_typed_batch, gs.B, gs.n, gs.ecap, _abi.ptr(gs.count),
                              _abi.ptr(gs.x), _abi.ptr(gs.edge_index), _abi.ptr(gs.edge_attr), _abi.ptr(inv_cnt),
                              _abi.ptr(depth), _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_resync_typed_batch")
    else:
        ws = _abi.workspace(lib.aig_resync_batch_ws_bytes(gs.B, gs.n, gs.ecap), dev)
        _abi.check(_abi.timed("aig_resync_batch", lib.aig_resync_batch, gs.B, gs.n, gs.ecap, _abi.ptr(gs.count),
                              _abi.ptr(gs.edge_index), _abi.ptr(gs.edge_attr), _abi.ptr(inv_cnt), _abi.ptr(depth),
                              _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_resync_batch")
    gs.x[:, 1] = inv_cnt
    gs.node_depth = depth
    return gs


def repair_and_resync_set(gs: GraphSet) -> GraphSet:
    """``repair_and_resync`` on every graph of the set: two calls, no host read in between."""
    return resync_set(enforce_set(gs, want_depth=False))


def pad_set(gs: GraphSet, max_edges: int):
    """Edge tensors of the padded minibatch made of the set's graphs - what ``pad_data`` on every graph and
    the batch concatenation give: ``edge_index [2, B*max_edges]`` (node offsets added, ``-1`` pad columns shifted
    like the rest) and ``edge_attr [B*max_edges, 1]``."""
    lib = _abi.lib()
    dev = gs.x.device
    ei = torch.empty((2, gs.B * max_edges), dtype=torch.int64, device=dev)
    ea = torch.empty((gs.B * max_edges, 1), dtype=torch.int64, device=dev)
    _abi.check(lib.aig_pad_batch(gs.B, gs.n, gs.ecap, _abi.ptr(gs.count), _abi.ptr(gs.edge_index), _abi.ptr(gs.edge_attr),
                                 int(max_edges), _abi.ptr(ei), _abi.ptr(ea), _abi.stream()), "aig_pad_batch")
    return ei, ea
