"""TEST INFRASTRUCTURE ONLY - ctypes front end of the C/C++ CPU checker (liboracle.so).

Imported by tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; never by anything under ``gan_amd/``.  Every wrapper mirrors the
calling convention of the reference function it checks (cited per function) so
parity tests read like calls into the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc/g++ (seconds)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        for fn in ("orc_build_order", "orc_enforce", "orc_penalties", "orc_validate", "orc_resync",
                   "orc_topk", "orc_topk_rows", "orc_node_proj", "orc_node_proj_z", "orc_score_rows", "orc_inv_bits"):
            getattr(_lib, fn).restype = C.c_int
        _lib.orc_candidates.restype = C.c_int64
    return _lib


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def _p(a: Optional[np.ndarray], typ):
    return None if a is None else a.ctypes.data_as(typ)


def _types_of(x: torch.Tensor) -> np.ndarray:
    return _i64(x[:, 0].long().cpu().numpy())


def _depth_of(data, n: int) -> Optional[np.ndarray]:
    nd = getattr(data, "node_depth", None)
    if nd is None or nd.numel() != n:
        return None
    return np.ascontiguousarray(nd.detach().cpu().to(torch.float64).numpy())


# ----------------------------------------------------------------------------- generator pieces
def build_order(types: np.ndarray, depth: Optional[np.ndarray]):
    """models/generator.py:70-82."""
    n = int(types.shape[0])
    types = _i64(types)
    depth = None if depth is None else np.ascontiguousarray(depth, dtype=np.float64)
    order, rank = np.empty(n, np.int64), np.empty(n, np.int64)
    rc = lib().orc_build_order(C.c_int64(n), _p(types, i64p), _p(depth, f64p), _p(order, i64p), _p(rank, i64p))
    assert rc == 0
    return order, rank


def candidates(types: np.ndarray, order: np.ndarray, K: int):
    """models/generator.py:131-158 -> (tgt[T], cand[T,K] (-1 padded), cnt[T], fallback[T])."""
    n = int(types.shape[0])
    types, order = _i64(types), _i64(order)
    tgt = np.empty(n, np.int64)
    cand = np.full((max(n, 1), K), -1, np.int64)
    cnt, fb = np.empty(n, np.int64), np.empty(n, np.int64)
    T = lib().orc_candidates(C.c_int64(n), _p(types, i64p), _p(order, i64p), C.c_int64(K),
                             _p(tgt, i64p), _p(cand, i64p), _p(cnt, i64p), _p(fb, i64p))
    assert T >= 0
    return tgt[:T].copy(), cand[:T].copy(), cnt[:T].copy(), fb[:T].copy()


def topk(values: np.ndarray, k: int) -> np.ndarray:
    """torch.topk(values, k, largest=True, sorted=True).indices on a 1-D float32 row."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    out = np.empty(max(k, 1), np.int64)
    rc = lib().orc_topk(_p(v, f32p), C.c_int64(v.shape[0]), C.c_int64(k), _p(out, i64p))
    assert rc == 0
    return out[:k]


def topk_rows(values: np.ndarray, cnt: np.ndarray, k: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(values, dtype=np.float32)
    cnt, k = _i64(cnt), _i64(k)
    out = np.empty((v.shape[0], 2), np.int64)
    rc = lib().orc_topk_rows(_p(v, f32p), C.c_int64(v.shape[0]), C.c_int64(v.shape[1]), _p(cnt, i64p),
                             _p(k, i64p), _p(out, i64p))
    assert rc == 0
    return out


GEN_KEYS = ["node_enc.0", "node_enc.2", "node_proj.0", "node_proj.2", "edge_scorer.0", "edge_attr_pred.0"]


def _f32(a) -> np.ndarray:
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def node_proj(state, x: np.ndarray, z: np.ndarray, dedup_rows: Optional[bool] = None):
    """Position-independent node pipeline (see aig_oracle.c) -> (h_cond [n,128], proj [4,n,128]).

    The pipeline is a function of a node's own feature row (and z) alone, evaluated by one fixed chain of
    operations per row, so equal rows give equal bits: with ``dedup_rows`` (default: from 4096 rows on) it is
    evaluated once per DISTINCT row and copied out - what makes million-node templates checkable in seconds
    (AIG features take a handful of distinct values)."""
    x, z = _f32(x), _f32(z).reshape(-1)
    n = int(x.shape[0])
    if dedup_rows is None:
        dedup_rows = n >= 4096
    if dedup_rows and n > 0:
        # rows compared by their bits (so that -0.0 / NaN payloads are not merged with their look-alikes)
        keys = np.ascontiguousarray(x).view(np.uint32).reshape(n, -1)
        uniq, inverse = np.unique(keys, axis=0, return_inverse=True)
        if uniq.shape[0] < n:
            h_u, proj_u = node_proj(state, uniq.view(np.float32), z, dedup_rows=False)
            inverse = inverse.reshape(-1)
            return h_u[inverse], proj_u[:, inverse]
    ws = []
    for k in GEN_KEYS:
        ws += [_f32(state[k + ".weight"]), _f32(state[k + ".bias"])]
    h_cond = np.empty((n, 128), np.float32)
    proj = np.empty((4, max(n, 1), 128), np.float32)
    zd = int(z.size)        # (the weights' row lengths must agree with it: node_proj.0 [128,128+zd], edge MLPs [128,256+zd])
    assert ws[4].shape == (128, 128 + zd) and ws[8].shape == (128, 256 + zd) and ws[10].shape == (128, 256 + zd)
    rc = lib().orc_node_proj_z(C.c_int64(n), _p(x, f32p), _p(z, f32p), C.c_int(zd), *[_p(w, f32p) for w in ws],
                               _p(h_cond, f32p), _p(proj, f32p))
    assert rc == 0
    return h_cond, proj[:, :n]


def score_rows(tgt, cnt, cand, A, B, w2, b2) -> np.ndarray:
    tgt, cnt, cand = _i64(tgt), _i64(cnt), _i64(cand)
    T, K = cand.shape
    A, B, w2, b2 = _f32(A), _f32(B), _f32(w2).reshape(-1), _f32(b2).reshape(-1)
    logits = np.full((T, K), np.nan, np.float32)
    rc = lib().orc_score_rows(C.c_int64(T), C.c_int64(K), _p(tgt, i64p), _p(cnt, i64p), _p(cand, i64p),
                              _p(A, f32p), _p(B, f32p), _p(w2, f32p), _p(b2, f32p), _p(logits, f32p))
    assert rc == 0
    return logits


def inv_bits(src, dst, A2, B2, w2a, b2a) -> np.ndarray:
    src, dst = _i64(src), _i64(dst)
    A2, B2, w2a, b2a = _f32(A2), _f32(B2), _f32(w2a).reshape(-1), _f32(b2a).reshape(-1)
    bits = np.empty(src.shape[0], np.int64)
    rc = lib().orc_inv_bits(C.c_int64(src.shape[0]), _p(src, i64p), _p(dst, i64p), _p(A2, f32p), _p(B2, f32p),
                            _p(w2a, f32p), _p(b2a, f32p), _p(bits, i64p))
    assert rc == 0
    return bits


# ----------------------------------------------------------------------------- constraints
def _edges_of(data):
    ei = getattr(data, "edge_index", None)
    if ei is None or ei.numel() == 0:
        z = np.zeros(0, np.int64)
        return z, z, z
    src, dst = _i64(ei[0].cpu().numpy()), _i64(ei[1].cpu().numpy())
    ea = getattr(data, "edge_attr", None)
    inv = np.zeros_like(src) if ea is None else _i64(ea.view(-1).cpu().numpy().astype(np.int64))
    return src, dst, inv


def enforce_aig_constraints(data, strict: bool = False, check_all_on_path: bool = True):
    """models/constraints.py:508-669; mutates and returns ``data`` like the reference."""
    dev = data.x.device
    n = int(data.x.size(0))
    types = _types_of(data.x)
    src, dst, inv = _edges_of(data)
    e = int(src.shape[0])
    depth = _depth_of(data, n) if hasattr(data, "node_depth") else None
    osrc, odst, oinv = (np.empty(2 * n + 1, np.int64) for _ in range(3))
    oe = C.c_int64(0)
    ndep, icnt = np.empty(n, np.int64), np.empty(n, np.int64)
    rc = lib().orc_enforce(C.c_int64(n), _p(types, i64p), C.c_int64(e), _p(src, i64p), _p(dst, i64p), _p(inv, i64p),
                           _p(depth, f64p), _p(osrc, i64p), _p(odst, i64p), _p(oinv, i64p), C.byref(oe),
                           _p(ndep, i64p), _p(icnt, i64p))
    assert rc == 0
    m = oe.value
    data.edge_index = torch.from_numpy(np.stack([osrc[:m], odst[:m]])).to(dev)
    data.edge_attr = torch.from_numpy(oinv[:m].copy()).view(-1, 1).to(dev)
    data.node_depth = torch.from_numpy(ndep).to(dev)
    if e > 0:
        data.x = data.x.clone()
    data.x[:, 1] = torch.from_numpy(icnt).to(dev).to(data.x.dtype)
    return data


PEN_KEYS = ["pi_in", "po_out", "backedge", "selfloop", "dup_fanin", "indeg_mse", "fanin_type", "reach_violation",
            "total"]


def aig_constraint_penalties(data, weights=None) -> Dict[str, torch.Tensor]:
    """models/constraints.py:253-366 (default weights only)."""
    assert not weights, "the checker restates the default weights"
    n = int(data.x.size(0))
    types = _types_of(data.x)
    src, dst, _ = _edges_of(data)
    out = np.zeros(9, np.float64)
    rc = lib().orc_penalties(C.c_int64(n), _p(types, i64p), C.c_int64(src.shape[0]), _p(src, i64p), _p(dst, i64p),
                             _p(out, f64p))
    assert rc == 0
    return {k: torch.tensor(float(out[i]), dtype=torch.float32, device=data.x.device) for i, k in enumerate(PEN_KEYS)}


def format_violations(n: int, types: np.ndarray, src: np.ndarray, dst: np.ndarray, eflag: np.ndarray,
                      ncode: np.ndarray, indeg: np.ndarray, on_path: np.ndarray, has_cycle: bool,
                      check_all_on_path: bool) -> List[str]:
    """Message texts of models/constraints.py:398-503 from per-edge / per-node verdicts."""
    errs: List[str] = []
    bad_types = np.nonzero(ncode == 1)[0].tolist()
    if bad_types:
        errs.append(f"Unknown node_type indices: {bad_types[:10]} ...")

    def ex(bit, k=5):
        idx = np.nonzero(eflag & bit)[0][:k]
        return [(int(src[i]), int(dst[i])) for i in idx]

    if (eflag & 1).any():
        errs.append(f"Self-loops not allowed: examples {ex(1)} ...")
    if (eflag & 2).any():
        errs.append(f"Duplicate edges found: examples {ex(2)} ...")
    if (eflag & 4).any():
        errs.append(f"Edges into PI not allowed: examples {ex(4)} ...")
    if (eflag & 8).any():
        errs.append(f"Edges out of PO not allowed: examples {ex(8)} ...")
    if (eflag & 16).any():
        errs.append(f"Edges with invalid src type: examples {ex(16)} ...")
    for v in np.nonzero((types == 1) & (ncode == 2))[0].tolist():
        errs.append(f"[PO] node {v} indegree={int(indeg[v])} (must be 1)")
    and_bad = np.nonzero((types == 2) & ((ncode == 3) | (ncode == 4)))[0].tolist()
    if and_bad:
        order = np.argsort(dst, kind="stable")
        sdst = dst[order]
        for v in and_bad:
            if ncode[v] == 3:
                errs.append(f"[AND] node {v} indegree={int(indeg[v])} (must be 2)")
            else:
                lo, hi = np.searchsorted(sdst, v), np.searchsorted(sdst, v, side="right")
                lst = [int(src[i]) for i in order[lo:hi]]
                errs.append(f"[AND] node {v} duplicated fanin {lst}")
    if has_cycle:
        errs.append("Graph contains a cycle (fails DAG).")
    if check_all_on_path:
        bad = np.nonzero(on_path == 0)[0].tolist()
        if bad:
            errs.append(f"Dangling/uncovered nodes (not on PI->PO path): {bad[:10]} ... (count={len(bad)})")
    return errs


def validate_strict_aig(data, check_all_on_path: bool = True) -> List[str]:
    """models/constraints.py:375-505."""
    x = data.x
    if x is None or x.size(1) == 0:
        return ["data.x missing or empty"]
    n = int(x.size(0))
    types = _types_of(x)
    src, dst, _ = _edges_of(data)
    e = int(src.shape[0])
    eflag = np.zeros(max(e, 1), np.uint8)
    ncode, on_path = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    indeg = np.zeros(max(n, 1), np.int64)
    summary = np.zeros(2, np.int64)
    rc = lib().orc_validate(C.c_int64(n), _p(types, i64p), C.c_int64(e), _p(src, i64p), _p(dst, i64p),
                            _p(eflag, u8p), _p(ncode, u8p), _p(indeg, i64p), _p(on_path, u8p), _p(summary, i64p))
    assert rc == 0
    return format_violations(n, types, src, dst, eflag[:e], ncode[:n], indeg[:n], on_path[:n],
                             bool(summary[0]), check_all_on_path)


def resync_attrs(data):
    """AIGgan.py:29-62; mutates and returns ``data``."""
    n = int(data.x.size(0))
    src, dst, inv = _edges_of(data)
    icnt, dep = np.empty(n, np.int64), np.empty(n, np.int64)
    rc = lib().orc_resync(C.c_int64(n), C.c_int64(src.shape[0]), _p(src, i64p), _p(dst, i64p), _p(inv, i64p),
                          _p(icnt, i64p), _p(dep, i64p))
    assert rc == 0
    dev = data.x.device
    data.x[:, 1] = torch.from_numpy(icnt).to(dev).to(data.x.dtype)
    data.node_depth = torch.from_numpy(dep).to(dev)
    return data
