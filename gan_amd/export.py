"""``.pt`` graph -> ISCAS ``.bench`` text, the export chain of the reference's README restated on
the ``Data`` tensors directly (host side, O(N + E) with two sorts; no GraphML file in between).

The reference goes ``pt2graphml.pt_to_graphml`` (utilities/pt2graphml.py:10-48) and then
``graphml2bench.convert_graphml_to_bench`` (utilities/graphml2bench.py:403-580).  What those two
steps amount to, and what ``bench_text`` reproduces byte for byte:

* graph: one node per row of ``x`` with type ``int(x[i,0])`` (0 PI, 1 PO, 2 AND, anything else is
  never emitted); every edge ``(u,v)`` with ``u != v`` is kept once (a ``DiGraph``: repeats
  collapse); an endpoint outside ``[0, N)`` - the ``-1`` padding of a padded sample, say - creates
  an attribute-less node, which reads back as type 0, i.e. one more primary input (pt2graphml.py:37-43,
  graphml2bench.py:438-445);
* ``INPUT(n<id>)`` for the PIs by id; then the ANDs by id: the two smallest-id predecessors, or -
  with ``repair_dag`` - the one predecessor plus a round-robin pick among the ANDs written so far
  (the PIs while there is none), two such picks without any predecessor (graphml2bench.py:485-528);
* ``OUTPUT(n<id>)`` per PO by id: the first AND met walking back from the PO's smallest-id
  predecessor along smallest-id predecessors (``None`` on a dead end or a cycle), else a
  round-robin pick among the ANDs, else the last PI, else the PO itself (graphml2bench.py:532-575).
  Inversion values do not reach this format (graphml2bench.py:413).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

PI, PO, AND = 0, 1, 2


def _np(t, dtype=None):
    if t is None:
        return None
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return a if dtype is None else a.astype(dtype)


def bench_text(data, repair_dag: bool = True, diversify_pad: bool = True) -> str:
    """The ``.bench`` file content the reference's two-step export writes for ``data``."""
    x = _np(data.x)
    n = int(x.shape[0])
    ei = _np(getattr(data, "edge_index", None), np.int64)
    if ei is None or ei.size == 0:
        src = dst = np.zeros(0, np.int64)
    else:
        src, dst = ei[0].reshape(-1), ei[1].reshape(-1)
    keep = src != dst
    src, dst = src[keep], dst[keep]

    # node table: ids 0..N-1, then the strays an edge endpoint brought in (all of type PI)
    stray = np.setdiff1d(np.concatenate([src, dst]), np.arange(n, dtype=np.int64)) if src.size else np.zeros(0, np.int64)
    ids = np.concatenate([np.arange(n, dtype=np.int64), stray])          # index -> id
    types = np.concatenate([x[:, 0].astype(np.int64) if n else np.zeros(0, np.int64), np.zeros(stray.size, np.int64)])
    total = int(ids.size)

    def index_of(v):                                                       # id -> index
        out = v.copy()
        out_of = (v < 0) | (v >= n)
        if out_of.any():
            out[out_of] = n + np.searchsorted(stray, v[out_of])
        return out

    # predecessor lists: unique (target, source) pairs, sources ascending by id
    if src.size:
        pairs = np.unique(np.stack([index_of(dst), src], 1), axis=0)       # sorted by target index, then source id
        p_dst, p_src = pairs[:, 0], pairs[:, 1]
    else:
        p_dst = p_src = np.zeros(0, np.int64)
    ptr = np.searchsorted(p_dst, np.arange(total + 1))
    npred = np.diff(ptr)
    first = np.where(npred > 0, p_src[np.minimum(ptr[:-1], max(p_src.size - 1, 0))] if p_src.size else 0, -1)

    def by_id(mask):
        sel = np.nonzero(mask)[0]
        return sel[np.argsort(ids[sel], kind="stable")]

    pi_idx, and_idx, po_idx = by_id(types == PI), by_id(types == AND), by_id(types == PO)
    pi_ids, and_ids = ids[pi_idx], ids[and_idx]
    lines = [f"INPUT(n{int(u)})" for u in pi_ids]

    # ---- AND gates
    m = int(and_idx.size)
    if m:
        cnt = npred[and_idx]
        beg = ptr[and_idx]
        a = np.where(cnt >= 1, p_src[np.minimum(beg, max(p_src.size - 1, 0))] if p_src.size else 0, 0)
        b = np.where(cnt >= 2, p_src[np.minimum(beg + 1, max(p_src.size - 1, 0))] if p_src.size else 0, 0)
        short = np.nonzero(cnt < 2)[0]
        if short.size:
            if repair_dag:
                # pool of gate j = ANDs written before it, or the PIs while there is none; a pick
                # advances the round-robin pointer only when its pool is not empty
                rr = 0
                npi = int(pi_ids.size)
                for j in short.tolist():                     # in id order, as the pointer is shared
                    pool_len = j if j > 0 else npi

                    def pick():
                        nonlocal rr
                        if pool_len == 0:
                            return None
                        k = rr % pool_len if diversify_pad else pool_len - 1
                        if diversify_pad:
                            rr += 1
                        return int(and_ids[k] if j > 0 else pi_ids[k])
                    if cnt[j] == 1:
                        p = pick()
                        b[j] = a[j] if p is None else p
                    else:
                        pa, pb = pick(), pick()
                        if pa is None and pb is None:
                            pa = pb = int(and_ids[j])
                        a[j] = pb if pa is None else pa
                        b[j] = a[j] if pb is None else pb
            else:
                one = short[cnt[short] == 1]
                none = short[cnt[short] == 0]
                b[one] = a[one]
                a[none] = b[none] = and_ids[none]
        lines.extend(f"n{u} = AND(n{p},n{q})" for u, p, q in zip(and_ids.tolist(), a.tolist(), b.tolist()))

    # ---- outputs
    if po_idx.size:
        is_and = types == AND
        # walk back from the PO's first predecessor along first predecessors until an AND
        # (a stray id may itself be -1, so "no predecessor" is read from the counts, not from `first`)
        target = np.zeros(po_idx.size, np.int64)
        has = npred[po_idx] > 0
        cur_idx = np.where(has, index_of(np.where(has, first[po_idx], 0)), 0)
        active = has.copy()
        found = np.zeros(po_idx.size, bool)
        steps = 0
        while active.any() and steps <= total:
            hit = active & is_and[cur_idx]
            target[hit] = ids[cur_idx[hit]]
            found |= hit
            active &= ~hit
            dead = active & (npred[cur_idx] == 0)
            active &= ~dead
            nxt = first[cur_idx[active]]
            cur_idx[active] = index_of(nxt)
            steps += 1
        # whoever is still walking after `total` steps is on a cycle without an AND
        rr = 0
        out_ids = []
        last_pi = int(pi_ids[-1]) if pi_ids.size else None
        for k in range(po_idx.size):
            if found[k]:
                out_ids.append(int(target[k]))
            elif m:
                if diversify_pad:
                    out_ids.append(int(and_ids[rr % m]))
                    rr += 1
                else:
                    out_ids.append(int(and_ids[-1]))
            elif last_pi is not None:
                out_ids.append(last_pi)
            else:
                out_ids.append(int(ids[po_idx[k]]))
        lines.extend(f"OUTPUT(n{t})" for t in out_ids)
    return "\n".join(lines) + "\n"


def write_bench(data, bench_path: str, repair_dag: bool = True, diversify_pad: bool = True) -> None:
    os.makedirs(os.path.dirname(bench_path) or ".", exist_ok=True)
    with open(bench_path, "w", encoding="utf-8") as f:
        f.write(bench_text(data, repair_dag=repair_dag, diversify_pad=diversify_pad))


def pt_to_bench_file(pt_path: str, bench_path: Optional[str] = None, repair_dag: bool = True,
                     diversify_pad: bool = True) -> str:
    """``.pt`` (dict form, or a pickled ``Data`` where its class is importable) -> ``.bench`` file."""
    from .dataset import _read, _to_graph
    data = _to_graph(_read(pt_path))
    if data is None:
        raise ValueError(f"{pt_path}: no graph found (expected a dict with 'x' or a Data object)")
    if bench_path is None:
        bench_path = os.path.splitext(pt_path)[0] + ".bench"
    write_bench(data, bench_path, repair_dag=repair_dag, diversify_pad=diversify_pad)
    return bench_path


# ---------------------------------------------------------------------------------- GraphML files
# The reference's two CLIs exchange a GraphML file.  These two functions keep that file format
# available (they need networkx, as the reference does); `bench_text` above does not.
def pt_to_graphml(pt_path: str, out_path: str, verbose: bool = True) -> None:
    """GraphML with the node / edge attributes of utilities/pt2graphml.py:28-46 (``node_type``,
    ``inverted_preds``, ``depth`` per node; ``edge_type`` per edge; self loops skipped, a repeated
    edge keeps its last attribute)."""
    import networkx as nx
    from .dataset import _read, _to_graph
    data = _to_graph(_read(pt_path))
    x = _np(data.x)
    ei = _np(data.edge_index, np.int64)
    ea = _np(getattr(data, "edge_attr", None))
    ea = np.zeros(ei.shape[1], np.int64) if ea is None else ea.reshape(-1)
    nd = _np(getattr(data, "node_depth", None))
    nd = np.arange(len(x), dtype=np.int64) if nd is None else nd
    g = nx.DiGraph()
    g.add_nodes_from((str(i), {"node_type": int(x[i, 0]), "inverted_preds": int(x[i, 1]), "depth": int(nd[i])})
                     for i in range(len(x)))
    g.add_edges_from((str(int(u)), str(int(v)), {"edge_type": int(t)})
                     for u, v, t in zip(ei[0].tolist(), ei[1].tolist(), ea.tolist()) if u != v)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    nx.write_graphml(g, out_path)
    if verbose:
        print(f"[INFO] Saved GraphML: {out_path} (nodes={len(x)}, edges={ei.shape[1]})")


def convert_graphml_to_bench(graphml_path: str, bench_path: str, repair_dag: bool = True,
                             diversify_pad: bool = True) -> None:
    """GraphML (integer node ids, ``node_type`` or ``d0`` per node) -> ``.bench`` with the rules of
    utilities/graphml2bench.py:403-580."""
    import networkx as nx
    g = nx.read_graphml(graphml_path)

    def kind(attrs):
        try:
            return int(attrs.get("d0", attrs.get("node_type", 0)))
        except (TypeError, ValueError):
            return 0
    try:
        kinds = {int(u): kind(a) for u, a in g.nodes(data=True)}
    except ValueError as exc:
        raise ValueError(f"{graphml_path}: node ids must be integers") from exc
    n = max([u for u in kinds if u >= 0], default=-1) + 1
    x = np.full((n, 2), -1.0, np.float32)          # ids missing from the file: a type nothing emits
    for u, k in kinds.items():
        if u >= 0:
            x[u, 0] = k
        elif k != PI:
            raise ValueError(f"{graphml_path}: node {u} has a negative id and type {k}")
    edges = np.asarray([(int(u), int(v)) for u, v in g.edges()], dtype=np.int64).reshape(-1, 2)
    from .data import Data
    write_bench(Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(edges.T.copy())), bench_path,
                repair_dag=repair_dag, diversify_pad=diversify_pad)


def main(argv=None) -> None:
    import argparse
    p = argparse.ArgumentParser(description="AIG sample (.pt) -> ISCAS .bench, same text as pt2graphml + graphml2bench")
    p.add_argument("-i", "--input", required=True, help=".pt file")
    p.add_argument("-o", "--output", default=None, help=".bench path (default: next to the input)")
    p.add_argument("--no-repair", action="store_true")
    p.add_argument("--no-diversify", action="store_true")
    a = p.parse_args(argv)
    out = pt_to_bench_file(a.input, a.output, repair_dag=not a.no_repair, diversify_pad=not a.no_diversify)
    print(f"[OK] Wrote {out}")


if __name__ == "__main__":
    main()
