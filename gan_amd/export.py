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


# ---------------------------------------------------------------------------------- direct exporter
# utilities/pt2bench.py of the reference: `.pt` -> `.bench` in one step, with inverters written as
# NOT gates.  Edges are first cleaned and oriented (pt2bench.py:9-173), then only the region on a
# PI -> PO path is written (pt2bench.py:216-428).
def _orient_edges(src, dst, inv, types, depth):
    """Cleaning and orientation of pt2bench.py:34-102 as array operations: drop out-of-range and
    self edges and PI-PI / PO-PO pairs, turn an edge round when it enters a PI, leaves a PO, goes
    down in depth or down in type priority (PI < AND < PO), drop exact repeats (first kept), and
    keep only depth-increasing edges unless that would cost more than 30 % of them."""
    n = types.shape[0]
    ok = (src >= 0) & (dst >= 0) & (src < n) & (dst < n) & (src != dst)
    src, dst, inv = src[ok], dst[ok], inv[ok]
    ts, td = types[src], types[dst]
    same = ((ts == PI) & (td == PI)) | ((ts == PO) & (td == PO))
    src, dst, inv, ts, td = src[~same], dst[~same], inv[~same], ts[~same], td[~same]

    def swap(mask):
        nonlocal src, dst, ts, td
        src, dst = np.where(mask, dst, src), np.where(mask, src, dst)
        ts, td = np.where(mask, td, ts), np.where(mask, ts, td)
    swap((td == PI) & (ts != PI))
    swap((ts == PO) & (td != PO))
    if depth is not None:
        swap(depth[src] > depth[dst])
    rank_s = np.select([ts == PI, ts == AND, ts == PO], [0, 1, 2], 9)
    rank_d = np.select([td == PI, td == AND, td == PO], [0, 1, 2], -1)
    swap(rank_s > rank_d)
    same = ((ts == PI) & (td == PI)) | ((ts == PO) & (td == PO))
    src, dst, inv = src[~same], dst[~same], inv[~same]
    if src.size:
        _, first = np.unique(np.stack([src, dst, inv], 1), axis=0, return_index=True)
        keep = np.sort(first)
        src, dst, inv = src[keep], dst[keep], inv[keep]
    if depth is not None and src.size:
        up = depth[src] < depth[dst]
        if (~up).sum() <= 0.3 * src.size:
            src, dst, inv = src[up], dst[up], inv[up]
    return src, dst, inv


def _break_cycles(src, dst, inv, n, depth):
    """pt2bench.py:105-152: while a Kahn peel leaves nodes behind, take the smallest such node and
    delete one edge entering it - the first that does not go up in depth, else the first."""
    src, dst, inv = src.tolist(), dst.tolist(), inv.tolist()
    while src:
        indeg = np.bincount(np.asarray(dst), minlength=n)
        order = np.argsort(np.asarray(src), kind="stable")
        s_sorted = np.asarray(src)[order]
        d_sorted = np.asarray(dst)[order]
        ptr = np.searchsorted(s_sorted, np.arange(n + 1))
        indeg = indeg.copy()
        frontier = np.nonzero(indeg == 0)[0]
        seen = 0
        while frontier.size:
            seen += frontier.size
            beg, end = ptr[frontier], ptr[frontier + 1]
            cnt = end - beg
            if cnt.sum() == 0:
                break
            idx = np.repeat(beg, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
            tgt = d_sorted[idx]
            np.subtract.at(indeg, tgt, 1)
            cand = np.unique(tgt)
            frontier = cand[indeg[cand] == 0]
        if seen == n:
            break
        left = np.nonzero(indeg > 0)[0]
        if left.size == 0:
            break
        victim = int(left[0])
        pick = -1
        for k, d in enumerate(dst):
            if d == victim:
                if depth is not None and not (float(depth[src[k]]) < float(depth[d])):
                    pick = k
                    break
                if pick == -1:
                    pick = k
        if pick == -1:
            break
        del src[pick], dst[pick], inv[pick]
    return np.asarray(src, np.int64), np.asarray(dst, np.int64), np.asarray(inv, np.int64)


def _closure(starts, ptr, adj, n):
    seen = np.zeros(n, bool)
    seen[starts] = True
    frontier = starts
    while frontier.size:
        beg, end = ptr[frontier], ptr[frontier + 1]
        cnt = end - beg
        if cnt.sum() == 0:
            break
        idx = np.repeat(beg, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        nb = np.unique(adj[idx])
        frontier = nb[~seen[nb]]
        seen[frontier] = True
    return seen


def bench_text_with_inverters(data, name: str = "graph.pt") -> str:
    """The text utilities/pt2bench.py:216-428 writes for ``data`` (``name`` = basename of the
    ``.pt`` file, which the header line quotes).  No trailing newline, as there."""
    x = _np(data.x)
    n = int(x.shape[0])
    types = x[:, 0].astype(np.int64)
    depth = _np(getattr(data, "node_depth", None))
    if depth is not None:
        depth = depth.astype(np.float64).reshape(-1)
    ei = _np(getattr(data, "edge_index", None), np.int64)
    if ei is None or ei.size == 0:
        src = dst = inv = np.zeros(0, np.int64)
    else:
        src, dst = ei[0].reshape(-1), ei[1].reshape(-1)
        ea = _np(getattr(data, "edge_attr", None))
        inv = np.zeros(src.size, np.int64) if ea is None else ea.reshape(-1).astype(np.int64)
    src, dst, inv = _orient_edges(src, dst, inv, types, depth)
    src, dst, inv = _break_cycles(src, dst, inv, n, depth)

    # predecessor lists in edge order; successor / predecessor CSR for the two closures
    by_dst = np.argsort(dst, kind="stable")
    p_src, p_inv = src[by_dst], inv[by_dst]
    p_ptr = np.searchsorted(dst[by_dst], np.arange(n + 1))
    by_src = np.argsort(src, kind="stable")
    s_ptr = np.searchsorted(src[by_src], np.arange(n + 1))
    reach = _closure(np.nonzero(types == PI)[0], s_ptr, dst[by_src], n)
    rev = _closure(np.nonzero(types == PO)[0], p_ptr, p_src, n)
    on_path = reach & rev
    if on_path.any():
        active = on_path
    elif reach.any():
        active = reach
    elif rev.any():
        active = rev
    else:
        active = (types == PI) | (types == AND) | (types == PO)

    lines = [f'# Benchmark "{name}" written by AIG-GAN']
    names = [f"n{i}" for i in range(n)]
    for k, i in enumerate(np.nonzero(types == PI)[0].tolist(), start=1):
        names[i] = f"N{k}"
        lines.append(f"INPUT(N{k})")

    def preds_of(i):
        return list(zip(p_src[p_ptr[i]:p_ptr[i + 1]].tolist(), p_inv[p_ptr[i]:p_ptr[i + 1]].tolist()))

    body, nots = [], set()
    defined_and = np.zeros(n, bool)
    usable = active & ((types == PI) | (types == AND))
    for i in np.nonzero((types == AND) & active)[0].tolist():
        preds = [(s, v) for s, v in preds_of(i) if usable[s]]
        if not preds:
            continue
        preds.sort(key=(lambda t: (depth[t[0]], t[0])) if depth is not None else (lambda t: t[0]))
        terms = []
        for s, v in preds:
            sig = names[s]
            if v:
                sig += "_not"
                if sig not in nots:
                    body.append(f"{sig} = NOT({names[s]})")
                    nots.add(sig)
            terms.append(sig)
        if len(terms) == 1:
            terms.append(terms[0])
        acc = terms[0]
        for k, term in enumerate(terms[1:-1], start=1):
            body.append(f"{names[i]}_and{k} = AND({acc}, {term})")
            acc = f"{names[i]}_and{k}"
        body.append(f"{names[i]} = AND({acc}, {terms[-1]})")
        defined_and[i] = True

    counter, emitted, used = 1, 0, set()
    any_on_path = bool(on_path.any())
    for i in np.nonzero(types == PO)[0].tolist():
        every = preds_of(i)
        if not (on_path[i] or (not any_on_path and (active[i] or every))):
            continue
        preds = [t for t in every if on_path[t[0]]] or [t for t in every if active[t[0]]] or every
        if not preds:
            continue
        preds.sort(key=lambda t: (0 if types[t[0]] == AND else 1, depth[t[0]] if depth is not None else 0.0, t[0]))
        # an AND that was never written cannot drive an output; anything else can
        drive = next(((s, v) for s, v in preds if not (types[s] == AND and not defined_and[s])), None)
        if drive is None:
            continue
        sig = names[drive[0]]
        if drive[1]:
            body.append(f"{sig}_not_out{counter} = NOT({sig})")
            sig = f"{sig}_not_out{counter}"
        if sig in used:
            sig = f"{sig}_dup{counter}"
        lines.append(f"OUTPUT({sig})")
        used.add(sig)
        counter += 1
        emitted += 1

    if emitted == 0:
        cand = np.nonzero((types == AND) & active & defined_and)[0]
        if cand.size == 0:
            cand = np.nonzero((types == AND) & defined_and)[0]
        if cand.size:
            key = sorted(cand.tolist(), key=(lambda i: (depth[i], i)) if depth is not None else (lambda i: i), reverse=True)
            for i in key[:2]:
                if names[i] not in used:
                    lines.append(f"OUTPUT({names[i]})")
                    used.add(names[i])
        elif reach.any():
            pis = np.nonzero(types == PI)[0].tolist()
            for i in pis[:max(1, min(2, len(pis)))]:
                if names[i] not in used:
                    lines.append(f"OUTPUT({names[i]})")
                    used.add(names[i])
    return "\n".join(lines + body)


def pt_to_bench(pt_path: str, out_path: str, verbose: bool = True) -> None:
    """``utilities/pt2bench.py:216`` - same arguments, same file content."""
    from .dataset import _read, _to_graph
    data = _to_graph(_read(pt_path))
    text = bench_text_with_inverters(data, os.path.basename(pt_path))
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        f.write(text)
    if verbose:
        ands = sum(" = AND(" in ln for ln in text.splitlines())
        outs = sum(ln.startswith("OUTPUT(") for ln in text.splitlines())
        print(f"[INFO] Saved .bench: {out_path} (nodes={int(data.x.size(0))}, outputs={outs}, ANDs={ands})")
