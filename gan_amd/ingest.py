"""Dataset ingest: ABC-style AIG ``.bench`` text -> the training graph (``Data``), without the
GraphML file the reference writes in between.

The reference prepares its dataset in two steps: ``data_files/andAIG2Graphml.py`` parses the
``.bench`` into a networkx ``MultiDiGraph`` (andAIG2Graphml.py:74-168) and ``data_files/generate_pt.py``
turns the GraphML into tensors (generate_pt.py:9-61).  ``bench_to_data`` gives the same tensors:

* one node per ``INPUT`` (and per ``x = vdd`` line) in file order, type 0; one node per ``AND``
  gate, type 2; ``NOT`` and ``BUFF`` gates are not nodes, they only rename a signal;
* an output driven by an AND gets an extra node ``<name>_buff`` (type 1) right after the gate; an
  output that is a ``NOT`` / ``BUFF`` gate becomes a type-1 node ``<name>_inv`` / ``<name>_buff``;
* edges run from a gate to its fan-ins (the dataset orientation), in gate order, ``edge_attr`` 1
  when the fan-in is reached through a renamed signal (andAIG2Graphml.py:49-56 - this includes a
  ``BUFF`` alias, which the reference also marks as inverted), else 0; ``AND(a, a)`` keeps both edges;
* ``x = [type, number of inverted fan-ins]`` (float32), ``edge_attr`` float32 ``[E,1]``,
  ``node_depth`` float32 from the Kahn sweep of generate_pt.py:9-22, including its treatment of
  parallel edges (a node entered twice by one gate is counted twice but released once, so it
  never leaves the queue's waiting list and passes no depth on).

Differences, on purpose: blank lines and ``#`` comment lines are skipped (the reference stops at
them unless they mention ABC); a signal that is used before it is defined, or an inverter chain
the reference cannot resolve either, raises ``ValueError`` with the line number.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Tuple

import numpy as np
import torch

from .data import Data

PI, PO, INTERNAL = 0, 1, 2
_RE_IN = re.compile(r"INPUT\((.*?)\)")
_RE_OUT = re.compile(r"OUTPUT\((.*?)\)")
_RE_LHS = re.compile(r"(.*?)=")
_RE_AND1 = re.compile(r"AND\((.*?),")
_RE_AND2 = re.compile(r",(.*?)\)")
_RE_NOT = re.compile(r"NOT\((.*?)\)")
_RE_BUFF = re.compile(r"BUFF\((.*?)\)")


def parse_bench(text: str) -> Tuple[List[int], List[int], List[Tuple[int, int, int]], List[str]]:
    """``(node types, inverted fan-in counts, edges (gate, fan-in, inverted), node names)``."""
    ids: Dict[str, int] = {}
    alias: Dict[str, str] = {}
    outputs = set()
    types: List[int] = []
    ninv: List[int] = []
    names: List[str] = []
    edges: List[Tuple[int, int, int]] = []

    def add(name: str, kind: int, inverted: int = 0) -> int:
        ids[name] = len(types)
        types.append(kind)
        ninv.append(inverted)
        names.append(name)
        return len(types) - 1

    def resolve(sig: str, lineno: int) -> Tuple[int, int]:
        """node index and inversion flag of a gate input"""
        if sig in ids:
            return ids[sig], 0
        if sig in alias and alias[sig] in ids:
            return ids[alias[sig]], 1
        raise ValueError(f"line {lineno}: signal {sig!r} is not defined by an INPUT, an AND or a single NOT/BUFF of one")

    for lineno, raw in enumerate(text.splitlines(), start=1):
        if "ABC" in raw or not raw.strip() or raw.lstrip().startswith("#"):
            continue
        line = raw.replace(" ", "").strip()
        if "vdd" in line:
            add(_RE_LHS.search(line).group(1), PI)
        elif "INPUT" in line:
            add(_RE_IN.search(line).group(1), PI)
        elif "OUTPUT" in line:
            outputs.add(_RE_OUT.search(line).group(1))
        elif "AND" in line:
            out = _RE_LHS.search(line).group(1)
            fanins = [_RE_AND1.search(line).group(1), _RE_AND2.search(line).group(1)]
            node = add(out, INTERNAL)
            for sig in fanins:
                s, inv = resolve(sig, lineno)
                edges.append((node, s, inv))
                ninv[node] += inv
            if out in outputs:
                po = add(out + "_buff", PO)
                edges.append((po, node, 0))
        elif "NOT" in line:
            out = _RE_LHS.search(line).group(1)
            pin = _RE_NOT.search(line).group(1)
            alias[out] = pin
            if out in outputs:
                if pin not in ids:
                    raise ValueError(f"line {lineno}: output inverter {out!r} on undefined or renamed signal {pin!r}")
                po = add(out + "_inv", PO, 1)
                edges.append((po, ids[pin], 1))
        elif "BUFF" in line:
            out = _RE_LHS.search(line).group(1)
            pin = _RE_BUFF.search(line).group(1)
            alias[out] = pin
            if out in outputs:
                s, inv = resolve(pin, lineno)
                po = add(out + "_buff", PO, inv)
                edges.append((po, s, inv))
        else:
            raise ValueError(f"line {lineno}: not an AIG bench statement: {raw!r}")
    return types, ninv, edges, names


def _kahn_depth(n: int, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """generate_pt.py:9-22 on a multigraph: in-degrees count parallel edges, successors are unique."""
    indeg = np.bincount(dst, minlength=n).astype(np.int64)
    depth = np.zeros(n, np.int64)
    if src.size:
        pairs = np.unique(np.stack([src, dst], 1), axis=0)
        u_src, u_dst = pairs[:, 0], pairs[:, 1]
    else:
        u_src = u_dst = np.zeros(0, np.int64)
    ptr = np.searchsorted(u_src, np.arange(n + 1))
    frontier = np.nonzero(indeg == 0)[0]
    while frontier.size:
        beg, end = ptr[frontier], ptr[frontier + 1]
        cnt = end - beg
        if cnt.sum() == 0:
            break
        idx = np.repeat(beg, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        tgt = u_dst[idx]
        np.maximum.at(depth, tgt, np.repeat(depth[frontier], cnt) + 1)
        np.subtract.at(indeg, tgt, 1)
        cand = np.unique(tgt)
        frontier = cand[indeg[cand] == 0]
    return depth


def bench_to_data(text: str) -> Data:
    types, ninv, edges, names = parse_bench(text)
    n = len(types)
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 3)
    # the reference's edge list is grouped by gate in node order (adjacency order of the graph)
    if e.shape[0]:
        e = e[np.argsort(e[:, 0], kind="stable")]
    src, dst = e[:, 0].copy(), e[:, 1].copy()
    # a FIFO Kahn sweep gives every released node its longest-path level, and nodes that are never
    # released keep the running maximum - both are what the level-synchronous sweep computes
    depth = _kahn_depth(n, src, dst)
    d = Data(x=torch.tensor(np.stack([types, ninv], 1) if n else np.zeros((0, 2)), dtype=torch.float32),
             edge_index=torch.from_numpy(np.stack([src, dst]) if e.shape[0] else np.zeros((2, 0), np.int64)),
             edge_attr=torch.from_numpy(e[:, 2].astype(np.float32)).view(-1, 1),
             node_depth=torch.from_numpy(depth.astype(np.float32)))
    d.node_names = names
    return d


def bench_to_pt(bench_path: str, out_path: str) -> None:
    """``.bench`` file -> ``.pt`` in the dict form ``{x, edge_index, edge_attr, node_depth}``."""
    with open(bench_path, "r") as f:
        d = bench_to_data(f.read())
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    torch.save({"x": d.x, "edge_index": d.edge_index, "edge_attr": d.edge_attr, "node_depth": d.node_depth}, out_path)


def main(argv=None) -> None:
    import argparse
    p = argparse.ArgumentParser(description="AIG .bench -> training graph .pt (andAIG2Graphml.py + generate_pt.py in one step)")
    p.add_argument("--bench", required=True)
    p.add_argument("--out", required=True)
    a = p.parse_args(argv)
    bench_to_pt(a.bench, a.out)
    print(f"[OK] Saved {a.out}")


if __name__ == "__main__":
    main()
