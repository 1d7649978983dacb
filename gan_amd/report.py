"""Inspection reports of the reference's helper scripts, computed from the graph tensors.

* ``pt_parameters`` / ``write_pt_parameters_csv`` - the per-file circuit parameters of
  data_files/check_pt.py:35-129 (``PI, PO, N, E, I, D`` and a status column, CSV header in the
  order ``filename,PI,PO,N,E,I,D,status``) with the depth diagnosis it prints (:40-51);
* ``check_graphml_lines`` / ``check_bench_lines`` - the quick checks of
  utilities/scripts_check.py:8-67, returned as the lines that script prints.

Host-side bookkeeping over a few integers per file; nothing here touches the GPU.  ``.pt`` files
are read with the package's loader (dict form, or a pickled ``Data`` when torch_geometric is
importable), never by unpickling arbitrary objects.
"""
from __future__ import annotations

import csv
import os
from typing import Dict, List, Optional

import torch

from .data import Data

CSV_FIELDS = ["filename", "PI", "PO", "N", "E", "I", "D", "status"]


def depth_diagnosis(d: Data) -> List[str]:
    """The three lines of check_pt.py:40-51 about the depth attribute."""
    max_depth = d.node_depth.max().item()
    zero_ratio = (d.node_depth == 0).sum().item() / d.x.shape[0]
    lines = [f"最大深度: {max_depth}", f"深度为0的节点比例: {zero_ratio:.2%}"]
    if max_depth == 0:
        lines.append("❌ 警告：所有节点深度为0，数据可能有问题！")
    elif zero_ratio > 0.1:
        lines.append("⚠️  深度为0的节点过多，可能存在问题")
    else:
        lines.append("✅ 深度分布正常")
    return lines


def pt_parameters(d: Data, filename: str = "") -> Dict[str, object]:
    """Row of the parameter table for one graph (check_pt.py:52-80); any failure becomes a row of
    ``None`` with ``status = 'error: ...'`` (check_pt.py:81-91)."""
    try:
        depth_diagnosis(d)   # the reference computes these first, so a graph without depths is an error row
        types = d.x[:, 0]
        return {
            "filename": filename,
            "PI": int((types == 0).sum().item()),
            "PO": int((types == 1).sum().item()),
            "N": int((types == 2).sum().item()),
            "E": d.edge_index.shape[1] if d.edge_index.numel() > 0 else 0,
            "I": int((d.edge_attr == 1).sum().item()) if d.edge_attr.numel() > 0 else 0,
            "D": int(d.node_depth.max().item()) if d.node_depth.numel() > 0 else 0,
            "status": "success",
        }
    except Exception as e:
        row: Dict[str, object] = {k: None for k in CSV_FIELDS}
        row["filename"], row["status"] = filename, f"error: {e}"
        return row


def _load_pt(path: str) -> Data:
    from .dataset import _read, _to_graph
    g = _to_graph(_read(path))
    if g is None:
        raise ValueError("no graph in file")
    return g


def write_pt_parameters_csv(pt_dir: str, output_csv: str, verbose: bool = True) -> List[Dict[str, object]]:
    """One row per ``.pt`` file of ``pt_dir`` (directory order, as the reference), written to ``output_csv``."""
    files = [f for f in os.listdir(pt_dir) if f.endswith(".pt")]
    if not files:
        print(f"在目录 {pt_dir} 中未找到任何.pt文件")
        return []
    rows = []
    for f in files:
        try:
            d = _load_pt(os.path.join(pt_dir, f))
            if verbose:
                for line in depth_diagnosis(d):
                    print(line)
            rows.append(pt_parameters(d, f))
        except Exception as e:
            row: Dict[str, object] = {k: None for k in CSV_FIELDS}
            row["filename"], row["status"] = f, f"error: {e}"
            rows.append(row)
    with open(output_csv, "w", newline="", encoding="utf-8") as fh:
        w = csv.DictWriter(fh, fieldnames=CSV_FIELDS)
        w.writeheader()
        w.writerows(rows)
    print(f"处理完成，结果已保存到 {output_csv}")
    return rows


def check_graphml_lines(graphml_path: str) -> List[str]:
    """AND nodes need exactly two predecessors, POs at least one, nobody isolated (scripts_check.py:8-48)."""
    import networkx as nx
    G = nx.read_graphml(graphml_path)
    out = [f"[INFO] Checking GraphML: {graphml_path}", f"Nodes: {len(G.nodes())}, Edges: {len(G.edges())}"]
    and_bad, po_bad = [], []
    for node, attr in G.nodes(data=True):
        kind = int(attr.get("node_type", 2))
        indeg = len(list(G.in_edges(node)))
        if kind == 2 and indeg != 2:
            and_bad.append((node, indeg))
        elif kind == 1 and indeg < 1:
            po_bad.append(node)
    if and_bad:
        out += [f"[WARN] AND node {node} has {k} predecessors (expect 2)" for node, k in and_bad[:20]]
        if len(and_bad) > 20:
            out.append(f"... and {len(and_bad)-20} more AND nodes with wrong indegree")
    else:
        out.append("[OK] All AND nodes have exactly 2 predecessors.")
    if po_bad:
        out += [f"[WARN] PO node {node} has 0 predecessors" for node in po_bad]
    else:
        out.append("[OK] All PO nodes have at least 1 predecessor.")
    iso = [node for node, deg in G.degree() if deg == 0]
    if iso:
        out.append(f"[WARN] Isolated nodes: {iso[:30]}{' ...' if len(iso)>30 else ''}")
    else:
        out.append("[OK] No isolated nodes.")
    return out


def check_bench_lines(bench_path: str) -> List[str]:
    """Gate / output counts of a ``.bench`` text and its first ten AND lines (scripts_check.py:51-67)."""
    with open(bench_path, "r", encoding="utf-8") as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    ands = [ln for ln in lines if " = AND(" in ln]
    outs = sum(1 for ln in lines if ln.startswith("OUTPUT("))
    out = [f"[INFO] .bench file: {bench_path}", f"AND gates (text count): {len(ands)}, OUTPUTs: {outs}"]
    if ands:
        out.append("[SAMPLE AND]:")
        out += ["  " + ln for ln in ands[:10]]
    return out


def main(argv: Optional[List[str]] = None) -> None:
    import argparse
    p = argparse.ArgumentParser(description="AIG inspection reports (check_pt.py / scripts_check.py of the reference)")
    sub = p.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("pt", help="circuit parameters of every .pt file of a directory -> CSV")
    a.add_argument("--pt-dir", required=True)
    a.add_argument("--output-csv", default="pt_parameters.csv")
    b = sub.add_parser("check", help="quick check of a GraphML file and a .bench file")
    b.add_argument("--graphml", required=True)
    b.add_argument("--bench", required=True)
    args = p.parse_args(argv)
    if args.cmd == "pt":
        write_pt_parameters_csv(args.pt_dir, args.output_csv)
    else:
        for path in (args.graphml, args.bench):
            if not os.path.exists(path):
                raise FileNotFoundError(path)
        print("\n".join(check_graphml_lines(args.graphml) + check_bench_lines(args.bench)))


if __name__ == "__main__":
    main()
