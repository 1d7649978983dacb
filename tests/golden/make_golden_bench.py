#!/usr/bin/env python3
"""Regenerates ``bench_text.json``: the ``.bench`` text the reference's export chain
(utilities/pt2graphml.py + utilities/graphml2bench.py, run unmodified from /root/reference in the
build container) writes for the graphs already held by the ``cons_*`` / ``gen_*`` / ``pad_c17``
fixtures and for a few hand-made corner cases.  Only inputs and the produced text are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bench.py
"""
import glob
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

from make_golden import import_reference, load_reference_file  # noqa: E402  (installs the torch_geometric stand-in)


def main():
    import_reference()
    pt2graphml = load_reference_file("utilities/pt2graphml.py")
    graphml2bench = load_reference_file("utilities/graphml2bench.py")

    def ref_bench(x, ei, ea, nd, **kw):
        with tempfile.TemporaryDirectory() as tmp:
            d = {"x": torch.from_numpy(np.asarray(x)), "edge_index": torch.from_numpy(np.asarray(ei, dtype=np.int64))}
            if ea is not None:
                d["edge_attr"] = torch.from_numpy(np.asarray(ea))
            if nd is not None:
                d["node_depth"] = torch.from_numpy(np.asarray(nd))
            torch.save(d, os.path.join(tmp, "g.pt"))
            pt2graphml.pt_to_graphml(os.path.join(tmp, "g.pt"), os.path.join(tmp, "g.graphml"), verbose=False)
            graphml2bench.convert_graphml_to_bench(os.path.join(tmp, "g.graphml"), os.path.join(tmp, "g.bench"), **kw)
            with open(os.path.join(tmp, "g.bench"), encoding="utf-8") as f:
                return f.read()

    pt2bench = load_reference_file("utilities/pt2bench.py")

    def ref_direct(x, ei, ea, nd):
        with tempfile.TemporaryDirectory() as tmp:
            d = {"x": torch.from_numpy(np.asarray(x)), "edge_index": torch.from_numpy(np.asarray(ei, dtype=np.int64))}
            if ea is not None:
                d["edge_attr"] = torch.from_numpy(np.asarray(ea))
            if nd is not None:
                d["node_depth"] = torch.from_numpy(np.asarray(nd))
            torch.save(d, os.path.join(tmp, "g.pt"))
            pt2bench.pt_to_bench(os.path.join(tmp, "g.pt"), os.path.join(tmp, "g.bench"), verbose=False)
            with open(os.path.join(tmp, "g.bench")) as f:
                return f.read()

    direct = {}
    out = {}
    for path in sorted(glob.glob(os.path.join(HERE, "cons_*.npz"))):
        z = np.load(path)
        name = os.path.basename(path)[:-4]
        nd = z["node_depth"] if "node_depth" in z.files else None
        out[name + "|in"] = ref_bench(z["x"], z["edge_index"], z["edge_attr"], nd)
        out[name + "|enf"] = ref_bench(z["enf_x"], z["enf_edge_index"], z["enf_edge_attr"], z["enf_node_depth"])
        direct[name + "|in"] = ref_direct(z["x"], z["edge_index"], z["edge_attr"], nd)
        direct[name + "|in|nodepth"] = ref_direct(z["x"], z["edge_index"], z["edge_attr"], None)
        direct[name + "|enf"] = ref_direct(z["enf_x"], z["enf_edge_index"], z["enf_edge_attr"], z["enf_node_depth"])
        direct[name + "|rs"] = ref_direct(z["rs_x"], z["enf_edge_index"], z["enf_edge_attr"], z["rs_node_depth"])
        if name in ("cons_junk03", "cons_junk10", "cons_only_and", "cons_no_edges"):
            out[name + "|in|norepair"] = ref_bench(z["x"], z["edge_index"], z["edge_attr"], nd, repair_dag=False)
            out[name + "|in|nodiversify"] = ref_bench(z["x"], z["edge_index"], z["edge_attr"], nd, diversify_pad=False)
    for path in sorted(glob.glob(os.path.join(HERE, "gen_n*.npz"))):
        z = np.load(path)
        name = os.path.basename(path)[:-4]
        out[name + "|out"] = ref_bench(z["out_x"], z["out_edge_index"], z["out_edge_attr"], None)
        direct[name + "|out"] = ref_direct(z["out_x"], z["out_edge_index"], z["out_edge_attr"],
                                           z["node_depth"] if "node_depth" in z.files else None)
    z = np.load(os.path.join(HERE, "pad_c17.npz"))
    out["pad_c17|padded"] = ref_bench(z["pad_x"], z["pad_edge_index"], z["pad_edge_attr"], z["pad_node_depth"])
    out["pad_c17|plain"] = ref_bench(z["x"], z["edge_index"], z["edge_attr"], z["node_depth"])
    direct["pad_c17|padded"] = ref_direct(z["pad_x"], z["pad_edge_index"], z["pad_edge_attr"], z["pad_node_depth"])
    direct["pad_c17|plain"] = ref_direct(z["x"], z["edge_index"], z["edge_attr"], z["node_depth"])

    # corner cases kept inline (inputs are stored next to the text)
    extra = {
        "po_chain": ([[0, 0], [2, 0], [1, 0], [1, 0], [1, 0]], [[0, 0, 1, 2, 3], [1, 1, 2, 3, 4]]),
        "po_cycle": ([[0, 0], [1, 0], [1, 0], [2, 0]], [[1, 2, 0], [2, 1, 3]]),
        "no_and": ([[0, 0], [0, 0], [1, 0]], [[0], [2]]),
        "nothing_but_po": ([[1, 0], [1, 0]], [[0], [1]]),
        "and_without_pi": ([[2, 0], [2, 0], [2, 0], [1, 0]], [[0], [3]]),
        "stray_ids": ([[0, 0], [2, 0], [1, 0], [3, 0]], [[-1, 7, 0, 3, 1, 9], [1, 1, 1, 2, 9, 2]]),
        "odd_types": ([[0, 0], [5, 0], [2, 0], [2, 0], [1, 0]], [[0, 1, 1, 2, 3], [2, 2, 3, 3, 4]]),
    }
    cases = {}
    for name, (x, ei) in extra.items():
        x = np.asarray(x, dtype=np.float32)
        ei = np.asarray(ei, dtype=np.int64).reshape(2, -1)
        cases[name] = {"x": x.tolist(), "edge_index": ei.tolist()}
        for tag, kw in (("", {}), ("|norepair", {"repair_dag": False}), ("|nodiversify", {"diversify_pad": False})):
            out["extra_" + name + tag] = ref_bench(x, ei, None, None, **kw)
        direct["extra_" + name] = ref_direct(x, ei, None, None)
    with open(os.path.join(HERE, "bench_text.json"), "w") as f:
        json.dump({"text": out, "direct": direct, "extra_inputs": cases}, f, indent=0, sort_keys=True)
    print(f"wrote bench_text.json: {len(out)} + {len(direct)} texts, {os.path.getsize(os.path.join(HERE, 'bench_text.json')) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
