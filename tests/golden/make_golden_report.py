#!/usr/bin/env python3
"""Regenerates ``report_cases.json``: what the reference's inspection scripts
(data_files/check_pt.py:35-91 and utilities/scripts_check.py:8-67, run unmodified from
/root/reference in the build container) report for graphs of the ``cons_*`` fixtures.  Only the
fixture names, the returned rows and the printed lines are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_report.py
"""
import contextlib
import io
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

from make_golden import Data, import_reference, load_reference_file  # noqa: E402

CASES = ["cons_junk03", "cons_junk10", "cons_raw_c17_fwd_k4", "cons_no_edges", "cons_only_and", "cons_one_pi_two_and"]


def main():
    import_reference()
    check_pt = load_reference_file("data_files/check_pt.py")
    scripts_check = load_reference_file("utilities/scripts_check.py")
    pt2graphml = load_reference_file("utilities/pt2graphml.py")
    graphml2bench = load_reference_file("utilities/graphml2bench.py")
    out = {}
    for name in CASES:
        z = np.load(os.path.join(HERE, name + ".npz"))
        for tag, keys in (("in", ("x", "edge_index", "edge_attr", "node_depth")),
                          ("enf", ("enf_x", "enf_edge_index", "enf_edge_attr", "enf_node_depth"))):
            if keys[3] not in z.files:
                continue
            x, ei, ea, nd = (torch.from_numpy(np.asarray(z[k])) for k in keys)
            with tempfile.TemporaryDirectory() as tmp:
                pt = os.path.join(tmp, "g.pt")
                # the reference reads an attribute-style graph object: a file written here, by this script
                torch.save(Data(x=x, edge_index=ei, edge_attr=ea, node_depth=nd), pt)
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    row = check_pt.calculate_parameters(pt)
                torch.save({"x": x, "edge_index": ei, "edge_attr": ea, "node_depth": nd}, pt)
                gml, bench = os.path.join(tmp, "g.graphml"), os.path.join(tmp, "g.bench")
                pt2graphml.pt_to_graphml(pt, gml, verbose=False)
                graphml2bench.convert_graphml_to_bench(gml, bench)
                buf2 = io.StringIO()
                with contextlib.redirect_stdout(buf2):
                    scripts_check.check_aig_graphml(gml)
                    scripts_check.check_bench_file(bench)
            out[f"{name}|{tag}"] = {"row": row, "depth_lines": buf.getvalue().splitlines(),
                                    "check_lines": [ln.replace(tmp, "<tmp>") for ln in buf2.getvalue().splitlines()]}
    with open(os.path.join(HERE, "report_cases.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=1, sort_keys=True)
    print(f"wrote report_cases.json ({len(out)} cases)")


if __name__ == "__main__":
    main()
