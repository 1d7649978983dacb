#!/usr/bin/env python3
"""Regenerates ``ingest_cases.json``: ABC-style AIG ``.bench`` texts and the tensors the reference's
dataset chain (data_files/andAIG2Graphml.py -> GraphML -> data_files/generate_pt.py, run unmodified
from /root/reference in the build container) makes of them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ingest.py
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import networkx as nx
import numpy as np

from make_golden import import_reference, load_reference_file  # noqa: E402


def abc_bench(rng, n_pi, n_and, n_po, window, vdd=False, twin=False):
    """An AIG in the style ABC writes: header, INPUT/OUTPUT lines, NOT aliases next to their use."""
    lines = ['# Benchmark "case" written by ABC on a rainy day']
    pis = [f"pi{i}" for i in range(n_pi)]
    lines += [f"INPUT({p})" for p in pis]
    gates = [f"n{n_pi + k}" for k in range(n_and)]
    kinds = rng.integers(0, 4, n_po)              # 0 AND itself, 1 NOT, 2 BUFF, 3 BUFF of a NOT
    drivers = rng.choice(np.arange(max(n_and // 2, 0), n_and), n_po, replace=n_po > n_and - n_and // 2)
    po_names = []
    for j in range(n_po):
        po_names.append(gates[drivers[j]] if kinds[j] == 0 else f"po{j}")
    po_names = list(dict.fromkeys(po_names))
    lines += [f"OUTPUT({p})" for p in po_names]
    body = []
    if vdd:
        body.append("one = vdd")
        pis = pis + ["one"]
    pool = list(pis)
    made_not = set()
    for k, g in enumerate(gates):
        lo = max(0, len(pool) - window)
        a, b = rng.choice(np.arange(lo, len(pool)), 2, replace=len(pool) - lo < 2)
        if twin and k % 17 == 5:
            b = a
        ins = []
        for s, inv in ((pool[a], rng.random() < 0.5), (pool[b], rng.random() < 0.5)):
            if inv:
                if s not in made_not:
                    body.append(f"{s}_n = NOT({s})")
                    made_not.add(s)
                ins.append(f"{s}_n")
            elif twin and rng.random() < 0.05:
                body.append(f"{s}_b{k} = BUFF({s})")
                ins.append(f"{s}_b{k}")
            else:
                ins.append(s)
        body.append(f"{g} = AND({ins[0]}, {ins[1]})")
        pool.append(g)
    for j in range(n_po):
        d = gates[drivers[j]]
        if kinds[j] == 1 and f"po{j}" in po_names:
            body.append(f"po{j} = NOT({d})")
        elif kinds[j] == 2 and f"po{j}" in po_names:
            body.append(f"po{j} = BUFF({d})")
        elif kinds[j] == 3 and f"po{j}" in po_names:
            if d not in made_not:
                body.append(f"{d}_n = NOT({d})")
                made_not.add(d)
            body.append(f"po{j} = BUFF({d}_n)")
    return "\n".join(lines + body) + "\n"


def main():
    import_reference()
    a2g = load_reference_file("data_files/andAIG2Graphml.py")
    gpt = load_reference_file("data_files/generate_pt.py")
    rng = np.random.default_rng(2024)
    texts = {
        "c17_like": abc_bench(rng, 5, 6, 2, 8),
        "small_vdd": abc_bench(rng, 4, 12, 3, 6, vdd=True),
        "twins_and_buffers": abc_bench(rng, 8, 60, 6, 16, twin=True),
        "medium": abc_bench(rng, 30, 400, 20, 64),
        "deep": abc_bench(rng, 6, 300, 9, 3, twin=True),
        "po_is_pi_not": '# ABC\nINPUT(a)\nINPUT(b)\nOUTPUT(x)\nOUTPUT(y)\nOUTPUT(g)\ng = AND(a, b)\nx = NOT(a)\ny = BUFF(b)\n',
    }
    out = {}
    for name, text in texts.items():
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, name + ".bench")
            with open(path, "w") as f:
                f.write(text)
            a2g.INPUT_BENCH, a2g.GML_DUMP_LOC = path, tmp
            a2g.dumpGMLGraph(a2g.parseAIGBenchAndCreateNetworkXGraph())
            data = gpt.pygDataFromNetworkx(nx.read_graphml(os.path.join(tmp, name + ".bench.graphml")))
        out[name] = {"bench": text, "x": data.x.numpy().tolist(), "edge_index": data.edge_index.numpy().tolist(),
                     "edge_attr": data.edge_attr.numpy().tolist(), "node_depth": data.node_depth.numpy().tolist(),
                     "dtypes": [str(data.x.dtype), str(data.edge_index.dtype), str(data.edge_attr.dtype), str(data.node_depth.dtype)]}
        print(name, "nodes", data.x.shape[0], "edges", data.edge_index.shape[1])
    with open(os.path.join(HERE, "ingest_cases.json"), "w") as f:
        json.dump(out, f)
    print("wrote ingest_cases.json", os.path.getsize(os.path.join(HERE, "ingest_cases.json")) // 1024, "KiB")


if __name__ == "__main__":
    main()
