"""`.pt` -> `.bench` export against the text the reference's own chain (pt2graphml.pt_to_graphml +
graphml2bench.convert_graphml_to_bench) wrote for the same graphs (tests/golden/bench_text.json,
made by tests/golden/make_golden_bench.py): byte for byte."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gan_amd.data import Data
from gan_amd.export import bench_text, pt_to_bench_file, write_bench

with open(os.path.join(GOLDEN, "bench_text.json")) as f:
    _G = json.load(f)
TEXT, EXTRA = _G["text"], _G["extra_inputs"]


def _graph(key):
    parts = key.split("|")
    name, which = parts[0], parts[1] if len(parts) > 1 else ""
    if name.startswith("extra_"):
        c = EXTRA[name[len("extra_"):]]
        return Data(x=torch.tensor(c["x"], dtype=torch.float32),
                    edge_index=torch.tensor(c["edge_index"], dtype=torch.long).reshape(2, -1))
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    pre = {"in": "", "enf": "enf_", "out": "out_", "padded": "pad_", "plain": ""}[which]
    return Data(x=torch.from_numpy(z[pre + "x"]), edge_index=torch.from_numpy(z[pre + "edge_index"]),
                edge_attr=torch.from_numpy(z[pre + "edge_attr"]))


def _kwargs(key):
    return {"repair_dag": "norepair" not in key, "diversify_pad": "nodiversify" not in key}


@pytest.mark.parametrize("key", sorted(TEXT))
def test_bench_text_matches_reference_chain(key):
    assert bench_text(_graph(key), **_kwargs(key)) == TEXT[key]


def test_every_golden_family_is_covered():
    fams = {k.split("|")[0].split("_")[0] for k in TEXT}
    assert {"cons", "gen", "pad", "extra"} <= fams and len(TEXT) >= 100


def test_files_round_trip(tmp_path):
    key = "cons_valid_fwd|in"
    g = _graph(key)
    write_bench(g, str(tmp_path / "a" / "g.bench"))
    assert (tmp_path / "a" / "g.bench").read_text() == TEXT[key]
    torch.save({"x": g.x, "edge_index": g.edge_index, "edge_attr": g.edge_attr}, str(tmp_path / "g.pt"))
    out = pt_to_bench_file(str(tmp_path / "g.pt"))
    assert out.endswith("g.bench") and open(out).read() == TEXT[key]


def test_large_graph_is_linear_time():
    from gan_amd.synth import synth_aig
    g = synth_aig(1000, 98000, 1000, seed=0, window=4096)
    txt = bench_text(g)
    lines = txt.splitlines()
    assert len(lines) == 100000 and lines[0] == "INPUT(n0)" and lines[-1].startswith("OUTPUT(n")
    assert sum(l.startswith("n") for l in lines) == 98000


@pytest.mark.parametrize("key", ["cons_junk03|in", "cons_oob_enforce_only|in", "pad_c17|padded", "extra_stray_ids",
                                 "extra_odd_types", "cons_valid_fwd|in"])
def test_graphml_file_chain_gives_the_same_text(key, tmp_path):
    pytest.importorskip("networkx")
    from utilities.graphml2bench import convert_graphml_to_bench
    from utilities.pt2graphml import pt_to_graphml
    g = _graph(key)
    d = {"x": g.x, "edge_index": g.edge_index}
    if getattr(g, "edge_attr", None) is not None:
        d["edge_attr"] = g.edge_attr
    torch.save(d, str(tmp_path / "g.pt"))
    pt_to_graphml(str(tmp_path / "g.pt"), str(tmp_path / "g.graphml"), verbose=False)
    convert_graphml_to_bench(str(tmp_path / "g.graphml"), str(tmp_path / "g.bench"))
    assert (tmp_path / "g.bench").read_text() == TEXT[key]


# ------------------------------------------------------------------ direct exporter (pt2bench.pt_to_bench)
DIRECT = _G["direct"]


def _graph_direct(key):
    parts = key.split("|")
    name, which = parts[0], parts[1] if len(parts) > 1 else ""
    if name.startswith("extra_"):
        return _graph(key)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    if which == "rs":
        d = Data(x=torch.from_numpy(z["rs_x"]), edge_index=torch.from_numpy(z["enf_edge_index"]),
                 edge_attr=torch.from_numpy(z["enf_edge_attr"]))
        d.node_depth = torch.from_numpy(z["rs_node_depth"])
        return d
    d = _graph("|".join(parts[:2]))
    pre = {"in": "", "enf": "enf_", "out": "", "padded": "pad_", "plain": ""}[which]
    if "nodepth" not in parts and (pre + "node_depth") in z.files:
        d.node_depth = torch.from_numpy(z[pre + "node_depth"])
    return d


@pytest.mark.parametrize("key", sorted(DIRECT))
def test_direct_exporter_matches_reference(key):
    from gan_amd.export import bench_text_with_inverters
    assert bench_text_with_inverters(_graph_direct(key), "g.pt") == DIRECT[key]


def test_direct_exporter_file(tmp_path):
    from utilities.pt2bench import pt_to_bench
    key = "cons_valid_fwd|enf"
    g = _graph_direct(key)
    torch.save({"x": g.x, "edge_index": g.edge_index, "edge_attr": g.edge_attr, "node_depth": g.node_depth},
               str(tmp_path / "g.pt"))
    pt_to_bench(str(tmp_path / "g.pt"), str(tmp_path / "o" / "g.bench"), verbose=False)
    assert (tmp_path / "o" / "g.bench").read_text() == DIRECT[key]
