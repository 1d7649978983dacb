"""`.bench` -> training tensors against the reference's own dataset chain (andAIG2Graphml.py ->
GraphML -> generate_pt.py; fixtures from tests/golden/make_golden_ingest.py): identical x, edge_index,
edge_attr, node_depth and dtypes."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gan_amd.ingest import bench_to_data, bench_to_pt

with open(os.path.join(GOLDEN, "ingest_cases.json")) as f:
    CASES = json.load(f)


@pytest.mark.parametrize("name", sorted(CASES))
def test_bench_to_data_matches_reference_chain(name):
    c = CASES[name]
    d = bench_to_data(c["bench"])
    assert [str(d.x.dtype), str(d.edge_index.dtype), str(d.edge_attr.dtype), str(d.node_depth.dtype)] == c["dtypes"]
    assert np.array_equal(d.x.numpy(), np.asarray(c["x"], dtype=np.float32))
    assert np.array_equal(d.edge_index.numpy(), np.asarray(c["edge_index"], dtype=np.int64).reshape(2, -1))
    assert np.array_equal(d.edge_attr.numpy(), np.asarray(c["edge_attr"], dtype=np.float32).reshape(-1, 1))
    assert np.array_equal(d.node_depth.numpy(), np.asarray(c["node_depth"], dtype=np.float32))


def test_file_round_trip_and_loader(tmp_path):
    from gan_amd.dataset import load_pt_dataset
    c = CASES["medium"]
    (tmp_path / "m.bench").write_text(c["bench"])
    bench_to_pt(str(tmp_path / "m.bench"), str(tmp_path / "graph" / "m.pt"))
    (g,) = load_pt_dataset(str(tmp_path / "graph"))
    assert np.array_equal(g.edge_index.numpy(), np.asarray(c["edge_index"]).reshape(2, -1))
    assert g.node_depth.dtype == torch.float32


def test_rejects_what_the_reference_cannot_resolve():
    with pytest.raises(ValueError, match="line 4"):
        bench_to_data("INPUT(a)\nINPUT(b)\nOUTPUT(g)\ng = AND(a, zz)\n")
    with pytest.raises(ValueError, match="not an AIG bench statement"):
        bench_to_data("INPUT(a)\nfoo bar\n")
    d = bench_to_data("# comment\n\nINPUT(a)\nINPUT(b)\nOUTPUT(g)\ng = AND(a, b)\n")
    assert d.x[:, 0].tolist() == [0.0, 0.0, 2.0, 1.0]
