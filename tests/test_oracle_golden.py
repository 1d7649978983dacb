"""Pins the CPU checker (oracle/) to vectors captured from the reference itself.

Every later GPU parity test trusts the checker; this file is what earns that trust.
No GPU needed.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from gan_amd.data import Data
from oracle import cpu as ocpu
from oracle import model as omodel


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def gen_state():
    w = load("gen_weights")
    return {k: torch.from_numpy(w[k]) for k in w.files}


@pytest.mark.parametrize("case", golden_names("gen_"))
def test_generator_oracle_matches_reference(case):
    if case == "weights":
        pytest.skip("weight file")
    d = load("gen_" + case)
    nd = torch.from_numpy(d["node_depth"]) if "node_depth" in d.files else None
    keep = "rec_tgt" in d.files
    out = omodel.generator_forward(gen_state(), torch.from_numpy(d["x"]), nd, torch.from_numpy(d["z"]),
                                   int(d["candidate_k"]), keep=keep)
    assert np.array_equal(out["edge_index"].numpy(), d["out_edge_index"])
    assert np.array_equal(out["edge_attr"].numpy(), d["out_edge_attr"])
    assert np.array_equal(out["x"].numpy(), d["out_x"])
    if keep:
        assert np.array_equal(np.asarray(out["rec"]["tgt"]), d["rec_tgt"])
        assert np.array_equal(np.concatenate(out["rec"]["cand"]), d["rec_cand"])
        np.testing.assert_allclose(np.concatenate(out["rec"]["logit"]), d["rec_logit"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out["h_cond"].numpy(), d["h_cond"], rtol=1e-5, atol=1e-6)


def graph_of(d):
    g = Data(x=torch.from_numpy(d["x"].copy()), edge_index=torch.from_numpy(d["edge_index"].copy()),
             edge_attr=torch.from_numpy(d["edge_attr"].copy()))
    if int(d["has_depth_attr"]):
        g.node_depth = torch.from_numpy(d["node_depth"].copy()) if "node_depth" in d.files else None
    return g


@pytest.mark.parametrize("case", golden_names("cons_"))
def test_constraints_oracle_matches_reference(case):
    d = load("cons_" + case)
    fixed = ocpu.enforce_aig_constraints(graph_of(d))
    assert np.array_equal(fixed.edge_index.numpy().reshape(2, -1), d["enf_edge_index"].reshape(2, -1))
    assert np.array_equal(fixed.edge_attr.numpy().reshape(-1), d["enf_edge_attr"].reshape(-1))
    assert np.array_equal(fixed.node_depth.numpy(), d["enf_node_depth"])
    assert np.array_equal(fixed.x.numpy(), d["enf_x"])
    assert fixed.edge_index.dtype == torch.int64 and fixed.node_depth.dtype == torch.int64
    # attributes of the repaired graph
    assert float(ocpu.aig_constraint_penalties(fixed)["total"]) == float(d["enf_pen_total"])
    errs = ocpu.validate_strict_aig(fixed, check_all_on_path=True)
    assert "\n".join(errs) == str(d["enf_val_text"])
    rs = ocpu.resync_attrs(fixed)
    assert np.array_equal(rs.x.numpy(), d["rs_x"])
    assert np.array_equal(rs.node_depth.numpy(), d["rs_node_depth"])
    if case == "oob_enforce_only":
        return
    pen = ocpu.aig_constraint_penalties(graph_of(d))
    for k in ocpu.PEN_KEYS:
        assert float(pen[k]) == float(d["pen_" + k]), k
    errs = ocpu.validate_strict_aig(graph_of(d), check_all_on_path=True)
    assert len(errs) == int(d["val_n"])
    assert "\n".join(errs) == str(d["val_text"])
    assert len(ocpu.validate_strict_aig(graph_of(d), check_all_on_path=False)) == int(d["val_n_nopath"])


def test_topk_oracle_matches_torch_table():
    d = load("topk_kat")
    offs, vals = d["offsets"], d["values"]
    for r in range(len(offs) - 1):
        row = vals[offs[r]:offs[r + 1]]
        assert int(ocpu.topk(row, 1)[0]) == int(d["top1"][r]), (d["meta"][r], 1)
        if row.shape[0] >= 2:
            assert ocpu.topk(row, 2).tolist() == d["top2"][r].tolist(), (d["meta"][r], 2)


def test_topk_oracle_matches_live_torch_topk():
    rng = np.random.default_rng(0)
    for _ in range(400):
        c = int(rng.integers(1, 300))
        row = rng.choice(np.linspace(-1, 1, int(rng.integers(1, 12))).astype(np.float32), size=c)
        for k in (1, 2):
            if k <= c:
                want = torch.topk(torch.from_numpy(row), k).indices.tolist()
                assert ocpu.topk(row, k).tolist() == want
