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
from util_golden import gen_state, golden_targets, load, tie_consistent

GEN_CASES = [c for c in golden_names("gen_") if c != "weights"]


@pytest.mark.parametrize("case", GEN_CASES)
def test_generator_oracle_matches_reference(case):
    """The restatement that mirrors the reference op for op (same dense ops, same shapes, one
    thread) reproduces the reference's edges bit for bit, BLAS rounding artefacts included."""
    d = load("gen_" + case)
    nd = torch.from_numpy(d["node_depth"]) if "node_depth" in d.files else None
    out = omodel.generator_forward(gen_state(), torch.from_numpy(d["x"]), nd, torch.from_numpy(d["z"]),
                                   int(d["candidate_k"]), keep=True)
    assert np.array_equal(out["edge_index"].numpy(), d["out_edge_index"])
    assert np.array_equal(out["edge_attr"].numpy(), d["out_edge_attr"])
    assert np.array_equal(out["x"].numpy(), d["out_x"])
    assert np.array_equal(np.asarray(out["rec"]["tgt"]), d["rec_tgt"])
    assert np.array_equal(np.concatenate(out["rec"]["cand"]), d["rec_cand"])
    np.testing.assert_allclose(np.concatenate(out["rec"]["logit"]), d["rec_logit"], rtol=1e-5, atol=1e-6)
    if "h_cond" in d.files:
        np.testing.assert_allclose(out["h_cond"].numpy(), d["h_cond"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", GEN_CASES)
def test_position_independent_generator_oracle_against_reference(case):
    """The fixed-order form the HIP kernels are compared with: same windows, scores within 1e-4
    relative of the reference's, and the same picks wherever the reference's own scores treat
    equal candidates equally (elsewhere the reference's pick depends on BLAS row blocking)."""
    d = load("gen_" + case)
    nd = torch.from_numpy(d["node_depth"]) if "node_depth" in d.files else None
    out = omodel.generator_forward_canonical(gen_state(), torch.from_numpy(d["x"]), nd, torch.from_numpy(d["z"]),
                                             int(d["candidate_k"]))
    assert out["edge_index"].shape == d["out_edge_index"].shape
    assert np.array_equal(out["edge_index"][1].numpy(), d["out_edge_index"][1])
    slot = {int(v): t for t, v in enumerate(out["tgt"])}
    checked = skipped = 0
    for v, cand, logit, picks in golden_targets(d):
        t = slot[v]
        c = int(out["cnt"][t])
        assert np.array_equal(out["cand"][t, :c], cand)
        np.testing.assert_allclose(out["logits"][t, :c], logit, rtol=1e-4, atol=1e-6)
        # exact scores in -> exact picks out (the select routine is the one ATen uses)
        assert ocpu.topk(logit, len(picks)).tolist() == picks
        if tie_consistent(d["x"], cand, logit):
            assert out["sel"][t, :len(picks)].tolist() == picks, (case, v)
            checked += 1
        else:
            skipped += 1
    assert checked > 0
    if skipped == 0:
        assert np.array_equal(out["edge_index"].numpy(), d["out_edge_index"])
        assert np.array_equal(out["edge_attr"].numpy(), d["out_edge_attr"])
        assert np.array_equal(out["x"].numpy(), d["out_x"])


def test_tie_inconsistency_of_reference_scores_is_real():
    """Documents why bit-exact picks need exact scores: in this fixture the reference gave
    candidates with identical features different logits."""
    d = load("gen_pofirst_nodepth_k4")
    flags = [tie_consistent(d["x"], cand, logit) for _, cand, logit, _ in golden_targets(d)]
    assert not all(flags)


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
