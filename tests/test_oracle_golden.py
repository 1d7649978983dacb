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


@pytest.mark.parametrize("zd", [4, 48, 200])
def test_position_independent_generator_oracle_at_other_z_widths(zd):
    """``--z-dim`` other than 128 (train.py:374): the fixed-order form against the restatement of the reference's own
    formulation (``cat[h, z]`` through the linear layers, models/generator.py:115-117, 94-98) - same windows and
    targets, h_cond and scores within 1e-4 relative.  No vector of the reference exists at these widths: the two
    restatements agree with each other, and at width 128 each of them is held to the reference's vectors above."""
    from gan_amd.generator import AIGGenerator
    from gan_amd.synth import synth_aig
    torch.manual_seed(zd)
    state = {k: v.detach().clone() for k, v in AIGGenerator(z_dim=zd, candidate_k=16).state_dict().items()}
    assert tuple(state["node_proj.0.weight"].shape) == (128, 128 + zd)
    g = synth_aig(10, 220, 8, seed=zd, window=60, orientation="dataset")
    z = torch.randn(zd, generator=torch.Generator().manual_seed(1))
    fixed = omodel.generator_forward_canonical(state, g.x, g.node_depth, z, 16)
    plain = omodel.generator_forward(state, g.x, g.node_depth, z, 16, keep=True)
    np.testing.assert_allclose(fixed["h_cond"], plain["h_cond"].numpy(), rtol=1e-4, atol=1e-6)
    assert np.array_equal(fixed["edge_index"][1].numpy(), plain["edge_index"][1].numpy())
    slot = {int(v): t for t, v in enumerate(fixed["tgt"])}
    for v, cand, logit in zip(plain["rec"]["tgt"], plain["rec"]["cand"], plain["rec"]["logit"]):
        t = slot[v]
        c = int(fixed["cnt"][t])
        assert np.array_equal(fixed["cand"][t, :c], cand)
        np.testing.assert_allclose(fixed["logits"][t, :c], logit, rtol=1e-4, atol=1e-6)


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


# ------------------------------------------------------------------------------------------------------
# The critic side against the reference's own discriminator.py / train.py (run with the restated attention layer
# standing in for torch_geometric's; tests/golden/make_golden_critic.py): level readout, head, sanitising,
# gradient penalty, loss assembly, optimiser steps.
from util_golden import TrainRun, critic_case_graph, critic_cases  # noqa: E402

_CD, _CPARAMS, _CNAMES = critic_cases()


def _close(a, b, rtol, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert err <= rtol, f"{what}: max error {err:.3e}"


@pytest.mark.parametrize("name", _CNAMES)
def test_critic_oracle_matches_reference_forward(name):
    from gan_amd.train import sanitize_for_D
    g = critic_case_graph(_CD, name)
    p = {k: v.clone().requires_grad_(True) for k, v in _CPARAMS.items()}
    pre = name + "/"
    if name.startswith("padded"):
        # sanitize_for_D of the product (plain tensor code, device-agnostic) and of the checker against the reference's
        s = sanitize_for_D(g)
        assert np.array_equal(s.edge_index.numpy(), _CD[pre + "san_edge_index"])
        assert np.array_equal(s.edge_attr.numpy(), _CD[pre + "san_edge_attr"])
        ei, ea = omodel.sanitize_edges(g.edge_index, g.edge_attr)
        assert np.array_equal(ei.numpy(), _CD[pre + "san_edge_index"]) and np.array_equal(ea.numpy(), _CD[pre + "san_edge_attr"])
    else:
        ei, ea = g.edge_index, g.edge_attr
    score = omodel.discriminator_forward(p, g.x, ei, ea, getattr(g, "node_depth", None) if hasattr(g, "node_depth") else None)
    _close(score.detach().numpy(), _CD[pre + "score"], 1e-6, "score")
    if pre + "grad/out.weight" in _CD.files:
        score.sum().backward()
        for k in p:
            _close(p[k].grad.numpy(), _CD[pre + "grad/" + k], 1e-5, "grad " + k)
    if pre + "gp" in _CD.files:
        p2 = {k: v.clone().requires_grad_(True) for k, v in _CPARAMS.items()}
        gp, gn = omodel.gradient_penalty(p2, g.x, torch.from_numpy(_CD[pre + "gp_fake_x"]), g.node_depth,
                                         torch.from_numpy(_CD[pre + "gp_alpha"]))
        _close(gp.detach().numpy(), _CD[pre + "gp"], 1e-5, "gp")
        _close(gn.numpy(), _CD[pre + "gp_grad_norm"], 1e-6, "grad norm")
        if pre + "gp_grad/out.weight" in _CD.files:
            gp.backward()
            for k in p2:
                if pre + "gp_grad/" + k in _CD.files:
                    _close(p2[k].grad.numpy(), _CD[pre + "gp_grad/" + k], 1e-4, "gp grad " + k)


def replay_cpu(run: TrainRun):
    """The recorded run stepped through by the CPU checker's trainer, fed the run's draws and raw graphs."""
    from gan_amd.data import Batch
    from gan_amd.dataset import pad_data, unpad_data
    from oracle.train_oracle import CpuTrainer
    mn = max(g.x.size(0) for g in run.graphs)
    me = max(g.edge_index.size(1) for g in run.graphs)
    padded = [pad_data(g, mn, me) for g in run.graphs]
    tr = CpuTrainer(gen_state(), run.d_init, run.arg("--candidate-k", 512), mn, me, lr=run.arg("--lr", 1e-4, float),
                    gp_lambda=run.arg("--gp-lambda", 10.0, float), lambda_cons=run.arg("--lambda-cons", 1e-2, float),
                    n_critic=run.arg("--n-critic", 3))
    rows, states, last_epoch = [], {}, 0
    for e, members, call, a, validate in run.steps():
        if e != last_epoch:
            states[last_epoch] = {k: v.detach().clone() for k, v in tr.dp.items()}
            last_epoch = e
        real = Batch.from_data_list([padded[i] for i in members])
        templates = [unpad_data(d) for d in real.to_data_list()]
        k = len(members)
        raw_c = [run.raw(call + j, templates[j].node_depth) for j in range(k)]
        raw_g = [run.raw(call + k + j, templates[j].node_depth) for j in range(k)]
        rows.append(tr.step(real, templates, run.z[call:call + k], run.z[call + k:call + 2 * k],
                            run.alpha[a:a + tr.n_critic], validate=validate, raw_critic=raw_c, raw_generator=raw_g))
    states[last_epoch] = {k: v.detach().clone() for k, v in tr.dp.items()}
    return rows, states


@pytest.mark.parametrize("name", ["bs1", "bs3"])
def test_cpu_trainer_follows_reference_training_run(name):
    """train.py:229-360 executed by the reference (4 minibatches of one graph; 2 epochs of one minibatch of three):
    every metrics.csv row and the critic saved at the end of each epoch."""
    run = TrainRun(name)
    rows, states = replay_cpu(run)
    assert len(rows) == len(run.metrics)
    for step, (got, ref) in enumerate(zip(rows, run.metrics)):
        for k in ("D_real", "D_fake", "gap", "grad_norm", "loss_D", "loss_G", "gp", "pen"):
            assert abs(got[k] - ref[k]) <= 2e-5 * max(abs(ref[k]), 1e-3), (step, k, got[k], ref[k])
    # Adam divides every gradient element by its own running magnitude, so an element that is rounding noise
    # (threads add in another order than the recording's single thread) still moves its weight by up to lr per
    # step: weights are compared to a tenth of one such step
    lr = run.arg("--lr", 1e-4, float)
    for e, sd in states.items():
        ref = run.epoch_state(e)
        assert set(ref) == set(sd)
        for k in ref:
            diff = float((sd[k] - ref[k]).abs().max())
            assert diff <= 1e-6 * float(ref[k].abs().max()) + 0.1 * lr, (e, k, diff)


@pytest.mark.parametrize("case", ["ties", "random", "gaps_and_padding", "one_level"])
def test_segmented_readout_follows_the_level_loop(case):
    """``oracle.model.level_readout_segmented`` (what the CPU step uses on 100k-node graphs) against the literal
    loop over levels of models/discriminator.py:78-87: value, gradient and the gradient of a gradient norm (the
    shape of the gradient penalty's double backward).  Tied maxima - frequent here: nodes with equal features and
    equal neighbourhoods carry equal rows - must send their gradient to the same single row."""
    import torch
    from oracle import model as omodel
    g = torch.Generator().manual_seed({"ties": 1, "random": 2, "gaps_and_padding": 3, "one_level": 4}[case])
    n, hdim = 300, 16
    if case == "ties":
        h = torch.randint(0, 3, (n, hdim), generator=g).double()
        nd = torch.randint(0, 7, (n,), generator=g)
    elif case == "random":
        h = torch.randn(n, hdim, generator=g, dtype=torch.float64)
        nd = torch.randint(0, 40, (n,), generator=g)
    elif case == "gaps_and_padding":
        h = torch.randn(n, hdim, generator=g, dtype=torch.float64).round(decimals=1)
        nd = torch.randint(0, 5, (n,), generator=g) * 3 - 1          # levels 2, 5, 8, 11 filled, the rest empty; -1 = padding
    else:
        h = torch.randn(n, hdim, generator=g, dtype=torch.float64)
        nd = torch.zeros(n, dtype=torch.long)
    w = torch.randn(2 * hdim, generator=g, dtype=torch.float64)
    outs = []
    for fn in (omodel.level_readout, omodel.level_readout_segmented):
        hh = h.clone().requires_grad_(True)
        out = fn(torch.nn.functional.elu(hh * 1.5), nd)
        val = (out.reshape(-1) * w).sum()
        (g1,) = torch.autograd.grad(val, hh, create_graph=True)
        (g2,) = torch.autograd.grad((g1 ** 2).sum().sqrt(), hh)
        outs.append((out.detach(), g1.detach(), g2.detach()))
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-14)
