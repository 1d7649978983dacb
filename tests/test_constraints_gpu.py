"""GPU parity of the constraint kernels: bit-exact against the reference's own outputs
(tests/golden/cons_*.npz) and, beyond those sizes, against the CPU checker."""
import numpy as np
import pytest
import torch

from conftest import golden_names
from gan_amd import _abi
from gan_amd.constraints import (aig_constraint_penalties, enforce_aig_constraints, repair_and_resync, resync_attrs,
                                 validate_strict_aig)
from gan_amd.data import Data
from gan_amd.synth import synth_aig
from oracle import cpu as ocpu
from util_golden import load

pytestmark = pytest.mark.gpu
DEV = "cuda"
PEN_KEYS = ["pi_in", "po_out", "backedge", "selfloop", "dup_fanin", "indeg_mse", "fanin_type", "reach_violation",
            "total"]


def graph_of(d, dev=DEV):
    g = Data(x=torch.from_numpy(d["x"].copy()).to(dev), edge_index=torch.from_numpy(d["edge_index"].copy()).to(dev),
             edge_attr=torch.from_numpy(d["edge_attr"].copy()).to(dev))
    if int(d["has_depth_attr"]):
        g.node_depth = torch.from_numpy(d["node_depth"].copy()).to(dev) if "node_depth" in d.files else None
    return g


@pytest.mark.parametrize("case", golden_names("cons_"))
def test_constraints_match_reference_outputs(case):
    d = load("cons_" + case)
    g = graph_of(d)
    same = enforce_aig_constraints(g)
    assert same is g
    assert g.edge_index.dtype == torch.int64 and g.edge_attr.dtype == torch.int64 and g.node_depth.dtype == torch.int64
    assert np.array_equal(g.edge_index.cpu().numpy().reshape(2, -1), d["enf_edge_index"].reshape(2, -1))
    assert np.array_equal(g.edge_attr.cpu().numpy().reshape(-1), d["enf_edge_attr"].reshape(-1))
    assert tuple(g.edge_attr.shape) == (g.edge_index.size(1), 1)
    assert np.array_equal(g.node_depth.cpu().numpy(), d["enf_node_depth"])
    assert np.array_equal(g.x.cpu().numpy(), d["enf_x"])
    # attributes of the repaired graph
    assert float(aig_constraint_penalties(g)["total"]) == float(d["enf_pen_total"])
    assert "\n".join(validate_strict_aig(g, check_all_on_path=True)) == str(d["enf_val_text"])
    rs = resync_attrs(g)
    assert np.array_equal(rs.x.cpu().numpy(), d["rs_x"])
    assert np.array_equal(rs.node_depth.cpu().numpy(), d["rs_node_depth"])
    if case == "oob_enforce_only":
        return
    pen = aig_constraint_penalties(graph_of(d))
    for k in PEN_KEYS:
        assert pen[k].dim() == 0 and pen[k].dtype == torch.float32
        assert float(pen[k]) == float(d["pen_" + k]), k
    errs = validate_strict_aig(graph_of(d), check_all_on_path=True)
    assert "\n".join(errs) == str(d["val_text"])
    assert len(validate_strict_aig(graph_of(d), check_all_on_path=False)) == int(d["val_n_nopath"])


def junk_graph(n, e, seed, depth_mode):
    rng = np.random.default_rng(seed)
    types = rng.choice([0, 1, 2], size=n, p=[0.25, 0.15, 0.6])
    x = np.stack([types, rng.integers(0, 3, n)], 1).astype(np.float32)
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    k = e // 5
    src[:k], dst[:k] = src[k:2 * k], dst[k:2 * k]          # repeats
    dst[-(e // 50 + 1):] = src[-(e // 50 + 1):]            # self loops
    g = Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.stack([src, dst]).astype(np.int64)),
             edge_attr=torch.from_numpy(rng.integers(0, 2, e).astype(np.int64)).view(-1, 1))
    if depth_mode == "long":
        g.node_depth = torch.from_numpy(rng.integers(-1, 9, n).astype(np.int64))
    elif depth_mode == "float":
        g.node_depth = torch.from_numpy((rng.random(n) * 9 - 1).astype(np.float32))
    return g


def to_dev(g):
    return g.clone().to(DEV)


def assert_same_graph(a: Data, b: Data):
    assert np.array_equal(a.edge_index.cpu().numpy().reshape(2, -1), b.edge_index.cpu().numpy().reshape(2, -1))
    assert np.array_equal(a.edge_attr.cpu().numpy().reshape(-1), b.edge_attr.cpu().numpy().reshape(-1))
    assert np.array_equal(a.node_depth.cpu().numpy(), b.node_depth.cpu().numpy())
    assert np.array_equal(a.x.cpu().numpy(), b.x.cpu().numpy())


@pytest.mark.parametrize("n,e,depth_mode", [(300, 900, "long"), (5000, 30000, "float"), (20000, 50000, None),
                                            (777, 40000, "long"), (300, 70000, None)])
def test_constraints_match_checker_on_random_graphs(n, e, depth_mode):
    g = junk_graph(n, e, seed=n + e, depth_mode=depth_mode)
    pen = aig_constraint_penalties(to_dev(g))
    ref = ocpu.aig_constraint_penalties(g.clone())
    for k in PEN_KEYS:
        assert float(pen[k]) == float(ref[k]), k
    assert validate_strict_aig(to_dev(g)) == ocpu.validate_strict_aig(g.clone())
    fixed = enforce_aig_constraints(to_dev(g))
    fixed_ref = ocpu.enforce_aig_constraints(g.clone())
    assert_same_graph(fixed, fixed_ref)
    assert validate_strict_aig(fixed) == ocpu.validate_strict_aig(fixed_ref.clone())
    assert_same_graph(resync_attrs(fixed), ocpu.resync_attrs(fixed_ref))


@pytest.mark.parametrize("n,e,depth_mode", [(300, 900, "long"), (5000, 30000, "float"), (777, 40000, None)])
def test_repair_and_resync_is_repair_then_resync(n, e, depth_mode):
    """The sampling tail leaves out the repair's own depth walk, whose result the resync replaces
    (AIGgan.py:105-110): every tensor of the result equals the two calls in a row and the CPU checker's."""
    g = junk_graph(n, e, seed=3 * n + e, depth_mode=depth_mode)
    fused = repair_and_resync(to_dev(g))
    two = resync_attrs(enforce_aig_constraints(to_dev(g)))
    assert_same_graph(fused, two)
    assert_same_graph(fused, ocpu.resync_attrs(ocpu.enforce_aig_constraints(g.clone())))
    # an edge-free graph takes the early exits of both halves
    empty = Data(x=g.x.clone(), edge_index=torch.zeros((2, 0), dtype=torch.long), edge_attr=torch.zeros((0, 1), dtype=torch.long))
    assert_same_graph(repair_and_resync(to_dev(empty)), resync_attrs(enforce_aig_constraints(to_dev(empty))))


@pytest.mark.parametrize("orientation", ["forward", "dataset"])
def test_constraints_on_full_size_aig(orientation):
    """100k nodes: a valid forward AIG validates clean and is a fixed point of the repair."""
    g = synth_aig(1000, 98000, 1000, seed=1, window=4096, orientation=orientation)
    g.edge_attr = g.edge_attr.long()
    gd = to_dev(g)
    errs = validate_strict_aig(gd)
    pen = aig_constraint_penalties(gd)
    ref_pen = ocpu.aig_constraint_penalties(g.clone())
    for k in PEN_KEYS:
        assert float(pen[k]) == float(ref_pen[k]), k
    assert errs == ocpu.validate_strict_aig(g.clone())
    if orientation == "forward":
        assert errs == [] or all("Dangling" in e for e in errs)   # unused ANDs may dangle
        before = gd.edge_index.clone()
        fixed = enforce_aig_constraints(gd)
        # same edge set, regrouped by target id and ordered by source position
        a = set(zip(before[0].tolist(), before[1].tolist()))
        b = set(zip(fixed.edge_index[0].tolist(), fixed.edge_index[1].tolist()))
        assert a == b
    fixed_ref = ocpu.enforce_aig_constraints(g.clone())
    assert_same_graph(enforce_aig_constraints(to_dev(g)), fixed_ref)
    assert_same_graph(resync_attrs(to_dev(g)), ocpu.resync_attrs(g.clone()))


@pytest.mark.parametrize("fanin", [255, 256, 700])
def test_kahn_depth_with_wide_fanin(fanin):
    """An acyclic graph with one in-degree around / beyond what a byte counter holds: the Kahn sweep
    (resync depths, cycle check of the validator) must not depend on where its counters live."""
    n = fanin + 40
    src = list(range(fanin)) + [fanin + i for i in range(39)] + [3, 5]
    dst = [fanin] * fanin + [fanin + i + 1 for i in range(39)] + [fanin + 20, fanin + 20]
    x = np.zeros((n, 2), np.float32)
    x[fanin:, 0] = 2
    x[-1, 0] = 1
    rng = np.random.default_rng(fanin)
    g = Data(x=torch.from_numpy(x), edge_index=torch.tensor([src, dst], dtype=torch.long),
             edge_attr=torch.from_numpy(rng.integers(0, 2, len(src)).astype(np.int64)).view(-1, 1))
    rs, ref = resync_attrs(to_dev(g)), ocpu.resync_attrs(g.clone())
    assert int(ref.node_depth.max()) == 40
    assert_same_graph(rs, ref)
    assert validate_strict_aig(to_dev(g)) == ocpu.validate_strict_aig(g.clone())


def test_empty_and_cpu_inputs():
    g = Data(x=torch.tensor([[0., 1.], [2., 2.], [1., 0.]], device=DEV),
             edge_index=torch.zeros((2, 0), dtype=torch.long, device=DEV),
             edge_attr=torch.zeros((0, 1), dtype=torch.long, device=DEV))
    ref = Data(x=g.x.cpu().clone(), edge_index=g.edge_index.cpu(), edge_attr=g.edge_attr.cpu())
    assert validate_strict_aig(g) == ocpu.validate_strict_aig(ref)
    assert float(aig_constraint_penalties(g)["total"]) == 0.0
    out = enforce_aig_constraints(g)
    assert out.node_depth.tolist() == [-1, -1, -1] and out.x[:, 1].tolist() == [0, 0, 0]
    with pytest.raises(_abi.AigAbiError):
        enforce_aig_constraints(ref)
