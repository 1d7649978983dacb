"""The graph walks and the generator / critic at the sizes of the million-node configuration, against the CPU
checker.  The level-by-level traversals (breadth-first depth of the repair, forward / backward reachability of the
penalties and the validator, Kahn peel of the validator and of the attribute resync) run in three regimes
depending on the node count:

    n <= 131,072                 frontier queues, visited bitmap and byte-wide Kahn counters in LDS
    131,072 < n <= 1,048,576     bitmap and queues in LDS, Kahn counters in global memory
    n > 1,048,576                everything in global memory (k_traverse)

and fall back in other ways on an in-degree above 255 (Kahn counters no longer fit a byte) or a frontier above
2,048 nodes (queue entries spill to global memory).  Every regime is exercised with a cyclic graph, an acyclic one,
a fan-in above 255 and a frontier of tens of thousands of nodes; results are bit-exact (integers)."""
import numpy as np
import pytest
import torch

from gan_amd.constraints import (aig_constraint_penalties, enforce_aig_constraints, repair_and_resync, resync_attrs,
                                 validate_strict_aig)
from gan_amd.data import Data
from gan_amd.discriminator import AIGDiscriminator
from gan_amd.generator import AIGGenerator
from gan_amd.synth import synth_aig
from oracle import cpu as ocpu
from oracle import model as omodel
from test_constraints_gpu import PEN_KEYS, assert_same_graph, to_dev
from util_golden import gen_state

pytestmark = pytest.mark.gpu
DEV = "cuda"

# one below / above each regime boundary, the middle of the second regime, the million-node size itself
SIZES = [131072, 131073, 300000, 1000000, 1048576, 1048577, 1300000]


def junk_graph(n, seed):
    """Random typed nodes (a quarter of them PIs: the first frontier is n/4 nodes wide), 2.5 n random edges with
    repeats and self loops - cycles everywhere - and one AND that 300 distinct nodes feed."""
    rng = np.random.default_rng(seed)
    e = int(2.5 * n)
    types = rng.choice([0, 1, 2], size=n, p=[0.25, 0.15, 0.6])
    x = np.stack([types, rng.integers(0, 3, n)], 1).astype(np.float32)
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    k = e // 10
    src[:k], dst[:k] = src[k:2 * k], dst[k:2 * k]
    dst[-(e // 100):] = src[-(e // 100):]
    hub = n // 2
    x[hub, 0] = 2
    src[2 * k:2 * k + 300] = rng.choice(n, 300, replace=False)
    dst[2 * k:2 * k + 300] = hub
    g = Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.stack([src, dst]).astype(np.int64)),
             edge_attr=torch.from_numpy(rng.integers(0, 2, e).astype(np.int64)).view(-1, 1))
    g.node_depth = torch.from_numpy(rng.integers(-1, 40, n).astype(np.int64))
    return g


def wide_acyclic_graph(n, seed):
    """A valid forward AIG of n - 1 nodes plus one last node with 400 fan-ins (acyclic: every edge goes up in id),
    so that the Kahn peels run to the end with an in-degree no byte holds."""
    n_pi, n_po = max(2, n // 100), max(1, n // 100)
    g = synth_aig(n_pi, n - 1 - n_pi - n_po, n_po, seed=seed, window=2048, compute_depth=False)
    rng = np.random.default_rng(seed)
    extra_src = rng.choice(n_pi + (n - 1 - n_pi - n_po), 400, replace=False)     # PIs and ANDs only
    x = torch.cat([g.x, torch.tensor([[2.0, 0.0]])])
    ei = torch.cat([g.edge_index, torch.from_numpy(np.stack([extra_src, np.full(400, n - 1)]).astype(np.int64))], 1)
    ea = torch.cat([g.edge_attr.long(), torch.zeros((400, 1), dtype=torch.long)])
    return Data(x=x, edge_index=ei, edge_attr=ea)


def check_all(g):
    pen, ref = aig_constraint_penalties(to_dev(g)), ocpu.aig_constraint_penalties(g.clone())
    for k in PEN_KEYS:
        assert float(pen[k]) == float(ref[k]), k
    assert validate_strict_aig(to_dev(g)) == ocpu.validate_strict_aig(g.clone())
    fixed, fixed_ref = enforce_aig_constraints(to_dev(g)), ocpu.enforce_aig_constraints(g.clone())
    assert_same_graph(fixed, fixed_ref)
    rs_ref = ocpu.resync_attrs(g.clone())
    assert_same_graph(resync_attrs(to_dev(g)), rs_ref)
    assert_same_graph(repair_and_resync(to_dev(g)), ocpu.resync_attrs(fixed_ref))
    return rs_ref


@pytest.mark.parametrize("n", SIZES)
def test_walks_on_cyclic_junk(n):
    g = junk_graph(n, seed=n % 1000)
    rs = check_all(g)
    assert int(rs.node_depth.max()) == 0            # a cycle: resync_attrs reports zeros (AIGgan.py:51-61)


@pytest.mark.parametrize("n", SIZES)
def test_walks_on_acyclic_graph_with_a_wide_fanin(n):
    g = wide_acyclic_graph(n, seed=7 + n % 1000)
    rs = check_all(g)
    assert int(rs.node_depth.max()) > 100           # the longest-path peel went all the way


@pytest.mark.parametrize("orientation", ["forward", "dataset"])
def test_million_node_training_graph(orientation):
    """The graph of the million-node configuration (10k PI / 980k AND / 10k PO, window 4096) in both edge
    orientations: constraints, one generator pass at candidate-k 64 and the critic's score."""
    g = synth_aig(10000, 980000, 10000, seed=3, window=4096, orientation=orientation, compute_depth=False)
    g.edge_attr = g.edge_attr.long()
    g.node_depth = check_all(g).node_depth.float()          # longest-path levels, as the dataset files carry them
    n = g.x.size(0)
    # generator: plan (order, windows) and one pass, bit for bit
    state = gen_state()
    G = AIGGenerator(candidate_k=64).to(DEV)
    G.load_state_dict(state)
    z = torch.randn(128, generator=torch.Generator().manual_seed(11))
    tpl = Data(x=g.x.to(DEV), node_depth=g.node_depth.to(DEV))
    out = G(tpl, z.to(DEV), return_debug=True)
    ref = omodel.generator_forward_canonical(state, g.x, g.node_depth, z, 64)
    plan = out._debug["plan"]
    assert np.array_equal(plan.order.cpu().numpy(), ref["order"])
    assert np.array_equal(out.edge_index.cpu().numpy(), ref["edge_index"].numpy())
    assert np.array_equal(out.edge_attr.cpu().numpy(), ref["edge_attr"].numpy())
    assert np.array_equal(out.x.cpu().numpy(), ref["x"].numpy())
    raw = Data(x=ref["x"], edge_index=ref["edge_index"], edge_attr=ref["edge_attr"], node_depth=g.node_depth.clone())
    check_all(raw)                                           # the wiring an untrained generator emits: two huge hubs
    # critic score on the real graph
    p = omodel.init_discriminator_params(4)
    D = AIGDiscriminator()
    D.load_state_dict(p)
    D = D.to(DEV)
    with torch.no_grad():
        score = D(Data(x=g.x.to(DEV), edge_index=g.edge_index.to(DEV), edge_attr=g.edge_attr.to(DEV),
                       node_depth=g.node_depth.long().to(DEV)))
        ref_score = omodel.discriminator_forward(p, g.x, g.edge_index, g.edge_attr, g.node_depth.long())
    assert abs(float(score) - float(ref_score)) <= 1e-4 * max(abs(float(ref_score)), 1e-3), (float(score), float(ref_score))
    assert n == 1000000
