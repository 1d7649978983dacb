"""Sets of graphs through one call (``gan_amd.graphset``, the ``*_batch`` entry points): every graph of a set must
come out exactly as the single-graph function - itself held to the reference's outputs in
test_constraints_gpu.py - returns it.  Integers, bit-exact."""
import numpy as np
import pytest
import torch

from gan_amd.constraints import (aig_constraint_penalties, clean_depth_set, enforce_aig_constraints, enforce_set, pad_set,
                                 penalties_set, repair_and_resync, repair_and_resync_set, resync_attrs, resync_set)
from gan_amd.data import Batch, Data
from gan_amd.dataset import pad_data
from gan_amd.graphset import GraphSet
from gan_amd.synth import synth_aig
from oracle import cpu as ocpu
from test_constraints_gpu import PEN_KEYS, assert_same_graph, junk_graph, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"


def mixed_graphs(n, depth_mode, seed):
    """Graphs of n nodes: random junk with different edge counts (cycles, repeats, self loops), valid forward
    AIGs (acyclic: the resync reports levels), one graph with a single edge and one without any."""
    gs = [junk_graph(n, e, seed + i, depth_mode) for i, e in enumerate((3 * n, n // 2, 7 * n, 5))]
    for i in range(2):
        n_pi, n_po = max(2, n // 20), max(1, n // 25)
        a = synth_aig(n_pi, n - n_pi - n_po, n_po, seed=seed + 10 + i, window=64)
        a.edge_attr = a.edge_attr.long()
        if depth_mode == "float":
            a.node_depth = a.node_depth.float()
        elif depth_mode is None:
            a.node_depth = None
        else:
            a.node_depth = a.node_depth.long()
        gs.append(a)
    empty = junk_graph(n, 10, seed + 99, depth_mode)
    empty.edge_index = torch.zeros((2, 0), dtype=torch.long)
    empty.edge_attr = torch.zeros((0, 1), dtype=torch.long)
    gs.insert(2, empty)
    return gs


@pytest.mark.parametrize("n,depth_mode", [(60, "long"), (700, "float"), (3000, None), (20000, "long")])
def test_every_graph_of_a_set_matches_its_single_graph_result(n, depth_mode):
    graphs = mixed_graphs(n, depth_mode, seed=n)
    on_dev = [to_dev(g) for g in graphs]
    gset = GraphSet.from_data_list(on_dev)
    assert gset.B == len(graphs) and gset.edge_index.shape == (gset.B, 2, gset.ecap)
    # penalties
    pen = penalties_set(gset).cpu().numpy()
    for g, d in enumerate(on_dev):
        ref = aig_constraint_penalties(d.clone())
        for i, k in enumerate(PEN_KEYS):
            assert float(pen[g, i]) == float(ref[k]), (g, k)
    # repair (with its breadth-first depths), then attributes of the repaired graphs
    fixed = enforce_set(gset)
    singles = [enforce_aig_constraints(d.clone()) for d in on_dev]
    for g, ref in enumerate(singles):
        assert_same_graph(fixed.graph(g), ref)
    assert fixed.counts() == [int(r.edge_index.size(1)) for r in singles]
    # the repair without its level walk, and the walk by itself afterwards (how the training step hands fakes over)
    late = enforce_set(gset, want_depth=False)
    assert late.node_depth is None and torch.equal(late.x, fixed.x) and late.counts() == fixed.counts()
    assert torch.equal(clean_depth_set(gset), fixed.node_depth)
    pen_fixed = penalties_set(fixed).cpu().numpy()
    for g, ref in enumerate(singles):
        assert float(pen_fixed[g, 8]) == float(aig_constraint_penalties(ref.clone())["total"])
    # resync of the raw graphs (cycles: zeros; the valid AIGs: their levels) and the fused sampling tail
    rs = resync_set(GraphSet.from_data_list([to_dev(g) for g in graphs]))
    deep = 0
    for g, d in enumerate(graphs):
        ref = resync_attrs(to_dev(d))
        assert_same_graph(rs.graph(g), ref)
        deep = max(deep, int(ref.node_depth.max()))
    assert deep > 3
    tail = repair_and_resync_set(GraphSet.from_data_list([to_dev(g) for g in graphs]))
    for g, d in enumerate(graphs):
        assert_same_graph(tail.graph(g), repair_and_resync(to_dev(d)))
    # against the CPU checker directly for two of the graphs
    for g in (0, 4):
        assert_same_graph(fixed.graph(g), ocpu.enforce_aig_constraints(graphs[g].clone()))


@pytest.mark.parametrize("max_edges", [40, 5000])
def test_padded_minibatch_of_a_set(max_edges):
    n = 300
    graphs = mixed_graphs(n, "long", seed=5)
    fixed = enforce_set(GraphSet.from_data_list([to_dev(g) for g in graphs]))
    ei, ea = pad_set(fixed, max_edges)
    ref = Batch.from_data_list([pad_data(d, n, max_edges) for d in fixed.graphs()])
    assert torch.equal(ei, ref.edge_index) and torch.equal(ea, ref.edge_attr)
    assert torch.equal(fixed.x, ref.x) and torch.equal(fixed.node_depth, ref.node_depth)


def test_shared_depth_vector_and_single_graph_sets():
    """Samples of one template share its depth vector (stride 0); a set of one graph is the plain call."""
    n = 500
    base = junk_graph(n, 4 * n, 3, "float")
    variants = []
    for i in range(5):
        v = junk_graph(n, 3 * n + 50 * i, 40 + i, "float")
        v.x = base.x.clone()                      # same node types, different edges
        v.node_depth = base.node_depth.clone()
        variants.append(v)
    gset = GraphSet.from_data_list([to_dev(v) for v in variants])
    gset.node_depth = base.node_depth.to(DEV)     # one vector for all
    assert gset.depth_shared
    fixed = enforce_set(gset)
    for g, v in enumerate(variants):
        ref = enforce_aig_constraints(to_dev(v))
        got = fixed.graph(g)
        assert_same_graph(got, ref)
    one = enforce_set(GraphSet.from_data_list([to_dev(variants[0])]))
    assert_same_graph(one.graph(0), enforce_aig_constraints(to_dev(variants[0])))


def test_large_set_runs_in_the_global_memory_regimes():
    """64 graphs of 20,000 nodes: 1.28M union nodes, beyond what the LDS-resident walks hold."""
    n = 20000
    graphs = [junk_graph(n, 2 * n + 100 * i, 900 + i, "long") for i in range(62)]
    for i in range(2):
        a = synth_aig(200, n - 400, 200, seed=i, window=512)
        a.edge_attr = a.edge_attr.long()
        a.node_depth = a.node_depth.long()
        graphs.append(a)
    gset = GraphSet.from_data_list([to_dev(g) for g in graphs])
    tail = repair_and_resync_set(gset)
    pen = penalties_set(gset).cpu().numpy()
    for g in (0, 17, 61, 62, 63):
        assert_same_graph(tail.graph(g), repair_and_resync(to_dev(graphs[g])))
        ref = aig_constraint_penalties(to_dev(graphs[g]))
        for i, k in enumerate(PEN_KEYS):
            assert float(pen[g, i]) == float(ref[k]), (g, k)


# ------------------------------------------------------------------------------------------ generator
def _templates(n_list_seed, n_pi, n_and, n_po, count, orientation="dataset"):
    out = []
    for i in range(count):
        g = synth_aig(n_pi + i, n_and - i, n_po, seed=n_list_seed + i, window=48 + 16 * i, orientation=orientation)
        out.append(Data(x=g.x.to(DEV), node_depth=g.node_depth.to(DEV)))
    return out


@pytest.mark.parametrize("K", [16, 64, 200, 300])
@pytest.mark.parametrize("score_dtype", ["fp32", "bf16"])
def test_generator_on_a_set_equals_one_graph_at_a_time(K, score_dtype):
    from gan_amd.generator import AIGGenerator
    from util_golden import gen_state
    G = AIGGenerator(candidate_k=K, score_dtype=score_dtype).to(DEV)
    G.load_state_dict(gen_state())
    gen = torch.Generator().manual_seed(K)
    # (a) the graphs of a padded minibatch: different templates of one node count
    tpls = _templates(70, 20, 900, 12, 5)
    zs = torch.randn(5, 128, generator=gen).to(DEV)
    gset = G.forward_set(tpls, zs)
    assert gset.B == 5 and gset.n == tpls[0].x.size(0)
    for g, t in enumerate(tpls):
        assert_same_graph(gset.graph(g), G(t, zs[g]))
    assert gset.counts() == [int(c) for c in gset.count.tolist()]
    again = G.forward_set(tpls, zs)                     # the stacked plans come from the cache now
    for g in range(5):                                  # (slots behind a graph's count are unspecified)
        assert_same_graph(again.graph(g), gset.graph(g))
    # (b) samples of one template
    zs = torch.randn(7, 128, generator=gen).to(DEV)
    sset = G.forward_set(tpls[2], zs)
    for g in range(7):
        assert_same_graph(sset.graph(g), G(tpls[2], zs[g]))
    # ... and straight through repair + resync, against the one-at-a-time sampling tail
    tail = repair_and_resync_set(sset)
    for g in range(7):
        assert_same_graph(tail.graph(g), repair_and_resync(G(tpls[2], zs[g])))


def test_generator_set_of_one_and_templates_without_targets():
    from gan_amd.generator import AIGGenerator
    from util_golden import gen_state
    G = AIGGenerator(candidate_k=32).to(DEV)
    G.load_state_dict(gen_state())
    tpl = _templates(5, 6, 130, 4, 1)[0]
    z = torch.randn(1, 128, generator=torch.Generator().manual_seed(1)).to(DEV)
    assert_same_graph(G.forward_set([tpl], z).graph(0), G(tpl, z[0]))
    assert_same_graph(G.forward_set(tpl, z).graph(0), G(tpl, z[0]))
    # only PIs: no edge is emitted, the repair and the penalties take their early exits
    bare = Data(x=torch.zeros((40, 2), device=DEV), node_depth=torch.zeros(40, device=DEV))
    zs = torch.randn(3, 128, generator=torch.Generator().manual_seed(2)).to(DEV)
    gset = G.forward_set(bare, zs)
    assert gset.counts() == [0, 0, 0]
    ref = enforce_aig_constraints(G(bare, zs[0]))
    fixed = enforce_set(gset)
    for g in range(3):
        assert_same_graph(fixed.graph(g), ref)
    assert float(penalties_set(gset).abs().sum()) == 0.0


def test_bulk_block_at_the_bulk_sampling_configuration():
    """A block of 64 draws from a c7552-sized template at candidate-k 256 with bf16 scoring - the bulk-sampling
    configuration - through ``sample_sets``: spot checks against one sample at a time, and the properties every
    repaired sample has whatever its size (fan-ins of PO / AND nodes, acyclic, levels consistent with the edges)."""
    from gan_amd.generator import AIGGenerator
    from gan_amd.sample import sample_one, sample_sets
    from gan_amd.synth import ISCAS85_SHAPES
    from util_golden import gen_state
    name, n_pi, n_po, n_and = ISCAS85_SHAPES[-1]
    g = synth_aig(n_pi, n_and, n_po, seed=0, window=max(8, n_and // 8 + n_pi), orientation="dataset")
    tpl = Data(x=g.x.to(DEV), node_depth=g.node_depth.to(DEV))
    G = AIGGenerator(candidate_k=256, score_dtype="bf16").to(DEV).eval()
    G.load_state_dict(gen_state())
    zs = torch.randn(64, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
    (gset,) = list(sample_sets(G, tpl, [zs]))
    assert gset.B == 64 and gset.n == g.x.size(0)
    for i in (0, 31, 63):
        assert_same_graph(gset.graph(i), sample_one(G, tpl, zs[i]))
    types = g.x[:, 0].long().numpy()
    for i in range(0, 64, 7):
        d = gset.graph(i)
        src, dst = d.edge_index.cpu().numpy()
        indeg = np.bincount(dst, minlength=types.size)
        assert (indeg[types == 0] == 0).all() and (indeg[types == 1] <= 1).all() and (indeg[types == 2] <= 2).all()
        assert set(indeg[types == 2].tolist()) <= {0, 2}
        depth = d.node_depth.cpu().numpy()
        assert (depth[dst] > depth[src]).all()              # longest-path levels: every edge goes up
        has_in = indeg > 0
        assert (depth[~has_in] == 0).all() and (depth[has_in] == np.maximum.reduceat(depth[src][np.argsort(dst, kind="stable")] + 1, np.flatnonzero(np.diff(np.concatenate([[-1], np.sort(dst)])) > 0))).all()
        inv = np.bincount(dst, weights=d.edge_attr.cpu().numpy().reshape(-1), minlength=types.size)
        assert np.array_equal(d.x[:, 1].cpu().numpy(), inv.astype(np.float32))


@pytest.mark.parametrize("n_and,k", [(300, 16), (30000, 64)])
def test_penalties_with_an_order_hint(n_and, k):
    """The reachability term of the penalties as two sweeps along an order hint (the generator's processing rank,
    which its sets carry): same nine numbers as the CPU checker, and the same whatever the hint holds - no hint, the
    reverse order (a permutation every edge violates), vectors that are no permutation, one template's rank shared by
    all graphs - because the hint is checked on the device and a graph it does not fit is walked level by level."""
    from gan_amd.generator import AIGGenerator
    from util_golden import gen_state
    G = AIGGenerator(candidate_k=k).to(DEV)
    G.load_state_dict(gen_state())
    templates = []
    for i, orientation in enumerate(("dataset", "forward", "dataset")):
        t = synth_aig(40, n_and, 30, seed=60 + i, window=max(8, n_and // 6), orientation=orientation)
        templates.append(Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV)))
    zs = torch.randn(3, 128, generator=torch.Generator().manual_seed(4)).to(DEV)
    raw = G.forward_set(templates, zs)
    n = raw.n
    rank = raw.order_hint
    assert rank is not None and rank.dtype == torch.int32 and rank.numel() == 3 * n
    want = []
    for g in range(3):
        d = raw.graph(g)
        ref = ocpu.aig_constraint_penalties(Data(x=d.x.cpu(), edge_index=d.edge_index.cpu(), edge_attr=d.edge_attr.cpu()))
        want.append([float(ref[key]) for key in PEN_KEYS])
    want = np.asarray(want, dtype=np.float32)
    assert want[:, 7].min() > 0          # some nodes are off every PI -> PO path: the term under test is not trivially zero
    got = penalties_set(raw).cpu().numpy()
    assert np.array_equal(got, want), (got, want)
    flat = rank.reshape(3, n)
    perm = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(s)) for s in range(3)]).to(DEV).int()
    variants = {
        "none": None,
        "reversed": (n - 1 - flat).contiguous(),
        "all_zero": torch.zeros_like(flat),
        "out_of_range": (flat + n).contiguous(),
        "one_value_twice": torch.cat([flat[:, :1], flat[:, :-1]], 1).contiguous(),
        "random_permutation": perm,
        "first_graphs_rank_for_all": flat[0].contiguous(),
        "wrong_dtype": flat.long(),
    }
    for name, hint in variants.items():
        raw.order_hint = hint
        assert np.array_equal(penalties_set(raw).cpu().numpy(), got), name
    # graphs full of cycles, with hints of every kind
    junk = GraphSet.from_data_list([to_dev(junk_graph(n, 3 * n, 200 + i, "long")) for i in range(3)])
    base = penalties_set(junk).cpu().numpy()
    for g in range(3):
        ref = ocpu.aig_constraint_penalties(junk.graph(g).cpu())
        assert [float(ref[key]) for key in PEN_KEYS] == [float(v) for v in base[g, :9]]
    for name, hint in variants.items():
        junk.order_hint = hint
        assert np.array_equal(penalties_set(junk).cpu().numpy(), base), name
