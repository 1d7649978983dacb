"""The repair's node_depth (breadth-first level from the PIs over the cleaned edges, models/constraints.py:541-546,
641) is computed by a sweep over the nodes in PI / AND / PO order, a tile of 4,096 positions at a time with the
levels of the last 16,384 positions in LDS.  Graphs built to reach its corners, against the CPU checker (bit-exact):
chains that run through a whole tile and across tiles (one relaxation round per hop), predecessor lists longer than
the two a thread keeps in registers - inside the tile, in the LDS window and from before it - unreachable stretches,
node counts around the tile and window sizes, and a set of graphs through one call."""
import numpy as np
import pytest
import torch

from gan_amd.constraints import clean_depth_set, enforce_aig_constraints, enforce_set
from gan_amd.data import Data
from gan_amd.graphset import GraphSet
from oracle import cpu as ocpu
from test_constraints_gpu import assert_same_graph, to_dev

pytestmark = pytest.mark.gpu


def typed(n, n_pi, n_po):
    x = np.zeros((n, 2), np.float32)
    x[n_pi:n - n_po, 0] = 2
    x[n - n_po:, 0] = 1
    return x


def graph(x, src, dst, seed=0):
    rng = np.random.default_rng(seed)
    e = len(src)
    return Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.stack([src, dst]).astype(np.int64)),
                edge_attr=torch.from_numpy(rng.integers(0, 2, e).astype(np.int64)).view(-1, 1))


def chain_graph(n, stride, seed):
    """Every AND is fed by the node `stride` positions before it; a few also by a node near the front, and past
    position 20,000 a third also by a random node more than 20,000 positions back (outside the LDS window).  With
    stride 1 the level rises by one per position between two shortcuts: chains of hundreds of hops inside a tile."""
    n_pi, n_po = 4, 3
    x = typed(n, n_pi, n_po)
    rng = np.random.default_rng(seed)
    v = np.arange(n_pi, n)
    near = np.maximum(v - stride, 0)
    far = (rng.random(len(v)) * np.maximum(v - 20000, 1)).astype(np.int64)
    keep = (rng.random(len(v)) < 0.002) | ((v > 20000) & (rng.random(len(v)) < 0.3))   # few shortcuts to the front
    src = np.concatenate([near, far[keep]])
    dst = np.concatenate([v, v[keep]])
    p = rng.permutation(len(src))
    return graph(x, src[p], dst[p], seed)


def long_list_graph(n, seed):
    """Nodes with dozens of predecessors: some inside their tile, some in the LDS window, some from long before,
    some unreachable (fed only by a node that nothing feeds), plus edges the cleaning drops (downward in position)."""
    n_pi, n_po = max(3, n // 50), max(2, n // 100)
    x = typed(n, n_pi, n_po)
    rng = np.random.default_rng(seed)
    src, dst = [], []
    for v in range(n_pi, n):
        r = rng.random()
        if r < 0.05:
            continue                                       # no predecessor at all: unreachable unless fed later
        k = 40 if r < 0.15 else 2
        lo = n_pi if (r < 0.5 and v > n_pi) else 0         # the first kind is only reachable through ANDs
        near = rng.integers(max(lo, v - 3000), v, size=k // 2)
        far = rng.integers(lo, v - 15000 if v - 15000 > lo else v, size=k - k // 2)
        for u in np.concatenate([near, far]):
            src.append(int(u)); dst.append(v)
    m = len(src) // 20
    src += list(rng.integers(0, n, m)); dst += list(rng.integers(0, n, m))      # any direction, self loops too
    return graph(x, np.array(src), np.array(dst), seed)


@pytest.mark.parametrize("n,stride", [(4095, 1), (4096, 1), (4097, 1), (9000, 1), (20000, 3), (40000, 22), (16385, 1)])
def test_chains_through_tiles(n, stride):
    g = chain_graph(n, stride, seed=n + stride)
    ref = ocpu.enforce_aig_constraints(g.clone())
    assert_same_graph(enforce_aig_constraints(to_dev(g)), ref)
    if stride == 1:
        assert int(ref.node_depth.max()) > 300              # the levels really do rise along the chain


@pytest.mark.parametrize("n", [3000, 12289, 50000])
def test_long_predecessor_lists(n):
    g = long_list_graph(n, seed=n)
    ref = ocpu.enforce_aig_constraints(g.clone())
    assert int((ref.node_depth < 0).sum()) > 0 and int(ref.node_depth.max()) > 2
    assert_same_graph(enforce_aig_constraints(to_dev(g)), ref)


def test_set_of_graphs_in_one_call():
    n = 6000
    graphs = [chain_graph(n, 1 + 2 * i, seed=90 + i) for i in range(3)] + [long_list_graph(n, seed=5)]
    gs = GraphSet.from_data_list([to_dev(g) for g in graphs])
    depth_alone = clean_depth_set(gs)
    fixed = enforce_set(gs).graphs()
    refs = [ocpu.enforce_aig_constraints(g.clone()) for g in graphs]
    for got, ref in zip(fixed, refs):
        assert_same_graph(got, ref)
    # the levels by themselves (what the training step asks for after a repair without them)
    assert np.array_equal(depth_alone.cpu().numpy(), np.concatenate([r.node_depth.numpy() for r in refs]))


# ---- resync_attrs: longest-path levels by the same sweep when the PI / AND / PO order is a topological one
def relabel(g, perm):
    """The same graph with node i renamed perm[i]."""
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    x = g.x.numpy()[inv]
    ei = perm[g.edge_index.numpy()]
    return Data(x=torch.from_numpy(x.copy()), edge_index=torch.from_numpy(ei.astype(np.int64)), edge_attr=g.edge_attr.clone())


@pytest.mark.parametrize("n,stride", [(16385, 1), (20000, 1), (30000, 7)])
def test_resync_levels_by_the_sweep(n, stride):
    from gan_amd.constraints import resync_attrs
    g = chain_graph(n, stride, seed=3 * n)                  # every edge runs upward in position: the sweep takes it
    ref = ocpu.resync_attrs(g.clone())
    assert int(ref.node_depth.max()) > (n - 10) // stride - 2     # longest path: the whole chain
    assert_same_graph(resync_attrs(to_dev(g)), ref)
    g2 = long_list_graph(n, seed=n + 1)                      # has downward edges and self loops: the walk takes it
    assert_same_graph(resync_attrs(to_dev(g2)), ocpu.resync_attrs(g2.clone()))


def test_resync_of_an_acyclic_graph_whose_order_is_not_topological():
    from gan_amd.constraints import resync_attrs
    n = 19000
    g = chain_graph(n, 2, seed=11)
    rng = np.random.default_rng(4)
    perm = np.arange(n)
    ands = np.arange(4, n - 3)
    perm[ands] = rng.permutation(ands)                      # ANDs renamed at random: types keep their places
    h = relabel(g, perm)
    ref = ocpu.resync_attrs(h.clone())
    assert int(ref.node_depth.max()) > 1000                  # still acyclic, still deep
    assert_same_graph(resync_attrs(to_dev(h)), ref)


def test_resync_set_chooses_per_graph():
    from gan_amd.constraints import resync_set
    n = 17000
    ordered = chain_graph(n, 1, seed=21)
    rng = np.random.default_rng(5)
    perm = np.arange(n)
    perm[4:n - 3] = rng.permutation(np.arange(4, n - 3))
    unordered = relabel(chain_graph(n, 3, seed=22), perm)
    cyclic = chain_graph(n, 5, seed=23)
    ei = cyclic.edge_index.clone()
    ei[:, 0] = torch.tensor([n // 2, n // 2 - 5])            # one edge turned around: a cycle
    cyclic.edge_index = ei
    graphs = [ordered, unordered, cyclic, chain_graph(n, 40, seed=24)]
    gs = resync_set(GraphSet.from_data_list([to_dev(g) for g in graphs]))
    for got, g in zip(gs.graphs(), graphs):
        ref = ocpu.resync_attrs(g.clone())
        assert np.array_equal(got.node_depth.cpu().numpy(), ref.node_depth.numpy())
        assert np.array_equal(got.x.cpu().numpy(), ref.x.numpy())
    assert int(gs.graphs()[2].node_depth.max()) == 0         # the cyclic one reports zeros
