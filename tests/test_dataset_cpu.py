"""Host-side data plumbing against the reference's own outputs (no GPU needed)."""
import os

import numpy as np
import torch

from gan_amd.data import Batch, Data, GraphLoader
from gan_amd.dataset import load_pt_dataset, pad_data, save_data_list_as_pt, unpad_data
from util_golden import load


def test_pad_unpad_match_reference():
    d = load("pad_c17")
    g = Data(x=torch.from_numpy(d["x"]), edge_index=torch.from_numpy(d["edge_index"]),
             edge_attr=torch.from_numpy(d["edge_attr"]), node_depth=torch.from_numpy(d["node_depth"]))
    p = pad_data(g, 20, 30)
    for k in ("x", "edge_index", "edge_attr", "node_depth"):
        got = getattr(p, k)
        assert np.array_equal(got.numpy(), d["pad_" + k]), k
        assert got.dtype == torch.from_numpy(d["pad_" + k]).dtype, k
    u = unpad_data(p)
    for k in ("x", "edge_index", "edge_attr", "node_depth"):
        assert np.array_equal(getattr(u, k).numpy(), d["unpad_" + k]), k
    # private size attributes do not survive batching: the graph stays padded
    (back,) = Batch.from_data_list([p]).to_data_list()
    assert not hasattr(back, "_orig_n")
    u2 = unpad_data(back)
    assert np.array_equal(u2.x.numpy(), d["unpad2_x"])
    assert np.array_equal(u2.edge_index.numpy(), d["unpad2_edge_index"])


def test_pad_truncates_and_defaults_depth():
    g = Data(x=torch.ones(3, 2), edge_index=torch.tensor([[0, 1, 2], [1, 2, 0]]), edge_attr=torch.ones(3, 1))
    p = pad_data(g, 5, 2)
    assert p.edge_index.tolist() == [[0, 1], [1, 2]] and p._orig_e == 3
    assert p.node_depth.tolist() == [0, 0, 0, -1, -1]


def test_batch_shifts_every_index_including_sentinels():
    a = pad_data(Data(x=torch.zeros(2, 2), edge_index=torch.tensor([[0], [1]]), edge_attr=torch.ones(1, 1)), 3, 2)
    b = Batch.from_data_list([a, a])
    assert b.edge_index.tolist() == [[0, -1, 3, 2], [1, -1, 4, 2]]   # second graph: -1 + 3 = 2, a self loop
    assert b.batch.tolist() == [0, 0, 0, 1, 1, 1] and b.ptr.tolist() == [0, 3, 6]
    parts = b.to_data_list()
    assert parts[1].edge_index.tolist() == [[0, -1], [1, -1]]


def test_pt_round_trip(tmp_path):
    g = Data(x=torch.rand(4, 2), edge_index=torch.tensor([[0, 1], [2, 3]]), edge_attr=torch.tensor([[1], [0]]),
             node_depth=torch.tensor([0, 0, 1, 1]))
    save_data_list_as_pt([g, g], str(tmp_path), prefix="s", start_idx=3)
    assert sorted(os.listdir(tmp_path)) == ["s_000003.pt", "s_000004.pt"]
    back = load_pt_dataset(str(tmp_path), max_files=1)
    assert len(back) == 1 and torch.equal(back[0].x, g.x) and torch.equal(back[0].node_depth, g.node_depth)


def test_loader_shards_batches_across_ranks():
    items = [Data(x=torch.full((1, 2), float(i))) for i in range(10)]
    full = [b.sample_ids for b in GraphLoader(items, batch_size=2, seed=5)]
    r0 = [b.sample_ids for b in GraphLoader(items, batch_size=2, seed=5, rank=0, world_size=2)]
    r1 = [b.sample_ids for b in GraphLoader(items, batch_size=2, seed=5, rank=1, world_size=2)]
    assert len(r0) == len(r1) == 2                      # 5 batches -> 4 usable, 2 per rank
    assert r0 == full[0:4:2] and r1 == full[1:4:2]


def test_batch_vector_and_offsets_are_built_on_first_use():
    from gan_amd.data import Batch, Data
    parts = [Data(x=torch.zeros(k, 2), edge_index=torch.zeros(2, 1, dtype=torch.long)) for k in (3, 1, 4)]
    b = Batch.from_data_list(parts)
    assert "batch" not in b.__dict__ and "ptr" not in b.__dict__
    assert {"batch", "ptr"} <= set(b.keys()) and hasattr(b, "batch")
    assert b.ptr.tolist() == [0, 3, 4, 8]
    assert b.batch.tolist() == [0, 0, 0, 1, 2, 2, 2, 2]
    assert [int(d.x.size(0)) for d in b.to_data_list()] == [3, 1, 4]
    with __import__("pytest").raises(AttributeError):
        b.no_such_attribute


def test_uniform_batch_and_graph_set_views_match_the_list_forms():
    """``Batch.from_uniform`` (tensors that are already concatenated) carries the bookkeeping of
    ``Batch.from_data_list`` on equally sized graphs, and a ``GraphSet`` hands its graphs back unchanged."""
    import torch
    from gan_amd.data import Batch, Data
    from gan_amd.dataset import pad_data
    from gan_amd.graphset import GraphSet
    from gan_amd.sample import block_size
    from gan_amd.synth import synth_aig
    graphs = [synth_aig(4, 20 + i, 3, seed=i, window=8) for i in range(3)]
    n = max(g.x.size(0) for g in graphs)
    me = max(g.edge_index.size(1) for g in graphs)
    padded = [pad_data(g, n, me) for g in graphs]
    ref = Batch.from_data_list(padded)
    uni = Batch.from_uniform(3, n, me, x=ref.x, edge_index=ref.edge_index, edge_attr=ref.edge_attr,
                             node_depth=ref.node_depth)
    assert uni.num_graphs == 3 and torch.equal(uni.batch, ref.batch) and torch.equal(uni.ptr, ref.ptr)
    for a, b in zip(uni.to_data_list(), ref.to_data_list()):
        assert all(torch.equal(getattr(a, k), getattr(b, k)) for k in ("x", "edge_index", "edge_attr", "node_depth"))
    gset = GraphSet.from_data_list([Data(x=p.x, edge_index=g.edge_index, edge_attr=g.edge_attr.long(),
                                         node_depth=p.node_depth) for p, g in zip(padded, graphs)])
    assert gset.B == 3 and gset.n == n and gset.counts() == [int(g.edge_index.size(1)) for g in graphs]
    for g, (p, src) in enumerate(zip(padded, graphs)):
        d = gset.graph(g)
        assert torch.equal(d.edge_index, src.edge_index) and torch.equal(d.x, p.x)
        assert torch.equal(d.edge_attr.view(-1), src.edge_attr.long().view(-1)) and torch.equal(d.node_depth, p.node_depth)
        v = gset.graph(g, copy=False)
        assert torch.equal(v.edge_index, d.edge_index) and v.x.data_ptr() == gset.x[g * n:].data_ptr()
    assert block_size(2389, 64) == 64 and block_size(100000, 64) == 10 and block_size(5_000_000, 64) == 1
