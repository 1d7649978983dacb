#!/usr/bin/env python3
"""Regenerates critic_*.npz / train_*.npz from the reference checkout: everything AROUND the attention
layer, executed by the reference's own code.

Run in the build container only (``/root/reference`` must exist):  ``python tests/golden/make_golden_critic.py``

``models/discriminator.py`` and ``train.py`` of the reference are imported unmodified.  What they import
from torch_geometric - which is not installable here - is supplied in memory:

  torch_geometric.data.Data / Batch     this repository's attribute-bag classes (as make_golden.py does)
  torch_geometric.loader.DataLoader     a loader that yields ``Batch.from_data_list`` of consecutive graphs
                                        (PyG's shuffles; the order of the minibatches is not arithmetic)
  torch_geometric.utils.scatter         imported by the reference but never called; a stub that raises
  torch_geometric.nn.GATConv            an nn.Module with PyG's parameter names whose forward is the
                                        restated layer, ``oracle.model.gat_conv``

So the attention layer's arithmetic stays what this repository restated (PARITY UNPINNED for that layer
alone), while the level readout, the head, ``sanitize_for_D``, ``gradient_penalty(use_edges=False)``, the
loss assembly, the order of optimiser steps and the random draws of the loop (train.py:108-152, 229-360;
discriminator.py:42-100) are the reference's own execution.  Only tensors are stored.

What is captured
  critic_cases.npz   critic parameters; per graph: inputs, ``AIGDiscriminator.forward`` score,
                     ``sanitize_for_D`` output, ``gradient_penalty(..., use_edges=False)`` with the alpha drawn
  train_<run>.npz    a whole ``train(args)`` run on a tiny dataset: dataset tensors, initial critic
                     parameters, every z and alpha the loop drew, the raw graphs its generator emitted,
                     every ``metrics.csv`` row, the critic parameters saved at the end of each epoch
"""
import csv
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn as nn

from gan_amd.data import Batch, Data
from gan_amd.synth import synth_aig
from oracle import model as omodel

import make_golden as mg


class StandInGATConv(nn.Module):
    """Parameter names and shapes of ``GATConv(in, out, heads, edge_dim=1, concat=True)``; the forward pass is
    the restated layer of oracle/model.py."""

    def __init__(self, in_channels, out_channels, heads=1, edge_dim=None, concat=True):
        super().__init__()
        assert edge_dim == 1 and concat
        self.heads = heads
        hc = heads * out_channels
        self.lin = nn.Linear(in_channels, hc, bias=False)
        self.lin_edge = nn.Linear(edge_dim, hc, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_edge = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.empty(hc))
        with torch.no_grad():
            for t in (self.lin.weight, self.lin_edge.weight, self.att_src, self.att_dst, self.att_edge):
                a = (6.0 / (t.size(-2) + t.size(-1))) ** 0.5
                t.uniform_(-a, a)
            self.bias.uniform_(0.0, 0.1)      # PyG starts at zero; a non-zero bias makes the fixture tell more

    def forward(self, x, edge_index, edge_attr):
        p = {k: v for k, v in self.named_parameters()}
        return omodel.gat_conv(p, "", x, edge_index, edge_attr, self.heads)


class SequentialLoader:
    def __init__(self, dataset, batch_size=1, **_ignored):
        self.dataset, self.batch_size = list(dataset), batch_size

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, len(self.dataset), self.batch_size):
            yield Batch.from_data_list(self.dataset[i:i + self.batch_size])


def import_reference_critic():
    mg.import_reference()          # torch_geometric.data, models.generator / constraints, utilities.data_process
    tg = sys.modules["torch_geometric"]
    for name, attrs in (("nn", {"GATConv": StandInGATConv}), ("loader", {"DataLoader": SequentialLoader}),
                        ("utils", {"scatter": lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("scatter"))})):
        m = types.ModuleType("torch_geometric." + name)
        for k, v in attrs.items():
            setattr(m, k, v)
        setattr(tg, name, m)
        sys.modules["torch_geometric." + name] = m
    from models import discriminator
    assert os.path.realpath(discriminator.__file__).startswith(os.path.realpath(mg.REF)), discriminator.__file__
    train_mod = mg.load_reference_file("train.py")
    return discriminator, train_mod


class DrawRecorder:
    """Records what ``torch.randn`` / ``torch.rand`` return while active (the loop's z and alpha draws)."""

    def __init__(self):
        self.z, self.alpha = [], []

    def __enter__(self):
        self._randn, self._rand = torch.randn, torch.rand

        def randn(*a, **k):
            out = self._randn(*a, **k)
            self.z.append(out.detach().clone())
            return out

        def rand(*a, **k):
            out = self._rand(*a, **k)
            self.alpha.append(out.detach().clone())
            return out

        torch.randn, torch.rand = randn, rand
        return self

    def __exit__(self, *exc):
        torch.randn, torch.rand = self._randn, self._rand


def graph_arrays(prefix, d):
    out = {prefix + "x": mg.np_of(d.x), prefix + "edge_index": mg.np_of(d.edge_index),
           prefix + "edge_attr": mg.np_of(d.edge_attr)}
    if getattr(d, "node_depth", None) is not None:
        out[prefix + "node_depth"] = mg.np_of(d.node_depth)
    return out


# --------------------------------------------------------------------------- critic cases
def critic_cases(discriminator, train_mod, data_process):
    torch.manual_seed(77)
    D = discriminator.AIGDiscriminator(in_dim=2, hidden_dim=128, heads=4)
    arrays = {"param/" + k: mg.np_of(v) for k, v in D.state_dict().items()}
    cases = {}
    cases["fwd_small"] = synth_aig(6, 40, 4, seed=1, window=16)
    cases["dataset_mid"] = synth_aig(20, 300, 12, seed=2, window=60, orientation="dataset")
    t = synth_aig(12, 90, 6, seed=3, window=30)
    nd = t.node_depth.clone().long()
    nd[::5] = -1                                                      # nodes without a level
    nd[nd == 3] = 9                                                   # an empty level in between, a far one
    cases["gaps_and_ignored"] = Data(x=t.x, edge_index=t.edge_index, edge_attr=t.edge_attr, node_depth=nd)
    t = synth_aig(8, 50, 4, seed=4, window=20)
    cases["all_ignored"] = Data(x=t.x, edge_index=t.edge_index, edge_attr=t.edge_attr,
                                node_depth=torch.full((t.x.size(0),), -1, dtype=torch.long))
    t = synth_aig(8, 50, 4, seed=5, window=20)
    cases["no_depth"] = Data(x=t.x, edge_index=t.edge_index, edge_attr=t.edge_attr)
    t = synth_aig(8, 50, 4, seed=6, window=20)
    cases["no_edge_attr"] = Data(x=t.x, edge_index=t.edge_index, node_depth=t.node_depth)
    parts = [data_process.pad_data(synth_aig(5 + i, 30 + 9 * i, 3, seed=10 + i, window=12, orientation="dataset"), 70, 140)
             for i in range(3)]
    cases["padded_batch3"] = Batch.from_data_list(parts)             # -1 columns; shifted ones in graphs 1, 2
    cases["padded_single"] = data_process.pad_data(synth_aig(7, 44, 5, seed=20, window=14, orientation="dataset"), 80, 120)
    names = []
    for name, d in cases.items():
        names.append(name)
        pre = f"{name}/"
        padded = name.startswith("padded")
        arrays.update(graph_arrays(pre, d) if d.edge_attr is not None else
                      {pre + "x": mg.np_of(d.x), pre + "edge_index": mg.np_of(d.edge_index),
                       pre + "node_depth": mg.np_of(d.node_depth)})
        inp = train_mod.sanitize_for_D(d) if padded else d
        if padded:
            arrays[pre + "san_edge_index"] = mg.np_of(inp.edge_index)
            arrays[pre + "san_edge_attr"] = mg.np_of(inp.edge_attr)
        D.zero_grad()
        score = D(inp)
        assert score.shape == (1,)
        arrays[pre + "score"] = mg.np_of(score)
        score.sum().backward()
        if name in ("dataset_mid", "gaps_and_ignored", "padded_batch3"):
            for k, v in D.named_parameters():
                arrays[pre + "grad/" + k] = mg.np_of(v.grad)
        if padded:
            # gradient penalty against a perturbed copy of the node features, as the loop calls it
            g = torch.Generator().manual_seed(len(names))
            fake = Data(x=d.x + torch.rand(d.x.shape, generator=g), edge_index=d.edge_index, edge_attr=d.edge_attr,
                        node_depth=d.node_depth)
            D.zero_grad()
            with DrawRecorder() as rec:
                gp, gn = train_mod.gradient_penalty(D, d, fake, torch.device("cpu"), use_edges=False)
            assert len(rec.alpha) == 1 and not rec.z
            gp.backward()
            arrays[pre + "gp_fake_x"] = mg.np_of(fake.x)
            arrays[pre + "gp_alpha"] = mg.np_of(rec.alpha[0])
            arrays[pre + "gp"] = mg.np_of(gp)
            arrays[pre + "gp_grad_norm"] = mg.np_of(gn)
            if name == "padded_batch3":
                for k, v in D.named_parameters():
                    if v.grad is not None:
                        arrays[pre + "gp_grad/" + k] = mg.np_of(v.grad)
    arrays["names"] = np.array(names)
    mg.save("critic_cases", **arrays)


# --------------------------------------------------------------------------- whole training runs
def train_run(name, train_mod, AIGGenerator, discriminator, graphs, argv, seed):
    """One ``train(args)`` of the reference on ``graphs``, everything a replay needs recorded."""
    gen_state = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, "gen_weights.npz")).items()}
    made = {}

    def make_G(*a, **k):
        G = AIGGenerator(*a, **k)
        G.load_state_dict(gen_state)       # after its constructor consumed the RNG, as any loaded checkpoint would be
        made["G"], made["raw"] = G, []
        G.register_forward_hook(lambda m, i, out: made["raw"].append(
            (mg.np_of(out.x).copy(), mg.np_of(out.edge_index).copy(), mg.np_of(out.edge_attr).copy())))
        return G

    def make_D(*a, **k):
        D = discriminator.AIGDiscriminator(*a, **k)
        made["D_init"] = {kk: vv.detach().clone() for kk, vv in D.state_dict().items()}
        return D

    train_mod.AIGGenerator, train_mod.AIGDiscriminator = make_G, make_D
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        ds = os.path.join(tmp, "dataset")
        os.makedirs(ds)
        for i, g in enumerate(graphs):
            torch.save(g, os.path.join(ds, f"g{i:03d}.pt"))
        os.chdir(tmp)                       # the loop writes results/ISCAS85/pt_fake relative to the cwd
        try:
            args = build_args(train_mod, ["--dataset-dir", ds, "--out-dir", os.path.join(tmp, "out")] + argv)
            torch.manual_seed(seed)
            with DrawRecorder() as rec:
                train_mod.train(args)
        finally:
            os.chdir(cwd)
        with open(os.path.join(tmp, "out", "metrics.csv")) as f:
            rows = list(csv.DictReader(f))
        finals = [torch.load(os.path.join(tmp, "out", f"discriminator_epoch{e}.pt")) for e in range(args.epochs)]
        g_final = torch.load(os.path.join(tmp, "out", f"generator_epoch{args.epochs - 1}.pt"))
    for k, v in g_final.items():
        assert torch.equal(v, gen_state[k]), "the reference's generator moved: " + k
    fields = list(rows[0].keys())
    arrays = {"metrics": np.array([[float(r[k]) for k in fields] for r in rows], dtype=np.float64),
              "metric_fields": np.array(fields), "argv": np.array(argv),
              "z": np.stack([mg.np_of(z) for z in rec.z]), "alpha": np.array([float(a) for a in rec.alpha], dtype=np.float32),
              "n_graphs": np.array(len(graphs))}
    for i, g in enumerate(graphs):
        arrays.update(graph_arrays(f"graph{i}/", g))
    for k, v in made["D_init"].items():
        arrays["d_init/" + k] = mg.np_of(v)
    for e, sd in enumerate(finals):
        for k, v in sd.items():
            arrays[f"d_epoch{e}/" + k] = mg.np_of(v)
    # The raw graphs the reference's generator emitted, call by call (per minibatch: the critic's fakes of every
    # member, then the generator step's).  Its CPU BLAS scores equal candidate rows unequally now and then (see
    # tests/util_golden.tie_consistent), so a position-independent generator cannot be asked to re-derive them:
    # replays feed these graphs in.  `raw_canonical[c]` says whether the fixed-order restatement emits the same graph.
    K = args.candidate_k
    max_nodes = max(g.x.size(0) for g in graphs)
    max_edges = max(g.edge_index.size(1) for g in graphs)
    padded = [mg_data_process.pad_data(g, max_nodes, max_edges) for g in graphs]
    assert len(made["raw"]) == len(rec.z)
    bs = args.batch_size
    call, same = 0, []
    for epoch in range(args.epochs):
        for b0 in range(0, len(graphs), bs):
            members = list(range(b0, min(len(graphs), b0 + bs)))
            for _phase in range(2):
                for i in members:
                    tpl = padded[i]          # templates keep their padded size (pad_data's _orig_n does not survive batching)
                    x_ref, ei_ref, ea_ref = made["raw"][call]
                    op = omodel.generator_forward(gen_state, tpl.x, tpl.node_depth, rec.z[call], K)
                    assert np.array_equal(op["edge_index"].numpy(), ei_ref) and np.array_equal(op["x"].numpy(), x_ref), \
                        (name, call, "the op-for-op restatement does not follow the reference's generator")
                    can = omodel.generator_forward_canonical(gen_state, tpl.x, tpl.node_depth, rec.z[call], K)
                    same.append(np.array_equal(can["edge_index"].numpy(), ei_ref) and
                                np.array_equal(can["edge_attr"].numpy(), ea_ref))
                    arrays[f"raw{call}/x"], arrays[f"raw{call}/edge_index"], arrays[f"raw{call}/edge_attr"] = x_ref, ei_ref, ea_ref
                    call += 1
    assert call == len(rec.z)
    arrays["raw_canonical"] = np.array(same)
    arrays["seed"] = np.array(seed)
    mg.save("train_" + name, **arrays)
    print(f"    seed {seed}: {len(rows)} steps, {len(rec.z)} z draws ({int(np.sum(same))} raw graphs equal to the "
          f"fixed-order restatement's), {len(rec.alpha)} alpha draws")


def build_args(train_mod, argv):
    """The argparse block of the reference's ``__main__`` section (train.py:366-387), which cannot be imported:
    flags and defaults as read from that file."""
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--dataset-dir", type=str, required=True)
    p.add_argument("--out-dir", type=str, default="out_gen")
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--batch-size", type=int, default=1)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--g-hidden", type=int, default=128)
    p.add_argument("--d-hidden", type=int, default=128)
    p.add_argument("--z-dim", type=int, default=128)
    p.add_argument("--candidate-k", type=int, default=512)
    p.add_argument("--max-nodes", type=int, default=None)
    p.add_argument("--max-edges", type=int, default=None)
    p.add_argument("--lambda-cons", type=float, default=1e-2)
    p.add_argument("--gp-lambda", type=float, default=10.0)
    p.add_argument("--n-critic", type=int, default=3)
    p.add_argument("--save-every", type=int, default=10)
    p.add_argument("--save-num", type=int, default=2)
    p.add_argument("--log-every", type=int, default=1)
    p.add_argument("--max-files", type=int, default=None)
    return p.parse_args(argv)


def main():
    global mg_data_process
    if not os.path.isdir(mg.REF):
        raise SystemExit(f"reference checkout not found at {mg.REF}")
    torch.set_num_threads(1)
    discriminator, train_mod = import_reference_critic()
    AIGGenerator, _, mg_data_process, _ = mg.import_reference()
    print("critic cases")
    critic_cases(discriminator, train_mod, mg_data_process)
    print("training runs")
    graphs_a = [synth_aig(5 + i, 26 + 7 * i, 3 + (i % 2), seed=300 + i, window=10 + 4 * i, orientation="dataset")
                for i in range(4)]
    train_run("bs1", train_mod, AIGGenerator, discriminator, graphs_a,
              ["--epochs", "1", "--batch-size", "1", "--candidate-k", "16", "--save-every", "2"], seed=2024)
    graphs_b = [synth_aig(6 + i, 30 + 5 * i, 4, seed=320 + i, window=12, orientation="dataset") for i in range(3)]
    train_run("bs3", train_mod, AIGGenerator, discriminator, graphs_b,
              ["--epochs", "2", "--batch-size", "3", "--candidate-k", "8", "--save-every", "10", "--lr", "0.001"], seed=7)


if __name__ == "__main__":
    main()
