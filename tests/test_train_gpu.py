"""One full training step on the GPU against the CPU checker's step (same weights, same random
draws): losses and scores within 1e-4 relative, critic weights after the step within 1e-4,
generated graphs identical."""
import argparse

import numpy as np
import pytest
import torch

from gan_amd.data import Batch, Data
from gan_amd.dataset import pad_data, unpad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser, gradient_penalty, sanitize_for_D
from oracle import model as omodel
from oracle.train_oracle import CpuTrainer
from util_golden import gen_state

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-6)


@pytest.mark.parametrize("gp", ["closed_form", "autograd"])
@pytest.mark.parametrize("capture", [False, True], ids=["eager", "hip_graph"])
@pytest.mark.parametrize("batch", [1, 3])
def test_training_step_matches_cpu_checker(batch, capture, gp):
    graphs = [synth_aig(6 + 2 * i, 50 + 20 * i, 4 + i, seed=40 + i, window=24, orientation="dataset")
              for i in range(batch)]
    max_nodes = max(g.x.size(0) for g in graphs)
    max_edges = max(g.edge_index.size(1) for g in graphs)
    padded = [pad_data(g, max_nodes, max_edges) for g in graphs]
    args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "16", "--save-every", "1"] +
                                     (["--autograd-gp"] if gp == "autograd" else []))
    tr = Trainer(args, max_nodes, max_edges, DEV)
    gs = gen_state()
    dparams = omodel.init_discriminator_params(11)
    tr.G.load_state_dict(gs)
    tr.D.load_state_dict(dparams)
    g = torch.Generator().manual_seed(5)
    zs = [torch.randn(128, generator=g) for _ in range(2 * batch)]
    alphas = [torch.rand(1, 1, generator=g) for _ in range(args.n_critic)]
    zi, ai = iter(zs), iter(alphas)
    tr.draw_z = lambda: next(zi).to(DEV)
    tr.draw_alpha = lambda: next(ai).to(DEV)
    real = Batch.from_data_list(padded)
    real.sample_ids = list(range(batch))
    step_batch = Batch.from_data_list(padded).to(DEV)
    if capture:                      # minibatches that carry sample ids replay the recorded critic terms
        step_batch.sample_ids = list(range(batch))
    out = tr.step(step_batch)
    assert (len(tr._captured) == 1) == capture
    got = {k: float(v) for k, v in out.items()}

    cpu_tr = CpuTrainer(gs, dparams, 16, max_nodes, max_edges)
    templates = [unpad_data(d) for d in real.to_data_list()]           # stay padded, as in the reference
    ref = cpu_tr.step(real, templates, zs[:batch], zs[batch:], alphas, validate=True)
    for k in ("D_real", "D_fake", "gap", "gp", "grad_norm", "loss_D", "loss_G", "pen"):
        assert rel(got[k], ref[k]) <= 1e-4, (k, got[k], ref[k])
    assert tr.edges_last_step == ref["edges"]
    # three Adam steps later the critic weights still agree
    for k, v in tr.D.state_dict().items():
        a, b = v.cpu().numpy(), cpu_tr.dp[k].detach().numpy()
        assert np.abs(a - b).max() <= 1e-4 * max(np.abs(b).max(), 1e-3), k
    # the generator did not move (its output carries no gradient)
    for k, v in tr.G.state_dict().items():
        assert torch.equal(v.cpu(), gs[k]), k
    assert all(p.grad is None for p in tr.G.parameters())


def test_gradient_penalty_rejects_edge_branch():
    with pytest.raises(NotImplementedError):
        gradient_penalty(None, None, None, DEV, use_edges=True)


def test_sanitize_drops_padding_edges():
    g = pad_data(synth_aig(5, 6, 2, seed=1, window=8), 20, 40).to(DEV)
    s = sanitize_for_D(g)
    assert s.edge_index.size(1) == 14 and int(s.edge_index.min()) >= 0 and s.edge_attr.size(0) == 14
    assert s.x is g.x and s.node_depth is g.node_depth
