"""One full training step on the GPU against the CPU checker's step (same weights, same random
draws): losses and scores within 1e-4 relative, critic weights after the step within 1e-4,
generated graphs identical."""
import argparse

import numpy as np
import pytest
import torch

from gan_amd.graphset import GraphSet
from gan_amd.data import Batch, Data
from gan_amd.dataset import pad_data, unpad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser, gradient_penalty, sanitize_for_D
from oracle import model as omodel
from oracle.train_oracle import CpuTrainer
from util_golden import TrainRun, gen_state

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-6)


@pytest.fixture(autouse=True, scope="module")
def _one_cpu_thread():
    """The CPU checker's sums run in one thread here, so that they come out the same every time: with several
    threads its last bits change from run to run, and Adam turns a last-bit difference in a near-zero gradient
    element into a full step, which after a dozen steps can reach the 1e-4 these tests hold the scalars to."""
    before = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(before)


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
                                     (["--autograd-gp"] if gp == "autograd" else []) +
                                     ([] if capture else ["--no-graph-capture"]))
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
    step_batch.sample_ids = list(range(batch))
    out = tr.step(step_batch)
    assert (len(tr._captured) == 1) == capture and tr.captures == int(capture)
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


def test_training_step_with_another_z_width_matches_cpu_checker():
    """``--z-dim 48`` (train.py:374): the step of the default recording path against the CPU checker."""
    batch, zd = 2, 48
    graphs = [synth_aig(6 + 2 * i, 60 + 25 * i, 4 + i, seed=90 + i, window=24, orientation="dataset") for i in range(batch)]
    max_nodes = max(g.x.size(0) for g in graphs)
    max_edges = max(g.edge_index.size(1) for g in graphs)
    padded = [pad_data(g, max_nodes, max_edges) for g in graphs]
    args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "16", "--save-every", "1", "--z-dim", str(zd)])
    tr = Trainer(args, max_nodes, max_edges, DEV)
    torch.manual_seed(17)
    from gan_amd.generator import AIGGenerator
    gs = {k: v.detach().clone() for k, v in AIGGenerator(z_dim=zd, candidate_k=16).state_dict().items()}
    assert tuple(gs["node_proj.0.weight"].shape) == (128, 128 + zd)
    dparams = omodel.init_discriminator_params(13)
    tr.G.load_state_dict(gs)
    tr.D.load_state_dict(dparams)
    g = torch.Generator().manual_seed(6)
    zs = [torch.randn(zd, generator=g) for _ in range(2 * batch)]
    alphas = [torch.rand(1, 1, generator=g) for _ in range(args.n_critic)]
    zi, ai = iter(zs), iter(alphas)
    tr.draw_z = lambda: next(zi).to(DEV)
    tr.draw_alpha = lambda: next(ai).to(DEV)
    real = Batch.from_data_list(padded)
    real.sample_ids = list(range(batch))
    step_batch = Batch.from_data_list(padded).to(DEV)
    step_batch.sample_ids = list(range(batch))
    got = {k: float(v) for k, v in tr.step(step_batch).items()}
    cpu_tr = CpuTrainer(gs, dparams, 16, max_nodes, max_edges)
    templates = [unpad_data(d) for d in real.to_data_list()]
    ref = cpu_tr.step(real, templates, zs[:batch], zs[batch:], alphas, validate=True)
    for k in ("D_real", "D_fake", "gap", "gp", "grad_norm", "loss_D", "loss_G", "pen"):
        assert rel(got[k], ref[k]) <= 1e-4, (k, got[k], ref[k])
    assert tr.edges_last_step == ref["edges"]


@pytest.mark.parametrize("gp", ["closed_form", "autograd"])
@pytest.mark.parametrize("batch", [1, 2])
def test_rotating_minibatches_share_one_recording(batch, gp):
    """Six different minibatches of one padded shape, stepped through twice in a different order (what a
    shuffling loader feeds train.py:229-233): the critic terms are recorded once for the shape and every step
    agrees with the CPU checker run over the same sequence (its Adam state included)."""
    n_batches = 6
    graphs = [synth_aig(5 + (i % 4), 40 + 13 * i, 3 + (i % 3), seed=70 + i, window=8 + 6 * (i % 5), orientation="dataset")
              for i in range(n_batches * batch)]
    max_nodes = max(g.x.size(0) for g in graphs)
    max_edges = max(g.edge_index.size(1) for g in graphs)
    padded = [pad_data(g, max_nodes, max_edges) for g in graphs]
    assert len({int(p.node_depth.max()) for p in padded}) > 2          # the level counts differ between minibatches
    args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "16", "--save-every", "1"] +
                                     (["--autograd-gp"] if gp == "autograd" else []))
    args.level_cap = max(int(p.node_depth.max()) for p in padded)
    tr = Trainer(args, max_nodes, max_edges, DEV)
    gs, dparams = gen_state(), omodel.init_discriminator_params(13)
    tr.G.load_state_dict(gs)
    tr.D.load_state_dict(dparams)
    cpu_tr = CpuTrainer(gs, dparams, 16, max_nodes, max_edges)
    g = torch.Generator().manual_seed(9)
    order = list(range(n_batches)) + [4, 2, 5, 0, 3, 1]
    for step, b in enumerate(order):
        members = list(range(b * batch, (b + 1) * batch))
        zs = [torch.randn(128, generator=g) for _ in range(2 * batch)]
        alphas = [torch.rand(1, 1, generator=g) for _ in range(args.n_critic)]
        zi, ai = iter(zs), iter(alphas)
        tr.draw_z = lambda: next(zi).to(DEV)
        tr.draw_alpha = lambda: next(ai).to(DEV)
        real = Batch.from_data_list([padded[i] for i in members])
        step_batch = Batch.from_data_list([padded[i] for i in members]).to(DEV)
        step_batch.sample_ids = members
        got = {k: float(v) for k, v in tr.step(step_batch).items()}
        templates = [unpad_data(d) for d in real.to_data_list()]
        ref = cpu_tr.step(real, templates, zs[:batch], zs[batch:], alphas, validate=True)
        for k in ("D_real", "D_fake", "gap", "gp", "grad_norm", "loss_D", "loss_G", "pen"):
            assert rel(got[k], ref[k]) <= 1e-4, (step, k, got[k], ref[k])
        assert tr.edges_last_step == ref["edges"], step
        assert tr.captures == 1 and len(tr._captured) == 1, step
        assert not tr._captured[(max_nodes * batch, max_edges * batch)].plan.overflowed()
    # 36 Adam steps later: Adam divides every gradient element by its own running magnitude, so an element that
    # is rounding noise still moves its weight by up to lr per step - weights are held to a tenth of that
    opt_steps = len(order) * args.n_critic
    for k, v in tr.D.state_dict().items():
        a, b_ = v.cpu().numpy(), cpu_tr.dp[k].detach().numpy()
        assert np.abs(a - b_).max() <= 1e-4 * max(np.abs(b_).max(), 1e-3) + 0.1 * args.lr * opt_steps, k


def test_recording_outgrown_by_a_deeper_minibatch_is_replaced():
    """A minibatch whose deepest level exceeds what the recording was sized for gets a new recording (and
    correct numbers), never a silently truncated readout."""
    shallow = synth_aig(8, 220, 4, seed=3, window=120, orientation="dataset")
    deep = synth_aig(8, 220, 4, seed=4, window=2, orientation="dataset")
    assert int(deep.node_depth.max()) > 64 + 2 * int(shallow.node_depth.max())
    mn, me = max(shallow.x.size(0), deep.x.size(0)), max(shallow.edge_index.size(1), deep.edge_index.size(1))
    args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "16", "--save-every", "1000"])
    tr = Trainer(args, mn, me, DEV)
    tr.global_step = 1
    gs, dparams = gen_state(), omodel.init_discriminator_params(3)
    tr.G.load_state_dict(gs)
    tr.D.load_state_dict(dparams)
    cpu_tr = CpuTrainer(gs, dparams, 16, mn, me)
    g = torch.Generator().manual_seed(2)
    for step, graph in enumerate([shallow, deep, shallow]):
        zs = [torch.randn(128, generator=g) for _ in range(2)]
        alphas = [torch.rand(1, 1, generator=g) for _ in range(3)]
        zi, ai = iter(zs), iter(alphas)
        tr.draw_z = lambda: next(zi).to(DEV)
        tr.draw_alpha = lambda: next(ai).to(DEV)
        real = Batch.from_data_list([pad_data(graph, mn, me)])
        got = {k: float(v) for k, v in tr.step(Batch.from_data_list([pad_data(graph, mn, me)]).to(DEV)).items()}
        ref = cpu_tr.step(real, [unpad_data(d) for d in real.to_data_list()], zs[:1], zs[1:], alphas, validate=False)
        for k in ("D_real", "D_fake", "gp", "loss_D", "loss_G"):
            assert rel(got[k], ref[k]) <= 1e-4, (step, k, got[k], ref[k])
        assert tr.captures == (1 if step == 0 else 2)


def test_generated_side_deeper_than_its_recording_is_scored_eagerly_then_recorded_anew():
    """A generated minibatch with more levels than its recording was sized for (forced here by shrinking the
    recording's level capacity) is scored eagerly in that step - correct numbers - and the next step records
    the generated side anew with room for it."""
    graph = synth_aig(8, 220, 4, seed=3, window=120, orientation="forward")
    mn, me = graph.x.size(0), graph.edge_index.size(1)
    args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "2", "--save-every", "1000",
                                      "--capture-fake", "on"])
    tr = Trainer(args, mn, me, DEV)
    tr.global_step = 1
    gs, dparams = gen_state(), omodel.init_discriminator_params(3)
    tr.G.load_state_dict(gs)
    tr.D.load_state_dict(dparams)
    cpu_tr = CpuTrainer(gs, dparams, 2, mn, me)
    g = torch.Generator().manual_seed(2)
    key = (mn, me)
    for step in range(4):
        zs = [torch.randn(128, generator=g) for _ in range(2)]
        alphas = [torch.rand(1, 1, generator=g) for _ in range(3)]
        zi, ai = iter(zs), iter(alphas)
        tr.draw_z = lambda: next(zi).to(DEV)
        tr.draw_alpha = lambda: next(ai).to(DEV)
        if step == 1:
            tr._captured[key].plan_f.L = 5            # the fakes of this template are about 15 levels deep
        real = Batch.from_data_list([pad_data(graph, mn, me)])
        got = {k: float(v) for k, v in tr.step(Batch.from_data_list([pad_data(graph, mn, me)]).to(DEV)).items()}
        ref = cpu_tr.step(real, [unpad_data(d) for d in real.to_data_list()], zs[:1], zs[1:], alphas, validate=False)
        for k in ("D_real", "D_fake", "gp", "loss_D", "loss_G"):
            assert rel(got[k], ref[k]) <= 1e-4, (step, k, got[k], ref[k])
        if step == 0:
            assert tr.captures == 1 and tr._captured[key].fake_side and not tr._fake_caps
        elif step == 1:
            assert tr.captures == 1 and key not in tr._captured and tr._fake_caps[key] > 10
        else:
            assert tr.captures == 2 and tr._captured[key].fake_cap >= tr._fake_caps[key]
            assert not tr._captured[key].plan_f.overflowed()
    for k, v in tr.D.state_dict().items():
        a, b_ = v.cpu().numpy(), cpu_tr.dp[k].detach().numpy()
        assert np.abs(a - b_).max() <= 1e-4 * max(np.abs(b_).max(), 1e-3) + 0.1 * args.lr * 12, k


@pytest.mark.parametrize("name", ["bs1", "bs3"])
@pytest.mark.parametrize("mode", ["hip_graph", "hip_graph_eager_fakes", "eager", "autograd_gp", "skip_unused",
                                  "eager_skip_unused", "separate_update", "gp_all_nodes"])
def test_training_follows_reference_run(name, mode):
    """A whole ``train(args)`` run recorded from the reference's own train.py (tests/golden/make_golden_critic.py;
    its attention layer is the restated one): the same dataset, initial critic, draws and raw generator outputs
    fed to ``Trainer.step`` give the same metrics.csv rows within 1e-4 and the same critic at each epoch's end."""
    run = TrainRun(name)
    mn = max(g.x.size(0) for g in run.graphs)
    me = max(g.edge_index.size(1) for g in run.graphs)
    padded = [pad_data(g, mn, me) for g in run.graphs]
    extra = {"hip_graph": [], "hip_graph_eager_fakes": ["--capture-fake", "off"], "eager": ["--no-graph-capture"],
             "autograd_gp": ["--autograd-gp"],
             # the generator step without its backward pass (it only reaches critic gradients that are zeroed
             # before their next use, train.py:253,298): the run's numbers must not notice
             "skip_unused": ["--skip-unused-backward"],
             "eager_skip_unused": ["--no-graph-capture", "--skip-unused-backward"],
             # zero_grad, the gradient sums and torch's own Adam as separate launches instead of aig_adam_sum3: the
             # two forms differ in last bits, both follow the recorded run to the same bound
             "separate_update": ["--no-fused-update"],
             # the recorded penalty term over every node instead of over its distinct rows (the default)
             "gp_all_nodes": ["--gp-all-nodes"]}[mode]
    args = build_parser().parse_args(["--dataset-dir", "unused"] + run.argv + extra)
    tr = Trainer(args, mn, me, DEV)
    tr.G.load_state_dict(gen_state())
    tr.D.load_state_dict(run.d_init)
    zi, ai = iter(run.z), iter(run.alpha)
    tr.draw_z = lambda: next(zi).to(DEV)
    tr.draw_alpha = lambda: next(ai).to(DEV)
    feed = []

    class RecordedGenerator:
        """The recorded run's raw graphs stand in for the generator pass (graph by graph, or as the set of a
        minibatch the way the trainer asks for them)."""

        def __call__(self, template, z):
            return feed.pop(0)

        def forward_set(self, templates, zs):
            return GraphSet.from_data_list([feed.pop(0) for _ in templates])

    tr.G = RecordedGenerator()
    last_epoch = 0

    def check_epoch(e):
        ref = run.epoch_state(e)
        for k, v in tr.D.state_dict().items():
            diff = float((v.cpu() - ref[k]).abs().max())
            assert diff <= 1e-4 * float(ref[k].abs().max()) + 0.1 * args.lr, (e, k, diff)

    for step, (e, members, call, a, validate) in enumerate(run.steps()):
        if e != last_epoch:
            check_epoch(last_epoch)
            last_epoch = e
        assert validate == (tr.global_step % args.save_every == 0)
        b = Batch.from_data_list([padded[i] for i in members]).to(DEV)
        b.sample_ids = members
        k = len(members)
        depth = [padded[i].node_depth for i in members]
        feed[:] = [run.raw(call + j, depth[j % k]).to(DEV) for j in range(2 * k)]
        got = {key: float(v) for key, v in tr.step(b).items()}
        assert not feed
        ref = run.metrics[step]
        for key in ("D_real", "D_fake", "gap", "grad_norm", "loss_D", "loss_G", "gp", "pen"):
            assert abs(got[key] - ref[key]) <= 1e-4 * max(abs(ref[key]), 1e-3), (step, key, got[key], ref[key])
    check_epoch(last_epoch)


def test_sanitize_drops_padding_edges():
    g = pad_data(synth_aig(5, 6, 2, seed=1, window=8), 20, 40).to(DEV)
    s = sanitize_for_D(g)
    assert s.edge_index.size(1) == 14 and int(s.edge_index.min()) >= 0 and s.edge_attr.size(0) == 14
    assert s.x is g.x and s.node_depth is g.node_depth


def test_one_launch_update_matches_separate_launches():
    """The one-launch form of a critic iteration's update (gradient sums + Adam, aig_adam_sum3) against the separate
    launches it replaces (zero_grad, three multi-tensor adds, torch's fused Adam): the same steps from the same
    weights and draws give the same critic, optimizer state and metrics."""
    graphs = [synth_aig(6 + i, 50 + 9 * i, 3, seed=90 + i, window=10, orientation="dataset") for i in range(4)]
    mn = max(g.x.size(0) for g in graphs)
    me = max(g.edge_index.size(1) for g in graphs)
    padded = [pad_data(g, mn, me) for g in graphs]
    results = []
    for extra in ([], ["--no-fused-update"]):
        args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "16", "--save-every", "1000"] + extra)
        args.level_cap = max(int(p.node_depth.max()) for p in padded)
        tr = Trainer(args, mn, me, DEV)
        tr.global_step = 1
        tr.G.load_state_dict(gen_state())
        tr.D.load_state_dict(omodel.init_discriminator_params(21))
        g = torch.Generator().manual_seed(3)
        draws_z = [torch.randn(128, generator=g) for _ in range(64)]
        draws_a = [torch.rand(1, 1, generator=g) for _ in range(64)]
        zi, ai = iter(draws_z), iter(draws_a)
        tr.draw_z = lambda: next(zi).to(DEV)
        tr.draw_alpha = lambda: next(ai).to(DEV)
        rows = []
        for step in range(6):
            b = Batch.from_data_list([padded[(2 * step) % 4], padded[(2 * step + 1) % 4]]).to(DEV)
            b.sample_ids = [(2 * step) % 4, (2 * step + 1) % 4]
            rows.append({k: float(v) for k, v in tr.step(b).items()})
        used = [t.__dict__.get("_fused_update") is not None for t in tr._captured.values()]
        state = {k: v.detach().cpu().clone() for k, v in tr.D.state_dict().items()}
        moments = [(tr.opt_D.state[p]["exp_avg"].cpu().clone(), tr.opt_D.state[p]["exp_avg_sq"].cpu().clone(),
                    float(tr.opt_D.state[p]["step"])) for p in tr._d_params]
        grads = [p.grad.detach().cpu().clone() for p in tr._d_params]
        results.append((rows, state, moments, grads, used, tr._d_updates))
    (r1, s1, m1, g1, used1, n1), (r2, s2, m2, g2, used2, n2) = results
    assert any(used1) and not any(used2)
    assert n1 == n2 == 18 and all(m[2] == 18.0 for m in m1) and all(m[2] == 18.0 for m in m2)
    for a_, b_ in zip(r1, r2):
        for k in a_:
            assert rel(a_[k], b_[k]) <= 1e-5, (k, a_[k], b_[k])
    for k in s1:
        assert float((s1[k] - s2[k]).abs().max()) <= 1e-6 + 1e-5 * float(s2[k].abs().max()), k
    for (ea1, es1, _), (ea2, es2, _) in zip(m1, m2):
        assert float((ea1 - ea2).abs().max()) <= 1e-6 + 1e-4 * float(ea2.abs().max())
        assert float((es1 - es2).abs().max()) <= 1e-9 + 1e-4 * float(es2.abs().max())


# ---------------------------------------------------------------------------------------------------------------
# The step as bench.py runs it (recorded critic terms over all three sides, one-launch update, two streams, level cap
# from the data), at the sizes bench.py runs it on.  Only there do the position sweeps (from 16,384 nodes), the
# hub-chunk backward pass, multi-tile sorts and real overlap between the two streams happen inside a step.
def _bench_trainer(max_nodes, max_edges, level_cap, seed_d, capture_fake="auto"):
    args = build_parser().parse_args(["--dataset-dir", "synthetic", "--candidate-k", "64", "--save-every", "1000000000",
                                      "--capture-fake", capture_fake])
    args.level_cap = level_cap
    tr = Trainer(args, max_nodes, max_edges, DEV)
    tr.G.load_state_dict(gen_state())
    tr.D.load_state_dict(omodel.init_discriminator_params(seed_d))
    return tr, args


def _bench_minibatches(groups):
    mn = max(g.x.size(0) for gs in groups for g in gs)
    me = max(g.edge_index.size(1) for gs in groups for g in gs)
    top = max(int(g.node_depth.max()) for gs in groups for g in gs)
    host, dev, next_id = [], [], 0
    for gs in groups:
        padded = [pad_data(g, mn, me) for g in gs]
        host.append(Batch.from_data_list(padded))
        b = Batch.from_data_list(padded).to(DEV)
        b.sample_ids = list(range(next_id, next_id + len(gs)))
        next_id += len(gs)
        dev.append(b)
    return host, dev, mn, me, top


def _steps_against_cpu(groups, steps, seed_d):
    host, dev, mn, me, top = _bench_minibatches(groups)
    tr, args = _bench_trainer(mn, me, top, seed_d)
    cpu_tr = CpuTrainer(gen_state(), omodel.init_discriminator_params(seed_d), 64, mn, me, segmented_readout=True)
    g = torch.Generator().manual_seed(17)
    for step in range(steps):
        k = len(groups[step % len(groups)])
        zs = [torch.randn(128, generator=g) for _ in range(2 * k)]
        alphas = [torch.rand(1, 1, generator=g) for _ in range(args.n_critic)]
        zi, ai = iter(zs), iter(alphas)
        tr.draw_z = lambda: next(zi).to(DEV)
        tr.draw_alpha = lambda: next(ai).to(DEV)
        got = {key: float(v) for key, v in tr.step(dev[step % len(dev)]).items()}
        real = host[step % len(host)]
        ref = cpu_tr.step(real, [unpad_data(d) for d in real.to_data_list()], zs[:k], zs[k:], alphas,
                          validate=(step == 0))          # (step 0 is a save step: the validator's verdict scales the penalty)
        for key in ("D_real", "D_fake", "gap", "gp", "grad_norm", "loss_D", "loss_G", "pen"):
            assert rel(got[key], ref[key]) <= 1e-4, (step, key, got[key], ref[key])
        assert tr.edges_last_step == ref["edges"], step
    # (one recording per minibatch shape; a second one when the repaired fakes turned out deeper than the data's
    #  deepest level, which sizes the first)
    assert tr.captures <= 2 and all(t.fk is not None and t.rw is not None for t in tr._captured.values())
    assert any(t.__dict__.get("_fused_update") is not None for t in tr._captured.values())       # the one-launch update ran
    opt_steps = steps * args.n_critic
    for key, v in tr.D.state_dict().items():
        a, b_ = v.cpu().numpy(), cpu_tr.dp[key].detach().numpy()
        assert np.abs(a - b_).max() <= 1e-4 * max(np.abs(b_).max(), 1e-3) + 0.1 * args.lr * opt_steps, key


@pytest.mark.timeout(900)
def test_bench_step_on_100k_node_minibatches_matches_cpu_checker():
    """BASELINE.json configs[2]: two steps over two different 100k-node minibatches (train.py:229-299)."""
    groups = [[synth_aig(1000, 98000, 1000, seed=s, window=4096, orientation="dataset")] for s in (0, 1)]
    _steps_against_cpu(groups, steps=2, seed_d=11)


@pytest.mark.timeout(900)
def test_bench_step_on_the_iscas85_sized_minibatch_matches_cpu_checker():
    """BASELINE.json configs[1]: the eleven ISCAS85-sized graphs as one minibatch padded to 2,401 nodes, two steps."""
    from gan_amd.synth import iscas85_like
    _steps_against_cpu([iscas85_like(seed=0), iscas85_like(seed=1)], steps=2, seed_d=12)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("capture_fake", ["auto", "off"])
def test_bench_step_is_bitwise_reproducible_across_trainers(capture_fake):
    """(``off``: the pass over the generated minibatch runs eagerly on the first stream, as in the second timed loop of
    bench.py - it reads what the second stream pads.)  Two fresh trainers, same weights and draws, four steps over three rotating 100k-node minibatches: critic,
    Adam moments and every logged scalar bit-identical.  No kernel of the library adds floats atomically, so a
    difference between two runs can only come from a missing wait between the step's two streams (a kernel reading
    what another stream has not finished writing)."""
    groups = [[synth_aig(1000, 98000, 1000, seed=s, window=4096, orientation="dataset")] for s in (3, 4, 5)]
    _, dev, mn, me, top = _bench_minibatches(groups)
    runs = []
    for _ in range(2):
        tr, args = _bench_trainer(mn, me, top, seed_d=5, capture_fake=capture_fake)
        g = torch.Generator().manual_seed(23)
        rows = []
        for step in range(4):
            zs = [torch.randn(128, generator=g) for _ in range(2)]
            alphas = [torch.rand(1, 1, generator=g) for _ in range(args.n_critic)]
            zi, ai = iter(zs), iter(alphas)
            tr.draw_z = lambda: next(zi).to(DEV)
            tr.draw_alpha = lambda: next(ai).to(DEV)
            out = tr.step(dev[step % 3])
            rows.append({k: v.detach().cpu().clone() for k, v in out.items()})
            rows[-1]["edges"] = torch.tensor(tr.edges_last_step)
        torch.cuda.synchronize()
        state = {k: v.detach().cpu().clone() for k, v in tr.D.state_dict().items()}
        moments = [(tr.opt_D.state[p]["exp_avg"].cpu().clone(), tr.opt_D.state[p]["exp_avg_sq"].cpu().clone())
                   for p in tr._d_params]
        runs.append((rows, state, moments))
        del tr
    (r1, s1, m1), (r2, s2, m2) = runs
    for step, (a_, b_) in enumerate(zip(r1, r2)):
        for k in a_:
            assert torch.equal(a_[k], b_[k]), (step, k, a_[k], b_[k])
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    for (ea1, es1), (ea2, es2) in zip(m1, m2):
        assert torch.equal(ea1, ea2) and torch.equal(es1, es2)
