"""Helpers shared by the golden-vector tests."""
import os

import numpy as np
import torch

from conftest import GOLDEN


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def gen_state():
    w = load("gen_weights")
    return {k: torch.from_numpy(w[k]) for k in w.files}


def golden_targets(d):
    """Per scored target of a gen_* fixture: (node, candidates, reference logits, reference picks).

    Picks are positions in the candidate list, recovered from the reference's emitted edges
    (each node is a target once, its edges are adjacent and ordered [top1, top2])."""
    ei = d["out_edge_index"]
    types = d["x"][:, 0].astype(np.int64)
    by_dst = {}
    for s, t in zip(ei[0].tolist(), ei[1].tolist()):
        by_dst.setdefault(t, []).append(s)
    out = []
    offs = d["rec_off"]
    for t, v in enumerate(d["rec_tgt"].tolist()):
        cand = d["rec_cand"][offs[t]:offs[t + 1]]
        logit = d["rec_logit"][offs[t]:offs[t + 1]]
        want = 2 if types[v] == 2 else 1
        k_eff = min(want, len(cand))
        srcs = by_dst[v][:k_eff]
        picks = [int(np.nonzero(cand == s)[0][0]) for s in srcs]
        out.append((v, cand, logit, picks))
    return out


def tie_consistent(x, cand, logit):
    """True when candidates with equal feature rows received bit-equal reference logits.

    The reference's CPU BLAS rounds remainder rows differently, so this can fail; the
    reference's pick among such candidates is then not a function of the inputs."""
    rows = x[cand]
    _, inv = np.unique(rows, axis=0, return_inverse=True)
    bits = logit.view(np.int32)
    for g in np.unique(inv):
        if np.unique(bits[inv.reshape(-1) == g]).size > 1:
            return False
    return True


# ---------------------------------------------------------------------------------------------
# critic_cases.npz / train_<run>.npz: captured by tests/golden/make_golden_critic.py from the reference's own
# models/discriminator.py and train.py (the attention layer inside them is the restated one)
def critic_cases():
    d = load("critic_cases")
    params = {k[len("param/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("param/")}
    return d, params, [str(n) for n in d["names"]]


def critic_case_graph(d, name):
    from gan_amd.data import Data
    pre = name + "/"
    g = Data(x=torch.from_numpy(d[pre + "x"]), edge_index=torch.from_numpy(d[pre + "edge_index"]))
    if pre + "edge_attr" in d.files:
        g.edge_attr = torch.from_numpy(d[pre + "edge_attr"])
    if pre + "node_depth" in d.files:
        g.node_depth = torch.from_numpy(d[pre + "node_depth"])
    return g


class TrainRun:
    """One recorded ``train(args)`` run of the reference: dataset, initial critic, draws, raw graphs, metrics."""

    def __init__(self, name):
        from gan_amd.data import Data
        d = self.d = load("train_" + name)
        self.argv = [str(a) for a in d["argv"]]
        n = int(d["n_graphs"])
        self.graphs = [Data(x=torch.from_numpy(d[f"graph{i}/x"]), edge_index=torch.from_numpy(d[f"graph{i}/edge_index"]),
                            edge_attr=torch.from_numpy(d[f"graph{i}/edge_attr"]),
                            node_depth=torch.from_numpy(d[f"graph{i}/node_depth"])) for i in range(n)]
        self.d_init = {k[len("d_init/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("d_init/")}
        self.z = [torch.from_numpy(z) for z in d["z"]]
        self.alpha = [torch.tensor([[float(a)]]) for a in d["alpha"]]
        self.fields = [str(f) for f in d["metric_fields"]]
        self.metrics = [dict(zip(self.fields, row)) for row in d["metrics"]]

    def arg(self, flag, default, cast=int):
        return cast(self.argv[self.argv.index(flag) + 1]) if flag in self.argv else default

    def epoch_state(self, e):
        pre = f"d_epoch{e}/"
        return {k[len(pre):]: torch.from_numpy(self.d[k]) for k in self.d.files if k.startswith(pre)}

    def raw(self, call, node_depth):
        """Raw graph the reference's generator emitted at its ``call``-th forward (it hands the template's
        depths through, generator.py:211-217)."""
        from gan_amd.data import Data
        return Data(x=torch.from_numpy(self.d[f"raw{call}/x"].copy()),
                    edge_index=torch.from_numpy(self.d[f"raw{call}/edge_index"].copy()),
                    edge_attr=torch.from_numpy(self.d[f"raw{call}/edge_attr"].copy()), node_depth=node_depth.clone())

    def steps(self):
        """Per loop step: (epoch, member graph ids, first z/raw call index, first alpha index, validate?)."""
        bs, epochs = self.arg("--batch-size", 1), self.arg("--epochs", 10)
        n_critic, save_every = self.arg("--n-critic", 3), self.arg("--save-every", 10)
        out, call, a, step = [], 0, 0, 0
        for e in range(epochs):
            for b0 in range(0, len(self.graphs), bs):
                members = list(range(b0, min(len(self.graphs), b0 + bs)))
                out.append((e, members, call, a, step % save_every == 0))
                call += 2 * len(members)
                a += n_critic
                step += 1
        return out
