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
