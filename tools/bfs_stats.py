"""Level structure of the repaired fake graph of the bench workload (tuning aid, not a test)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.synth import synth_aig
from gan_amd.generator import AIGGenerator
from gan_amd.data import Data

dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
raw = G(Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev)), torch.randn(128, device=dev))
ei = raw.edge_index.cpu().numpy(); x = raw.x.cpu().numpy(); n = x.shape[0]
types = x[:, 0].astype(int)
def levels(src, dst, starts, kahn=False):
    order = np.argsort(src, kind="stable"); s, d = src[order], dst[order]
    ptr = np.searchsorted(s, np.arange(n + 1))
    deg = np.diff(ptr)
    out = []
    if not kahn:
        seen = np.zeros(n, bool); seen[starts] = True; fr = starts
        while fr.size:
            dg = deg[fr]
            nb = np.concatenate([d[ptr[u]:ptr[u + 1]] for u in fr]) if fr.size else np.array([], int)
            out.append((fr.size, int(dg.sum()), int(dg.max()), int((dg > 2).sum()), int((dg > 32).sum())))
            nb = np.unique(nb); nb = nb[~seen[nb]]; seen[nb] = True; fr = nb
    else:
        ind = np.bincount(dst, minlength=n); fr = np.nonzero(ind == 0)[0]
        while fr.size:
            dg = deg[fr]
            nb = np.concatenate([d[ptr[u]:ptr[u + 1]] for u in fr]) if fr.size else np.array([], int)
            out.append((fr.size, int(dg.sum()), int(dg.max()), int((dg > 2).sum()), int((dg > 32).sum())))
            np.subtract.at(ind, nb, 1); nb = np.unique(nb); fr = nb[ind[nb] == 0]
    return np.array(out)
for name, (s, d, st, k) in {"fwd from PI": (ei[0], ei[1], np.nonzero(types == 0)[0], False),
                            "rev from PO": (ei[1], ei[0], np.nonzero(types == 1)[0], False),
                            "kahn": (ei[0], ei[1], None, True)}.items():
    L = levels(s, d, st, k)
    print(name, "levels", len(L), "frontier mean/max", L[:, 0].mean().round(1), L[:, 0].max(), "edges/level mean/max",
          L[:, 1].mean().round(1), L[:, 1].max(), "maxdeg mean/max", L[:, 2].mean().round(1), L[:, 2].max(),
          "nodes deg>2 per level", L[:, 3].mean().round(1), "deg>32 per level", L[:, 4].mean().round(1))
    print("   sum over levels of max-degree:", L[:, 2].sum(), " levels with maxdeg>32:", int((L[:, 2] > 32).sum()))
