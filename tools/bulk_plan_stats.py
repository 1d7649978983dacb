"""Tuning aid: candidate counts of the bulk workload's template (which selection path a slot takes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from gan_amd.data import Data
from gan_amd.generator import AIGGenerator
from gan_amd.synth import ISCAS85_SHAPES, synth_aig
name, n_pi, n_po, n_and = ISCAS85_SHAPES[-1]
g = synth_aig(n_pi, n_and, n_po, seed=0, window=max(8, n_and // 8 + n_pi), orientation="dataset")
dev = torch.device("cuda")
tpl = Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev))
G = AIGGenerator(candidate_k=256).to(dev).eval()
plan = G.plan_for(tpl, tpl.x.float().contiguous(), tpl.node_depth)
cnt, want = plan.cnt.cpu().numpy(), plan.want.cpu().numpy()
k = np.minimum(cnt, want)
print("slots", len(cnt), "targets", int((k > 0).sum()), "partial_sort slots", int(((k > 0) & (k * 64 <= cnt)).sum()),
      "cnt quantiles", np.percentile(cnt[k > 0], [0, 10, 50, 90, 100]).tolist(), "pairs", int(cnt.sum()))
