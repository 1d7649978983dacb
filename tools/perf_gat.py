"""Timing aid (not a test): forward/backward GAT kernels on the bench graphs, median of many launches."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _abi
from gan_amd.synth import synth_aig
from gan_amd.discriminator import GraphCSR, AIGDiscriminator
from gan_amd.generator import AIGGenerator
from gan_amd.constraints import enforce_aig_constraints
from gan_amd.data import Data

dev = "cuda"
torch.manual_seed(123)
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
G = AIGGenerator(candidate_k=64).to(dev)
fake = enforce_aig_constraints(G(Data(x=g.x.to(dev), node_depth=g.node_depth.to(dev)), torch.randn(128, device=dev)))
D = AIGDiscriminator().to(dev)
graphs = {"real": (g.edge_index.to(dev), g.edge_attr.to(dev)), "fake": (fake.edge_index, fake.edge_attr)}
n = g.x.size(0)
only = sys.argv[1] if len(sys.argv) > 1 else None      # "real" / "fake": one graph kind (for per-kernel profiles)
for name, (ei, ea) in graphs.items():
    if only and name != only:
        continue
    csr = GraphCSR(ei, ea, n)
    x = torch.randn(n, 128, device=dev, requires_grad=True)
    for _ in range(3):
        h = D.gat2(x, csr); h.sum().backward()
    _abi.timeline.reset(); _abi.timeline.enabled = True
    for _ in range(30):
        h = D.gat2(x, csr); h.sum().backward()
    _abi.timeline.enabled = False
    torch.cuda.synchronize()
    for k, v in sorted(_abi.timeline.events.items()):
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in v)
        print(f"{name:5s} {k:20s} median {statistics.median(t):8.1f} us  min {t[0]:8.1f} us  (E={ei.size(1)})")
