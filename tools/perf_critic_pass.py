"""Per-kernel times of one critic pass (forward + backward) on a 100k-node real graph and on a graph wired the way
an untrained generator does it (two hub sources), from the events the library's kernels carry (aig_kernel_timing).
Tuning aid: AIG_GAT_GRID_CAP and friends can be swept from the environment."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gan_amd import _abi
from gan_amd.data import Data
from gan_amd.discriminator import AIGDiscriminator
from gan_amd.synth import synth_aig

dev = torch.device("cuda")
lib = _abi.lib()
names = []
k = 0
while lib.aig_kernel_timing_name(k):
    names.append(lib.aig_kernel_timing_name(k).decode())
    k += 1


def graphs():
    real = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
    n = 100000
    rng = np.random.default_rng(8)
    dst = np.repeat(np.arange(1000, n), 2)
    src = np.where(rng.random(dst.size) < 0.85, rng.integers(0, 2, dst.size), rng.integers(0, 1000, dst.size))
    x = np.stack([rng.choice([0, 1, 2], n), rng.integers(0, 3, n)], 1).astype(np.float32)
    fake = Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.stack([src, dst])),
                edge_attr=torch.from_numpy(rng.integers(0, 2, src.size).astype(np.float32)).view(-1, 1),
                node_depth=torch.from_numpy(rng.integers(0, 300, n)))
    return {"real": real, "two_hubs": fake}


torch.manual_seed(0)
D = AIGDiscriminator().to(dev)
for name, d in graphs().items():
    dd = Data(x=d.x.to(dev), edge_index=d.edge_index.to(dev), edge_attr=d.edge_attr.to(dev), node_depth=d.node_depth.to(dev).long())
    for _ in range(3):
        D.zero_grad()
        D(dd).sum().backward()
    torch.cuda.synchronize()
    lib.aig_kernel_timing((1 << len(names)) - 1)
    for _ in range(20):
        D.zero_grad()
        D(dd).sum().backward()
    torch.cuda.synchronize()
    out = {}
    for k, nm in enumerate(names):
        ms, cnt = C.c_double(), C.c_int64()
        if lib.aig_kernel_timing_read(k, C.byref(ms), C.byref(cnt)) == 0 and cnt.value:
            out[nm] = round(ms.value * 1e3 / cnt.value, 1)
    lib.aig_kernel_timing(0)
    print(name, out, "sum", round(sum(out.values()), 1))
