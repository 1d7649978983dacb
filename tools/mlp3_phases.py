"""Timing aid (needs a library built with -DMLP3_TIMING): cycle stamps of workgroup 0 of the fused
node-MLP kernel, per tile: layer 1, barrier, layer 2, barrier, layer 3, stash; and the 100 MHz wall clock."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _abi
from gan_amd.generator import AIGGenerator

dev = "cuda"
n = 100000
G = AIGGenerator(candidate_k=64).to(dev)
lib = _abi.lib()
x = torch.rand(n, 2, device=dev)
z = torch.randn(128, device=dev)
proj = torch.empty(4, n, 128, device=dev)
hc = torch.empty(n, 128, device=dev)
ws = _abi.workspace(lib.aig_gen_node_proj_ws_bytes(n), dev)
w = G._weights()
for _ in range(3):
    _abi.check(lib.aig_gen_node_proj(n, _abi.ptr(x), _abi.ptr(z), C.byref(w), _abi.ptr(hc), _abi.ptr(proj), _abi.ptr(ws),
                                     ws.numel(), _abi.stream()), "np")
torch.cuda.synchronize()
dbg = ws[:8 * 8 * 8].view(torch.int64).cpu().view(-1, 8)[:7]
names = ["L1", "bar", "L2", "bar", "L3", "stash"]
for it, row in enumerate(dbg.tolist()):
    d = [row[i + 1] - row[i] for i in range(6)]
    tot = row[6] - row[0]
    print(f"tile {it}: " + "  ".join(f"{nm} {v:6d}" for nm, v in zip(names, d)) + f"  total {tot}  wall {row[7] * 10} ns")
