"""Tuning aid: raw hipMemsetAsync nodes in a recorded graph - do several of them survive replay?"""
import ctypes
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device("cuda")

def memset(t, value, nbytes, offset=0):
    rc = hip.hipMemsetAsync(ctypes.c_void_p(t.data_ptr() + offset), value, nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc

def trial(name, sizes, pair=False):
    bufs = [torch.full((max(s // 4, 4),), 7, dtype=torch.int32, device=dev) for s in sizes]
    def build():
        for b, s in zip(bufs, sizes):
            memset(b, 0, s)
            if pair:
                memset(b, 0xFF, 4)
            b.add_(1)
    s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        build()
    torch.cuda.current_stream().wait_stream(s0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    out = []
    for r in range(3):
        for b in bufs:
            b.fill_(7)
        g.replay(); torch.cuda.synchronize()
        out.append([(int(b[0]), int(b[1:].min()), int(b[1:].max())) for b in bufs])
    print(name, out, flush=True)

trial("one 16B", [16])
trial("one 16B + 4B 0xFF pair", [16], pair=True)
trial("one 64B pair", [64], pair=True)
trial("four 16B", [16] * 4)
trial("mixed", [16, 4096, 64, 1 << 20, 8, 256])
trial("mixed pairs", [16, 4096, 64, 1 << 20], pair=True)
