"""Small-M GEMMs of the transformer decoder (50 queries x 20 views = 1000 rows) on xm3d_gemm_bf16 vs the library: HIP events over
back-to-back launches and over a HIP-graph replay of the same launches.  python tools/gemm_small_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from xmask3d_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
REPS = 50


def ev(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


for M, K, N, act in [(1000, 256, 256, None), (1000, 256, 2048, "relu"), (1000, 2048, 256, None), (1000, 256, 512, None), (1540, 768, 2560, None),
                     (6140, 1024, 1024, None), (107520, 256, 256, None), (20480, 256, 256, None), (81920, 320, 320, None)]:
    x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.zeros(N, device=dev)
    packed, tile = ops.gemm_pack_weight(w)

    def own():
        for _ in range(REPS):
            ops.gemm(x, packed, N, tile, bias=b, act=act)

    def lib():
        for _ in range(REPS):
            y = F.linear(x, w)
            if act:
                torch.relu_(y)

    res = []
    for fn in (own, lib):
        t_stream = ev(fn) / REPS
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                fn()
        t_graph = ev(gr.replay) / REPS
        res.append((t_stream, t_graph))
    print(f"M{M} K{K} N{N} {act or '-':5s}: own {res[0][0]:7.1f} us stream / {res[0][1]:7.1f} us graph   library {res[1][0]:7.1f} / {res[1][1]:7.1f}", flush=True)
