"""A few launches of k_gemm on ONE shape, for rocprofv3 --pmc runs (tools/pmc_summary.py reads the counters).
python tools/gemm_pmc.py [M K N [act]]   default: mask-CLIP c_fc, 20 views (5140, 1024, 4096, quick_gelu)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xmask3d_amd import ops

M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5140, 1024, 4096)
act = sys.argv[4] if len(sys.argv) > 4 else ("quick_gelu" if len(sys.argv) <= 3 else None)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
bias = torch.randn(N, generator=g).to(dev)
packed, tile = ops.gemm_pack_weight(w, act)
for _ in range(6):
    y = ops.gemm(x, packed, N, tile, bias=bias, act=act)
torch.cuda.synchronize()
print("done", tuple(y.shape), tile)
