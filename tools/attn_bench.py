"""HIP flash attention (xm3d_attention_fwd) vs the library path (torch SDPA -> AOTriton) on the shapes of the bench forward
(20 views), HIP events.  usage: python tools/attn_bench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from xmask3d_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


shapes = [("unet self 64^2", 20, 8, 4096, 4096, 40), ("unet self 32^2", 20, 8, 1024, 1024, 80), ("unet self 16^2", 20, 8, 256, 256, 160),
          ("unet cross 64^2", 20, 8, 4096, 77, 40), ("unet cross 32^2", 20, 8, 1024, 77, 80), ("clip vit-l", 20, 16, 307, 307, 64),
          ("decoder x-attn 64^2", 20, 8, 50, 4096, 32), ("decoder x-attn 16^2", 20, 8, 50, 256, 32)]
g = torch.Generator().manual_seed(0)
for name, B, H, Nq, Nk, D in shapes:
    q = torch.randn(B, Nq, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nq, H, D)
    k = torch.randn(B, Nk, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nk, H, D)
    v = torch.randn(B, Nk, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nk, H, D)
    t_hip = timeit(lambda: ops.attention(q, k, v))
    qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    t_lib = timeit(lambda: F.scaled_dot_product_attention(qt, kt, vt).transpose(1, 2).reshape(B, Nq, -1))
    flop = 4.0 * B * H * Nq * Nk * D
    print(f"{name:22s} B{B} H{H} Nq{Nq} Nk{Nk} D{D}: hip {t_hip:8.1f} us ({flop / t_hip / 1e6:6.1f} TF)   library {t_lib:8.1f} us ({flop / t_lib / 1e6:6.1f} TF)   x{t_lib / t_hip:4.2f}", flush=True)

# additive-bias forms of the forward: mask-CLIP (per-image mask shared by the heads) and the decoder's masked cross-attention
print("with additive bias (bf16 / f32 inputs):", flush=True)
for name, B, H, Nq, Nk, D in [("clip vit-l + mask", 20, 16, 307, 307, 64), ("decoder x-attn 64^2 + mask", 20, 8, 50, 4096, 32),
                              ("decoder x-attn 32^2 + mask", 20, 8, 50, 1024, 32)]:
    q = torch.randn(B, Nq, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nq, H, D)
    k = torch.randn(B, Nk, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nk, H, D)
    v = torch.randn(B, Nk, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nk, H, D)
    bias = torch.where(torch.rand(B, 1, Nq, Nk, generator=g) < 0.3, float("-inf"), 0.0)
    bias[..., 0] = 0.0
    bias = bias.to(dev)
    b16 = bias.to(torch.bfloat16)
    t16 = timeit(lambda: ops.attention(q, k, v, bias=b16))
    qf, kf, vf = q.float(), k.float(), v.float()
    t32 = timeit(lambda: ops.attention_f32(qf, kf, vf, bias=bias)) if D <= 64 else float("nan")
    flop = 4.0 * B * H * Nq * Nk * D
    print(f"{name:28s} B{B} H{H} Nq{Nq} Nk{Nk} D{D}: bf16 {t16:8.1f} us ({flop / t16 / 1e6:6.1f} TF)   f32-accurate {t32:8.1f} us", flush=True)
