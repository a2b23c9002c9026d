"""Fused GroupNorm-SiLU-conv3x3 kernel (csrc/conv.hip) against the chain it replaces (HIP GroupNorm apply + MIOpen/CK NHWC
convolution + bias/residual pass) on the ResnetBlock shapes of the SD VAE / UNet at `views` views.
usage: python tools/conv_bench.py [views=20] [iters=10]"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import xmask3d_amd  # noqa: F401  (MIOpen env)
from xmask3d_amd import ops

views = int(sys.argv[1]) if len(sys.argv) > 1 else 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda:0")
SHAPES = [("vae 128@512", 128, 128, 512, 512), ("vae 256@256", 256, 256, 256, 256), ("vae 512@128", 512, 512, 128, 128),
          ("vae 512@64", 512, 512, 64, 64), ("vae 128>256@256", 128, 256, 256, 256), ("vae 256>512@128", 256, 512, 128, 128),
          ("unet 640@32", 640, 640, 32, 32), ("unet 1280@32", 1280, 1280, 32, 32), ("unet 1920>640@32", 1920, 640, 32, 32),
          ("unet 320@64", 320, 320, 64, 64), ("unet 640>320@64", 640, 320, 64, 64), ("unet 960>320@64", 960, 320, 64, 64)]


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


for name, cin, cout, H, W in SHAPES:
    if only and only not in name:
        continue
    B, G = views, 32
    x = torch.randn(B, cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    gb, bb = gamma.bfloat16(), beta.bfloat16()
    bias = torch.randn(cout, device=dev)
    res = torch.randn(B, cout, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    packed, tile = ops.conv3x3_pack_weight(w)
    stats = ops.gn_stats_of(x, G)
    flop = 2.0 * B * H * W * 9 * cin * cout

    def fused(waves=8):
        return ops.conv3x3(x, packed, cout, tile, bias=bias, gn=(stats, gamma, beta, 1e-6, G), residual=res, stats_groups=G if (cout // G) % 4 == 0 else None, waves=waves)

    def plain(waves=8):
        return ops.conv3x3(x, packed, cout, tile, bias=bias, waves=waves)

    def chain():  # what the round-2 path runs per convolution: apply pass (statistics known), library conv, bias + residual + stats pass
        y = ops.group_norm(x, G, gb, bb, 1e-6, 1)
        h = F.conv2d(y, w, None, padding=1)
        return ops.bias_residual(res, h, bias.bfloat16(), stats_groups=G)

    def lib_conv():
        return F.conv2d(x, w, None, padding=1)

    with torch.no_grad():
        t_f, t_p, t_c, t_l = timeit(fused, iters), timeit(plain, iters), timeit(chain, iters), timeit(lib_conv, iters)
        t_f4, t_p4 = timeit(lambda: fused(4), iters), timeit(lambda: plain(4), iters)
        o1, o2 = fused().float(), chain().float()
        err = (o1 - o2).abs().max().item() / o2.abs().max().item()
    print(f"{name:18s} B={B} fused {t_f:8.1f} us {flop / t_f / 1e6:7.1f} TF | plain {t_p:8.1f} us {flop / t_p / 1e6:7.1f} TF | "
          f"4-wave: fused {t_f4:8.1f} us {flop / t_f4 / 1e6:7.1f} TF plain {t_p4:8.1f} us | library conv {t_l:8.1f} us {flop / t_l / 1e6:7.1f} TF | round-2 chain {t_c:8.1f} us | fused/chain x{t_c / t_f:.2f} | diff {err:.1e}", flush=True)
    del x, w, res, packed
    torch.cuda.empty_cache()
