"""The operand-split pass of the f32-accurate convolution (xm3d_split_f16_nhwc with the GroupNorm affine + SiLU folded in) on the VAE's
shapes, GB/s of its 8 bytes per element.  python tools/split_bench.py [B=25]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xmask3d_amd import ops
from xmask3d_amd._lib import lib
from xmask3d_amd.ops import _ptr, _stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for C, H in [(512, 128), (256, 256), (128, 512), (512, 64)]:
    x = torch.randn(B, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    stats = ops.gn_stats_of(x, 32)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    hi = torch.empty((B, C, H, H), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
    lo = torch.empty_like(hi)
    ws = torch.empty(B * C * 2, dtype=torch.float32, device=dev)

    def run(act):
        ops.check(lib().xm3d_split_f16_nhwc(_ptr(x), B, H * H, C, _ptr(stats), _ptr(gamma), _ptr(beta), None, 0, 1e-6, 32, act, ops.F16_X_SCALE,
                                            _ptr(hi), _ptr(lo), _ptr(ws), _stream()), "split")

    res = []
    for act in (1, 0):
        run(act)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run(act)
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 10 * 1e3)
    nbytes = x.numel() * 8
    print(f"C{C} {H}x{H} B{B}: GroupNorm+SiLU {res[0]:8.1f} us ({nbytes / res[0] / 1e3:6.0f} GB/s)   GroupNorm only {res[1]:8.1f} us ({nbytes / res[1] / 1e3:6.0f} GB/s)", flush=True)
print("check flag:", lib().xm3d_check_flag())
