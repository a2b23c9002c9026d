"""nearest-2x-upsample + conv3x3 (ldm Upsample) on the HIP kernel, both geometries, against F.interpolate + library conv.
usage: python tools/conv_ups_bench.py [views=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import xmask3d_amd  # noqa: F401
from xmask3d_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, c, h in (("vae 512 64->128", 512, 64), ("vae 512 128->256", 512, 128), ("vae 256 256->512", 256, 256), ("unet 1280 16->32", 1280, 16), ("unet 640 32->64", 640, 32)):
    x = torch.randn(B, c, h, h, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.zeros(c, device=dev)
    packed, tile = ops.conv3x3_pack_weight(w)
    flop = 2.0 * B * (2 * h) ** 2 * 9 * c * c
    t = {wv: timeit(lambda: ops.conv3x3(x, packed, c, tile, bias=bias, upsample=True, waves=wv, stats_groups=32)) for wv in (8, 4)}
    tl = timeit(lambda: F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, padding=1))
    print(f"{name:18s} 8-wave {t[8]:8.1f} us {flop / t[8] / 1e6:7.1f} TF | 4-wave {t[4]:8.1f} us {flop / t[4] / 1e6:7.1f} TF | interpolate + library conv {tl:8.1f} us", flush=True)
