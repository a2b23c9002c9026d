"""Which GEMM / implicit-GEMM convolution shapes the dense branch launches, with the device time of each (HIP events around every call,
eager forward at B views, bf16).  python tools/gemm_shapes.py [B=20]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xmask3d_amd import ops, pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = pipeline.make_inference_model(XMASK3d(cfg).eval(), dev, torch.bfloat16, channels_last=True, graphs=False)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
T = np.diag([50.0, 50.0, 50.0, 1.0])
nv = len(sd.views)
batch = pipeline.build_group_batch([(sd, list(range(nv)))] * (B // nv), vox, [[T] * nv] * (B // nv))

rec = collections.defaultdict(lambda: [0, 0.0])
on = [False]


def wrap(name, fn, key):
    def f(*a, **k):
        if not on[0]:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        e1.synchronize()
        v = rec[(name,) + key(*a, **k)]
        v[0] += 1
        v[1] += e0.elapsed_time(e1) * 1e3
        return r
    return f


def rows(x):
    return x.numel() // x.shape[-1]


ops.gemm = wrap("gemm", ops.gemm, lambda x, packed, n_rows, tile, bias=None, act=None, residual=None, waves=0:
                (rows(x), x.shape[-1], n_rows, act or "-", "res" if residual is not None else "-"))
ops.conv_gemm = wrap("conv_gemm", ops.conv_gemm, lambda x, packed, tile, n32, cout, ksize, stride=1, padding=(0, 0, 0, 0), bias=None, residual=None:
                     (tuple(x.shape), cout, ksize, stride, "res" if residual is not None else "-"))
ops.attention = wrap("attention", ops.attention, lambda q, k, v, bias=None, scale=None, out=None:
                     (tuple(q.shape), k.shape[1], "bias" if bias is not None else "-"))
with torch.no_grad():
    _, cond, _ = model.encode_3d(batch["sinput"], batch["inds_reconstruct"], B)
    for _ in range(2):
        model.dense_forward(batch["img"], cond)
    torch.cuda.synchronize()
    on[0] = True
    model.dense_forward(batch["img"], cond)
    torch.cuda.synchronize()
tot = sum(v[1] for v in rec.values())
print(f"{B} views, one dense forward: {tot / 1e3:.2f} ms in {sum(v[0] for v in rec.values())} wrapped calls")
for k, (c, t) in sorted(rec.items(), key=lambda kv: -kv[1][1])[:60]:
    flop = 2.0 * k[1] * k[2] * k[3] if k[0] == "gemm" else 0.0
    print(f"  {t / 1e3:7.3f} ms  x{c:<3d} avg {t / c:7.1f} us  {('%6.0f TF' % (flop * c / t / 1e6)) if flop else '         '}  {k}")
