"""Per-layer timing of the sparse-conv kernels on the S1-full cloud (HIP events): algo 3 (exact f32 MFMA) vs algo 4 (bf16 split).
usage: python tools/spconv_bench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xmask3d_amd import ops, synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
sc = synthetic.scene_s1()
grid, inds, inv = ops.voxelize(torch.from_numpy(sc.points).to(dev), np.diag([50.0, 50.0, 50.0, 1.0]))
coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=dev), grid], 1).contiguous()
cm = ops.CoordinateManager(coords)


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


# (ts_in, ts_out, ksize, transposed, cin, cout) of MinkUNet34C layers that matter
layers = [(1, 1, 3, False, 96, 96), (1, 1, 3, False, 128, 96), (2, 2, 3, False, 96, 96), (2, 2, 3, False, 128, 96), (2, 2, 3, False, 32, 32),
          (4, 4, 3, False, 64, 64), (4, 4, 3, False, 192, 128), (4, 4, 3, False, 128, 128), (8, 8, 3, False, 128, 128),
          (8, 8, 3, False, 384, 256), (16, 16, 3, False, 256, 256), (1, 2, 2, False, 32, 32), (2, 1, 2, True, 96, 96), (1, 1, 1, False, 96, 256)]
g = torch.Generator().manual_seed(1)
for ts_in, ts_out, ks, tr, cin, cout in layers:
    n_in, n_out = cm.num(ts_in), cm.num(ts_out)
    ident = ks == 1 and ts_in == ts_out
    nbr = None if ident else cm.kernel_map(ts_in, ts_out, ks, tr)
    tiles, order = cm.tiles(ts_in, ts_out, ks, tr), cm.order(ts_out)
    pairs = n_out if ident else int((nbr >= 0).sum())
    K = 1 if ident else nbr.shape[0]
    feats = torch.randn(n_in, cin, generator=g).to(dev)
    W = (torch.randn(K, cin, cout, generator=g) * 0.05).to(dev)
    p3, p4 = ops.pack_weight(W), ops.pack_weight_split(W)
    t3 = timeit(lambda: ops.spconv_fwd(feats, W, nbr, n_out, order=order, packed=p3, tiles=tiles, relu=True, algo=ops.ALGO_TILES))
    t4 = timeit(lambda: ops.spconv_fwd(feats, W, nbr, n_out, order=order, packed=p4, tiles=tiles, relu=True, algo=ops.ALGO_SPLIT))
    fs = torch.stack([feats.bfloat16(), (feats - feats.bfloat16().float()).bfloat16()]).contiguous()
    t5 = timeit(lambda: ops.spconv_fwd(feats, W, nbr, n_out, order=order, packed=p4, tiles=tiles, relu=True, algo=ops.ALGO_SPLIT,
                                       feats_split=fs, want_split=True))
    fb = feats.bfloat16().contiguous()
    t6 = timeit(lambda: ops.spconv_fwd_bf16(fb, tuple(W.shape), p4, tiles, n_out, order=order, relu=True))
    o3 = ops.spconv_fwd(feats, W, nbr, n_out, order=order, packed=p3, tiles=tiles, relu=True, algo=ops.ALGO_TILES)
    o4 = ops.spconv_fwd(feats, W, nbr, n_out, order=order, packed=p4, tiles=tiles, relu=True, algo=ops.ALGO_SPLIT)
    err = ((o3 - o4).abs().max() / o3.abs().max()).item()
    flop = 2.0 * pairs * cin * cout
    print(f"ts {ts_in}->{ts_out} k{ks}{'T' if tr else ' '} {cin:3d}->{cout:3d} rows {n_out:6d} pairs {pairs:7d}: f32 {t3:7.1f} us ({flop / t3 / 1e6:6.1f} TF)  "
          f"split {t4:7.1f} us ({flop / t4 / 1e6:6.1f} TF)  x{t3 / t4:4.2f}  pre-split in+out {t5:7.1f} us ({flop / t5 / 1e6:6.1f} TF)  bf16 form {t6:7.1f} us ({flop / t6 / 1e6:6.1f} TF)  |f32-split| {err:.1e}", flush=True)
