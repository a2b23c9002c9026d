"""Diagnostic: per-wave timeline of k_spconv_split's barrier intervals (needs the stamped build:
   make -C xmask3d_amd/csrc clean && make -C xmask3d_amd/csrc EXTRA=-DXM3D_SPLIT_STAMPS).  Never part of the product build."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xmask3d_amd import ops, synthetic
from xmask3d_amd._lib import lib

dev = torch.device("cuda:0")
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 96)
sc = synthetic.scene_s1()
grid, inds, inv = ops.voxelize(torch.from_numpy(sc.points).to(dev), np.diag([50.0, 50.0, 50.0, 1.0]))
coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=dev), grid], 1).contiguous()
cm = ops.CoordinateManager(coords)
n = coords.shape[0]
nbr, tiles, order = cm.kernel_map(1, 1, 3), cm.tiles(1, 1, 3), cm.order(1)
g = torch.Generator().manual_seed(1)
feats = torch.randn(n, cin, generator=g).to(dev)
W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev)
p4 = ops.pack_weight_split(W)
fs = torch.stack([feats.bfloat16(), (feats - feats.bfloat16().float()).bfloat16()]).contiguous() if os.environ.get("PRESPLIT") else None
run = lambda: ops.spconv_fwd(feats, W, nbr, n, order=order, packed=p4, tiles=tiles, relu=True, algo=ops.ALGO_SPLIT, feats_split=fs, ksplit=1)
for _ in range(3):
    run()
torch.cuda.synchronize()
L = lib()
L.xm3d_debug_clear_stamps()
run()
WGS, WAVES, SLOTS = 8, 16, 256
buf = np.zeros(WGS * WAVES * SLOTS, dtype=np.int64)
L.xm3d_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.size))
st = buf.reshape(WGS, WAVES, SLOTS)
for wg in (0, 3):
    t0 = st[wg][st[wg] > 0].min()
    waves = [w for w in range(WAVES) if st[wg, w, 0] > 0]
    nint = max(int(((st[wg, w, 4:254] > 0).sum()) // 2) for w in waves)
    print(f"== workgroup {wg}: {len(waves)} waves, {nint} intervals, total {st[wg, waves[0], 255] - t0} cycles")
    print("   init->setup / prologue (work, barrier wait):")
    for w in waves:
        s = st[wg, w]
        print(f"   wave {w:2d}: start {s[0]-t0:6d} setup {s[1]-s[0]:6d} prologue_work {s[2]-s[1]:6d} wait {s[3]-s[2]:6d} end {s[255]-t0:7d}")
    print("   per interval: work cycles per wave (arrival - previous release) | barrier release - last arrival")
    for ch in range(min(nint, 24)):
        works = []
        for w in waves:
            s = st[wg, w]
            prev = s[3] if ch == 0 else s[5 + 2 * (ch - 1)]
            works.append(int(s[4 + 2 * ch] - prev))
        rel = max(int(st[wg, w, 5 + 2 * ch]) for w in waves)
        arr = max(int(st[wg, w, 4 + 2 * ch]) for w in waves)
        print(f"   int {ch:2d}: " + " ".join(f"{x:5d}" for x in works) + f" | len {rel - max(int(st[wg, w, 3]) if ch == 0 else int(st[wg, w, 5 + 2 * (ch - 1)]) for w in waves):6d}")
