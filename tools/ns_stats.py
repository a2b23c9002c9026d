import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmask3d_amd import ops, pipeline, synthetic
from xmask3d_amd._lib import lib, check
dev = torch.device("cuda:0")
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
xyz = sd.points.float().contiguous()
seen = torch.zeros(sd.n, dtype=torch.bool, device=dev)
for v in sd.views: seen |= v["vis"]
g = torch.Generator().manual_seed(0)
for name, valid in (("S1", seen), ("rand35", (torch.rand(sd.n, generator=g) < 0.35).to(dev)), ("rand90", (torch.rand(sd.n, generator=g) < 0.9).to(dev))):
    n = sd.n
    out = torch.empty(n, dtype=torch.int64, device=dev)
    ws = torch.zeros(lib().xm3d_nearest_valid_fill_sorted_workspace_bytes(n), dtype=torch.uint8, device=dev)
    v8 = valid.view(torch.uint8)
    check(lib().xm3d_nearest_valid_fill_sorted(ctypes.c_void_p(xyz.data_ptr()), n, ctypes.c_void_p(v8.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                             ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
    torch.cuda.synchronize()
    p = ws[:64].view(torch.int32).cpu().tolist()
    nref = p[6]
    print(f"{name}: refs {nref} ({(nref+63)//64} tiles), waves {p[10]}, tiles scanned total {p[8]} mean {p[8]/max(p[10],1):.1f} max {p[9]}")
