"""Micro-benchmark of the sparse 3D branch: S1-full (whole ~107k-voxel cloud) and S1-view (one view)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import ops, synthetic, me_compat as ME
from xmask3d_amd.pc_processor import PC_Processor, PC_Binary_Processor

dev = torch.device("cuda:0")
sc = synthetic.scene_s1()
T = np.diag([50.0, 50.0, 50.0, 1.0])
torch.manual_seed(0)
net = PC_Processor().eval().to(dev); net2 = PC_Binary_Processor().eval().to(dev)

def prep(pts, cols):
    grid, inds, inv = ops.voxelize(torch.from_numpy(pts).to(dev), T)
    coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=dev), grid], 1).contiguous()
    feats = (torch.from_numpy(cols).to(dev)[inds] / 127.5 - 1).float()
    return coords, feats, inv

def run(coords, feats, reps, tag):
    def once():
        with torch.no_grad():
            s = ME.SparseTensor(feats, coords)
            a = net(s); b = net2(s)
        return a, b
    for _ in range(3): once()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): once()
    torch.cuda.synchronize(); dt = (time.time() - t) / reps
    print(f"{tag}: N={coords.shape[0]} {dt*1e3:.2f} ms per (34C+18A) forward", flush=True)

which = sys.argv[2] if len(sys.argv) > 2 else "both"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if which in ("both", "full"):
    coords, feats, _ = prep(sc.points, sc.colors)
    run(coords, feats, reps, "S1-full")
if which in ("batch",):  # the bench forward: 4 scenes x 5 views in one batch
    from xmask3d_amd import pipeline
    sd = pipeline.SceneOnDevice(sc, dev)
    vox = pipeline.default_voxelizer(device=dev)
    b = pipeline.build_group_batch([(sd, list(range(5)))] * 4, vox, [[T] * 5] * 4)
    run(b["coords"], b["sinput"].F, reps, "S1-batch20")
if which in ("both", "view"):
    vis, r, c = synthetic.view_subset(sc, 3)
    coords, feats, _ = prep(sc.points[vis], sc.colors[vis])
    run(coords, feats, reps, "S1-view3")
