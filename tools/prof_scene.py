"""Wall-clock breakdown of one batched scene inference, synchronising between stages (so overlap is removed: this is the
serial cost of each stage, not the pipelined scene time bench.py reports).  python tools/prof_scene.py [cl]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = XMASK3d(cfg).eval().to(dev).set_dense_dtype(torch.bfloat16)
if "cl" in sys.argv[1:]:
    model.set_channels_last(True)
model.enable_dense_graph()
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
np.random.seed(1)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for rep in range(4):
        acc = {}
        t0 = T()
        batch = pipeline.build_scene_batch(sd, list(range(5)), vox); batch["compact_outputs"] = False
        t1 = T(); acc["build_scene_batch (voxelize x5)"] = t1 - t0
        pred_3d, cond, bs = model.encode_3d(batch["sinput"], batch["inds_reconstruct"], 5); t2 = T(); acc["encode_3d (34C+18A), serial"] = t2 - t1
        _, out = model(batch); t3 = T(); acc["model(batch) (3D || VAE-enc, dense graph, fuse)"] = t3 - t2
        preds = [pipeline.postprocess_view(cfg, out, batch, True, s) for s in range(5)]; t4 = T(); acc["postprocess_view x5"] = t4 - t3
        votes = [torch.zeros((sd.n, 19), dtype=torch.int32, device=dev) for _ in range(3)]
        seen = torch.zeros(sd.n, dtype=torch.bool, device=dev)
        for s in range(5):
            idx = sd.views[s]["idx"]
            for vt, p in zip(votes, preds[s]):
                vt.index_put_((idx, p), torch.ones_like(p, dtype=torch.int32), accumulate=True)
            seen[idx] = True
        t5 = T(); acc["votes"] = t5 - t4
        fill = pipeline.nearest_valid_fill(sd.points, seen); res = [vt.argmax(1)[fill] for vt in votes]
        t6 = T(); acc["nearest fill + argmax"] = t6 - t5
        if rep == 3:
            for k, v in acc.items(): print(f"{k:50s} {v*1e3:7.2f} ms")
            print(f"{'total (minus the duplicate encode_3d)':50s} {(t6-t0-(t2-t1))*1e3:7.2f} ms; seen {int(seen.sum())} of {sd.n}")
