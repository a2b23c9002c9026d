"""Wall-clock breakdown of one batched scene inference (synchronising between stages)."""
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
model = XMASK3d(cfg).eval().to(dev).set_dense_dtype(torch.bfloat16).enable_dense_graph()
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
np.random.seed(1)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for rep in range(3):
        acc = {}
        t0 = T()
        batch = pipeline.build_scene_batch(sd, list(range(5)), vox); t1 = T(); acc["build_batch(voxelize x5)"] = t1 - t0
        sinput = batch["sinput"]; inds = batch["inds_reconstruct"]
        pred_3d, cond, bs = model.encode_3d(sinput, inds, 5); t2 = T(); acc["encode_3d (34C+18A)"] = t2 - t1
        out = model._dense_graphed(batch["img"], cond); t3 = T(); acc["dense graph"] = t3 - t2
        out["pred_3d"] = pred_3d
        fused = model.fuse_eval(out, batch, bs); out.update(fused); out["binary_pred"] = (torch.sigmoid(bs) > 0.5).long(); t4 = T(); acc["fuse_eval"] = t4 - t3
        preds = [pipeline.postprocess_view(cfg, out, batch, True, s) for s in range(5)]; t5 = T(); acc["postprocess x5"] = t5 - t4
        votes = torch.zeros((sd.n, 19), dtype=torch.int32, device=dev); seen = torch.zeros(sd.n, dtype=torch.bool, device=dev)
        for s in range(5):
            idx = sd.views[s]["idx"]; votes.index_put_((idx, preds[s][0]), torch.ones_like(preds[s][0], dtype=torch.int32), accumulate=True); seen[idx] = True
        src = torch.nonzero(seen)[:, 0]; xyz = sd.points.float(); fill = src[pipeline.nearest_index(xyz[~seen], xyz[src])]
        p = votes.argmax(1); p[~seen] = p[fill]; t6 = T(); acc["vote + nearest fill"] = t6 - t5
        if rep == 2:
            for k, v in acc.items(): print(f"{k:32s} {v*1e3:7.2f} ms")
            print(f"{'total':32s} {(t6-t0)*1e3:7.2f} ms; seen {int(seen.sum())} of {sd.n}")
