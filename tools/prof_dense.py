"""Time the dense 2D branch (encode_2d) and mask-CLIP of one view under a few settings."""
import sys, os, copy, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = XMASK3d(cfg).eval().to(dev)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
batch = pipeline.build_view_batch(sd, 3, vox, np.diag([50.0, 50.0, 50.0, 1.0]))
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, (time.perf_counter() - t) / reps * 1e3
which = sys.argv[1:] or ["fp32", "bf16", "bf16cl"]
with torch.no_grad():
    _, cond, _ = model.encode_3d(batch["sinput"], batch["inds_reconstruct"], 1)
    for w in which:
        model.set_dense_dtype(torch.float32 if w == "fp32" else torch.bfloat16)
        model.set_channels_last(w.endswith("cl"))
        d, wall = ev(lambda: model.encode_2d(batch["img"], cond))
        out = model.encode_2d(batch["img"], cond)
        c, _ = ev(lambda: model.clip_head(out["images"], out["pred_masks"]))
        ext = model.backbone.feature_extractor
        img = (batch["img"].float() / 255).to(model.dense_dtype)
        if model.channels_last: img = img.contiguous(memory_format=torch.channels_last)
        x, _ = ev(lambda: ext(dict(img=img), cond))
        print(f"{w}: encode_2d {d:.1f} ms (wall {wall:.1f}) of which SD extractor {x:.1f} ms; mask-CLIP {c:.1f} ms", flush=True)
