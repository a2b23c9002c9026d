"""Time the dense branch for B views under a few settings: python tools/prof_dense.py B mode..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]); modes = sys.argv[2:]
if "bench" in modes: torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = XMASK3d(cfg).eval().to(dev).set_dense_dtype(torch.bfloat16)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
batch = pipeline.build_scene_batch(sd, list(range(B)), vox, [np.diag([50.0, 50.0, 50.0, 1.0])] * B)
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
with torch.no_grad():
    _, cond, _ = model.encode_3d(batch["sinput"], batch["inds_reconstruct"], B)
    t = time.time(); model.dense_forward(batch["img"], cond); torch.cuda.synchronize(); print(f"first call {time.time()-t:.1f} s", flush=True)
    ext = model.backbone.feature_extractor
    img = (batch["img"].float() / 255).to(torch.bfloat16)
    x = ev(lambda: ext(dict(img=img), cond))
    feats = ext(dict(img=img), cond)
    def proj():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return model.backbone.forward_features(feats, (512, 512))
    pr = ev(proj)
    d = ev(lambda: model.encode_2d(batch["img"], cond))
    out = model.encode_2d(batch["img"], cond)
    c = ev(lambda: model.clip_head(out["images"], out["pred_masks"]))
    g = ev(lambda: model.enable_dense_graph()._dense_graphed(batch["img"], cond)) if "graph" in modes else float("nan")
    print(f"B={B} {modes}: SD extractor {x:.1f} | projections(autocast) {pr:.1f} | encode_2d {d:.1f} | mask-CLIP {c:.1f} | graph total {g:.1f} ms", flush=True)
