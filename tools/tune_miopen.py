"""Generate the MIOpen user find-db for the conv shapes of the dense branch (B=5 and B=1, bf16) with exhaustive find
(torch.backends.cudnn.benchmark).  Run on a GPU box; the db lands in $MIOPEN_USER_DB_PATH."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.backends.cudnn.benchmark = True
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("MIOPEN_USER_DB_PATH =", os.environ.get("MIOPEN_USER_DB_PATH"), flush=True)
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
with torch.device(dev):
    model = XMASK3d(cfg, dense_dtype=torch.float32 if os.environ.get("XM3D_TUNE_DTYPE") == "fp32" else torch.bfloat16).eval()
model = model.to(dev)
if os.environ.get('XM3D_CL') == '1':
    model.set_channels_last(True)
    print('channels_last on', flush=True)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
T = np.diag([50.0, 50.0, 50.0, 1.0])
with torch.no_grad():
    for B in ([int(a) for a in sys.argv[1:]] or [5, 1]):
        batch = pipeline.build_scene_batch(sd, [i % 5 for i in range(B)], vox, [T] * B)
        t = time.time()
        model(batch)
        torch.cuda.synchronize()
        print(f"B={B}: tuned in {time.time()-t:.1f} s", flush=True)
        t = time.time()
        for _ in range(3):
            model(batch)
        torch.cuda.synchronize()
        print(f"B={B}: {(time.time()-t)/3*1e3:.1f} ms per forward (eager, tuned)", flush=True)
