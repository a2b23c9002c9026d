"""cProfile of the sparse 3D nets' training forward+backward (host side): python tools/prof_sparse_train_host.py"""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = XMASK3d(cfg).to(dev).train()
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
def step(i):
    batch = pipeline.build_train_batch(sd, [i % 5], vox, seed=i)
    p, c, b = model.encode_3d(batch["sinput"], batch["inds_reconstruct"].to(dev), 1)
    (p.sum() + c.sum() + b.sum()).backward()
for i in range(3):
    step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(5):
    step(i)
torch.cuda.synchronize()
print(f"sparse fwd+bwd: {(time.perf_counter()-t)/5*1e3:.1f} ms/iter")
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    step(i)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
