"""Wall-clock breakdown of one training iteration (synchronising between stages): python tools/prof_train_stages.py [B] [bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = torch.bfloat16 if "bf16" in sys.argv[2:] else torch.float32
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(cfg.manual_seed)
model = XMASK3d(cfg).to(dev).set_dense_dtype(dtype).train()
if "cl" in sys.argv[2:]:
    model.set_channels_last(True)
if "graph" in sys.argv[2:]:
    model.enable_train_graphs()
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
np.random.seed(0)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
import xmask3d_amd.criterion as crit
acc = {}
def timed(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = T(); r = fn(*a, **k); acc[label] = acc.get(label, 0.0) + T() - t
        return r
    setattr(obj, name, w)
timed(model, "encode_3d", "fwd sparse 3D nets")
timed(model, "encode_2d", "fwd dense 2D branch")
timed(model.criterion, "forward", "fwd criterion (matcher + losses)")
timed(model.criterion.matcher, "forward", "  of which Hungarian matcher")
for i in range(4):
    acc.clear()
    t0 = T()
    batch = pipeline.build_train_batch(sd, [(i + j) % 5 for j in range(B)], vox, seed=i); t1 = T()
    losses, _ = model(batch); loss = sum(losses.values()); t2 = T()
    opt.zero_grad(set_to_none=True); loss.backward(); t3 = T()
    opt.step(); t4 = T()
if "syncs" in sys.argv[2:]:
    import traceback, warnings, collections
    sites = collections.Counter()
    def showwarning(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack() if "/xmask3d_amd/" in f.filename]
        sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-2:]) if st else f"{filename}:{lineno}"] += 1
    warnings.showwarning = showwarning
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    saved = dict(acc)
    batch = pipeline.build_train_batch(sd, [j % 5 for j in range(B)], vox, seed=9)
    losses, _ = model(batch); loss = sum(losses.values())
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.set_sync_debug_mode("default")
    acc.clear(); acc.update(saved)
    print("host-synchronising torch calls in one iteration (site: count):")
    for k, v in sites.most_common(25):
        print(f"   {v:4d}  {k}")
print(f"B={B} dtype={dtype}: build batch {1e3*(t1-t0):.1f} | forward {1e3*(t2-t1):.1f} | backward {1e3*(t3-t2):.1f} | AdamW {1e3*(t4-t3):.1f} | total {1e3*(t4-t0):.1f} ms")
for k, v in acc.items():
    print(f"   {k:40s} {1e3*v:8.1f} ms")
