"""torch.profiler attribution of the dense branch (eager, B views, bf16, channels_last): device time per aten op + input
shapes, and per source line of xmask3d_amd.  python tools/prof_ops.py [B]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = pipeline.make_inference_model(XMASK3d(cfg).eval(), dev, torch.bfloat16, channels_last=True, graphs=False)  # the bench configuration, eager
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
T = np.diag([50.0, 50.0, 50.0, 1.0])
nv = len(sd.views)
batch = pipeline.build_group_batch([(sd, list(range(nv)))] * (B // nv), vox, [[T] * nv] * (B // nv)) if B > nv else \
    pipeline.build_scene_batch(sd, list(range(B)), vox, [T] * B)
# label every module call so that device time can be read per module (inclusive), independent of python stack capture
import torch.autograd.profiler as ap
def add_hooks(root):
    for name, m in root.named_modules():
        if not name or name.count(".") > 7:
            continue
        ctx = {}
        def pre(mod, inp, _n=name, _c=ctx):
            r = ap.record_function("M:" + _n); r.__enter__(); _c.setdefault("s", []).append(r)
        def post(mod, inp, out, _c=ctx):
            _c["s"].pop().__exit__(None, None, None)
        m.register_forward_pre_hook(pre); m.register_forward_hook(post)
add_hooks(model)
with torch.no_grad():
    _, cond, _ = model.encode_3d(batch["sinput"], batch["inds_reconstruct"], B)
    for _ in range(2):
        model.dense_forward(batch["img"], cond)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        model.dense_forward(batch["img"], cond)
        torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
print(ka.table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=50, max_shapes_column_width=80))
mods = [(e.key, e.device_time_total, e.count) for e in prof.key_averages() if e.key.startswith("M:")]
mods.sort()
print("inclusive device time per module (>= 0.25 ms):")
for k, t, c in mods:
    if t >= 250:
        print(f"  {t/1e3:8.2f} ms x{c:<4d} {'  ' * k.count('.')}{k[2:]}")
