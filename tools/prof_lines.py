"""Device time of the elementwise / copy aten ops of the dense branch (eager, bench configuration) attributed to the
xmask3d_amd source line that issued them.  python tools/prof_lines.py [B] [fp32|bf16] [lib] [all]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16  # python tools/prof_lines.py 20 fp32
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = pipeline.make_inference_model(XMASK3d(cfg).eval(), dev, DT, channels_last=True, graphs=False)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
T = np.diag([50.0, 50.0, 50.0, 1.0])
nv = len(sd.views)
batch = pipeline.build_group_batch([(sd, list(range(nv)))] * (B // nv), vox, [[T] * nv] * (B // nv))
import torch.autograd.profiler as ap
def add_hooks(root):
    for name, m in root.named_modules():
        if not name:
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        if "all" in sys.argv[2:]:  # python tools/prof_lines.py 20 bf16 all: the sparse 3D branch inside the profile too
            model.encode_3d(batch["sinput"], batch["inds_reconstruct"], B)
        model.dense_forward(batch["img"], cond)
        torch.cuda.synchronize()
skip = ("convolution", "mm", "bmm", "linear", "layer_norm", "group_norm", "attention", "softmax", "MSDeform", "conv2d", "matmul")
ONLY_LIB = "lib" in sys.argv[2:]  # python tools/prof_lines.py 20 fp32 lib: the library GEMMs / convolutions / norms instead (who still calls them)
acc = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith("aten::") or (any(k in e.name for k in skip) != ONLY_LIB):
        continue
    t = e.self_device_time_total
    if t <= 0:
        continue
    par, frame, via = e.cpu_parent, "?", ""
    while par is not None:
        if par.name.startswith("M:"):
            frame = par.name[2:]
            break
        if par.name.startswith("aten::") and not via:
            via = " <" + par.name[6:]
        par = par.cpu_parent
    frame = (frame.replace("backbone.feature_extractor.ldm_extractor.ldm.", "ldm.") + via)[-70:]
    shape = str(e.input_shapes)[:60]
    a = acc[(e.name, frame, shape)]
    a[0] += t; a[1] += 1
tot = sum(v[0] for v in acc.values())
print(f"elementwise / copy aten ops: {tot/1e3:.2f} ms of device time per dense forward (B={B})")
for (name, frame, shape), (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"  {t/1e3:7.3f} ms x{c:<4d} avg {t/c:6.1f} us  {name:22s} {frame:70s} {shape}")
