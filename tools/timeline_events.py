"""Device-side timeline of the pipelined scene loop from HIP events (no profiler attached): when do graph A (VAE encoder),
the sparse front S, graph B (UNet || VAE decoder, projections), graph C (decoders, mask-CLIP) and the post-processing P of
consecutive scenes start and end?  python tools/timeline_events.py [n_scenes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
G = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1   # scenes per forward
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = XMASK3d(cfg).eval().to(dev).set_dense_dtype(torch.bfloat16).set_channels_last(True).cast_head_weights().enable_dense_graph()
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
np.random.seed(1)
with torch.no_grad():
    def go():
        if G == 1:
            pipeline.infer_scene(model, sd, cfg, vox, next_scene=sd)
        else:
            pipeline.infer_scenes(model, [sd] * G, cfg, vox, next_scenes=[sd] * G)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    if "nogc" in sys.argv:
        import gc
        gc.collect(); gc.freeze(); gc.disable()
    model._trace = []
    base = torch.cuda.Event(enable_timing=True); base.record()
    host = []
    t0 = time.perf_counter()
    for i in range(n):
        go()
        host.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    total = 1e3 * (time.perf_counter() - t0)
tr = [(lab, base.elapsed_time(ev)) for lab, ev in model._trace]
print(f"{n} forwards of {G} scene(s) in {total:.1f} ms ({total / n / G:.1f} ms/scene); host returned from scene k at:", [round(h, 1) for h in host])
print("device time stamps (ms since start), in issue order:")
line = []
for lab, t in tr:
    line.append(f"{lab}@{t:.1f}")
    if lab == "S1":
        print("  " + "  ".join(line)); line = []
if line:
    print("  " + "  ".join(line))
