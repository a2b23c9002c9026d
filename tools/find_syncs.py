"""List the host-synchronising torch calls of one steady-state scene inference (torch.cuda.set_sync_debug_mode) and time
how far the host runs ahead of the device.  python tools/find_syncs.py"""
import sys, os, time, warnings, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(0)
model = XMASK3d(cfg).eval().to(dev).set_dense_dtype(torch.bfloat16).set_channels_last(True).enable_dense_graph()
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
np.random.seed(1)
with torch.no_grad():
    for _ in range(2):
        pipeline.infer_scene(model, sd, cfg, vox, next_scene=sd)
    torch.cuda.synchronize()
    seen = set()
    def showwarning(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack() if "/xmask3d_amd/" in f.filename]
        key = (st[-1].filename, st[-1].lineno) if st else (filename, lineno)
        if key not in seen:
            seen.add(key)
            print("SYNC:", str(message)[:80], "<-", " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:]), flush=True)
    warnings.showwarning = showwarning
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    t0 = time.perf_counter()
    pipeline.infer_scene(model, sd, cfg, vox, next_scene=sd)
    t1 = time.perf_counter()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    # host time per stage (no device synchronisation added): where does the host wait?
    host = {}
    def wrap(obj, name):
        fn = getattr(obj, name)
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            host[name] = host.get(name, 0.0) + time.perf_counter() - t
            return r
        setattr(obj, name, w)
    for o, n in ((model, "eval_front"), (model, "eval_dense"), (model, "eval_fuse"), (pipeline, "postprocess_view"),
                 (pipeline, "nearest_valid_fill"), (pipeline, "build_scene_batch"), (model, "encode_3d")):
        wrap(o, n)
    for rep in range(3):
        host.clear()
        t0 = time.perf_counter()
        pipeline.infer_scene(model, sd, cfg, vox, next_scene=sd)
        t1 = time.perf_counter()
    print("host ms per stage:", {k: round(1e3 * v, 2) for k, v in host.items()}, "total", round(1e3 * (t1 - t0), 2))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host returned after {1e3*(t1-t0):.1f} ms, device drained {1e3*(t2-t1):.1f} ms later")
