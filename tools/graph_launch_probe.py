"""Does hipGraphLaunch block the host while earlier work is still running on the stream?  Host time of replaying the dense
graph B (a) on an idle device, (b) behind ~80 ms of queued matmuls, (c) behind another replay."""
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
model = XMASK3d(cfg).eval().to(dev).set_dense_dtype(torch.bfloat16).set_channels_last(True).enable_dense_graph()
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
with torch.no_grad():
    pipeline.infer_scene(model, sd, cfg, vox)
    torch.cuda.synchronize()
    gs = [v for v in model._dense_graphs.values()]
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    def busy(n=40):
        for _ in range(n):
            a @ a
    def host(fn):
        t = time.perf_counter(); fn(); return 1e3 * (time.perf_counter() - t)
    for g in gs:
        g["ga"].replay()
    torch.cuda.synchronize()
    print(f"(a) idle device:        replay host {host(gs[0]['gb'].replay):.2f} ms"); t = time.perf_counter(); torch.cuda.synchronize(); print(f"    drain {1e3*(time.perf_counter()-t):.1f} ms")
    hb = host(busy); h = host(gs[0]["gb"].replay); t = time.perf_counter(); torch.cuda.synchronize()
    print(f"(b) behind matmuls:     matmul enqueue {hb:.2f} ms, replay host {h:.2f} ms, drain {1e3*(time.perf_counter()-t):.1f} ms")
    h0 = host(gs[0]["gb"].replay); h1 = host(gs[1]["gb"].replay); t = time.perf_counter(); torch.cuda.synchronize()
    print(f"(c) two slots back to back: {h0:.2f} ms, {h1:.2f} ms, drain {1e3*(time.perf_counter()-t):.1f} ms")
    h0 = host(gs[0]["gb"].replay); small = host(lambda: [a[:64, :64].add_(1) for _ in range(400)]); t = time.perf_counter(); torch.cuda.synchronize()
    print(f"(d) replay then 400 small kernels: replay {h0:.2f} ms, small kernels host {small:.2f} ms, drain {1e3*(time.perf_counter()-t):.1f} ms")
    os.environ  # (e) small kernels first, then replay
    small = host(lambda: [a[:64, :64].add_(1) for _ in range(400)]); h0 = host(gs[0]["gb"].replay); t = time.perf_counter(); torch.cuda.synchronize()
    print(f"(e) 400 small kernels then replay: small {small:.2f} ms, replay {h0:.2f} ms, drain {1e3*(time.perf_counter()-t):.1f} ms")
