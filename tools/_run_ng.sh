set -e
timeout -k 10 300 python -m pytest tests/test_gpu_nearest_grid.py tests/test_gpu_msda_fuse.py -x -q 2>&1 | tail -5
timeout -k 10 120 python - <<'PY'
import torch
from xmask3d_amd import ops, pipeline, synthetic
dev = torch.device("cuda:0")
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
xyz = sd.points.float().contiguous()
def ev(f, reps=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
seen = torch.zeros(sd.n, dtype=torch.bool, device=dev)
for v in sd.views: seen |= v["vis"]
g = torch.Generator(device="cpu").manual_seed(0)
for name, valid in (("S1 vote fill (5 views seen)", seen), ("random 35% valid", (torch.rand(sd.n, generator=g) < 0.35).to(dev)),
                    ("random 90% valid", (torch.rand(sd.n, generator=g) < 0.9).to(dev))):
    a = pipeline.nearest_valid_fill(xyz, valid, "scan"); b = pipeline.nearest_valid_fill(xyz, valid, "octree")
    assert torch.equal(a, b) and torch.equal(a, pipeline.nearest_valid_fill(xyz, valid, 'sorted'))
    print(f"{name}: n={sd.n} valid={int(valid.sum())} scan {ev(lambda: pipeline.nearest_valid_fill(xyz, valid, 'scan')):.0f} us  octree {ev(lambda: pipeline.nearest_valid_fill(xyz, valid, 'octree')):.0f} us  sorted {ev(lambda: pipeline.nearest_valid_fill(xyz, valid, 'sorted')):.0f} us")
PY
