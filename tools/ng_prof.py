import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmask3d_amd import ops, pipeline, synthetic
dev = torch.device("cuda:0")
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 5
seen = torch.zeros(sd.n, dtype=torch.bool, device=dev)
for v in sd.views[:nv]: seen |= v["vis"]
xyz = sd.points.float().contiguous()
print("n", sd.n, "seen", int(seen.sum()))
meth = sys.argv[2] if len(sys.argv) > 2 else "octree"
for _ in range(5): ops.nearest_valid_fill(xyz, seen, method=meth)
torch.cuda.synchronize()
