"""Run training iterations and report the first non-finite loss term / gradient (debug aid).  python tools/train_nan_probe.py [iters] [graph]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xmask3d_amd import pipeline, synthetic
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(cfg.manual_seed)
model = XMASK3d(cfg).to(dev).train()
if "graph" in sys.argv[2:]:
    model.enable_train_graphs()
p3d = [p for n, p in model.named_parameters() if p.requires_grad and ("pc_decoder" in n or "pc_binary_head" in n)]
rest = [p for n, p in model.named_parameters() if p.requires_grad and not ("pc_decoder" in n or "pc_binary_head" in n)]
opt = torch.optim.AdamW([{"params": p3d, "lr": cfg.lr_3d}, {"params": rest, "lr": cfg.lr_others}], fused=True)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
np.random.seed(cfg.manual_seed)
for i in range(iters):
    batch = pipeline.build_train_batch(sd, [i % 5], vox, seed=cfg.manual_seed + i)
    losses, _ = model(batch)
    bad = [k for k, v in losses.items() if not torch.isfinite(v)]
    loss = sum(losses.values())
    opt.zero_grad(set_to_none=True)
    loss.backward()
    badg = [(n, float(p.grad.abs().max())) for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    gmax = max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)
    if gmax > 1e4:
        big = [(n, float(p.grad.abs().max()), int((p.grad.abs() > 1e4).sum()), p.grad.numel()) for n, p in model.named_parameters()
               if p.grad is not None and float(p.grad.abs().max()) > 1e4]
        print("   large gradients:", big[:8], flush=True)
    print(f"iter {i}: loss {float(loss):.4f} max|grad| {gmax:.3e} bad losses {bad[:6]} bad grads {len(badg)} {[n for n, _ in badg[:4]]}", flush=True)
    if bad or badg:
        wmax = [(n, float(p.detach().abs().max())) for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        print("non-finite parameters:", wmax[:6])
        break
    opt.step()
    badw = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    if badw:
        print("non-finite parameters after the step:", badw[:8])
        break
