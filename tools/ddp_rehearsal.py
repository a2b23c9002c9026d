"""2-rank DDP + SyncBatchNorm training rehearsal (torch.distributed.run, XM3D_DIST_BACKEND=gloo on a 1-GPU box):
checks that after one step every rank holds identical parameters and that losses are finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from xmask3d_amd import config, driver
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = config.load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
logs = []
model = driver.train(cfg, epochs=1, iters_per_epoch=2, views_per_gpu=1, log=lambda s: (logs.append(s), print(s, flush=True)))
core = model.module
flat = torch.cat([p.detach().float().reshape(-1) for n, p in core.named_parameters() if p.requires_grad])
probe = torch.stack([flat.sum(), flat.abs().sum(), flat[::1000].sum()]).double().cpu()
gathered = [torch.zeros_like(probe) for _ in range(dist.get_world_size())]
dist.all_gather(gathered, probe)
bn = core.pc_decoder.encoder.bn0
same = all(torch.equal(g, gathered[0]) for g in gathered)
if dist.get_rank() == 0:
    print("sync BN class:", type(bn).__name__, "| params identical across ranks:", same, "| probe", gathered[0].tolist(), flush=True)
assert same and type(bn).__name__ == "MinkowskiSyncBatchNorm"
dist.destroy_process_group()
