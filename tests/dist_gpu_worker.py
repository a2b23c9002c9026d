"""Worker of tests/test_gpu_distributed.py: launched twice by torch.distributed.run (gloo, both ranks on the one device of the box).
  mode "spnet": MinkUNet18A under MinkowskiSyncBatchNorm + DDP, one sparse cloud per rank: forward rows, loss, BatchNorm running
                statistics and the (DDP-averaged) parameter gradients go to <out>/spnet_rank<r>.pt
  mode "infer": driver.infer over 4 synthetic scenes sharded across the ranks: the all-reduced scores go to <out>/infer_rank<r>.pt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from xmask3d_amd import config, driver
from xmask3d_amd import me_compat as ME

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sparse_cloud(seed, batch_index, dev, n=6000):
    """a seeded cloud on the surface of a box (distinct integer coordinates), features in [-1, 1]"""
    rng = np.random.RandomState(seed)
    pts = rng.randint(0, 48, size=(n, 3))
    pts[:, rng.randint(0, 3)] = 0  # a face, so that the cloud is surface-like (neighbourhoods are populated)
    pts = np.unique(np.concatenate([pts, rng.randint(0, 48, size=(n, 3)) * np.array([1, 1, 0])]), axis=0)
    coords = torch.from_numpy(np.concatenate([np.full((len(pts), 1), batch_index), pts], 1)).int().to(dev)
    feats = torch.from_numpy(rng.uniform(-1, 1, size=(len(pts), 3))).float().to(dev)
    return coords, feats


def spnet(out, rank, world, dev):
    from xmask3d_amd.pc_processor import PC_Binary_Processor

    torch.manual_seed(3)
    net = PC_Binary_Processor(arch_3d="MinkUNet18A").to(dev).train()
    ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(net)
    torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index], find_unused_parameters=False)
    coords, feats = sparse_cloud(100 + rank, 0, dev)
    y = ddp(ME.SparseTensor(feats, coords))
    w = torch.linspace(-1, 1, y.shape[0], device=dev)[:, None]
    loss = (y * w).sum()  # DDP averages gradients over ranks: the reference objective is the sum over both clouds / world
    loss.backward()
    grads = {n: p.grad.detach().cpu() for n, p in net.named_parameters() if p.grad is not None}
    bufs = {n: b.detach().cpu() for n, b in net.named_buffers() if "running" in n}
    torch.save({"y": y.detach().cpu(), "loss": float(loss), "grads": grads, "bufs": bufs,
                "sync": type(net.encoder.bn0).__name__}, os.path.join(out, f"spnet_rank{rank}.pt"))


def infer(out, rank, world, dev):
    cfg = config.load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
    cfg.scenes_per_forward = 1  # one scene per forward on every world size: identical shapes, identical kernels
    res = driver.infer(cfg, scenes=4, log=lambda s: None)
    torch.save(res, os.path.join(out, f"infer_rank{rank}.pt"))


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    cfg0 = config.load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
    rank, world, dev = driver.setup_distributed(cfg0)
    {"spnet": spnet, "infer": infer}[mode](out, rank, world, dev)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
