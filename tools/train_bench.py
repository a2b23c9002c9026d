"""Training-iteration timing (SURVEY §8d metric (ii)): B views per GPU, fwd + losses + bwd + AdamW step.
python tools/train_bench.py [B] [iters]   (single GPU; multi-GPU via torch.distributed.run uses DDP over RCCL)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
from xmask3d_amd import pipeline, synthetic, me_compat as ME
from xmask3d_amd.config import load_cfg_from_cfg_file
from xmask3d_amd.xmask3d import XMASK3d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dtype = torch.bfloat16 if "bf16" in sys.argv[3:] else torch.float32
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    dist.init_process_group("nccl", device_id=dev)
cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
torch.manual_seed(cfg.manual_seed)
model = XMASK3d(cfg).to(dev).set_dense_dtype(dtype).train()
if "cl" in sys.argv[3:]:
    model.set_channels_last(True)
if "graph" in sys.argv[3:]:
    model.enable_train_graphs()
if world > 1:
    ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)      # per-GPU batch < 4 (run/train.py:185-187)
    model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=False)
core = model.module if world > 1 else model
# two parameter groups as run/train.py:152-169
p3d = [p for n, p in core.named_parameters() if p.requires_grad and ("pc_decoder" in n or "pc_binary_head" in n)]
rest = [p for n, p in core.named_parameters() if p.requires_grad and not ("pc_decoder" in n or "pc_binary_head" in n)]
opt = torch.optim.AdamW([{"params": p3d, "lr": cfg.lr_3d}, {"params": rest, "lr": cfg.lr_others}], fused=True)
sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
vox = pipeline.default_voxelizer(device=dev)
np.random.seed(cfg.manual_seed + rank)

def it(i):
    batch = pipeline.build_train_batch(sd, [(i + j + rank) % 5 for j in range(B)], vox, seed=cfg.manual_seed + i)
    losses, _ = model(batch)
    loss = sum(losses.values())
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return float(loss.detach())

for i in range(2):
    l = it(i)
torch.cuda.synchronize()
if world > 1: dist.barrier()
t = time.perf_counter()
for i in range(iters):
    l = it(i + 2)
    if "trace" in sys.argv[3:] and rank == 0:
        print(f"  iter {i + 2}: loss {l:.4f}", flush=True)
torch.cuda.synchronize()
if world > 1: dist.barrier()
dt = (time.perf_counter() - t) / iters
if rank == 0:
    print(f"train[{'bf16' if dtype == torch.bfloat16 else 'fp32'} frozen nets]: world {world} x {B} views/GPU: {dt*1e3:.1f} ms/iter = {1/dt:.2f} iters/s ({world*B/dt:.2f} views/s), last loss {l:.3f}, "
          f"max mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
if world > 1: dist.destroy_process_group()
