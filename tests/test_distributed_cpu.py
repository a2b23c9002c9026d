"""CPU, world_size 2, gloo: the cross-rank pieces of the path: BatchNorm statistics all-reduce,
the num_masks normaliser, DDP gradient averaging of a trainable head, and the weak-scaling shard/timing logic of bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xmask3d_amd.criterion import FeatureMerger
        from xmask3d_amd.me_compat import sync_moments

        g = torch.Generator().manual_seed(7)
        full = torch.randn(90, 6, generator=g, dtype=torch.float64)
        mine = full[:30] if rank == 0 else full[30:]  # ragged shards
        mean, var, n = sync_moments(mine.sum(0), (mine * mine).sum(0), mine.shape[0], True)
        ok_bn = torch.allclose(mean, full.mean(0)) and torch.allclose(var, full.var(0, unbiased=False)) and float(n) == 90
        num_masks = torch.tensor([3.0 if rank == 0 else 5.0])
        dist.all_reduce(num_masks)
        ok_nm = float(torch.clamp(num_masks / world, min=1)) == 4.0
        torch.manual_seed(0)
        net = torch.nn.parallel.DistributedDataParallel(FeatureMerger(4))
        x = torch.full((2, 4), float(rank + 1))
        net(x, x).sum().backward()
        gw = net.module.linear.weight.grad.clone()
        ref = torch.full_like(gw, 2 * 1.5)  # mean over ranks of sum over 2 rows of inputs (1 or 2)
        ok_ddp = torch.allclose(gw, ref)
        t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok_max = abs(float(t) - 0.2) < 1e-12
        q.put((rank, ok_bn, ok_nm, ok_ddp, ok_max))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r
