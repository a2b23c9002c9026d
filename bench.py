#!/usr/bin/env python3
"""XMask3D scene inference benchmark on MI355X.

python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path over one synthetic ScanNet-shaped scene S1 (119,963
points, 5 views of 240x320 -> 512x512): per view voxelise the visible subset (HIP), MinkUNet34C +
MinkUNet18A (HIP sparse conv), 3D-conditioned SD-v1 VAE/UNet feature extractor, Mask2Former pixel +
transformer decoder (HIP deformable attention), mask-CLIP ViT-L/14, 2D->3D fusion (HIP), open-vocabulary
logits; then vote over views and fill unseen points.  Inputs are resident in HBM before the timed region.
Weights are seeded random (no checkpoints offline); config = ScanNet B15N4.

Prints ONE JSON line (rank 0).  value = scenes/s over all ranks (weak scaling: every rank runs K scenes).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16
DENSE_TFLOP_PER_VIEW_MIN = 2.83  # SURVEY.md §8d, dead compute pruned
DENSE_TFLOP_PER_VIEW_REF = 4.79  # as the reference computes


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """threads for the CPU baseline: the cores this process may run on, capped at the GPU box's per-GPU share"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def event_ms(fn, reps, stream=None):
    """average device time of fn() over reps launches, HIP events on torch's current stream"""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / reps


def spconv_roofline(dev):
    """Roofline of the dominant hand-written kernel, k_spconv_tiles<6>, on the S1-full 96->96 k=3 layer at
    tensor stride 1 (MinkUNet34C block8): algorithmic FLOP = 2*P*Cin*Cout, bytes = gather+scatter model."""
    from xmask3d_amd import ops, synthetic

    sc = synthetic.scene_s1()
    T = np.diag([50.0, 50.0, 50.0, 1.0])
    grid, inds, inv = ops.voxelize(torch.from_numpy(sc.points).to(dev), T)
    coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=dev), grid], 1).contiguous()
    cm = ops.CoordinateManager(coords)
    n = coords.shape[0]
    nbr, tiles, order = cm.kernel_map(1, 1, 3), cm.tiles(1, 1, 3), cm.order(1)
    pairs = int((nbr >= 0).sum().item())
    cin = cout = 96
    g = torch.Generator(device="cpu").manual_seed(1)
    feats = torch.randn(n, cin, generator=g).to(dev)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev)
    packed = ops.pack_weight(W)
    ms = event_ms(lambda: ops.spconv_fwd(feats, W, nbr, n, order=order, packed=packed, tiles=tiles, relu=True), 20)
    flop = 2.0 * pairs * cin * cout
    gs_bytes = pairs * (cin + cout) * 4 + pairs * 8 + 27 * cin * cout * 4
    return {"kernel": "xm3d::k_spconv_tiles<6>", "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
            # HBM-side bytes per launch from rocprofv3 PMC passes on `bench.py --roofline-only` (profiles/r01_roofline_spconv_pmc.txt):
            # FETCH_SIZE 207164 KiB + WRITE_SIZE 40130 KiB; algorithmic gather+scatter model below for comparison
            "traffic": 253.2e6, "algorithmic_bytes_gather_scatter": gs_bytes,
            "algorithmic_bytes_compulsory": (2 * n * cin + 27 * cin * cout) * 4 + 8 * pairs,
            "avg_launch_us": ms * 1e3, "pairs": pairs, "voxels": n, "cin": cin, "cout": cout,
            "gather_scatter_GBps": gs_bytes / (ms * 1e-3) / 1e9}


def balanced_groups(n_scenes, per_forward):
    """n scenes in ceil(n / per_forward) forwards, as even as possible (5 scenes at 4 per forward -> 3 + 2, not 4 + 1)"""
    if n_scenes <= 0:
        return []
    ng = -(-n_scenes // per_forward)
    return [n_scenes // ng + (1 if i < n_scenes % ng else 0) for i in range(ng)]


def train_leg(args, cfg, dev, rank, world, backend, sd, voxelizer, log):
    """SURVEY 8d metric (ii): training iterations per second at one view per GPU (BASELINE config 3: global batch = one
    scene per GPU), forward + 37 weighted losses + backward + AdamW, gradients all-reduced by DDP when world > 1."""
    import torch.distributed as dist
    from xmask3d_amd import me_compat as ME, pipeline
    from xmask3d_amd.driver import build_optimizer
    from xmask3d_amd.xmask3d import XMASK3d

    tdt = torch.bfloat16 if args.train_dtype == "bf16" else torch.float32
    torch.manual_seed(cfg.manual_seed)
    model = XMASK3d(cfg).to(dev).set_dense_dtype(tdt).train()
    model.backbone.feature_extractor.ldm_extractor.enable_train_graph()
    if world > 1:
        ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)  # per-GPU batch < 4 (run/train.py:185-187)
        torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=False)
    opt = build_optimizer(model, cfg)
    nv = len(sd.views)

    def it(i):
        batch = pipeline.build_train_batch(sd, [(i + rank) % nv], voxelizer, seed=cfg.manual_seed + i)
        losses, _ = model(batch)
        loss = sum(losses.values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for i in range(2):
        it(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.train_steps):
        last = it(i + 2)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not bool(torch.isfinite(last.detach())):
        raise SystemExit("bench: training loss is not finite")
    log(f"training leg: {args.train_steps} iterations in {dt:.3f} s")
    return {"iters_per_s": args.train_steps / dt, "ms_per_iter": dt / args.train_steps * 1e3, "steps": args.train_steps,
            "views_per_gpu": 1, "global_batch_views": world, "frozen_nets_dtype": args.train_dtype,
            "scope": "forward + 37 weighted losses (CPU Hungarian matching like the reference) + backward + AdamW; frozen UNet "
                     "forward/backward replayed as HIP graphs; DDP gradient all-reduce + MinkowskiSyncBatchNorm when n_gpus > 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="dtype of the frozen dense nets (SD, CLIP)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--views-per-batch", type=int, default=0, help="views per forward (0 = all views of the scene; 1 = reference loop)")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find through the shipped find-db)")
    ap.add_argument("--scenes-per-forward", type=int, default=4,
                    help="scenes whose views share one forward (views are independent until the vote); 1 = one scene per forward")
    ap.add_argument("--train-steps", type=int, default=0,
                    help="also time this many training iterations (1 view per GPU, DDP when --gpus > 1) and report them under \"train\"")
    ap.add_argument("--train-dtype", default="fp32", choices=["fp32", "bf16"], help="dtype of the frozen nets in the training leg")
    ap.add_argument("--nchw", action="store_true", help="keep NCHW activations in the frozen conv nets (default: channels-last)")
    ap.add_argument("--no-graph", action="store_true", help="launch the dense branch eagerly instead of replaying a HIP graph")
    ap.add_argument("--faithful-dead-compute", action="store_true", help="also run what the reference computes and discards")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the kernel-level roofline run of k_spconv_tiles (the command profiles/ rocprof summaries are taken on)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU fallback for the product path)")
    backend = os.environ.get("XM3D_DIST_BACKEND", "nccl")  # "gloo" only to rehearse N>1 on a box with fewer GPUs than ranks
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and backend == "nccl":
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {ndev} GPUs visible")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    if args.roofline_only:
        print(json.dumps({"roofline": spconv_roofline(dev)}))
        return

    import __graft_entry__

    if rank == 0 and not os.path.exists(os.path.join(ROOT, "xmask3d_amd", "libxm3d_hip.so")):
        __graft_entry__.build()
    if world > 1:
        dist.barrier()

    from xmask3d_amd import pipeline, synthetic  # noqa: E402  (sets MIOPEN_USER_DB_PATH before the first convolution)
    from xmask3d_amd.config import load_cfg_from_cfg_file

    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
    log(f"MIOPEN_USER_DB_PATH={os.environ.get('MIOPEN_USER_DB_PATH')} cudnn.benchmark={torch.backends.cudnn.benchmark}")
    from xmask3d_amd.xmask3d import XMASK3d

    cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "configs", "xmask3d_scannet_B15N4.yaml"))
    torch.manual_seed(cfg.manual_seed)
    torch.set_num_threads(host_threads())
    log(f"building model (seeded random weights), host threads {host_threads()}")
    cpu_model = XMASK3d(cfg, prune_dead_compute=not args.faithful_dead_compute).eval()
    dense_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = pipeline.make_inference_model(cpu_model, dev, dense_dtype, channels_last=not args.nchw, graphs=not args.no_graph)

    scene = synthetic.scene_s1()
    sd = pipeline.SceneOnDevice(scene, dev)
    voxelizer = pipeline.default_voxelizer(cfg.voxel_size, dev)
    np.random.seed(cfg.manual_seed + rank)

    G = 1 if (args.no_graph or args.views_per_batch) else max(1, args.scenes_per_forward)

    def group_sizes(n_scenes):
        return balanced_groups(n_scenes, G)

    def run(n_scenes):
        """n_scenes steps (one step = one scene).  Scenes go through the model in groups of G (all views of the group in
        one forward); consecutive groups are software-pipelined (the next group's front is issued on side streams)."""
        if G == 1 and (args.no_graph or args.views_per_batch):
            for k in range(n_scenes):
                out = pipeline.infer_scene(model, sd, cfg, voxelizer, views_per_batch=args.views_per_batch or None)
            return out
        sizes = group_sizes(n_scenes)
        for gi, g in enumerate(sizes):
            nxt = [sd] * sizes[gi + 1] if gi + 1 < len(sizes) else None
            out = pipeline.infer_scenes(model, [sd] * g, cfg, voxelizer, next_scenes=nxt)[-1]
        return out

    if not args.no_graph and not args.views_per_batch:
        # setup, not a step: capture the HIP graphs of every batch shape the timed region will meet
        with torch.no_grad():
            for g in sorted(set(group_sizes(args.steps)) | set(group_sizes(1))):
                model._graphs_for(torch.cat([sd.img_all] * g), torch.zeros(g * len(sd.views), 768, device=dev))
        torch.cuda.synchronize()
        log(f"HIP graphs captured for {G} scene(s) per forward")
    log("model on device; warmup")
    for i in range(args.warmup):
        run(1)
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    torch.cuda.synchronize()
    # the model, its HIP graphs and the scene tables are millions of long-lived python objects: move them to the permanent
    # generation so that a cyclic-GC pass inside the timed loop stays cheap (an unfrozen gen-2 pass stalls the host ~50 ms,
    # which delays the next forward: tools/timeline_events.py with and without "nogc")
    import gc

    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    from xmask3d_amd import ops as _ops

    marker = torch.zeros(1, 3, dtype=torch.int32, device=dev)
    _ops.fnv_keys(marker)  # k_fnv_only: a dispatch that only ever marks the timed window in kernel traces (profiles/)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    preds = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _ops.fnv_keys(marker)
    ok = all(bool(torch.isfinite(p.float()).all()) and int(p.min()) >= 0 and int(p.max()) < cfg.test_classes for p in preds)
    if not ok:
        raise SystemExit("bench: scene predictions are not finite class ids - refusing to report a number")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    train = train_leg(args, cfg, dev, rank, world, backend, sd, voxelizer, log) if args.train_steps > 0 else None
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    log(f"timed {args.steps} steps in {elapsed:.3f} s")
    n_views = len(sd.views)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed

    # stage-level roofline of the dominant stage (dense 2D branch) + kernel-level roofline of the dominant HIP kernel
    vb = args.views_per_batch or len(sd.views) * G
    mats = [np.diag([50.0, 50.0, 50.0, 1.0])] * vb
    batch = pipeline.build_scene_batch(sd, [i % len(sd.views) for i in range(vb)], voxelizer, mats)
    with torch.no_grad():
        pred_3d, cond, bs = model.encode_3d(batch["sinput"], batch["inds_reconstruct"], vb)
        dense_fn = (lambda: model._dense_graphed(batch["img"], cond)) if not args.no_graph else (lambda: model.dense_forward(batch["img"], cond))
        dense_ms = event_ms(dense_fn, 5) / vb
        sparse_ms = event_ms(lambda: model.encode_3d(pipeline.build_scene_batch(sd, [i % len(sd.views) for i in range(vb)], voxelizer, mats)["sinput"],
                                                      batch["inds_reconstruct"], vb), 3) / vb
    dense_tflop = DENSE_TFLOP_PER_VIEW_REF if args.faithful_dead_compute else DENSE_TFLOP_PER_VIEW_MIN
    peak = BF16_MFMA_PEAK_TF if args.dtype == "bf16" else FP32_MFMA_PEAK_TF
    log(f"dense branch {dense_ms:.1f} ms/view, sparse branch {sparse_ms:.1f} ms/view; kernel roofline")
    roof_kernel = spconv_roofline(dev)
    # `roofline`: kernel-level, the dominant hand-written kernel (k_spconv_tiles), HIP events live + PMC traffic from profiles/
    roofline = dict(roof_kernel)
    roofline["scope"] = ("dominant hand-written HIP kernel; algorithmic FLOP = 2*pairs*cin*cout per launch (SURVEY 8d), one launch = "
                         "one sparse-conv layer of MinkUNet34C block8 on the full S1 cloud; the scene-level time is dominated by the "
                         "library-kernel dense stage reported under roofline_dense_stage")
    roofline_stage = {"bound": "mfma", "achieved": dense_tflop / (dense_ms * 1e-3), "peak": peak, "unit": "TFLOP/s",
                      "frac": dense_tflop / (dense_ms * 1e-3) / peak, "traffic": None, "views_per_forward": vb,
                      "scope": "dense 2D branch per view (SD VAE+UNet, projections, pixel+transformer decoder, mask-CLIP): MIOpen / hipBLASLt / "
                               "AOTriton kernels + HIP GroupNorm, " + ("HIP graph replay" if not args.no_graph else "eager launches"),
                      "ms_per_view": dense_ms, "algorithmic_tflop_per_view": dense_tflop, "sparse3d_ms_per_view": sparse_ms}

    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
        from oracle import model_oracle, voxel_oracle

        log("cpu baseline (1 view through the oracle)")
        v = 3
        vis, rows, cols = synthetic.view_subset(scene, v)
        pts = scene.points[vis]
        T = np.diag([50.0, 50.0, 50.0, 1.0])
        t1 = time.perf_counter()
        grid, inds, inv = voxel_oracle.voxelize_with_matrix(pts, T)
        coords = torch.from_numpy(np.concatenate([np.zeros((len(grid), 1)), grid], 1).astype(np.int32))
        feats = torch.from_numpy((scene.colors[vis][inds] / 127.5 - 1).astype(np.float32))
        cbatch = {"sinput": model_oracle.CpuSparseTensor(feats, coords), "img": torch.from_numpy(scene.images[v]).permute(2, 0, 1)[None],
                  "x_label": torch.from_numpy(rows).long(), "y_label": torch.from_numpy(cols).long(),
                  "inds_reconstruct": torch.from_numpy(inv), "captions": ("a room",),
                  "ori_coords": torch.cat([torch.zeros(len(pts), 1), torch.from_numpy(pts).float()], 1)}
        model_oracle.forward_cpu(cpu_model, cbatch)
        t_view = time.perf_counter() - t1
        cpu_baseline = {"value": 1.0 / (t_view * n_views), "unit": "scenes/s", "cores": torch.get_num_threads(), "kind": "port",
                        "sample": f"1 of {n_views} views of S1 (view {v}, {len(pts)} points) through oracle/model_oracle.py "
                                  f"(fp32, same weights), {t_view:.1f} s, extrapolated x{n_views}; vote/fill excluded"}

    out = {
        "metric": "ScanNet scenes/sec (infer)", "value": value, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype + " (frozen SD/CLIP nets, decoder GEMMs) + f32 (sparse 3D, deformable attention, statistics, logits)", "data": "synthetic",
        "config": {"workload": "ScanNet B15N4 inference, synthetic scene S1 (119963 pts, 5 views 240x320->512x512), "
                               f"{vb} views ({G} scene{'s' if G > 1 else ''}) per forward, seeded random weights", "views_per_scene": n_views, "parallelism": f"dp{world} (scene level, no collective)",
                   "dead_compute": "as reference" if args.faithful_dead_compute else "pruned (SURVEY F7)",
                   "layout": "NCHW" if args.nchw else "channels-last (NHWC) frozen nets",
                   "schedule": "eager launches" if args.no_graph else "3 HIP graphs per forward (2 slots), next forward's front software-pipelined on side streams"},
        "roofline": roofline, "roofline_dense_stage": roofline_stage, "cpu_baseline": cpu_baseline, "train": train,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
