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
HBM_PEAK_GBPS = 8000.0      # HBM3E spec (6.3 TB/s measured copy)
L2_PEAK_GBPS = 34500.0      # aggregate L2 rate (MI355X_MICROARCH.md, L2 section)
PMC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "roofline_pmc.json")
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


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC summary (profiles/roofline_pmc.json, written by
    tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes of `bench.py --roofline-only`, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950); None if the file has no entry for it"""
    try:
        with open(PMC_FILE) as f:
            ent = json.load(f).get(kernel_key)
        return None if ent is None else float(ent["traffic_bytes"])
    except (OSError, ValueError, KeyError):
        return None


def conv_roofline(dev):
    """Roofline of the time-dominant hand-written kernel, k_conv3x3<256,2,false,NW> (csrc/conv.hip), on the shape with the largest
    share of the forward: the 512 -> 512 channel ResnetBlock convolution of the SD VAE at 128 x 128, 20 views, with everything
    the bench forward fuses into it (GroupNorm affine + SiLU on the staged input, bias, residual, output moments).
    Algorithmic FLOP = 2 * B*H*W * 9*Cin * Cout per launch; bytes = input + residual + output (bf16) + weights."""
    from xmask3d_amd import ops

    B, C, H, W, G = 20, 512, 128, 128, 32
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(B, C, H, W, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn(B, C, H, W, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev)
    gamma, beta, bias = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    packed, tile = ops.conv3x3_pack_weight(w)
    stats = ops.gn_stats_of(x, G)
    ms = event_ms(lambda: ops.conv3x3(x, packed, C, tile, bias=bias, gn=(stats, gamma, beta, 1e-6, G), residual=res, stats_groups=G), 10)
    ms_plain = event_ms(lambda: ops.conv3x3(x, packed, C, tile, bias=bias), 10)
    wl = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ms_lib = event_ms(lambda: torch.nn.functional.conv2d(x, wl, None, padding=1), 10)
    flop = 2.0 * B * H * W * 9 * C * C
    nbytes = B * H * W * (C + 2 * C) * 2 + 9 * C * C * 2
    tf = flop / (ms * 1e-3) / 1e12
    waves = ops.lib().xm3d_conv3x3_default_waves(H, W, C, C)  # the geometry the forward uses for this layer
    other = 4 if waves == 8 else 8  # the other geometry, timed beside it (and so present in the PMC passes of tools/roofline_profile.sh)
    ms_other = event_ms(lambda: ops.conv3x3(x, packed, C, tile, bias=bias, gn=(stats, gamma, beta, 1e-6, G), residual=res, stats_groups=G,
                                            waves=other), 5)
    kname = f"k_conv3x3<256,2,false,{waves}>"
    return {"kernel": "xm3d::" + kname, "bound": "mfma", "achieved": tf, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": tf / BF16_MFMA_PEAK_TF, "traffic": pmc_traffic(kname), "algorithmic_bytes": nbytes,
            "avg_launch_us": ms * 1e3, "shape": f"{B} x {H}x{W} x {C}->{C}, GroupNorm(32)+SiLU in, bias+residual+moments out",
            f"geometry_{other}_waves_us": ms_other * 1e3, f"geometry_{other}_waves_traffic": pmc_traffic(f"k_conv3x3<256,2,false,{other}>"),
            "plain_conv_us": ms_plain * 1e3, "plain_conv_frac": flop / (ms_plain * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF,
            "library_conv_alone_us": ms_lib * 1e3, "library_conv_alone_frac": flop / (ms_lib * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF}


def _spconv_forms(dev, cm, ts, cin, cout, seed, reps=20):
    """One sparse-conv layer (k = 3 at tensor stride `ts`) in both forms the forward uses, measured live with HIP events:
    the plain-bf16 form (bf16 configuration: one bf16 plane in / out, one MFMA per product) and the f32-accurate split form (fp32
    configuration and training: f32 rows + pre-split bf16 hi / lo copies in and out, three MFMAs per product).  SURVEY 8d prices the
    sparse convolution against HBM on the gather + scatter model: bytes = pairs * (cin + cout) * e + 8 * pairs + K * cin * cout * e;
    the matrix rate executed is listed against the bf16 MFMA peak the instructions actually run on."""
    from xmask3d_amd import ops

    nbr, tiles, order = cm.kernel_map(ts, ts, 3), cm.tiles(ts, ts, 3), cm.order(ts)
    n = cm.num(ts)
    pairs = int((nbr >= 0).sum().item())
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = torch.randn(n, cin, generator=g).to(dev)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev)
    p4 = ops.pack_weight_split(W)
    fb = feats.bfloat16().contiguous()
    fs = torch.stack([feats.bfloat16(), (feats - feats.bfloat16().float()).bfloat16()]).contiguous()
    ms_bf = event_ms(lambda: ops.spconv_fwd_bf16(fb, tuple(W.shape), p4, tiles, n, order=order, relu=True), reps)
    ms_sp = event_ms(lambda: ops.spconv_fwd(feats, W, nbr, n, order=order, packed=p4, tiles=tiles, relu=True, algo=ops.ALGO_SPLIT, feats_split=fs,
                                            want_split=True), reps)
    flop = 2.0 * pairs * cin * cout

    def model_bytes(e):
        return pairs * (cin + cout) * e + pairs * 8 + 27 * cin * cout * e

    def form(ms, e, mfmas, compulsory):
        gs = model_bytes(e)
        tf = flop / (ms * 1e-3) / 1e12
        return {"avg_launch_us": ms * 1e3, "algorithmic_bytes_gather_scatter": gs, "gather_scatter_GBps": gs / (ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": gs / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_compulsory": compulsory,
                "algorithmic_TFLOPs": tf, "executed_bf16_TFLOPs": mfmas * tf, "executed_frac_of_bf16_mfma_peak": mfmas * tf / BF16_MFMA_PEAK_TF}

    bf = form(ms_bf, 2, 1, 2 * n * cin * 2 + 27 * cin * cout * 2 + 8 * pairs)
    sp = form(ms_sp, 4, 3, (2 * n * cin + 27 * cin * cout) * 4 + 2 * n * cout * 2 + 8 * pairs)
    sp["algorithmic_frac_of_f32_matrix_peak"] = sp["algorithmic_TFLOPs"] / FP32_MFMA_PEAK_TF
    return {"pairs": pairs, "voxels": n, "cin": cin, "cout": cout}, bf, sp, (feats, W, nbr, n, order, tiles)


def _spconv_object(kernel_bf, kernel_sp, meta, bf, sp):
    out = {"kernel": kernel_bf, "bound": "hbm", "achieved": bf["gather_scatter_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": bf["frac_of_hbm_peak"], "traffic": pmc_traffic(kernel_bf)}
    out.update(meta)
    out.update({k: v for k, v in bf.items() if k not in ("gather_scatter_GBps", "frac_of_hbm_peak")})
    sp = dict(sp, kernel=kernel_sp, traffic=pmc_traffic(kernel_sp))
    out["f32_accurate_form"] = sp
    return out


def spconv_roofline(dev):
    """`roofline_spconv`: the dominant kernel of the sparse 3D branch on the S1-full 96 -> 96 k = 3 layer at tensor stride 1 (MinkUNet34C
    block8, 107 k voxels, 418 k pairs): the plain-bf16 form the bf16 configuration runs (headline object, HBM roofline on the gather +
    scatter model as SURVEY 8d prices it) and, under `f32_accurate_form`, the split-operand form of the fp32 configuration."""
    from xmask3d_amd import ops, synthetic

    sc = synthetic.scene_s1()
    grid, inds, inv = ops.voxelize(torch.from_numpy(sc.points).to(dev), np.diag([50.0, 50.0, 50.0, 1.0]))
    coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=dev), grid], 1).contiguous()
    cm = ops.CoordinateManager(coords)
    meta, bf, sp, (feats, W, nbr, n, order, tiles) = _spconv_forms(dev, cm, 1, 96, 96, 1)
    out = _spconv_object("k_spconv_split<6,1,96,4,3,2,true,true,1>", "k_spconv_split<6,1,96,4,3,2,true,false,1>", meta, bf, sp)
    p3 = ops.pack_weight(W)
    ms3 = event_ms(lambda: ops.spconv_fwd(feats, W, nbr, n, order=order, packed=p3, tiles=tiles, relu=True, algo=ops.ALGO_TILES), 10)
    out["exact_f32_kernel_us"] = ms3 * 1e3
    return out


def sparse_network_roofline(dev, model):
    """`roofline_sparse_network`: both sparse U-Nets (MinkUNet34C + MinkUNet18A, heads included) on the FULL S1 cloud (S1-full, SURVEY
    8d: 106 950 voxels) in the form the benched model runs them (bf16 configuration: plain-bf16 convolutions): algorithmic 379.3 GFLOP
    (convolutions, measured pair counts) + 42.6 GFLOP (heads) per forward / measured time, against the bf16 matrix peak the kernels
    execute on.  HIP events around whole forwards (rulebooks included: they are rebuilt every forward, as in training)."""
    from xmask3d_amd import me_compat as ME, ops, synthetic

    sc = synthetic.scene_s1()
    grid, inds, inv = ops.voxelize(torch.from_numpy(sc.points).to(dev), np.diag([50.0, 50.0, 50.0, 1.0]))
    coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=dev), grid], 1).contiguous()
    feats = (torch.from_numpy(sc.colors).to(dev)[inds] / 127.5 - 1).float().contiguous()

    def once():
        with torch.no_grad():
            sp = ME.SparseTensor(feats, coords)
            model.pc_decoder(sp)
            model.pc_binary_head(sp)

    ms = event_ms(once, 5)
    gflop = 379.3 + 42.6
    tf = gflop / ms  # GFLOP / ms = TFLOP/s
    form = "plain-bf16" if getattr(model, "sparse_dtype", None) == torch.bfloat16 else "f32-accurate split (3 MFMAs per product)"
    mf = 1 if form == "plain-bf16" else 3
    return {"kernel": "MinkUNet34C + MinkUNet18A on S1-full (k_spconv_split + rulebook kernels), " + form + " form", "bound": "mfma",
            "achieved": mf * tf, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s (executed)", "frac": mf * tf / BF16_MFMA_PEAK_TF, "algorithmic_TFLOPs": tf,
            "ms_per_forward": ms, "voxels": int(coords.shape[0]), "algorithmic_gflop": gflop,
            "note": "network level: coordinate / rulebook kernels, BatchNorm folding and the two Linear heads inside the time; the sparse branch is "
                    "latency / gather bound, nowhere near the matrix roof: per-layer HBM rooflines are `roofline_spconv` / `roofline_spconv_window`"}


def spconv_window_roofline(dev, sd, voxelizer, n_views):
    """`roofline_spconv_window`: the sparse-conv instantiation with the largest share of the timed window - the 64 -> 64 channel k = 3 layers at
    tensor stride 2 (MinkUNet block 2) - on the coordinates of the bench forward itself (`n_views` views in one batch), pair count read
    live from the kernel map; both forms as in `roofline_spconv`."""
    from xmask3d_amd import ops, pipeline

    mats = [np.diag([50.0, 50.0, 50.0, 1.0])] * n_views
    batch = pipeline.build_scene_batch(sd, [i % len(sd.views) for i in range(n_views)], voxelizer, mats)
    cm = ops.CoordinateManager(batch["sinput"].C)
    meta, bf, sp, _ = _spconv_forms(dev, cm, 2, 64, 64, 2)
    out = _spconv_object("k_spconv_split<4,1,64,8,3,2,true,true,1>", "k_spconv_split<4,1,64,8,3,2,true,false,1>", meta, bf, sp)
    out["views"] = n_views
    return out


def kernel_rooflines(dev):
    """`roofline_kernels`: the other hand-written kernels that show up in the timed window, each measured live (HIP events)
    on the shape it runs at in the bench forward (20 views) against the roof that bounds it."""
    from xmask3d_amd import ops

    out = []
    g = torch.Generator(device="cpu").manual_seed(3)
    # deformable attention forward, pixel-decoder shape, 20 views: gather model 5376*8*12*4 taps x 128 B per view and layer
    B, S, H, D, L, P = 20, 5376, 8, 32, 3, 4
    shapes = torch.tensor([[16, 16], [32, 32], [64, 64]], device=dev)
    lsi = torch.tensor([0, 256, 1280], device=dev)
    value = torch.randn(B, S, H, D, generator=g).to(dev)
    loc = torch.rand(B, S, H, L, P, 2, generator=g).to(dev)
    w = torch.softmax(torch.randn(B, S, H, L * P, generator=g), -1).view(B, S, H, L, P).to(dev)
    ms = event_ms(lambda: ops.msda_forward(value, shapes, lsi, loc, w), 10)
    gathered = B * S * H * L * P * 4 * D * 4
    comp = (value.numel() + loc.numel() + w.numel() + B * S * H * D) * 4
    out.append({"kernel": "xm3d::k_msda_fwd<float,4>", "bound": "l2", "achieved": gathered / (ms * 1e-3) / 1e9, "peak": L2_PEAK_GBPS,
                "unit": "GB/s", "frac": gathered / (ms * 1e-3) / 1e9 / L2_PEAK_GBPS, "avg_launch_us": ms * 1e3,
                "algorithmic_bytes": gathered, "compulsory_bytes": comp, "compulsory_GBps": comp / (ms * 1e-3) / 1e9,
                "note": "bytes = bilinear taps gathered (the value tensor, 5.5 MB per view, is L2 / Infinity-Cache resident, so the "
                        "gather is priced against the aggregate L2 rate, 34.5 TB/s); 20 views"})
    # GroupNorm (statistics + apply), channels-last bf16, the VAE's 256^2 x 256-channel maps, 5 views: 3 passes over the tensor
    x = torch.randn(5, 256, 256, 256, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gw, gb = torch.ones(256, device=dev, dtype=torch.bfloat16), torch.zeros(256, device=dev, dtype=torch.bfloat16)
    ms = event_ms(lambda: ops.group_norm(x, 32, gw, gb, 1e-6, True), 10)
    nbytes = 3 * x.numel() * 2
    out.append({"kernel": "xm3d::k_gn_stats_nhwc + k_gn_apply_nhwc <bf16>", "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "avg_launch_us": ms * 1e3,
                "algorithmic_bytes": nbytes, "note": "two reads + one write of a (5,256,256,256) bf16 map, both launches"})
    # implicit-GEMM convolution (k_gemm, rows gathered from the NHWC image while staging): the two instantiations with the largest share
    for name, cin, cout, h, k, stride, pad in (("SD VAE downsample 512 -> 512, 3x3 stride 2, 128^2 -> 64^2", 512, 512, 128, 3, 2, (0, 0, 1, 1)),
                                                ("SD UNet ResnetBlock 2560 -> 1280, 3x3, 16^2 (split-K 2, f32 slabs + fixed-order finish)", 2560, 1280, 16, 3, 1,
                                                 (1, 1, 1, 1))):
        xc = torch.randn(20, cin, h, h, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wc = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev, torch.bfloat16)
        packed, tile, n32 = ops.conv_gemm_pack_weight(wc)
        bpad = torch.zeros(n32, device=dev)
        ms = event_ms(lambda: ops.conv_gemm(xc, packed, tile, n32, cout, k, stride, pad, bias=bpad), 10)
        ho = (h + pad[0] + pad[2] - k) // stride + 1
        fl = 2.0 * 20 * ho * ho * cin * k * k * cout
        nb = (xc.numel() + wc.numel() + 20 * ho * ho * cout) * 2
        out.append({"kernel": "xm3d::k_gemm<..., GF_CONV> (implicit GEMM convolution)", "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12,
                    "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, "avg_launch_us": ms * 1e3,
                    "algorithmic_bytes": nb, "note": name + ", 20 views, bf16 NHWC in / out, f32 accumulation, bias in the epilogue"})
    # exact nearest neighbour fill: 120 k queries x 40 k references, VALU bound: 8 flop per (query, reference)
    q = torch.rand(120000, 3, generator=g).to(dev)
    r = torch.rand(40000, 3, generator=g).to(dev)
    ms = event_ms(lambda: ops.nearest_index(q, r), 10)
    fl = 8.0 * q.shape[0] * r.shape[0]
    out.append({"kernel": "xm3d::k_nearest", "bound": "valu", "achieved": fl / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF,
                "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, "avg_launch_us": ms * 1e3,
                "note": "brute force 120 k x 40 k, 3 sub + 3 fma + compare per pair; peak = f32 vector rate"})
    # the same question as the pipeline asks it (every invalid point <- nearest valid point), Morton-sorted and tile-pruned:
    # no closed-form roofline (the work depends on the geometry); reported as time and as the scan-equivalent pair rate
    pts = torch.cat([q, r])
    valid = torch.cat([torch.zeros(q.shape[0], dtype=torch.bool, device=dev), torch.ones(r.shape[0], dtype=torch.bool, device=dev)])
    ms2 = event_ms(lambda: ops.nearest_valid_fill(pts, valid, method="sorted"), 10)
    out.append({"kernel": "xm3d_nearest_valid_fill_sorted (k_ns_query + sort)", "bound": "valu", "achieved": fl / (ms2 * 1e-3) / 1e12,
                "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s (scan-equivalent)", "frac": None, "avg_launch_us": ms2 * 1e3,
                "note": "same 120 k queries / 40 k references (uniform random cloud): exact answer with a fraction of the pair tests; "
                        "scan-equivalent rate = what the brute-force scan would need to match this time"})
    return out


def balanced_groups(n_scenes, per_forward):
    """n scenes in ceil(n / per_forward) forwards, as even as possible (5 scenes at 4 per forward -> 3 + 2, not 4 + 1)"""
    if n_scenes <= 0:
        return []
    ng = -(-n_scenes // per_forward)
    return [n_scenes // ng + (1 if i < n_scenes % ng else 0) for i in range(ng)]


def train_leg(args, cfg, dev, rank, world, backend, sd, voxelizer, log, train_dtype="fp32"):
    """SURVEY 8d metric (ii): training iterations per second at one view per GPU (BASELINE config 3: global batch = one
    scene per GPU), forward + 37 weighted losses + backward + AdamW, gradients all-reduced by DDP when world > 1."""
    import torch.distributed as dist
    from xmask3d_amd import me_compat as ME, pipeline
    from xmask3d_amd.driver import build_optimizer
    from xmask3d_amd.xmask3d import XMASK3d

    tdt = torch.bfloat16 if train_dtype == "bf16" else torch.float32
    torch.manual_seed(cfg.manual_seed)
    model = XMASK3d(cfg).to(dev).set_dense_dtype(tdt).train()
    model.enable_train_graphs()
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
    import gc

    gc.collect()
    gc.freeze()  # the step is host-bound: a generation-2 pass of python's cyclic GC over the model objects inside 4-8 timed steps is +30 %
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
            "views_per_gpu": 1, "global_batch_views": world, "frozen_nets_dtype": train_dtype,
            "scope": "forward + 37 weighted losses (Hungarian matching of all 10 decoder outputs in one HIP launch) + backward + AdamW; frozen UNet "
                     "forward/backward and the frozen VAE stages (inference kernels) replayed as HIP graphs; LayerNorm / GroupNorm forward + backward and "
                     "the linear layers' bias gradients on own fixed-order kernels; DDP gradient all-reduce + MinkowskiSyncBatchNorm when n_gpus > 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="dtype of the frozen dense nets (SD, CLIP)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--views-per-batch", type=int, default=0, help="views per forward (0 = all views of the scene; 1 = reference loop)")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find through the shipped find-db)")
    ap.add_argument("--scenes-per-forward", type=int, default=5,
                    help="scenes whose views share one forward (views are independent until the vote); 1 = one scene per forward.  "
                         "5 (25 views): 43.2 scenes/s against 42.7 at 4 and 43.2 at 6 on one box after the round-4 attention kernels "
                         "(profiles/r04_bench_scenes_per_forward_ab2.log); the driver's 20 steps are four whole groups of 5")
    ap.add_argument("--train-steps", type=int, default=8,
                    help="training iterations timed after the inference steps (1 view per GPU, DDP when --gpus > 1), reported under "
                         "\"train\" (fp32 as the reference trains, plus a bf16-frozen-nets run); 0 = skip")
    ap.add_argument("--train-deadline", type=int, default=300,
                    help="seconds after which a training leg that has not finished is abandoned (the inference line is printed regardless)")
    ap.add_argument("--fp32-steps", type=int, default=20, help="inference steps of the fp32 configuration reported under \"fp32\" (0 = skip)")
    ap.add_argument("--scene-pool", type=int, default=8,
                    help="distinct seeded scenes the timed loop cycles through (8 = two different 4-scene groups alternate)")
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
        from xmask3d_amd import pipeline as _pl, synthetic as _syn

        _sd = _pl.SceneOnDevice(_syn.scene_s1(seed=5557), dev)
        print(json.dumps({"roofline": conv_roofline(dev), "roofline_spconv": spconv_roofline(dev),
                          "roofline_spconv_window": spconv_window_roofline(dev, _sd, _pl.default_voxelizer(0.02, dev), 20)}))
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

    # distinct seeded scenes (same generator, different seeds: different geometry, point counts and images), resident in
    # HBM before the timed region; the timed loop cycles through them
    scenes = [synthetic.scene_s1(seed=cfg.manual_seed + 101 * i) for i in range(max(1, args.scene_pool))]
    scene = scenes[0]
    sds = [pipeline.SceneOnDevice(sc, dev) for sc in scenes]
    sd = sds[0]
    voxelizer = pipeline.default_voxelizer(cfg.voxel_size, dev)
    np.random.seed(cfg.manual_seed + rank)

    G = 1 if (args.no_graph or args.views_per_batch) else max(1, args.scenes_per_forward)

    def group_sizes(n_scenes, g=None):
        return balanced_groups(n_scenes, g or G)

    def run(mdl, n_scenes, g=None):
        """n_scenes steps (one step = one scene).  Scenes go through the model in groups of g (all views of the group in
        one forward); consecutive groups are software-pipelined (the next group's front is issued on side streams)."""
        g = g or G
        if g == 1 and (args.no_graph or args.views_per_batch):
            for k in range(n_scenes):
                out = pipeline.infer_scene(mdl, sds[k % len(sds)], cfg, voxelizer, views_per_batch=args.views_per_batch or None)
            return out
        sizes = group_sizes(n_scenes, g)
        groups, k = [], 0
        for sz in sizes:
            groups.append([sds[(k + j) % len(sds)] for j in range(sz)])
            k += sz
        for gi, grp in enumerate(groups):
            nxt = groups[gi + 1] if gi + 1 < len(groups) else None
            out = pipeline.infer_scenes(mdl, grp, cfg, voxelizer, next_scenes=nxt)[-1]
        return out

    def capture(mdl, n_steps, g=None):
        # setup, not a step: capture the HIP graphs of every batch shape the timed region will meet
        with torch.no_grad():
            for gs in sorted(set(group_sizes(n_steps, g)) | set(group_sizes(1, g))):
                mdl._graphs_for(torch.cat([sd.img_all] * gs), torch.zeros(gs * len(sd.views), 768, device=dev))
        torch.cuda.synchronize()

    def timed(mdl, n_steps, g=None):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        preds = run(mdl, n_steps, g)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = all(bool(torch.isfinite(p.float()).all()) and int(p.min()) >= 0 and int(p.max()) < cfg.test_classes for p in preds)
        if not ok:
            raise SystemExit("bench: scene predictions are not finite class ids - refusing to report a number")
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    graphed = not args.no_graph and not args.views_per_batch
    if graphed:
        capture(model, args.steps)
        capture(model, 1, 1)
        log(f"HIP graphs captured for {G} scene(s) per forward")
    log("model on device; warmup")
    # warm-up on the SCHEDULE the timed region runs: whole groups of G scenes per forward, consecutive groups pipelined.  (Until
    # round 3 the W warm-up steps ran one scene per forward, so the first G-scene forwards - allocator growth for the 4-scene
    # sparse batch, lazily loaded library kernels for its shapes - fell inside the timed region: one run in eight lost ~120 ms
    # there, profiles/r03_bench_geometry_ab.log group 0.)  W steps rounded up to whole groups, at least two groups.
    n_warm = max(2, -(-args.warmup // G)) * G if graphed else args.warmup
    if graphed:
        run(model, n_warm)
        torch.cuda.synchronize()
        log(f"warmup: {n_warm} scenes in groups of {G} done")
    else:
        for i in range(args.warmup):
            run(model, 1)
            torch.cuda.synchronize()
            log(f"warmup step {i} done")
    torch.cuda.synchronize()
    # the model, its HIP graphs and the scene tables are millions of long-lived python objects: move them to the permanent
    # generation so that a cyclic-GC pass inside the timed loop stays cheap (an unfrozen gen-2 pass stalls the host ~50 ms,
    # which delays the next forward: tools/timeline_events.py with and without "nogc")
    import gc

    gc.collect()
    gc.freeze()
    from xmask3d_amd import ops as _ops

    marker = torch.zeros(1, 3, dtype=torch.int32, device=dev)
    _ops.fnv_keys(marker)  # k_fnv_only: a dispatch that only ever marks the timed window in kernel traces (profiles/)
    elapsed = timed(model, args.steps)
    _ops.fnv_keys(marker)
    # latency of ONE scene (one scene per forward, nothing to pipeline against): reported beside the throughput number
    latency_ms = None
    if graphed and world == 1:
        run(model, 1, 1)
        latency_ms = timed(model, 3, 1) / 3 * 1e3
    # the same pipeline with every net in fp32 (the reference's arithmetic; BASELINE config 2 names bf16, hence not `value`)
    fp32 = None
    if args.dtype == "bf16" and args.fp32_steps > 0 and graphed:
        # the SAME schedule as the headline number (G scenes per forward, the same number of timed steps): the shipped MIOpen
        # find-db covers the fp32 NHWC convolutions at 20 views since round 3 (tools/tune_miopen.py 20 with XM3D_TUNE_DTYPE=fp32).
        # This is the configuration inside north_star's 1e-3 on the per-point logits (tests/test_gpu_bench_parity.py).
        m32 = pipeline.make_inference_model(cpu_model, dev, torch.float32, channels_last=not args.nchw, graphs=True)
        capture(m32, args.fp32_steps, G)
        run(m32, G, G)
        dt32 = timed(m32, args.fp32_steps, G)
        vb32 = len(sd.views) * G
        b32 = pipeline.build_scene_batch(sd, [i % len(sd.views) for i in range(vb32)], voxelizer, [np.diag([50.0, 50.0, 50.0, 1.0])] * vb32)
        with torch.no_grad():
            _, cond32, _ = m32.encode_3d(b32["sinput"], b32["inds_reconstruct"], vb32)
            dense32_ms = event_ms(lambda: m32._dense_graphed(b32["img"], cond32), 3) / vb32
        from xmask3d_amd._lib import lib as _xm3d_lib

        range_flag = int(_xm3d_lib().xm3d_check_flag())  # sticky device flag: an operand left the half's range in a split (must be 0)
        if range_flag:
            log(f"fp32 configuration: RANGE FLAG {range_flag} - an activation beyond the split operands' half range; results of this leg are invalid")
        fp32 = {"value": world * args.fp32_steps / dt32, "unit": "scenes/s", "ms_per_step": dt32 / args.fp32_steps * 1e3, "steps": args.fp32_steps,
                "split_range_flag": range_flag,
                "scenes_per_forward": G,
                "dtype": "f32 everywhere, on the 16-bit matrix cores: the sparse 3D convolutions (bf16 x 3 split operands), every dense convolution and "
                         "GEMM of the frozen nets (two-term split in IEEE halves, three accumulating passes, f32 accumulation; <= 1e-6 per layer against f64); "
                         "softmax attention = torch MATH (two library f32 GEMMs around an aten softmax)",
                "roofline_dense_stage": {"bound": "mfma", "achieved": DENSE_TFLOP_PER_VIEW_MIN / (dense32_ms * 1e-3), "peak": FP32_MFMA_PEAK_TF,
                                         "unit": "TFLOP/s", "frac": DENSE_TFLOP_PER_VIEW_MIN / (dense32_ms * 1e-3) / FP32_MFMA_PEAK_TF,
                                         "ms_per_view": dense32_ms, "views_per_forward": vb32,
                                         "scope": "dense 2D branch per view in f32 (convolutions / GEMMs of the frozen nets: three half-precision MFMA passes over "
                                                  "two-term split operands; trainable heads' GEMMs and the attention products: library, f32 matrix path)"}}
        del b32, cond32
        del m32
        torch.cuda.empty_cache()
        log(f"fp32 configuration: {fp32['value']:.2f} scenes/s")
    out = None

    def finish(train):
        """the optional training leg comes LAST and under a deadline: whatever happens in it (a collective that never
        completes on some fabric, a capture that stalls), the inference line measured above is still printed"""
        import faulthandler
        import threading

        def on_timeout():
            # a leg that does not finish is a FAILURE of this run: the stacks of every python thread name the blocking call, the
            # inference line measured above is still printed (with train_error), and the exit code is non-zero
            print(f"[bench] rank {rank}: training leg exceeded {args.train_deadline} s; python stacks follow", file=sys.stderr, flush=True)
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            if rank == 0:
                out["train"] = None
                out["train_error"] = (f"training leg did not finish within {args.train_deadline} s (python stacks of all threads on stderr); "
                                      "inference line unaffected; exit code 3")
                print(json.dumps(out), flush=True)
            sys.stderr.flush()
            os._exit(3)

        timer = None
        if args.train_steps > 0:
            if world > 1:
                # rank 0 reaches this point after its roofline / baseline measurements, the other ranks right after the timed
                # region: meet here, so that every rank's deadline starts when the leg does
                dist.barrier()
            timer = threading.Timer(args.train_deadline, on_timeout)
            timer.daemon = True
            timer.start()
            train = train_leg(args, cfg, dev, rank, world, backend, sd, voxelizer, log, "fp32")
            torch.cuda.empty_cache()
            train["bf16_frozen_nets"] = train_leg(args, cfg, dev, rank, world, backend, sd, voxelizer, log, "bf16")
            torch.cuda.empty_cache()
            timer.cancel()
        if rank == 0:
            out["train"] = train
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()

    if rank != 0:
        return finish(None)

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
    # `roofline`: kernel-level, the time-dominant hand-written kernel (k_conv3x3: 28 % of the device time of the timed window),
    # HIP events live + PMC traffic from profiles/; `roofline_spconv`: the dominant kernel of the sparse 3D branch
    roofline = conv_roofline(dev)
    roofline["scope"] = ("time-dominant hand-written HIP kernel (fused GroupNorm-SiLU-conv3x3 of the SD VAE / UNet ResnetBlocks); algorithmic "
                         "FLOP = 2*B*H*W*9*Cin*Cout per launch, one launch = one ResnetBlock convolution of the VAE at 128x128 on 20 views")
    roofline_spconv = spconv_roofline(dev)
    roofline_spconv["scope"] = ("dominant kernel of the sparse 3D branch, HBM roofline on the gather + scatter byte model of SURVEY 8d (pairs*(cin+cout)*e "
                                "+ 8*pairs + K*cin*cout*e per launch), one launch = one sparse-conv layer of MinkUNet34C block8 on the full S1 cloud; "
                                "headline object = the plain-bf16 form the bf16 configuration runs, `f32_accurate_form` = the split-operand form of the "
                                "fp32 configuration and of training; matrix rates are listed against the bf16 MFMA peak the instructions execute on")
    roofline_stage = {"bound": "mfma", "achieved": dense_tflop / (dense_ms * 1e-3), "peak": peak, "unit": "TFLOP/s",
                      "frac": dense_tflop / (dense_ms * 1e-3) / peak, "traffic": None, "views_per_forward": vb,
                      "scope": "dense 2D branch per view (SD VAE+UNet, projections, pixel+transformer decoder, mask-CLIP): HIP fused GroupNorm-SiLU-conv3x3 "
                               "for the ResnetBlocks, HIP implicit-GEMM k_gemm for the other convolutions and the linear layers (MIOpen only for the cin < 64 stems), HIP flash attention (VAE d=512 head: GEMM + HIP row softmax + GEMM) + HIP GroupNorm / LayerNorm / pointwise kernels, " + ("HIP graph replay" if not args.no_graph else "eager launches"),
                      "ms_per_view": dense_ms, "algorithmic_tflop_per_view": dense_tflop, "sparse3d_ms_per_view": sparse_ms}

    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
        from oracle import infer_oracle

        log("cpu baseline: one whole scene (5 views, post-processing, votes, fill) through the oracle")
        T = np.diag([50.0, 50.0, 50.0, 1.0])
        t1 = time.perf_counter()
        labels, _ = infer_oracle.scene_forward_cpu(cpu_model, cfg, scene, [T] * n_views)
        t_scene = time.perf_counter() - t1
        cpu_baseline = {"value": 1.0 / t_scene, "unit": "scenes/s", "cores": torch.get_num_threads(), "kind": "port",
                        "sample": f"one whole scene S1 ({scene.points.shape[0]} points, {n_views} views): oracle/model_oracle.py per view "
                                  f"(fp32, same weights) + oracle/infer_oracle.py (hole filling, ensembling, votes, nearest fill), "
                                  f"{t_scene:.1f} s, no extrapolation"}

    out = {
        "metric": "ScanNet scenes/sec (infer)", "value": value, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype + " (frozen SD/CLIP nets, decoder GEMMs) + f32 (deformable attention, statistics, logits); sparse 3D: " + ("bf16 rows, bf16 products, f32 accumulation"
                 if getattr(model, "sparse_dtype", None) == torch.bfloat16 else "f32 rows, products as bf16x3 split operands with f32 accumulation"), "data": "synthetic",
        "config": {"workload": f"ScanNet B15N4 inference, synthetic scenes S1 ({len(scenes)} distinct seeds, ~120k pts, 5 views 240x320->512x512), "
                               f"{vb} views ({G} scene{'s' if G > 1 else ''}) per forward, seeded random weights", "views_per_scene": n_views, "parallelism": f"dp{world} (scene level, no collective)",
                   "dead_compute": "as reference" if args.faithful_dead_compute else "pruned (SURVEY F7)",
                   "layout": "NCHW" if args.nchw else "channels-last (NHWC) frozen nets",
                   "schedule": "eager launches" if args.no_graph else "3 HIP graphs per forward (2 slots), next forward's front software-pipelined on side streams"},
        "roofline": roofline, "roofline_spconv": roofline_spconv, "roofline_spconv_window": spconv_window_roofline(dev, sd, voxelizer, vb),
        "roofline_sparse_network": sparse_network_roofline(dev, model), "roofline_dense_stage": roofline_stage, "roofline_kernels": kernel_rooflines(dev),
        "cpu_baseline": cpu_baseline, "latency_ms_single_scene": latency_ms, "fp32": fp32, "train": None,
    }
    finish(None)


if __name__ == "__main__":
    main()
