"""Train / infer drivers: the counterparts of /root/reference/run/train.py (main_worker :126-393, train_net :403-878)
and run/infer.py (validate :338-911) on synthetic ScanNet-shaped scenes (no dataset offline).

Kept from the reference: one process per GPU, ``dist_backend`` from the yaml ("nccl" == RCCL on ROCm), DDP (without the
reference's ``find_unused_parameters=True``: every trainable parameter takes part in every iteration here, so the extra
autograd-graph traversal per iteration buys nothing), MinkowskiSyncBatchNorm when the per-GPU batch is < 4 (train.py:185-187), AdamW with the two
parameter groups of train.py:152-169 (3D nets at lr_3d, everything trainable else at lr_others, frozen SD/CLIP skipped),
cosine / poly LR per iteration (train.py:575-586), checkpoint ``model/model_last.pth.tar`` every epoch (train.py:354-390),
metrics all-reduced as SUMs (train.py:640-652, infer.py:717-726).
Changed on purpose: no ``torch.cuda.empty_cache()`` per iteration (train.py:842), the twelve metric all-reduces are one
coalesced tensor, scene post-processing stays on the device (pipeline.py).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import checkpoint as ckpt_io
from . import me_compat as ME
from . import metrics, pipeline, synthetic
from .xmask3d import XMASK3d


def setup_distributed(cfg):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("XM3D_DIST_BACKEND", cfg.dist_backend)  # "gloo": rehearse N ranks on fewer GPUs
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    return rank, world, dev


def build_optimizer(model, cfg):
    core = model.module if hasattr(model, "module") else model
    g3d, rest = [], []
    for name, p in core.named_parameters():
        if not p.requires_grad or "ldm_extractor.ldm" in name or "clip.clip" in name:
            continue
        (g3d if ("pc_decoder" in name or "pc_binary_head" in name) else rest).append(p)
    fused = all(p.is_cuda for p in g3d + rest)  # one multi-tensor kernel per group instead of ~10 foreach launches
    return torch.optim.AdamW([{"params": g3d, "lr": cfg.lr_3d}, {"params": rest, "lr": cfg.lr_others}], fused=fused)


def set_contra_schedule(core, cfg, epoch):
    """run/train.py:292-307: the mask-level 3D contrastive loss is switched off (weight 0, not even computed) before
    ``cfg.start_contra`` and enters the objective with ``cfg.loss_weight.loss_3d_contra`` from that epoch on."""
    if not bool(getattr(cfg, "mask_contra_3d", False)):
        return
    crit = core.criterion
    if epoch < int(getattr(cfg, "start_contra", 0)):
        crit.weight_dict["loss_3d_contra"] = 0
        crit.mask_contra_3d = False
    else:
        crit.weight_dict["loss_3d_contra"] = cfg.loss_weight["loss_3d_contra"]
        crit.mask_contra_3d = True


def synthetic_labels(scene, n_classes=19, seed=0):
    """deterministic per-point ground truth for the synthetic room: class by height band x quadrant"""
    p = scene.points
    band = np.clip((p[:, 2] / 2.6 * 4).astype(int), 0, 3)
    quad = (p[:, 0] > 3.0).astype(int) * 2 + (p[:, 1] > 2.5).astype(int)
    return torch.from_numpy(((band * 4 + quad + seed) % n_classes).astype(np.int64))


def _log_pretrained(model, log):
    """say which frozen-net files the model was built from (checkpoint.load_pretrained raises on a refused file / partial set)"""
    rep = getattr(model, "pretrained_report", None)
    if rep is None:
        log("pretrained: no SD / CLIP / tokenizer files found - seeded random frozen nets, stand-in tokenizer (synthetic runs only)")
        return
    log(f"pretrained: SD {rep['sd']}, CLIP {rep['clip']}, tokenizer {rep['tokenizer']}, uncond_inputs {rep['uncond']}")
    for pr in rep.get("problems", []):
        log("pretrained: WARNING " + pr)


def train(cfg, epochs=1, iters_per_epoch=4, views_per_gpu=2, save_path=None, resume=None, log=print):
    rank, world, dev = setup_distributed(cfg)
    torch.manual_seed(cfg.manual_seed)
    np.random.seed(cfg.manual_seed + rank)
    model = XMASK3d(cfg).to(dev)
    _log_pretrained(model, log)
    if dev.type == "cuda" and bool(getattr(cfg, "train_unet_graph", True)):
        # static-shape stages replay as HIP graphs: frozen UNet forward + backward, frozen VAE stages, the trainable dense heads
        # (XMASK3d.enable_train_graphs; parameter gradients still arrive through AccumulateGrad, so DDP is unchanged)
        model.enable_train_graphs()
    if world > 1:
        if views_per_gpu < 4:
            ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)
            torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=False)
    opt = build_optimizer(model, cfg)
    start_epoch, best = 0, 0.0
    if resume:
        info = ckpt_io.load_checkpoint(resume, model, opt, map_location=dev)
        start_epoch, best = info["start_epoch"], info["best_iou"]
    sd = pipeline.SceneOnDevice(synthetic.scene_s1(), dev)
    vox = pipeline.default_voxelizer(cfg.voxel_size, dev)
    max_iter = epochs * iters_per_epoch
    core = model.module if world > 1 else model
    K = cfg.classes
    for epoch in range(start_epoch, epochs):
        model.train()
        set_contra_schedule(core, cfg, epoch)
        meter = metrics.AverageMeter()
        t0 = time.perf_counter()
        for i in range(iters_per_epoch):
            it = epoch * iters_per_epoch + i
            sched = metrics.cosine_learning_rate if cfg.learning_rate_type == "cosine" else metrics.poly_learning_rate
            opt.param_groups[0]["lr"] = sched(cfg.lr_3d, it, max_iter)
            opt.param_groups[1]["lr"] = sched(cfg.lr_others, it, max_iter)
            views = [(it * views_per_gpu * world + rank * views_per_gpu + j) % len(sd.views) for j in range(views_per_gpu)]
            batch = pipeline.build_train_batch(sd, views, vox, seed=cfg.manual_seed + it * world + rank)
            losses, outputs = model(batch)
            loss = sum(losses.values())
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            with torch.no_grad():  # train-mIoU bookkeeping: one coalesced all-reduce instead of twelve (train.py:624-652)
                text = torch.nn.functional.normalize(torch.cat([outputs["text_embed"], outputs["null_embed"]]), dim=-1)
                feat = torch.nn.functional.normalize(torch.cat(outputs["fused_pred_feature"]), dim=-1)
                pred = (outputs["logit_scale"] * feat @ text.t()).argmax(1)
                stats = torch.stack(metrics.intersection_and_union(pred, batch["labels_3d"], K, (cfg.ignore_label,)))
                red = torch.cat([stats.reshape(-1), loss.detach().reshape(1)])
                if world > 1:
                    dist.all_reduce(red)
                meter.update(float(red[-1]) / world)
            if rank == 0:
                log(f"epoch {epoch} iter {i} loss {meter.val:.4f} lr {opt.param_groups[0]['lr']:.2e}/{opt.param_groups[1]['lr']:.2e}")
        if rank == 0 and save_path:
            ckpt_io.save_checkpoint(os.path.join(save_path, "model", "model_last.pth.tar"), core, opt, epoch + 1, best)
        if rank == 0:
            log(f"epoch {epoch}: {iters_per_epoch / (time.perf_counter() - t0):.2f} iters/s, mean loss {meter.avg:.4f}")
    return model


def scannet_scene_list(cfg, split="val"):
    """Scene names of the reference's on-disk validation set (<data_root>/<split>/*_vh_clean_2.pth, sorted like
    ScannetLoaderFull: data_loader_infer.py:60-75) when both data directories exist, else None (synthetic scenes)."""
    import glob

    d3, d2 = str(getattr(cfg, "data_root", "") or ""), str(getattr(cfg, "data_root_2d", "") or "")
    if not (os.path.isdir(os.path.join(d3, split)) and os.path.isdir(d2)):
        return None
    names = sorted(os.path.basename(p)[:-len("_vh_clean_2.pth")] for p in glob.glob(os.path.join(d3, split, "*_vh_clean_2.pth")))
    return [n for n in names if os.path.isdir(os.path.join(d2, n))] or None


@torch.no_grad()
def infer(cfg, model=None, scenes=1, resume=None, dense_dtype=torch.bfloat16, log=print):
    """-> dict of open-vocabulary scores for the fused / 2D / 3D predictions (infer.py:696-911 bookkeeping).
    When cfg.data_root / cfg.data_root_2d hold the reference's ScanNet layout (scannet.py), the validation scenes found there
    are read (point cloud, frames, depth, poses, captions, the reference's frame filter) and scored against their labels;
    `scenes` then caps how many (None / 0 = all).  Otherwise `scenes` synthetic rooms with synthetic labels."""
    rank, world, dev = setup_distributed(cfg)
    own = model is None
    if own:
        torch.manual_seed(cfg.manual_seed)
        model = XMASK3d(cfg).to(dev)
        _log_pretrained(model, log)
        if resume:
            ckpt_io.load_checkpoint(resume, model, eval=True, map_location=dev)
    model = model.module if hasattr(model, "module") else model
    restore = None if own else (model.dense_dtype, model.channels_last, model.training, model._dense_graphs is not None)
    model.eval().set_dense_dtype(dense_dtype)
    if dense_dtype == torch.bfloat16 and dev.type == "cuda":
        model.set_channels_last(True)   # NHWC convolutions + folded GroupNorm / residual kernels
        if own:                         # a model handed in may be trained further: keep its fp32 head weights
            model.cast_head_weights()
    model.enable_dense_graph()
    import gc

    gc.collect()
    gc.freeze()  # long-lived model / graph objects out of the cyclic GC's way (a gen-2 pass otherwise stalls the host ~50 ms)
    K = cfg.test_classes
    names = ("fused", "2d", "3d")
    acc = torch.zeros(3, 3, K, device=dev)
    real = scannet_scene_list(cfg)
    if real is not None:
        if scenes:
            real = real[:scenes]
        scenes = len(real)
        if rank == 0:
            log(f"infer: {scenes} ScanNet scene(s) from {cfg.data_root} / {cfg.data_root_2d}")
    mine = list(range(rank, scenes, world))  # DistributedSampler(shuffle=False) partition
    G = max(1, int(getattr(cfg, "scenes_per_forward", 5)))  # scenes whose views share one forward (pipeline.infer_scenes)
    chunks = [mine[i:i + G] for i in range(0, len(mine), G)]

    vox = pipeline.default_voxelizer(cfg.voxel_size, dev)

    def load_real(s):
        from . import scannet

        sc, frames = scannet.load_scene(cfg.data_root, cfg.data_root_2d, real[s], split="val", caption_path=getattr(cfg, "caption_path", None),
                                        ignore_label=cfg.category_split["ignore_category"][-1], device=dev,
                                        val_keep=int(getattr(cfg, "val_keep", 10000000)), ignore_categories=cfg.category_split["ignore_category"])
        if not frames:
            raise RuntimeError(f"scene {real[s]}: no frame passes the visibility filter (data_loader_infer.py:199-209)")
        return sc

    def upload(chunk):
        scs = [load_real(s) if real is not None else synthetic.scene_s1(seed=cfg.manual_seed + s) for s in chunk]
        mats = []
        for s, sc in zip(chunk, scs):  # the augmentation draws of a scene depend on its index alone (not on grouping / prefetch order)
            np.random.seed(cfg.manual_seed + s)
            mats.append([vox.rigid_matrix()[0] for _ in sc.poses])
        return scs, [pipeline.SceneOnDevice(sc, dev) for sc in scs], mats

    nxt = upload(chunks[0]) if chunks else None
    for ci, chunk in enumerate(chunks):
        scs, sds, mats = nxt
        nxt = upload(chunks[ci + 1]) if ci + 1 < len(chunks) else None  # resident before this chunk's forward is issued
        if real is not None:
            # real scenes have tens to hundreds of frames each: a fixed number of views per forward (one graph shape), scene by scene
            vpf = max(1, int(getattr(cfg, "views_per_forward", 20)))
            results = [pipeline.infer_scene(model, sd_, cfg, vox, m_, views_per_batch=None if len(sd_.views) <= vpf else vpf)
                       for sd_, m_ in zip(sds, mats)]
        else:
            results = pipeline.infer_scenes(model, sds, cfg, vox, mats, next_scenes=None if nxt is None else nxt[1],
                                            next_matrices=None if nxt is None else nxt[2])
        for s, scene, preds in zip(chunk, scs, results):
            gt = (torch.from_numpy(scene.labels) if real is not None else synthetic_labels(scene, K, s)).to(dev)
            for j, p in enumerate(preds):
                acc[j] += torch.stack(metrics.intersection_and_union(p, gt, K, tuple(cfg.test_ignore_label)))
    if world > 1:
        dist.all_reduce(acc)  # nine SUM all-reduces of the reference (infer.py:717-726) as one
    if restore is not None:  # a borrowed model goes back the way it came (a training run continues in its own precision)
        model.set_dense_dtype(restore[0])
        model.set_channels_last(restore[1])
        model.train(restore[2])
        if not restore[3]:
            model.enable_dense_graph(False)
    if dev.type == "cuda":
        # sticky device flag (coordinates outside the packable range; an activation beyond the half range of a split operand in the fp32
        # configuration): scores computed past it are not to be trusted - fail loudly instead of reporting them
        from ._lib import check, lib

        check(lib().xm3d_check_flag(), "inference (device range flag)")
    cs = cfg.category_split
    out = {n: metrics.open_vocab_scores(acc[j, 0], acc[j, 1], cs["base_category"], cs["novel_category"]) for j, n in enumerate(names)}
    if rank == 0:
        for n, v in out.items():
            log(f"{n}: hIoU {v['hIoU']:.4f} mIoU_base {v['mIoU_base']:.4f} mIoU_novel {v['mIoU_novel']:.4f}")
    return out
