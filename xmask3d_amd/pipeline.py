"""Scene-level inference: per-view sample assembly, model forward, open-vocabulary ensembling, voting.

The counterpart of the reference's driver loop ``validate`` (/root/reference/run/infer.py:338-911) and of
the per-view assembly in ``ScannetLoaderFull.__getitem__`` (dataset/data_loader_infer.py:161-283), kept
on the device: no ``.cpu()`` round trips, no sklearn KD-tree (nearest-neighbour fill is a chunked
distance arg-min on the GPU), no per-mask Python loop (final masks are pixel-disjoint, so the sequential
update of run/infer.py:585-601 equals one gather).  Results follow the reference formulae:
  logits      = logit_scale * norm(f) @ norm(text).T                          (infer.py:556-558)
  ensembling  = p^r * p_open^(1-r) in log space, r = base_ratio / novel_ratio (infer.py:585-601)
  gating      = binary_pred ? base columns : novel columns                    (infer.py:603-612)
  vote        = scene_pred[visible, cls] += 1 ; unseen points <- nearest seen (infer.py:642-694)
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F

from . import me_compat as ME
from . import ops
from . import synthetic
from .voxelizer import Voxelizer

ROT_BOUND = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))


def default_voxelizer(voxel_size=0.02, device="cuda"):
    """dataset/point_loader.py:52-60,100-107"""
    return Voxelizer(voxel_size=voxel_size, clip_bound=None, use_augmentation=True, scale_augmentation_bound=(0.9, 1.1),
                     rotation_augmentation_bound=ROT_BOUND,
                     translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)), device=device)


def make_inference_model(cpu_model, device, dense_dtype=torch.bfloat16, channels_last=True, graphs=True):
    """The inference configuration bench.py measures (and tests/test_gpu_bench_parity.py checks against the oracle): a copy of
    `cpu_model` on `device` with the frozen SD / CLIP nets in `dense_dtype`, channels-last conv nets, bf16 copies of the
    head GEMM weights when the dense dtype is bf16, and the dense branch replayed as HIP graphs."""
    import copy

    model = copy.deepcopy(cpu_model).to(device).eval()
    model.set_dense_dtype(dense_dtype)
    if channels_last:
        model.set_channels_last(True)
    if dense_dtype == torch.bfloat16:
        model.cast_head_weights()
        if os.environ.get("XM3D_SPARSE", "bf16") != "f32":  # A/B switch: keep the f32-accurate (split-operand) sparse branch under the bf16 nets
            model.set_sparse_dtype(torch.bfloat16)
    if graphs:
        model.enable_dense_graph()
    return model


class SceneOnDevice:
    """A scene uploaded once: points (f64), colours, per-view visibility / pixel labels / images."""

    def __init__(self, scene: synthetic.Scene, device):
        self.device = device
        self.points = torch.from_numpy(scene.points).to(device)
        self.points_f32 = self.points.float().contiguous()  # the nearest-neighbour kernels work in f32
        self.colors = torch.from_numpy(scene.colors).to(device)
        self.n = scene.points.shape[0]
        self.views = []
        for v in range(len(scene.poses)):
            vis, rows, cols = synthetic.view_subset(scene, v)
            self.views.append(dict(
                vis=torch.from_numpy(vis).to(device),
                idx=torch.from_numpy(np.nonzero(vis)[0]).to(device),
                x=torch.from_numpy(rows).long().to(device), y=torch.from_numpy(cols).long().to(device),
                img=torch.from_numpy(scene.images[v]).permute(2, 0, 1)[None].contiguous().to(device),
                caption=scene.captions[v]))
        self.img_all = torch.cat([v["img"] for v in self.views])  # all views as one batch (stable storage: prefetch key)
        # all-views batch: concatenated per-point tables, built once (per-call torch.cat / fills are launches on the hot path)
        self.idx_all = torch.cat([v["idx"] for v in self.views])
        self.x_all = torch.cat([v["x"] for v in self.views])
        self.y_all = torch.cat([v["y"] for v in self.views])
        self.view_all = torch.cat([torch.full((v["idx"].shape[0],), i, dtype=torch.long, device=device) for i, v in enumerate(self.views)])


def build_view_batch(sd: SceneOnDevice, view: int, voxelizer: Voxelizer, matrix=None):
    """-> batch_input dict of XMASK3d.forward for one view (batch 1), everything on the device."""
    v = sd.views[view]
    pts = sd.points[v["idx"]].contiguous()
    grid, inds, inverse = voxelizer.voxelize_device(pts, matrix)
    coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=sd.device), grid], 1).contiguous()
    feats = (sd.colors[v["idx"]][inds] / 127.5 - 1.0).float().contiguous()
    ori = torch.cat([torch.zeros(pts.shape[0], 1, device=sd.device), pts.float()], 1)
    return {"sinput": ME.SparseTensor(feats, coords), "img": v["img"], "x_label": v["x"], "y_label": v["y"],
            "inds_reconstruct": inverse, "ori_coords": ori, "captions": (v["caption"],), "coords": coords,
            "label_2d": None, "labels_3d": None, "use_pure_3d": False}


def build_scene_batch(sd: SceneOnDevice, views, voxelizer: Voxelizer, matrices=None):
    """All (or some) views of a scene as ONE batch: views are independent until the vote, so they are collated like a
    DataLoader batch (batch index in column 0 of coords / ori_coords, inds_reconstruct offset per view)."""
    return build_group_batch([(sd, list(views))], voxelizer, None if matrices is None else [matrices])


def build_group_batch(groups, voxelizer: Voxelizer, matrices=None):
    """Views of one or several scenes as ONE batch.  groups: [(SceneOnDevice, [view, ...]), ...]; matrices: per group, a list
    of 4x4 voxelisation transforms per view (None: drawn from np.random in scene-by-scene, view-by-view order, as a
    sequential run would).  Batch entry b = position in the flattened (scene, view) list."""
    coords, feats, inv, ori, caps = [], [], [], [], []
    xs, ys, imgs, vids = [], [], [], []
    base, b = 0, 0
    offsets = [0]
    dev = groups[0][0].device
    vrows, row_base = [], [0]   # vote-table rows of the visible points: scene offset + point index (xm3d_scene_votes)
    for gi, (sd, views) in enumerate(groups):
        whole = list(views) == list(range(len(sd.views)))
        b0 = b
        idx = sd.idx_all if whole else torch.cat([sd.views[v]["idx"] for v in views])
        vrows.append(idx if row_base[-1] == 0 else idx + row_base[-1])
        row_base.append(row_base[-1] + sd.n)
        for vi, v in enumerate(views):
            vw = sd.views[v]
            pts = sd.points[vw["idx"]].contiguous()
            offsets.append(offsets[-1] + pts.shape[0])
            grid, inds, inverse = voxelizer.voxelize_device(pts, None if matrices is None or matrices[gi] is None else matrices[gi][vi])
            coords.append(torch.cat([torch.full((grid.shape[0], 1), b, dtype=torch.int32, device=dev), grid], 1))
            feats.append((sd.colors[vw["idx"]][inds] / 127.5 - 1.0).float())
            inv.append(inverse + base)
            base += grid.shape[0]
            ori.append(torch.cat([torch.full((pts.shape[0], 1), float(b), device=dev), pts.float()], 1))
            caps.append(vw["caption"])
            if not whole:
                xs.append(vw["x"]); ys.append(vw["y"]); imgs.append(vw["img"])
                vids.append(torch.full((pts.shape[0],), b, dtype=torch.long, device=dev))
            b += 1
        if whole:  # per-scene tables built once at upload
            xs.append(sd.x_all); ys.append(sd.y_all); imgs.append(sd.img_all)
            vids.append(sd.view_all if b0 == 0 else sd.view_all + b0)
    coords = torch.cat(coords).contiguous()
    one = len(xs) == 1
    return {"sinput": ME.SparseTensor(torch.cat(feats).contiguous(), coords), "img": imgs[0] if one else torch.cat(imgs),
            "x_label": xs[0] if one else torch.cat(xs), "y_label": ys[0] if one else torch.cat(ys),
            "inds_reconstruct": torch.cat(inv), "ori_coords": torch.cat(ori), "captions": tuple(caps),
            "coords": coords, "label_2d": None, "labels_3d": None, "use_pure_3d": False, "point_offsets": offsets,
            "point_view": vids[0] if one else torch.cat(vids),
            "vote_rows": vrows[0] if len(vrows) == 1 else torch.cat(vrows), "vote_row_base": row_base}


def build_train_batch(sd: SceneOnDevice, views, voxelizer: Voxelizer, seed=0, n_classes=15, ignore=(19, 20)):
    """A training batch of several views (one sample = one view, dataset/data_loader.py:85-316, collated as
    data_loader.py:319-357 / run/train.py:462-502): batch index in column 0, inds_reconstruct offset per sample, the random
    voxel-coordinate offset of run/train.py:481, synthetic 2D/3D labels."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    coords, feats, inv, ori, xs, ys, imgs, l3d, bl3d, l2d, caps = [], [], [], [], [], [], [], [], [], [], []
    base = 0
    for b, v in enumerate(views):
        vw = sd.views[v]
        pts = sd.points[vw["idx"]].contiguous()
        grid, inds, inverse = voxelizer.voxelize_device(pts)
        c = torch.cat([torch.full((grid.shape[0], 1), b, dtype=torch.int32, device=sd.device), grid], 1)
        coords.append(c)
        feats.append((sd.colors[vw["idx"]][inds] / 127.5 - 1.0).float())
        inv.append(inverse + base)
        base += grid.shape[0]
        ori.append(torch.cat([torch.full((pts.shape[0], 1), float(b), device=sd.device), pts.float()], 1))
        xs.append(vw["x"]); ys.append(vw["y"]); imgs.append(vw["img"]); caps.append(vw["caption"])
        lab = torch.randint(0, n_classes + 1, (pts.shape[0],), generator=g).to(sd.device)  # n_classes = ignore label
        l3d.append(lab)
        bl3d.append(torch.where(lab == n_classes, torch.full_like(lab, ignore[0]), (lab % 4 != 0).long()).float())
        blocks = torch.randint(0, n_classes + 1, (8, 8), generator=g)
        l2d.append(blocks.repeat_interleave(64, 0).repeat_interleave(64, 1).to(sd.device))
    coords = torch.cat(coords)
    coords[:, 1:] += torch.randint(0, 100, (1, 3), generator=g).int().to(sd.device)  # run/train.py:481
    return {"sinput": ME.SparseTensor(torch.cat(feats).contiguous(), coords.contiguous()), "img": torch.cat(imgs),
            "x_label": torch.cat(xs), "y_label": torch.cat(ys), "inds_reconstruct": torch.cat(inv), "ori_coords": torch.cat(ori),
            "captions": tuple(caps), "labels_3d": torch.cat(l3d), "binary_label_3d": torch.cat(bl3d), "label_2d": torch.stack(l2d),
            "binary_label_2d": None, "coords": coords}


def nearest_index(query: torch.Tensor, ref: torch.Tensor, ref_valid=None):
    """index into ref of the nearest (valid) reference point for every query point (exact; xm3d_nearest_index)"""
    return ops.nearest_index(query.float().contiguous(), ref.float().contiguous(), ref_valid)


def nearest_valid_fill(xyz: torch.Tensor, valid: torch.Tensor, method: str = "sorted"):
    """For every point: its own index if `valid`, else the index of the nearest valid point (exact, lowest index on ties;
    identity when nothing is valid).  No host synchronisation either way.
    method "sorted" (default): xm3d_nearest_valid_fill_sorted - queries and valid points sorted along a Morton curve, waves of
    64 neighbouring queries scan only the 64-point tiles they cannot exclude: 0.37 ms on the S1 vote fill (two thirds of the
    room never seen), 0.5-0.9 ms on random masks;
    method "octree": xm3d_nearest_valid_fill - Morton-ordered cells, depth-first descent per query: 0.15-0.2 ms when the holes
    are small against the cloud, 1.7 ms on the S1 vote fill;
    method "scan": every query scans every valid point through LDS (xm3d_nearest_index on device-partitioned rows): 1.06 ms flat.
    All three return the same indices (tests/test_gpu_nearest_grid.py)."""
    if method in ("octree", "sorted"):
        return ops.nearest_valid_fill(xyz.float().contiguous(), valid, method=method)
    n = xyz.shape[0]
    xyz = xyz.float()
    q_order = torch.argsort(valid.to(torch.uint8), stable=True)          # invalid points first = the queries
    r_order = torch.argsort((~valid).to(torch.uint8), stable=True)       # valid points first = the references
    n_valid = valid.sum()
    counts = torch.stack([n - n_valid, n_valid]).to(torch.int64)
    nn = ops.nearest_index(xyz[q_order].contiguous(), xyz[r_order].contiguous(), None, counts)
    fill = torch.arange(n, device=xyz.device)
    live = (torch.arange(n, device=xyz.device) < (n - n_valid)) & (n_valid > 0)  # nothing valid anywhere: identity
    # scatter the answers of the live queries back; every other point keeps its own index
    return fill.scatter(0, q_order, torch.where(live, r_order[nn.clamp_(0, n - 1)], q_order))


def nearest_valid_fill_segmented(xyz, valid, seg, n_seg, max_seg_points):
    """nearest_valid_fill inside every segment (view) of a concatenated point set, one launch: xyz (n,3), valid (n,) bool,
    seg (n,) long segment id per point, max_seg_points = host-side bound of a segment's size.  Same result as calling
    nearest_valid_fill per segment (stable partition -> lowest original index wins ties); no host synchronisation."""
    n = xyz.shape[0]
    key = seg * 2 + valid.long()                      # per segment: invalid points (queries) first, valid ones (references) after
    order = torch.argsort(key, stable=True)
    cnt = torch.zeros(2 * n_seg, dtype=torch.int64, device=xyz.device).index_add_(0, key, torch.ones_like(key))
    off = torch.cumsum(cnt, 0) - cnt
    desc = torch.stack([off[0::2], cnt[0::2], off[1::2], cnt[1::2]], 1).contiguous()
    out = torch.arange(n, device=xyz.device)          # identity for reference points and for segments without references
    ops.nearest_index_segmented(xyz.float()[order].contiguous(), desc, max_seg_points, out)
    return torch.empty_like(out).scatter_(0, order, order[out])


_CONSTS = {}


def _class_consts(cfg, ncols, device):
    """Column masks of the base / novel classes and the base-overlap vector, built once per device: creating them per call
    means a pageable host->device copy, which blocks the host until the stream has drained (the host could then never run
    ahead of the dense graph)."""
    key = (id(cfg), ncols, str(device))
    if key not in _CONSTS:
        cs = cfg.category_split
        base, novel, allc = list(cs["base_category"]), list(cs["novel_category"]), list(cs["all_category"])
        bm, nm = torch.zeros(ncols, dtype=torch.bool), torch.zeros(ncols, dtype=torch.bool)
        bm[base] = True
        nm[novel] = True
        overlap = torch.tensor([float(c in base) for c in allc], dtype=torch.float32)
        _CONSTS[key] = (bm.to(device), nm.to(device), overlap.to(device))
    return _CONSTS[key]


def _gate(logits, binary_pred, base_mask, novel_mask):
    """base-predicted points may only take base classes, the others only novel ones (infer.py:489-507)"""
    return torch.where(binary_pred.bool(), logits.masked_fill(novel_mask, -1e10), logits.masked_fill(base_mask, -1e10))


def postprocess_view(cfg, outputs, batch, with_ablations=True, s=0):
    """-> class id per visible point of batch entry `s` for the fused / 2D-only / 3D-only predictions.  Branch-free (no
    host synchronisation): an empty mask set or a fully covered view simply selects nothing."""
    text = F.normalize(outputs["text_embed"], dim=-1)
    base, novel, overlap = _class_consts(cfg, text.shape[0], text.device)
    scale = outputs["logit_scale"]
    offsets = batch.get("point_offsets")
    sel = slice(offsets[s], offsets[s + 1]) if offsets is not None else (batch["ori_coords"][:, 0] == s)
    binary_pred = outputs["binary_pred"][sel]
    fused = F.normalize(outputs["fused_pred_feature"][s], dim=-1)
    probs = (scale * (fused @ text.t())).softmax(dim=-1)
    open_emb = outputs["final_pred_open_embedding"][s]
    masks = outputs["final_mask_3d"][s]
    if masks.shape[0] > 0:
        open_p = (scale * (F.normalize(open_emb, dim=-1) @ text.t())).softmax(dim=-1)
        covered = masks.any(0)
        q = masks.to(torch.uint8).argmax(0)  # masks are pixel-disjoint: at most one per point
        po = open_p[q]
        b = (probs ** cfg.base_ratio * po ** (1 - cfg.base_ratio)).log() * overlap
        n = (probs ** cfg.novel_ratio * po ** (1 - cfg.novel_ratio)).log() * (1 - overlap)
        probs = torch.where(covered[:, None], b + n, probs)
    pred = _gate(probs, binary_pred, base, novel).argmax(1)
    if not with_ablations:
        return pred, None, None
    f2d = outputs["2d_pred_feature"][s]
    empty = f2d.sum(1) == 0
    xyz = batch["ori_coords"][sel][:, 1:]
    f2d = f2d[nearest_valid_fill(xyz, ~empty)]  # points without a 2D feature take the nearest covered point's (infer.py:523-553)
    pred2d = _gate(scale * (F.normalize(f2d, dim=-1) @ text.t()), binary_pred, base, novel).argmax(1)
    f3d = F.normalize(outputs["pure3d_pred_feature"][s], dim=-1)
    pred3d = _gate(scale * (f3d @ text.t()), binary_pred, base, novel).argmax(1)
    return pred, pred2d, pred3d


def postprocess_scene(cfg, outputs, batch, with_ablations=True):
    """postprocess_view for all batch entries at once, on the concatenated tensors of XMASK3d.fuse_eval_batched:
    -> class id per visible point (all entries back to back) for the fused / 2D-only / 3D-only predictions."""
    text = F.normalize(outputs["text_embed"], dim=-1)
    base, novel, overlap = _class_consts(cfg, text.shape[0], text.device)
    scale = outputs["logit_scale"]
    vid = batch["point_view"]
    offsets = batch["point_offsets"]
    binary_pred = outputs["binary_pred"]
    open_p = (scale * (F.normalize(outputs["open_embedding_all"], dim=-1) @ text.t())).softmax(dim=-1)   # (B, Q, C)
    masks = outputs["mask_3d_cat"]                     # (Np, Q), pixel-disjoint: at most one query per point
    fast = (os.environ.get("XM3D_POINT_CLASS", "hip") != "library" and ops.point_class_supported(outputs["fused_cat"], text)
            and masks.dtype == torch.bool and masks.is_contiguous() and binary_pred.dtype == torch.int64 and torch.is_tensor(scale))
    if fast:
        # xm3d_point_class: each (Np, 768) feature table is read once, the (Np, C) logits / probabilities never exist
        sc = scale.detach().float().reshape(1)
        pred = ops.point_class(outputs["fused_cat"], text, binary_pred, base, novel,
                               ensemble=(sc, masks, vid, open_p.float(), overlap, cfg.base_ratio, cfg.novel_ratio))
        if not with_ablations:
            return pred, None, None
        f2d = outputs["feat2d_cat"]
        n_seg = len(offsets) - 1
        fill = nearest_valid_fill_segmented(batch["ori_coords"][:, 1:], f2d.sum(1) != 0, vid, n_seg,
                                            max(offsets[i + 1] - offsets[i] for i in range(n_seg)))
        pred2d = ops.point_class(f2d, text, binary_pred, base, novel, row_index=fill)
        pred3d = ops.point_class(outputs["pure3d_cat"], text, binary_pred, base, novel)
        return pred, pred2d, pred3d
    probs = (scale * (F.normalize(outputs["fused_cat"], dim=-1) @ text.t())).softmax(dim=-1)
    covered = masks.any(1)
    po = open_p[vid, masks.to(torch.uint8).argmax(1)]
    b = (probs ** cfg.base_ratio * po ** (1 - cfg.base_ratio)).log() * overlap
    n = (probs ** cfg.novel_ratio * po ** (1 - cfg.novel_ratio)).log() * (1 - overlap)
    probs = torch.where(covered[:, None], b + n, probs)
    pred = _gate(probs, binary_pred, base, novel).argmax(1)
    if not with_ablations:
        return pred, None, None
    f2d = outputs["feat2d_cat"]
    empty = f2d.sum(1) == 0
    n_seg = len(offsets) - 1
    fill = nearest_valid_fill_segmented(batch["ori_coords"][:, 1:], ~empty, vid, n_seg,
                                        max(offsets[i + 1] - offsets[i] for i in range(n_seg)))
    # The 2D-only and 3D-only labels are arg-maxima of scale * cos(feature, text) over the gated classes: the positive per-point
    # factor scale / |feature| does not move an arg-max, so the (Np, 768) features are neither normalised (a read + write pass
    # each) nor gathered - the (Np, C) products are, after the GEMM.  Points without a 2D feature take the nearest covered
    # point's of the same view (infer.py:523-553): its row of the product.
    pred2d = _gate((f2d @ text.t())[fill], binary_pred, base, novel).argmax(1)
    pred3d = _gate(outputs["pure3d_cat"] @ text.t(), binary_pred, base, novel).argmax(1)
    return pred, pred2d, pred3d


@torch.no_grad()
def infer_scene(model, sd: SceneOnDevice, cfg, voxelizer=None, matrices=None, with_ablations=True, views_per_batch=None,
                next_scene: SceneOnDevice | None = None, next_matrices=None):
    """All views of one scene -> per-point class votes -> arg-max, unseen points take the nearest seen point's label.
    views_per_batch: how many views go through the model together (default: all of them, one forward per scene; 1 =
    the reference's batch-1 loop, run/infer.py:428-482).  Per-view results do not depend on the grouping.
    matrices: optional list of 4x4 voxelisation transforms (otherwise drawn from np.random like the reference).
    next_scene (+ next_matrices): the scene that will be inferred next.  Its shape-dynamic front (voxelisation, sparse 3D
    nets) is issued on a side stream and its VAE-encoder graph on another once this scene's work is enqueued (the host runs
    ~40 ms ahead of the device there); they start when this scene's convolution-bound graph B is done and overlap its
    latency-bound graph C (software pipelining across scenes); the next call picks them up.
    (Measured and rejected: fusion/votes/fill on a stream of their own beside the next scene's graph B - the small kernels
    double the duration of the convolution kernels they share the device with, tools/timeline_events.py.)"""
    voxelizer = voxelizer or default_voxelizer(cfg.voxel_size, sd.device)
    ncls = len(cfg.category_split["base_category"]) + len(cfg.category_split["novel_category"])
    votes = seen = labels = None

    def tables():
        return ([torch.zeros((sd.n, ncls), dtype=torch.int32, device=sd.device) for _ in range(3 if with_ablations else 1)],
                torch.zeros(sd.n, dtype=torch.bool, device=sd.device))

    nv = len(sd.views)
    step = views_per_batch or nv
    staged = step >= nv and sd.device.type == "cuda" and getattr(model, "_dense_graphs", None) is not None and not model.training
    pending, model._next_front = getattr(model, "_next_front", None), None
    for v0 in range(0, nv, step):
        views = list(range(v0, min(v0 + step, nv)))
        if staged and pending is not None and pending["scene"] is sd and pending["matrices"] is matrices:
            batch, front = pending["batch"], pending["front"]  # issued during the previous call
        else:
            batch = build_scene_batch(sd, views, voxelizer, None if matrices is None else [matrices[v] for v in views])
            batch["compact_outputs"] = False  # keep all Q mask rows (dropped ones all-False): no host sync in the fusion stage
            front = None
        if staged:
            front = front or model.eval_front(batch)
            outputs = model.eval_dense(batch, front)
            outputs = model.eval_fuse(batch, front, outputs)
            model.mark("F1")  # fusion done (timeline tracing only)
        else:
            _, outputs = model(batch)
        if "fused_cat" in outputs:  # batched fusion ran: post-process and vote for all views of the batch in one go
            preds = postprocess_scene(cfg, outputs, batch, with_ablations)
            if len(views) == nv and "vote_rows" in batch:  # the whole scene in one forward: votes + labels in two launches
                label, seen = ops.scene_votes(batch["vote_rows"], torch.stack([p for p in preds if p is not None]), sd.n, ncls)
                labels = [label[k] for k in range(label.shape[0])]
                continue
            if votes is None:
                votes, seen = tables()
            idx = sd.idx_all if len(views) == nv else torch.cat([sd.views[v]["idx"] for v in views])
            for vt, p in zip(votes, preds):
                if p is not None:
                    vt.index_put_((idx, p), torch.ones_like(p, dtype=torch.int32), accumulate=True)
            seen.index_fill_(0, idx, True)
            continue
        if votes is None:
            votes, seen = tables()
        for s, v in enumerate(views):
            preds = postprocess_view(cfg, outputs, batch, with_ablations, s)
            idx = sd.views[v]["idx"]
            for vt, p in zip(votes, preds):
                if p is not None:
                    vt.index_put_((idx, p), torch.ones_like(p, dtype=torch.int32), accumulate=True)
            seen.index_fill_(0, idx, True)  # (`seen[idx] = True` uploads the scalar: a host-blocking copy)
    if hasattr(model, "mark"):
        model.mark("V1")  # per-view post-processing and votes done
    fill = nearest_valid_fill(sd.points_f32, seen)  # unseen points take the label of the nearest seen point (infer.py:682-694)
    result = [lb[fill] for lb in labels] if labels is not None else [vt.argmax(1)[fill] for vt in votes]
    if hasattr(model, "mark"):
        model.mark("P1")  # end of this scene's post-processing (tools/timeline_events.py)
    if staged and next_scene is not None:
        # everything of this scene is enqueued behind its dense graph; the host is free to issue the next scene's front
        fs = model.front_stream()
        with torch.cuda.stream(fs):
            nbatch = build_scene_batch(next_scene, list(range(len(next_scene.views))), voxelizer, next_matrices)
            nbatch["compact_outputs"] = False
        model._next_front = dict(scene=next_scene, matrices=next_matrices, batch=nbatch, front=model.eval_front(nbatch, stream=fs))
    return result


# XM3D_MAIN_STREAMS=2 (A/B switch): consecutive forwards alternate between two main streams, so that the convolution-bound graph B of forward
# i + 1 can start as soon as its front is done - beside the latency-bound graph C, fusion and post-processing of forward i - instead of behind them
_MAIN_STREAMS = int(os.environ.get("XM3D_MAIN_STREAMS", "1"))


def _main_stream(model, caller):
    st = model.__dict__.setdefault("_main_streams", None)
    if st is None:
        st = model.__dict__["_main_streams"] = [torch.cuda.Stream(), torch.cuda.Stream()]
        for x in st:
            x.wait_stream(caller)  # once: everything the caller prepared before the first forward (weights, scene data)
        model.__dict__["_main_turn"] = 0
    model.__dict__["_main_turn"] ^= 1
    return st[model.__dict__["_main_turn"]]


@torch.no_grad()
def infer_scenes(model, sds, cfg, voxelizer=None, matrices=None, with_ablations=True, next_scenes=None, next_matrices=None):
    """infer_scene for a GROUP of scenes whose views all go through ONE forward (e.g. 2 scenes x 5 views = batch 10): the
    views are independent until the vote, and the latency-bound stages (decoders, post-processing) cost almost the same for
    10 views as for 5 - 36.5 instead of 46.5 ms of dense-branch time per scene (tools/prof_stages.py).  Per-view results
    do not depend on the grouping.  matrices: per scene, a list of 4x4 transforms per view (or None).  next_scenes
    (+ next_matrices): the group inferred next; its front is issued on side streams like infer_scene(next_scene=...).
    -> list (per scene) of [fused, 2D-only, 3D-only] per-point predictions."""
    sds = list(sds)
    dev = sds[0].device
    if dev.type != "cuda" or getattr(model, "_dense_graphs", None) is None or model.training:
        return [infer_scene(model, sd, cfg, voxelizer, None if matrices is None else matrices[j], with_ablations)
                for j, sd in enumerate(sds)]
    voxelizer = voxelizer or default_voxelizer(cfg.voxel_size, dev)
    ncls = len(cfg.category_split["base_category"]) + len(cfg.category_split["novel_category"])

    def same(a, b):
        return a is not None and b is not None and len(a) == len(b) and all(x is y for x, y in zip(a, b))

    pending, model._next_front = getattr(model, "_next_front", None), None
    caller = torch.cuda.current_stream()
    ms = _main_stream(model, caller) if _MAIN_STREAMS == 2 else caller
    with torch.cuda.stream(ms):
        if pending is not None and same(pending.get("scenes"), sds) and pending["matrices"] is matrices:
            batch, front = pending["batch"], pending["front"]
        else:
            batch = build_group_batch([(sd, list(range(len(sd.views)))) for sd in sds], voxelizer, matrices)
            batch["compact_outputs"] = False
            front = model.eval_front(batch)
        dense = model.eval_dense(batch, front)
        prev = getattr(model, "_main_done", None)
        if ms is not caller and prev is not None:
            ms.wait_event(prev)  # the eager stages of consecutive forwards share cached workspaces: fusion i + 1 starts after post-processing i
        outputs = model.eval_fuse(batch, front, dense)
        model.mark("F1")
        preds = postprocess_scene(cfg, outputs, batch, with_ablations)
        # votes of all views of all scenes, first-max label per scene point and the "seen" flags: two launches (xm3d_scene_votes)
        rb = batch["vote_row_base"]
        label, seen = ops.scene_votes(batch["vote_rows"], torch.stack([p for p in preds if p is not None]), rb[-1], ncls)
        results = []
        for j, sd in enumerate(sds):
            fill = nearest_valid_fill(sd.points_f32, seen[rb[j]:rb[j + 1]])  # unseen points take the nearest seen point's label
            results.append([label[k, rb[j]:rb[j + 1]][fill] for k in range(label.shape[0])] + [None] * (3 - label.shape[0]))
        model.mark("P1")
        if ms is not caller:
            model._main_done = torch.cuda.Event()
            model._main_done.record(ms)
    if ms is not caller:  # the caller reads the labels on its own stream
        caller.wait_event(model._main_done)
        for r in results:
            for t in r:
                if t is not None:
                    t.record_stream(caller)
    if next_scenes is not None:
        nxt = list(next_scenes)
        fs = model.front_stream()
        with torch.cuda.stream(fs):
            nbatch = build_group_batch([(sd, list(range(len(sd.views)))) for sd in nxt], voxelizer, next_matrices)
            nbatch["compact_outputs"] = False
        model._next_front = dict(scenes=nxt, scene=None, matrices=next_matrices, batch=nbatch, front=model.eval_front(nbatch, stream=fs))
    return results
