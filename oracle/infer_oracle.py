"""CPU oracle for the driver's per-view post-processing and scene voting.

TEST INFRASTRUCTURE ONLY (tests/, bench.py cpu_baseline) - never imported by ``xmask3d_amd``.

Loop-form restatement of /root/reference/run/infer.py:484-694 in plain torch-CPU / numpy: per view
  * hole filling of the 2D feature with the nearest covered point (:523-553; sklearn KDTree there, scipy cKDTree here:
    both exact 1-NN; ties resolve to the lowest index in the GPU path, so the tests compare labels, not indices, on ties)
  * logits = logit_scale * norm(f) @ norm(text).T, softmax (:556-571)
  * the sequential per-mask geometric ensembling with the mask-CLIP logits (:585-601), base_ratio / novel_ratio
  * base / novel gating by the binary head (:603-640) and arg-max
then per scene: votes scene_pred[visible, cls] += 1 (:642-647), arg-max, unseen points <- nearest seen (:682-694).
PARITY UNPINNED in the strict sense (the reference driver cannot run here: MinkowskiEngine, tensorboardX, imageio are
absent); this file follows the reference text line by line and is what the device path is compared with.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from scipy.spatial import cKDTree


def _gate(logits, binary_pred, base_cat, novel_cat):
    novel, base = logits.clone(), logits.clone()
    novel[:, base_cat] = -1e10   # infer.py:606-607: the novel branch may not take base classes
    base[:, novel_cat] = -1e10
    bp = binary_pred.reshape(-1, 1).to(logits.dtype)
    return bp * base + (1 - bp) * novel


def postprocess_view(cfg, outputs, xyz, s=0):
    """outputs: dict of CPU tensors of ONE forward; entry s.  -> (pred, pred_2d, pred_3d) int64 class ids per visible point."""
    cs = cfg.category_split
    base_cat, novel_cat, all_cat = list(cs["base_category"]), list(cs["novel_category"]), list(cs["all_category"])
    text = F.normalize(outputs["text_embed"].float(), dim=-1)
    scale = outputs["logit_scale"].float()
    binary_pred = outputs["binary_pred_view"].float()
    fused = F.normalize(outputs["fused_pred_feature"][s].float(), dim=-1)
    f2d = outputs["2d_pred_feature"][s].float().clone()
    empty = f2d.sum(1) == 0
    if bool(empty.any()) and not bool(empty.all()):  # infer.py:523-553
        true_idx = torch.where(~empty)[0]
        _, ind = cKDTree(xyz[~empty].numpy()).query(xyz[empty].numpy(), k=1)
        f2d[torch.where(empty)[0]] = f2d[true_idx[torch.from_numpy(np.asarray(ind)).long()]]
    f2d = F.normalize(f2d, dim=-1)
    f3d = F.normalize(outputs["pure3d_pred_feature"][s].float(), dim=-1)
    logits = (scale * (fused @ text.t())).softmax(dim=-1)
    open_emb = F.normalize(outputs["final_pred_open_embedding"][s].float(), dim=-1)
    open_logits = (scale * (open_emb @ text.t())).softmax(dim=-1)
    overlap = torch.tensor([int(c in base_cat) for c in all_cat], dtype=torch.long)
    for single_mask, open_logit in zip(outputs["final_mask_3d"][s], open_logits):  # infer.py:585-601, sequential
        if not bool(single_mask.any()):
            continue
        b = (logits[single_mask] ** cfg.base_ratio * open_logit ** (1 - cfg.base_ratio)).log() * overlap
        n = (logits[single_mask] ** cfg.novel_ratio * open_logit ** (1 - cfg.novel_ratio)).log() * (1 - overlap)
        logits[single_mask] = b + n
    pred = _gate(logits, binary_pred, base_cat, novel_cat).argmax(1)
    pred_2d = _gate(scale * (f2d @ text.t()), binary_pred, base_cat, novel_cat).argmax(1)
    pred_3d = _gate(scale * (f3d @ text.t()), binary_pred, base_cat, novel_cat).argmax(1)
    return pred, pred_2d, pred_3d


def vote_scene(n_points, n_classes, per_view, scene_xyz):
    """per_view: list of (visible point indices (int64), (pred, pred_2d, pred_3d)) -> three (n_points,) label arrays."""
    votes = [torch.zeros(n_points, n_classes) for _ in range(3)]
    counter = torch.zeros(n_points)
    for idx, preds in per_view:  # infer.py:642-647
        for v, p in zip(votes, preds):
            v[idx, p] += 1
        counter[idx] += 1
    seen = counter != 0
    out = []
    match = None
    if bool((~seen).any()) and bool(seen.any()):  # infer.py:682-694
        true_idx = torch.where(seen)[0]
        _, ind = cKDTree(scene_xyz[seen.numpy()]).query(scene_xyz[(~seen).numpy()], k=1)
        match = true_idx[torch.from_numpy(np.asarray(ind)).long()]
    for v in votes:
        lab = v.argmax(1)
        if match is not None:
            lab[~seen] = lab[match]
        out.append(lab)
    return out


def scene_forward_cpu(cpu_model, cfg, scene, matrices):
    """The whole scene on the CPU: per view the eval forward through oracle/model_oracle.py, per-view post-processing, votes,
    fill.  `matrices`: one 4x4 voxelisation transform per view.  -> (labels [fused, 2d, 3d], per-view outputs)."""
    from xmask3d_amd import synthetic

    from . import model_oracle, voxel_oracle

    per_view, outs = [], []
    for v in range(len(scene.poses)):
        vis, rows, cols = synthetic.view_subset(scene, v)
        pts = scene.points[vis]
        grid, inds, inv = voxel_oracle.voxelize_with_matrix(pts, matrices[v])
        coords = torch.from_numpy(np.concatenate([np.zeros((len(grid), 1)), grid], 1).astype(np.int32))
        feats = torch.from_numpy((scene.colors[vis][inds] / 127.5 - 1).astype(np.float32))
        cbatch = {"sinput": model_oracle.CpuSparseTensor(feats, coords), "img": torch.from_numpy(scene.images[v]).permute(2, 0, 1)[None],
                  "x_label": torch.from_numpy(rows).long(), "y_label": torch.from_numpy(cols).long(),
                  "inds_reconstruct": torch.from_numpy(inv), "captions": (scene.captions[v],),
                  "ori_coords": torch.cat([torch.zeros(len(pts), 1), torch.from_numpy(pts).float()], 1),
                  "point_offsets": [0, len(pts)], "compact_outputs": False}
        _, ref = model_oracle.forward_cpu(cpu_model, cbatch)
        ref["binary_pred_view"] = ref["binary_pred"]
        preds = postprocess_view(cfg, ref, torch.from_numpy(pts).float())
        per_view.append((torch.from_numpy(np.nonzero(vis)[0]).long(), preds))
        outs.append(ref)
    ncls = len(cfg.category_split["base_category"]) + len(cfg.category_split["novel_category"])
    return vote_scene(scene.points.shape[0], ncls, per_view, scene.points.astype(np.float32)), outs
