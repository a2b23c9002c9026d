"""Training losses of XMask3D.

Restated from the reference (same names of losses / weights / attributes):
  * point sampling helpers       detectron2.projects.point_rend.point_features (point_sample,
                                 get_uncertain_point_coords_with_randomness) as used by
                                 third_party/Mask2Former/mask2former/modeling/criterion.py:13-16,156-181
  * HungarianMatcher             third_party/Mask2Former/mask2former/modeling/matcher.py:70-156 (scipy assignment)
  * SetCriterion losses          .../criterion.py:20-60 (dice, sigmoid-CE), :129-197 (labels, masks)
  * Criterion (XMask3D)          /root/reference/models/utils/criterion.py:11-376: weight_dict, fuser/fc1/fc2/clip,
                                 forward (per-scene mask selection :245-328, mask_mapper :330, loss_exact :184-207,
                                 loss_contra :39-182, aux losses :366-374), and models/utils/fuser.py:6-53 mask_mapper
PARITY: ``mask_mapper`` is pinned by tests/golden/fuser.npz; the point-sampled losses are unpinned (detectron2 absent)
and checked against closed forms in tests/test_criterion.py.  The mask->point accumulation of mask_mapper is written
as one masked matmul (identical sums, differentiable w.r.t. mask embeddings) instead of a per-query index loop.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from .clip_model import MaskCLIP


# ----------------------------------------------------------------------------- point sampling
def point_sample(inp, point_coords, **kwargs):
    """inp (N,C,H,W), point_coords (N,P,2) in [0,1]x[0,1] (x,y) -> (N,C,P) bilinear samples."""
    out = F.grid_sample(inp, 2.0 * point_coords.unsqueeze(2) - 1.0, **kwargs)
    return out.squeeze(3)


def _rand(shape, device):
    """uniform [0, 1) point coordinates (detectron2's point sampling draws torch.rand on the logits' device); one place, so that a test
    can make the device run and the CPU oracle draw the SAME points from a host generator"""
    return torch.rand(*shape, device=device)


def get_uncertain_point_coords_with_randomness(logits, uncertainty_func, num_points, oversample_ratio, importance_sample_ratio):
    n = logits.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    coords = _rand((n, num_sampled, 2), logits.device)
    unc = uncertainty_func(point_sample(logits, coords, align_corners=False))
    num_uncertain = int(importance_sample_ratio * num_points)
    num_random = num_points - num_uncertain
    idx = torch.topk(unc[:, 0, :], k=num_uncertain, dim=1)[1]
    idx = idx + num_sampled * torch.arange(n, dtype=torch.long, device=logits.device)[:, None]
    coords = coords.view(-1, 2)[idx.view(-1), :].view(n, num_uncertain, 2)
    if num_random > 0:
        coords = torch.cat([coords, _rand((n, num_random, 2), logits.device)], dim=1)
    return coords


def dice_loss(inputs, targets, num_masks):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * (inputs * targets).sum(-1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    return (1 - (numerator + 1) / (denominator + 1)).sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks):
    return F.binary_cross_entropy_with_logits(inputs, targets, reduction="none").mean(1).sum() / num_masks


def batch_dice_loss(inputs, targets):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * torch.einsum("nc,mc->nm", inputs, targets)
    denominator = inputs.sum(-1)[:, None] + targets.sum(-1)[None, :]
    return 1 - (numerator + 1) / (denominator + 1)


def batch_sigmoid_ce_loss(inputs, targets):
    hw = inputs.shape[1]
    pos = F.binary_cross_entropy_with_logits(inputs, torch.ones_like(inputs), reduction="none")
    neg = F.binary_cross_entropy_with_logits(inputs, torch.zeros_like(inputs), reduction="none")
    return (torch.einsum("nc,mc->nm", pos, targets) + torch.einsum("nc,mc->nm", neg, 1 - targets)) / hw


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class=1.0, cost_mask=1.0, cost_dice=1.0, num_points=0):
        super().__init__()
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.cost_class, self.cost_mask, self.cost_dice, self.num_points = cost_class, cost_mask, cost_dice, num_points

    @torch.no_grad()
    def cost_matrices(self, outputs, targets):
        """matcher.py:103-149: per image the (Q, T) matching cost 5*BCE + 5*dice + 2*(-prob) on num_points random points"""
        bs, num_queries = outputs["pred_logits"].shape[:2]
        costs = []
        for b in range(bs):
            out_prob = outputs["pred_logits"][b].softmax(-1)
            cost_class = -out_prob[:, targets[b]["labels"]]
            out_mask = outputs["pred_masks"][b][:, None].float()
            tgt_mask = targets[b]["masks"].to(out_mask)[:, None]
            pts = _rand((1, self.num_points, 2), out_mask.device)  # shared by all masks of the image
            tgt = point_sample(tgt_mask, pts.repeat(tgt_mask.shape[0], 1, 1), align_corners=False).squeeze(1)
            out = point_sample(out_mask, pts.repeat(out_mask.shape[0], 1, 1), align_corners=False).squeeze(1)
            C = self.cost_mask * batch_sigmoid_ce_loss(out, tgt) + self.cost_class * cost_class + self.cost_dice * batch_dice_loss(out, tgt)
            costs.append(C.reshape(num_queries, -1))
        return costs

    @staticmethod
    def assign(costs):
        """list of (Q, T_i) cost matrices -> list of (query idx, target idx) int64 pairs sorted by query (scipy's order).
        Device matrices: ONE xm3d_linear_sum_assignment launch for all of them, no host round trip (the reference moves each
        matrix to the host for scipy, matcher.py:151-152); CPU matrices: scipy, as the reference."""
        if not costs:
            return []
        if costs[0].device.type == "cpu":
            out = []
            for C in costs:
                i, j = linear_sum_assignment(C)
                out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
            return out
        from . import ops

        Q = costs[0].shape[0]
        sizes = [int(C.shape[1]) for C in costs]
        Tm = max(max(sizes), 1)
        if Q > 64 or Tm > 256:
            raise RuntimeError(f"HungarianMatcher.assign: Q={Q}, T={Tm} outside the device solver's range (Q <= 64, T <= 256)")
        dev = costs[0].device
        stack = torch.zeros((len(costs), Q, Tm), dtype=torch.float32, device=dev)
        for m, C in enumerate(costs):
            stack[m, :, : sizes[m]] = C
        key = (tuple(sizes), str(dev))
        cache = HungarianMatcher._nt_cache
        if key not in cache:  # the target counts of a batch repeat for all ten decoder outputs: one upload per distinct tuple
            if len(cache) > 64:
                cache.clear()
            host = torch.tensor(sizes, dtype=torch.int32).pin_memory()  # pinned + non_blocking: the host does not wait for the stream
            cache[key] = (host.to(dev, non_blocking=True), host)
        oq, ot = ops.linear_sum_assignment(stack, cache[key][0])
        return [(oq[m, : min(n, Q)], ot[m, : min(n, Q)]) for m, n in enumerate(sizes)]

    _nt_cache = {}

    @torch.no_grad()
    def forward(self, outputs, targets):
        return self.assign(self.cost_matrices(outputs, targets))


def mask_mapper(x_list, y_list, masks, mask_embeds, pred_3ds, fuser, fc1, fc2, cfg):
    """Per scene: binary masks (Q,H,W) -> per-point mean mask embedding, fused with the 3D feature where covered."""
    output, output_2d, output_3d, output_2d_pre = [], [], [], []
    for x_label, y_label, mask, mask_embed, pred_3d in zip(x_list, y_list, masks, mask_embeds, pred_3ds):
        mask_3d = mask[:, x_label, y_label] >= 0.5
        if not bool(mask_3d.any()):
            mask_3d = mask_3d.clone()
            mask_3d[0, 0] = True
        m = mask_3d.to(pred_3d.dtype)                      # (Q, Np)
        counter = m.sum(0)[:, None]                        # queries covering each point
        feat = m.t() @ mask_embed                          # sum of embeddings of the covering queries
        feat = feat / torch.where(counter == 0, torch.full_like(counter, 1e-5), counter)
        need = counter[:, 0] >= 1
        fused = pred_3d.clone()
        fused[need] = fuser(feat[need], pred_3d[need])
        output.append(fused)
        output_2d.append(fc2(feat))
        output_3d.append(fc1(pred_3d))
        if cfg.caption_contra_2d_pre:
            output_2d_pre.append(feat[need])
    return output, output_2d, output_3d, output_2d_pre


class FeatureMerger(nn.Module):
    def __init__(self, feature_dim):
        super().__init__()
        self.linear = nn.Linear(feature_dim * 2, feature_dim)

    def forward(self, X, Y):
        return self.linear(torch.cat((X, Y), dim=1))


class Criterion(nn.Module):
    """SetCriterion (labels + point-sampled masks, with aux layers) + the XMask3D 3D losses."""

    def __init__(self, num_classes, matcher, class_weight, mask_weight, dice_weight, num_layers, eos_coef, losses, num_points,
                 oversample_ratio, importance_sample_ratio, cfg):
        super().__init__()
        self.num_classes, self.matcher = num_classes, matcher
        wd = {"loss_ce": class_weight, "loss_mask": mask_weight, "loss_dice": dice_weight}
        aux = {}
        for i in range(num_layers):
            aux.update({f"{k}_{i}": v for k, v in wd.items()})
        wd.update(aux)
        lw = cfg.loss_weight
        for k in ("loss_3d", "loss_3d_pure", "loss_explicit_contra", "loss_explicit_contra_3d", "loss_explicit_contra_2d_pre",
                  "loss_binary"):
            wd[k] = lw[k]
        self.weight_dict = wd
        self.eos_coef, self.losses = eos_coef, losses
        empty_weight = torch.ones(num_classes + 1)
        empty_weight[-1] = eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio, self.importance_sample_ratio = num_points, oversample_ratio, importance_sample_ratio
        self.fuser = FeatureMerger(feature_dim=768)
        self.criterion = nn.CrossEntropyLoss(ignore_index=cfg.ignore_label)
        self.ignore_label, self.mask_contra_3d = cfg.ignore_label, cfg.mask_contra_3d
        self.fc1, self.fc2 = nn.Identity(), nn.Identity()
        self.contra_criterion = nn.CosineSimilarity()
        self.cfg = cfg
        self.clip = MaskCLIP(name=cfg.clip_name)

    # -------- SetCriterion pieces
    @staticmethod
    def _src_idx(indices):
        return (torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)]), torch.cat([src for src, _ in indices]))

    @staticmethod
    def _tgt_idx(indices):
        return (torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)]), torch.cat([tgt for _, tgt in indices]))

    def loss_labels(self, outputs, targets, indices, num_masks):
        src_logits = outputs["pred_logits"].float()
        idx = self._src_idx(indices)
        target_o = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
        target = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64, device=src_logits.device)
        target[idx] = target_o
        return {"loss_ce": F.cross_entropy(src_logits.transpose(1, 2), target, self.empty_weight)}

    def loss_masks(self, outputs, targets, indices, num_masks):
        src_idx, tgt_idx = self._src_idx(indices), self._tgt_idx(indices)
        src_masks = outputs["pred_masks"][src_idx][:, None]
        target_masks = torch.stack([t["masks"] for t in targets]) if len({t["masks"].shape for t in targets}) == 1 else None
        if target_masks is None:  # ragged number of masks per image: pad to the largest
            q = max(t["masks"].shape[0] for t in targets)
            h, w = targets[0]["masks"].shape[-2:]
            target_masks = torch.zeros(len(targets), q, h, w, device=src_masks.device)
            for i, t in enumerate(targets):
                target_masks[i, : t["masks"].shape[0]] = t["masks"]
        target_masks = target_masks.to(src_masks)[tgt_idx][:, None]
        with torch.no_grad():
            coords = get_uncertain_point_coords_with_randomness(src_masks, lambda lg: -torch.abs(lg), self.num_points,
                                                                self.oversample_ratio, self.importance_sample_ratio)
            point_labels = point_sample(target_masks, coords, align_corners=False).squeeze(1)
        point_logits = point_sample(src_masks, coords, align_corners=False).squeeze(1)
        return {"loss_mask": sigmoid_ce_loss(point_logits, point_labels, num_masks),
                "loss_dice": dice_loss(point_logits, point_labels, num_masks)}

    def get_loss(self, loss, outputs, targets, indices, num_masks):
        return {"labels": self.loss_labels, "masks": self.loss_masks}[loss](outputs, targets, indices, num_masks)

    # -------- XMask3D pieces
    def loss_exact(self, outputs, gt):
        fused = F.normalize(torch.cat(outputs["fused_pred_feature"]), dim=-1)
        f3d = F.normalize(torch.cat(outputs["pure3d_pred_feature"]), dim=-1)
        text = torch.cat([F.normalize(outputs["text_embed"], dim=-1), F.normalize(outputs["null_embed"], dim=-1)])
        scale = outputs["logit_scale"]
        if bool((gt == self.ignore_label).all()):
            gt = gt.clone()
            gt[0] = self.ignore_label - 1
        return {"loss_3d": self.criterion(scale * (fused @ text.t()), gt), "loss_3d_pure": self.criterion(scale * (f3d @ text.t()), gt)}

    def loss_contra(self, x_list, y_list, binary_gts, outputs):
        masks = F.interpolate(outputs["pred_masks"], size=tuple(self.cfg.mask_shape), mode="bilinear", align_corners=False)
        emb_3d, emb_gt, final_2d_mask = [], [], []
        last_mask_embed = None
        for b, (x_label, y_label, mask, feat_fused, mask_embed, feat_3d, clip_emb, binary_gt) in enumerate(
                zip(x_list, y_list, masks, outputs["fused_pred_feature"], outputs["mask_embed"], outputs["pure3d_pred_feature"],
                    outputs["mask_embed_clip"], binary_gts)):
            mask_3d = mask[:, x_label, y_label].sigmoid() >= 0.5
            if not bool((mask_3d.sum(1) >= 10).any()):
                mask_3d = mask_3d.clone()
                mask_3d[0, :] = True
            keep = mask_3d.sum(1) >= 10
            mask_k, clip_k, mask_3d = mask[keep], clip_emb[keep], mask_3d[keep]
            last_mask_embed = mask_embed[keep]
            # per-mask statistics for all kept masks at once, ONE device->host copy (the reference's per-mask python loop,
            # criterion.py:186-215, costs two host syncs per mask)
            bg = binary_gt.view(1, -1)
            tot = mask_3d.sum(1)
            n_novel = (mask_3d & bg.eq(0)).sum(1)
            n_base_ = (mask_3d & bg.eq(1)).sum(1)
            p = mask_k.sigmoid().flatten(1)
            hi = p > 0.5
            conf = (p * hi).sum(1) / hi.sum(1)  # mean of p[p > 0.5]; NaN for an empty set, like .mean() of nothing
            st = torch.stack([tot.double(), n_novel.double(), n_base_.double(), conf.double()]).cpu().tolist()
            novel, base = [], []
            for i in range(mask_3d.shape[0]):
                novel_num, base_num_ = int(st[1][i]), int(st[2][i])
                base_num, novel_num_ = int(st[0][i]) - novel_num, int(st[0][i]) - base_num_
                if novel_num > 1.8 * base_num and novel_num > 10:
                    novel.append((i, st[3][i]))
                elif base_num_ > 20 * novel_num_ and base_num_ > 150:
                    base.append((i, st[3][i]))
            if novel or base:
                pick = [i for i, _ in sorted(novel, key=lambda t: t[1], reverse=True)][:4]
                pick += [i for i, _ in sorted(base, key=lambda t: t[1], reverse=True)][:1]
                emb_gt.append(torch.stack([clip_k[i] for i in pick]))
                emb_3d.append(torch.stack([feat_3d[mask_3d[i]].mean(0) for i in pick]))
                final_2d_mask.append((b, torch.stack([mask_k[i] for i in pick])))
        if emb_3d:
            loss = (1 - self.contra_criterion(torch.cat(emb_3d), torch.cat(emb_gt).detach())).mean()
        else:
            inv = last_mask_embed[:1]
            loss = (1 - self.contra_criterion(inv, inv)).mean()
        return {"loss_3d_contra": loss}, final_2d_mask

    def select_masks(self, outputs, batch_input):
        """criterion.py:245-328: per scene, arg-max ownership of pixels among queries, binary masks of the kept queries."""
        masks = F.interpolate(outputs["pred_masks"], size=tuple(self.cfg.mask_shape), mode="bilinear", align_corners=False)
        ori = batch_input["ori_coords"]
        x_list, y_list, p3d_list, m_list, e_list, o_list, bg_list = [], [], [], [], [], [], []
        for s in ori[:, 0].unique():
            sel = ori[:, 0] == s
            si = int(s)
            x_list.append(batch_input["x_label"][sel])
            y_list.append(batch_input["y_label"][sel])
            p3d_list.append(outputs["pred_3d"][sel])
            if batch_input.get("binary_label_3d") is not None:
                bg_list.append(batch_input["binary_label_3d"][sel])
            mask_pred = masks[si].sigmoid()
            scores = F.softmax(outputs["pred_logits"][si], dim=-1).max(-1)[0]
            emb, emb_open = outputs["mask_embed"][si], outputs["mask_embed_clip"][si]
            ids = (scores.view(-1, 1, 1) * mask_pred).argmax(0)
            q = torch.arange(mask_pred.shape[0], device=ids.device).view(-1, 1, 1)
            final = (ids[None] == q) & (mask_pred >= 0.5)
            keep = final.flatten(1).any(1)
            if bool(keep.any()):
                m_list.append(final[keep].float())
                e_list.append(emb[keep])
                o_list.append(emb_open[keep])
            else:
                m_list.append(torch.zeros_like(mask_pred))
                e_list.append(torch.zeros_like(emb))
                o_list.append(torch.zeros_like(emb_open))
        return x_list, y_list, p3d_list, m_list, e_list, o_list, bg_list

    def forward(self, outputs, targets, batch_input):
        import torch.distributed as dist

        no_aux = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        aux_list = list(outputs.get("aux_outputs", [])) if self.training else []
        # all matchings of the iteration (main + nine auxiliary decoder outputs) in one batched device launch
        costs = self.matcher.cost_matrices(no_aux, targets)
        nb = len(costs)
        for aux in aux_list:
            costs += self.matcher.cost_matrices(aux, targets)
        matched = self.matcher.assign(costs)
        indices = matched[:nb]
        num_masks = torch.as_tensor([sum(len(t["labels"]) for t in targets)], dtype=torch.float,
                                    device=outputs["pred_masks"].device)
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(num_masks)
            world = dist.get_world_size()
        num_masks = torch.clamp(num_masks / world, min=1).item()
        losses = {}
        if self.training:
            for loss in self.losses:
                losses.update(self.get_loss(loss, outputs, targets, indices, num_masks))
        outputs.update(self.clip(outputs["images"], outputs["pred_masks"]))
        x_list, y_list, p3d_list, m_list, e_list, o_list, bg_list = self.select_masks(outputs, batch_input)
        fused, out2d, out3d, out2d_pre = mask_mapper(x_list, y_list, m_list, e_list, p3d_list, self.fuser, self.fc1, self.fc2, self.cfg)
        outputs.update({"fused_pred_feature": fused, "2d_pred_feature": out2d, "pure3d_pred_feature": out3d,
                        "2d_pred_feature_pre": out2d_pre, "final_pred_mask": m_list})
        if not self.training:
            return outputs
        losses.update(self.loss_exact(outputs, batch_input["labels_3d"]))
        if self.mask_contra_3d:
            lc, final_2d_mask = self.loss_contra(x_list, y_list, bg_list, outputs)
            losses.update(lc)
            outputs["final_pred_mask"] = final_2d_mask
        for i, aux in enumerate(aux_list):
            idx = matched[nb * (i + 1): nb * (i + 2)]
            for loss in self.losses:
                losses.update({f"{k}_{i}": v for k, v in self.get_loss(loss, aux, targets, idx, num_masks).items()})
        return losses, outputs
