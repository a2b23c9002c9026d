"""XMASK3d: the drop-in model surface.

Mirror of /root/reference/models/xmask3d.py:27-489: same constructor (``XMASK3d(cfg)`` with the flat
CfgNode of util/config.py), same sub-module attribute names (``pc_decoder``, ``pc_binary_head``,
``backbone``, ``sem_seg_head``, ``criterion.{fuser,fc1,fc2,clip}``, ``category_head``, ``clip_head``) so
``named_parameters()`` / ``state_dict()`` keys line up with the released checkpoints, same
``forward(batch_input) -> (losses | None, outputs)`` contract and output keys.

MI355X-first differences (results unchanged):
  * the 3D nets run on the HIP sparse-conv kernels behind ``sinput`` (xmask3d_amd.me_compat.SparseTensor);
    both U-Nets share one coordinate manager / rulebook set
  * per-scene max of the implicit caption rows is one scatter-reduce, not a Python loop (xmask3d.py:153-159)
  * the <=50-iteration boolean-index loops of the eval fusion (xmask3d.py:421-451) are one HIP kernel
    (xm3d_mask_point_fuse) and the FeatureMerger is applied with a select instead of three index copies
  * frozen SD / CLIP nets can run in bf16 (``dense_dtype``), the reference is fp32 throughout
Reference quirk NOT kept: its eval fusion reads ``binary_scores`` of ALL points in the batch for every scene and
re-applies the sigmoid per scene (xmask3d.py:363), which only works for the batch-1 inference its drivers use; here
each scene reads its own slice, so a batch of B views gives exactly the B batch-1 results.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

import numpy as np

from . import ops
from .clip_model import CategoryEmbed
from .criterion import Criterion, HungarianMatcher
from .image_branch import FeatureExtractorBackbone, LdmImplicitCaptionerExtractor
from .mask_head import (MaskFormerHead, MSDeformAttnPixelDecoder, ODISEMultiScaleMaskedTransformerDecoder, PooledMaskEmbed,
                        PseudoClassEmbed)
from .pc_processor import PC_Binary_Processor, PC_Processor


def ensemble_logits_with_labels(logits, labels, ensemble_method="max"):
    """per-label max/mean over its synonyms (models/modeling/meta_arch/helper.py:72-97; pinned by golden/ensemble.npz)."""
    lens = [len(l) for l in labels]
    assert logits.shape[-1] == sum(lens), f"{logits.shape[-1]} != {sum(lens)}"
    assert ensemble_method in ("mean", "max")
    if all(n == 1 for n in lens):
        return logits.clone()
    outs, start = [], 0
    for n in lens:
        chunk = logits[..., start:start + n]
        outs.append(chunk.max(dim=-1).values if ensemble_method == "max" else chunk.mean(dim=-1))
        start += n
    return torch.stack(outs, dim=-1)


def _record_stream(obj, stream):
    """tensors allocated on one stream and read on another: tell the caching allocator about the second reader"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class XMASK3d(nn.Module):
    def __init__(self, cfg=None, dense_dtype=torch.float32, prune_dead_compute=True):
        super().__init__()
        self.cfg = cfg
        num_classes, num_queries = cfg.classes, cfg.num_queries
        self.pixel_mean, self.pixel_std = cfg.pixel_mean, cfg.pixel_std
        self.seq_len, self.size_divisibility = 77, 64
        self.dense_dtype = dense_dtype
        self.prune_dead_compute = prune_dead_compute
        self.channels_last = False
        self.low_precision_heads = True
        self.register_buffer("_pixel_mean", torch.tensor(cfg.pixel_mean, dtype=torch.float32).view(1, 3, 1, 1), False)
        self.register_buffer("_pixel_std", torch.tensor(cfg.pixel_std, dtype=torch.float32).view(1, 3, 1, 1), False)
        self._dense_graphs = None
        self.pc_decoder = PC_Processor(arch_3d=cfg.arch_3d)
        self.pc_binary_head = PC_Binary_Processor(arch_3d=cfg.arch_binary_head)
        cs = cfg.category_split
        self.ignore_label = cs["ignore_category"] if isinstance(cs, dict) else cs.ignore_category
        self.binary_loss_func = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([cfg.data_ratio]))
        self.backbone = FeatureExtractorBackbone(
            feature_extractor=LdmImplicitCaptionerExtractor(
                encoder_block_indices=(5, 7), unet_block_indices=(2, 5, 8, 11), decoder_block_indices=(2, 5), steps=(0,),
                learnable_time_embed=True, num_timesteps=1, dim_latent=768, clip=None, prune_dead_compute=prune_dead_compute),
            out_features=["s2", "s3", "s4", "s5"],
            # the reference re-computes the projections in backward (use_checkpoint=True, odise backbone) to fit 24-80 GB cards;
            # with 288 GB of HBM the activations simply stay resident (same values, one forward less per iteration)
            use_checkpoint=bool(getattr(cfg, "activation_checkpointing", False)), slide_training=False)
        self.sem_seg_head = MaskFormerHead(
            ignore_value=255, num_classes=num_classes,
            pixel_decoder=MSDeformAttnPixelDecoder(
                conv_dim=256, mask_dim=256, norm="GN", transformer_dropout=0.0, transformer_nheads=8,
                transformer_dim_feedforward=1024, transformer_enc_layers=6, transformer_in_features=["s3", "s4", "s5"],
                common_stride=4, input_shape=self.backbone.output_shape()),
            loss_weight=1.0, transformer_in_feature="multi_scale_pixel_decoder",
            transformer_predictor=ODISEMultiScaleMaskedTransformerDecoder(
                class_embed=PseudoClassEmbed(num_classes=num_classes), hidden_dim=256,
                post_mask_embed=PooledMaskEmbed(hidden_dim=256, mask_dim=256, projection_dim=768), in_channels=256,
                mask_classification=True, num_classes=num_classes, num_queries=num_queries, nheads=8, dim_feedforward=2048,
                dec_layers=9, pre_norm=False, enforce_input_project=False, mask_dim=256),
            input_shape=self.backbone.output_shape())
        self.sem_seg_head.predictor.prune_aux_embed = prune_dead_compute
        self.criterion = Criterion(
            num_layers=9, class_weight=2.0, mask_weight=5.0, dice_weight=5.0, num_classes=num_classes,
            matcher=HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=12544), eos_coef=0.1,
            losses=["labels", "masks"], num_points=12544, oversample_ratio=3.0, importance_sample_ratio=0.75, cfg=cfg)
        self.category_head = CategoryEmbed(clip_model_name=self.criterion.clip, labels=[[l] for l in cfg.label],
                                           test_labels=[[l] for l in cfg.all_label], projection_dim=-1)
        self.clip_head = self.criterion.clip
        self.set_dense_dtype(dense_dtype)
        # frozen SD / CLIP weights, BPE vocabulary and the empty-prompt conditioning from local files when they are there
        # (sd_model/sd-v1-3.ckpt, openai/: ldm.py:105-114, clip.py:69-73); seeded random weights + stand-in tokenizer otherwise
        from . import checkpoint as _ckpt

        self.pretrained_report = None
        if any(_ckpt.find_pretrained(cfg).values()):
            _ckpt.load_pretrained(self, cfg)

    def set_dense_dtype(self, dense_dtype):
        """frozen SD / CLIP-visual nets hold their weights in the dense compute dtype (fp32 or bf16)"""
        self.dense_dtype = dense_dtype
        ldm = self.backbone.feature_extractor.ldm_extractor.ldm
        ldm.first_stage_model.to(dense_dtype)
        ldm.unet_model.to(dense_dtype)
        self.criterion.clip.clip.visual.to(dense_dtype)
        return self

    def set_sparse_dtype(self, dtype):
        """bf16: the two sparse U-Nets run the plain-bf16 form of the sparse convolution at inference (activations one bf16 plane, one
        MFMA per product: csrc/spconv_split.hip BF) - the bf16 configuration; f32 (default): f32 activations, split-operand products
        (~f32 accuracy: the fp32 configuration and training)."""
        from . import me_compat

        on = dtype == torch.bfloat16
        for net in (self.pc_decoder, self.pc_binary_head):
            for m in net.modules():
                if isinstance(m, me_compat._ConvBase):
                    m.bf16_io = on
        self.sparse_dtype = dtype
        return self

    def cast_head_weights(self, dtype=torch.bfloat16):
        """Inference only: hold the GEMM / convolution weights of the trainable heads that run under bf16 autocast (feature
        projections, pixel decoder, transformer decoder) in bf16, so that autocast finds nothing to cast - otherwise every
        forward (and every graph replay) re-casts ~500 fp32 weight tensors.  Normalisation, embedding and positional
        parameters keep fp32.  The values are the ones autocast would have produced; fp32 masters are lost, so do not train
        this instance afterwards."""
        for root in (self.backbone.feature_projections, self.sem_seg_head):
            for m in root.modules():
                if isinstance(m, (nn.Linear, nn.Conv2d)):
                    m.weight.data = m.weight.data.to(dtype)
                    if m.bias is not None:
                        m.bias.data = m.bias.data.to(dtype)
                elif isinstance(m, nn.MultiheadAttention):
                    m.in_proj_weight.data = m.in_proj_weight.data.to(dtype)
                    m.in_proj_bias.data = m.in_proj_bias.data.to(dtype)
        self.heads_cast = dtype
        return self

    # ------------------------------------------------------------------ helpers
    def cal_pred_logits(self, outputs):
        mask_embed = F.normalize(outputs["mask_embed"], dim=-1)
        text_embed = F.normalize(outputs["text_embed"], dim=-1)
        logit_scale = outputs["logit_scale"]
        pred = logit_scale * (mask_embed @ text_embed.t())
        pred = ensemble_logits_with_labels(pred, outputs["labels"], ensemble_method="max")
        null_pred = logit_scale * (mask_embed @ F.normalize(outputs["null_embed"], dim=-1).t())
        return torch.cat([pred, null_pred], dim=-1)

    def set_channels_last(self, on=True):
        """NHWC activations/weights for the frozen conv nets (MIOpen's bf16 implicit-GEMM kernels are NHWC)"""
        self.channels_last = on
        fmt = torch.channels_last if on else torch.contiguous_format
        ldm = self.backbone.feature_extractor.ldm_extractor.ldm
        for mod in (ldm.first_stage_model, ldm.unet_model, self.backbone.feature_projections):
            mod.to(memory_format=fmt)
        return self

    def encode_3d(self, sinput, inds_reconstruct, batch_size):
        imp_condition, pred_3d, idx = self.pc_decoder(sinput)
        pred_3d = pred_3d[inds_reconstruct, :]
        cond = torch.full((batch_size, imp_condition.shape[1]), float("-inf"), dtype=imp_condition.dtype, device=pred_3d.device)
        cond = cond.scatter_reduce(0, idx.long()[:, None].expand_as(imp_condition), imp_condition, reduce="amax")
        binary_scores = self.pc_binary_head(sinput)[inds_reconstruct, :]
        return pred_3d, cond, binary_scores

    def normalize_images(self, img):
        img = img.float()
        images = (img - self._pixel_mean) / self._pixel_std
        h, w = images.shape[-2:]
        ph, pw = (-h) % self.size_divisibility, (-w) % self.size_divisibility
        if ph or pw:
            images = F.pad(images, (0, pw, 0, ph))
        images = images.to(self.dense_dtype)
        if self.channels_last:
            images = images.contiguous(memory_format=torch.channels_last)
        return images

    def encode_vae(self, img):
        """VAE-encoder stage of the dense branch (no dependence on the 3D nets)."""
        images = self.backbone.prepare(self.normalize_images(img))
        return self.backbone.feature_extractor.ldm_extractor.encode(images)

    def dense_features(self, img, imp_condition_input, encoded=None, fork_stream=None):
        """First half of the dense branch (rows a8-a12): SD feature extractor + projections.  Convolution-bound: fills the
        device on its own."""
        img = img.to(imp_condition_input.device).float()
        return self.backbone(self.normalize_images(img), imp_condition_input, encoded, fork_stream)

    def _decode_heads(self, img, feature):
        """pixel decoder + transformer decoder on the projected features -> decoder outputs (+ the [0,1] images mask-CLIP reads)"""
        low = self.dense_dtype != torch.float32 and self.low_precision_heads and not torch.is_grad_enabled()
        # bf16 mode: GEMMs of the pixel / transformer decoder run in bf16 too (the reference keeps them fp32; sampling in
        # xm3d_msda_forward, LayerNorm statistics and the mask logits stay f32)
        with torch.autocast(device_type=img.device.type, dtype=torch.bfloat16, enabled=low):
            # (low: the projected features are bf16 and every consumer runs under autocast - an f32 copy would be cast straight back)
            outputs = self.sem_seg_head(feature if low else {k: v.float() for k, v in feature.items()})
        for k in ("pred_masks", "mask_embed", "mask_pooled_features"):
            outputs[k] = outputs[k].float()
        outputs["images"] = img.float() / 255.0
        return outputs

    def dense_heads(self, img, feature):
        """Second half (rows a13-a17): pixel decoder, transformer decoder, category logits, mask-CLIP.  Many small
        dependent kernels: latency-bound, leaves most of the device idle."""
        outputs = self._decode_heads(img, feature)
        outputs.update(self.category_head(outputs))
        outputs["pred_logits"] = self.cal_pred_logits(outputs)
        clip_embed = self.clip_head(outputs["images"], outputs["pred_masks"])  # casts to the visual tower's dtype inside
        outputs["mask_embed_clip"] = clip_embed["mask_embed_clip"].float()
        return outputs

    def encode_2d(self, img, imp_condition_input, encoded=None, fork_stream=None):
        """img (B,3,H,W) 0..255 -> decoder outputs (the training forward adds category head and losses itself)."""
        img = img.to(imp_condition_input.device)
        if self._head_graphs is not None and torch.is_grad_enabled() and img.is_cuda and self.dense_dtype == torch.float32:
            return self._decode_heads_graphed(img, imp_condition_input)
        return self._decode_heads(img, self.dense_features(img, imp_condition_input, encoded, fork_stream))

    _head_graphs = None

    def enable_train_graphs(self, on=True, heads=False):
        """Training at one static view shape per GPU: the frozen UNet's forward + backward (image_branch._GraphedTaps) and the frozen VAE
        stages (gradient-free, on the inference kernels) replay as HIP graphs.  The sparse 3D nets, the Hungarian matching and the losses
        have data-dependent shapes and stay eager.

        heads=True (OPT-IN, not validated): also the TRAINABLE dense heads - feature projections, pixel decoder, transformer decoder,
        forward and backward - as one graph pair (train_graph.GraphedRegion; parameter gradients arrive through their ordinary
        AccumulateGrad nodes).  Measured in round 4: with the frozen stages graphed the iteration is device-bound (157 ms of kernels in
        a 158 ms iteration), so this buys nothing - and torch's multi-block reductions (bias gradients of convolutions, gradients of
        broadcast parameters) return stale values when replayed from a HIP graph on this stack (tools/graph_reduce_probe.py;
        tests/test_gpu_train.py found mask_features.bias and level_embed wrong).  The linear layers, LayerNorm and GroupNorm are safe
        (own fixed-order reductions: xm3d_column_sum, xm3d_layer_norm_bwd, xm3d_group_norm_bwd); the remaining aten reductions are not."""
        ext = self.backbone.feature_extractor.ldm_extractor
        ext.enable_train_graph(on)
        ext.enable_vae_train_path(on)
        self._head_graphs = {} if (on and heads) else None
        return self

    def _decode_heads_graphed(self, img, cond):
        from .train_graph import GraphedRegion

        x = self.normalize_images(img.float())
        feats = self.backbone.extract(x, cond)
        size = tuple(x.shape[-2:])
        key = (size, tuple(tuple(f.shape) for f in feats), tuple(f.requires_grad for f in feats))
        g = self._head_graphs.get(key)
        if g is None:
            def region(*fs):
                out = self.sem_seg_head({k: v.float() for k, v in self.backbone.project(list(fs), size, False).items()})
                for k in ("pred_masks", "mask_embed", "mask_pooled_features"):
                    out[k] = out[k].float()
                return out

            params = list(self.backbone.feature_projections.parameters()) + list(self.sem_seg_head.parameters())
            g = self._head_graphs[key] = GraphedRegion(region, feats, params)
        outputs = g(*feats)
        outputs["images"] = img.float() / 255.0
        return outputs

    def dense_forward(self, img, cond, encoded=None, fork_stream=None):
        """The static-shape part of the eval forward (rows a8-a17) in one call: dense_features + dense_heads.
        img (B,3,H,W) 0..255 on device, cond (B,768)."""
        img = img.to(cond.device)
        return self.dense_heads(img, self.dense_features(img, cond, encoded, fork_stream))

    def enable_dense_graph(self, on=True, slots=2):
        """Replay the static-shape dense branch as one HIP graph per input shape (inference only): ~2000 launch-bound
        kernels per view stop paying host launch cost.  Outputs are static buffers, valid until the next call."""
        self._dense_graphs = {} if on else None
        self._graph_slots = max(1, int(slots))
        return self

    def _graphs_for(self, img, cond, slot=0):
        """Three HIP graphs per input shape and slot: A = VAE encoder (independent of the 3D branch); B = UNet with the VAE
        decoder forked beside it inside the capture, then the projections (convolution-bound); C = pixel / transformer
        decoder, category logits, mask-CLIP (latency-bound).  Slots are independent copies (own static buffers and side
        stream) so that the next forward's front can run while this one's graph C is still executing; the event recorded
        between B and C tells the next front when the device starts to have room."""
        key = (tuple(img.shape), img.dtype, self.dense_dtype, self.channels_last)
        if (key, slot) not in self._dense_graphs:
            for sl in range(self._graph_slots):  # capture every slot now: nothing else is in flight at the first call
                s_img, s_cond = img.clone(), cond.clone()
                side, fork = torch.cuda.Stream(), torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2 if sl == 0 else 1):  # warm-up outside capture: library algorithm selection, caches
                        self.dense_forward(s_img, s_cond, self.encode_vae(s_img))
                torch.cuda.current_stream().wait_stream(side)
                ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    s_enc = self.encode_vae(s_img)
                with torch.cuda.graph(gb):
                    s_feat = self.dense_features(s_img, s_cond, s_enc, fork)
                with torch.cuda.graph(gc):
                    s_out = self.dense_heads(s_img, s_feat)
                self._dense_graphs[(key, sl)] = dict(ga=ga, gb=gb, gc=gc, img=s_img, cond=s_cond, out=s_out, side=side, b_done=None,
                                                     keep=(s_enc, s_feat, fork))
        return self._dense_graphs[(key, slot)]

    def _dense_graphed(self, img, cond):
        g = self._graphs_for(img, cond)
        g["img"].copy_(img)
        g["cond"].copy_(cond)
        g["ga"].replay()
        g["gb"].replay()
        g["gc"].replay()
        return dict(g["out"])

    def mark(self, label, stream=None):
        """timeline tracing (tools/timeline_events.py): a timed event on `stream` when tracing is on, nothing otherwise"""
        tr = getattr(self, "_trace", None)
        if tr is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            tr.append((label, ev))

    def front_stream(self):
        """The stream `pipeline.infer_scene` runs the NEXT scene's front on (voxelisation, sparse 3D nets)."""
        if getattr(self, "_front_stream", None) is None:
            # XM3D_FRONT_PRIORITY=1: a high-priority queue for the front's ~1000 small kernels (A/B switch)
            self._front_stream = torch.cuda.Stream(priority=-1) if os.environ.get("XM3D_FRONT_PRIORITY", "0") == "1" else torch.cuda.Stream()
        return self._front_stream

    # ------------------------------------------------------------------ forward
    # The eval forward has three stages.  forward() runs them back to back; pipeline.infer_scene interleaves two scenes:
    # front(i+1) is issued on a side stream right after dense(i) is launched, so the host-bound, shape-dynamic sparse
    # front of the next scene and its VAE-encoder graph overlap the long static graph of this one.
    def eval_front(self, batch_input, stream=None):
        """Stage 1 (rows a4-a7 + a8 encoder): VAE-encoder graph on its side stream, sparse 3D nets on `stream` (default:
        the current stream).  Returns the hand-over dict for eval_dense / eval_fuse."""
        sinput = batch_input["sinput"]
        dev = sinput.F.device
        img = batch_input["img"].to(dev)
        B = img.shape[0]
        inds = batch_input["inds_reconstruct"].to(dev)
        graphed = self._dense_graphs is not None and not torch.is_grad_enabled()
        front = {"graphed": graphed, "img": img, "event": None}
        tail = None
        producer = stream if stream is not None else (torch.cuda.current_stream() if img.is_cuda else None)
        if graphed:
            slot = self._slot = (getattr(self, "_slot", -1) + 1) % self._graph_slots
            g = front["g"] = self._graphs_for(img, torch.zeros(B, 768, device=dev), slot)
            g["side"].wait_stream(producer)          # the images are ready
            tail = getattr(self, "_tail_event", None) if stream is not None else None
            if tail is not None:                     # prefetch: hold back until the previous forward's conv-bound part is done
                g["side"].wait_event(tail)
            if g["b_done"] is not None:
                g["side"].wait_event(g["b_done"])    # graph B that last read this slot's buffers has finished
            with torch.cuda.stream(g["side"]):
                self.mark("A0", g["side"])
                g["img"].copy_(img)
                g["ga"].replay()
                self.mark("A1", g["side"])
        if stream is not None:
            with torch.cuda.stream(stream):
                if tail is not None:
                    stream.wait_event(tail)
                if graphed and os.environ.get("XM3D_FRONT_PARALLEL", "0") != "1":
                    # the sparse front queues up behind this forward's VAE-encoder graph: when both run beside graph C
                    # at once the front's small kernels starve (30 -> 100 ms at 20 views) and delay the next graph B
                    # (XM3D_FRONT_PARALLEL=1: A/B switch, issue it beside the encoder graph)
                    stream.wait_stream(g["side"])
                self.mark("S0", stream)
                front["pred_3d"], front["cond"], front["binary_scores"] = self.encode_3d(sinput, inds, B)
                self.mark("S1", stream)
                front["event"] = torch.cuda.Event()
                front["event"].record(stream)
        else:
            front["pred_3d"], front["cond"], front["binary_scores"] = self.encode_3d(sinput, inds, B)
        return front

    def eval_dense(self, batch_input, front):
        """Stage 2 (rows a8-a17): the static-shape dense branch on the current stream."""
        if front["event"] is not None:  # the front ran on another stream: order after it, tell the allocator about the new reader
            cur = torch.cuda.current_stream()
            cur.wait_event(front["event"])
            _record_stream((batch_input, front["pred_3d"], front["cond"], front["binary_scores"]), cur)
        if not self.prune_dead_compute:  # the reference embeds the captions in eval and never uses the result
            self.category_head.clip.embed_text(batch_input["captions"])
        if front["graphed"]:
            g = front["g"]
            cur = torch.cuda.current_stream()
            g["cond"].copy_(front["cond"])
            cur.wait_stream(g["side"])
            self.mark("B0")
            g["gb"].replay()
            # from here on this forward is latency-bound: the next forward's front (issued on side streams) starts here
            self._tail_event = torch.cuda.Event()
            self._tail_event.record(cur)
            self.mark("C0")
            g["gc"].replay()
            self.mark("C1")
            g["b_done"] = torch.cuda.Event()
            g["b_done"].record(cur)
            return dict(g["out"])
        return self.dense_forward(front["img"], front["cond"])

    def eval_fuse(self, batch_input, front, outputs):
        """Stage 3 (rows a18-a19): masks -> points, 2D/3D fusion."""
        outputs["pred_3d"] = front["pred_3d"]
        binary_pred = (torch.sigmoid(front["binary_scores"]) > 0.5).long()
        mask_cls_results = outputs["pred_logits"]
        outputs.update(self.fuse_eval(outputs, batch_input, front["binary_scores"]))
        outputs.update({"mask_cls_results": mask_cls_results, "binary_pred": binary_pred})
        return outputs

    def forward(self, batch_input):
        if self.training:
            return self.forward_train(batch_input)
        front = self.eval_front(batch_input)
        return None, self.eval_fuse(batch_input, front, self.eval_dense(batch_input, front))

    def fuse_eval_batched(self, outputs, batch_input, binary_scores):
        """fuse_eval over all batch entries at once: the same arithmetic as the per-entry loop (models/xmask3d.py:326-487),
        organised per point (Np_total, Q) with the entry index of each point (`point_view`) instead of per entry (Q, Np):
        ~25 launches for the whole batch instead of ~40 per entry.  Needs `point_offsets` (host) and `point_view` (device);
        keeps all Q mask rows (no host synchronisation).  Besides the reference's per-entry lists it returns the
        concatenated tensors (`*_cat`) that pipeline.postprocess_scene consumes."""
        cfg = self.cfg
        dev = outputs["pred_3d"].device
        masks = F.interpolate(outputs["pred_masks"], size=tuple(cfg.mask_shape), mode="bilinear", align_corners=False)
        B, Q = masks.shape[:2]
        offsets = batch_input["point_offsets"]
        sizes = [offsets[i + 1] - offsets[i] for i in range(B)]
        vid = batch_input["point_view"]
        x, y = batch_input["x_label"].to(dev), batch_input["y_label"].to(dev)
        cs = cfg.category_split
        base_cat, novel_cat = list(cs["base_category"]), list(cs["novel_category"])
        num_classes = cfg.test_ignore_label[0]
        ck = (str(dev), outputs["pred_logits"].shape[-1])
        if getattr(self, "_fuse_cols", (None,))[0] != ck:  # built once: a per-call host->device copy would stall the host
            bc, nc = torch.zeros(ck[1], dtype=torch.bool), torch.zeros(ck[1], dtype=torch.bool)
            bc[base_cat + [num_classes]] = True
            nc[novel_cat] = True
            self._fuse_cols = (ck, bc.to(dev), nc.to(dev))
        base_cols, novel_cols = self._fuse_cols[1:]
        p3d = outputs["pred_3d"]
        m3d_full = masks[vid, :, x, y].sigmoid() > 0.5                       # (Np, Q): mask q covers point p
        cover = m3d_full.float()
        weighted = torch.sigmoid(binary_scores).view(-1, 1) * cover
        # per-entry sums over the entry's own contiguous point range (deterministic reductions, no float atomics)
        cnt_q = torch.stack([cover[offsets[i]:offsets[i + 1]].sum(0) for i in range(B)])        # (B, Q)
        num_q = torch.stack([weighted[offsets[i]:offsets[i + 1]].sum(0) for i in range(B)])
        keep_full = cnt_q > 0
        is_base = (num_q / (cnt_q + 1e-10) > cfg.binary_2d_thresh).unsqueeze(-1)
        cls = outputs["pred_logits"]
        modified = torch.where(is_base, cls.masked_fill(novel_cols, -1e10), cls.masked_fill(base_cols, -1e10))
        scores = F.softmax(modified, dim=-1).max(-1)[0]                        # (B, Q)
        keep = keep_full & (scores > cfg.scores_keep_thresh)
        # pixel ownership in one pass over the logits (xm3d_mask_owner): arg-max over queries of (kept ? score : -1) * sigmoid,
        # owner = that query where its sigmoid >= 0.5 and it is kept, else -1; the reference's binary masks are owner == q
        owner = ops.mask_owner(masks.float().contiguous(), scores.float().contiguous(), keep)
        own_p = owner[vid, x, y].long()                                        # (Np,) query owning each point's pixel, or -1
        covered = own_p >= 0
        emb = outputs["mask_embed"].float()
        feat2d = torch.where(covered.view(-1, 1), emb[vid, own_p.clamp_min(0)], torch.zeros((), dtype=emb.dtype, device=dev))
        cnt = covered.to(torch.int32)
        mask_3d = own_p.view(-1, 1) == torch.arange(Q, device=dev).view(1, -1)  # (Np, Q)
        # FeatureMerger = Linear(cat(feat2d, p3d)) (models/utils/fuser.py:6-14).  feat2d takes one of only B*Q distinct rows
        # (the owning query's embedding), so its half of the product is a (B*Q, 768) GEMM + a gather; the point-wise half
        # is p3d @ W[:, 768:]^T - half the work of the concatenated form and no (Np, 1536) copy.  With bf16 head weights
        # (cast_head_weights: the benched throughput configuration) that GEMM runs in bf16 with f32 accumulation.
        lin = self.criterion.fuser.linear
        C = emb.shape[-1]
        hd = getattr(self, "heads_cast", None)
        if hd is not None:
            if getattr(self, "_fuser_w3d", None) is None or self._fuser_w3d.device != dev:
                self._fuser_w3d = lin.weight[:, C:].to(hd).contiguous()      # (out, in): the point-wise half of the merger's weight
            from .sd_model import flinear

            part3d = flinear(p3d.to(hd), self._fuser_w3d).float()            # k_gemm (bf16, f32 accumulation)
        else:
            part3d = p3d @ lin.weight[:, C:].t()
        part2d = (emb @ lin.weight[:, :C].t().float() + lin.bias.float())[vid, own_p.clamp_min(0)]   # (B, Q, 768) -> per point
        fused = torch.where(covered.view(-1, 1), part2d + part3d, p3d)
        pure3d = self.criterion.fc1(p3d)
        emb_open = outputs["mask_embed_clip"]
        return {"fused_pred_feature": list(fused.split(sizes)), "2d_pred_feature": list(feat2d.split(sizes)),
                "pure3d_pred_feature": list(pure3d.split(sizes)), "final_mask_3d": [m.t() for m in mask_3d.split(sizes)],
                "final_pred_open_embedding": [emb_open[i] for i in range(B)],
                "fused_cat": fused, "feat2d_cat": feat2d, "pure3d_cat": pure3d, "mask_3d_cat": mask_3d, "open_embedding_all": emb_open}

    def forward_train(self, batch_input):
        """models/xmask3d.py:182-305: Hungarian-matched mask losses (main + 9 aux), 3D CE losses on fused / pure-3D point
        features, caption cosine losses, binary base/novel loss; returns weighted losses."""
        cfg = self.cfg
        sinput = batch_input["sinput"]
        dev = sinput.F.device
        img = batch_input["img"].to(dev)
        B = img.shape[0]
        inds = batch_input["inds_reconstruct"].to(dev)
        batch_input = dict(batch_input)
        for k in ("x_label", "y_label", "ori_coords", "labels_3d", "binary_label_3d"):
            batch_input[k] = batch_input[k].to(dev)
        pred_3d, cond, binary_scores = self.encode_3d(sinput, inds, B)
        caption_embed = self.category_head.text_proj(self.category_head.clip.embed_text(batch_input["captions"]))
        outputs = self.encode_2d(img, cond)
        outputs["pred_3d"] = pred_3d
        binary_pred = (torch.sigmoid(binary_scores) > 0.5).long()
        label_2d = batch_input["label_2d"].to(dev)
        targets = []
        for i in range(B):  # one binary mask per label value present in the view (xmask3d.py:189-224)
            values = torch.unique(label_2d[i])
            targets.append({"labels": values.long(), "masks": (label_2d[i][None] == values.view(-1, 1, 1)).float()})
        head = self.category_head(outputs, targets)
        outputs.update(head)
        outputs["pred_logits"] = self.cal_pred_logits(outputs)
        for aux in outputs.get("aux_outputs", []):
            aux.update(head)
            aux["pred_logits"] = self.cal_pred_logits(aux)
        losses, outputs = self.criterion(outputs, targets, batch_input)
        cos = nn.CosineSimilarity()

        def caption_loss(feats):
            return (1 - cos(torch.stack([f.mean(0) for f in feats]), caption_embed)).mean()

        if cfg.caption_contra:
            losses["loss_explicit_contra"] = caption_loss(outputs["fused_pred_feature"])
        if cfg.caption_contra_2d_pre:
            losses["loss_explicit_contra_2d_pre"] = caption_loss(outputs["2d_pred_feature_pre"])
        if cfg.caption_contra_3d:
            losses["loss_explicit_contra_3d"] = caption_loss(outputs["pure3d_pred_feature"])
        labels = batch_input["binary_label_3d"]
        valid = ~torch.isin(labels, torch.tensor(self.ignore_label).to(labels))
        self.binary_loss_func.pos_weight = self.binary_loss_func.pos_weight.to(dev)
        losses["loss_binary"] = self.binary_loss_func(binary_scores[valid], labels[valid].reshape(-1, 1))
        outputs["binary_pred"] = binary_pred
        wd = self.criterion.weight_dict
        losses = {k: v * wd[k] for k, v in losses.items() if k in wd}
        return losses, outputs

    def fuse_eval(self, outputs, batch_input, binary_scores):
        """models/xmask3d.py:326-487 for every batch entry.  No host synchronisation when the batch carries
        ``point_offsets`` (per-view point ranges) and ``compact_outputs`` is False: the reference's dynamic-size outputs
        ``final_mask_3d`` (Qk,Np) / ``final_pred_open_embedding`` (Qk,768) are then returned over all Q queries with
        all-False rows for the queries that were dropped (same votes downstream)."""
        cfg = self.cfg
        dev = outputs["pred_3d"].device
        if batch_input.get("point_offsets") is not None and not batch_input.get("compact_outputs", True) \
                and batch_input.get("point_view") is not None:
            return self.fuse_eval_batched(outputs, batch_input, binary_scores)
        masks = F.interpolate(outputs["pred_masks"], size=tuple(cfg.mask_shape), mode="bilinear", align_corners=False)
        ori_coords = batch_input["ori_coords"].to(dev)
        x_all, y_all = batch_input["x_label"].to(dev), batch_input["y_label"].to(dev)
        cs = cfg.category_split
        base_cat, novel_cat = list(cs["base_category"]), list(cs["novel_category"])
        num_classes = cfg.test_ignore_label[0]
        offsets = batch_input.get("point_offsets")
        compact = batch_input.get("compact_outputs", True)
        n_scene = (len(offsets) - 1) if offsets is not None else int(ori_coords[:, 0].max().item()) + 1
        Q = masks.shape[1]
        qidx = torch.arange(Q, device=dev).view(-1, 1, 1)
        ck = (str(dev), outputs["pred_logits"].shape[-1])
        if getattr(self, "_fuse_cols", (None,))[0] != ck:  # built once: a per-call host->device copy would stall the host
            bc, nc = torch.zeros(ck[1], dtype=torch.bool), torch.zeros(ck[1], dtype=torch.bool)
            bc[base_cat + [num_classes]] = True
            nc[novel_cat] = True
            self._fuse_cols = (ck, bc.to(dev), nc.to(dev))
        base_cols, novel_cols = self._fuse_cols[1:]
        out, out2d, out3d, mask3d_l, open_l = [], [], [], [], []
        for s in range(n_scene):
            sel = slice(offsets[s], offsets[s + 1]) if offsets is not None else (ori_coords[:, 0] == s)
            x_label, y_label = x_all[sel].contiguous(), y_all[sel].contiguous()
            p3d = outputs["pred_3d"][sel].contiguous()
            emb, emb_open = outputs["mask_embed"][s], outputs["mask_embed_clip"][s]
            m = masks[s]
            cls = outputs["pred_logits"][s]
            m3d_full = m[:, x_label, y_label].sigmoid() > 0.5
            keep_full = m3d_full.sum(1) > 0
            scene_scores = torch.sigmoid(binary_scores[sel]).view(1, -1)
            cover = m3d_full.float()
            bp = (scene_scores * cover).sum(1) / (cover.sum(1) + 1e-10)
            is_base = (bp > cfg.binary_2d_thresh).view(-1, 1)
            modified = torch.where(is_base, cls.masked_fill(novel_cols, -1e10), cls.masked_fill(base_cols, -1e10))
            scores = F.softmax(modified, dim=-1).max(-1)[0]
            mask_pred = m.sigmoid()
            keep = keep_full & (scores > cfg.scores_keep_thresh)
            # queries not kept must not win the per-pixel arg-max: give them score -1 (kept scores are > 0)
            prob = torch.where(keep, scores, torch.full_like(scores, -1.0)).view(-1, 1, 1) * mask_pred
            ids = prob.argmax(0)
            final = (ids[None] == qidx) & (mask_pred >= 0.5) & keep.view(-1, 1, 1)  # all False when nothing is kept
            feat2d, cnt = ops.mask_point_fuse(final.to(torch.uint8).contiguous(), x_label, y_label, emb.float().contiguous())
            mask_3d = final[:, x_label, y_label]
            if compact:
                final_keep = final.flatten(1).any(1)
                mask_3d, open_sel = mask_3d[final_keep], emb_open[final_keep]
            else:
                open_sel = emb_open
            need = (cnt >= 1).view(-1, 1)
            fused = torch.where(need, self.criterion.fuser(feat2d, p3d), p3d)
            out.append(fused)
            out2d.append(feat2d)
            out3d.append(self.criterion.fc1(p3d))
            mask3d_l.append(mask_3d)
            open_l.append(open_sel)
        return {"fused_pred_feature": out, "2d_pred_feature": out2d, "pure3d_pred_feature": out3d, "final_mask_3d": mask3d_l,
                "final_pred_open_embedding": open_l}
