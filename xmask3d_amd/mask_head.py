"""Mask2Former head of XMask3D: MSDeformAttn pixel decoder + masked transformer decoder with
pooled mask-CLIP embedding.

Same module / parameter names as the reference so checkpoints map 1:1:
  * MSDeformAttnPixelDecoder   third_party/Mask2Former/mask2former/modeling/pixel_decoder/msdeformattn.py:23-358
  * PositionEmbeddingSine      .../transformer_decoder/position_encoding.py:11-52   (pinned: tests/golden/sine_pe.npz)
  * Self/Cross-attention, FFN, MLP, MultiScaleMaskedTransformerDecoder
                               .../transformer_decoder/mask2former_transformer_decoder.py:17-204,225-365
  * ODISEMultiScaleMaskedTransformerDecoder, PseudoClassEmbed, MaskPooling, PooledMaskEmbed
                               /root/reference/models/modeling/meta_arch/odise.py:329-597
  * MaskFormerHead glue        .../meta_arch/mask_former_head.py:115-132
The deformable-attention sampling runs on the HIP kernels of xmask3d_amd.msda.
"""
from __future__ import annotations

import math

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .sd_model import ACT_NONE, ACT_RELU, Conv2d, GroupNorm, LayerNorm, Linear, flinear, fused_conv_ok, gn_act, own_conv, plain_conv3x3
from .msda import MSDeformAttn


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, x, mask=None):
        b, _, h, w = x.shape
        dev = x.device
        if mask is None and x.is_cuda:
            # no padding mask: the embedding depends on (h, w) only - computed once per shape and device (12 launches per call before:
            # ~0.35 ms of every forward), handed out as a batch-broadcast view.  Not cached from inside a graph capture (the tensor
            # would live in the capture's pool).
            key = (h, w, str(dev))
            cache = self.__dict__.setdefault("_pe_cache", {})
            pe = cache.get(key)
            if pe is None:
                pe = self._compute(1, h, w, dev, None)
                if not torch.cuda.is_current_stream_capturing():
                    cache[key] = pe
            return pe.expand(b, -1, -1, -1)
        return self._compute(b, h, w, dev, mask)

    def _compute(self, b, h, w, dev, mask):
        if mask is None:
            y_embed = torch.arange(1, h + 1, dtype=torch.float32, device=dev).view(1, h, 1).expand(b, h, w)
            x_embed = torch.arange(1, w + 1, dtype=torch.float32, device=dev).view(1, 1, w).expand(b, h, w)
        else:
            not_mask = ~mask
            y_embed = not_mask.cumsum(1, dtype=torch.float32)
            x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=dev)
        dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / self.num_pos_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def _stream_mode(x):
    """Inference over an f32 residual stream: the post-norm blocks below run as GEMMs (flinear) + ONE fused residual-add + LayerNorm launch per
    block (ops.add_layer_norm) that also emits what the next GEMMs read - instead of cast + add + LayerNorm + add + casts.
    "bf16" (autocast region of XMASK3d._decode_heads): the stream travels as (f32, bf16 copy, bf16 of stream + pos);
    "f32" (the fp32 configuration, f32-accurate GEMMs): as (f32, the same f32 tensor, f32 stream + pos);  None: the plain path."""
    if not (x.is_cuda and not torch.is_grad_enabled() and ops.add_layer_norm_supported(x, x.shape[-1]) and os.environ.get("XM3D_FUSED_LN", "hip") != "library"):
        return None
    if torch.is_autocast_enabled("cuda"):
        return "bf16" if torch.get_autocast_dtype("cuda") == torch.bfloat16 else None
    from .sd_model import gemm_f32_on

    return "f32" if gemm_f32_on() else None


def _bf16_stream_ok(x):
    return _stream_mode(x) == "bf16"


_MODE_DTYPE = {"bf16": torch.bfloat16, "f32": torch.float32}


def _add_ln(norm, x32, delta, pos=None, want=("f32", "bf16"), mode="bf16"):
    w, b = norm.weight.float(), norm.bias.float()
    if mode == "bf16":
        return ops.add_layer_norm(x32, delta, w, b, norm.eps, pos=pos, want=want)
    res = ops.add_layer_norm(x32, delta, w, b, norm.eps, pos=pos, want=tuple(k for k in want if k != "bf16"), out_dtype=torch.float32)
    res = dict(zip((k for k in want if k != "bf16"), res if isinstance(res, tuple) else (res,)))
    out = tuple(res["f32"] if k == "bf16" else res[k] for k in want)   # f32 mode: the "bf16 copy" is the stream itself
    return out[0] if len(out) == 1 else out


def _stream_attention(q4, k4, v4, bias, out_shape):
    """flash attention on (B, L, H, d) views in the stream's dtype (bf16 kernel / f32-accurate kernel); None when neither takes the shapes"""
    if ops.attention_supported(q4, k4, v4):
        o = torch.empty(out_shape, dtype=q4.dtype, device=q4.device)
        ops.attention(q4, k4, v4, bias=bias, out=o.view(out_shape[0], out_shape[1], q4.shape[2], q4.shape[3]).transpose(0, 1))
        return o
    if ops.attention_f32_supported(q4, k4, v4) and (bias is None or bias.dtype == torch.float32):
        o = torch.empty(out_shape, dtype=torch.float32, device=q4.device)
        ops.attention_f32(q4, k4, v4, bias=bias, out=o.view(out_shape[0], out_shape[1], q4.shape[2], q4.shape[3]).transpose(0, 1))
        return o
    return None


# ----------------------------------------------------------------------------- pixel decoder
class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        src2 = self.self_attn(src + pos, reference_points, src, spatial_shapes, level_start_index, padding_mask)
        src = self.norm1(src + self.dropout1(src2))
        src2 = self.linear2(self.dropout2(flinear(src, self.linear1.weight, self.linear1.bias, act="relu")))
        return self.norm2(src + self.dropout3(src2))

    def forward_stream(self, st, pos, reference_points, spatial_shapes, level_start_index, mode):
        """inference on the (f32, copy, copy + pos) stream (_stream_mode): same arithmetic, 2 fused add + LayerNorm launches per layer"""
        src32, src_b, srcpos_b = st
        delta = self.self_attn(srcpos_b, reference_points, src_b, spatial_shapes, level_start_index, None)
        src32, src_b = _add_ln(self.norm1, src32, delta.contiguous(), mode=mode)
        h = flinear(src_b, self.linear1.weight, self.linear1.bias, act="relu")
        delta = flinear(h, self.linear2.weight, self.linear2.bias)
        return _add_ln(self.norm2, src32, delta.contiguous(), pos=pos, want=("f32", "bf16", "pos"), mode=mode)


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, layer_fn, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(layer_fn() for _ in range(num_layers))
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        refs = []
        for lvl, (H_, W_) in enumerate(spatial_shapes):
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            refs.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(refs, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes_list, spatial_shapes, level_start_index, valid_ratios, pos):
        ref = self.get_reference_points(spatial_shapes_list, valid_ratios, src.device)
        out = src
        src32 = src.float().contiguous() if (src.is_cuda and not torch.is_grad_enabled()) else None
        mode = _stream_mode(src32) if src32 is not None else None
        if mode is not None and pos.shape == src.shape and all(l.linear1.weight.dtype == _MODE_DTYPE[mode] for l in self.layers):
            pos = pos.float().contiguous()
            st = (src32, src32.to(torch.bfloat16), (src32 + pos).to(torch.bfloat16)) if mode == "bf16" else (src32, src32, src32 + pos)
            for layer in self.layers:
                st = layer.forward_stream(st, pos, ref, spatial_shapes, level_start_index, mode)
            return st[0]
        for layer in self.layers:
            out = layer(out, pos, ref, spatial_shapes, level_start_index, None)
        return out


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = MSDeformAttnTransformerEncoder(
            lambda: MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, num_feature_levels, nhead, enc_n_points),
            num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.normal_(self.level_embed)

    def _shape_tensors(self, shapes, device):
        """(spatial_shapes, level_start_index) device constants, cached so graph capture sees no H2D copy"""
        key = (tuple(shapes), str(device))
        cache = self.__dict__.setdefault("_shape_cache", {})
        if key not in cache:
            ss = torch.as_tensor(shapes, dtype=torch.long, device=device)
            cache[key] = (ss, torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1])))
        return cache[key]

    def forward(self, srcs, pos_embeds):
        shapes = [(s.shape[2], s.shape[3]) for s in srcs]
        src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[l].view(1, 1, -1) for l, p in enumerate(pos_embeds)], 1)
        spatial_shapes, level_start_index = self._shape_tensors(shapes, src.device)
        valid_ratios = torch.ones(src.shape[0], len(srcs), 2, dtype=torch.float32, device=src.device)  # no padding on this path
        memory = self.encoder(src, shapes, spatial_shapes, level_start_index, valid_ratios, pos)
        return memory, shapes, level_start_index


class _Conv(nn.Conv2d):
    """detectron2 Conv2d wrapper: optional GroupNorm under ``.norm`` and ReLU."""

    def __init__(self, cin, cout, k, padding=0, bias=True, gn=False, relu=False):
        super().__init__(cin, cout, k, padding=padding, bias=bias)
        if gn:
            self.norm = GroupNorm(32, cout)
        self._gn, self._relu = gn, relu

    def _conv(self, x):
        out = _head_conv(self, x)
        return out if out is not None else nn.Conv2d.forward(self, x)

    def forward(self, x, residual=None):
        """residual: added after the norm, before the ReLU (the FPN's top-down term rides in the GroupNorm apply pass)"""
        x = self._conv(x)
        if self._gn:
            if _fused_inference(x) or (_fused_inference(x, torch.float32) and self.norm.weight.dtype == torch.float32
                                       and (residual is None or residual.dtype == torch.float32)):
                # bf16 inference under autocast: the fused channels-last GroupNorm(+residual)(+ReLU) kernel, bf16 out - the
                # next convolution would round torch's f32 GroupNorm output to bf16 anyway.  f32 (the fp32 configuration): the same
                # kernel in f32 - two passes instead of the library's five, the moments from the convolution's epilogue when it left them
                return gn_act(self.norm, x, ACT_RELU if self._relu else ACT_NONE, residual=residual)
            x = self.norm(x)
        if residual is not None:
            x = x + residual
        return F.relu_(x) if self._relu else x


def _head_conv(conv, x):
    """a convolution of the pixel decoder on the own kernels where they apply: channels-last inference with bf16 activations and bf16
    weight copies (cast_head_weights), or f32 activations and weights (the fp32 configuration: the f32-accurate forms).  The 3x3 output
    convolutions run on the halo-tile kernel - the moments of the result for the GroupNorm behind it come out of the epilogue - the 1x1
    lateral / projection convolutions on the implicit-GEMM kernel.  None: use torch."""
    if not (x.is_cuda and not torch.is_grad_enabled() and x.dim() == 4 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last)):
        return None
    if not ((x.dtype == torch.bfloat16 and conv.weight.dtype == torch.bfloat16) or (x.dtype == torch.float32 and conv.weight.dtype == torch.float32)):
        return None
    if conv.kernel_size == (3, 3) and fused_conv_ok(x, conv):
        return plain_conv3x3(conv, x)
    return own_conv(conv, x)


def _fused_inference(x, dtype=torch.bfloat16):
    return x.is_cuda and not torch.is_grad_enabled() and x.dtype == dtype and x.dim() == 4 \
        and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


def _c2_xavier_fill(m):
    nn.init.kaiming_uniform_(m.weight, a=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, input_shape, *, transformer_dropout, transformer_nheads, transformer_dim_feedforward,
                 transformer_enc_layers, conv_dim, mask_dim, norm="GN", transformer_in_features, common_stride):
        """input_shape: {name: (channels, stride)}"""
        super().__init__()
        items = sorted(input_shape.items(), key=lambda kv: kv[1][1])
        self.in_features = [k for k, _ in items]
        self.feature_channels = [v[0] for _, v in items]
        titems = [(k, v) for k, v in items if k in transformer_in_features]
        self.transformer_in_features = [k for k, _ in titems]
        self.transformer_feature_strides = [v[1] for _, v in titems]
        self.transformer_num_feature_levels = len(titems)
        self.input_proj = nn.ModuleList(
            nn.Sequential(Conv2d(v[0], conv_dim, kernel_size=1), GroupNorm(32, conv_dim)) for _, v in titems[::-1])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(conv_dim, transformer_nheads, transformer_enc_layers,
                                                              transformer_dim_feedforward, transformer_dropout,
                                                              self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = _Conv(conv_dim, mask_dim, 1)
        _c2_xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        self.num_fpn_levels = int(np.log2(min(self.transformer_feature_strides)) - np.log2(common_stride))
        lateral, output = [], []
        gn = norm == "GN"
        for idx, cin in enumerate(self.feature_channels[: self.num_fpn_levels]):
            lat = _Conv(cin, conv_dim, 1, bias=not gn, gn=gn)
            out = _Conv(conv_dim, conv_dim, 3, padding=1, bias=not gn, gn=gn, relu=True)
            _c2_xavier_fill(lat)
            _c2_xavier_fill(out)
            self.add_module(f"adapter_{idx + 1}", lat)
            self.add_module(f"layer_{idx + 1}", out)
            lateral.append(lat)
            output.append(out)
        self.lateral_convs, self.output_convs = lateral[::-1], output[::-1]

    def forward_features(self, features):
        srcs, pos = [], []
        low = torch.is_autocast_enabled("cuda") and not torch.is_grad_enabled()

        def f32(t):  # `.float()` as the reference does (msdeformattn.py:320) - except where autocast would round the result
            return t if (low and t.dtype == torch.bfloat16) else t.float()   # straight back to bf16 for the convolution

        for idx, f in enumerate(self.transformer_in_features[::-1]):
            x = f32(features[f])
            srcs.append(self.input_proj[idx](x))
            pos.append(self.pe_layer(x))
        y, shapes, level_start_index = self.transformer(srcs, pos)
        bs = y.shape[0]
        sizes = [h * w for h, w in shapes]
        out = [z.transpose(1, 2).reshape(bs, -1, h, w) for z, (h, w) in zip(torch.split(y, sizes, dim=1), shapes)]
        for idx, f in enumerate(self.in_features[: self.num_fpn_levels][::-1]):
            x = f32(features[f])
            if low and _fused_inference(x):
                # lateral conv -> GroupNorm with the top-down term as the kernel's residual: y = GN(conv(x)) + up
                with torch.autocast(device_type="cuda", enabled=False):  # bilinear resampling in bf16, not via f32 copies
                    up = F.interpolate(out[-1].to(torch.bfloat16).contiguous(memory_format=torch.channels_last), size=x.shape[-2:],
                                       mode="bilinear", align_corners=False)
                y = self.lateral_convs[idx](x, residual=up)
            else:
                cur = self.lateral_convs[idx](x)
                y = cur + F.interpolate(out[-1], size=cur.shape[-2:], mode="bilinear", align_corners=False)
            out.append(self.output_convs[idx](y))
        return self.mask_features(out[-1]), out[0], out[: self.maskformer_num_feature_levels]


# ----------------------------------------------------------------------------- transformer decoder
def _mha_train_ok(mha, x):
    return (x.is_cuda and torch.is_grad_enabled() and x.dtype == torch.float32 and mha.in_proj_weight is not None and mha.in_proj_weight.dtype == torch.float32
            and not torch.is_autocast_enabled("cuda") and mha.dropout == 0.0 and not mha.batch_first)


def _mha_train(mha, query, key, value, attn_mask=None):
    """nn.MultiheadAttention(query, key, value, attn_mask)[0] for f32 training on a device, (L, B, E) layout, with the three input projections
    and the output projection through flinear (their bias gradients then come from xm3d_column_sum: torch's own column reduction does not
    replay from a HIP graph on this stack) and the attention itself through scaled_dot_product_attention (exact f32 math backend).
    attn_mask: (B * heads, Lq, Lk) bool, True = may NOT attend (the module's convention)."""
    E, H = mha.embed_dim, mha.num_heads
    w, b = mha.in_proj_weight, mha.in_proj_bias
    q = flinear(query, w[:E], None if b is None else b[:E])
    k = flinear(key, w[E:2 * E], None if b is None else b[E:2 * E])
    v = flinear(value, w[2 * E:], None if b is None else b[2 * E:])
    Lq, B = q.shape[:2]
    Lk = k.shape[0]
    q, k, v = (t.reshape(t.shape[0], B, H, E // H).permute(1, 2, 0, 3) for t in (q, k, v))
    allowed = None if attn_mask is None else ~attn_mask.view(B, H, Lq, Lk)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=allowed)
    return flinear(o.permute(2, 0, 1, 3).reshape(Lq, B, E), mha.out_proj.weight, mha.out_proj.bias)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, query_pos=None):
        q = k = tgt + query_pos
        mha = self.self_attn
        if tgt.is_cuda and not torch.is_grad_enabled() and (torch.is_autocast_enabled("cuda") or mha.in_proj_weight.dtype == torch.float32):
            # bf16 inference (autocast region of XMASK3d._decode_heads): the projections nn.MultiheadAttention runs, HIP flash
            # attention in between (xm3d_attention_fwd)
            E, H = mha.embed_dim, mha.num_heads
            w, b = mha.in_proj_weight, mha.in_proj_bias
            L, B = tgt.shape[:2]
            qk = flinear(q, w[: 2 * E], b[: 2 * E])                       # (L, B, 2E): q and k share their input
            v = flinear(tgt, w[2 * E:], b[2 * E:])
            q4 = qk[..., :E].unflatten(-1, (H, E // H)).transpose(0, 1)    # (B, L, H, d) views
            k4 = qk[..., E:].unflatten(-1, (H, E // H)).transpose(0, 1)
            v4 = v.view(L, B, H, E // H).transpose(0, 1)
            if ops.attention_supported(q4, k4, v4):
                o = torch.empty((L, B, E), dtype=q4.dtype, device=tgt.device)
                ops.attention(q4, k4, v4, out=o.view(L, B, H, E // H).transpose(0, 1))
                return self.norm(tgt + flinear(o, mha.out_proj.weight, mha.out_proj.bias).float())  # (.float(): a mixed f32 + bf16 add of this size takes the 35 us generic kernel)
            if ops.attention_f32_supported(q4, k4, v4):  # fp32 configuration: the f32-accurate flash attention, same data flow
                o = torch.empty((L, B, E), dtype=torch.float32, device=tgt.device)
                ops.attention_f32(q4, k4, v4, out=o.view(L, B, H, E // H).transpose(0, 1))
                return self.norm(tgt + flinear(o, mha.out_proj.weight, mha.out_proj.bias))
        if _mha_train_ok(mha, tgt):
            return self.norm(tgt + _mha_train(mha, q, k, tgt))
        return self.norm(tgt + self.self_attn(q, k, value=tgt, need_weights=False)[0])

    def forward_stream(self, st, query_pos, mode):
        """inference on the (f32, copy, copy + query_pos) stream; None when no flash-attention kernel takes the shapes"""
        tgt32, tgt_b, tgtpos_b = st
        mha = self.self_attn
        E, H = mha.embed_dim, mha.num_heads
        w, b = mha.in_proj_weight, mha.in_proj_bias
        L, B = tgt_b.shape[:2]
        qk = flinear(tgtpos_b, w[: 2 * E], b[: 2 * E])
        v = flinear(tgt_b, w[2 * E:], b[2 * E:])
        q4 = qk[..., :E].unflatten(-1, (H, E // H)).transpose(0, 1)
        k4 = qk[..., E:].unflatten(-1, (H, E // H)).transpose(0, 1)
        v4 = v.view(L, B, H, E // H).transpose(0, 1)
        o = _stream_attention(q4, k4, v4, None, (L, B, E))
        if o is None:
            return None
        delta = flinear(o, mha.out_proj.weight, mha.out_proj.bias)
        return _add_ln(self.norm, tgt32, delta, pos=query_pos, want=("f32", "bf16", "pos"), mode=mode)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, memory, memory_mask=None, pos=None, query_pos=None, key=None, memory_bias=None):
        """key: memory + pos when the caller has it already (the same level feeds three layers).
        memory_bias (B, Q, K) additive float mask (xm3d_attn_mask_bias) instead of the boolean memory_mask (B*heads, Q, K):
        inference path, the same projections / scaled_dot_product_attention / out-projection nn.MultiheadAttention runs,
        with the bias broadcast over the heads instead of replicated and converted per call."""
        key = memory + pos if key is None else key
        if memory_bias is None:
            if _mha_train_ok(self.multihead_attn, tgt) and (memory_mask is None or memory_mask.dtype == torch.bool):
                return self.norm(tgt + _mha_train(self.multihead_attn, tgt + query_pos, key, memory, memory_mask))
            tgt2 = self.multihead_attn(query=tgt + query_pos, key=key, value=memory, attn_mask=memory_mask, need_weights=False)[0]
            return self.norm(tgt + tgt2)
        mha = self.multihead_attn
        E, H = mha.embed_dim, mha.num_heads
        w, b = mha.in_proj_weight, mha.in_proj_bias
        q = flinear(tgt + query_pos, w[:E], b[:E])
        k = flinear(key, w[E:2 * E], b[E:2 * E])
        v = flinear(memory, w[2 * E:], b[2 * E:])
        Lq, B = q.shape[:2]
        Lk = k.shape[0]
        q4, k4, v4 = (t.view(t.shape[0], B, H, E // H).transpose(0, 1) for t in (q, k, v))  # (B, L, H, d) views of the (L, B, E) rows
        if ops.attention_supported(q4, k4, v4):
            # bf16 inference: HIP flash attention on the projections in place, the additive mask shared by the heads, the
            # output written straight in (Lq, B, E) order for the out-projection
            o = torch.empty((Lq, B, E), dtype=q.dtype, device=q.device)
            ops.attention(q4, k4, v4, bias=memory_bias.view(B, 1, Lq, Lk), out=o.view(Lq, B, H, E // H).transpose(0, 1))
            return self.norm(tgt + flinear(o, mha.out_proj.weight, mha.out_proj.bias).float())  # (.float(): a mixed f32 + bf16 add of this size takes the 35 us generic kernel)
        if ops.attention_f32_supported(q4, k4, v4) and memory_bias.dtype == torch.float32:  # fp32 configuration
            o = torch.empty((Lq, B, E), dtype=torch.float32, device=q.device)
            ops.attention_f32(q4, k4, v4, bias=memory_bias.view(B, 1, Lq, Lk), out=o.view(Lq, B, H, E // H).transpose(0, 1))
            return self.norm(tgt + flinear(o, mha.out_proj.weight, mha.out_proj.bias))
        q = q.view(Lq, B, H, E // H).permute(1, 2, 0, 3)
        k = k.view(Lk, B, H, E // H).permute(1, 2, 0, 3)
        v = v.view(Lk, B, H, E // H).permute(1, 2, 0, 3)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=memory_bias.view(B, 1, Lq, Lk).to(q.dtype))
        tgt2 = flinear(o.permute(2, 0, 1, 3).reshape(Lq, B, E), mha.out_proj.weight, mha.out_proj.bias)
        return self.norm(tgt + tgt2)

    def forward_stream(self, st, memory_b, key_b, memory_bias, query_pos, mode):
        """inference on the (f32, copy, copy + query_pos) stream; memory_b / key_b: contiguous (Lk, B, E) copies of the level's memory and
        memory + pos in the stream's GEMM dtype (made once per level, three layers read each).  None when no flash-attention kernel applies."""
        tgt32, tgt_b, tgtpos_b = st
        mha = self.multihead_attn
        E, H = mha.embed_dim, mha.num_heads
        w, b = mha.in_proj_weight, mha.in_proj_bias
        q = flinear(tgtpos_b, w[:E], b[:E])
        k = flinear(key_b, w[E:2 * E], b[E:2 * E])
        v = flinear(memory_b, w[2 * E:], b[2 * E:])
        Lq, B = q.shape[:2]
        Lk = k.shape[0]
        q4, k4, v4 = (t.view(t.shape[0], B, H, E // H).transpose(0, 1) for t in (q, k, v))
        o = _stream_attention(q4, k4, v4, memory_bias.view(B, 1, Lq, Lk), (Lq, B, E))
        if o is None:
            return None
        delta = flinear(o, mha.out_proj.weight, mha.out_proj.bias)
        return _add_ln(self.norm, tgt32, delta, pos=query_pos, want=("f32", "bf16", "pos"), mode=mode)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048):
        super().__init__()
        self.linear1 = Linear(d_model, dim_feedforward)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm = LayerNorm(d_model)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt):
        return self.norm(tgt + self.linear2(flinear(tgt, self.linear1.weight, self.linear1.bias, act="relu")).float())

    def forward_stream(self, st, query_pos, mode):
        tgt32, tgt_b, _ = st
        h = flinear(tgt_b, self.linear1.weight, self.linear1.bias, act="relu")
        delta = flinear(h, self.linear2.weight, self.linear2.bias)
        return _add_ln(self.norm, tgt32, delta, pos=query_pos, want=("f32", "bf16", "pos"), mode=mode)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):  # (the ReLU rides in the GEMM epilogue where the own kernel runs)
            x = flinear(x, layer.weight, layer.bias, act="relu") if i < self.num_layers - 1 else layer(x)
        return x


class PseudoClassEmbed(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.num_classes = num_classes

    def forward(self, x):
        fg = torch.ones((*x.shape[:-1], self.num_classes), dtype=x.dtype, device=x.device)
        bg = torch.zeros((*x.shape[:-1], 1), dtype=x.dtype, device=x.device)
        return torch.cat([fg, bg], dim=-1)


class MaskPooling(nn.Module):
    def __init__(self, hard_pooling=True, mask_threshold=0.5):
        super().__init__()
        self.hard_pooling, self.mask_threshold = hard_pooling, mask_threshold

    def forward(self, x, mask):
        if self.hard_pooling and self.mask_threshold == 0.5 and x.is_cuda and not torch.is_grad_enabled() and mask.dtype == torch.bfloat16 \
                and mask.is_contiguous() and mask.shape[1] <= 64 and x.shape[1] == 256 and (x.shape[2] * x.shape[3]) % 16 == 0 \
                and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and x.dim() == 4:
            from . import ops

            # sigmoid(m) > 0.5 <=> m > 0: threshold in registers, masked feature sums on the matrix cores (xm3d_mask_pool), f32 mean
            return {"mask_pooled_features": ops.mask_pool(mask.detach(), x)}
        mask = mask.detach().sigmoid()
        if self.hard_pooling:
            mask = (mask > self.mask_threshold).to(mask.dtype)
        denorm = mask.sum(dim=(-1, -2), keepdim=True) + 1e-8
        return {"mask_pooled_features": torch.einsum("bchw,bqhw->bqc", x, mask / denorm)}


class PooledMaskEmbed(nn.Module):
    def __init__(self, hidden_dim, mask_dim, projection_dim, temperature=0.07):
        super().__init__()
        self.pool_proj = nn.Sequential(LayerNorm(hidden_dim), Linear(hidden_dim, hidden_dim))
        self.mask_embed = nn.Sequential(LayerNorm(mask_dim), MLP(mask_dim, hidden_dim, projection_dim, 3))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / temperature))
        self.mask_pooling = MaskPooling()

    def forward(self, decoder_output, input_mask_embed, mask_features, pred_logits, pred_masks):
        pooled = self.mask_pooling(mask_features, pred_masks)["mask_pooled_features"]
        pooled = self.pool_proj(pooled) + decoder_output
        return {"mask_embed": self.mask_embed(pooled), "mask_pooled_features": pooled,
                "logit_scale": torch.clamp(self.logit_scale.exp(), max=100)}


def bilinear_down(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) for the case this decoder meets: shrinking by an even
    integer factor s.  The source coordinate of output pixel i is s*i + s/2 - 1/2, i.e. exactly half way between input
    pixels s*i + s/2 - 1 and s*i + s/2: both weights are 0.5, so the result is the mean of the central 2x2 of every s x s
    block.  Scaling by 0.5 commutes with rounding, so summed in the library kernel's order - row pairs first on the
    device, left to right on the host, f32 accumulation for 16-bit inputs - the result is bit-identical
    (tests/test_gpu_msda_fuse.py, tests/test_host.py) at a fraction of the cost: the device kernel parallelises over output
    pixels only and loops over all B*Q maps."""
    H, W = x.shape[-2:]
    h, w = int(size[0]), int(size[1])
    if torch.is_grad_enabled() or H % h or W % w or (H // h) % 2 or (W // w) % 2:
        return F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
    sh, sw = H // h, W // w
    oh, ow = sh // 2 - 1, sw // 2 - 1
    xf = x.float() if x.dtype in (torch.bfloat16, torch.float16) else x
    a, b = xf[..., oh::sh, ow::sw], xf[..., oh::sh, ow + 1::sw]
    c, d = xf[..., oh + 1::sh, ow::sw], xf[..., oh + 1::sh, ow + 1::sw]
    y = ((a + b) + (c + d)) * 0.25 if x.is_cuda else (a + b + c + d) * 0.25
    return y.to(x.dtype)


class ODISEMultiScaleMaskedTransformerDecoder(nn.Module):
    def __init__(self, *, in_channels, mask_classification=True, num_classes, hidden_dim, num_queries, nheads,
                 dim_feedforward, dec_layers, pre_norm, mask_dim, enforce_input_project, class_embed=None,
                 post_mask_embed=None):
        super().__init__()
        assert mask_classification and not pre_norm
        self.mask_classification = True
        self.pe_layer = PositionEmbeddingSine(hidden_dim // 2, normalize=True)
        self.num_heads, self.num_layers = nheads, dec_layers
        self.transformer_self_attention_layers = nn.ModuleList(SelfAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_cross_attention_layers = nn.ModuleList(CrossAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_ffn_layers = nn.ModuleList(FFNLayer(hidden_dim, dim_feedforward) for _ in range(dec_layers))
        self.decoder_norm = LayerNorm(hidden_dim)
        self.num_queries = num_queries
        self.query_feat = nn.Embedding(num_queries, hidden_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.num_feature_levels = 3
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                conv = nn.Conv2d(in_channels, hidden_dim, kernel_size=1)
                _c2_xavier_fill(conv)
                self.input_proj.append(conv)
            else:
                self.input_proj.append(nn.Sequential())
        self.class_embed = class_embed if class_embed is not None else Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self.post_mask_embed = post_mask_embed
        # eval only: the pooled mask-CLIP embedding of the 9 intermediate layers feeds nothing but the training losses;
        # the reference computes it anyway (odise.py:445-491).  True = skip it outside training (like SURVEY F7).
        self.prune_aux_embed = False

    def forward(self, x, mask_features, mask=None):
        assert len(x) == self.num_feature_levels
        src, pos, size_list = [], [], []
        for i in range(self.num_feature_levels):
            size_list.append(x[i].shape[-2:])
            pos.append(self.pe_layer(x[i], None).flatten(2).permute(2, 0, 1))
            src.append((self.input_proj[i](x[i]).flatten(2) + self.level_embed.weight[i][None, :, None]).permute(2, 0, 1))
        bs = src[0].shape[1]
        keys = [s_ + p_ for s_, p_ in zip(src, pos)]  # once per level instead of once per layer (same values)
        query_embed = self.query_embed.weight.unsqueeze(1).repeat(1, bs, 1)
        output = self.query_feat.weight.unsqueeze(1).repeat(1, bs, 1)
        cls_l, mask_l, extra_l = [], [], []
        skip = self.prune_aux_embed and not self.training
        c, m, attn_mask, e = self.forward_prediction_heads(output, mask_features, size_list[0], not skip)
        cls_l.append(c), mask_l.append(m), extra_l.append(e)
        st = mem_b = key_b = None
        mode = _stream_mode(output) if attn_mask.dtype != torch.bool else None
        if mode is not None and self.transformer_ffn_layers[0].linear1.weight.dtype == _MODE_DTYPE[mode]:
            query_embed = query_embed.contiguous()
            dt = _MODE_DTYPE[mode]
            st = (output, output.to(dt), (output + query_embed).to(dt))
            mem_b = [s_.to(dt).contiguous() for s_ in src]    # once per level: three layers read each
            key_b = [k_.to(dt).contiguous() for k_ in keys]
        for i in range(self.num_layers):
            lvl = i % self.num_feature_levels
            if st is not None and attn_mask.dtype != torch.bool:
                # bf16 inference: the layer triple on the (f32, bf16, bf16 + query_pos) stream - one fused add + LayerNorm per block
                s1 = self.transformer_cross_attention_layers[i].forward_stream(st, mem_b[lvl], key_b[lvl], attn_mask, query_embed, mode)
                s2 = self.transformer_self_attention_layers[i].forward_stream(s1, query_embed, mode) if s1 is not None else None
                if s2 is not None:
                    st = self.transformer_ffn_layers[i].forward_stream(s2, query_embed, mode)
                    output = st[0]
                    c, m, attn_mask, e = self.forward_prediction_heads(output, mask_features, size_list[(i + 1) % self.num_feature_levels],
                                                                       not skip or i == self.num_layers - 1)
                    cls_l.append(c), mask_l.append(m), extra_l.append(e)
                    continue
                st = None  # a shape the kernels do not take: the plain path from here on
            if attn_mask.dtype == torch.bool:
                # a query whose mask is empty everywhere attends to everything (odise.py:395)
                full = attn_mask.all(dim=-1, keepdim=True)
                attn_mask = attn_mask & ~full
                output = self.transformer_cross_attention_layers[i](output, src[lvl], memory_mask=attn_mask, pos=pos[lvl],
                                                                    query_pos=query_embed, key=keys[lvl])
            else:  # additive bias from xm3d_attn_mask_bias (that rule is applied inside the kernel)
                output = self.transformer_cross_attention_layers[i](output, src[lvl], memory_bias=attn_mask, pos=pos[lvl],
                                                                    query_pos=query_embed, key=keys[lvl])
            output = self.transformer_self_attention_layers[i](output, query_pos=query_embed)
            output = self.transformer_ffn_layers[i](output)
            c, m, attn_mask, e = self.forward_prediction_heads(output, mask_features, size_list[(i + 1) % self.num_feature_levels],
                                                               not skip or i == self.num_layers - 1)
            cls_l.append(c), mask_l.append(m), extra_l.append(e)
        out = {"pred_logits": cls_l[-1], "pred_masks": mask_l[-1],
               "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(cls_l[:-1], mask_l[:-1])]}
        for k in extra_l[-1]:
            out[k] = extra_l[-1][k]
            for i in range(len(extra_l) - 1):
                if k in extra_l[i]:
                    out["aux_outputs"][i][k] = extra_l[i][k]
        return out

    def forward_prediction_heads(self, output, mask_features, attn_mask_target_size, with_embed=True):
        hmode = _stream_mode(output)
        if hmode is not None and self.mask_embed.layers[0].weight.dtype == _MODE_DTYPE[hmode]:
            # inference: the fused LayerNorm hands the f32 rows (pooled-embedding residual) and the GEMMs' input (bf16 copy / the same f32 rows)
            dec32, dec_b = _add_ln(self.decoder_norm, output, None, mode=hmode)
            decoder_output, dec_in = dec32.transpose(0, 1), dec_b.transpose(0, 1).contiguous()
        else:
            decoder_output = dec_in = self.decoder_norm(output).transpose(0, 1)
        # (the class logits of a layer whose pooled embedding is pruned feed nothing either: same switch)
        outputs_class = self.class_embed(dec_in) if with_embed else None
        mask_embed = self.mask_embed(dec_in)
        extra = {}
        if mask_embed.is_cuda and not torch.is_grad_enabled() and os.environ.get("XM3D_MASK_HEADS", "hip") != "library":
            from . import ops

            qdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else mask_embed.dtype
            if qdt == torch.bfloat16 and ops.mask_heads_supported(mask_embed, mask_features, attn_mask_target_size):
                # fused heads (csrc/maskhead.hip): logits on the matrix cores + the attention bias from them in one launch; a layer
                # whose embeddings are pruned (with_embed False: evaluation, 9 of 10 layers) needs only the bias - its logits are
                # never written, and only the rows the bias reads are computed.  bf16 logits as under autocast.
                outputs_mask, bias = ops.mask_logits_bias(mask_embed, mask_features, attn_mask_target_size, want_logits=with_embed, bias_dtype=qdt)
                if self.post_mask_embed is not None and with_embed:
                    extra.update(self.post_mask_embed(decoder_output, mask_embed, mask_features, outputs_class, outputs_mask))
                return outputs_class, outputs_mask, bias, extra
        outputs_mask = torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features)
        if self.post_mask_embed is not None and with_embed:
            extra.update(self.post_mask_embed(decoder_output, mask_embed, mask_features, outputs_class, outputs_mask))
        if outputs_mask.is_cuda and not torch.is_grad_enabled() and outputs_mask.dtype in (torch.float32, torch.bfloat16):
            from . import ops

            if ops.attn_mask_bias_supported(outputs_mask.shape, attn_mask_target_size):
                qdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
                if qdt in (torch.float32, torch.bfloat16):  # one launch: shrink, threshold, empty-mask rule, additive bias
                    return outputs_class, outputs_mask, ops.attn_mask_bias(outputs_mask, attn_mask_target_size, qdt), extra
        attn_mask = bilinear_down(outputs_mask, attn_mask_target_size)
        attn_mask = (attn_mask.sigmoid().flatten(2).unsqueeze(1).repeat(1, self.num_heads, 1, 1).flatten(0, 1) < 0.5).bool()
        return outputs_class, outputs_mask, attn_mask.detach(), extra


class MaskFormerHead(nn.Module):
    def __init__(self, input_shape=None, *, num_classes, pixel_decoder, loss_weight=1.0, ignore_value=-1,
                 transformer_predictor, transformer_in_feature):
        super().__init__()
        self.ignore_value, self.loss_weight = ignore_value, loss_weight
        self.pixel_decoder, self.predictor = pixel_decoder, transformer_predictor
        self.transformer_in_feature, self.num_classes = transformer_in_feature, num_classes
        assert transformer_in_feature == "multi_scale_pixel_decoder"

    def forward(self, features, mask=None):
        mask_features, _, multi_scale_features = self.pixel_decoder.forward_features(features)
        return self.predictor(multi_scale_features, mask_features, mask)
