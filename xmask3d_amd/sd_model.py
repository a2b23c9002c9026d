"""Stable-Diffusion v1 autoencoder (KL-f8) and denoising UNet, restated in plain torch.

The reference reaches these through the un-vendored dependency
``stable-diffusion-sdkit==2.1.3`` (``ldm.*``; /root/reference/setup.py:33,
call sites /root/reference/models/modeling/meta_arch/ldm.py:7-10,386-490).  That
package is absent here, so the published SD-v1 architecture
(``v1-inference.yaml``: VAE ch=128, ch_mult (1,2,4,4), 2 res blocks, z=4; UNet
model_channels=320, channel_mult (1,2,4,4), 2 res blocks, attention at ds 1/2/4,
8 heads, context_dim 768) is written out again.  Module / parameter names follow
the ldm state-dict layout (``encoder.down.0.block.0.norm1.weight``,
``input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight`` ...) so that
``sd-v1-3.ckpt`` can be mapped onto it once it is available.
PARITY UNPINNED for the numerics of this file: nothing in /root/reference pins
them; tests check structure (tap shapes/strides) and GPU-vs-CPU agreement of the
same modules.

Only what the feature extractor executes is built: the encoder, the UNet, the
decoder; `taps` arguments name the blocks whose INPUT is returned as a feature.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def swish(x):
    return x * torch.sigmoid(x)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm whose f32 training forward / backward run on the HIP kernels (norm_train.LayerNormFn); torch otherwise"""

    def forward(self, x):
        from . import norm_train

        if norm_train.layer_norm_ok(x, self.weight, self.bias) and tuple(self.normalized_shape) == (x.shape[-1],):
            return norm_train.layer_norm(x, self.weight, self.bias, self.eps)
        return super().forward(x)


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm whose f32 training forward / backward run on the HIP kernels (norm_train.GroupNormActFn); torch otherwise"""

    def forward(self, x):
        from . import norm_train

        if norm_train.group_norm_ok(x, self):
            return norm_train.group_norm_act(x, self)
        return super().forward(x)


def group_norm(c, eps):
    return GroupNorm(32, c, eps=eps, affine=True)


ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2


def fused_nhwc(x):
    """inference on a ROCm device with channels-last activations: the path where conv biases / embedding terms are folded
    into the GroupNorm and residual kernels (xm3d_group_norm_nhwc shift, xm3d_bias_residual_nhwc)"""
    return (x.is_cuda and not torch.is_grad_enabled() and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 4
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 8 == 0)


_CONV_GEMM_LIBRARY = os.environ.get("XM3D_CONV_GEMM", "hip") == "library"  # A/B switch: strided / small-map / 1x1 convolutions back on MIOpen


def _packed_cg(conv):
    """(packed image, column tile, padded cout, padded f32 bias or None) of a frozen Conv2d for ops.conv_gemm, once per weight storage"""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.dtype)
    c = conv.__dict__.get("_xm3d_cg")
    if c is None or c[0] != key:
        packed, tile, n32 = ops.conv_gemm_pack_weight(w)
        bias = None
        if conv.bias is not None:
            bias = torch.zeros(n32, dtype=torch.float32, device=w.device)
            bias[:conv.out_channels] = conv.bias.detach().float()
        c = conv.__dict__["_xm3d_cg"] = (key, packed, tile, n32, bias)
    return c[1:]


def _packed_cg32(conv):
    """([packed hi, packed lo], column tile, padded cout, padded f32 bias or None) of a frozen f32 Conv2d for ops.conv_gemm_f32"""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.dtype)
    fused = gemm_f32_fused_on()
    key = key + (fused,)
    c = conv.__dict__.get("_xm3d_cg32")
    if c is None or c[0] != key:
        if fused:
            packs, _, n32, sw = ops.gemm_pack_weight_f16(w, one_scale=True)
            tile = ("sw", sw)
        else:
            packs, tile, n32 = ops.gemm_pack_weight_f16(w)
        bias = None
        if conv.bias is not None:
            bias = torch.zeros(n32, dtype=torch.float32, device=w.device)
            bias[:conv.out_channels] = conv.bias.detach().float()
        c = conv.__dict__["_xm3d_cg32"] = (key, packs, tile, n32, bias)
    return c[1:]


def own_conv(conv, x, with_bias=True, residual=None, padding=None):
    """Conv2d on the implicit-GEMM kernel (csrc/gemm.hip GF_CONV, ops.conv_gemm) - the convolutions the halo-tile kernel does not
    take: strided Downsample, the 16^2 / 8^2 UNet levels, 1x1 - or None when the call is not channels-last bf16 inference on a shape
    the kernel takes (the caller then uses torch).  padding: (top, left, bottom, right) overriding the module's symmetric padding
    (the VAE Downsample pads bottom / right only).  Unlike the library's split-K convolutions these are bit-reproducible."""
    if _CONV_GEMM_LIBRARY or torch.is_grad_enabled() or not fused_nhwc(x):
        return None
    f32 = x.dtype == torch.float32 and conv.weight.dtype == torch.float32
    if f32:
        if not gemm_f32_on():
            return None
    elif conv.weight.dtype != torch.bfloat16:
        return None
    elif x.dtype != torch.bfloat16:
        # f32 activations into a bf16-weight convolution only happen under bf16 autocast (the trainable heads' inference): autocast
        # itself would round x to bf16 here - do the same, one pass, and stay on the own kernel
        if not (x.dtype == torch.float32 and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            return None
        x = x.to(torch.bfloat16)
    k = conv.kernel_size
    if not (k[0] == k[1] and k[0] <= 3 and conv.stride[0] == conv.stride[1] and conv.dilation == (1, 1) and conv.groups == 1
            and isinstance(conv.padding, tuple) and conv.in_channels % 64 == 0 and x.shape[1] == conv.in_channels):
        return None
    pad = padding if padding is not None else (conv.padding[0], conv.padding[1], conv.padding[0], conv.padding[1])
    if f32:  # fp32 configuration: the same kernel to f32 accuracy (three passes over operands split in halves)
        packs, tile, n32, bias = _packed_cg32(conv)
        if residual is not None and n32 != conv.out_channels:
            return None
        if isinstance(tile, tuple):
            return ops.conv_gemm_f32_fused(x, packs, n32, tile[1], conv.out_channels, k[0], conv.stride[0], pad, bias=bias if with_bias else None,
                                           residual=residual)
        return ops.conv_gemm_f32(x, packs, tile, n32, conv.out_channels, k[0], conv.stride[0], pad, bias=bias if with_bias else None, residual=residual)
    packed, tile, n32, bias = _packed_cg(conv)
    if residual is not None and n32 != conv.out_channels:
        return None
    return ops.conv_gemm(x, packed, tile, n32, conv.out_channels, k[0], conv.stride[0], pad, bias=bias if with_bias else None, residual=residual)


_TRACE_LIBRARY_CONV = os.environ.get("XM3D_TRACE_CONV", "") == "1"  # diagnostic: name every convolution that still runs on the library
_traced = set()


def _trace_library_conv(conv, x):
    key = (tuple(x.shape), str(x.dtype), tuple(conv.weight.shape), conv.stride, x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())
    if key not in _traced:
        _traced.add(key)
        print(f"[xm3d] library convolution: x {key[0]} {key[1]} channels_last={key[4]} weight {key[2]} {conv.weight.dtype} stride {key[3]} "
              f"grad={torch.is_grad_enabled()}", flush=True)


class Conv2d(nn.Conv2d):
    """nn.Conv2d of the frozen nets: channels-last bf16 inference runs the own kernels (own_conv), everything else torch"""

    def forward(self, x):
        out = own_conv(self, x)
        if out is not None:
            return out
        if _TRACE_LIBRARY_CONV:
            _trace_library_conv(self, x)
        return super().forward(x)


def conv_nobias(conv: nn.Conv2d, x):
    out = own_conv(conv, x, with_bias=False)
    if out is not None:
        return out
    if _TRACE_LIBRARY_CONV:
        _trace_library_conv(conv, x)
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


# ---- fused GroupNorm -> SiLU -> conv3x3 (csrc/conv.hip, xm3d_conv3x3_nhwc): the ResnetBlock halves of both frozen nets
def fused_conv_ok(x, conv, upsample=False):
    """channels-last inference on a shape the HIP convolution takes: bf16 (xm3d_conv3x3_nhwc; XM3D_CONV=library switches it off for A/B
    runs) or f32 through the split-operand form (ops.conv3x3_f32, the fp32 configuration): by default two terms in IEEE halves per
    operand and three matrix-core passes, ~1e-6 per layer - the rounding level of an f32 convolution, per-stage parity of the fp32
    forward unchanged against the library's f32 convolutions (conv_f32_terms: the other forms, A/B switches)."""
    if not (fused_nhwc(x) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and x.shape[1] == conv.in_channels and os.environ.get("XM3D_CONV", "hip") != "library"):
        return False
    if x.dtype == torch.bfloat16:
        return ops.conv3x3_supported(x, conv.out_channels, upsample)
    return (x.dtype == torch.float32 and conv.weight.dtype == torch.float32 and conv_f32_terms() != 0
            and ops.conv3x3_f32_supported(x, conv.out_channels, upsample))


def _packed(conv):
    """(packed weight, cout tile, f32 bias) of a frozen Conv2d, built once per weight storage"""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.dtype)
    c = conv.__dict__.get("_xm3d_pack")
    if c is None or c[0] != key:
        packed, tile = ops.conv3x3_pack_weight(w)
        c = conv.__dict__["_xm3d_pack"] = (key, packed, tile, None if conv.bias is None else conv.bias.detach().float().contiguous())
    return c[1], c[2], c[3]


def conv_f32_terms():
    """XM3D_CONV_F32: "f16" (default: the two-term split in IEEE halves, three matrix-core passes, ~1e-6 per layer - the rounding level of
    an f32 convolution), "hip3" (three bf16 terms, six passes, the same accuracy: round 3's default), "hip" (two bf16 terms, three passes,
    2e-5 per layer: A/B only), "library" (torch / MIOpen f32).  -> "f16" | 3 | 2 | 0"""
    return {"f16": "f16", "hip": 2, "hip2": 2, "hip3": 3}.get(os.environ.get("XM3D_CONV_F32", "f16"), 0)


def _packed_split(conv):
    """(packed weight terms, cout tile, f32 bias) of a frozen f32 Conv2d: the bf16 split of its weight, built once per weight storage"""
    w = conv.weight
    terms = conv_f32_terms() or "f16"
    key = (w.data_ptr(), w._version, w.dtype, terms)
    c = conv.__dict__.get("_xm3d_pack_split")
    if c is None or c[0] != key:
        packs, tile = ops.conv3x3_pack_weight_split(w, terms)
        c = conv.__dict__["_xm3d_pack_split"] = (key, packs, tile, None if conv.bias is None else conv.bias.detach().float().contiguous())
    return c[1:]


def _gn_f32(norm):
    w = norm.weight
    key = (w.data_ptr(), w._version, w.dtype)
    c = norm.__dict__.get("_xm3d_f32")
    if c is None or c[0] != key:
        c = norm.__dict__["_xm3d_f32"] = (key, norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous())
    return c[1], c[2]


_GN_SEPARATE = int(os.environ.get("XM3D_CONV_GN_SEPARATE", "0"))


def gn_silu_conv3x3(norm, conv, x, pend=None, bias=None, residual=None):
    """conv(SiLU(norm(x + pend))) + bias (+ residual) in one launch; bias None = the convolution's own.  The moments of x come from
    the kernel that produced it when it left them on the tensor, and the moments of the result are left on it for the next norm."""
    if _GN_SEPARATE and pend is None and x.dtype == torch.bfloat16 and x.shape[1] >= _GN_SEPARATE:
        # A/B (XM3D_CONV_GN_SEPARATE=<min cin>): the normalisation as its own apply pass (moments from the producer's epilogue) + the plain kernel
        packed, tile, own_bias = _packed(conv)
        return ops.conv3x3(gn_act(norm, x, ACT_SILU), packed, conv.out_channels, tile, bias=own_bias if bias is None else bias, residual=residual,
                           stats_groups=32 if (conv.out_channels // 32) % 4 == 0 else None)
    gamma, beta = _gn_f32(norm)
    stats = ops.gn_stats_of(x, norm.num_groups, shift=pend)
    if x.dtype == torch.float32:
        packs, tile, own_bias = _packed_split(conv)
        return ops.conv3x3_f32(x, packs, conv.out_channels, tile, bias=own_bias if bias is None else bias,
                               gn=(stats, gamma, beta, norm.eps, norm.num_groups), residual=residual,
                               stats_groups=32 if (conv.out_channels // 32) % 4 == 0 else None,
                               in_shift=None if pend is None else pend.detach().float().contiguous())
    packed, tile, own_bias = _packed(conv)
    return ops.conv3x3(x, packed, conv.out_channels, tile, bias=own_bias if bias is None else bias, gn=(stats, gamma, beta, norm.eps, norm.num_groups),
                       residual=residual, stats_groups=32 if (conv.out_channels // 32) % 4 == 0 else None,
                       in_shift=None if pend is None else pend.detach().float().contiguous())


def plain_conv3x3(conv, x, upsample=False):
    """conv(x) + bias (x nearest-upsampled 2x first if asked), moments of the result left on it"""
    if x.dtype == torch.float32:
        packs, tile, own_bias = _packed_split(conv)
        return ops.conv3x3_f32(x, packs, conv.out_channels, tile, bias=own_bias, upsample=upsample,
                               stats_groups=32 if (conv.out_channels // 32) % 4 == 0 else None)
    packed, tile, own_bias = _packed(conv)
    return ops.conv3x3(x, packed, conv.out_channels, tile, bias=own_bias, upsample=upsample,
                       stats_groups=32 if (conv.out_channels // 32) % 4 == 0 else None)


# ---- linear layers / 1x1 convolutions on the HIP GEMM (csrc/gemm.hip, xm3d_gemm_bf16)
_GEMM_LIBRARY = os.environ.get("XM3D_GEMM", "hip") == "library"  # A/B switch: every projection back on torch (hipBLASLt)
# every eligible bf16 GEMM on k_gemm (round 4; end to end 39.49 vs 39.21 scenes/s against the round-3 rule "only where k_gemm wins alone",
# profiles/r04_bench_gemm_all_ab.log); XM3D_GEMM=wins restores that rule for A/B runs
_GEMM_ALL = os.environ.get("XM3D_GEMM", "hip") != "wins"


def gemm_ok(x, n_rows, act=None, fused_residual=False):
    """inference rows that go to the own GEMM kernels: every bf16 (K % 64, N % 32) product on k_gemm, every f32 one on the f32-accurate GEMM.
    Alone, k_gemm beats the library chain on the HBM-balanced projections (K <= 768, fused residual adds: x1.03 - 1.7) and is
    0.8 - 0.95 x hipBLASLt on the MFMA-bound ones (the 16^2 level, the wide GEGLUs, mask-CLIP: tools/gemm_bench.py,
    profiles/r03_gemm_bench.log; 30 % matrix-pipe utilisation on CLIP c_fc, profiles/r04_gemm_pmc.txt: 1.31 tile waves + ~50 % inside a
    workgroup); in the forward the two balance out (39.49 vs 39.21 scenes/s), so round 4 routes them all here - the library stays
    behind XM3D_GEMM=wins (the round-3 rule) / =library for A/B runs."""
    if _GEMM_LIBRARY or torch.is_grad_enabled() or not x.is_cuda:
        return False
    k = x.shape[-1]
    if x.dtype == torch.float32:
        # fp32 configuration: every projection on the f32-accurate GEMM (three matrix-core passes over operands split in halves,
        # ops.gemm_f32): 1.3 - 2 x the library's f32 GEMM, and bit-reproducible
        return gemm_f32_on() and x.dim() >= 2 and k % 64 == 0 and n_rows % 32 == 0 and x.numel() > 0
    if x.dtype != torch.bfloat16:
        return False
    if _GEMM_ALL:
        wins = True
    elif act == "geglu":
        wins = k <= 320
    else:
        wins = k <= 768 or (fused_residual and k <= 2560 and n_rows <= 640)
    return wins and ops.gemm_supported(x, n_rows, k)


def gemm_f32_on():
    """XM3D_GEMM_F32=library: the fp32 configuration's GEMMs / non-3x3 convolutions back on torch (hipBLASLt / MIOpen f32), for A/B runs"""
    return os.environ.get("XM3D_GEMM_F32", "hip") != "library"


def gemm_f32_fused_on():
    """the one-launch form of the f32-accurate GEMM (both half planes at one scale, three MFMAs per k-step into one accumulator:
    ops.gemm_f32_fused) unless XM3D_GEMM_F32=passes selects the three accumulating passes (wider activation range: 4e6 instead of 4094)"""
    return os.environ.get("XM3D_GEMM_F32", "hip") != "passes"


def _packed_lin(mods, act=None):
    """(packed weight, column tile, f32 bias or None, rows) of one frozen Linear / 1x1 Conv2d, or of several stacked along the
    output dimension (one GEMM for q, k, v); built once per weight storage and epilogue"""
    mods = list(mods) if isinstance(mods, (list, tuple)) else [mods]
    key = tuple((m.weight.data_ptr(), m.weight._version, m.weight.dtype) for m in mods)
    cache = mods[0].__dict__.setdefault("_xm3d_gemm", {})
    c = cache.get((len(mods), act))
    if c is None or c[0] != key:
        w = torch.cat([m.weight.detach().reshape(m.weight.shape[0], -1) for m in mods], 0)
        packed, tile = ops.gemm_pack_weight(w, act)
        bias = None
        if any(m.bias is not None for m in mods):
            bias = torch.cat([m.bias.detach().float() if m.bias is not None else torch.zeros(m.weight.shape[0], device=w.device)
                              for m in mods]).contiguous()
        c = cache[(len(mods), act)] = (key, packed, tile, bias, w.shape[0])
    return c[1:]


def _packed_lin_f32(mods):
    """([packed hi, packed lo], column tile, f32 bias or None, rows) of frozen f32 Linear / 1x1 Conv2d weights (stacked) for ops.gemm_f32"""
    mods = list(mods) if isinstance(mods, (list, tuple)) else [mods]
    key = tuple((m.weight.data_ptr(), m.weight._version, m.weight.dtype) for m in mods)
    cache = mods[0].__dict__.setdefault("_xm3d_gemm", {})
    fused = gemm_f32_fused_on()
    c = cache.get((len(mods), "f16split", fused))
    if c is None or c[0] != key:
        w = torch.cat([m.weight.detach().reshape(m.weight.shape[0], -1) for m in mods], 0)
        if fused:
            packs, _, n32, tile = ops.gemm_pack_weight_f16(w, one_scale=True)  # `tile` slot carries the weight scale ("sw", ...) for the fused form
            tile = ("sw", tile)
        else:
            packs, tile, n32 = ops.gemm_pack_weight_f16(w)
        assert n32 == w.shape[0]
        bias = None
        if any(m.bias is not None for m in mods):
            bias = torch.cat([m.bias.detach().float() if m.bias is not None else torch.zeros(m.weight.shape[0], device=w.device)
                              for m in mods]).contiguous()
        c = cache[(len(mods), "f16split", fused)] = (key, packs, tile, bias, w.shape[0])
    return c[1:]


def _gemm_f32_any(x, packs, n, tile, bias=None, act=None, residual=None):
    """ops.gemm_f32_fused when `tile` carries a weight scale (("sw", t): the one-launch form), else the three passes of ops.gemm_f32"""
    if isinstance(tile, tuple):
        return ops.gemm_f32_fused(x, packs, n, tile[1], bias=bias, act=act, residual=residual)
    return ops.gemm_f32(x, packs, n, tile, bias=bias, act=act, residual=residual)


def lin(mods, x, act=None, residual=None, with_bias=True):
    """act(x @ W^T + b) (+ residual) over the last dimension in one launch; mods: a Linear / 1x1 Conv2d or a list of them (stacked).
    with_bias False: the product alone (the caller folds the bias into a later epilogue).  f32 rows: the f32-accurate GEMM."""
    if x.dtype == torch.float32:
        packs, tile, bias, n = _packed_lin_f32(mods)
        if act == "geglu":
            return ops.geglu(_gemm_f32_any(x, packs, n, tile, bias=bias if with_bias else None)) if residual is None else \
                ops.geglu(_gemm_f32_any(x, packs, n, tile, bias=bias if with_bias else None)) + residual
        if residual is not None and not residual.is_contiguous():
            residual = residual.contiguous()
        return _gemm_f32_any(x, packs, n, tile, bias=bias if with_bias else None, act=act, residual=residual)
    packed, tile, bias, n = _packed_lin(mods, act)
    return ops.gemm(x, packed, n, tile, bias=bias if with_bias else None, act=act, residual=residual)


_FLIN = {}


def flinear(x, weight, bias=None, act=None, residual=None):
    """F.linear(x, weight, bias) -> act -> + residual on the own GEMM kernels where gemm_ok() says so (GPU inference: bf16 rows on k_gemm
    where it wins, f32 rows on the f32-accurate GEMM), else torch.  For call sites that hold weight tensors rather than modules
    (nn.MultiheadAttention's packed in_proj, slices of it); packed images are cached per (storage, shape, version)."""
    n = weight.shape[0]
    if x.dtype == torch.float32 and weight.dtype == torch.bfloat16 and x.is_cuda and not torch.is_grad_enabled() \
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        x = x.to(torch.bfloat16)  # what autocast does in front of F.linear (bf16 rows out either way); then the own kernel can take it
    if gemm_ok(x, n, act, residual is not None) and weight.dtype == x.dtype and weight.is_contiguous():
        key = (weight.data_ptr(), tuple(weight.shape), weight._version, weight.dtype, bias.data_ptr() if bias is not None else 0,
               weight.dtype == torch.float32 and gemm_f32_fused_on())
        c = _FLIN.get(key)
        if c is None:
            b32 = None if bias is None else bias.detach().float().contiguous()
            if weight.dtype == torch.float32:
                if gemm_f32_fused_on():
                    packs, _, n32, sw = ops.gemm_pack_weight_f16(weight, one_scale=True)
                    tile = ("sw", sw)
                else:
                    packs, tile, n32 = ops.gemm_pack_weight_f16(weight)
                c = (weight, packs, tile, b32) if n32 == n else None
            else:
                packed, tile = ops.gemm_pack_weight(weight.detach(), act)
                c = (weight, packed, tile, b32)
            _FLIN[key] = c
        if c is not None:
            if x.dtype == torch.float32:
                if residual is not None and not residual.is_contiguous():
                    residual = residual.contiguous()
                return _gemm_f32_any(x, c[1], n, c[2], bias=c[3], act=act, residual=residual)
            return ops.gemm(x, c[1], n, c[2], bias=c[3], act=act, residual=residual)
    from . import norm_train

    y = norm_train.LinearFn.apply(x, weight, bias) if norm_train.linear_ok(x, weight) else F.linear(x, weight, bias)
    if act == "quick_gelu":
        y = y * torch.sigmoid(1.702 * y)
    elif act == "gelu":
        y = F.gelu(y)
    elif act == "relu":
        y = F.relu(y)
    return y if residual is None else y + residual


class Linear(nn.Linear):
    """nn.Linear whose GPU-inference forward runs on the own GEMM kernels where gemm_ok() says so (flinear); torch otherwise"""

    def forward(self, x):
        return flinear(x, self.weight, self.bias)


def conv1x1_nobias(conv, x):
    """a 1x1 convolution without its bias: the token GEMM on k_gemm where that wins (channels-last bf16 inference), else the library"""
    if fused_nhwc(x) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and gemm_ok(tokens_of(x), conv.out_channels) \
            and os.environ.get("XM3D_GEMM_1X1", "hip") != "library":
        return image_of(lin(conv, tokens_of(x), with_bias=False), x.shape[2], x.shape[3])
    return conv_nobias(conv, x)


def tokens_of(x):
    """channels-last (B, C, H, W) -> its (B, H*W, C) token rows, a view"""
    return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1, x.shape[1])


def image_of(t, h, w):
    """(B, H*W, C) contiguous token rows -> the channels-last (B, C, H, W) tensor on the same storage"""
    return t.view(t.shape[0], h, w, t.shape[2]).permute(0, 3, 1, 2)


def bias_residual(skip, h, bias):
    """skip + h + bias[c] in one pass (skip may be None); the GroupNorm(32) statistics of the result are taken on the way - every
    consumer of a block output in these nets that normalises it uses 32 groups (group_norm())"""
    from . import ops

    return ops.bias_residual(skip, h, bias.to(h.dtype), stats_groups=32)


def gn_act(norm: nn.GroupNorm, x, act=ACT_NONE, shift=None, residual=None):
    """GroupNorm followed by an activation.  Inference on a ROCm device runs the fused HIP kernel (xm3d_group_norm:
    two streaming passes instead of five library kernels); under autograd, or on the CPU-baseline path, the torch ops.
    shift: (C,) or (B,C) term added to x first (a folded conv bias / embedding term).
    residual: tensor like x added after the normalisation, before the activation: act(GN(x) + residual)."""
    if x.is_cuda and not torch.is_grad_enabled() and x.dtype in (torch.float32, torch.bfloat16) and (
            (x.numel() // (x.shape[0] * x.shape[1])) % 8 == 0 or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 8 == 0)):
        from . import ops

        w, b = norm.weight, norm.bias
        if w is not None and w.dtype != x.dtype:
            w, b = w.to(x.dtype), b.to(x.dtype)
        if residual is not None and residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
        return ops.group_norm(x, norm.num_groups, w, b, norm.eps, act, shift, residual)
    if shift is not None:
        x = x + shift.to(x.dtype).reshape(-1, x.shape[1], 1, 1)
    from . import norm_train

    if norm_train.group_norm_ok(x, norm):
        # f32 training: HIP forward + backward, the activation (and its backward) inside the GroupNorm's passes
        if residual is None:
            return norm_train.group_norm_act(x, norm, act)
        y = norm_train.group_norm_act(x, norm, ACT_NONE) + residual
    else:
        y = norm(x)
        if residual is not None:
            y = y + residual
    if act == ACT_SILU:
        return y * torch.sigmoid(y)
    return F.relu(y) if act == ACT_RELU else y


# ----------------------------------------------------------------------------- VAE
class VaeResBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = group_norm(cin, 1e-6)
        self.conv1 = Conv2d(cin, cout, 3, padding=1)
        self.norm2 = group_norm(cout, 1e-6)
        self.conv2 = Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = Conv2d(cin, cout, 1)

    def forward(self, x, temb=None, pend=None):
        """pend: (C,) bias of the convolution that produced x and has NOT been added yet (conv_in / Downsample / Upsample of
        the fused channels-last path): it rides in norm1's shift and in the residual add instead of a pass of its own."""
        if fused_conv_ok(x, self.conv1) and self.conv2.out_channels % 64 == 0:
            # both halves on the HIP convolution: normalisation on the staged input tile, bias / skip in the epilogue, and the
            # moments for the next GroupNorm from the epilogue as well
            h = gn_silu_conv3x3(self.norm1, self.conv1, x, pend=pend)
            bias, skip = None, x
            if self.in_channels != self.out_channels:
                skip = conv1x1_nobias(self.nin_shortcut, x)
                bias = self.conv2.bias.float() + self.nin_shortcut.bias.float()
                if pend is not None:  # the 1x1 shortcut of a constant: W @ pend
                    bias = bias + (self.nin_shortcut.weight.flatten(1) @ pend.to(self.nin_shortcut.weight.dtype)).float()
            elif pend is not None:
                bias = self.conv2.bias.float() + pend.float()
            return gn_silu_conv3x3(self.norm2, self.conv2, h, bias=bias, residual=skip)
        if fused_nhwc(x):  # conv1's bias rides in norm2's shift, conv2's (and the shortcut's) in the residual add
            h = conv_nobias(self.conv1, gn_act(self.norm1, x, ACT_SILU, pend))
            h = conv_nobias(self.conv2, gn_act(self.norm2, h, ACT_SILU, self.conv1.bias))
            bias = self.conv2.bias
            if self.in_channels != self.out_channels:
                x = conv_nobias(self.nin_shortcut, x)
                bias = bias + self.nin_shortcut.bias
                if pend is not None:  # the 1x1 shortcut of a constant: W @ pend
                    bias = bias + self.nin_shortcut.weight.flatten(1) @ pend.to(self.nin_shortcut.weight.dtype)
            elif pend is not None:
                bias = bias + pend
            return bias_residual(x, h, bias)
        if pend is not None:
            x = x + pend.to(x.dtype).view(1, -1, 1, 1)
        h = self.conv1(gn_act(self.norm1, x, ACT_SILU))
        h = self.conv2(gn_act(self.norm2, h, ACT_SILU))
        if self.in_channels != self.out_channels:
            x = self.nin_shortcut(x)
        return x + h


class VaeAttnBlock(nn.Module):
    """single-head attention over all H*W positions"""

    def __init__(self, c):
        super().__init__()
        self.norm = group_norm(c, 1e-6)
        self.q = Conv2d(c, c, 1)
        self.k = Conv2d(c, c, 1)
        self.v = Conv2d(c, c, 1)
        self.proj_out = Conv2d(c, c, 1)

    def forward(self, x):
        h = gn_act(self.norm, x)
        b, c, hh, ww = h.shape
        if fused_nhwc(h) and h.dtype == torch.float32 and gemm_ok(tokens_of(h), 3 * c):
            # fp32 configuration: q, k, v from one f32-accurate token GEMM, softmax attention through torch (MATH), proj_out + bias + skip
            # in the epilogue of another
            qkv = lin([self.q, self.k, self.v], tokens_of(h))
            q, k, v = (qkv[..., i * c:(i + 1) * c].unsqueeze(1) for i in range(3))
            o = F.scaled_dot_product_attention(q, k, v)[:, 0]
            return image_of(lin(self.proj_out, o, residual=tokens_of(x)), hh, ww)
        if fused_nhwc(h) and h.dtype == torch.bfloat16 and gemm_ok(tokens_of(h), 3 * c) and (hh * ww) % 4 == 0 and hh * ww <= 8192 \
                and os.environ.get("XM3D_GEMM_1X1", "hip") != "library":
            # channels-last: q, k, v from ONE token GEMM (k_gemm), scores / softmax / product as below, proj_out + bias + skip in the
            # epilogue of another - no transposes, no separate bias / residual pass
            qkv = lin([self.q, self.k, self.v], tokens_of(h))
            q, k, v = (qkv[..., i * c:(i + 1) * c] for i in range(3))
            s_ = torch.bmm(q, k.transpose(1, 2), out_dtype=torch.float32)
            o = torch.bmm(ops.softmax_rows(s_, c ** -0.5), v)
            return image_of(lin(self.proj_out, o, residual=tokens_of(x)), hh, ww)
        q = self.q(h).reshape(b, 1, c, hh * ww).transpose(2, 3)
        k = self.k(h).reshape(b, 1, c, hh * ww).transpose(2, 3)
        v = self.v(h).reshape(b, 1, c, hh * ww).transpose(2, 3)
        if h.is_cuda and h.dtype == torch.bfloat16 and not torch.is_grad_enabled() and (hh * ww) % 4 == 0 and hh * ww <= 8192 \
                and fused_nhwc(h):
            # one head of c = 512 channels: two plain GEMMs around the HIP row softmax (f32 scores, bf16 probabilities) are
            # ~1 ms per call faster than the fused library kernel at this head width (pointwise.hip)
            s_ = torch.bmm(q[:, 0], k[:, 0].transpose(1, 2), out_dtype=torch.float32)
            o = torch.bmm(ops.softmax_rows(s_, c ** -0.5), v[:, 0]).unsqueeze(1)
        else:
            o = F.scaled_dot_product_attention(q, k, v)  # scale = c^-0.5
        o = o.transpose(2, 3).reshape(b, c, hh, ww)
        if fused_nhwc(x) and fused_nhwc(o):
            return bias_residual(x, conv_nobias(self.proj_out, o), self.proj_out.bias)
        return x + self.proj_out(o)


class VaeDownsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x, defer_bias=False):
        """defer_bias: return (conv output WITHOUT its bias, bias) when the fused channels-last path applies, else (out, None)"""
        if fused_nhwc(x) and not torch.is_grad_enabled():
            out = own_conv(self.conv, x, with_bias=not defer_bias, padding=(0, 0, 1, 1))  # the (0,1,0,1) padding is the kernel's border handling
            if out is not None:
                return (out, self.conv.bias) if defer_bias else out
            xp = ops.pad_bottom_right_nhwc(x, 1, 1)  # one pass instead of F.pad's fill + strided copy
            if defer_bias:
                return conv_nobias(self.conv, xp), self.conv.bias
            return self.conv(xp)
        out = self.conv(F.pad(x, (0, 1, 0, 1)))
        return (out, None) if defer_bias else out


class VaeUpsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3, padding=1)

    def forward(self, x, defer_bias=False):
        if fused_conv_ok(x, self.conv, upsample=True):  # the nearest 2x upsampling is a shift of the staging address
            out = plain_conv3x3(self.conv, x, upsample=True)
            return (out, None) if defer_bias else out
        up = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if defer_bias and fused_nhwc(up):
            return conv_nobias(self.conv, up), self.conv.bias
        out = self.conv(up)
        return (out, None) if defer_bias else out


class _Level(nn.Module):
    def __init__(self):
        super().__init__()
        self.block = nn.ModuleList()
        self.attn = nn.ModuleList()


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.block_1 = VaeResBlock(c, c)
        self.attn_1 = VaeAttnBlock(c)
        self.block_2 = VaeResBlock(c, c)

    def forward(self, h):
        return self.block_2(self.attn_1(self.block_1(h)))


class VaeEncoder(nn.Module):
    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.conv_in = Conv2d(in_channels, ch, 3, padding=1)
        self.down = nn.ModuleList()
        cin = ch
        for i, m in enumerate(ch_mult):
            lvl = _Level()
            for _ in range(num_res_blocks):
                lvl.block.append(VaeResBlock(cin, ch * m))
                cin = ch * m
            if i != len(ch_mult) - 1:
                lvl.downsample = VaeDownsample(cin)
            self.down.append(lvl)
        self.mid = _Mid(cin)
        self.norm_out = group_norm(cin, 1e-6)
        self.conv_out = Conv2d(cin, 2 * z_channels, 3, padding=1)

    def forward(self, x, taps=()):
        """taps: flat block indices (level*num_res_blocks + block) whose input is recorded."""
        feats = []
        # fused channels-last inference: the bias of conv_in / every Downsample conv is not added by a pass of its own (PyTorch-
        # ROCm adds a conv bias in a separate broadcast kernel, 0.53 ms at 20 x 512^2 x 128) but handed to the next ResBlock
        h, pend = conv_nobias(self.conv_in, x), self.conv_in.bias
        if not fused_nhwc(h):
            h, pend = h + pend.to(h.dtype).view(1, -1, 1, 1), None
        for i, lvl in enumerate(self.down):
            for j, blk in enumerate(lvl.block):
                if i * self.num_res_blocks + j in taps:
                    if pend is not None:
                        h, pend = h + pend.to(h.dtype).view(1, -1, 1, 1), None
                    feats.append(h)
                h, pend = blk(h, pend=pend), None
            if i != self.num_resolutions - 1:
                h, pend = lvl.downsample(h, defer_bias=True)
        assert pend is None
        h = self.mid(h)
        return self.conv_out(gn_act(self.norm_out, h, ACT_SILU)), feats


class VaeDecoder(nn.Module):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        cin = ch * ch_mult[-1]
        self.conv_in = Conv2d(z_channels, cin, 3, padding=1)
        self.mid = _Mid(cin)
        self.up = nn.ModuleList([_Level() for _ in ch_mult])
        for i in reversed(range(len(ch_mult))):
            for _ in range(num_res_blocks + 1):
                self.up[i].block.append(VaeResBlock(cin, ch * ch_mult[i]))
                cin = ch * ch_mult[i]
            if i != 0:
                self.up[i].upsample = VaeUpsample(cin)
        self.norm_out = group_norm(cin, 1e-6)
        self.conv_out = Conv2d(cin, out_ch, 3, padding=1)

    def forward(self, z, taps=(), stop_after_taps=False):
        """taps: flat indices over (level descending, block ascending).  With stop_after_taps the decoder
        returns as soon as the last tap has been recorded (the image itself is never used by XMask3D)."""
        feats = []
        h = self.mid(self.conv_in(z))
        idx = 0
        pend = None
        last = max(taps) if taps else -1
        for i in reversed(range(self.num_resolutions)):
            for blk in self.up[i].block:
                if idx in taps:
                    if pend is not None:
                        h, pend = h + pend.to(h.dtype).view(1, -1, 1, 1), None
                    feats.append(h)
                    if stop_after_taps and idx == last:
                        return None, feats
                h, pend = blk(h, pend=pend), None
                idx += 1
            if i != 0:
                h, pend = self.up[i].upsample(h, defer_bias=True)  # bias handed to the next level's first ResBlock
        return self.conv_out(gn_act(self.norm_out, h, ACT_SILU)), feats


class AutoencoderKL(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = VaeEncoder()
        self.decoder = VaeDecoder()
        self.quant_conv = Conv2d(8, 8, 1)
        self.post_quant_conv = Conv2d(4, 4, 1)


# ----------------------------------------------------------------------------- UNet
def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class UNetResBlock(nn.Module):
    def __init__(self, cin, emb_ch, cout):
        super().__init__()
        self.channels, self.out_channels = cin, cout
        self.in_layers = nn.Sequential(group_norm(cin, 1e-5), nn.SiLU(), Conv2d(cin, cout, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, cout))
        self.out_layers = nn.Sequential(group_norm(cout, 1e-5), nn.SiLU(), nn.Dropout(0.0), Conv2d(cout, cout, 3, padding=1))
        self.skip_connection = nn.Identity() if cin == cout else Conv2d(cin, cout, 1)

    def forward(self, x, emb):
        if fused_conv_ok(x, self.in_layers[2]) and self.out_channels % 64 == 0:
            # HIP convolutions: the timestep-embedding term is a per-sample bias of the first one
            bias1 = (self.emb_layers(emb).float() + self.in_layers[2].bias.float()).contiguous()
            h = gn_silu_conv3x3(self.in_layers[0], self.in_layers[2], x, bias=bias1)
            bias, skip = None, x
            if not isinstance(self.skip_connection, nn.Identity):
                skip = conv1x1_nobias(self.skip_connection, x)
                bias = self.out_layers[3].bias.float() + self.skip_connection.bias.float()
            return gn_silu_conv3x3(self.out_layers[0], self.out_layers[3], h, bias=bias, residual=skip)
        if fused_nhwc(x):  # first conv's bias + the embedding term ride in the second GroupNorm's shift
            h = conv_nobias(self.in_layers[2], gn_act(self.in_layers[0], x, ACT_SILU))
            shift = self.emb_layers(emb).to(h.dtype) + self.in_layers[2].bias
            h = conv_nobias(self.out_layers[3], gn_act(self.out_layers[0], h, ACT_SILU, shift))
            bias = self.out_layers[3].bias
            if not isinstance(self.skip_connection, nn.Identity):
                x = conv_nobias(self.skip_connection, x)
                bias = bias + self.skip_connection.bias
            return bias_residual(x, h, bias)
        h = self.in_layers[2](gn_act(self.in_layers[0], x, ACT_SILU))
        h = h + self.emb_layers(emb).to(h.dtype)[:, :, None, None]
        return self.skip_connection(x) + self.out_layers[3](gn_act(self.out_layers[0], h, ACT_SILU))


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        context_dim = context_dim or query_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))

    def forward(self, x, context=None):
        self_attn = context is None
        context = x if context is None else context
        b, n, _ = x.shape
        h = self.heads
        inner = self.to_q.out_features
        if gemm_ok(x, 3 * inner if self_attn else inner) and gemm_ok(context, 2 * inner) and inner % (8 * h) == 0:
            # HIP GEMMs: q, k, v of the self attention from ONE pass over x (the attention kernel reads them as column slices of the
            # fused projection), k, v of the cross attention from one pass over the context
            if self_attn:
                qkv = lin([self.to_q, self.to_k, self.to_v], x)
                q, k, v = (qkv[..., i * inner:(i + 1) * inner].unflatten(-1, (h, -1)) for i in range(3))
            else:
                q = lin(self.to_q, x).view(b, n, h, -1)
                kv = lin([self.to_k, self.to_v], context)
                k, v = (kv[..., i * inner:(i + 1) * inner].unflatten(-1, (h, -1)) for i in range(2))
            if ops.attention_supported(q, k, v):
                return lin(self.to_out[0], ops.attention(q, k, v).view(b, n, -1))
            if x.dtype == torch.float32:  # fp32 configuration: f32-accurate projections around the f32-accurate attention
                if ops.attention_f32_supported(q, k, v):
                    o = ops.attention_f32(q, k, v).view(b, n, -1)
                else:  # head dims > 64 (the <= 1024-token levels): torch's MATH attention
                    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(b, n, -1)
                return lin(self.to_out[0], o)
        q = self.to_q(x).view(b, n, h, -1)
        k = self.to_k(context).view(b, context.shape[1], h, -1)
        v = self.to_v(context).view(b, context.shape[1], h, -1)
        if ops.attention_supported(q, k, v):  # bf16 inference: HIP flash attention on the (B, N, H*D) projections in place
            return self.to_out(ops.attention(q, k, v).view(b, n, -1))
        if ops.attention_train_supported(q, k, v):  # bf16 under autograd (training through the frozen UNet): HIP forward + backward
            return self.to_out(ops.attention_train(q, k, v).reshape(b, n, -1))
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(o)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        y = self.proj(x)
        if y.is_cuda and not torch.is_grad_enabled() and y.dtype in (torch.float32, torch.bfloat16) and y.shape[-1] % 16 == 0:
            from . import ops

            return ops.geglu(y.contiguous())
        x, gate = y.chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))

    def forward(self, x, residual=None):
        """net(x) (+ residual).  Inference: GEGLU in the epilogue of its projection where that wins (no (M, 8 dim) intermediate),
        the residual add in the epilogue of the output projection"""
        g, out = self.net[0], self.net[2]
        y = lin(g.proj, x, act="geglu") if gemm_ok(x, g.proj.out_features, "geglu") else g(x)
        if gemm_ok(y, out.out_features, fused_residual=residual is not None):
            return lin(out, y, residual=residual)
        y = out(y)
        return y if residual is None else y + residual


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads, dim_head)
        self.norm1, self.norm2, self.norm3 = LayerNorm(dim), LayerNorm(dim), LayerNorm(dim)

    def forward(self, x, context):
        n1, n2, n3 = self.norm1, self.norm2, self.norm3
        if ops.layer_norm_supported(x, x.shape[-1]) and n1.weight.dtype == x.dtype:
            # inference: HIP LayerNorm (xm3d_layer_norm); the residual adds in front of norm2 / norm3 ride in their kernels
            a = self.attn1(ops.layer_norm(x, n1.weight, n1.bias, n1.eps))
            h, x = ops.layer_norm(x, n2.weight, n2.bias, n2.eps, delta=a.contiguous(), want_sum=True)
            a = self.attn2(h, context)
            h, x = ops.layer_norm(x, n3.weight, n3.bias, n3.eps, delta=a.contiguous(), want_sum=True)
            return self.ff(h, residual=x)
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    def __init__(self, c, heads, dim_head, context_dim=768):
        super().__init__()
        self.norm = group_norm(c, 1e-6)
        self.proj_in = Conv2d(c, heads * dim_head, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(heads * dim_head, heads, dim_head, context_dim)])
        self.proj_out = Conv2d(heads * dim_head, c, 1)

    def forward(self, x, context):
        b, c, h, w = x.shape
        xn = gn_act(self.norm, x)
        if fused_nhwc(xn) and gemm_ok(tokens_of(xn), self.proj_in.out_channels) and self.proj_out.out_channels % 32 == 0:
            # 1x1 convolutions of channels-last images are GEMMs over their token rows: proj_in + bias, proj_out + bias + skip
            y = lin(self.proj_in, tokens_of(xn))
            for blk in self.transformer_blocks:
                y = blk(y, context.to(y.dtype))
            if gemm_ok(y, self.proj_out.out_channels):
                return image_of(lin(self.proj_out, y, residual=tokens_of(x)), h, w)
        else:
            y = self.proj_in(xn).flatten(2).transpose(1, 2)
            for blk in self.transformer_blocks:
                y = blk(y, context.to(y.dtype))
        y = y.transpose(1, 2).reshape(b, -1, h, w)
        if fused_nhwc(x) and fused_nhwc(y):
            return bias_residual(x, conv_nobias(self.proj_out, y), self.proj_out.bias)
        return x + self.proj_out(y)


class UNetDownsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.op = Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.op(x)


class UNetUpsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv2d(c, c, 3, padding=1)

    def forward(self, x, defer_bias=False):
        if fused_conv_ok(x, self.conv, upsample=True):
            out = plain_conv3x3(self.conv, x, upsample=True)
            return (out, None) if defer_bias else out
        up = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if defer_bias and fused_nhwc(up):
            return conv_nobias(self.conv, up), self.conv.bias
        out = self.conv(up)
        return (out, None) if defer_bias else out


class TimestepSeq(nn.Sequential):
    def forward(self, x, emb, context):
        for layer in self:
            if isinstance(layer, UNetResBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    def __init__(self, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1),
                 channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768):
        super().__init__()
        self.model_channels = model_channels
        emb = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, emb), nn.SiLU(), nn.Linear(emb, emb))
        self.input_blocks = nn.ModuleList([TimestepSeq(Conv2d(in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [UNetResBlock(ch, emb, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, context_dim))
                self.input_blocks.append(TimestepSeq(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepSeq(UNetDownsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepSeq(UNetResBlock(ch, emb, ch), SpatialTransformer(ch, num_heads, ch // num_heads, context_dim),
                                        UNetResBlock(ch, emb, ch))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [UNetResBlock(ch + ich, emb, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, context_dim))
                if level and i == num_res_blocks:
                    layers.append(UNetUpsample(ch))
                    ds //= 2
                self.output_blocks.append(TimestepSeq(*layers))
        self.out = nn.Sequential(group_norm(ch, 1e-5), nn.SiLU(), Conv2d(model_channels, out_channels, 3, padding=1))

    def forward(self, x, timesteps, context, cond_emb=None, taps=(), stop_after_taps=False):
        """taps: indices of output_blocks whose (concatenated) input is recorded."""
        feats = []
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels).to(x.dtype))
        if cond_emb is not None:
            emb = emb + cond_emb.to(emb.dtype)
        hs = []
        h = x
        for m in self.input_blocks:
            h = m(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        last = max(taps) if taps else -1
        for i, m in enumerate(self.output_blocks):
            h = torch.cat([h, hs.pop()], dim=1)
            if i in taps:
                feats.append(h)
                if stop_after_taps and i == last:
                    return None, feats
            h = m(h, emb, context)
        return self.out[2](gn_act(self.out[0], h, ACT_SILU)), feats
