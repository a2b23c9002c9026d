"""MultiScaleDeformableAttention-compatible surface over libxm3d_hip.so.

Replaces the reference's pybind module and its Python wrappers:
  * ``ms_deform_attn_forward`` / ``ms_deform_attn_backward`` - same positional
    signature as ops/src/vision.cpp:18-21 / ms_deform_attn.h:25-66
    (under /root/reference/third_party/Mask2Former/mask2former/modeling/pixel_decoder/)
  * ``MSDeformAttnFunction`` - ops/functions/ms_deform_attn_func.py:32-49
  * ``MSDeformAttn`` module   - ops/modules/ms_deform_attn.py:34-125, same
    parameter names (sampling_offsets, attention_weights, value_proj,
    output_proj) and initialisation.

Differences from the reference on purpose: no bare ``except`` fallback to a slow
PyTorch path (ops/modules/ms_deform_attn.py:116-121) - a failing kernel raises;
CPU tensors raise "Not implemented on the CPU" exactly like the reference op.
``install_as_msda()`` registers this module as ``MultiScaleDeformableAttention``.
"""
from __future__ import annotations

import math
import sys

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.init import constant_, xavier_uniform_

from . import ops
from .sd_model import Linear, flinear


def _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, extra=()):
    names = ("value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight") + tuple(n for n, _ in extra)
    tensors = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight) + tuple(t for _, t in extra)
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    for n, t in zip(names, tensors):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
    batch = value.size(0)
    step = min(batch, int(im2col_step))
    if step > 0 and batch % step != 0:
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    if value.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"ms_deform_attn: unsupported dtype {value.dtype}")


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    # float and double both have device kernels (the reference dispatches over both, ms_deform_attn_cuda.cu:64)
    return ops.msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, (("grad_output", grad_output),))
    return list(ops.msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output))


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = Linear(d_model, d_model)
        self.output_proj = Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        H, L, P = self.n_heads, self.n_levels, self.n_points
        value = value.view(N, Len_in, H, self.d_model // H)
        fast = query.is_cuda and not torch.is_grad_enabled() and torch.is_autocast_enabled("cuda")
        if fast:
            # bf16 inference under autocast: the two projections of `query` as ONE GEMM on one bf16 copy of it (autocast casts
            # the activation once per linear), concatenated weight cached per parameter version
            so, aw = self.sampling_offsets, self.attention_weights
            key = (so.weight.data_ptr(), so.weight._version, aw.weight.data_ptr(), aw.weight._version, so.weight.device)
            if getattr(self, "_cat_key", None) != key:
                dt = torch.get_autocast_dtype("cuda")
                self._cat_w = torch.cat([so.weight.detach(), aw.weight.detach()]).to(dt).contiguous()
                self._cat_b = torch.cat([so.bias.detach(), aw.bias.detach()]).to(dt).contiguous()
                self._cat_key = key
            ow = flinear(query.to(self._cat_w.dtype), self._cat_w, self._cat_b)
            n_off = H * L * P * 2
            offsets = ow[..., :n_off].float().view(N, Len_q, H, L, P, 2)
            weights = F.softmax(ow[..., n_off:].float().view(N, Len_q, H, L * P), -1).view(N, Len_q, H, L, P)
        else:
            offsets = self.sampling_offsets(query).view(N, Len_q, H, L, P, 2)
            weights = F.softmax(self.attention_weights(query).view(N, Len_q, H, L * P), -1).view(N, Len_q, H, L, P)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            if fast:  # reference + offsets / normalizer in one pass
                loc = torch.addcdiv(reference_points[:, :, None, :, None, :].float(), offsets,
                                    normalizer[None, None, None, :, None, :].to(offsets.dtype))
            else:
                loc = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))
        out = MSDeformAttnFunction.apply(value.float().contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.float().contiguous(), weights.float().contiguous(), self.im2col_step)
        return self.output_proj(out.to(query.dtype))


def install_as_msda():
    sys.modules["MultiScaleDeformableAttention"] = sys.modules[__name__]
    return sys.modules[__name__]
