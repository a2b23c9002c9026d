"""LayerNorm / GroupNorm(+activation) with HIP forward AND backward for the training iteration (f32, as the reference trains:
/root/reference/run/train.py:178,504-540 runs nn.LayerNorm / nn.GroupNorm through torch's autograd).  csrc/layernorm.hip + groupnorm.hip
forward, csrc/norm_bwd.hip backward: the statistics of a LayerNorm row are recomputed in the backward (only x is kept), a GroupNorm keeps x
and its f64 moments and its activation's backward rides in the same passes (the activated and the pre-activation tensor are never both
alive).  All reductions in a fixed order - no atomics.  XM3D_NORM_BWD=library puts torch's kernels back (A/B runs)."""
from __future__ import annotations

import os

import torch

from . import ops

_OFF = os.environ.get("XM3D_NORM_BWD", "hip") == "library"


def layer_norm_ok(x, weight, bias):
    C = x.shape[-1]
    affine_grad = (weight is not None and weight.requires_grad) or (bias is not None and bias.requires_grad)
    return (not _OFF and x.is_cuda and torch.is_grad_enabled() and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")
            and (weight is None or (weight.dtype == torch.float32 and bias is not None and bias.dtype == torch.float32))
            and C % 8 == 0 and C <= (1024 if affine_grad else 2048) and x.numel() > 0)


def group_norm_ok(x, norm):
    if _OFF or not (x.is_cuda and torch.is_grad_enabled() and x.dtype == torch.float32 and x.dim() >= 3) or torch.is_autocast_enabled("cuda"):
        return False
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // max(B * C, 1)
    w = norm.weight
    return (hw % 8 == 0 and B * C <= 65535 and x.numel() > 0 and x.is_contiguous() and (w is None or (w.dtype == torch.float32 and norm.bias is not None)))


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xc = x.contiguous()
        y = ops.layer_norm(xc, None if weight is None else weight.detach().contiguous(), None if bias is None else bias.detach().contiguous(), eps)
        ctx.save_for_backward(xc, weight)
        ctx.eps = eps
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        need = weight is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dx, dg, db = ops.layer_norm_bwd(x, dy.contiguous(), None if weight is None else weight.detach().contiguous(), ctx.eps, need)
        return dx, (dg if ctx.needs_input_grad[1] else None), (db if ctx.needs_input_grad[2] else None), None


class GroupNormActFn(torch.autograd.Function):
    """act(GroupNorm(x)), act 0 none / 1 SiLU / 2 ReLU, on a contiguous (B, C, ...) f32 tensor"""

    @staticmethod
    def forward(ctx, x, weight, bias, num_groups, eps, act):
        w = None if weight is None else weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        y, stats = ops.group_norm_nchw_stats(x, num_groups, w, b, eps, act)
        ctx.save_for_backward(x, weight, bias, stats)
        ctx.cfg = (num_groups, eps, act)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, bias, stats = ctx.saved_tensors
        G, eps, act = ctx.cfg
        need = weight is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dx, dg, db = ops.group_norm_bwd(x, dy.contiguous(), stats, G, None if weight is None else weight.detach().contiguous(),
                                        None if bias is None else bias.detach().contiguous(), eps, act, need)
        return dx, (dg if ctx.needs_input_grad[1] else None), (db if ctx.needs_input_grad[2] else None), None, None, None


def layer_norm(x, weight, bias, eps):
    return LayerNormFn.apply(x, weight, bias, eps)


def group_norm_act(x, norm, act=0):
    return GroupNormActFn.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps, act)


class LinearFn(torch.autograd.Function):
    """F.linear for f32 training with the bias gradient from xm3d_column_sum (fixed order; torch's multi-block column reduction does not
    replay correctly from a HIP graph on this stack, tools/graph_reduce_probe.py); the two products stay on the library's GEMMs"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        d2 = dy.reshape(-1, dy.shape[-1])
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        gx = (d2 @ weight).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = d2.t() @ x.reshape(-1, x.shape[-1]) if ctx.needs_input_grad[1] else None
        gb = ops.column_sum(d2) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def linear_ok(x, weight):
    return (not _OFF and x.is_cuda and torch.is_grad_enabled() and x.dtype == torch.float32 and weight.dtype == torch.float32
            and not torch.is_autocast_enabled("cuda") and weight.requires_grad and x.numel() > 0)
