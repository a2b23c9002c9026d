"""CPU oracle for multi-scale deformable attention sampling.

TEST INFRASTRUCTURE ONLY - never imported by ``xmask3d_amd``.

Numpy restatement of the algorithm of the reference's device kernels
(/root/reference/third_party/Mask2Former/mask2former/modeling/pixel_decoder/ops/
src/cuda/ms_deform_im2col_cuda.cuh:38-89 bilinear tap rule, :92-157 its
backward, :242-304 the per-(b,q,head,channel) accumulation), vectorised over all
samples instead of one thread per output element.

Parity is PINNED: tests/golden/msda_*.npz hold outputs and all three gradients
of the reference's own CPU implementation ``ms_deform_attn_core_pytorch``
(ops/functions/ms_deform_attn_func.py:52-72, reached on CPU through
ops/modules/ms_deform_attn.py:116-121) including the toy shape / seed of the
reference's ops/test.py:21-31; tests/test_oracle_msda.py compares against them.
"""
from __future__ import annotations

import numpy as np


def _taps(shapes, level_start, loc):
    """Per-sample tap geometry.  loc (B,Lq,H,L,P,2) holds (x, y) in [0,1] units."""
    B, Lq, H, L, P, _ = loc.shape
    hs = shapes[:, 0].reshape(1, 1, 1, L, 1).astype(loc.dtype)
    ws = shapes[:, 1].reshape(1, 1, 1, L, 1).astype(loc.dtype)
    w_im = loc[..., 0] * ws - 0.5
    h_im = loc[..., 1] * hs - 0.5
    inside = (h_im > -1) & (w_im > -1) & (h_im < hs) & (w_im < ws)
    h_low = np.floor(h_im)
    w_low = np.floor(w_im)
    lh, lw = h_im - h_low, w_im - w_low
    hh, hw = 1 - lh, 1 - lw
    h_low = h_low.astype(np.int64)
    w_low = w_low.astype(np.int64)
    Hs = shapes[:, 0].reshape(1, 1, 1, L, 1)
    Ws = shapes[:, 1].reshape(1, 1, 1, L, 1)
    start = level_start.reshape(1, 1, 1, L, 1)
    taps = []
    for dy, dx, wt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
        y, x = h_low + dy, w_low + dx
        ok = inside & (y >= 0) & (y <= Hs - 1) & (x >= 0) & (x <= Ws - 1)
        idx = start + np.clip(y, 0, Hs - 1) * Ws + np.clip(x, 0, Ws - 1)
        taps.append((ok, idx, wt))
    return taps, (lh, lw, hh, hw), inside, Hs, Ws


def forward(value, shapes, level_start, loc, attn):
    """value (B,S,H,D); loc (B,Lq,H,L,P,2); attn (B,Lq,H,L,P) -> (B,Lq,H*D)."""
    B, S, H, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    taps, _, _, _, _ = _taps(shapes, level_start, loc)
    b_idx = np.arange(B).reshape(B, 1, 1, 1, 1)
    h_idx = np.arange(H).reshape(1, 1, H, 1, 1)
    out = np.zeros((B, Lq, H, D), dtype=value.dtype)
    for ok, idx, wt in taps:
        v = value[b_idx, idx, h_idx]  # (B,Lq,H,L,P,D)
        out += ((wt * attn * ok)[..., None] * v).sum(axis=(3, 4))
    return out.reshape(B, Lq, H * D)


def backward(value, shapes, level_start, loc, attn, grad_out):
    """-> (grad_value, grad_loc, grad_attn), same shapes as the inputs."""
    B, S, H, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    go = grad_out.reshape(B, Lq, H, 1, 1, D)
    taps, (lh, lw, hh, hw), inside, Hs, Ws = _taps(shapes, level_start, loc)
    b_idx = np.broadcast_to(np.arange(B).reshape(B, 1, 1, 1, 1), attn.shape)
    h_idx = np.broadcast_to(np.arange(H).reshape(1, 1, H, 1, 1), attn.shape)
    g_value = np.zeros_like(value)
    vals = []
    for ok, idx, wt in taps:
        v = value[b_idx, idx, h_idx] * ok[..., None]
        vals.append(v)
        contrib = (wt * attn * ok)[..., None] * go  # (B,Lq,H,L,P,D)
        np.add.at(g_value, (b_idx, idx, h_idx), contrib)
    v1, v2, v3, v4 = vals
    sampled = sum(t[2][..., None] * v for t, v in zip(taps, vals))
    g_attn = (go * sampled).sum(-1) * inside
    gh = (-hw[..., None] * v1 - lw[..., None] * v2 + hw[..., None] * v3 + lw[..., None] * v4)
    gw = (-hh[..., None] * v1 + hh[..., None] * v2 - lh[..., None] * v3 + lh[..., None] * v4)
    top = go * attn[..., None]
    g_loc = np.zeros_like(loc)
    g_loc[..., 0] = (Ws * (gw * top).sum(-1)) * inside
    g_loc[..., 1] = (Hs * (gh * top).sum(-1)) * inside
    return g_value, g_loc, g_attn
