"""CPU oracle / CPU baseline for the whole XMASK3d eval forward.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline) - never imported by
``xmask3d_amd``.

What runs where:
  * sparse 3D nets      -> oracle/spconv_oracle.py (per-offset gather-matmul-scatter, the CPU algorithm of
                           MinkowskiEngine) driven by the model's own state_dict
  * deformable attention-> oracle/msda_oracle.py (pinned to the reference's CPU path by golden vectors)
  * mask->point fusion  -> the reference's own loop structure (models/xmask3d.py:421-451,
                           models/utils/fuser.py:24-35: per query boolean-index add + count), restated
  * dense 2D nets       -> the model's torch.nn modules on CPU in fp32 with the same weights (what
                           BASELINE.md §2 names as the CPU baseline for SD / CLIP / Mask2Former parts)
PARITY UNPINNED for the dense nets and the sparse ops (see the headers of xmask3d_amd/sd_model.py,
clip_model.py and oracle/spconv_oracle.py); pinned for MSDeformAttn, the fusion loop, voxelisation.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch

from . import msda_oracle, spconv_oracle


class CpuSparseTensor:
    """Duck-typed stand-in for me_compat.SparseTensor on the CPU path."""

    def __init__(self, feats, coords):
        self.F = feats.float().cpu()
        self.C = coords.int().cpu()
        self.cache = spconv_oracle.CoordCache(self.C.numpy())


def _msda_forward_cpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    out = msda_oracle.forward(value.detach().double().numpy(), spatial_shapes.numpy(), level_start_index.numpy(),
                              sampling_loc.detach().double().numpy(), attn_weight.detach().double().numpy())
    return torch.from_numpy(out).to(value.dtype)


def _mask_point_fuse_cpu(masks_u8, x_label, y_label, embed):
    mask_3d = masks_u8[:, x_label, y_label] >= 1
    feat = torch.zeros(x_label.numel(), embed.shape[1])
    counter = torch.zeros(x_label.numel(), 1)
    for single_mask, mask_emb in zip(mask_3d, embed):
        feat[single_mask] += mask_emb
        counter[single_mask] += 1
    cnt = counter[:, 0].to(torch.int32)
    counter[counter == 0] = 1e-5
    return feat / counter, cnt


@contextlib.contextmanager
def cpu_ops(model):
    """Route the model's HIP-backed ops to the CPU oracles for the duration of the block."""
    from xmask3d_amd import msda, ops

    saved = (msda.ms_deform_attn_forward, ops.mask_point_fuse, model.pc_decoder.forward, model.pc_binary_head.forward)

    def pc_decoder(s):
        p = {k: v.detach() for k, v in model.pc_decoder.state_dict().items()}
        return spconv_oracle.pc_processor_forward(p, s.C.numpy(), s.F, model.cfg.arch_3d, cache=s.cache)

    def pc_binary(s):
        p = {k: v.detach() for k, v in model.pc_binary_head.state_dict().items()}
        return spconv_oracle.pc_binary_forward(p, s.C.numpy(), s.F, model.cfg.arch_binary_head, cache=s.cache)

    msda.ms_deform_attn_forward = _msda_forward_cpu
    ops.mask_point_fuse = _mask_point_fuse_cpu
    model.pc_decoder.forward = pc_decoder
    model.pc_binary_head.forward = pc_binary
    try:
        yield
    finally:
        msda.ms_deform_attn_forward, ops.mask_point_fuse = saved[0], saved[1]
        model.pc_decoder.forward, model.pc_binary_head.forward = saved[2], saved[3]


def forward_cpu(model, batch_input):
    """Eval forward of a CPU-resident fp32 XMASK3d on a batch whose ``sinput`` is a CpuSparseTensor."""
    assert not model.training
    with torch.no_grad(), cpu_ops(model):
        return model(batch_input)
