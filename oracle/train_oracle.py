"""CPU oracle of ONE TRAINING ITERATION of XMASK3d: forward with batch statistics, the 37 weighted losses
and their gradients, for the device-vs-oracle comparison of tests/test_gpu_train.py.

TEST INFRASTRUCTURE ONLY - never imported by ``xmask3d_amd``.

What runs where (like oracle/model_oracle.py, but differentiable and in training mode):
  * sparse 3D nets       -> oracle/spconv_oracle.py: per-offset gather / matmul / scatter with torch ops (autograd), BatchNorm with BATCH
                            statistics (biased variance, as ME.MinkowskiBatchNorm / nn.BatchNorm1d normalise in training)
  * deformable attention -> oracle/msda_oracle.py forward AND backward (numpy, pinned by the reference's own CPU path through
                            tests/golden/msda_*.npz) behind the model's own MSDeformAttnFunction
  * Hungarian matching   -> scipy.optimize.linear_sum_assignment (the reference's matcher.py:95-156 calls exactly that)
  * dense 2D nets, losses-> the model's torch.nn modules / criterion on CPU in fp32 with the same weights (their CPU branch: no HIP op)
The random point sets of the mask losses (detectron2's point sampling, criterion.py) come from ONE generator on the host in both runs:
the test installs ``criterion._rand`` to draw there.  PARITY UNPINNED for the dense nets' numerics (packages absent); the loss formulas
are pinned by tests/test_criterion.py's closed forms and the reference-generated mask_mapper fixture.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from . import msda_oracle, spconv_oracle
from .model_oracle import CpuSparseTensor


def _msda_forward_cpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    out = msda_oracle.forward(value.detach().double().numpy(), spatial_shapes.numpy(), level_start_index.numpy(),
                              sampling_loc.detach().double().numpy(), attn_weight.detach().double().numpy())
    return torch.from_numpy(out).to(value.dtype)


def _msda_backward_cpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step=64):
    gv, gl, ga = msda_oracle.backward(value.detach().double().numpy(), spatial_shapes.numpy(), level_start_index.numpy(),
                                      sampling_loc.detach().double().numpy(), attn_weight.detach().double().numpy(),
                                      grad_output.detach().double().numpy())
    return [torch.from_numpy(gv).to(value.dtype), torch.from_numpy(gl).to(sampling_loc.dtype), torch.from_numpy(ga).to(attn_weight.dtype)]


def _params(module):
    p = dict(module.named_parameters())
    p.update(dict(module.named_buffers()))
    return p


@contextlib.contextmanager
def cpu_train_ops(model):
    """Route the model's HIP-backed ops to differentiable CPU oracles (training mode) for the duration of the block."""
    from xmask3d_amd import msda

    saved = (msda.ms_deform_attn_forward, msda.ms_deform_attn_backward, model.pc_decoder.forward, model.pc_binary_head.forward)

    def pc_decoder(s):
        p = _params(model.pc_decoder)
        enc = {k[len("encoder."):]: v for k, v in p.items() if k.startswith("encoder.")}
        bott, c16, out = spconv_oracle.minkunet_forward(enc, s.C.numpy(), s.F, model.cfg.arch_3d, training=True, cache=s.cache)
        imp = bott @ p["point2text_adapter.weight"].T + p["point2text_adapter.bias"]
        x = out @ p["decoder.weight"].T + p["decoder.bias"]
        return imp, x, torch.from_numpy(c16[:, 0].astype(np.int64))

    def pc_binary(s):
        p = _params(model.pc_binary_head)
        enc = {k[len("encoder."):]: v for k, v in p.items() if k.startswith("encoder.")}
        _, _, out = spconv_oracle.minkunet_forward(enc, s.C.numpy(), s.F, model.cfg.arch_binary_head, training=True, cache=s.cache)
        x = F.batch_norm(out, None, None, p["batch_norm.weight"], p["batch_norm.bias"], True, 0.0, model.pc_binary_head.batch_norm.eps)
        return torch.relu(x) @ p["fc.weight"].T + p["fc.bias"]

    msda.ms_deform_attn_forward, msda.ms_deform_attn_backward = _msda_forward_cpu, _msda_backward_cpu  # MSDeformAttnFunction calls these
    model.pc_decoder.forward = pc_decoder
    model.pc_binary_head.forward = pc_binary
    try:
        yield
    finally:
        msda.ms_deform_attn_forward, msda.ms_deform_attn_backward = saved[0], saved[1]
        model.pc_decoder.forward, model.pc_binary_head.forward = saved[2], saved[3]


def train_step_cpu(model, batch_input):
    """-> (weighted losses dict, outputs) of a CPU-resident fp32 XMASK3d in training mode on a batch whose ``sinput`` is a
    CpuSparseTensor; the backward of the summed losses runs here too (the deformable-attention backward is an oracle as well), so the
    parameters' .grad are filled on return."""
    assert model.training
    with cpu_train_ops(model):
        losses, outputs = model(batch_input)
        sum(losses.values()).backward()
    return losses, outputs


__all__ = ["CpuSparseTensor", "cpu_train_ops", "train_step_cpu"]
