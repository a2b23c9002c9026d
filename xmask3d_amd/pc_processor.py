"""3D heads of XMask3D (mirror of /root/reference/models/modeling/meta_arch/pc_processor.py:6-60)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .mink_unet import mink_unet


def _head_linear(lin, x):
    """nn.Linear of a 3D head on the net's features.  bf16 features (the bf16 configuration's sparse branch, inference): the product in
    bf16 on a bf16 copy of the trainable weight (k_gemm where it wins), f32 rows out; else the module itself on f32 rows."""
    if x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and x.is_cuda:
        from .sd_model import flinear

        key = (lin.weight.data_ptr(), lin.weight._version)
        c = lin.__dict__.get("_xm3d_bf16")
        if c is None or c[0] != key:
            c = lin.__dict__["_xm3d_bf16"] = (key, lin.weight.detach().to(torch.bfloat16).contiguous(),
                                              None if lin.bias is None else lin.bias.detach().to(torch.bfloat16).contiguous())
        return flinear(x, c[1], c[2]).float()
    return lin(x if x.dtype == lin.weight.dtype else x.float())


class PC_Processor(nn.Module):
    """MinkUNet34C -> (implicit caption rows (N16,768), per-voxel features (N1,768), batch ids (N16,))."""

    def __init__(self, adapter_proj_out_dim=768, decoder_proj_out_dim=768, last_dim=256, arch_3d="MinkUNet34C"):
        super().__init__()
        self.adapter_proj_out_dim = adapter_proj_out_dim
        self.encoder = mink_unet(in_channels=3, out_channels=last_dim, D=3, arch=arch_3d)
        self.point2text_adapter = nn.Linear(last_dim, adapter_proj_out_dim, bias=True)
        self.decoder = nn.Linear(last_dim, decoder_proj_out_dim, bias=True)

    def forward(self, x):
        high_x, out_x = self.encoder(x)
        idx = high_x.C[:, 0]
        return _head_linear(self.point2text_adapter, high_x.F), _head_linear(self.decoder, out_x.F), idx


class PC_Binary_Processor(nn.Module):
    """MinkUNet18A -> BatchNorm1d -> ReLU -> Linear(256,1): base/novel logit per voxel."""

    def __init__(self, in_channels=3, out_channels=256, arch_3d="MinkUNet18A"):
        super().__init__()
        self.encoder = mink_unet(in_channels=in_channels, out_channels=out_channels, D=3, arch=arch_3d)
        self.batch_norm = nn.BatchNorm1d(out_channels)
        self.relu = nn.ReLU()
        self.fc = nn.Linear(out_channels, 1)

    def forward(self, x):
        _, out_x = self.encoder(x)
        return self.fc(self.relu(self.batch_norm(out_x.F.float())))
