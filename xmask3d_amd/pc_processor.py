"""3D heads of XMask3D (mirror of /root/reference/models/modeling/meta_arch/pc_processor.py:6-60)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .mink_unet import mink_unet


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
        return self.point2text_adapter(high_x.F), self.decoder(out_x.F), idx


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
        return self.fc(self.relu(self.batch_norm(out_x.F)))
