"""Sparse 3D U-Nets of the XMask3D path, built on the ME-compatible HIP operator surface.

Table-driven restatement of the topology in
/root/reference/models/modeling/meta_arch/mink_unet.py:44-178 (+ resnet_base.py:55-96):
module names, parameter names and shapes match, so the released checkpoints'
``pc_decoder.encoder.*`` / ``pc_binary_head.encoder.*`` keys load unchanged.
"""
from __future__ import annotations

import torch.nn as nn

from . import me_compat as ME

ARCHS = {
    # name: (LAYERS, PLANES)
    "MinkUNet14A": ((1,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet14B": ((1,) * 8, (32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet14C": ((1,) * 8, (32, 64, 128, 256, 192, 192, 128, 128)),
    "MinkUNet14D": ((1,) * 8, (32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet18A": ((2,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet18B": ((2,) * 8, (32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet18D": ((2,) * 8, (32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet34A": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 64, 64)),
    "MinkUNet34B": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 64, 32)),
    "MinkUNet34C": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 96, 96)),
}
INIT_DIM = 32
# (down conv, bn, stage) names per encoder level and (up conv, bn, stage) per decoder level
_ENC = (("conv1p1s2", "bn1", "block1"), ("conv2p2s2", "bn2", "block2"), ("conv3p4s2", "bn3", "block3"),
        ("conv4p8s2", "bn4", "block4"))
_DEC = (("convtr4p16s2", "bntr4", "block5"), ("convtr5p8s2", "bntr5", "block6"), ("convtr6p4s2", "bntr6", "block7"),
        ("convtr7p2s2", "bntr7", "block8"))


class MinkUNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=20, D=3, arch="MinkUNet18A"):
        super().__init__()
        if arch not in ARCHS:
            raise Exception("architecture not supported yet: {}".format(arch))
        layers, planes = ARCHS[arch]
        self.arch, self.D = arch, D
        width = INIT_DIM
        self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, width, kernel_size=5, dimension=D)
        self.bn0 = ME.MinkowskiBatchNorm(width)
        skips = [width]
        for lvl, (cname, bname, sname) in enumerate(_ENC):
            setattr(self, cname, ME.MinkowskiConvolution(width, width, kernel_size=2, stride=2, dimension=D))
            setattr(self, bname, ME.MinkowskiBatchNorm(width))
            stage, width = self._stage(width, planes[lvl], layers[lvl])
            setattr(self, sname, stage)
            skips.append(width)
        skips.pop()  # the bottleneck is not a skip
        for lvl, (cname, bname, sname) in enumerate(_DEC):
            p = planes[4 + lvl]
            setattr(self, cname, ME.MinkowskiConvolutionTranspose(width, p, kernel_size=2, stride=2, dimension=D))
            setattr(self, bname, ME.MinkowskiBatchNorm(p))
            stage, width = self._stage(p + skips.pop(), p, layers[4 + lvl])
            setattr(self, sname, stage)
        self.final = ME.MinkowskiConvolution(planes[7], out_channels, kernel_size=1, dimension=D)
        self.final.emit_split = False  # consumed by dense linear heads, not by another sparse conv: no pre-split copy
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.weight_initialization()

    def _stage(self, inplanes, planes, blocks):
        mods = []
        for i in range(blocks):
            down = None
            if i == 0 and inplanes != planes:
                down = nn.Sequential(ME.MinkowskiConvolution(inplanes, planes, kernel_size=1, stride=1, dimension=self.D),
                                     ME.MinkowskiBatchNorm(planes))
            mods.append(ME.BasicBlock(inplanes if i == 0 else planes, planes, stride=1, dilation=1, downsample=down,
                                      dimension=self.D))
        return nn.Sequential(*mods), planes

    def weight_initialization(self):
        # resnet_base.py:55-62 (note: isinstance(..., MinkowskiConvolution) deliberately excludes transposed convs there)
        for m in self.modules():
            if type(m) is ME.MinkowskiConvolution:
                ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def forward(self, x):
        act = self.relu
        out = act(self.bn0(self.conv0p1s1(x)))
        skips = [out]
        for cname, bname, sname in _ENC:
            out = act(getattr(self, bname)(getattr(self, cname)(out)))
            out = getattr(self, sname)(out)
            skips.append(out)
        bottleneck = skips.pop()
        for cname, bname, sname in _DEC:
            out = act(getattr(self, bname)(getattr(self, cname)(out)))
            out = getattr(self, sname)(ME.cat(out, skips.pop()))
        return bottleneck, self.final(out)


def mink_unet(in_channels=3, out_channels=20, D=3, arch="MinkUNet18A"):
    return MinkUNet(in_channels, out_channels, D, arch)
