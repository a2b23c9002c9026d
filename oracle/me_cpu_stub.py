"""A CPU stand-in for the MinkowskiEngine names the reference's MinkUNet files import, built on the ops of
oracle/spconv_oracle.py.

TEST INFRASTRUCTURE ONLY (used by tests/golden/make_golden.py to run the REFERENCE's own
``models/modeling/meta_arch/mink_unet.py::MinkUNetBase.forward`` (:118-178) and ``resnet_base.py`` on the
CPU in the build container, where MinkowskiEngine itself is absent).  What that run pins: the network
TOPOLOGY as the reference's code executes it (layer order, strides, skip concatenations, BasicBlock wiring,
which rows ``temp_out`` / ``out`` are) against the restated topology of ``spconv_oracle.minkunet_forward``
and of ``xmask3d_amd/mink_unet.py``.  What it cannot pin: the operator semantics themselves (offset order,
even-kernel offsets, transposed map) - those are the documented ME-0.5 rules restated in spconv_oracle and
stay PARITY UNPINNED (MinkowskiEngine is not installable here; the reference holds no fixture for them).

Never imported by ``xmask3d_amd``; never travels into the product path.
"""
from __future__ import annotations

import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import spconv_oracle as so


class SparseTensor:
    def __init__(self, features, coordinates=None, tensor_stride=1, cache=None):
        self.F = features
        self.cache = cache if cache is not None else so.CoordCache(np.asarray(coordinates, dtype=np.int32))
        self.tensor_stride = tensor_stride

    @property
    def C(self):
        return torch.from_numpy(self.cache.level(self.tensor_stride))

    def _like(self, feats, ts=None):
        return SparseTensor(feats, tensor_stride=self.tensor_stride if ts is None else ts, cache=self.cache)

    def __add__(self, other):
        assert other.tensor_stride == self.tensor_stride
        return self._like(self.F + other.F)

    __iadd__ = __add__


class _Conv(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None):
        super().__init__()
        assert dilation == 1 and not bias and dimension == 3
        self.kernel_size, self.stride = kernel_size, stride
        kv = kernel_size ** 3
        self.kernel = nn.Parameter(torch.zeros((in_channels, out_channels) if kv == 1 else (kv, in_channels, out_channels)))

    def forward(self, x):
        ts_in = x.tensor_stride
        ts_out = ts_in // self.stride if self.transposed else ts_in * self.stride
        nbr = x.cache.map(ts_in, ts_out, self.kernel_size, self.transposed)
        return x._like(so.spconv(x.F, self.kernel.detach(), nbr), ts_out)


class MinkowskiConvolution(_Conv):
    pass


class MinkowskiConvolutionTranspose(_Conv):
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def forward(self, x):
        return x._like(self.bn(x.F))


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x._like(torch.relu(x.F))


def cat(*ts):
    return ts[0]._like(torch.cat([t.F for t in ts], 1))


class BasicBlock(nn.Module):
    """MinkowskiEngine.modules.resnet_block.BasicBlock wiring (conv-bn-relu-conv-bn (+downsample) +res, relu)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        return self.relu(out)


class _Unused(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def install():
    """register this module as ``MinkowskiEngine`` (generator script only) -> the module object"""
    me = sys.modules[__name__]
    me.MinkowskiAvgPooling = me.MinkowskiGlobalMaxPooling = me.MinkowskiLinear = _Unused
    me.Bottleneck = BasicBlock
    utils = types.ModuleType("MinkowskiEngine.utils")
    utils.kaiming_normal_ = lambda t, **k: t
    me.utils = utils
    modules = types.ModuleType("MinkowskiEngine.modules")
    rb = types.ModuleType("MinkowskiEngine.modules.resnet_block")
    rb.BasicBlock, rb.Bottleneck = BasicBlock, BasicBlock
    modules.resnet_block = rb
    for name, mod in (("MinkowskiEngine", me), ("MinkowskiEngine.modules", modules), ("MinkowskiEngine.modules.resnet_block", rb),
                      ("MinkowskiEngine.utils", utils)):
        sys.modules[name] = mod
    return me


# ----------------------------------------------------------------------------- closed-form parameters / inputs
def closed_form_state(shapes: dict) -> dict:
    """Deterministic, seed-free parameter values for a {key: shape} table (37.9 M parameters are too big to commit):
    kernels ~ cos ramp / sqrt(fan_in), BN weight 1 +- 0.1, bias / running_mean +- 0.1, running_var in [1, 1.3]."""
    out = {}
    for j, key in enumerate(sorted(shapes)):
        shape = tuple(shapes[key])
        n = int(np.prod(shape)) if shape else 1
        ramp = np.cos(np.linspace(0.0, 1000.0 + 7.0 * j, n, dtype=np.float64) + 0.37 * j)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros(shape, dtype=torch.int64)
            continue
        if key.endswith(".kernel"):
            fan_in = shape[-2] * (shape[0] if len(shape) == 3 else 1)
            v = ramp * (1.7 / np.sqrt(fan_in))
        elif key.endswith("running_var"):
            v = 1.0 + 0.3 * ramp * ramp
        elif key.endswith(".bn.weight"):
            v = 1.0 + 0.1 * ramp
        else:
            v = 0.1 * ramp
        out[key] = torch.from_numpy(v.reshape(shape).astype(np.float32))
    return out


def seam_cloud(n=2600, seed=11, extent=72):
    """unique int32 voxel coordinates [0, x, y, z] on two noisy surfaces inside an extent^3 grid + colour-like features"""
    r = np.random.RandomState(seed)
    xy = r.randint(0, extent, size=(n, 2))
    z = np.where(r.rand(n) < 0.5, (xy[:, 0] // 3 + r.randint(0, 2, n)) % extent, (extent - 1 - xy[:, 1] // 2 + r.randint(0, 2, n)) % extent)
    c = np.unique(np.concatenate([np.zeros((n, 1), np.int64), xy, z[:, None]], 1), axis=0).astype(np.int32)
    c = c[r.permutation(len(c))]
    feats = r.uniform(-1, 1, size=(len(c), 3)).astype(np.float32)
    return c, feats
