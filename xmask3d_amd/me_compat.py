"""MinkowskiEngine-compatible operator surface backed by libxm3d_hip.so.

This is the drop-in seam of SURVEY.md §8b "Sparse operator seam": the names,
constructor arguments, parameter names/shapes (``kernel`` (K,Cin,Cout) or
(Cin,Cout) for k=1, ``bn.*``) and call conventions the reference uses from
``import MinkowskiEngine as ME``
(/root/reference/models/modeling/meta_arch/mink_unet.py:25-26,47-116,
resnet_base.py:3-4,55-96, pc_processor.py:31-33,57, run/train.py:18,35,186,483).
``install_as_minkowski_engine()`` registers this module under that name so the
reference's model files import unchanged.

Design (MI355X-first, not ME's): coordinates are hashed once per forward by a
shared CoordinateManager; every conv is ONE output-stationary HIP kernel over a
neighbour table; and the module-by-module call pattern the reference writes
(conv -> bn -> relu, ``out += residual``) is folded into that kernel's epilogue
by lazy evaluation: a conv returns a SparseTensor whose features are *pending*;
eval-mode BatchNorm, ReLU and the residual add attach to the pending epilogue,
and the kernel launches when somebody needs the features (``.F``, the next
conv, ``cat``).  Training-mode BatchNorm materialises the conv, reduces
statistics with a HIP kernel and leaves a pending fused affine(+residual)(+ReLU).
"""
from __future__ import annotations

import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# ----------------------------------------------------------------------------- autograd (training) path
class _SpconvFn(torch.autograd.Function):
    """out = sum_k feats[nbr[k]] @ kernel[k];  dgrad = the same kernel over the inverse map with kernel[k]^T,
    wgrad = xm3d_spconv_bwd_weight (replaces ME's conv backward, reached from loss.backward() run/train.py:537)."""

    @staticmethod
    def forward(ctx, feats, kernel3, cm, key, n_out):
        ts_in, ts_out, ks, transposed = key
        identity = ks == 1 and ts_in == ts_out
        nbr = None if identity else cm.kernel_map(*key)
        K, cin, cout = kernel3.shape
        tiles = cm.tiles(*key) if ops.mfma_eligible(cin, cout) else None
        ctx.save_for_backward(feats, kernel3)
        ctx.cm, ctx.key, ctx.identity = cm, key, identity
        return ops.spconv_fwd(feats, kernel3, nbr, n_out, order=cm.order(ts_out), tiles=tiles)

    @staticmethod
    def backward(ctx, gout):
        feats, kernel3 = ctx.saved_tensors
        cm, key = ctx.cm, ctx.key
        ts_in = key[0]
        gout = gout.contiguous()
        K, cin, cout = kernel3.shape
        gin = gw = None
        if ctx.needs_input_grad[0]:
            wt = kernel3.transpose(1, 2).contiguous()
            if ctx.identity:
                nbr_t, tiles_t = None, (cm.tiles(*key) if ops.mfma_eligible(cout, cin) else None)
            else:
                nbr_t = cm.inverse_map(*key)
                tiles_t = cm.tiles(*key, inverse=True) if ops.mfma_eligible(cout, cin) else None
            gin = ops.spconv_fwd(gout, wt, nbr_t, feats.shape[0], order=cm.order(ts_in), tiles=tiles_t)
        if ctx.needs_input_grad[1]:
            gw = ops.spconv_bwd_weight(feats, gout, None if ctx.identity else cm.kernel_map(*key), K)
        return gin, gw, None, None, None


def sync_moments(s, ss, n, group=None):
    """(sum, sumsq, count) -> (mean, biased var, total count); all-reduced over `group` when given.
    Pure tensor math so the reduction logic is testable on CPU with gloo."""
    import torch.distributed as dist

    packed = torch.cat([s.double(), ss.double(), torch.full((1,), float(n), dtype=torch.float64, device=s.device)])  # (a fill, not a host->device upload)
    if group is not None:
        dist.all_reduce(packed, group=None if group is True else group)
    c = s.numel()
    total = packed[-1]
    mean = packed[:c] / total
    var = (packed[c:2 * c] / total - mean * mean).clamp_min(0.0)
    return mean, var, total


class _BatchNormFn(torch.autograd.Function):
    """training-mode BatchNorm over rows with optional cross-rank statistics (MinkowskiBatchNorm / SyncBatchNorm):
    statistics pass, one per-channel kernel (mean / invstd / affine terms / running buffers), affine pass; backward =
    one reduction + one elementwise kernel.  The counts stay on the device: no host synchronisation per layer."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group, running_mean, running_var, num_batches, momentum):
        import torch.distributed as dist

        c = x.shape[1]
        packed = ops.bn_stats_packed(x)
        total = float(x.shape[0])
        if group is not None:
            packed[2 * c:].fill_(total)
            dist.all_reduce(packed, group=None if group is True else group)
            total = -1.0  # read the all-reduced count from the buffer
        mean, invstd, scale, shift, tot = ops.bn_finalize(packed, c, total, weight, bias, eps, momentum, running_mean, running_var,
                                                          num_batches)
        y = ops.affine_act(x, scale, shift)
        ctx.save_for_backward(x, weight, mean, invstd, tot)
        ctx.group, ctx.has_bias = group, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist

        x, weight, mean, invstd, tot = ctx.saved_tensors
        gy = gy.contiguous()
        sums = ops.bn_bwd_reduce(gy, x, mean, invstd)
        local = None
        if ctx.group is not None:
            # the input gradient needs the sums over ALL ranks' rows (the statistics couple them); the affine gradients must stay
            # this rank's own share - DDP averages parameter gradients over the ranks, and torch.nn.SyncBatchNorm does the same.
            # (Taking them from the all-reduced sums made every rank hold the global value: world x the intended gradient -
            # found by tests/test_gpu_distributed.py against a single process on the union of the data.)
            local = sums.clone()
            dist.all_reduce(sums, group=None if ctx.group is True else ctx.group)
        gx, gw, gb = ops.bn_bwd_apply(gy, x, mean, invstd, weight, sums, tot, need_wb=weight is not None or ctx.has_bias)
        if local is not None and gw is not None:
            c = x.shape[1]
            gb, gw = local[:c].float(), local[c:].float()
        return gx, (gw if weight is not None else None), (gb if ctx.has_bias else None), None, None, None, None, None, None


# ----------------------------------------------------------------------------- tensors
class _PendingConv:
    def __init__(self, feats, kernel3, packed, nbr, order, n_out, tiles=None, feats_split=None, emit_split=False):
        self.feats, self.kernel3, self.packed = feats, kernel3, packed
        self.nbr, self.order, self.n_out, self.tiles = nbr, order, n_out, tiles
        self.feats_split, self.emit_split = feats_split, emit_split
        self.scale = self.shift = self.residual = None
        self.relu = False

    def can_fold_affine(self):
        return self.residual is None and not self.relu

    def run(self):
        """-> (features, pre-split bf16 copy or None)"""
        if self.feats.dtype == torch.bfloat16:  # the bf16 configuration: one bf16 plane in and out
            res = self.residual
            if res is not None and res.dtype != torch.bfloat16:
                res = res.to(torch.bfloat16)
            return ops.spconv_fwd_bf16(self.feats, tuple(self.kernel3.shape), self.packed, self.tiles, self.n_out, order=self.order,
                                       scale=self.scale, shift=self.shift, residual=res, relu=self.relu), None
        r = ops.spconv_fwd(self.feats, self.kernel3, self.nbr, self.n_out, order=self.order, scale=self.scale, shift=self.shift,
                           residual=self.residual, relu=self.relu, packed=self.packed, tiles=self.tiles,
                           feats_split=self.feats_split, want_split=self.emit_split)
        return r if self.emit_split else (r, None)


class _PendingAffine:
    def __init__(self, x, scale, shift):
        self.x, self.scale, self.shift = x, scale, shift
        self.residual = None
        self.relu = False

    def can_fold_affine(self):
        return False

    def run(self):
        if self.x.dtype == torch.bfloat16:  # bf16 configuration, rare (an affine / add / ReLU with no convolution to fold into): torch ops
            y = self.x.float()
            if self.scale is not None:
                y = y * self.scale
            if self.shift is not None:
                y = y + self.shift
            if self.residual is not None:
                y = y + self.residual.float()
            return (torch.relu_(y) if self.relu else y).to(torch.bfloat16), None
        res = self.residual
        return ops.affine_act(self.x, self.scale, self.shift, res.float() if res is not None and res.dtype != torch.float32 else res, self.relu), None


class SparseTensor:
    """Features (N,C) f32 + int32 coordinates (N,4) [batch,x,y,z] at a tensor stride."""

    def __init__(self, features=None, coordinates=None, tensor_stride=1, coordinate_manager=None, _pending=None):
        if coordinate_manager is None:
            if coordinates is None:
                raise ValueError("SparseTensor needs coordinates or a coordinate_manager")
            if coordinates.dtype != torch.int32:
                coordinates = coordinates.int()
            coordinate_manager = ops.CoordinateManager(coordinates.contiguous())
            coordinate_manager.order(1)  # also verifies uniqueness, like ME's default quantisation mode
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = tensor_stride
        self._F = None
        self._Fs = None  # (2, N, C) bf16 hi / lo copy of _F, written by the split-operand conv that produced it (eval path)
        self._pending = _pending
        if features is not None:
            if features.dtype not in (torch.float32, torch.bfloat16):  # (bf16: the plain-bf16 sparse path of the bf16 configuration)
                features = features.float()
            self._F = features.contiguous()
            if self._F.shape[0] != coordinate_manager.num(tensor_stride):
                raise RuntimeError("features and coordinates have different numbers of rows")

    # ME attribute names
    @property
    def F(self):
        if self._F is None:
            self._F, self._Fs = self._pending.run()
            self._pending = None
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.coords(self.tensor_stride)

    @property
    def features(self):
        return self.F

    @property
    def coordinates(self):
        return self.C

    @property
    def shape(self):
        if self._F is not None:
            return self._F.shape
        p = self._pending
        return torch.Size((p.n_out, p.kernel3.shape[2])) if isinstance(p, _PendingConv) else p.x.shape

    def _like(self, features=None, pending=None):
        return SparseTensor(features, tensor_stride=self.tensor_stride, coordinate_manager=self.coordinate_manager,
                            _pending=pending)

    def __add__(self, other):
        return _add(self, other)

    def __iadd__(self, other):
        return _add(self, other)


def _add(a: SparseTensor, b: SparseTensor) -> SparseTensor:
    if a.coordinate_manager is not b.coordinate_manager or a.tensor_stride != b.tensor_stride:
        raise RuntimeError("adding SparseTensors that live on different coordinate maps")
    if torch.is_grad_enabled():
        return a._like(a.F + b.F)
    if a._pending is not None and a._pending.residual is None and not a._pending.relu:
        a._pending.residual = b.F
        return a
    if b._pending is not None and b._pending.residual is None and not b._pending.relu:
        b._pending.residual = a.F
        return b
    p = _PendingAffine(a.F, None, None)
    p.residual = b.F
    return a._like(pending=p)


def cat(*tensors):
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    t0 = tensors[0]
    for t in tensors[1:]:
        if t.coordinate_manager is not t0.coordinate_manager or t.tensor_stride != t0.tensor_stride:
            raise RuntimeError("ME.cat needs tensors on the same coordinate map")
    fs = [t.F for t in tensors]
    if len({f.dtype for f in fs}) > 1:
        fs = [f.float() for f in fs]
    out = t0._like(torch.cat(fs, dim=1))
    if all(t._Fs is not None for t in tensors):  # keep the pre-split copies: hi and lo planes concatenated channel-wise
        out._Fs = torch.cat([t._Fs for t in tensors], dim=2)
    return out


# ----------------------------------------------------------------------------- modules
class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        if dimension not in (None, 3):
            raise NotImplementedError("only dimension=3 is on the XMask3D path")
        if dilation != 1 or kernel_generator is not None or expand_coordinates:
            raise NotImplementedError("dilation / custom kernel generators / expand_coordinates are not on the XMask3D path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dimension = int(kernel_size), int(stride), 3
        kv = self.kernel_size ** 3
        self.kernel_volume = kv
        shape = (in_channels, out_channels) if kv == 1 else (kv, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()
        self._packed = None
        self._packed_key = None

    def reset_parameters(self):
        with torch.no_grad():
            n = (self.out_channels if self.transposed else self.in_channels) * self.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def _kernel3(self):
        return self.kernel if self.kernel.dim() == 3 else self.kernel.unsqueeze(0)

    def _packed_weight(self, k3, cin=None, force_split=False):
        cin = self.in_channels if cin is None else cin
        if not ops.mfma_eligible(cin, self.out_channels):
            return None
        algo = ops.ALGO_SPLIT if force_split else ops.default_tiled_algo(cin, self.out_channels, self.kernel_volume)
        key = (k3.data_ptr(), self.kernel._version, k3.device, algo)
        if self._packed_key != key:
            pack = ops.pack_weight_split if algo == ops.ALGO_SPLIT else ops.pack_weight
            self._packed = pack(k3.detach().contiguous())
            self._packed_key = key
        return self._packed

    def _strides(self, ts_in):
        if self.transposed:
            if ts_in % self.stride:
                raise RuntimeError("transposed conv on a tensor stride that is not a multiple of its stride")
            return ts_in // self.stride
        return ts_in * self.stride

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm = x.coordinate_manager
        ts_in = x.tensor_stride
        ts_out = self._strides(ts_in)
        feats = x.F
        if torch.is_grad_enabled() and (feats.requires_grad or self.kernel.requires_grad):
            k3 = self._kernel3()
            out = _SpconvFn.apply(feats, k3 if k3.is_contiguous() else k3.contiguous(), cm,
                                  (ts_in, ts_out, self.kernel_size, self.transposed), cm.num(ts_out))
            if self.bias is not None:
                out = out + self.bias
            return SparseTensor(out, tensor_stride=ts_out, coordinate_manager=cm)
        k3 = self._kernel3().detach()
        if not k3.is_contiguous():
            k3 = k3.contiguous()
        n_out = cm.num(ts_out)
        cin = self.in_channels
        cin_pad = 32 if (cin < 32 and self.out_channels % 32 == 0 and self.kernel_volume > 1) else cin
        bf16 = (getattr(self, "bf16_io", False) or feats.dtype == torch.bfloat16) and ops.mfma_eligible(cin_pad, self.out_channels) \
            and self.kernel_volume <= 128
        if bf16 and feats.dtype != torch.bfloat16:
            feats = feats.to(torch.bfloat16)   # the net's entry: from here on activations are one bf16 plane
        elif not bf16 and feats.dtype == torch.bfloat16:
            feats = feats.float()              # a layer the MFMA kernel does not take: back to f32 for the rest of the net
        if cin < 32 and self.out_channels % 32 == 0 and self.kernel_volume > 1:
            # the 3 -> 32 stem (mink_unet.py conv0p1s1, 5^3 offsets): zero-pad the input channels to one 32-channel MFMA step
            # instead of the scalar kernel - same sums (the padded products are exact zeros)
            cin = 32
            key = (k3.data_ptr(), self.kernel._version, k3.device)
            if getattr(self, "_padded_key", None) != key:
                self._padded, self._padded_key = F.pad(k3, (0, 0, 0, 32 - self.in_channels)).contiguous(), key
            k3 = self._padded
            feats = F.pad(feats, (0, 32 - self.in_channels))
        packed = self._packed_weight(k3, cin, force_split=bf16)
        nbr = tiles = None
        if packed is not None:
            tiles = cm.tiles(ts_in, ts_out, self.kernel_size, self.transposed)
        if not (self.kernel_volume == 1 and self.stride == 1):
            nbr = cm.kernel_map(ts_in, ts_out, self.kernel_size, self.transposed)
        split = not bf16 and packed is not None and ops.default_tiled_algo(cin, self.out_channels, self.kernel_volume) == ops.ALGO_SPLIT
        pend = _PendingConv(feats, k3, packed, nbr, cm.order(ts_out), n_out, tiles, feats_split=x._Fs if split else None,
                            emit_split=split and getattr(self, "emit_split", True))
        if self.bias is not None:
            pend.shift = self.bias.detach().reshape(-1).contiguous()
        return SparseTensor(tensor_stride=ts_out, coordinate_manager=cm, _pending=pend)


class MinkowskiConvolution(_ConvBase):
    transposed = False


class MinkowskiConvolutionTranspose(_ConvBase):
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def _scale_shift(self, mean, var):
        bn = self.bn
        inv = torch.rsqrt(var + bn.eps)
        scale = inv * bn.weight.detach() if bn.affine else inv
        shift = -mean * scale
        if bn.affine:
            shift = shift + bn.bias.detach()
        return scale.float().contiguous(), shift.float().contiguous()

    def forward(self, x: SparseTensor) -> SparseTensor:
        bn = self.bn
        use_batch = self.training or not bn.track_running_stats
        if not use_batch and torch.is_grad_enabled() and (x.F.requires_grad or (bn.affine and (bn.weight.requires_grad or bn.bias.requires_grad))):
            # eval-mode statistics with gradients wanted (frozen-BN fine-tuning): the folded / detached affine below would cut
            # the graph, so run the differentiable torch form
            return x._like(F.batch_norm(x.F, bn.running_mean, bn.running_var, bn.weight if bn.affine else None,
                                        bn.bias if bn.affine else None, False, 0.0, bn.eps))
        if not use_batch:
            key = (bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(),
                   bn.weight._version if bn.affine else 0, bn.bias._version if bn.affine else 0)
            if getattr(self, "_fold_key", None) != key:
                self._fold = self._scale_shift(bn.running_mean, bn.running_var)
                self._fold_key = key
            scale, shift = self._fold
            p = x._pending
            if p is not None and p.can_fold_affine():
                if p.scale is None and p.shift is None:
                    p.scale, p.shift = scale, shift
                else:  # conv bias already sits in shift
                    s0 = p.scale if p.scale is not None else torch.ones_like(scale)
                    b0 = p.shift if p.shift is not None else torch.zeros_like(shift)
                    p.scale, p.shift = s0 * scale, b0 * scale + shift
                return x
            return x._like(pending=_PendingAffine(x.F, scale, shift))
        return self._batch_stats_forward(x, None)

    def _batch_stats_forward(self, x, group):
        bn = self.bn
        feats = x.F
        if torch.is_grad_enabled() and (bn.momentum is not None or not bn.track_running_stats):
            track = self.training and bn.track_running_stats
            y = _BatchNormFn.apply(feats, bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn.eps, group,
                                   bn.running_mean if track else None, bn.running_var if track else None,
                                   bn.num_batches_tracked if track else None, bn.momentum if track else None)
            return x._like(y)
        elif torch.is_grad_enabled():  # cumulative-average running statistics (momentum=None): torch ops
            feats_ = feats
            s, ss = ops.bn_stats(feats_.detach())
            mean, var, total = sync_moments(s, ss, feats.shape[0], group)
            y = F.batch_norm(feats, None, None, bn.weight if bn.affine else None, bn.bias if bn.affine else None, True, 0.0, bn.eps) \
                if group is None else None
            if y is None:
                raise NotImplementedError("SyncBatchNorm with momentum=None is not supported")
            out = x._like(y)
        else:
            s, ss = ops.bn_stats(feats)
            mean, var, total = sync_moments(s, ss, feats.shape[0], group)
            scale, shift = self._scale_shift(mean, var)
            out = x._like(pending=_PendingAffine(feats, scale, shift))
        if self.training and bn.track_running_stats:
            with torch.no_grad():
                m_ = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                unbiased = var * (total / (total - 1.0).clamp_min(1.0))
                bn.running_mean.mul_(1 - m_).add_(mean.float(), alpha=m_)
                bn.running_var.mul_(1 - m_).add_(unbiased.float(), alpha=m_)
                bn.num_batches_tracked += 1
        return out


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    """Statistics all-reduced over the default process group (RCCL on ROCm)."""

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        """Recursively swap MinkowskiBatchNorm for the synchronised variant (parameters are shared, not copied)."""
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, cls):
            bn = module.bn
            new = cls(bn.num_features, bn.eps, bn.momentum, bn.affine, bn.track_running_stats)
            new.bn = bn
            new.train(module.training)
            return new
        for name, child in list(module.named_children()):
            setattr(module, name, cls.convert_sync_batchnorm(child, process_group))
        return module

    def forward(self, x):
        import torch.distributed as dist

        if not (self.training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return super().forward(x)
        return self._batch_stats_forward(x, True)  # one fused (2C+1)-double all-reduce per layer (+ one in backward)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x: SparseTensor) -> SparseTensor:
        if torch.is_grad_enabled():
            return x._like(torch.relu(x.F))
        if x._pending is not None:
            x._pending.relu = True  # relu is idempotent, folding twice is harmless
            return x
        p = _PendingAffine(x.F, None, None)
        p.relu = True
        return x._like(pending=p)


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        f = x.F
        return x._like(self.linear(f.float() if f.dtype != self.linear.weight.dtype else f))


class _NotOnPath(nn.Module):
    """Imported by resnet_base.py but never executed by MinkUNet (constructed only by ResNetBase)."""

    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        raise NotImplementedError(f"{type(self).__name__} is not on the XMask3D hot path")


class MinkowskiAvgPooling(_NotOnPath):
    pass


class MinkowskiGlobalMaxPooling(_NotOnPath):
    pass


# ----------------------------------------------------------------------------- resnet blocks
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                          dimension=dimension)
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


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.relu(self.norm2(self.conv2(out)))
        out = self.norm3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        return self.relu(out)


# ----------------------------------------------------------------------------- utils
def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_: fans computed for (K, Cin, Cout) / (Cin, Cout) kernels."""
    if tensor.dim() == 2:
        fan_in, fan_out = tensor.size(0), tensor.size(1)
    else:
        kv = tensor.size(0)
        fan_in, fan_out = tensor.size(1) * kv, tensor.size(2) * kv
    fan = fan_in if mode == "fan_in" else fan_out
    std = nn.init.calculate_gain(nonlinearity, a) / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)


utils = types.ModuleType("MinkowskiEngine.utils")
utils.kaiming_normal_ = kaiming_normal_


def install_as_minkowski_engine():
    """Make ``import MinkowskiEngine as ME`` resolve to this module (drop-in for the reference's model files)."""
    me = sys.modules[__name__]
    modules = types.ModuleType("MinkowskiEngine.modules")
    resnet_block = types.ModuleType("MinkowskiEngine.modules.resnet_block")
    resnet_block.BasicBlock, resnet_block.Bottleneck = BasicBlock, Bottleneck
    modules.resnet_block = resnet_block
    me.modules = modules
    sys.modules["MinkowskiEngine"] = me
    sys.modules["MinkowskiEngine.modules"] = modules
    sys.modules["MinkowskiEngine.modules.resnet_block"] = resnet_block
    sys.modules["MinkowskiEngine.utils"] = utils
    return me
