"""Tensor-level wrappers over the C ABI (include/xm3d.h).

torch is used for device memory and streams only; every computation below is
a HIP kernel of libxm3d_hip.so launched on torch's current stream.  No CPU
fallbacks: tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from ._lib import c_i64, c_sz, check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr())


def _req(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a ROCm device tensor (got {t.device}); xmask3d_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    return t


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------- voxelisation
def voxelize(xyz: torch.Tensor, matrix) -> tuple:
    """xyz (n,3) f64 device, matrix 4x4 (host) -> (grid (Nv,3) i32, inds (Nv,) i64, inverse (n,) i64)."""
    _req(xyz, torch.float64, "xyz", 2)
    n = xyz.shape[0]
    T = np.ascontiguousarray(np.asarray(matrix, dtype=np.float64).reshape(4, 4))
    need = c_sz(0)
    check(lib().xm3d_voxelize_ws_bytes(n, ctypes.byref(need)), "xm3d_voxelize_ws_bytes")
    ws = _workspace(need.value, xyz.device)
    grid = torch.empty((n, 3), dtype=torch.int32, device=xyz.device)
    inds = torch.empty(n, dtype=torch.int64, device=xyz.device)
    inv = torch.empty(n, dtype=torch.int64, device=xyz.device)
    nu = c_i64(0)
    check(lib().xm3d_voxelize(_ptr(xyz), n, T.ctypes.data_as(ctypes.c_void_p), _ptr(grid), _ptr(inds), _ptr(inv),
                              ctypes.byref(nu), _ptr(ws), ws.numel(), _stream()), "xm3d_voxelize")
    return grid[: nu.value], inds[: nu.value], inv


def fnv_keys(grid: torch.Tensor) -> torch.Tensor:
    _req(grid, torch.int32, "grid", 2)
    keys = torch.empty(grid.shape[0], dtype=torch.int64, device=grid.device)  # raw u64 bits
    check(lib().xm3d_fnv_keys(_ptr(grid), grid.shape[0], _ptr(keys), _stream()), "xm3d_fnv_keys")
    return keys


# ---------------------------------------------------------------- coordinate manager
def _pow2_cap(n):
    cap = 16
    while cap < 2 * n:
        cap *= 2
    return cap


class CoordinateManager:
    """Per-forward cache of coordinate sets, hashes and neighbour tables, keyed like
    MinkowskiEngine's coordinate manager so every conv with the same
    (tensor_stride_in, tensor_stride_out, kernel_size, transposed) shares one table,
    and both MinkUNets running on the same `sinput` share everything."""

    def __init__(self, coords: torch.Tensor):
        _req(coords, torch.int32, "coordinates", 2)
        if coords.shape[1] != 4:
            raise RuntimeError("coordinates must be (N, 4) int32 rows [batch, x, y, z]")
        self.device = coords.device
        self._coords = {1: coords}
        self._hash = {}
        self._maps = {}
        self._order = {}
        self._inv = {}
        self._tiles = {}

    def _ws(self, n):
        need = c_sz(0)
        check(lib().xm3d_stride_ws_bytes(n, ctypes.byref(need)), "xm3d_stride_ws_bytes")
        return _workspace(need.value, self.device)

    def coords(self, ts: int) -> torch.Tensor:
        if ts not in self._coords:
            src = self.coords(ts // 2) if ts > 2 else self._coords[1]
            n = src.shape[0]
            out = torch.empty((n, 4), dtype=torch.int32, device=self.device)
            ws = self._ws(n)
            cnt = c_i64(0)
            check(lib().xm3d_coords_stride(_ptr(src), n, ts, _ptr(out), ctypes.byref(cnt), _ptr(ws), ws.numel(), _stream()),
                  "xm3d_coords_stride")
            self._coords[ts] = out[: cnt.value]
        return self._coords[ts]

    def num(self, ts):
        return self.coords(ts).shape[0]

    def order(self, ts: int):
        """Spatially sorted processing order (None when rows are already sorted, i.e. ts > 1)."""
        if ts != 1:
            return None
        if ts not in self._order:
            c = self._coords[1]
            n = c.shape[0]
            order = torch.empty(n, dtype=torch.int32, device=self.device)
            ws = self._ws(n)
            nu = c_i64(0)
            check(lib().xm3d_coords_order(_ptr(c), n, _ptr(order), ctypes.byref(nu), _ptr(ws), ws.numel(), _stream()),
                  "xm3d_coords_order")
            if nu.value != n:
                raise RuntimeError(f"SparseTensor coordinates must be unique ({n - nu.value} duplicates); "
                                   "the reference always passes voxelised (deduplicated) coordinates")
            self._order[ts] = order
        return self._order[ts]

    def hash(self, ts: int):
        if ts not in self._hash:
            c = self.coords(ts)
            cap = _pow2_cap(c.shape[0])
            tk = torch.empty(cap, dtype=torch.int64, device=self.device)
            tv = torch.empty(cap, dtype=torch.int32, device=self.device)
            check(lib().xm3d_hash_build(_ptr(c), c.shape[0], _ptr(tk), _ptr(tv), cap, _stream()), "xm3d_hash_build")
            self._hash[ts] = (tk, tv, cap)
        return self._hash[ts]

    def kernel_map(self, ts_in: int, ts_out: int, ksize: int, transposed: bool = False) -> torch.Tensor:
        """(K, n_out) int32 neighbour table; -1 marks holes."""
        key = (ts_in, ts_out, ksize, transposed)
        if key not in self._maps:
            out_c = self.coords(ts_out)
            tk, tv, cap = self.hash(ts_in)
            n_out = out_c.shape[0]
            K = ksize ** 3
            nbr = torch.empty((K, n_out), dtype=torch.int32, device=self.device)
            if transposed:
                step, sign = ts_out, -1
            else:
                step, sign = ts_in, 1
            check(lib().xm3d_kernel_map(_ptr(out_c), n_out, _ptr(tk), _ptr(tv), cap, ksize, step, sign, _ptr(nbr), _stream()),
                  "xm3d_kernel_map")
            self._maps[key] = nbr
        return self._maps[key]

    def inverse_map(self, ts_in, ts_out, ksize, transposed=False):
        """(K, n_in) table with nbr_t[k, i] = o  <=>  nbr[k, o] = i (for dgrad)."""
        key = (ts_in, ts_out, ksize, transposed)
        if key not in self._inv:
            nbr = self.kernel_map(*key)
            n_in = self.num(ts_in)
            K, n_out = nbr.shape
            nbr_t = torch.empty((K, n_in), dtype=torch.int32, device=self.device)
            check(lib().xm3d_kernel_map_invert(_ptr(nbr), K, n_out, n_in, _ptr(nbr_t), _stream()), "xm3d_kernel_map_invert")
            self._inv[key] = nbr_t
        return self._inv[key]

    def tiles(self, ts_in, ts_out, ksize, transposed=False, inverse=False):
        """Tiled rulebook (tsrc i32, tdst u8, tcnt i32) of a kernel map (or of its inverse, for dgrad);
        built once and shared by every conv on that map.  ksize=1 gives the identity map."""
        key = (ts_in, ts_out, ksize, transposed, inverse)
        if key not in self._tiles:
            if inverse:
                nbr, n_out, order = self.inverse_map(ts_in, ts_out, ksize, transposed), self.num(ts_in), self.order(ts_in)
            else:
                nbr = None if (ksize == 1 and ts_in == ts_out) else self.kernel_map(ts_in, ts_out, ksize, transposed)
                n_out, order = self.num(ts_out), self.order(ts_out)
            K = 1 if nbr is None else nbr.shape[0]
            ntiles = (n_out + 255) // 256
            tsrc = torch.empty(max(ntiles * K * 256, 1), dtype=torch.int32, device=self.device)
            tdst = torch.empty(max(ntiles * K * 256, 1), dtype=torch.uint8, device=self.device)
            tcnt = torch.empty(max(ntiles * K, 1), dtype=torch.int32, device=self.device)
            check(lib().xm3d_rulebook_tiles(_ptr(nbr), _ptr(order), n_out, K, _ptr(tsrc), _ptr(tdst), _ptr(tcnt), _stream()),
                  "xm3d_rulebook_tiles")
            self._tiles[key] = (tsrc, tdst, tcnt)
        return self._tiles[key]

    def check(self):
        check(lib().xm3d_check_flag(), "coordinate manager")


# ---------------------------------------------------------------- sparse conv
ALGO_AUTO, ALGO_SCALAR, ALGO_MFMA, ALGO_TILES, ALGO_SPLIT = 0, 1, 2, 3, 4


import os as _os

_SPLIT_WG_TARGET = int(_os.environ.get("XM3D_SPLIT_WG_TARGET", "512"))


def default_tiled_algo(cin=None, cout=None, K=None):
    """algo for MFMA-eligible layers that have a tiled rulebook: the split-operand bf16 kernel (algo 4) unless
    XM3D_SPCONV_ALGO=tiles selects the exact-f32 MFMA kernel (algo 3).  The 32 -> 32 k=3 layers stay on algo 3: with two
    16-channel tiles per row tile the producers' gather + split work is not amortised (61 vs 50 us at tensor stride 2)."""
    import os

    if os.environ.get("XM3D_SPCONV_ALGO", "split") == "tiles":
        return ALGO_TILES
    if cin == 32 and cout == 32 and K is not None and K > 8:
        return ALGO_TILES
    return ALGO_SPLIT


def pack_weight(kernel: torch.Tensor) -> torch.Tensor:
    """(K,Cin,Cout) f32 -> MFMA fragment layout (same numel)."""
    _req(kernel, torch.float32, "kernel", 3)
    out = torch.empty_like(kernel)
    K, cin, cout = kernel.shape
    check(lib().xm3d_spconv_pack_weight(_ptr(kernel), K, cin, cout, _ptr(out), _stream()), "xm3d_spconv_pack_weight")
    return out


def pack_weight_split(kernel: torch.Tensor) -> torch.Tensor:
    """(K,Cin,Cout) f32 -> bf16 hi/lo fragments for xm3d_spconv_fwd_split (same byte size; opaque f32-typed buffer)."""
    _req(kernel, torch.float32, "kernel", 3)
    out = torch.empty_like(kernel)
    K, cin, cout = kernel.shape
    check(lib().xm3d_spconv_pack_weight_split(_ptr(kernel), K, cin, cout, _ptr(out), _stream()), "xm3d_spconv_pack_weight_split")
    return out


def mfma_eligible(cin, cout):
    return cin % 32 == 0 and cout % 32 == 0


def spconv_fwd(feats, kernel, nbr, n_out, order=None, scale=None, shift=None, residual=None, relu=False,
               algo=ALGO_AUTO, packed=None, tiles=None, ksplit=None, feats_split=None, want_split=False):
    """out (n_out, Cout) = epi(sum_k feats[nbr[k]] @ kernel[k]).  `packed` = pack_weight(kernel) cache;
    `tiles` = CoordinateManager.tiles(...) selects the tiled-rulebook MFMA kernels (algo 3 / 4).
    algo 4 only: feats_split = the (2, n_in, Cin) bf16 hi/lo copy of feats a previous call returned; want_split=True returns
    (out, out_split) with the (2, n_out, Cout) bf16 copy of out for the next conv (None when another algo ran)."""
    _req(feats, torch.float32, "features", 2)
    _req(kernel, torch.float32, "kernel", 3)
    K, cin, cout = kernel.shape
    if feats.shape[1] != cin:
        raise RuntimeError(f"feature width {feats.shape[1]} != kernel Cin {cin}")
    if nbr is not None:
        _req(nbr, torch.int32, "nbr", 2)
        if tuple(nbr.shape) != (K, n_out):
            raise RuntimeError(f"nbr shape {tuple(nbr.shape)} != ({K}, {n_out})")
    for t, nm, ln in ((scale, "scale", cout), (shift, "shift", cout)):
        if t is not None:
            _req(t, torch.float32, nm, 1)
            assert t.numel() == ln
    if residual is not None:
        _req(residual, torch.float32, "residual", 2)
        assert tuple(residual.shape) == (n_out, cout)
    if order is not None:
        _req(order, torch.int32, "order", 1)
        assert order.numel() == n_out
    if algo == ALGO_AUTO:
        algo = (default_tiled_algo(cin, cout, K) if tiles is not None else ALGO_MFMA) if mfma_eligible(cin, cout) else ALGO_SCALAR
    w = kernel
    if algo in (ALGO_MFMA, ALGO_TILES):
        w = packed if packed is not None else pack_weight(kernel)
    elif algo == ALGO_SPLIT:
        w = packed if packed is not None else pack_weight_split(kernel)
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if algo in (ALGO_TILES, ALGO_SPLIT):
        if tiles is None:
            raise RuntimeError("ALGO_TILES / ALGO_SPLIT need the tiled rulebook (CoordinateManager.tiles)")
        tsrc, tdst, tcnt = tiles
        split = algo == ALGO_SPLIT
        if ksplit is None:
            ctt = lib().xm3d_spconv_split_channels(cout) if split else lib().xm3d_spconv_tile_channels(cin, cout)
            wgs = ((n_out + 255) // 256) * (cout // ctt)
            if split:
                # one 768-thread workgroup per CU, and a workgroup's time is a chain of ~2 barrier intervals per kernel offset:
                # small grids are spread over the offsets until ~2 workgroups per CU exist (tools/prof_3d.py batch: 13.3 / 12.7 / 13.9 ms
                # for targets 128 / 512 / 768 on the 20-view bench batch)
                # (grids that already give every CU a workgroup are left alone: split-K costs the slab round trip)
                target = _SPLIT_WG_TARGET
                ksplit = 1 if wgs >= 256 else max(1, min(K, -(-target // max(wgs, 1))))
            else:
                ksplit = 1 if wgs >= 256 else max(1, min(K, 768 // max(wgs, 1)))
        slab = torch.empty((ksplit, n_out, cout), dtype=torch.float32, device=feats.device) if ksplit > 1 else None
        if split:
            if feats_split is not None:
                _req(feats_split, torch.bfloat16, "feats_split", 3)
                if tuple(feats_split.shape) != (2, feats.shape[0], cin):
                    raise RuntimeError(f"feats_split shape {tuple(feats_split.shape)} != (2, {feats.shape[0]}, {cin})")
            out_split = torch.empty((2, n_out, cout), dtype=torch.bfloat16, device=feats.device) if want_split else None
            check(lib().xm3d_spconv_fwd_split2(_ptr(feats), _ptr(feats_split), feats.shape[0], cin, _ptr(w), K, cout, _ptr(tsrc), _ptr(tdst),
                                               _ptr(tcnt), _ptr(order), n_out, _ptr(scale), _ptr(shift), _ptr(residual), int(bool(relu)),
                                               _ptr(out), _ptr(out_split), ksplit, _ptr(slab), _stream()), "xm3d_spconv_fwd_split2")
            return (out, out_split) if want_split else out
        check(lib().xm3d_spconv_fwd_tiles(_ptr(feats), feats.shape[0], cin, _ptr(w), K, cout, _ptr(tsrc), _ptr(tdst), _ptr(tcnt),
                                          _ptr(order), n_out, _ptr(scale), _ptr(shift), _ptr(residual), int(bool(relu)),
                                          _ptr(out), ksplit, _ptr(slab), _stream()), "xm3d_spconv_fwd_tiles")
        return (out, None) if want_split else out
    check(lib().xm3d_spconv_fwd(_ptr(feats), feats.shape[0], cin, _ptr(w), K, cout, _ptr(nbr), _ptr(order), n_out,
                                _ptr(scale), _ptr(shift), _ptr(residual), int(bool(relu)), _ptr(out), algo, _stream()),
          "xm3d_spconv_fwd")
    return (out, None) if want_split else out


def spconv_fwd_bf16(feats, kernel_shape, packed, tiles, n_out, order=None, scale=None, shift=None, residual=None, relu=False, ksplit=None):
    """The plain-bf16 form of the split kernel (xm3d_spconv_fwd_bf16): feats (n_in, Cin) bf16 -> (n_out, Cout) bf16; residual bf16 like the
    output; packed = pack_weight_split(kernel) (its hi plane is used); kernel_shape = (K, Cin, Cout)."""
    _req(feats, torch.bfloat16, "features", 2)
    K, cin, cout = kernel_shape
    if feats.shape[1] != cin or not feats.is_contiguous():
        raise RuntimeError(f"spconv_fwd_bf16: contiguous (n, {cin}) features required")
    if residual is not None:
        _req(residual, torch.bfloat16, "residual", 2)
        assert tuple(residual.shape) == (n_out, cout) and residual.is_contiguous()
    tsrc, tdst, tcnt = tiles
    if ksplit is None:
        wgs = ((n_out + 255) // 256) * (cout // lib().xm3d_spconv_split_channels(cout))
        ksplit = 1 if wgs >= 256 else max(1, min(K, -(-_SPLIT_WG_TARGET // max(wgs, 1))))
    slab = torch.empty((ksplit, n_out, cout), dtype=torch.float32, device=feats.device) if ksplit > 1 else None
    out = torch.empty((n_out, cout), dtype=torch.bfloat16, device=feats.device)
    check(lib().xm3d_spconv_fwd_bf16(_ptr(feats), feats.shape[0], cin, _ptr(packed), K, cout, _ptr(tsrc), _ptr(tdst), _ptr(tcnt), _ptr(order), n_out,
                                     _ptr(scale), _ptr(shift), _ptr(residual), int(bool(relu)), _ptr(out), ksplit, _ptr(slab), _stream()),
          "xm3d_spconv_fwd_bf16")
    return out


def spconv_bwd_weight(feats, gout, nbr, K):
    """gW (K,Cin,Cout) = sum_o feats[nbr[k,o]]^T gout[o]  (nbr None = K=1 identity map)"""
    _req(feats, torch.float32, "features", 2)
    _req(gout, torch.float32, "grad_output", 2)
    n_out, cout = gout.shape
    cin = feats.shape[1]
    if nbr is not None:
        _req(nbr, torch.int32, "nbr", 2)
        assert tuple(nbr.shape) == (K, n_out)
    gw = torch.empty((K, cin, cout), dtype=torch.float32, device=feats.device)
    check(lib().xm3d_spconv_bwd_weight(_ptr(feats), feats.shape[0], cin, _ptr(gout), n_out, cout, _ptr(nbr), K, _ptr(gw),
                                       _stream()), "xm3d_spconv_bwd_weight")
    return gw


def bn_stats(x):
    """per-channel (sum, sumsq) in f64 over rows of an (n,c) f32 matrix."""
    _req(x, torch.float32, "x", 2)
    s = torch.empty(2 * x.shape[1], dtype=torch.float64, device=x.device)
    check(lib().xm3d_bn_stats(_ptr(x), x.shape[0], x.shape[1], _ptr(s), _stream()), "xm3d_bn_stats")
    return s[: x.shape[1]], s[x.shape[1]:]


def bn_stats_packed(x):
    """-> f64 (2c+1): [sum(c), sumsq(c), unset count slot] - the buffer SyncBatchNorm all-reduces"""
    _req(x, torch.float32, "x", 2)
    s = torch.empty(2 * x.shape[1] + 1, dtype=torch.float64, device=x.device)
    check(lib().xm3d_bn_stats(_ptr(x), x.shape[0], x.shape[1], _ptr(s), _stream()), "xm3d_bn_stats")
    return s


def bn_finalize(packed, c, total, weight, bias, eps, momentum=None, running_mean=None, running_var=None, num_batches=None):
    """-> mean, invstd, scale, shift (c,) f32 and total (1,) f32; momentum given: running buffers updated in place."""
    dev = packed.device
    out = torch.empty(4 * c + 4, dtype=torch.float32, device=dev)  # one allocation; slices stay 16-byte aligned for c % 4 == 0
    mean, invstd, scale, shift, tot = out[:c], out[c:2 * c], out[2 * c:3 * c], out[3 * c:4 * c], out[4 * c:4 * c + 1]
    check(lib().xm3d_bn_finalize(_ptr(packed), c, float(total), _ptr(weight), _ptr(bias), float(eps),
                                 -1.0 if momentum is None else float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(num_batches),
                                 _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), _ptr(tot), _stream()), "xm3d_bn_finalize")
    return mean, invstd, scale, shift, tot


def bn_bwd_reduce(gy, x, mean, invstd):
    _req(gy, torch.float32, "gy", 2)
    _req(x, torch.float32, "x", 2)
    sums = torch.empty(2 * x.shape[1], dtype=torch.float64, device=x.device)
    check(lib().xm3d_bn_bwd_reduce(_ptr(gy), _ptr(x), x.shape[0], x.shape[1], _ptr(mean), _ptr(invstd), _ptr(sums), _stream()),
          "xm3d_bn_bwd_reduce")
    return sums


def bn_bwd_apply(gy, x, mean, invstd, weight, sums, total, need_wb=True):
    c = x.shape[1]
    gx = torch.empty_like(x)
    gwb = torch.empty(2 * c, dtype=torch.float32, device=x.device) if need_wb else None
    check(lib().xm3d_bn_bwd_apply(_ptr(gy), _ptr(x), x.shape[0], c, _ptr(mean), _ptr(invstd), _ptr(weight), _ptr(sums), _ptr(total),
                                  _ptr(gx), _ptr(gwb[:c]) if need_wb else None, _ptr(gwb[c:]) if need_wb else None, _stream()),
          "xm3d_bn_bwd_apply")
    return gx, (gwb[:c] if need_wb else None), (gwb[c:] if need_wb else None)


def affine_act(x, scale=None, shift=None, residual=None, relu=False, out=None):
    _req(x, torch.float32, "x", 2)
    if out is None:
        out = torch.empty_like(x)
    check(lib().xm3d_affine_act(_ptr(x), x.shape[0], x.shape[1], _ptr(scale), _ptr(shift), _ptr(residual),
                                int(bool(relu)), _ptr(out), _stream()), "xm3d_affine_act")
    return out


# ---------------------------------------------------------------- fused GroupNorm(+SiLU)
def is_nhwc(x):
    """4-D device tensor stored channels-last (and not also NCHW-contiguous)"""
    return x.dim() == 4 and x.is_cuda and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


def _act_torch(y, act):
    act = int(act)
    return y * torch.sigmoid(y) if act == 1 else (torch.relu_(y) if act == 2 else y)


def _gn_stats_buf(x, B, C, hw, G):
    """statistics buffer of the channels-last GroupNorm kernels: B*G*2 f64 moments followed by the per-workgroup partial pairs they are
    summed from in fixed order (xm3d.h, "STATISTICS BUFFERS")"""
    return torch.empty(lib().xm3d_gn_stats_doubles_nhwc(B, C, hw, G, 0 if x.dtype == torch.float32 else 1), dtype=torch.float64, device=x.device)


def group_norm(x, num_groups, weight=None, bias=None, eps=1e-5, silu=False, shift=None, residual=None):
    """NCHW, channels-last or (B,C,L) f32/bf16 device tensor -> same shape/dtype/layout; statistics in f64, math in f32.
    silu: False/0 none, True/1 SiLU, 2 ReLU fused after the affine.
    shift: optional (C,) or (B,C) term added to x before normalising (conv bias / embedding term folded in).
    residual: optional tensor like x added after the affine, before the activation (channels-last kernel; else a torch add)."""
    if not x.is_cuda:
        raise RuntimeError("group_norm: ROCm device tensor required (no CPU path)")
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"group_norm: unsupported dtype {x.dtype}")
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // max(B * C, 1)
    for t in (weight, bias):
        if t is not None and (t.dtype != x.dtype or not t.is_contiguous() or t.numel() != C):
            raise TypeError("group_norm: weight/bias must be contiguous (C,) tensors of the input dtype")
    nvec = 4 if x.dtype == torch.float32 else 8
    if is_nhwc(x) and C % nvec == 0 and num_groups <= 64:
        stats = _gn_stats_buf(x, B, C, hw, num_groups)
        y = torch.empty_like(x)  # preserves channels_last
        bstride = 0
        if shift is not None:
            shift = shift.to(x.dtype).contiguous()
            if shift.numel() not in (C, B * C):
                raise TypeError("group_norm: shift must have C or B*C elements")
            bstride = C if (shift.numel() == B * C and B > 1) else 0
        fused_res = residual is not None and residual.dtype == x.dtype and residual.shape == x.shape and is_nhwc(residual)
        pre = getattr(x, "_xm3d_gn_stats", None)
        if pre is not None and shift is None and pre[1] == num_groups and pre[2] == x.data_ptr():
            # statistics already taken by the kernel that produced x (xm3d_bias_residual_stats_nhwc): apply pass only
            check(lib().xm3d_group_norm_nhwc_apply(_ptr(x), None, 0, 0 if x.dtype == torch.float32 else 1, B, C, hw, num_groups, _ptr(weight),
                                                   _ptr(bias), float(eps), 0 if (residual is not None and not fused_res) else int(silu),
                                                   _ptr(residual if fused_res else None), _ptr(y), _ptr(pre[0]), _stream()),
                  "xm3d_group_norm_nhwc_apply")
            if residual is not None and not fused_res:
                y = _act_torch(y + residual, silu)
            return y
        check(lib().xm3d_group_norm_nhwc_res(_ptr(x), _ptr(shift), bstride, 0 if x.dtype == torch.float32 else 1, B, C, hw, num_groups,
                                             _ptr(weight), _ptr(bias), float(eps), 0 if (residual is not None and not fused_res) else int(silu),
                                             _ptr(residual if fused_res else None), _ptr(y), _ptr(stats), _stream()),
              "xm3d_group_norm_nhwc_res")
        if residual is not None and not fused_res:
            y = _act_torch(y + residual, silu)
        return y
    if shift is not None:
        x = x + shift.to(x.dtype).reshape(-1, C, *([1] * (x.dim() - 2)))
    if not x.is_contiguous():
        x = x.contiguous()
    y = torch.empty_like(x)
    stats = torch.empty(lib().xm3d_gn_stats_doubles_nchw(B, C, hw, num_groups), dtype=torch.float64, device=x.device)
    check(lib().xm3d_group_norm(_ptr(x), 0 if x.dtype == torch.float32 else 1, B, C, hw, num_groups, _ptr(weight), _ptr(bias),
                                float(eps), 0 if residual is not None else int(silu), _ptr(y), _ptr(stats), _stream()), "xm3d_group_norm")
    return y if residual is None else _act_torch(y + residual, silu)


def group_norm_nchw_stats(x, num_groups, weight, bias, eps, act):
    """training forward: xm3d_group_norm on a contiguous (B, C, ...) f32 tensor -> (act(GroupNorm(x)), the moments buffer its backward reads)"""
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // max(B * C, 1)
    y = torch.empty_like(x)
    stats = torch.empty(lib().xm3d_gn_stats_doubles_nchw(B, C, hw, num_groups), dtype=torch.float64, device=x.device)
    check(lib().xm3d_group_norm(_ptr(x), 0, B, C, hw, num_groups, _ptr(weight), _ptr(bias), float(eps), int(act), _ptr(y), _ptr(stats), _stream()),
          "xm3d_group_norm")
    return y, stats


def group_norm_bwd(x, dy, stats, num_groups, weight, bias, eps, act, need_affine):
    """-> (dx, dgamma | None, dbeta | None) of act(GroupNorm(x)) for contiguous (B, C, ...) f32 tensors (xm3d_group_norm_bwd)"""
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // max(B * C, 1)
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
    db = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
    ws = torch.empty(lib().xm3d_group_norm_bwd_ws_floats(B, C, num_groups), dtype=torch.float32, device=x.device)
    check(lib().xm3d_group_norm_bwd(_ptr(x), _ptr(dy), _ptr(stats), _ptr(weight), _ptr(bias), B, C, hw, num_groups, float(eps), int(act), _ptr(dx), _ptr(dg),
                                    _ptr(db), _ptr(ws), _stream()), "xm3d_group_norm_bwd")
    return dx, dg, db


def layer_norm_bwd(x, dy, weight, eps, need_affine):
    """-> (dx, dgamma | None, dbeta | None) of LayerNorm over the last dimension of contiguous f32 tensors (xm3d_layer_norm_bwd)"""
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
    db = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
    ws = torch.empty(lib().xm3d_layer_norm_bwd_ws_floats(rows, C), dtype=torch.float32, device=x.device) if need_affine else None
    check(lib().xm3d_layer_norm_bwd(_ptr(x), _ptr(dy), _ptr(weight), rows, C, float(eps), _ptr(dx), _ptr(dg), _ptr(db), _ptr(ws), _stream()),
          "xm3d_layer_norm_bwd")
    return dx, dg, db


def column_sum(x):
    """(rows, n) contiguous f32 device tensor -> (n,) column sums in a fixed order (xm3d_column_sum)"""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()):
        raise TypeError("column_sum: contiguous 2-D f32 device tensor required")
    rows, n = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib().xm3d_column_sum_ws_floats(rows, n), dtype=torch.float32, device=x.device)
    check(lib().xm3d_column_sum(_ptr(x), rows, n, n, _ptr(out), _ptr(ws), _stream()), "xm3d_column_sum")
    return out


def bias_residual(a, b, bias, stats_groups=None):
    """out = a + b + bias[c] for channels-last (B,C,H,W) f32/bf16 device tensors; a may be None.
    stats_groups: G of the GroupNorm expected to read the result (its statistics are then computed here, on the way)."""
    if not is_nhwc(b) or (a is not None and (not is_nhwc(a) or a.shape != b.shape or a.dtype != b.dtype)):
        raise TypeError("bias_residual: channels-last device tensors of one shape/dtype required")
    if b.dtype not in (torch.float32, torch.bfloat16) or bias.dtype != b.dtype or bias.numel() != b.shape[1]:
        raise TypeError("bias_residual: unsupported dtype / bias shape")
    out = torch.empty_like(b)
    Bn, C, H, W = b.shape
    if stats_groups and C % stats_groups == 0 and stats_groups <= 64:
        # also accumulate the GroupNorm statistics of the result: the GroupNorm that reads it next skips its statistics pass
        # (group_norm() picks them up from the tensor; they describe exactly this storage, so never modify it in place)
        stats = _gn_stats_buf(b, Bn, C, H * W, int(stats_groups))
        check(lib().xm3d_bias_residual_stats_nhwc(_ptr(a), _ptr(b), _ptr(bias.contiguous()), 0 if b.dtype == torch.float32 else 1, Bn, C, H * W,
                                                  int(stats_groups), _ptr(out), _ptr(stats), _stream()), "xm3d_bias_residual_stats_nhwc")
        out._xm3d_gn_stats = (stats[:Bn * stats_groups * 2], int(stats_groups), out.data_ptr())
        return out
    check(lib().xm3d_bias_residual_nhwc(_ptr(a), _ptr(b), _ptr(bias.contiguous()), 0 if b.dtype == torch.float32 else 1, Bn * H * W, C,
                                        _ptr(out), _stream()), "xm3d_bias_residual_nhwc")
    return out


# ---------------------------------------------------------------- fused GroupNorm -> SiLU -> conv3x3 (conv.hip)
def gn_stats_of(x, num_groups=32, shift=None):
    """(B*G*2,) f64 GroupNorm moments (sum, sum of squares per (sample, group)) of a channels-last tensor: taken from the tensor
    if the kernel that produced it left them there (conv3x3 / bias_residual epilogues), else by one statistics pass.
    shift: (C,) or (B, C) tensor of x's dtype: moments of x + shift (always a pass)."""
    pre = getattr(x, "_xm3d_gn_stats", None)
    if pre is not None and shift is None and pre[1] == num_groups and pre[2] == x.data_ptr():
        return pre[0]
    if not is_nhwc(x) or x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("gn_stats_of: channels-last f32/bf16 device tensor required")
    B, C, H, W = x.shape
    stats = _gn_stats_buf(x, B, C, H * W, num_groups)
    bstride = 0
    if shift is not None:
        shift = shift.to(x.dtype).contiguous()
        if shift.numel() not in (C, B * C):
            raise TypeError("gn_stats_of: shift must have C or B*C elements")
        bstride = C if (shift.numel() == B * C and B > 1) else 0
    check(lib().xm3d_group_norm_nhwc_stats(_ptr(x), _ptr(shift), bstride, 0 if x.dtype == torch.float32 else 1, B, C, H * W, num_groups, _ptr(stats),
                                           _stream()), "xm3d_group_norm_nhwc_stats")
    return stats[:B * num_groups * 2]


def conv3x3_supported(x, cout, upsample=False):
    """shapes xm3d_conv3x3_nhwc takes: channels-last bf16, output H % 4 == 0, W % 32 == 0, cin % 64 == 0, cout % 32 == 0 (cout >= 128:
    a 64-channel layer would leave half of its only tile empty)"""
    if not (is_nhwc(x) and x.dtype == torch.bfloat16):
        return False
    _, cin, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    return H % 4 == 0 and W % 32 == 0 and cin % 64 == 0 and cout % 32 == 0 and cout >= 128


def conv3x3_pack_weight(weight):
    """Conv2d weight (cout, cin, 3, 3), any float dtype / memory format -> (packed bf16 tensor, cout tile)"""
    cout, cin = weight.shape[0], weight.shape[1]
    tile = lib().xm3d_conv3x3_cout_tile(cout)
    if tile == 0 or cin % 64 != 0 or tuple(weight.shape[2:]) != (3, 3):
        raise TypeError(f"conv3x3_pack_weight: unsupported weight shape {tuple(weight.shape)}")
    dt = torch.float16 if weight.dtype == torch.float16 else torch.bfloat16  # halves: a term of the f32-accurate split (the packer moves 16-bit words)
    w = weight.detach().to(dt).permute(0, 2, 3, 1).contiguous()  # OHWI
    packed = torch.empty(lib().xm3d_conv3x3_packed_elems(cout, cin, tile), dtype=dt, device=w.device)
    check(lib().xm3d_conv3x3_pack_weight(_ptr(w), cout, cin, tile, _ptr(packed), _stream()), "xm3d_conv3x3_pack_weight")
    return packed, tile


def conv3x3_f32_supported(x, cout, upsample=False):
    """shapes the f32-accurate convolution takes: channels-last f32, the kernel's tile constraints"""
    if not (is_nhwc(x) and x.dtype == torch.float32):
        return False
    _, cin, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    return H % 4 == 0 and W % 32 == 0 and cin % 64 == 0 and cout % 32 == 0 and cout >= 128


F16_LO_SCALE = 2048.0   # the second term of a split in halves is stored times 2^11 (both terms at one magnitude)
F16_X_SCALE = 2.0 ** -6  # activations are split as hi = half(x * 2^-6): |x| up to 4.2e6 before the half overflows (flagged by the kernel)


def split_f16(t, scale_hi=1.0, lo_mul=F16_LO_SCALE):
    """f32 tensor -> (hi, lo) halves with t = hi / scale_hi + lo / (scale_hi * lo_mul) to 2^-22 |t| (torch ops; weights, once)"""
    t = t.detach().float()
    hi = (t * scale_hi).to(torch.float16)
    lo = ((t - hi.float() / scale_hi) * (scale_hi * lo_mul)).to(torch.float16)
    return hi, lo


def conv3x3_pack_weight_split(weight, terms=3):
    """f32 Conv2d weight (cout, cin, 3, 3) -> ([packed term 0, 1(, 2)], cout tile): the bf16 split w = t0 + t1 (+ t2) of the f32-accurate
    convolution (two terms: 2^-18 |w| left over; three: 2^-25); terms = "f16": the two-term split in halves (split_f16: 2^-22)"""
    if terms == "f16":
        hi, lo = split_f16(weight)
        p0, tile = conv3x3_pack_weight(hi)
        p1, _ = conv3x3_pack_weight(lo)
        return [p0, p1], tile
    r = weight.detach().float()
    packs, tile = [], None
    for _ in range(terms):
        t = r.to(torch.bfloat16)
        r = r - t.float()
        p_, tile = conv3x3_pack_weight(t)
        packs.append(p_)
    return packs, tile


def conv3x3_f32(x, packs, cout, tile, bias=None, gn=None, residual=None, stats_groups=None, upsample=False, in_shift=None, waves=0):
    """conv3x3(act(GroupNorm(x))) + bias (+ residual) on channels-last f32 tensors, to f32 accuracy, on the bf16 matrix cores: one split
    pass (x -> bf16 terms, the GroupNorm affine + activation applied on the way) and accumulating convolution launches over the
    term pairs - packs in halves (conv3x3_pack_weight_split(w, "f16"), the default): two terms of 11 bits each, three passes, ~1e-6 (the
    rounding level of an f32 convolution); bf16 packs: len(packs) == 2: x0 w0 + x0 w1 + x1 w0 (2e-5 of max|out|); 3: + x0 w2 + x2 w0 + x1 w1
    (~1e-6 in six passes).  Arguments as conv3x3; bias / residual / result f32."""
    if not conv3x3_f32_supported(x, cout, upsample):
        raise TypeError(f"conv3x3_f32: unsupported input {tuple(x.shape)} {x.dtype}")
    f16 = packs[0].dtype == torch.float16
    terms = len(packs)
    B, cin, Hi, Wi = x.shape
    H, W = (2 * Hi, 2 * Wi) if upsample else (Hi, Wi)
    xt = [torch.empty((B, cin, Hi, Wi), dtype=packs[0].dtype, device=x.device, memory_format=torch.channels_last) for _ in range(terms)]
    stats_in = gamma = beta = ws = None
    eps, G, act, sstride = 0.0, 0, 0, 0
    if gn is not None:
        stats_in, gamma, beta, eps, G = gn[:5]
        act = 2 if (len(gn) > 5 and gn[5] == "relu") else 1
        ws = torch.empty(B * cin * 2, dtype=torch.float32, device=x.device)
        if in_shift is not None:
            sstride = cin if (in_shift.numel() == B * cin and B > 1) else 0
    elif in_shift is not None:
        raise TypeError("conv3x3_f32: in_shift needs gn")
    if f16:
        check(lib().xm3d_split_f16_nhwc(_ptr(x), B, Hi * Wi, cin, _ptr(stats_in), _ptr(gamma), _ptr(beta), _ptr(in_shift), sstride, float(eps), int(G), act,
                                        F16_X_SCALE, _ptr(xt[0]), _ptr(xt[1]), _ptr(ws), _stream()), "xm3d_split_f16_nhwc")
    else:
        check(lib().xm3d_split_bf16_nhwc(_ptr(x), B, Hi * Wi, cin, _ptr(stats_in), _ptr(gamma), _ptr(beta), _ptr(in_shift), sstride, float(eps), int(G), act,
                                         _ptr(xt[0]), _ptr(xt[1]), _ptr(xt[2]) if terms == 3 else None, _ptr(ws), _stream()), "xm3d_split_bf16_nhwc")
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    bstride = 0
    if bias is not None:
        if bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() not in (cout, B * cout):
            raise TypeError("conv3x3_f32: bias must be a contiguous f32 (cout,) or (B, cout) tensor")
        bstride = cout if (bias.numel() == B * cout and B > 1) else 0
    if residual is not None and not (is_nhwc(residual) and residual.dtype == torch.float32 and residual.shape == out.shape):
        raise TypeError("conv3x3_f32: residual must be a channels-last f32 tensor of the output's shape")
    u, wv = int(bool(upsample)), int(waves) or _CONV_WAVES
    stats_out = torch.empty(lib().xm3d_conv3x3_stats_doubles(B, H, W, cout, tile, int(stats_groups), wv), dtype=torch.float64,
                            device=x.device) if stats_groups else None
    L = lib().xm3d_conv3x3_nhwc_f32acc2
    if f16:
        # (x term, w term, alpha): the two small products first, the leading one last (statistics of the final values in its epilogue)
        a_small = 1.0 / (F16_X_SCALE * F16_LO_SCALE)
        pairs = [(0, 1, a_small), (1, 0, a_small), (0, 0, 1.0 / F16_X_SCALE)]
    else:
        pairs = [(0, 0, 1.0), (0, 1, 1.0), (1, 0, 1.0)] + ([(0, 2, 1.0), (2, 0, 1.0), (1, 1, 1.0)] if terms == 3 else [])
    for n, (i, j, alpha) in enumerate(pairs):
        first, last = n == 0, n == len(pairs) - 1
        check(L(_ptr(xt[i]), B, H, W, cin, _ptr(packs[j]), cout, tile, _ptr(bias) if first else None, bstride if first else 0,
                _ptr(residual) if first else _ptr(out), _ptr(out), _ptr(stats_out) if last else None, int(stats_groups or 0) if last else 0, u, wv,
                int(f16), float(alpha), _stream()), "xm3d_conv3x3_nhwc_f32acc2")
    if stats_out is not None:
        out._xm3d_gn_stats = (stats_out[:B * stats_groups * 2], int(stats_groups), out.data_ptr())
    return out


# ---- linear layer / 1x1 convolution with fused epilogue (csrc/gemm.hip)
GEMM_ACTS = {None: 0, "none": 0, "gelu": 1, "quick_gelu": 2, "geglu": 3, "relu": 4}


def gemm_supported(x, n_rows, k=None):
    """x: (..., K) bf16 on the GPU whose rows are contiguous and evenly strided; K % 64, N % 32"""
    k = x.shape[-1] if k is None else k
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() >= 2 and x.shape[-1] == k and k % 64 == 0 and n_rows % 32 == 0
            and x.numel() > 0 and _rows_of(x) is not None and x.data_ptr() % 16 == 0)


def _rows_of(x):
    """(M, row stride) if x (..., K) is a stack of rows at a constant stride (contiguous tensors and column slices of one), else None"""
    if x.stride(-1) != 1:
        return None
    if x.dim() == 2:
        ld = x.stride(0)
        return (x.shape[0], ld) if ld % 8 == 0 and ld >= x.shape[1] else None
    ld = x.stride(-2)
    expect = ld
    for d in range(x.dim() - 2, -1, -1):
        if x.shape[d] != 1 and x.stride(d) != expect:
            return None
        expect *= x.shape[d]
    return (x.numel() // x.shape[-1], ld) if ld % 8 == 0 and ld >= x.shape[-1] else None


def gemm_pack_weight(weight, act=None):
    """Linear / 1x1 Conv2d weight (N, K[, 1, 1]), f32 or bf16 -> (packed bf16 tensor, column tile) for gemm(..., act=act).
    The packed image depends on `act` only through "geglu" (value / gate rows interleaved)."""
    w = weight.detach().reshape(weight.shape[0], -1)
    n, k = w.shape
    if k % 64 != 0 or n % 32 != 0 or w.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"gemm_pack_weight: unsupported weight {tuple(weight.shape)} {weight.dtype}")
    w = w.contiguous()
    tile = lib().xm3d_gemm_col_tile(n)
    packed = torch.empty(lib().xm3d_gemm_packed_elems(n, k, tile), dtype=torch.bfloat16, device=w.device)
    check(lib().xm3d_gemm_pack_weight(_ptr(w), int(w.dtype == torch.float32), n, k, 3 if act == "geglu" else 0, tile, _ptr(packed), _stream()),
          "xm3d_gemm_pack_weight")
    return packed, tile


_GEMM_WAVES_GEGLU = _os.environ.get("XM3D_GEMM_WAVES_GEGLU", "0") == "1"  # tuning runs: let XM3D_GEMM_WAVES also move the GEGLU launches
_GEMM_WAVES = int(_os.environ.get("XM3D_GEMM_WAVES", "0"))  # A/B switch for bench runs: force one workgroup geometry on every plain GEMM (0 = per-shape choice)


def gemm(x, packed, n_rows, tile, bias=None, act=None, residual=None, waves=0):
    """out = act(x @ W^T + bias) (+ residual) over the last dimension of x; "geglu": out = (x Wv^T + bv) * GELU(x Wg^T + bg) with
    W = [Wv; Wg] (n_rows = 2 * out features).  x (..., K) bf16, bias f32 (n_rows) or None, residual like the output or None.
    waves: 0 = the library's choice of workgroup geometry, 8 / 4 (same results)."""
    k = x.shape[-1]
    rows = _rows_of(x) if x.is_cuda and x.dtype == torch.bfloat16 else None
    if rows is None or k % 64 != 0:
        raise TypeError(f"gemm: unsupported input {tuple(x.shape)} {x.dtype} strides {x.stride()}")
    m, ldx = rows
    if waves == 0 and _GEMM_WAVES and (act != "geglu" or _GEMM_WAVES_GEGLU):
        waves = _GEMM_WAVES
    a = GEMM_ACTS[act]
    nout = n_rows // 2 if a == 3 else n_rows
    out = torch.empty(x.shape[:-1] + (nout,), dtype=torch.bfloat16, device=x.device)
    ldr = 0
    if residual is not None:
        rr = _rows_of(residual) if residual.dtype == torch.bfloat16 and residual.shape == out.shape else None
        if rr is None:
            raise TypeError("gemm: residual must be a bf16 tensor of the output's shape with evenly strided rows")
        ldr = rr[1]
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != n_rows):
        raise TypeError("gemm: bias must be a contiguous f32 (n_rows,) tensor")
    if tile == 256 and (waves == 4 or (waves == 0 and ((m + 255) // 256) * (n_rows // 256) < 200)):
        tile = 128  # fewer 256 x 256 tiles than CUs: 128-row, 4-wave workgroups on 128-column tiles (the packed image serves both)
    check(lib().xm3d_gemm_bf16(_ptr(x), m, k, ldx, _ptr(packed), n_rows, tile, _ptr(bias), a, _ptr(residual), ldr, _ptr(out), nout, int(waves), _stream()),
          "xm3d_gemm_bf16")
    return out


def conv_gemm_supported(x, weight_shape):
    """channels-last bf16 image and a Conv2d weight (cout, cin, k, k), k <= 3, cin % 64 == 0 (xm3d_conv_gemm_bf16)"""
    return (is_nhwc(x) and x.dtype == torch.bfloat16 and len(weight_shape) == 4 and weight_shape[2] == weight_shape[3] and weight_shape[2] in (1, 2, 3)
            and weight_shape[1] == x.shape[1] and weight_shape[1] % 64 == 0)


def conv_gemm_pack_weight(weight):
    """Conv2d weight (cout, cin, k, k) -> (packed bf16 image, column tile, padded cout): rows in (ky, kx, cin) order, cout padded with
    zero rows to a multiple of 32"""
    cout, cin, k, _ = weight.shape
    w = weight.detach().float().permute(0, 2, 3, 1).reshape(cout, k * k * cin)
    n32 = (cout + 31) // 32 * 32
    if n32 != cout:
        w = torch.cat([w, torch.zeros(n32 - cout, w.shape[1], dtype=w.dtype, device=w.device)])
    packed, tile = gemm_pack_weight(w.contiguous())
    return packed, tile, n32


def conv_gemm(x, packed, tile, n32, cout, ksize, stride=1, padding=(0, 0, 0, 0), bias=None, residual=None):
    """conv2d(x, W, stride, zero padding (top, left, bottom, right)) + bias (+ residual) as an implicit GEMM (csrc/gemm.hip GF_CONV):
    x (B, cin, H, W) channels-last bf16 -> (B, cout, Ho, Wo) channels-last bf16.  bias f32 (n32,) or None; residual like the output
    (cout == n32 only).  Bit-reproducible (deterministic split-K on small grids)."""
    if not (is_nhwc(x) and x.dtype == torch.bfloat16):
        raise TypeError(f"conv_gemm: channels-last bf16 device tensor required, got {tuple(x.shape)} {x.dtype}")
    B, cin, H, W = x.shape
    pt, pl, pb, pr = (int(p) for p in padding)
    Ho, Wo = (H + pt + pb - ksize) // stride + 1, (W + pl + pr - ksize) // stride + 1
    out = torch.empty((B, Ho, Wo, n32), dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        if n32 != cout or residual.dtype != torch.bfloat16 or tuple(residual.shape) != (B, cout, Ho, Wo) or not is_nhwc(residual):
            raise TypeError("conv_gemm: residual must be a channels-last bf16 tensor of the output's shape")
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != n32):
        raise TypeError("conv_gemm: bias must be a contiguous f32 (padded cout,) tensor")
    nb = lib().xm3d_conv_gemm_ws_bytes(B * Ho * Wo, n32, ksize * ksize * cin, tile)
    ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device) if nb else None
    check(lib().xm3d_conv_gemm_bf16(_ptr(x), B, H, W, cin, _ptr(packed), n32, tile, ksize, stride, pt, pl, Ho, Wo, _ptr(bias), _ptr(residual), _ptr(out),
                                    _ptr(ws), _stream()), "xm3d_conv_gemm_bf16")
    img = out.permute(0, 3, 1, 2)
    return img if n32 == cout else img[:, :cout].contiguous(memory_format=torch.channels_last)


# ---- f32-accurate GEMM / implicit-GEMM convolution: three matrix-core passes over operands split in IEEE halves (xm3d_gemm_f32acc)
F16T_X_SCALE = 16.0  # one-launch form: activations split as x * 16 = hi + lo (both halves at one scale): |x| up to 4094, lo normal for |x| >= 2^-7


def gemm_pack_weight_f16(weight, one_scale=False):
    """f32 Linear / Conv2d weight (N, K) or (N, cin, k, k) -> ([packed hi, packed lo], column tile, padded N): the two-term split in
    halves (split_f16); convolution weights go in (ky, kx, cin) order, N is padded with zero rows to a multiple of 32.
    one_scale: the operand form of the one-launch GEMM (gemm_f32_fused): W t = hi + lo with t the largest power of two that keeps
    |W| t <= 32768 -> ([packed hi, packed lo], 128, padded N, t)"""
    w = weight.detach().float()
    if w.dim() == 4:
        w = w.permute(0, 2, 3, 1)
    w = w.reshape(w.shape[0], -1)
    n32 = (w.shape[0] + 31) // 32 * 32
    if n32 != w.shape[0]:
        w = torch.cat([w, torch.zeros(n32 - w.shape[0], w.shape[1], dtype=w.dtype, device=w.device)])
    packs, tile = [], None
    sw = 1.0
    if one_scale:
        amax = float(w.abs().max()) if w.numel() else 0.0
        sw = 2.0 ** math.floor(math.log2(32768.0 / amax)) if amax > 0 else 1.0
        sw = min(max(sw, 2.0 ** -14), 2.0 ** 24)
    for t in split_f16(w, sw, 1.0 if one_scale else F16_LO_SCALE):
        # the packer moves 16-bit words (its bf16 -> f32 -> bf16 round trip is the identity on them)
        p_, tile = gemm_pack_weight(t.contiguous().view(torch.bfloat16))
        packs.append(p_)
    return (packs, 128, n32, sw) if one_scale else (packs, tile, n32)


def _split_rows_f16(x2):
    """(M, K) contiguous f32 -> (hi, lo) halves at F16_X_SCALE (one pass)"""
    m, k = x2.shape
    hi, lo = torch.empty((m, k), dtype=torch.float16, device=x2.device), torch.empty((m, k), dtype=torch.float16, device=x2.device)
    check(lib().xm3d_split_f16_nhwc(_ptr(x2), 1, m, k, None, None, None, None, 0, 0.0, 0, 0, F16_X_SCALE, _ptr(hi), _ptr(lo), None, _stream()),
          "xm3d_split_f16_nhwc")
    return hi, lo


def _split_rows_f16t(x2):
    """(M, K) contiguous f32 -> (hi, lo) halves with x * F16T_X_SCALE = hi + lo (one pass)"""
    m, k = x2.shape
    hi, lo = torch.empty((m, k), dtype=torch.float16, device=x2.device), torch.empty((m, k), dtype=torch.float16, device=x2.device)
    check(lib().xm3d_split_f16t_nhwc(_ptr(x2), 1, m, k, None, None, None, None, 0, 0.0, 0, 0, F16T_X_SCALE, _ptr(hi), _ptr(lo), None, _stream()),
          "xm3d_split_f16t_nhwc")
    return hi, lo


# the one-launch f32-accurate GEMM / convolution splits the f32 activation while it stages it (xm3d_gemm_f32x); XM3D_GEMM_F32_SPLIT=pass restores
# the separate split pass (xm3d_split_f16t_nhwc + xm3d_gemm_f32: the same bits) for A/B timing
_F32_SPLIT_IN_KERNEL = _os.environ.get("XM3D_GEMM_F32_SPLIT", "kernel") != "pass"


def gemm_f32_fused(x, packs, n, sw, bias=None, act=None, residual=None):
    """gemm_f32 in ONE launch (xm3d_gemm_f32): packs, sw = gemm_pack_weight_f16(W, one_scale=True)[0, 3].  A third of the traffic of the three
    accumulating passes; activations beyond |x| = 4094 raise the sticky range flag."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 64 == 0):
        raise TypeError(f"gemm_f32_fused: f32 device tensor with K % 64 == 0 required, got {tuple(x.shape)} {x.dtype}")
    k = x.shape[-1]
    x2 = x.reshape(-1, k)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    m = x2.shape[0]
    out = torch.empty(x.shape[:-1] + (n,), dtype=torch.float32, device=x.device)
    ldr = 0
    if residual is not None:
        if residual.dtype != torch.float32 or residual.shape != out.shape or not residual.is_contiguous():
            raise TypeError("gemm_f32_fused: residual must be a contiguous f32 tensor of the output's shape")
        ldr = n
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != n):
        raise TypeError("gemm_f32_fused: bias must be a contiguous f32 (n,) tensor")
    if _F32_SPLIT_IN_KERNEL and x2.data_ptr() % 16 == 0:
        check(lib().xm3d_gemm_f32x(_ptr(x2), F16T_X_SCALE, m, k, k, _ptr(packs[0]), _ptr(packs[1]), n, _ptr(bias), GEMM_ACTS[act], 1.0 / (F16T_X_SCALE * sw),
                                   _ptr(residual), ldr, _ptr(out), n, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, _stream()), "xm3d_gemm_f32x")
        return out
    hi, lo = _split_rows_f16t(x2)
    check(lib().xm3d_gemm_f32(_ptr(hi), _ptr(lo), m, k, k, _ptr(packs[0]), _ptr(packs[1]), n, _ptr(bias), GEMM_ACTS[act], 1.0 / (F16T_X_SCALE * sw),
                              _ptr(residual), ldr, _ptr(out), n, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, _stream()), "xm3d_gemm_f32")
    return out


def conv_gemm_f32_fused(x, packs, n32, sw, cout, ksize, stride=1, padding=(0, 0, 0, 0), bias=None, residual=None):
    """conv_gemm_f32 in ONE launch (xm3d_gemm_f32, convolution form); packs, n32, sw from gemm_pack_weight_f16(W, one_scale=True)"""
    if not (is_nhwc(x) and x.dtype == torch.float32 and x.shape[1] % 64 == 0):
        raise TypeError(f"conv_gemm_f32_fused: channels-last f32 device tensor with cin % 64 == 0 required, got {tuple(x.shape)} {x.dtype}")
    B, cin, H, W = x.shape
    pt, pl, pb, pr = (int(p) for p in padding)
    Ho, Wo = (H + pt + pb - ksize) // stride + 1, (W + pl + pr - ksize) // stride + 1
    out = torch.empty((B, Ho, Wo, n32), dtype=torch.float32, device=x.device)
    ldr = 0
    if residual is not None:
        if n32 != cout or residual.dtype != torch.float32 or tuple(residual.shape) != (B, cout, Ho, Wo) or not is_nhwc(residual):
            raise TypeError("conv_gemm_f32_fused: residual must be a channels-last f32 tensor of the output's shape")
        ldr = n32
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != n32):
        raise TypeError("conv_gemm_f32_fused: bias must be a contiguous f32 (padded cout,) tensor")
    if _F32_SPLIT_IN_KERNEL and x.data_ptr() % 16 == 0:
        check(lib().xm3d_gemm_f32x(_ptr(x), F16T_X_SCALE, B * Ho * Wo, ksize * ksize * cin, 0, _ptr(packs[0]), _ptr(packs[1]), n32, _ptr(bias), 0,
                                   1.0 / (F16T_X_SCALE * sw), _ptr(residual), ldr, _ptr(out), n32, 0, 1, B, H, W, cin, ksize, stride, pt, pl, Ho, Wo,
                                   _stream()), "xm3d_gemm_f32x")
    else:
        hi = torch.empty((B, cin, H, W), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
        lo = torch.empty_like(hi)
        check(lib().xm3d_split_f16t_nhwc(_ptr(x), B, H * W, cin, None, None, None, None, 0, 0.0, 0, 0, F16T_X_SCALE, _ptr(hi), _ptr(lo), None, _stream()),
              "xm3d_split_f16t_nhwc")
        check(lib().xm3d_gemm_f32(_ptr(hi), _ptr(lo), B * Ho * Wo, ksize * ksize * cin, 0, _ptr(packs[0]), _ptr(packs[1]), n32, _ptr(bias), 0,
                                  1.0 / (F16T_X_SCALE * sw), _ptr(residual), ldr, _ptr(out), n32, 0, 1, B, H, W, cin, ksize, stride, pt, pl, Ho, Wo, _stream()),
              "xm3d_gemm_f32")
    img = out.permute(0, 3, 1, 2)
    return img if n32 == cout else img[:, :cout].contiguous(memory_format=torch.channels_last)


def _f32acc_passes(L_args, terms, packs, out, bias, act, residual, ldr):
    """the three term pairs of x w: small products first, the leading one - with the activation and the residual - last"""
    a_small = 1.0 / (F16_X_SCALE * F16_LO_SCALE)
    for n, (i, j, alpha) in enumerate(((0, 1, a_small), (1, 0, a_small), (0, 0, 1.0 / F16_X_SCALE))):
        first, last = n == 0, n == 2
        L_args(terms[i], packs[j], _ptr(bias) if first else None, GEMM_ACTS[act] if last else 0, float(alpha), None if first else _ptr(out),
               _ptr(residual) if last else None, ldr)


def gemm_f32(x, packs, n, tile, bias=None, act=None, residual=None):
    """act(x @ W^T + bias) (+ residual) over the last dimension of an f32 tensor, to f32 accuracy (~1e-6), on the matrix cores.
    packs: gemm_pack_weight_f16(W); n: its (padded) row count; act None | "gelu" | "quick_gelu"; bias f32 (n); residual f32 like the output."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 64 == 0):
        raise TypeError(f"gemm_f32: f32 device tensor with K % 64 == 0 required, got {tuple(x.shape)} {x.dtype}")
    k = x.shape[-1]
    x2 = x.reshape(-1, k)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    m = x2.shape[0]
    terms = _split_rows_f16(x2)
    out = torch.empty(x.shape[:-1] + (n,), dtype=torch.float32, device=x.device)
    ldr = 0
    if residual is not None:
        if residual.dtype != torch.float32 or residual.shape != out.shape or not residual.is_contiguous():
            raise TypeError("gemm_f32: residual must be a contiguous f32 tensor of the output's shape")
        ldr = n
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != n):
        raise TypeError("gemm_f32: bias must be a contiguous f32 (n,) tensor")

    def run(xt, wp, b, a, alpha, accin, res, ldr_):
        check(lib().xm3d_gemm_f32acc(_ptr(xt), m, k, k, _ptr(wp), n, tile, b, a, alpha, accin, res, ldr_, _ptr(out), n, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                     _stream()), "xm3d_gemm_f32acc")

    _f32acc_passes(run, terms, packs, out, bias, act, residual, ldr)
    return out


def conv_gemm_f32(x, packs, tile, n32, cout, ksize, stride=1, padding=(0, 0, 0, 0), bias=None, residual=None):
    """conv2d(x, W, stride, zero padding (top, left, bottom, right)) + bias (+ residual) on a channels-last f32 image, to f32 accuracy:
    the implicit-GEMM kernel over operands split in halves (three passes).  packs: gemm_pack_weight_f16(W)."""
    if not (is_nhwc(x) and x.dtype == torch.float32 and x.shape[1] % 64 == 0):
        raise TypeError(f"conv_gemm_f32: channels-last f32 device tensor with cin % 64 == 0 required, got {tuple(x.shape)} {x.dtype}")
    B, cin, H, W = x.shape
    pt, pl, pb, pr = (int(p) for p in padding)
    Ho, Wo = (H + pt + pb - ksize) // stride + 1, (W + pl + pr - ksize) // stride + 1
    hi = torch.empty((B, cin, H, W), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    lo = torch.empty_like(hi)
    check(lib().xm3d_split_f16_nhwc(_ptr(x), B, H * W, cin, None, None, None, None, 0, 0.0, 0, 0, F16_X_SCALE, _ptr(hi), _ptr(lo), None, _stream()),
          "xm3d_split_f16_nhwc")
    out = torch.empty((B, Ho, Wo, n32), dtype=torch.float32, device=x.device)
    ldr = 0
    if residual is not None:
        if n32 != cout or residual.dtype != torch.float32 or tuple(residual.shape) != (B, cout, Ho, Wo) or not is_nhwc(residual):
            raise TypeError("conv_gemm_f32: residual must be a channels-last f32 tensor of the output's shape")
        ldr = n32
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != n32):
        raise TypeError("conv_gemm_f32: bias must be a contiguous f32 (padded cout,) tensor")
    m, k = B * Ho * Wo, ksize * ksize * cin

    def run(xt, wp, b, a, alpha, accin, res, ldr_):
        check(lib().xm3d_gemm_f32acc(_ptr(xt), m, k, 0, _ptr(wp), n32, tile, b, a, alpha, accin, res, ldr_, _ptr(out), n32, 0, 1, B, H, W, cin, ksize, stride,
                                     pt, pl, Ho, Wo, _stream()), "xm3d_gemm_f32acc")

    _f32acc_passes(run, (hi, lo), packs, out, bias, None, residual, ldr)
    img = out.permute(0, 3, 1, 2)
    return img if n32 == cout else img[:, :cout].contiguous(memory_format=torch.channels_last)


_CONV_WAVES = int(_os.environ.get("XM3D_CONV_WAVES", "0"))  # A/B switch for bench runs: force one workgroup geometry (0 = per-layer choice)


def conv3x3(x, packed, cout, tile, bias=None, gn=None, residual=None, stats_groups=None, upsample=False, in_shift=None, waves=0):
    """out = conv3x3(SiLU(GroupNorm(x))) + bias (+ residual), channels-last bf16 (B, C, H, W) in and out.
    packed, tile: conv3x3_pack_weight(weight).  gn: None (plain convolution) or (stats f64 (B*G*2), gamma f32 (cin), beta f32 (cin),
    eps, G[, "relu"]) - the activation behind the GroupNorm is SiLU unless "relu" is given.  bias: None, (cout,) or (B, cout) f32.  residual: tensor like the output.  stats_groups: G of the GroupNorm that reads
    the result next - its moments are accumulated in the epilogue and attached to the returned tensor (gn_stats_of picks them up).
    upsample: x is nearest-upsampled 2x first (plain convolution only).
    in_shift: (cin,) or (B, cin) f32 added to x in front of the GroupNorm (gn's moments must be those of x + in_shift).
    waves: 0 = the library's choice, 8 / 4 = workgroup geometry (same results)."""
    if not conv3x3_supported(x, cout, upsample):
        raise TypeError(f"conv3x3: unsupported input {tuple(x.shape)} {x.dtype} (channels-last bf16, H % 4, W % 32, cin % 64, cout % 32)")
    B, cin, H, W = x.shape
    if upsample:
        H, W = 2 * H, 2 * W
    out = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    bstride = 0
    if bias is not None:
        if bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() not in (cout, B * cout):
            raise TypeError("conv3x3: bias must be a contiguous f32 (cout,) or (B, cout) tensor")
        bstride = cout if (bias.numel() == B * cout and B > 1) else 0
    if residual is not None and not (is_nhwc(residual) and residual.dtype == torch.bfloat16 and residual.shape == out.shape):
        raise TypeError("conv3x3: residual must be a channels-last bf16 tensor of the output's shape")
    stats_in = gamma = beta = None
    eps, G, act = 0.0, 0, 0
    if gn is not None:
        stats_in, gamma, beta, eps, G = gn[:5]
        if stats_in.dtype != torch.float64 or stats_in.numel() != B * G * 2 or gamma.dtype != torch.float32 or beta.dtype != torch.float32 \
                or gamma.numel() != cin or beta.numel() != cin:
            raise TypeError("conv3x3: gn = (f64 moments (B*G*2), f32 gamma (cin), f32 beta (cin), eps, G)")
        act = 2 if (len(gn) > 5 and gn[5] == "relu") else 1
    sstride = 0
    if in_shift is not None:
        if gn is None or in_shift.dtype != torch.float32 or not in_shift.is_contiguous() or in_shift.numel() not in (cin, B * cin):
            raise TypeError("conv3x3: in_shift needs gn and must be a contiguous f32 (cin,) or (B, cin) tensor")
        sstride = cin if (in_shift.numel() == B * cin and B > 1) else 0
    stats_out = ws = None
    if stats_groups:
        stats_out = torch.empty(lib().xm3d_conv3x3_stats_doubles(B, H, W, cout, tile, int(stats_groups), int(waves) or _CONV_WAVES),
                                dtype=torch.float64, device=x.device)
    if gn is not None:
        ws = torch.empty(B * cin * 2, dtype=torch.float32, device=x.device)
    check(lib().xm3d_conv3x3_nhwc(_ptr(x), B, H, W, cin, _ptr(packed), cout, tile, _ptr(stats_in), _ptr(gamma), _ptr(beta), _ptr(in_shift), sstride,
                                  float(eps), int(G), act, _ptr(bias), bstride, _ptr(residual), _ptr(out), _ptr(stats_out), int(stats_groups or 0), int(bool(upsample)),
                                  int(waves) or _CONV_WAVES, _ptr(ws), _stream()), "xm3d_conv3x3_nhwc")
    if stats_out is not None:
        out._xm3d_gn_stats = (stats_out[:B * stats_groups * 2], int(stats_groups), out.data_ptr())
    return out


def attn_mask_bias(logits, size, out_dtype):
    """mask logits (B,Q,H,W) f32/bf16 -> additive attention bias (B,Q,h*w) of dtype out_dtype (0 / -inf), see xm3d.h"""
    if not logits.is_cuda or logits.dtype not in (torch.float32, torch.bfloat16) or out_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("attn_mask_bias: f32/bf16 device tensors required")
    logits = logits.contiguous()
    B, Q, H, W = logits.shape
    h, w = int(size[0]), int(size[1])
    out = torch.empty((B, Q, h * w), dtype=out_dtype, device=logits.device)
    check(lib().xm3d_attn_mask_bias(_ptr(logits), 0 if logits.dtype == torch.float32 else 1, B * Q, H, W, h, w, _ptr(out),
                                    0 if out_dtype == torch.float32 else 1, _stream()), "xm3d_attn_mask_bias")
    return out


def point_class_supported(x, text):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and text.dtype == torch.float32 and text.dim() == 2 and text.shape[0] <= 32 and text.shape[1] == x.shape[1]
            and x.shape[1] % 8 == 0 and x.shape[1] <= 1024 and not torch.is_grad_enabled())


def point_class(x, text, binary_pred, base_mask, novel_mask, row_index=None, ensemble=None):
    """Class label per point in one pass over the (rows, K) f32 features x (see xm3d.h, xm3d_point_class).  text (C, K): UNIT rows.
    binary_pred (Np) int64, base_mask / novel_mask (C) bool.  row_index (Np) int64: point p reads row row_index[p].
    ensemble None: arg-max of the gated products.  ensemble = (scale 0-dim f32 tensor, masks (Np, Q) bool, vid (Np) int64,
    open_p (B, Q, C) f32, overlap (C) f32, base_ratio, novel_ratio): the fused prediction chain.  -> (Np) int64"""
    n = int(binary_pred.shape[0])
    C, K = text.shape
    text = text.contiguous()
    label = torch.empty(n, dtype=torch.int64, device=x.device)
    if binary_pred.dtype != torch.int64 or base_mask.dtype != torch.bool or novel_mask.dtype != torch.bool:
        raise TypeError("point_class: binary_pred int64, base_mask / novel_mask bool")
    if row_index is not None and (row_index.dtype != torch.int64 or row_index.numel() != n):
        raise TypeError("point_class: row_index must be int64 (Np)")
    if row_index is None and x.shape[0] != n:
        raise TypeError("point_class: one feature row per point (or a row_index)")
    mode, masks, Q, vid, open_p, overlap, scale, br, nr = 0, None, 0, None, None, None, None, 0.0, 0.0
    if ensemble is not None:
        scale, masks, vid, open_p, overlap, br, nr = ensemble
        if masks.dtype != torch.bool or not masks.is_contiguous() or masks.shape[0] != n or vid.dtype != torch.int64 or open_p.dtype != torch.float32 \
                or overlap.dtype != torch.float32 or scale.dtype != torch.float32 or scale.numel() != 1:
            raise TypeError("point_class: ensemble = (f32 scalar tensor, bool (Np,Q) masks, int64 vid, f32 open_p (B,Q,C), f32 overlap (C), r_base, r_novel)")
        mode, Q, open_p = 1, int(masks.shape[1]), open_p.contiguous()
    check(lib().xm3d_point_class(_ptr(x), x.stride(0), _ptr(row_index), n, _ptr(text), C, K, _ptr(scale), _ptr(binary_pred), _ptr(base_mask),
                                 _ptr(novel_mask), mode, _ptr(masks), Q, _ptr(vid), _ptr(open_p), _ptr(overlap), float(br), float(nr), _ptr(label),
                                 _stream()), "xm3d_point_class")
    return label


def mask_heads_supported(mask_embed, mask_features, size):
    """the fused prediction heads take: channels-last bf16 mask_features (B, 256, H, W), Q <= 64 queries, an even integer shrink to
    `size` that divides the kernel's 32-pixel segments; inference only"""
    if torch.is_grad_enabled() or not mask_features.is_cuda or mask_features.dtype != torch.bfloat16 or mask_features.dim() != 4:
        return False
    B, C, H, W = mask_features.shape
    h, w = int(size[0]), int(size[1])
    return (C == 256 and mask_features.is_contiguous(memory_format=torch.channels_last) and mask_embed.dim() == 3 and mask_embed.shape[0] == B
            and mask_embed.shape[1] <= 64 and mask_embed.shape[2] == C and attn_mask_bias_supported((H, W), (h, w)) and W % 32 == 0
            and 32 % (W // w) == 0)


def mask_logits_bias(mask_embed, mask_features, size, want_logits=True, bias_dtype=torch.float32):
    """einsum("bqc,bchw->bqhw") on the matrix cores + the additive attention bias of the (h, w) level from the logits while they are
    on chip (see xm3d.h).  -> (logits (B,Q,H,W) bf16 or None, bias (B,Q,h*w) of bias_dtype)"""
    B, C, H, W = mask_features.shape
    Q = mask_embed.shape[1]
    h, w = int(size[0]), int(size[1])
    me = mask_embed.detach().to(torch.bfloat16).contiguous()
    logits = torch.empty((B, Q, H, W), dtype=torch.bfloat16, device=me.device) if want_logits else None
    bias = torch.empty((B, Q, h * w), dtype=bias_dtype, device=me.device)
    check(lib().xm3d_mask_logits_bias(_ptr(me), _ptr(mask_features), B, Q, C, H, W, _ptr(logits), h, w, _ptr(bias),
                                      0 if bias_dtype == torch.float32 else 1, _stream()), "xm3d_mask_logits_bias")
    return logits, bias


def mask_pool(logits, mask_features):
    """MaskPooling with hard masks (sigmoid(logit) > 0.5): (B,Q,H,W) bf16 logits + channels-last bf16 mask_features (B,C,H,W) ->
    (B,Q,C) f32 mean of the feature rows under each mask (0 for an empty mask, like sum / (0 + 1e-8))"""
    B, Q, H, W = logits.shape
    C = mask_features.shape[1]
    if not (logits.is_cuda and logits.dtype == torch.bfloat16 and logits.is_contiguous() and mask_features.dtype == torch.bfloat16
            and mask_features.is_contiguous(memory_format=torch.channels_last) and mask_features.shape == (B, C, H, W)):
        raise TypeError("mask_pool: contiguous bf16 logits (B,Q,H,W) and channels-last bf16 mask_features (B,C,H,W) required")
    chunks = lib().xm3d_mask_pool_chunks(H * W)
    pooled = torch.empty((chunks, B, Q, C), dtype=torch.float32, device=logits.device)
    count = torch.empty((chunks, B, Q), dtype=torch.float32, device=logits.device)
    check(lib().xm3d_mask_pool(_ptr(logits), _ptr(mask_features), B, Q, C, H * W, _ptr(pooled), _ptr(count), _stream()), "xm3d_mask_pool")
    return pooled.sum(0) / (count.sum(0).unsqueeze(-1) + 1e-8)


def attn_mask_bias_supported(shape, size):
    H, W = shape[-2:]
    h, w = int(size[0]), int(size[1])
    return H % h == 0 and W % w == 0 and (H // h) % 2 == 0 and (W // w) % 2 == 0 and h * w <= 8192


def _layer_norm_ok(x, C):
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and x.shape[-1] == C and C % 8 == 0
            and C <= (2048 if x.dtype == torch.float32 else 4096))


def layer_norm_supported(x, C):
    """callers' gate: inference (no autograd graph wanted) on tensors xm3d_layer_norm takes as they are"""
    return not torch.is_grad_enabled() and _layer_norm_ok(x, C)


def layer_norm(x, weight, bias, eps=1e-5, delta=None, want_sum=False):
    """LayerNorm over the last dimension of a contiguous f32/bf16 device tensor (statistics in f32).  delta: tensor like x added
    first (the residual add in front of a pre-norm LayerNorm); want_sum: also return x + delta -> (y, x + delta)."""
    C = x.shape[-1]
    if not _layer_norm_ok(x, C):
        raise TypeError("layer_norm: contiguous f32/bf16 device tensor with C % 8 == 0 (<= 4096 bf16 / 2048 f32) required")
    for t in (weight, bias, delta):
        if t is not None and (t.dtype != x.dtype or not t.is_contiguous()):
            raise TypeError("layer_norm: weight / bias / delta must be contiguous tensors of the input dtype")
    if delta is not None and delta.shape != x.shape:
        raise TypeError("layer_norm: delta must have the shape of x")
    y = torch.empty_like(x)
    s = torch.empty_like(x) if (want_sum and delta is not None) else None
    check(lib().xm3d_layer_norm(_ptr(x), _ptr(delta), 0 if x.dtype == torch.float32 else 1, x.numel() // C, C, _ptr(weight), _ptr(bias),
                                float(eps), _ptr(s), _ptr(y), _stream()), "xm3d_layer_norm")
    return (y, s) if want_sum else y


def add_layer_norm_supported(x, C):
    """inference on an f32 residual stream xm3d_add_layer_norm takes as it is"""
    return (not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == C and C % 4 == 0
            and C <= 1024)


def add_layer_norm(x, delta, weight, bias, eps=1e-5, pos=None, want=("f32", "bf16"), out_dtype=torch.bfloat16):
    """LayerNorm(x + delta) over the last dimension of the contiguous f32 stream x; delta f32 / bf16 like x, or None; weight / bias f32.
    want: which forms to return, in order - "f32" (the new stream), "bf16" (its bf16 rounding), "pos" (bf16(y + pos); pos f32 / bf16
    whose rows repeat along the rows of x: the trailing rows.numel / C rows of x line up with pos).  out_dtype = torch.float32: "pos" comes
    back as f32 (y + pos, the fp32 configuration) and "bf16" is not available."""
    C = x.shape[-1]
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C % 4 == 0 and C <= 1024):
        raise TypeError("add_layer_norm: contiguous f32 device tensor with C % 4 == 0, C <= 1024 required")
    if delta is not None and (delta.shape != x.shape or not delta.is_contiguous() or delta.dtype not in (torch.float32, torch.bfloat16)):
        raise TypeError("add_layer_norm: delta must be a contiguous f32 / bf16 tensor of the shape of x")
    for t in (weight, bias):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != C):
            raise TypeError("add_layer_norm: weight / bias must be contiguous f32 (C,) tensors")
    rows = x.numel() // C
    prow = 0
    if "pos" in want:
        if pos is None or pos.shape[-1] != C or not pos.is_contiguous() or pos.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("add_layer_norm: pos must be a contiguous f32 / bf16 (..., C) tensor")
        prow = pos.numel() // C
        if prow == 0 or rows % prow:
            raise TypeError("add_layer_norm: the rows of pos must tile the rows of x")
    if out_dtype not in (torch.bfloat16, torch.float32) or (out_dtype == torch.float32 and "bf16" in want):
        raise TypeError("add_layer_norm: out_dtype bf16, or f32 without the \"bf16\" output")
    outs = {"f32": torch.empty_like(x) if "f32" in want else None,
            "bf16": torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if "bf16" in want else None,
            "pos": torch.empty(x.shape, dtype=out_dtype, device=x.device) if "pos" in want else None}
    check(lib().xm3d_add_layer_norm(_ptr(x), _ptr(delta), int(delta is not None and delta.dtype == torch.bfloat16), rows, C, _ptr(weight), _ptr(bias),
                                    float(eps), _ptr(pos) if "pos" in want else None, int(pos is not None and pos.dtype == torch.bfloat16), prow,
                                    _ptr(outs["f32"]), _ptr(outs["bf16"]), _ptr(outs["pos"]), int(out_dtype == torch.bfloat16), _stream()),
          "xm3d_add_layer_norm")
    res = tuple(outs[k] for k in want)
    return res[0] if len(res) == 1 else res


def clip_mask_blocked(logits, size, patch):
    """(B, Q, h, w) f32 contiguous mask logits -> (B, Q, (size/patch)^2) bool: max-pooled sigmoid of the bilinear resize to (size, size) < 0.5
    (xm3d_clip_mask_blocked: mask-CLIP's patch mask without the resized intermediate)"""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4 and logits.is_contiguous() and size % patch == 0):
        raise TypeError("clip_mask_blocked: contiguous (B, Q, h, w) f32 device tensor and size % patch == 0 required")
    B, Q, h, w = logits.shape
    n = (size // patch) ** 2
    out = torch.empty((B, Q, n), dtype=torch.uint8, device=logits.device)
    check(lib().xm3d_clip_mask_blocked(_ptr(logits), B * Q, h, w, int(size), int(patch), _ptr(out), _stream()), "xm3d_clip_mask_blocked")
    return out.view(torch.bool)


def pad_bottom_right_nhwc(x, pad_bottom, pad_right):
    """channels-last (B,C,H,W) f32/bf16 device tensor -> (B,C,H+pad_bottom,W+pad_right) channels-last, zero padded, one pass"""
    if x.dtype not in (torch.float32, torch.bfloat16) or not is_nhwc(x):
        raise TypeError("pad_bottom_right_nhwc: channels-last f32/bf16 device tensor required")
    B, C, H, W = x.shape
    out = torch.empty((B, C, H + pad_bottom, W + pad_right), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(lib().xm3d_pad_nhwc(_ptr(x), 0 if x.dtype == torch.float32 else 1, B, H, W, C, int(pad_bottom), int(pad_right), _ptr(out), _stream()),
          "xm3d_pad_nhwc")
    return out


def quick_gelu(x):
    """x * sigmoid(1.702 x) in one pass (f32 / bf16 contiguous device tensor, numel % 8 == 0)"""
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda or not x.is_contiguous():
        raise TypeError("quick_gelu: contiguous f32/bf16 device tensor required")
    out = torch.empty_like(x)
    check(lib().xm3d_quick_gelu(_ptr(x), 0 if x.dtype == torch.float32 else 1, x.numel(), _ptr(out), _stream()), "xm3d_quick_gelu")
    return out


def softmax_rows(scores, scale):
    """scores (..., cols) f32 contiguous -> bf16 softmax(scale * scores) over the last dimension (cols % 4 == 0, <= 8192)"""
    _req(scores, torch.float32, "scores")
    out = torch.empty(scores.shape, dtype=torch.bfloat16, device=scores.device)
    cols = scores.shape[-1]
    check(lib().xm3d_softmax_rows_f32_bf16(_ptr(scores), scores.numel() // cols, cols, float(scale), _ptr(out), _stream()),
          "xm3d_softmax_rows_f32_bf16")
    return out


def geglu(x):
    """x (..., 2D) contiguous f32/bf16 device tensor -> (..., D) = x[..., :D] * gelu(x[..., D:])"""
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous() or x.shape[-1] % 2:
        raise TypeError("geglu: contiguous f32/bf16 device tensor with an even last dimension required")
    D = x.shape[-1] // 2
    out = torch.empty((*x.shape[:-1], D), dtype=x.dtype, device=x.device)
    check(lib().xm3d_geglu(_ptr(x), 0 if x.dtype == torch.float32 else 1, x.numel() // (2 * D), D, _ptr(out), _stream()), "xm3d_geglu")
    return out


# ---------------------------------------------------------------- point -> pixel mapping
def compute_mapping(points, camera_to_world, intrinsic, image_dim=(320, 240), cut_bound=10, depth=None, vis_thres=0.25):
    """points (n,3) f64 device; camera_to_world / intrinsic 4x4 host arrays; depth None or (H,W) f64 device map
    -> (n,3) int32 [row, col, visible] like PointCloudToImageMapper.compute_mapping (fusion_util.py:46-142)."""
    _req(points, torch.float64, "points", 2)
    if points.shape[1] != 3:
        raise RuntimeError("compute_mapping: points must be (n, 3)")
    w2c = np.ascontiguousarray(np.linalg.inv(np.asarray(camera_to_world, dtype=np.float64).reshape(4, 4)))
    k4 = np.ascontiguousarray(np.asarray(intrinsic, dtype=np.float64).reshape(4, 4))
    dh = dw = 0
    if depth is not None:
        _req(depth, torch.float64, "depth", 2)
        dh, dw = depth.shape
    out = torch.empty((points.shape[0], 3), dtype=torch.int32, device=points.device)
    check(lib().xm3d_compute_mapping(_ptr(points), points.shape[0], w2c.ctypes.data_as(ctypes.c_void_p), k4.ctypes.data_as(ctypes.c_void_p),
                                     int(image_dim[0]), int(image_dim[1]), int(cut_bound), _ptr(depth), dh, dw, float(vis_thres), _ptr(out),
                                     _stream()), "xm3d_compute_mapping")
    return out


# ---------------------------------------------------------------- batched assignment
def linear_sum_assignment(cost, n_targets):
    """cost (M, Q, Tmax) f32 device, n_targets (M,) i32 device -> (query idx, target idx) (M, Tmax) i64, pairs sorted by query
    index in the first n_targets[m] slots, -1 elsewhere.  One launch, no host synchronisation."""
    _req(cost, torch.float32, "cost", 3)
    _req(n_targets, torch.int32, "n_targets", 1)
    M, Q, Tm = cost.shape
    if n_targets.numel() != M:
        raise RuntimeError("linear_sum_assignment: one target count per matrix expected")
    oq = torch.full((M, Tm), -1, dtype=torch.int64, device=cost.device)
    ot = torch.full((M, Tm), -1, dtype=torch.int64, device=cost.device)
    check(lib().xm3d_linear_sum_assignment(_ptr(cost), M, Q, Tm, _ptr(n_targets), _ptr(oq), _ptr(ot), _stream()),
          "xm3d_linear_sum_assignment")
    return oq, ot


# ---------------------------------------------------------------- fused softmax attention
_ATTENTION_OFF = _os.environ.get("XM3D_ATTENTION", "hip") == "library"  # A/B switch for measurements: library SDPA everywhere


def attention_supported(q, k, v, bias=None):
    """True when xm3d_attention_fwd takes these tensors as they are (bf16 device tensors (B,N,H,D), channels contiguous, 16-byte
    rows, D a multiple of 8 up to 160, no gradient wanted); callers keep the library path for everything else"""
    if _ATTENTION_OFF or (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)):
        return False
    for t in (q, k, v):
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 4 or t.stride(3) != 1 or t.data_ptr() % 16:
            return False
        if any(st % 8 for st in t.stride()[:3]):
            return False
    D = q.shape[3]
    if D % 8 or D > 160 or k.shape[3] != D or v.shape[3] != D or k.shape[1] != v.shape[1] or k.shape[1] < 1:
        return False
    if bias is not None and (not bias.is_cuda or bias.dtype not in (torch.float32, torch.bfloat16) or bias.dim() != 4 or bias.stride(3) != 1):
        return False
    return True


def attention(q, k, v, bias=None, scale=None, out=None):
    """softmax(q k^T * scale + bias) v per (batch, head).  q (B,Nq,H,D), k/v (B,Nk,H,D) bf16 views with contiguous channels
    (any batch / row / head strides); bias None or additive f32/bf16 (B|1, H|1, Nq, Nk) (broadcast over size-1 dims, -inf
    masks).  -> out (B,Nq,H,D) bf16 (a fresh contiguous tensor, or the given view)."""
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    if out is None:
        out = torch.empty((B, Nq, H, D), dtype=torch.bfloat16, device=q.device)
    if scale is None:
        scale = D ** -0.5

    def st(t):
        return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))

    bdt, bst = 0, None
    if bias is not None:
        if bias.shape[-2:] != (Nq, Nk) or bias.shape[0] not in (1, B) or bias.shape[1] not in (1, H):
            raise RuntimeError(f"attention: bias shape {tuple(bias.shape)} does not broadcast to ({B},{H},{Nq},{Nk})")
        bdt = 1 if bias.dtype == torch.float32 else 2
        bst = (ctypes.c_int64 * 3)(bias.stride(0) if bias.shape[0] > 1 else 0, bias.stride(1) if bias.shape[1] > 1 else 0, bias.stride(2))
    check(lib().xm3d_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, Nq, Nk, D, st(q), st(k), st(v), st(out), _ptr(bias), bdt,
                                   bst, float(scale), _stream()), "xm3d_attention_fwd")
    return out


def attention_f32_supported(q, k, v, bias=None):
    """True when xm3d_attention_fwd_f32 takes these tensors as they are: f32 device tensors (B,N,H,D), channels contiguous, 16-byte
    rows, D a multiple of 8 up to 64, no gradient wanted, bias None or additive f32 (B|1, H|1, Nq, Nk)"""
    if _ATTENTION_OFF or _os.environ.get("XM3D_ATTENTION_F32", "hip") == "library" or \
            (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)):
        return False
    for t in (q, k, v):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or t.stride(3) != 1 or t.data_ptr() % 16:
            return False
        if any(st % 4 for st in t.stride()[:3]):
            return False
    D = q.shape[3]
    if D % 8 or D > 64 or k.shape[3] != D or v.shape[3] != D or k.shape[1] != v.shape[1] or k.shape[1] < 1:
        return False
    if bias is not None and (not bias.is_cuda or bias.dtype != torch.float32 or bias.dim() != 4 or bias.stride(3) != 1):
        return False
    return True


def attention_f32(q, k, v, bias=None, scale=None, out=None):
    """softmax(q k^T * scale + bias) v per (batch, head) to f32 accuracy on the matrix cores (csrc/attention_f32.hip: operands split
    in IEEE halves, f32 softmax).  q (B,Nq,H,D), k/v (B,Nk,H,D) f32 views with contiguous channels; bias None or additive f32
    (B|1, H|1, Nq, Nk), -inf masks.  -> out (B,Nq,H,D) f32 (fresh contiguous, or the given view)."""
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    if out is None:
        out = torch.empty((B, Nq, H, D), dtype=torch.float32, device=q.device)
    if scale is None:
        scale = D ** -0.5

    def st(t):
        return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))

    bst = None
    if bias is not None:
        if bias.shape[-2:] != (Nq, Nk) or bias.shape[0] not in (1, B) or bias.shape[1] not in (1, H):
            raise RuntimeError(f"attention_f32: bias shape {tuple(bias.shape)} does not broadcast to ({B},{H},{Nq},{Nk})")
        bst = (ctypes.c_int64 * 3)(bias.stride(0) if bias.shape[0] > 1 else 0, bias.stride(1) if bias.shape[1] > 1 else 0, bias.stride(2))
    check(lib().xm3d_attention_fwd_f32(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, Nq, Nk, D, st(q), st(k), st(v), st(out), _ptr(bias), bst,
                                       float(scale), _stream()), "xm3d_attention_fwd_f32")
    return out


def _att_ok(q, k, v, bias):
    for t in (q, k, v):
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 4 or t.stride(3) != 1 or t.data_ptr() % 16:
            return False
        if any(st % 8 for st in t.stride()[:3]):
            return False
    D = q.shape[3]
    if D % 8 or D > 160 or k.shape[3] != D or v.shape[3] != D or k.shape[1] != v.shape[1] or k.shape[1] < 1:
        return False
    if bias is not None and (not bias.is_cuda or bias.dtype not in (torch.float32, torch.bfloat16) or bias.dim() != 4 or bias.stride(3) != 1
                             or bias.requires_grad):
        return False
    return True


def attention_train_supported(q, k, v, bias=None):
    """True when the differentiable HIP attention (forward with log-sum-exp + xm3d_attention_bwd) takes these bf16 tensors: the
    training path through the bf16 frozen UNet.  f32 operands keep torch's math backend (exact in f32)."""
    return (not _ATTENTION_OFF and torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
            and _att_ok(q, k, v, bias))


def _st3(t):
    return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def _bias_args(bias, B, H, Nq, Nk):
    if bias is None:
        return 0, None
    if bias.shape[-2:] != (Nq, Nk) or bias.shape[0] not in (1, B) or bias.shape[1] not in (1, H):
        raise RuntimeError(f"attention: bias shape {tuple(bias.shape)} does not broadcast to ({B},{H},{Nq},{Nk})")
    return (1 if bias.dtype == torch.float32 else 2,
            (ctypes.c_int64 * 3)(bias.stride(0) if bias.shape[0] > 1 else 0, bias.stride(1) if bias.shape[1] > 1 else 0, bias.stride(2)))


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, bias, scale):
        B, Nq, H, D = q.shape
        Nk = k.shape[1]
        out = torch.empty((B, Nq, H, D), dtype=torch.bfloat16, device=q.device)
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        bdt, bst = _bias_args(bias, B, H, Nq, Nk)
        check(lib().xm3d_attention_fwd_lse(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, Nq, Nk, D, _st3(q), _st3(k), _st3(v), _st3(out), _ptr(bias),
                                           bdt, bst, float(scale), _ptr(lse), _stream()), "xm3d_attention_fwd_lse")
        ctx.save_for_backward(q, k, v, out, lse, bias)
        ctx.scale = float(scale)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        q, k, v, out, lse, bias = ctx.saved_tensors
        B, Nq, H, D = q.shape
        Nk = k.shape[1]
        dout = dout.to(torch.bfloat16)
        if dout.stride(3) != 1 or any(s % 8 for s in dout.stride()[:3]) or dout.data_ptr() % 16:
            dout = dout.contiguous()
        dq = torch.empty((B, Nq, H, D), dtype=torch.bfloat16, device=q.device)
        dk = torch.empty((B, Nk, H, D), dtype=torch.bfloat16, device=q.device)
        dv = torch.empty((B, Nk, H, D), dtype=torch.bfloat16, device=q.device)
        ws = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        bdt, bst = _bias_args(bias, B, H, Nq, Nk)
        check(lib().xm3d_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse), B, H, Nq, Nk, D, _st3(q), _st3(k), _st3(v),
                                       _st3(out), _st3(dout), _ptr(bias), bdt, bst, ctx.scale, _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws), _stream()),
              "xm3d_attention_bwd")
        return dq, dk, dv, None, None


def attention_train(q, k, v, bias=None, scale=None):
    """differentiable softmax(q k^T * scale + bias) v on the HIP kernels (bf16 (B,N,H,D) views, see attention()); the bias is a constant"""
    return _AttentionFn.apply(q, k, v, bias, q.shape[3] ** -0.5 if scale is None else scale)


# ---------------------------------------------------------------- deformable attention
def _msda_dtype(value):
    if value.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"msda: float32 or float64 tensors required, got {value.dtype}")
    return value.dtype, ("" if value.dtype == torch.float32 else "_f64")


def msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    dt, sfx = _msda_dtype(value)
    _req(value, dt, "value", 4)
    _req(spatial_shapes, torch.int64, "spatial_shapes", 2)
    _req(level_start_index, torch.int64, "level_start_index", 1)
    _req(sampling_loc, dt, "sampling_loc", 6)
    _req(attn_weight, dt, "attn_weight", 5)
    B, S, H, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    out = torch.empty((B, Lq, H * D), dtype=dt, device=value.device)
    check(getattr(lib(), "xm3d_msda_forward" + sfx)(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(sampling_loc),
                                                     _ptr(attn_weight), B, S, H, D, L, Lq, P, _ptr(out), _stream()),
          "xm3d_msda_forward" + sfx)
    return out


def msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output):
    dt, sfx = _msda_dtype(value)
    _req(value, dt, "value", 4)
    _req(spatial_shapes, torch.int64, "spatial_shapes", 2)
    _req(level_start_index, torch.int64, "level_start_index", 1)
    _req(sampling_loc, dt, "sampling_loc", 6)
    _req(attn_weight, dt, "attn_weight", 5)
    _req(grad_output, dt, "grad_output")
    B, S, H, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    gv = torch.zeros_like(value)
    gl = torch.zeros_like(sampling_loc)
    ga = torch.zeros_like(attn_weight)
    check(getattr(lib(), "xm3d_msda_backward" + sfx)(_ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(sampling_loc),
                                                      _ptr(attn_weight), _ptr(grad_output), B, S, H, D, L, Lq, P, _ptr(gv), _ptr(gl),
                                                      _ptr(ga), _stream()), "xm3d_msda_backward" + sfx)
    return gv, gl, ga


# ---------------------------------------------------------------- 2D->3D fusion
def mask_owner(logits, score, keep):
    """logits (B,Q,H,W) f32, score (B,Q) f32, keep (B,Q) bool/u8 -> owner (B,H,W) i32 (query id or -1), see xm3d.h"""
    _req(logits, torch.float32, "logits", 4)
    _req(score, torch.float32, "score", 2)
    keep = keep.to(torch.uint8).contiguous()
    B, Q, H, W = logits.shape
    owner = torch.empty((B, H, W), dtype=torch.int32, device=logits.device)
    check(lib().xm3d_mask_owner(_ptr(logits), _ptr(score), _ptr(keep), B, Q, H * W, _ptr(owner), _stream()), "xm3d_mask_owner")
    return owner


def mask_point_fuse(masks_u8, x_label, y_label, embed):
    """masks (Q,H,W) uint8, x/y (n,) i64, embed (Q,C) f32 -> (feat2d (n,C), count (n,) i32)."""
    _req(masks_u8, torch.uint8, "masks", 3)
    _req(x_label, torch.int64, "x_label", 1)
    _req(y_label, torch.int64, "y_label", 1)
    _req(embed, torch.float32, "embed", 2)
    Q, Hm, Wm = masks_u8.shape
    n = x_label.numel()
    C = embed.shape[1]
    feat = torch.empty((n, C), dtype=torch.float32, device=embed.device)
    cnt = torch.empty(n, dtype=torch.int32, device=embed.device)
    check(lib().xm3d_mask_point_fuse(_ptr(masks_u8), Q, Hm, Wm, _ptr(x_label), _ptr(y_label), n, _ptr(embed), C, _ptr(feat),
                                     _ptr(cnt), _stream()), "xm3d_mask_point_fuse")
    return feat, cnt


# ---------------------------------------------------------------- nearest neighbour
def nearest_index_segmented(pts, desc, max_queries, out):
    """pts (n,3) f32 laid out per segment as [queries | references]; desc (S,4) i64 device {q_off,q_cnt,r_off,r_cnt};
    out (n,) i64 pre-filled by the caller: out[q_off+i] <- r_off + nearest reference of the same segment."""
    _req(pts, torch.float32, "pts", 2)
    _req(desc, torch.int64, "desc", 2)
    _req(out, torch.int64, "out", 1)
    if pts.shape[1] != 3 or desc.shape[1] != 4 or out.numel() != pts.shape[0]:
        raise RuntimeError("nearest_index_segmented: pts (n,3), desc (S,4), out (n,) expected")
    check(lib().xm3d_nearest_index_segmented(_ptr(pts), _ptr(desc), desc.shape[0], int(max_queries), _ptr(out), _stream()),
          "xm3d_nearest_index_segmented")
    return out


def scene_votes(rows, pred, n_rows, n_cls):
    """rows (Np,) i64 table row per visible point, pred (K, Np) i64 class ids -> (label (K, n_rows) i64 = first maximal class of
    the vote table, seen (n_rows,) bool = row got a vote).  Two launches, no host synchronisation."""
    _req(rows, torch.int64, "rows", 1)
    _req(pred, torch.int64, "pred", 2)
    if pred.shape[1] != rows.shape[0]:
        raise RuntimeError("scene_votes: pred (K, Np) and rows (Np,) expected")
    K = pred.shape[0]
    votes = torch.empty((K, n_rows, n_cls), dtype=torch.int32, device=rows.device)
    label = torch.empty((K, n_rows), dtype=torch.int64, device=rows.device)
    seen = torch.empty(n_rows, dtype=torch.uint8, device=rows.device)
    check(lib().xm3d_scene_votes(_ptr(rows), _ptr(pred), K, rows.shape[0], int(n_rows), int(n_cls), _ptr(votes), _ptr(label), _ptr(seen),
                                 _stream()), "xm3d_scene_votes")
    return label, seen.view(torch.bool)


def nearest_valid_fill(xyz, valid, cell=0.1, method="octree"):
    """xyz (n,3) f32, valid (n,) bool/uint8 -> (n,) i64: own index where valid, else the index of the nearest valid point
    (exact, lowest index on ties, identity when nothing is valid).  No host synchronisation.
    method "octree": Morton-ordered cells + depth-first descent per query (best for small holes);
    method "sorted": Morton-sorted queries and references, tile-pruned LDS scan (best for many far queries)."""
    _req(xyz, torch.float32, "xyz", 2)
    if valid.dtype == torch.bool:
        valid = valid.view(torch.uint8)
    _req(valid, torch.uint8, "valid", 1)
    if xyz.shape[1] != 3 or valid.numel() != xyz.shape[0]:
        raise RuntimeError("nearest_valid_fill: xyz (n,3) and valid (n,) expected")
    n = xyz.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=xyz.device)
    if method == "sorted":
        ws = torch.empty(lib().xm3d_nearest_valid_fill_sorted_workspace_bytes(n), dtype=torch.uint8, device=xyz.device)
        check(lib().xm3d_nearest_valid_fill_sorted(_ptr(xyz), n, _ptr(valid), _ptr(out), _ptr(ws), _stream()), "xm3d_nearest_valid_fill_sorted")
        return out
    if method != "octree":
        raise ValueError(f"nearest_valid_fill: unknown method {method!r}")
    ws = torch.empty(lib().xm3d_nearest_valid_fill_workspace_bytes(n), dtype=torch.uint8, device=xyz.device)
    check(lib().xm3d_nearest_valid_fill(_ptr(xyz), n, _ptr(valid), float(cell), _ptr(out), _ptr(ws), _stream()), "xm3d_nearest_valid_fill")
    return out


def nearest_index(query, ref, ref_valid=None, counts=None):
    """(n,3) f32, (m,3) f32 -> (n,) i64 index of the nearest reference point (exact, lowest index on ties).
    ref_valid (m,) uint8/bool: only reference points with a non-zero flag are considered.
    counts (2,) i64 device tensor {live queries, live refs}: only those leading rows take part (others: out = 0)."""
    _req(query, torch.float32, "query", 2)
    _req(ref, torch.float32, "ref", 2)
    if ref_valid is not None:
        if ref_valid.dtype == torch.bool:
            ref_valid = ref_valid.to(torch.uint8)
        _req(ref_valid, torch.uint8, "ref_valid", 1)
        assert ref_valid.numel() == ref.shape[0]
    if query.shape[1] != 3 or ref.shape[1] != 3:
        raise RuntimeError("nearest_index works on 3-D points")
    if counts is not None:
        _req(counts, torch.int64, "counts", 1)
        assert counts.numel() == 2
        out = torch.zeros(query.shape[0], dtype=torch.int64, device=query.device)
    else:
        out = torch.empty(query.shape[0], dtype=torch.int64, device=query.device)
    ws = None
    if query.shape[0] < 2048 * 256 and ref.shape[0] >= 8 * 1024:  # few query slabs, many reference tiles: reference slices
        ws = torch.empty(query.shape[0], dtype=torch.int64, device=query.device)
    check(lib().xm3d_nearest_index(_ptr(query), query.shape[0], _ptr(ref), ref.shape[0], _ptr(ref_valid), _ptr(counts), _ptr(out),
                                   _ptr(ws), _stream()), "xm3d_nearest_index")
    return out
