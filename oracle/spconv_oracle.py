"""CPU oracle for the sparse 3D backbone.

TEST INFRASTRUCTURE ONLY - never imported by ``xmask3d_amd``.

PARITY UNPINNED at the operator level: the reference reaches these ops through
MinkowskiEngine (un-vendored, unpinned git master ~v0.5.4,
/root/reference/installation.md:41) whose source is not under /root/reference
and which cannot be installed offline.  The reference holds no test, fixture or
golden vector at this boundary.  What is restated here:

* topology   - /root/reference/models/modeling/meta_arch/mink_unet.py:44-178,
               resnet_base.py:64-96, pc_processor.py:29-34,55-60
* semantics  - MinkowskiEngine 0.5 published behaviour:
    - stride-s output coordinates  = unique(floor(c / (s*ts)) * (s*ts)) per batch
    - odd kernel offsets centred {-(k//2)..k//2}*ts ; even kernel {0..k-1}*ts
    - offset index enumerates with the FIRST spatial axis fastest
    - transposed conv writes onto the already existing coordinate set of the
      target stride using the swapped (in<->out) map of the strided conv
    - out[o] += in[i] @ kernel[k]   with kernel (K, Cin, Cout), no bias
    - MinkowskiBatchNorm = BatchNorm1d(eps 1e-5, momentum 0.1) over rows
    - a SparseTensor built from unique coordinates keeps the input row order

Self-consistency pins available (tests/test_oracle_spconv.py): rulebook as a set
== brute-force neighbour search; odd-k stride-1 conv == torch conv3d on a
densified grid.  Row order at strides > 1 is a free choice (nothing downstream
depends on it): ascending (b, x, y, z).
"""
from __future__ import annotations

import numpy as np
import torch

_BIAS = 1 << 15


def pack_keys(c: np.ndarray) -> np.ndarray:
    """(N,4) int [b,x,y,z] -> uint64 sortable key, lexicographic in (b,x,y,z)."""
    c = c.astype(np.int64)
    return ((c[:, 0] << 48) | ((c[:, 1] + _BIAS) << 32) | ((c[:, 2] + _BIAS) << 16) | (c[:, 3] + _BIAS)).astype(np.uint64)


def unpack_keys(k: np.ndarray) -> np.ndarray:
    k = k.astype(np.int64)
    out = np.empty((k.shape[0], 4), dtype=np.int32)
    out[:, 0] = k >> 48
    out[:, 1] = ((k >> 32) & 0xFFFF) - _BIAS
    out[:, 2] = ((k >> 16) & 0xFFFF) - _BIAS
    out[:, 3] = (k & 0xFFFF) - _BIAS
    return out


def stride_coords(coords: np.ndarray, ts_out: int) -> np.ndarray:
    c = coords.copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], ts_out) * ts_out
    return unpack_keys(np.unique(pack_keys(c)))


def kernel_offsets(k: int, ts: int) -> np.ndarray:
    """(k^3, 3) spatial offsets, x fastest."""
    r = np.arange(k) - (k // 2 if k % 2 == 1 else 0)
    zz, yy, xx = np.meshgrid(r, r, r, indexing="ij")
    return np.stack([xx.ravel(), yy.ravel(), zz.ravel()], 1).astype(np.int32) * ts


def _lookup(table_coords: np.ndarray, query: np.ndarray) -> np.ndarray:
    tk = pack_keys(table_coords)
    order = np.argsort(tk, kind="stable")
    stk = tk[order]
    qk = pack_keys(query)
    pos = np.searchsorted(stk, qk)
    pos_c = np.minimum(pos, len(stk) - 1)
    hit = stk[pos_c] == qk
    return np.where(hit, order[pos_c], -1).astype(np.int32)


def kernel_map(in_coords: np.ndarray, out_coords: np.ndarray, k: int, ts_in: int) -> np.ndarray:
    """nbr (K, N_out): row of in_coords at out + offset_k, or -1.
    Serves stride-1 convs (out set == in set) and k=2 stride-2 down convs."""
    offs = kernel_offsets(k, ts_in)
    nbr = np.empty((offs.shape[0], out_coords.shape[0]), dtype=np.int32)
    for i, d in enumerate(offs):
        q = out_coords.copy()
        q[:, 1:] += d
        nbr[i] = _lookup(in_coords, q)
    return nbr


def kernel_map_transposed(coarse: np.ndarray, fine: np.ndarray, k: int, ts_fine: int) -> np.ndarray:
    """nbr (K, N_fine) for the k=2,s=2 transposed conv coarse->fine: the swap of
    kernel_map(fine, coarse): fine row o = coarse row p + offset_k  <=>  nbr[k, o] = p."""
    offs = kernel_offsets(k, ts_fine)
    nbr = np.full((offs.shape[0], fine.shape[0]), -1, dtype=np.int32)
    for i, d in enumerate(offs):
        q = fine.copy()
        q[:, 1:] -= d
        nbr[i] = _lookup(coarse, q)
    return nbr


def spconv(feats: torch.Tensor, kernel: torch.Tensor, nbr: np.ndarray) -> torch.Tensor:
    """Per-offset gather -> matmul -> scatter-add (the algorithm of ME's CPU path)."""
    if kernel.dim() == 2:
        kernel = kernel[None]
    K, _, cout = kernel.shape
    assert K == nbr.shape[0]
    out = torch.zeros(nbr.shape[1], cout, dtype=feats.dtype)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        if o.size == 0:
            continue
        i = torch.from_numpy(nbr[k][o].astype(np.int64))
        out.index_add_(0, torch.from_numpy(o), feats[i] @ kernel[k])
    return out


def batchnorm(x, p, prefix, training=False, eps=1e-5):
    w, b = p[prefix + ".bn.weight"], p[prefix + ".bn.bias"]
    if training:
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
    else:
        mean, var = p[prefix + ".bn.running_mean"], p[prefix + ".bn.running_var"]
    return (x - mean) / torch.sqrt(var + eps) * w + b


class CoordCache:
    """Per-forward cache of coordinate sets and kernel maps (ME coordinate manager)."""

    def __init__(self, coords: np.ndarray):
        self.coords = {1: np.asarray(coords, dtype=np.int32)}
        self.maps = {}

    def level(self, ts):
        if ts not in self.coords:
            self.coords[ts] = stride_coords(self.level(ts // 2), ts)
        return self.coords[ts]

    def map(self, ts_in, ts_out, k, transposed=False):
        key = (ts_in, ts_out, k, transposed)
        if key not in self.maps:
            if transposed:
                self.maps[key] = kernel_map_transposed(self.level(ts_in), self.level(ts_out), k, ts_out)
            elif ts_in == ts_out:
                if k == 1:
                    self.maps[key] = np.arange(self.level(ts_in).shape[0], dtype=np.int32)[None]
                else:
                    self.maps[key] = kernel_map(self.level(ts_in), self.level(ts_in), k, ts_in)
            else:
                self.maps[key] = kernel_map(self.level(ts_in), self.level(ts_out), k, ts_in)
        return self.maps[key]


ARCH = {
    "MinkUNet14A": ((1, 1, 1, 1, 1, 1, 1, 1), (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet18A": ((2, 2, 2, 2, 2, 2, 2, 2), (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet34C": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 96, 96)),
}


def minkunet_forward(params: dict, coords: np.ndarray, feats: torch.Tensor, arch="MinkUNet34C", training=False,
                     cache: CoordCache | None = None):
    """Returns (bottleneck_feats (N16,256), bottleneck_coords, out_feats (N1,Cout))."""
    layers, _planes = ARCH[arch]
    cm = cache or CoordCache(coords)
    relu = torch.relu

    def conv(x, name, ts_in, ts_out, k, transposed=False):
        return spconv(x, params[name + ".kernel"], cm.map(ts_in, ts_out, k, transposed))

    def bn(x, name):
        return batchnorm(x, params, name, training)

    def block(x, name, ts):
        res = x
        out = relu(bn(conv(x, name + ".conv1", ts, ts, 3), name + ".norm1"))
        out = bn(conv(out, name + ".conv2", ts, ts, 3), name + ".norm2")
        if (name + ".downsample.0.kernel") in params:
            res = bn(conv(x, name + ".downsample.0", ts, ts, 1), name + ".downsample.1")
        return relu(out + res)

    def stage(x, name, n, ts):
        for i in range(n):
            x = block(x, f"{name}.{i}", ts)
        return x

    out_p1 = relu(bn(conv(feats, "conv0p1s1", 1, 1, 5), "bn0"))
    out = relu(bn(conv(out_p1, "conv1p1s2", 1, 2, 2), "bn1"))
    b1 = stage(out, "block1", layers[0], 2)
    out = relu(bn(conv(b1, "conv2p2s2", 2, 4, 2), "bn2"))
    b2 = stage(out, "block2", layers[1], 4)
    out = relu(bn(conv(b2, "conv3p4s2", 4, 8, 2), "bn3"))
    b3 = stage(out, "block3", layers[2], 8)
    out = relu(bn(conv(b3, "conv4p8s2", 8, 16, 2), "bn4"))
    bott = stage(out, "block4", layers[3], 16)

    out = relu(bn(conv(bott, "convtr4p16s2", 16, 8, 2, True), "bntr4"))
    out = stage(torch.cat([out, b3], 1), "block5", layers[4], 8)
    out = relu(bn(conv(out, "convtr5p8s2", 8, 4, 2, True), "bntr5"))
    out = stage(torch.cat([out, b2], 1), "block6", layers[5], 4)
    out = relu(bn(conv(out, "convtr6p4s2", 4, 2, 2, True), "bntr6"))
    out = stage(torch.cat([out, b1], 1), "block7", layers[6], 2)
    out = relu(bn(conv(out, "convtr7p2s2", 2, 1, 2, True), "bntr7"))
    out = stage(torch.cat([out, out_p1], 1), "block8", layers[7], 1)
    out = conv(out, "final", 1, 1, 1)
    return bott, cm.level(16), out


def pc_processor_forward(params, coords, feats, arch="MinkUNet34C", cache=None):
    """pc_processor.py:29-34 -> (implicit_x (N16,768), x (N1,768), idx (N16,))."""
    enc = {k[len("encoder."):]: v for k, v in params.items() if k.startswith("encoder.")}
    bott, c16, out = minkunet_forward(enc, coords, feats, arch, cache=cache)
    imp = bott @ params["point2text_adapter.weight"].T + params["point2text_adapter.bias"]
    x = out @ params["decoder.weight"].T + params["decoder.bias"]
    return imp, x, torch.from_numpy(c16[:, 0].astype(np.int64))


def pc_binary_forward(params, coords, feats, arch="MinkUNet18A", cache=None, eps=1e-5):
    """pc_processor.py:55-60 -> (N1,1) logits (eval-mode BatchNorm1d)."""
    enc = {k[len("encoder."):]: v for k, v in params.items() if k.startswith("encoder.")}
    _, _, out = minkunet_forward(enc, coords, feats, arch, cache=cache)
    x = (out - params["batch_norm.running_mean"]) / torch.sqrt(params["batch_norm.running_var"] + eps)
    x = torch.relu(x * params["batch_norm.weight"] + params["batch_norm.bias"])
    return x @ params["fc.weight"].T + params["fc.bias"]
