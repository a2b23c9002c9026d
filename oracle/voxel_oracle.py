"""CPU oracle for the voxelisation stage.

TEST INFRASTRUCTURE ONLY.  Nothing under ``xmask3d_amd/`` may import this
module; it is the checker for ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

This is a numpy restatement (not a copy) of the reference algorithm:

* key folding        -> /root/reference/dataset/voxelization_utils.py:6-18
* unique / inverse   -> /root/reference/dataset/voxelization_utils.py:93-102
* rigid transform,
  floor, min shift   -> /root/reference/dataset/voxelizer.py:104-122
* augmentation draw  -> /root/reference/dataset/voxelizer.py:32-58
* coord packing      -> /root/reference/dataset/data_loader.py:262-265,319-357

Parity is PINNED: ``tests/golden/voxel_*.npz`` were produced by importing the
reference modules in the build container (``tests/golden/make_golden.py``) and
``tests/test_oracle_voxel.py`` checks this file against them bit for bit.
"""
from __future__ import annotations

import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)  # 0xcbf29ce484222325
FNV_PRIME = np.uint64(1099511628211)  # 0x100000001b3


def fnv_keys(grid: np.ndarray) -> np.ndarray:
    """uint64 key per row of an integer-valued (N, D) array.

    h = OFFSET; for each column: h = (h * PRIME) mod 2^64; h ^= uint64(v).
    Note the order (multiply, then xor) - it is *not* textbook FNV-1a.
    """
    g = np.ascontiguousarray(grid).astype(np.uint64)
    h = np.full(g.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(g.shape[1]):
            h = h * FNV_PRIME
            h = h ^ g[:, j]
    return h


def unique_first(keys: np.ndarray):
    """(inds, inverse): ascending-key order, first occurrence, rank of each key.

    Written with an explicit stable argsort so the tie-breaking rule the GPU
    path must reproduce (smallest original index wins) is visible.
    """
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    head = np.ones(sk.shape[0], dtype=bool)
    head[1:] = sk[1:] != sk[:-1]
    inds = order[head]
    rank_sorted = np.cumsum(head) - 1
    inverse = np.empty(keys.shape[0], dtype=np.int64)
    inverse[order] = rank_sorted
    return inds.astype(np.int64), inverse


def axis_rotation(axis_ind: int, theta: float) -> np.ndarray:
    """Rotation by ``theta`` about a coordinate axis via the matrix exponential
    of the cross-product matrix (voxelizer.py:7-8 uses scipy expm the same way)."""
    from scipy.linalg import expm

    axis = np.zeros(3)
    axis[axis_ind] = 1.0
    return expm(np.cross(np.eye(3), axis / np.linalg.norm(axis) * theta))


def draw_augmentation(
    voxel_size: float,
    rotation_bound=((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi)),
    scale_bound=(0.9, 1.1),
):
    """Consume the global ``np.random`` stream exactly as the reference does and
    return (M_v, M_r): three uniform angles, one list shuffle, one uniform scale."""
    mats = []
    for axis_ind, bound in enumerate(rotation_bound):
        theta = 0
        if bound is not None:
            theta = np.random.uniform(*bound)
        mats.append(axis_rotation(axis_ind, theta))
    np.random.shuffle(mats)
    M_r = np.eye(4)
    M_r[:3, :3] = mats[0] @ mats[1] @ mats[2]
    scale = 1 / voxel_size
    if scale_bound is not None:
        scale *= np.random.uniform(*scale_bound)
    M_v = np.eye(4)
    np.fill_diagonal(M_v[:3, :3], scale)
    return M_v, M_r


def voxelize_with_matrix(xyz: np.ndarray, T: np.ndarray):
    """Quantise points given the 4x4 rigid transform ``T = M_r @ M_v``.

    Returns (grid (Nv,3) float64 integer-valued, inds (Nv,), inverse (Np,)).
    """
    homo = np.hstack((xyz, np.ones((xyz.shape[0], 1), dtype=xyz.dtype)))
    grid = np.floor(homo @ T.T[:, :3])
    grid = np.floor(grid - grid.min(0))
    inds, inverse = unique_first(fnv_keys(grid))
    return grid[inds], inds, inverse


def voxelize(xyz, feats, labels, voxel_size=0.02, matrix=None):
    """Mirror of ``Voxelizer.voxelize`` for the configuration every loader uses
    (clip_bound None, augmentation on).  ``matrix`` (4x4) bypasses the RNG."""
    if matrix is None:
        M_v, M_r = draw_augmentation(voxel_size)
        matrix = M_r @ M_v
    grid, inds, inverse = voxelize_with_matrix(xyz, matrix)
    return grid, feats[inds], labels[inds], inverse, inds, matrix


def pack_batch(grids, feats_list, inverses):
    """Collate: int32 [b, x, y, z] rows, rgb/127.5-1 features, offset inverses."""
    coords, feats, invs = [], [], []
    base = 0
    for b, (g, f, inv) in enumerate(zip(grids, feats_list, inverses)):
        c = np.empty((g.shape[0], 4), dtype=np.int32)
        c[:, 0] = b
        c[:, 1:] = g.astype(np.int32)
        coords.append(c)
        feats.append((f / 127.5 - 1.0).astype(np.float32))
        invs.append(inv + base)
        base += g.shape[0]
    return np.concatenate(coords), np.concatenate(feats), np.concatenate(invs)
