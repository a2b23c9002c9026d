"""GPU voxeliser with the reference's ``Voxelizer`` interface.

Mirror of /root/reference/dataset/voxelizer.py:11-132 (constructor arguments,
``get_transformation_matrix``, ``voxelize`` return tuple) for the configuration
the loaders use (dataset/point_loader.py:100-107: clip_bound=None,
use_augmentation=True).  The random augmentation is drawn on the host from the
global ``np.random`` stream in the reference's order (three angles, one list
shuffle, one scale) so that the same seed gives the same 4x4 matrix; the
quantisation itself (transform, floor, min shift, FNV keys, unique) runs in
``xm3d_voxelize`` on the device and is bit-exact with the reference.
"""
from __future__ import annotations

import collections.abc

import numpy as np
import torch

from . import ops


def _axis_rotation(axis_ind, theta):
    from scipy.linalg import expm

    axis = np.zeros(3)
    axis[axis_ind] = 1
    return expm(np.cross(np.eye(3), axis / np.linalg.norm(axis) * theta))


class Voxelizer:
    def __init__(self, voxel_size=1, clip_bound=None, use_augmentation=False, scale_augmentation_bound=None,
                 rotation_augmentation_bound=None, translation_augmentation_ratio_bound=None, ignore_label=255,
                 device="cuda"):
        if clip_bound is not None:
            raise NotImplementedError("clip_bound is never set by the XMask3D loaders (dataset/point_loader.py:102)")
        self.voxel_size = voxel_size
        self.clip_bound = clip_bound
        self.ignore_label = ignore_label
        self.use_augmentation = use_augmentation
        self.scale_augmentation_bound = scale_augmentation_bound
        self.rotation_augmentation_bound = rotation_augmentation_bound
        self.translation_augmentation_ratio_bound = translation_augmentation_ratio_bound
        self.device = device

    def get_transformation_matrix(self):
        voxelization_matrix, rotation_matrix = np.eye(4), np.eye(4)
        rot = np.eye(3)
        if self.use_augmentation and self.rotation_augmentation_bound is not None:
            if not isinstance(self.rotation_augmentation_bound, collections.abc.Iterable):
                raise ValueError()
            mats = []
            for axis_ind, bound in enumerate(self.rotation_augmentation_bound):
                theta = 0
                if bound is not None:
                    theta = np.random.uniform(*bound)
                mats.append(_axis_rotation(axis_ind, theta))
            np.random.shuffle(mats)
            rot = mats[0] @ mats[1] @ mats[2]
        rotation_matrix[:3, :3] = rot
        scale = 1 / self.voxel_size
        if self.use_augmentation and self.scale_augmentation_bound is not None:
            scale *= np.random.uniform(*self.scale_augmentation_bound)
        np.fill_diagonal(voxelization_matrix[:3, :3], scale)
        return voxelization_matrix, rotation_matrix

    def rigid_matrix(self):
        M_v, M_r = self.get_transformation_matrix()
        return (M_r @ M_v) if self.use_augmentation else M_v, M_r

    def voxelize_device(self, coords, matrix=None):
        """coords (n,3) f64 numpy or device tensor -> (grid i32 (Nv,3), inds i64 (Nv,), inverse i64 (n,)) on device."""
        if matrix is None:
            matrix, _ = self.rigid_matrix()
        if isinstance(coords, np.ndarray):
            coords = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.float64)).to(self.device)
        if coords.shape[0] == 0 or coords.shape[1] != 3:
            raise AssertionError("voxelize needs a non-empty (n,3) array")
        return ops.voxelize(coords.contiguous(), matrix)

    def voxelize(self, coords, feats, labels, center=None, link=None, return_ind=False):
        assert coords.shape[1] == 3 and coords.shape[0] == feats.shape[0] and coords.shape[0]
        matrix, M_r = self.rigid_matrix()
        grid, inds, inverse = self.voxelize_device(coords, matrix)
        inds_h = inds.cpu().numpy()
        coords_aug = grid.cpu().numpy().astype(np.float64)
        feats, labels = feats[inds_h], labels[inds_h]
        if feats.shape[1] > 6:
            feats[:, 3:6] = feats[:, 3:6] @ (M_r[:3, :3].T)
        inv_h = inverse.cpu().numpy()
        if return_ind:
            return coords_aug, feats, labels, inv_h, inds_h
        if link is not None:
            return coords_aug, feats, labels, inv_h, link[inds_h]
        return coords_aug, feats, labels, inv_h
