"""CPU: host-side mirrors of the reference interfaces (no GPU)."""
import os

import numpy as np

from xmask3d_amd import synthetic as syn
from xmask3d_amd.voxelizer import Voxelizer


def test_projection_matches_reference_mapping(golden_dir):
    g = np.load(os.path.join(golden_dir, "mapping.npz"))
    assert np.abs(syn.scannet_intrinsics() - g["intrinsic"]).max() == 0
    assert (syn.project_points(g["pose"], g["pts"], None) == g["map_nodepth"]).all()
    assert (syn.project_points(g["pose"], g["pts"], g["depth"]) == g["map_depth"]).all()


def test_voxelizer_draws_reference_matrix(golden_dir):
    s = np.load(os.path.join(golden_dir, "voxel_scene_a.npz"))
    vox = Voxelizer(voxel_size=0.02, use_augmentation=True, scale_augmentation_bound=(0.9, 1.1),
                    rotation_augmentation_bound=((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi)),
                    translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)))
    np.random.seed(int(s["seed"]))
    M_v, M_r = vox.get_transformation_matrix()
    assert (M_v == s["M_v"]).all() and (M_r == s["M_r"]).all()
    np.random.seed(int(s["seed"]))
    T, _ = vox.rigid_matrix()
    assert (T == s["matrix"]).all()


def test_synthetic_scenes_are_deterministic_and_shaped():
    s0 = syn.scene_s0()
    assert 8000 < s0.points.shape[0] <= 8192 and s0.images[0].shape == (512, 512, 3)
    s1a, s1b = syn.scene_s1(), syn.scene_s1()
    assert (s1a.points == s1b.points).all() and 119000 < s1a.points.shape[0] <= 120000 and len(s1a.poses) == 5
    vis, rows, cols = syn.view_subset(s1a, 0)
    assert 400 < vis.sum() < 65000 and rows.shape == cols.shape == (int(vis.sum()),)
    assert rows.min() >= 10 and rows.max() < 230 and cols.min() >= 10 and cols.max() < 310


def test_bilinear_down_is_bit_identical_to_interpolate():
    """mask_head.bilinear_down replaces F.interpolate(bilinear) for even integer shrink factors (odise.py:445-491 attention masks)"""
    import torch
    import torch.nn.functional as F
    from xmask3d_amd.mask_head import bilinear_down

    torch.manual_seed(0)
    x = torch.randn(2, 5, 128, 128)
    with torch.no_grad():
        for t in (16, 32, 64):
            assert torch.equal(bilinear_down(x, (t, t)), F.interpolate(x, size=(t, t), mode="bilinear", align_corners=False))
        assert torch.equal(bilinear_down(x, (48, 48)), F.interpolate(x, size=(48, 48), mode="bilinear", align_corners=False))  # fallback
