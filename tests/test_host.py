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


def test_bench_scene_groups_are_balanced():
    import bench

    assert bench.balanced_groups(16, 4) == [4, 4, 4, 4]
    assert bench.balanced_groups(5, 4) == [3, 2] and bench.balanced_groups(1, 4) == [1] and bench.balanced_groups(0, 4) == []
    for n in range(1, 40):
        for g in (1, 2, 4):
            sizes = bench.balanced_groups(n, g)
            assert sum(sizes) == n and max(sizes) <= g and max(sizes) - min(sizes) <= 1


def test_gate_equals_the_reference_formula():
    """pipeline._gate (masked_fill + where) == binary_pred * logits_base + (1 - binary_pred) * logits_novel (infer.py:489-507)"""
    import torch
    from xmask3d_amd import pipeline

    torch.manual_seed(0)
    logits = torch.randn(200, 19)
    binary = (torch.rand(200, 1) > 0.5).long()
    base, novel = [0, 1, 2, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17, 18], [3, 4, 6, 16]
    bm, nm = torch.zeros(19, dtype=torch.bool), torch.zeros(19, dtype=torch.bool)
    bm[base], nm[novel] = True, True
    lb, ln = logits.clone(), logits.clone()
    ln[:, base] = -1e10
    lb[:, novel] = -1e10
    assert torch.equal(pipeline._gate(logits, binary, bm, nm), binary * lb + (1 - binary) * ln)


def test_scene_tables_concatenate_the_views():
    import torch
    from xmask3d_amd import pipeline, synthetic

    sd = pipeline.SceneOnDevice(synthetic.scene_s0(), torch.device("cpu"))
    assert torch.equal(sd.idx_all, torch.cat([v["idx"] for v in sd.views]))
    assert torch.equal(sd.x_all, torch.cat([v["x"] for v in sd.views])) and torch.equal(sd.y_all, torch.cat([v["y"] for v in sd.views]))
    off = 0
    for i, v in enumerate(sd.views):
        n = v["idx"].shape[0]
        assert bool((sd.view_all[off:off + n] == i).all())
        off += n
    assert off == sd.view_all.shape[0]
