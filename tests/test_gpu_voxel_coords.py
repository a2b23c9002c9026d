"""GPU parity: voxelisation and coordinate-manager kernels vs the oracle / golden vectors (bit-exact)."""
import os

import numpy as np
import pytest
import torch

from oracle import spconv_oracle as so
from oracle import voxel_oracle as vo

pytestmark = pytest.mark.gpu


def test_fnv_keys_known_answers(dev, golden_dir):
    from xmask3d_amd import ops

    g = np.load(os.path.join(golden_dir, "voxel_kat.npz"))
    for arr, want in ((g["kat_in"], g["kat_keys"]), (g["big"], g["big_keys"])):
        keys = ops.fnv_keys(torch.from_numpy(arr.astype(np.int32)).to(dev))
        assert (keys.cpu().numpy().view(np.uint64) == want).all()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_voxelize_golden_scene(dev, golden_dir, tag):
    from xmask3d_amd import ops

    s = np.load(os.path.join(golden_dir, f"voxel_scene_{tag}.npz"))
    grid, inds, inv = ops.voxelize(torch.from_numpy(s["pts"]).to(dev), s["matrix"])
    assert (grid.cpu().numpy() == s["locs"]).all()
    assert (inds.cpu().numpy() == s["inds"]).all()
    assert (inv.cpu().numpy() == s["inv"]).all()


def test_voxelizer_class_matches_reference_outputs(dev, golden_dir):
    from xmask3d_amd.voxelizer import Voxelizer

    s = np.load(os.path.join(golden_dir, "voxel_scene_a.npz"))
    vox = Voxelizer(voxel_size=0.02, use_augmentation=True, scale_augmentation_bound=(0.9, 1.1),
                    rotation_augmentation_bound=((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi)),
                    translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)))
    np.random.seed(int(s["seed"]))
    locs, feats, labels, inv, inds = vox.voxelize(s["pts"], s["feats"].copy(), s["labels"].copy(), return_ind=True)
    assert locs.dtype == np.float64 and (locs == s["locs"]).all() and (inv == s["inv"]).all()
    assert (feats == s["vfeats"]).all() and (labels == s["vlabels"]).all() and (inds == s["inds"]).all()


def test_voxelize_edge_cases(dev):
    from xmask3d_amd import ops
    from xmask3d_amd._lib import Xm3dError

    T = np.diag([50.0, 50.0, 50.0, 1.0])
    one = torch.tensor([[0.5, 0.25, 0.125]], dtype=torch.float64, device=dev)
    grid, inds, inv = ops.voxelize(one, T)
    assert grid.tolist() == [[0, 0, 0]] and inds.tolist() == [0] and inv.tolist() == [0]
    same = one.repeat(1000, 1)  # every point in one voxel: first index wins
    grid, inds, inv = ops.voxelize(same, T)
    assert grid.shape[0] == 1 and inds.tolist() == [0] and (inv == 0).all()
    with pytest.raises(Xm3dError):
        ops.voxelize(torch.zeros(0, 3, dtype=torch.float64, device=dev), T)  # the reference asserts n > 0
    # full-size property test: S1 (120k points): idempotence + inverse consistency
    from xmask3d_amd import synthetic

    pts = synthetic.scene_s1().points
    grid, inds, inv = ops.voxelize(torch.from_numpy(pts).to(dev), T)
    g_ref, i_ref, v_ref = vo.voxelize_with_matrix(pts, T)
    assert (grid.cpu().numpy() == g_ref).all() and (inds.cpu().numpy() == i_ref).all() and (inv.cpu().numpy() == v_ref).all()
    keys = ops.fnv_keys(grid.contiguous()).cpu().numpy().view(np.uint64)
    assert (np.diff(keys.astype(np.float64)) > 0).all() or (keys[1:] > keys[:-1]).all()  # ascending key order


def _coords(n, seed, lo=-5, hi=60, batches=3):
    r = np.random.RandomState(seed)
    c = np.unique(np.concatenate([r.randint(0, batches, (n, 1)), r.randint(lo, hi, (n, 3))], 1), axis=0)
    return c[r.permutation(len(c))].astype(np.int32)


@pytest.mark.parametrize("n,seed", [(1, 0), (700, 1), (40000, 2)])
def test_strided_coords_and_maps(dev, n, seed):
    from xmask3d_amd import ops

    c = _coords(n, seed)
    cm = ops.CoordinateManager(torch.from_numpy(c).to(dev))
    oc = so.CoordCache(c)
    order = cm.order(1).cpu().numpy()
    assert (order == np.argsort(so.pack_keys(c), kind="stable")).all()
    for ts in (2, 4, 8, 16):
        assert (cm.coords(ts).cpu().numpy() == oc.level(ts)).all(), ts
    for key in [(1, 1, 3, False), (1, 1, 5, False), (2, 2, 3, False), (1, 2, 2, False), (4, 8, 2, False), (2, 1, 2, True),
                (16, 8, 2, True), (8, 8, 3, False)]:
        got = cm.kernel_map(*key).cpu().numpy()
        assert (got == oc.map(*key)).all(), key
        inv = cm.inverse_map(*key).cpu().numpy()
        K, n_out = got.shape
        for k in range(0, K, max(1, K // 5)):
            o = np.nonzero(got[k] >= 0)[0]
            assert (inv[k][got[k][o]] == o).all()
            assert (inv[k] >= 0).sum() == len(o)
    cm.check()


def test_duplicate_and_out_of_range_coordinates_raise(dev):
    from xmask3d_amd import me_compat as ME
    from xmask3d_amd._lib import Xm3dError

    c = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 3]], dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="unique"):
        ME.SparseTensor(torch.zeros(2, 3, device=dev), c)
    far = torch.tensor([[0, 40000, 0, 0]], dtype=torch.int32, device=dev)
    with pytest.raises(Xm3dError):
        ME.SparseTensor(torch.zeros(1, 3, device=dev), far)


def test_compute_mapping_matches_reference_golden(dev, golden_dir):
    """xm3d_compute_mapping against PointCloudToImageMapper.compute_mapping captured from the imported reference
    (tests/golden/mapping.npz): with and without the depth-occlusion test, bit-exact pixel indices, and against the package's
    own numpy restatement (synthetic.project_points) on the S1 scene's five views"""
    import os

    from xmask3d_amd import ops, synthetic

    g = np.load(os.path.join(golden_dir, "mapping.npz"))
    pts = torch.from_numpy(g["pts"]).to(dev)
    dim = tuple(int(v) for v in g["image_dim"])
    got = ops.compute_mapping(pts, g["pose"], g["intrinsic"], dim)
    assert (got.cpu().numpy() == g["map_nodepth"]).all() and int(got[:, 2].sum()) > 100
    got_d = ops.compute_mapping(pts, g["pose"], g["intrinsic"], dim, depth=torch.from_numpy(g["depth"]).to(dev))
    assert (got_d.cpu().numpy() == g["map_depth"]).all() and 0 < int(got_d[:, 2].sum()) < int(got[:, 2].sum())
    sc = synthetic.scene_s1()
    P = torch.from_numpy(sc.points).to(dev)
    for v in range(len(sc.poses)):
        want = synthetic.project_points(sc.poses[v], sc.points)
        assert (ops.compute_mapping(P, sc.poses[v], synthetic.scannet_intrinsics()).cpu().numpy() == want).all()
