"""CPU: the voxelisation oracle against vectors captured from the reference itself."""
import os

import numpy as np

from oracle import voxel_oracle as vo


def test_fnv_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "voxel_kat.npz"))
    assert (vo.fnv_keys(g["kat_in"]) == g["kat_keys"]).all()
    # the values SURVEY.md §8c quotes
    assert int(vo.fnv_keys(np.array([[0.0, 0, 0]]))[0]) == 15658191375538532279
    assert int(vo.fnv_keys(np.array([[287.0, 130, 209]]))[0]) == 15383797253656502647
    assert (vo.fnv_keys(g["big"]) == g["big_keys"]).all()


def test_sparse_quantize_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "voxel_kat.npz"))
    inds, inv = vo.unique_first(vo.fnv_keys(g["sq_in"]))
    assert inds.tolist() == [5, 0, 3, 1] and inv.tolist() == [1, 3, 1, 2, 3, 0]
    inds, inv = vo.unique_first(vo.fnv_keys(g["big"]))
    assert (inds == g["big_inds"]).all() and (inv == g["big_inv"]).all()


def test_voxelize_scenes_bit_exact(golden_dir):
    for tag in "ab":
        s = np.load(os.path.join(golden_dir, f"voxel_scene_{tag}.npz"))
        np.random.seed(int(s["seed"]))
        grid, feats, labels, inv, inds, M = vo.voxelize(s["pts"], s["feats"], s["labels"])
        assert (M == s["matrix"]).all()
        assert (grid == s["locs"]).all() and (inv == s["inv"]).all() and (inds == s["inds"]).all()
        assert (feats == s["vfeats"]).all() and (labels == s["vlabels"]).all()
        grid2, inds2, inv2 = vo.voxelize_with_matrix(s["pts"], s["matrix"])
        assert (grid2 == s["locs"]).all() and (inv2 == s["inv"]).all()


def test_pack_batch_offsets():
    g = [np.array([[0.0, 1, 2], [3, 4, 5]]), np.array([[7.0, 7, 7]])]
    f = [np.full((2, 3), 255.0), np.zeros((1, 3))]
    inv = [np.array([0, 1, 1]), np.array([0, 0])]
    c, ff, ii = vo.pack_batch(g, f, inv)
    assert c.dtype == np.int32 and c.tolist() == [[0, 0, 1, 2], [0, 3, 4, 5], [1, 7, 7, 7]]
    assert np.allclose(ff[0], 1.0) and np.allclose(ff[2], -1.0)
    assert ii.tolist() == [0, 1, 1, 2, 2]
