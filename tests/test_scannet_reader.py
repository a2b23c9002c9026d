"""On-disk ScanNet reader (xmask3d_amd/scannet.py) on a script-written tiny scene (tests/scannet_fixture.py): what the
reference's loader does up to the per-view samples (data_loader_infer.py:88-308) - label remap, colour restore, numeric frame
order, visibility filter, depth-occluded mapping, 512x512 frames."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.scannet_fixture import write_scene


@pytest.fixture(scope="module")
def disk_scene(tmp_path_factory):
    return write_scene(str(tmp_path_factory.mktemp("scannet")))


def test_reader_follows_the_reference_loader(disk_scene):
    from PIL import Image
    from xmask3d_amd import scannet, synthetic

    fx = disk_scene
    sc, ids = scannet.load_scene(fx["data_root"], fx["data_root_2d"], fx["scene"], caption_path=fx["caption_path"])
    assert ids == ["0", "20", "100"]                      # numeric order; frame 7 sees < 400 points and is dropped
    assert sc.captions == [f"a room seen from frame {i}" for i in ids]
    assert np.array_equal(sc.points, fx["points"])
    np.testing.assert_allclose(sc.colors, (fx["feats"].astype(np.float64) + 1) * 127.5)
    want = fx["labels"].copy()
    want[(want == -100) | (want == 255)] = 20
    assert np.array_equal(sc.labels, want) and (sc.labels == 20).sum() > 0
    rejected = 0
    for i, fid in enumerate(ids):
        assert np.array_equal(sc.poses[i], np.loadtxt(f"{fx['data_root_2d']}/{fx['scene']}/pose/{fid}.txt"))
        depth = np.asarray(Image.open(f"{fx['data_root_2d']}/{fx['scene']}/depth/{fid}.png")) / 1000
        assert np.array_equal(sc.depths[i], depth)
        # the occlusion test rejects points: fewer visible than by the frustum alone, and the scene container's view_subset
        # (what SceneOnDevice uploads) applies it
        m_occ = synthetic.project_points(sc.poses[i], sc.points, depth)
        m_all = synthetic.project_points(sc.poses[i], sc.points)
        assert 400 <= m_occ[:, 2].sum() <= m_all[:, 2].sum()
        rejected += int(m_all[:, 2].sum() - m_occ[:, 2].sum())
        vis, rows, cols = synthetic.view_subset(sc, i)
        assert np.array_equal(vis, m_occ[:, 2] == 1)
        # 512x512 frame: bilinear with half-pixel centres (cv2.resize default), rounded back to uint8 levels
        raw = np.asarray(Image.open(f"{fx['data_root_2d']}/{fx['scene']}/color/{fid}.jpg").convert("RGB"))
        ref = F.interpolate(torch.from_numpy(raw.copy()).permute(2, 0, 1)[None].float(), size=(512, 512), mode="bilinear", align_corners=False)
        got = torch.from_numpy(sc.images[i]).permute(2, 0, 1)[None]
        assert sc.images[i].shape == (512, 512, 3) and (got - ref).abs().max() <= 0.5 + 1e-3
        assert np.array_equal(sc.images[i], np.rint(sc.images[i]))
    assert rejected > 0


def test_reader_filters(disk_scene):
    from xmask3d_amd import scannet

    fx = disk_scene
    _, ids = scannet.load_scene(fx["data_root"], fx["data_root_2d"], fx["scene"], val_keep=1)   # every view too large
    assert ids == []
    _, ids = scannet.load_scene(fx["data_root"], fx["data_root_2d"], fx["scene"], ignore_categories=range(0, 21))  # no valid point
    assert ids == []
    with pytest.raises(FileNotFoundError):
        scannet.load_scene(fx["data_root"], fx["data_root_2d"], "scene9999_00")


@pytest.mark.gpu
def test_disk_scene_through_the_device_pipeline(dev, disk_scene):
    """device mapping (xm3d_compute_mapping) keeps the same frames and the same visible sets as the numpy restatement, and the
    scene runs through the whole inference path"""
    import os
    from xmask3d_amd import pipeline, scannet, synthetic
    from xmask3d_amd.config import load_cfg_from_cfg_file
    from xmask3d_amd.xmask3d import XMASK3d

    fx = disk_scene
    sc, ids = scannet.load_scene(fx["data_root"], fx["data_root_2d"], fx["scene"], caption_path=fx["caption_path"], device=dev)
    assert ids == ["0", "20", "100"]
    from xmask3d_amd import ops
    for i in range(3):
        got = ops.compute_mapping(torch.from_numpy(sc.points).to(dev), sc.poses[i], synthetic.scannet_intrinsics(),
                                  depth=torch.from_numpy(sc.depths[i]).to(dev)).cpu().numpy()
        assert np.array_equal(got, synthetic.project_points(sc.poses[i], sc.points, sc.depths[i]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_cfg_from_cfg_file(os.path.join(root, "configs", "xmask3d_scannet_B15N4.yaml"))
    torch.manual_seed(0)
    model = pipeline.make_inference_model(XMASK3d(cfg).eval(), dev, torch.bfloat16, graphs=False)
    sd = pipeline.SceneOnDevice(sc, dev)
    vox = pipeline.default_voxelizer(cfg.voxel_size, dev)
    fused, p2d, p3d = pipeline.infer_scene(model, sd, cfg, vox, [np.diag([50.0, 50.0, 50.0, 1.0])] * 3, views_per_batch=1)
    assert fused.shape[0] == sc.points.shape[0] and fused.dtype == torch.long
