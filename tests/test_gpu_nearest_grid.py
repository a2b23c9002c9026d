"""xm3d_nearest_valid_fill (uniform grid) against the brute-force xm3d_nearest_index: the SAME index, not just the same
distance (both compute d = fma(dz,dz,fma(dy,dy,dx*dx)) in f32 and take the lowest index among equal distances)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def brute(xyz, valid):
    from xmask3d_amd import ops

    n = xyz.shape[0]
    if not bool(valid.any()):
        return torch.arange(n, device=xyz.device)
    nn = ops.nearest_index(xyz, xyz, valid.to(torch.uint8))
    return torch.where(valid, torch.arange(n, device=xyz.device), nn)


def cases():
    g = torch.Generator().manual_seed(3)
    out = {}
    xyz = torch.rand(20000, 3, generator=g) * torch.tensor([6.0, 5.0, 2.6])
    out["uniform_35pct"] = (xyz, torch.rand(20000, generator=g) < 0.35)
    out["sparse_valid"] = (xyz, torch.rand(20000, generator=g) < 0.002)                  # long shell walks
    v = torch.zeros(20000, dtype=torch.bool)
    v[123] = True
    out["single_valid"] = (xyz, v)
    out["none_valid"] = (xyz, torch.zeros(20000, dtype=torch.bool))
    out["all_valid"] = (xyz, torch.ones(20000, dtype=torch.bool))
    # two far clusters hold the valid points, the queries fill the gap and lie outside the clusters' box as well
    a = torch.rand(3000, 3, generator=g) * 0.5
    b = torch.rand(3000, 3, generator=g) * 0.5 + torch.tensor([5.0, 4.0, 2.0])
    q = torch.rand(9000, 3, generator=g) * torch.tensor([8.0, 7.0, 4.0]) - 1.0
    out["two_clusters"] = (torch.cat([a, q, b]), torch.cat([torch.ones(3000), torch.zeros(9000), torch.ones(3000)]).bool())
    # ties: points on an integer lattice (many exactly equal distances) and exact duplicates with different indices
    lat = torch.stack(torch.meshgrid(torch.arange(24.0), torch.arange(24.0), torch.arange(12.0), indexing="ij"), -1).reshape(-1, 3) * 0.25
    lat = torch.cat([lat, lat[:500]])[torch.randperm(lat.shape[0] + 500, generator=g)]
    out["lattice_ties"] = (lat, torch.rand(lat.shape[0], generator=g) < 0.3)
    # large offsets (ScanNet scenes are not centred) and a scene big enough to grow the cell edge past 0.1 m
    out["offset_scene"] = (xyz + torch.tensor([1000.0, -500.0, 30.0]), torch.rand(20000, generator=g) < 0.2)
    out["wide_scene"] = (xyz * 20.0, torch.rand(20000, generator=g) < 0.2)
    out["planar"] = (xyz * torch.tensor([1.0, 1.0, 0.0]), torch.rand(20000, generator=g) < 0.1)  # zero extent along z
    out["tiny"] = (xyz[:5], torch.tensor([False, True, False, False, True]))
    return out


@pytest.mark.parametrize("name", list(cases().keys()))
def test_grid_fill_equals_bruteforce(dev, name):
    from xmask3d_amd import ops

    xyz, valid = cases()[name]
    xyz, valid = xyz.to(dev).contiguous(), valid.to(dev)
    got = ops.nearest_valid_fill(xyz, valid)
    want = brute(xyz, valid)
    assert torch.equal(got, want), (name, int((got != want).sum()))
    for cell in (0.03, 0.5):  # the answer does not depend on the cell edge
        assert torch.equal(ops.nearest_valid_fill(xyz, valid, cell), want), (name, cell)
    got = ops.nearest_valid_fill(xyz, valid, method="sorted")  # Morton-sorted, tile-pruned scan: the same indices again
    assert torch.equal(got, want), (name, "sorted", int((got != want).sum()))


def test_grid_fill_at_scene_size(dev):
    """the vote fill of a whole scene (~120 k points, the seen ones are the references)"""
    from xmask3d_amd import ops, pipeline, synthetic

    sc = synthetic.scene_s1()
    sd = pipeline.SceneOnDevice(sc, dev)
    seen = torch.zeros(sd.n, dtype=torch.bool, device=dev)
    for v in sd.views[:3]:
        seen |= v["vis"]
    xyz = sd.points.float().contiguous()
    assert torch.equal(ops.nearest_valid_fill(xyz, seen), brute(xyz, seen))
    assert torch.equal(ops.nearest_valid_fill(xyz, seen, method="sorted"), brute(xyz, seen))


def test_grid_fill_argument_errors(dev):
    from xmask3d_amd import ops

    xyz = torch.rand(10, 3, device=dev)
    with pytest.raises(RuntimeError):
        ops.nearest_valid_fill(xyz, torch.ones(9, dtype=torch.bool, device=dev))
    with pytest.raises(RuntimeError):
        ops.nearest_valid_fill(xyz, torch.ones(10, dtype=torch.bool, device=dev), cell=0.0)
    assert ops.nearest_valid_fill(xyz[:0], torch.ones(0, dtype=torch.bool, device=dev)).shape == (0,)
