"""CPU: self-consistency pins of the sparse-conv oracle (parity unpinned vs MinkowskiEngine, see
oracle/spconv_oracle.py): rulebook == brute force, odd-k stride-1 conv == dense conv3d."""
import numpy as np
import torch

from oracle import spconv_oracle as so


def _random_coords(n, extent, seed, batches=2):
    r = np.random.RandomState(seed)
    c = np.unique(np.concatenate([r.randint(0, batches, (n, 1)), r.randint(-3, extent, (n, 3))], 1), axis=0)
    return c[r.permutation(len(c))].astype(np.int32)


def test_stride_and_offsets():
    c = np.array([[0, 0, 0, 0], [0, 1, 1, 1], [0, -1, 2, 3], [1, 5, 5, 5]], dtype=np.int32)
    s2 = so.stride_coords(c, 2)
    assert s2.tolist() == [[0, -2, 2, 2], [0, 0, 0, 0], [1, 4, 4, 4]]
    o3 = so.kernel_offsets(3, 1)
    assert o3[0].tolist() == [-1, -1, -1] and o3[1].tolist() == [0, -1, -1] and o3[13].tolist() == [0, 0, 0]
    o2 = so.kernel_offsets(2, 4)
    assert o2.tolist() == [[0, 0, 0], [4, 0, 0], [0, 4, 0], [4, 4, 0], [0, 0, 4], [4, 0, 4], [0, 4, 4], [4, 4, 4]]


def test_kernel_map_vs_bruteforce():
    c = _random_coords(400, 8, 1)
    nbr = so.kernel_map(c, c, 3, 1)
    offs = so.kernel_offsets(3, 1)
    lut = {tuple(r): i for i, r in enumerate(c.tolist())}
    for k in range(27):
        for o in range(len(c)):
            q = (c[o, 0], c[o, 1] + offs[k, 0], c[o, 2] + offs[k, 1], c[o, 3] + offs[k, 2])
            assert nbr[k, o] == lut.get(tuple(int(v) for v in q), -1)
    assert (nbr[13] == np.arange(len(c))).all()


def test_down_up_maps_are_transposes():
    c = _random_coords(500, 12, 2)
    cm = so.CoordCache(c)
    down = cm.map(1, 2, 2)           # (8, N2) rows of level 1
    up = cm.map(2, 1, 2, True)       # (8, N1) rows of level 2
    pairs_down = {(k, int(down[k, o]), o) for k in range(8) for o in range(down.shape[1]) if down[k, o] >= 0}
    pairs_up = {(k, o, int(up[k, o])) for k in range(8) for o in range(up.shape[1]) if up[k, o] >= 0}
    assert pairs_down == pairs_up
    assert (np.sum(up >= 0, 0) == 1).all()  # every fine voxel has exactly one parent
    assert len(pairs_down) == len(c)


def test_conv_matches_dense_conv3d():
    torch.manual_seed(0)
    c = _random_coords(300, 6, 3, batches=1)
    c[:, 1:] -= c[:, 1:].min(0)
    D = int(c[:, 1:].max()) + 1
    cin, cout = 5, 7
    f = torch.randn(len(c), cin, dtype=torch.float64)
    W = torch.randn(27, cin, cout, dtype=torch.float64)
    out = so.spconv(f, W, so.kernel_map(c, c, 3, 1))
    dense = torch.zeros(1, cin, D, D, D, dtype=torch.float64)  # [z][y][x]
    dense[0, :, c[:, 3], c[:, 2], c[:, 1]] = f.T
    # kernel index k = (dz+1)*9 + (dy+1)*3 + (dx+1): x fastest -> conv3d weight [cout, cin, kz, ky, kx]
    w3 = W.reshape(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2)
    ref = torch.nn.functional.conv3d(dense, w3, padding=1)[0][:, c[:, 3], c[:, 2], c[:, 1]].T
    assert torch.allclose(out, ref, atol=1e-10)


def test_minkunet_oracle_runs_and_shapes():
    torch.manual_seed(1)
    from xmask3d_amd.mink_unet import mink_unet

    net = mink_unet(3, 16, 3, "MinkUNet14A").eval()
    params = {k: v.detach() for k, v in net.state_dict().items()}
    c = _random_coords(600, 20, 4, batches=2)
    c[:, 1:] -= c[:, 1:].min(0)
    bott, c16, out = so.minkunet_forward(params, c, torch.randn(len(c), 3), "MinkUNet14A")
    assert out.shape == (len(c), 16) and bott.shape == (len(c16), 256)
    assert torch.isfinite(out).all()
