"""GPU parity: sparse convolution kernels and the full MinkUNets vs the CPU oracle.

fp32 tolerance: the kernels accumulate in a different (documented) order than the oracle's matmul, so
results agree to f32 rounding, not bitwise: |err| <= 2e-5 * max|ref| per conv, 1e-3 * max|ref| for the
whole 63-conv network (BASELINE.json north_star: "per-point logits within 1e-3 fp")."""
import numpy as np
import pytest
import torch

from oracle import spconv_oracle as so

pytestmark = pytest.mark.gpu


def _coords(n, seed, hi=40, batches=2):
    r = np.random.RandomState(seed)
    c = np.unique(np.concatenate([r.randint(0, batches, (n, 1)), r.randint(0, hi, (n, 3))], 1), axis=0)
    return c[r.permutation(len(c))].astype(np.int32)


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-20)


@pytest.mark.parametrize("cin,cout,ks,n", [(3, 32, 5, 3000), (32, 32, 3, 5000), (64, 96, 3, 3000), (128, 96, 3, 1000),
                                           (384, 256, 3, 700), (32, 64, 1, 2000), (96, 32, 3, 1), (32, 32, 3, 255),
                                           (32, 32, 3, 257)])
def test_conv_kernels_match_oracle(dev, cin, cout, ks, n):
    from xmask3d_amd import ops

    torch.manual_seed(n + cin)
    c = _coords(n, cin + ks)
    N = len(c)
    cm = ops.CoordinateManager(torch.from_numpy(c).to(dev))
    nbr = cm.kernel_map(1, 1, ks)
    W = torch.randn(ks ** 3, cin, cout) / (cin * 4) ** 0.5
    f = torch.randn(N, cin)
    scale, shift, res = torch.rand(cout) + 0.5, torch.randn(cout), torch.randn(N, cout)
    ref_plain = so.spconv(f.double(), W.double(), nbr.cpu().numpy()).float()
    ref_epi = torch.relu(ref_plain * scale + shift + res)
    algos = [ops.ALGO_SCALAR] + ([ops.ALGO_MFMA, ops.ALGO_TILES, ops.ALGO_SPLIT] if ops.mfma_eligible(cin, cout) else [])
    for algo in algos:
        for order in (None, cm.order(1)):
            if algo in (ops.ALGO_TILES, ops.ALGO_SPLIT) and order is None:
                continue  # the tiled rulebook is built for the manager's processing order
            tiles = cm.tiles(1, 1, ks) if algo in (ops.ALGO_TILES, ops.ALGO_SPLIT) else None
            out = ops.spconv_fwd(f.to(dev), W.to(dev), nbr, N, order=order, algo=algo, tiles=tiles)
            assert _rel(out.cpu(), ref_plain) < 2e-5, (algo, order is None)
            out = ops.spconv_fwd(f.to(dev), W.to(dev), nbr, N, order=order, scale=scale.to(dev), shift=shift.to(dev),
                                 residual=res.to(dev), relu=True, algo=algo, tiles=tiles)
            assert _rel(out.cpu(), ref_epi) < 2e-5, (algo, order is None)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32), (32, 64), (64, 64), (96, 96), (128, 96), (96, 256), (192, 128),
                                      (256, 256), (384, 256), (128, 192), (96, 128), (64, 96), (32, 96), (96, 32), (96, 64)])
def test_split_kernel_every_instantiation(dev, cin, cout):
    """algo 4 (bf16 split-operand kernel): every (output-channel tile, channel chunk) instantiation, ragged tiles, k = 3 / 2 / 1,
    with and without split-K, against the f64 oracle at the f32 kernel's own 2e-5 bound; bitwise reproducible."""
    from xmask3d_amd import ops

    torch.manual_seed(cin * 7 + cout)
    c = _coords(2300, cin + cout, hi=20)
    N = len(c)
    cm = ops.CoordinateManager(torch.from_numpy(c).to(dev))
    f = torch.randn(N, cin)
    scale, shift, res = torch.rand(cout) + 0.5, torch.randn(cout), torch.randn(N, cout)
    for ks in (3, 1):
        nbr = None if ks == 1 else cm.kernel_map(1, 1, ks)
        tiles = cm.tiles(1, 1, ks)
        W = torch.randn(ks ** 3, cin, cout) / (cin * 4) ** 0.5
        ident = torch.arange(N, dtype=torch.int32)[None].numpy()
        ref = so.spconv(f.double(), W.double(), ident if nbr is None else nbr.cpu().numpy()).float()
        ref_epi = torch.relu(ref * scale + shift + res)
        for ksplit in (1, 3 if ks == 3 else 1):
            out = ops.spconv_fwd(f.to(dev), W.to(dev), nbr, N, order=cm.order(1), algo=ops.ALGO_SPLIT, tiles=tiles, ksplit=ksplit)
            assert _rel(out.cpu(), ref) < 2e-5, (ks, ksplit, _rel(out.cpu(), ref))
            out2 = ops.spconv_fwd(f.to(dev), W.to(dev), nbr, N, order=cm.order(1), scale=scale.to(dev), shift=shift.to(dev),
                                  residual=res.to(dev), relu=True, algo=ops.ALGO_SPLIT, tiles=tiles, ksplit=ksplit)
            assert _rel(out2.cpu(), ref_epi) < 2e-5
            again = ops.spconv_fwd(f.to(dev), W.to(dev), nbr, N, order=cm.order(1), algo=ops.ALGO_SPLIT, tiles=tiles, ksplit=ksplit)
            assert torch.equal(out, again)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32), (32, 64), (64, 64), (96, 96), (128, 96), (96, 256), (192, 128), (256, 256),
                                      (384, 256), (96, 64)])
def test_bf16_form_every_instantiation(dev, cin, cout):
    """the plain-bf16 form (xm3d_spconv_fwd_bf16: one bf16 plane in / out, one MFMA per product - the bf16 configuration's sparse
    convolution): every instantiation, k = 3 / 1, with and without split-K.  Against the f64 oracle evaluated on the SAME bf16-rounded
    operands (features, weights, residual) the only differences are the f32 summation order and the bf16 rounding of the result:
    bound 2^-8 relative to max|out| (measured <= 2e-3); an integer-valued case must be exact; bitwise reproducible."""
    from xmask3d_amd import ops

    torch.manual_seed(cin * 5 + cout)
    c = _coords(2300, cin + cout + 1, hi=20)
    N = len(c)
    cm = ops.CoordinateManager(torch.from_numpy(c).to(dev))
    bf = lambda t: t.to(torch.bfloat16)
    f = bf(torch.randn(N, cin))
    scale, shift, res = torch.rand(cout) + 0.5, torch.randn(cout), bf(torch.randn(N, cout))
    for ks in (3, 1):
        nbr = None if ks == 1 else cm.kernel_map(1, 1, ks)
        tiles = cm.tiles(1, 1, ks)
        W = torch.randn(ks ** 3, cin, cout) / (cin * 4) ** 0.5
        packed = ops.pack_weight_split(W.to(dev))
        ident = torch.arange(N, dtype=torch.int32)[None].numpy()
        ref = so.spconv(f.double(), bf(W).double(), ident if nbr is None else nbr.cpu().numpy()).float()
        ref_epi = torch.relu(ref * scale + shift + res.float())
        for ksplit in (1, 3 if ks == 3 else 1):
            out = ops.spconv_fwd_bf16(f.to(dev), tuple(W.shape), packed, tiles, N, order=cm.order(1), ksplit=ksplit)
            assert out.dtype == torch.bfloat16 and _rel(out.float().cpu(), ref) < 4e-3, (ks, ksplit, _rel(out.float().cpu(), ref))
            out2 = ops.spconv_fwd_bf16(f.to(dev), tuple(W.shape), packed, tiles, N, order=cm.order(1), scale=scale.to(dev), shift=shift.to(dev),
                                       residual=res.to(dev), relu=True, ksplit=ksplit)
            assert _rel(out2.float().cpu(), ref_epi) < 4e-3
            again = ops.spconv_fwd_bf16(f.to(dev), tuple(W.shape), packed, tiles, N, order=cm.order(1), ksplit=ksplit)
            assert torch.equal(out, again)
    # exact on small integers: features in {-2..2}, sparse {-1, 0, 1} weights - every product and sum representable in bf16 / f32
    g = torch.Generator().manual_seed(1)
    fi = torch.randint(-2, 3, (N, cin), generator=g).float()
    Wi = (torch.randint(-1, 2, (27, cin, cout), generator=g) * (torch.rand(27, cin, cout, generator=g) < 4.0 / cin)).float()
    nbr, tiles = cm.kernel_map(1, 1, 3), cm.tiles(1, 1, 3)
    refi = so.spconv(fi.double(), Wi.double(), nbr.cpu().numpy()).float()
    assert float(refi.abs().max()) <= 256
    outi = ops.spconv_fwd_bf16(bf(fi).to(dev), tuple(Wi.shape), ops.pack_weight_split(Wi.to(dev)), tiles, N, order=cm.order(1))
    assert torch.equal(outi.float().cpu(), refi)


def test_bf16_sparse_nets_track_the_f32_nets(dev):
    """MinkUNet34C + heads and MinkUNet18A + head in the bf16 configuration's sparse mode (XMASK3d.set_sparse_dtype(bf16): bf16
    activations between the layers, plain-bf16 convolutions) against the same nets in f32 (split-operand kernels, ~f32 accuracy): the
    budget of the bf16 sparse branch, measured 1e-2 (34C features) / 4e-3 (18A logits) on S1-like clouds - the level of the bf16 dense
    branch it conditions, not north_star's 1e-3 (the fp32 configuration keeps the f32 form)."""
    from xmask3d_amd import me_compat as ME
    from xmask3d_amd.pc_processor import PC_Binary_Processor, PC_Processor

    torch.manual_seed(11)
    c = _coords(9000, 3, hi=48, batches=2)
    coords = torch.from_numpy(c).to(dev)
    feats = (torch.rand(len(c), 3) * 2 - 1).to(dev)
    for net in (PC_Processor().to(dev).eval(), PC_Binary_Processor().to(dev).eval()):
        with torch.no_grad():
            ref = net(ME.SparseTensor(feats, coords))
            for m in net.modules():
                if isinstance(m, ME._ConvBase):
                    m.bf16_io = True
            got = net(ME.SparseTensor(feats, coords))
            got2 = net(ME.SparseTensor(feats, coords))
        ref, got, got2 = (r if isinstance(r, tuple) else (r,) for r in (ref, got, got2))
        for a, b, b2 in zip(ref, got, got2):
            if a.dtype.is_floating_point:
                assert b.dtype == torch.float32 and torch.equal(b, b2)
                assert _rel(b, a) < 4e-2, _rel(b, a)
            else:
                assert torch.equal(a, b)


def test_presplit_activations_chain(dev):
    """algo 4 with activations kept pre-split between layers: conv1 emits the bf16 hi / lo copy of its output, conv2 consumes
    it (producers then only move fragments); same result as the on-the-fly split, the copy reconstructs the f32 output to
    2^-16, and both split-K and direct epilogues write it"""
    from xmask3d_amd import ops

    torch.manual_seed(5)
    c = _coords(4000, 17, hi=24)
    N = len(c)
    cm = ops.CoordinateManager(torch.from_numpy(c).to(dev))
    nbr, tiles, order = cm.kernel_map(1, 1, 3), cm.tiles(1, 1, 3), cm.order(1)
    f = torch.randn(N, 64).to(dev)
    W1, W2 = (torch.randn(27, 64, 96) * 0.05).to(dev), (torch.randn(27, 96, 64) * 0.05).to(dev)
    for ksplit in (1, 4):
        y1, y1s = ops.spconv_fwd(f, W1, nbr, N, order=order, tiles=tiles, relu=True, algo=ops.ALGO_SPLIT, ksplit=ksplit, want_split=True)
        assert y1s.shape == (2, N, 96) and y1s.dtype == torch.bfloat16
        rec = y1s[0].float() + y1s[1].float()
        assert _rel(rec.cpu(), y1.cpu()) < 2e-5
        a = ops.spconv_fwd(y1, W2, nbr, N, order=order, tiles=tiles, algo=ops.ALGO_SPLIT, ksplit=ksplit)                     # split on the fly
        b = ops.spconv_fwd(y1, W2, nbr, N, order=order, tiles=tiles, algo=ops.ALGO_SPLIT, ksplit=ksplit, feats_split=y1s)    # pre-split
        ref = so.spconv(y1.cpu().double(), W2.cpu().double(), nbr.cpu().numpy()).float()
        assert _rel(a.cpu(), ref) < 2e-5 and _rel(b.cpu(), ref) < 2e-5
        assert _rel(b.cpu(), a.cpu()) < 1e-5


def test_roofline_shape_s1_full_96_to_96(dev):
    """the layer bench.py's `roofline` object is measured on (S1-full, 107 k voxels, 96 -> 96, k = 3, 418 k pairs): both
    tiled kernels against the oracle's per-offset gather-matmul-scatter in f64"""
    from xmask3d_amd import ops, synthetic

    sc = synthetic.scene_s1()
    grid, inds, inv = ops.voxelize(torch.from_numpy(sc.points).to(dev), np.diag([50.0, 50.0, 50.0, 1.0]))
    coords = torch.cat([torch.zeros(grid.shape[0], 1, dtype=torch.int32, device=dev), grid], 1).contiguous()
    cm = ops.CoordinateManager(coords)
    n = coords.shape[0]
    nbr, tiles, order = cm.kernel_map(1, 1, 3), cm.tiles(1, 1, 3), cm.order(1)
    assert n > 100000 and int((nbr >= 0).sum()) > 400000
    assert (nbr.cpu().numpy() == so.kernel_map(coords.cpu().numpy(), coords.cpu().numpy(), 3, 1)).all()
    g = torch.Generator().manual_seed(1)
    feats, W = torch.randn(n, 96, generator=g), torch.randn(27, 96, 96, generator=g) * 0.05
    ref = torch.relu(so.spconv(feats.double(), W.double(), nbr.cpu().numpy())).float()
    for algo in (ops.ALGO_TILES, ops.ALGO_SPLIT):
        out = ops.spconv_fwd(feats.to(dev), W.to(dev), nbr, n, order=order, tiles=tiles, relu=True, algo=algo)
        err = _rel(out.cpu(), ref)
        print(f"[S1-full 96->96 algo {algo}] rel err {err:.3e}")
        assert err < 2e-5, (algo, err)


def test_split_k_is_deterministic_and_matches(dev):
    from xmask3d_amd import ops

    torch.manual_seed(9)
    c = _coords(900, 3, hi=14)
    N = len(c)
    cm = ops.CoordinateManager(torch.from_numpy(c).to(dev))
    nbr, tiles = cm.kernel_map(1, 1, 3), cm.tiles(1, 1, 3)
    W, f = (torch.randn(27, 256, 64) * 0.05).to(dev), torch.randn(N, 256).to(dev)
    sc, sh = (torch.rand(64) + 0.5).to(dev), torch.randn(64).to(dev)
    ref = torch.relu(so.spconv(f.cpu().double(), W.cpu().double(), nbr.cpu().numpy()).float() * sc.cpu() + sh.cpu())
    outs = [ops.spconv_fwd(f, W, nbr, N, order=cm.order(1), scale=sc, shift=sh, relu=True, tiles=tiles, ksplit=ks)
            for ks in (1, 2, 5, 27, 27)]
    for o in outs:
        assert _rel(o.cpu(), ref) < 2e-5
    assert torch.equal(outs[3], outs[4])  # same split -> bitwise identical


def test_strided_and_transposed_conv(dev):
    from xmask3d_amd import ops

    torch.manual_seed(3)
    c = _coords(6000, 9)
    cm = ops.CoordinateManager(torch.from_numpy(c).to(dev))
    oc = so.CoordCache(c)
    f1 = torch.randn(len(c), 32)
    Wd, Wu = torch.randn(8, 32, 64) * 0.1, torch.randn(8, 64, 32) * 0.1
    down = ops.spconv_fwd(f1.to(dev), Wd.to(dev), cm.kernel_map(1, 2, 2), cm.num(2))
    ref_down = so.spconv(f1, Wd, oc.map(1, 2, 2))
    assert _rel(down.cpu(), ref_down) < 2e-5
    up = ops.spconv_fwd(down, Wu.to(dev), cm.kernel_map(2, 1, 2, True), cm.num(1), order=cm.order(1))
    assert _rel(up.cpu(), so.spconv(ref_down, Wu, oc.map(2, 1, 2, True))) < 2e-5


def test_empty_input_is_a_noop(dev):
    from xmask3d_amd import ops

    out = ops.spconv_fwd(torch.zeros(0, 32, device=dev), torch.zeros(27, 32, 32, device=dev),
                         torch.zeros(27, 0, dtype=torch.int32, device=dev), 0)
    assert out.shape == (0, 32)


def test_bn_stats_and_affine(dev):
    from xmask3d_amd import ops

    torch.manual_seed(0)
    x = torch.randn(10007, 96) * 3 + 1
    s, ss = ops.bn_stats(x.to(dev))
    assert torch.allclose(s.cpu(), x.double().sum(0), rtol=1e-12) and torch.allclose(ss.cpu(), (x.double() ** 2).sum(0), rtol=1e-12)
    sc, sh, r = torch.rand(96), torch.randn(96), torch.randn_like(x)
    y = ops.affine_act(x.to(dev), sc.to(dev), sh.to(dev), r.to(dev), True)
    assert torch.allclose(y.cpu(), torch.relu(x * sc + sh + r), atol=1e-6)


def _randomise_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


@pytest.mark.parametrize("arch", ["MinkUNet34C", "MinkUNet18A"])
def test_full_network_matches_oracle(dev, arch):
    from xmask3d_amd import me_compat as ME
    from xmask3d_amd.mink_unet import mink_unet

    torch.manual_seed(11)
    net = mink_unet(3, 256, 3, arch).eval()
    _randomise_bn(net, 5)
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    c = _coords(9000, 21, hi=48, batches=2)
    f = torch.rand(len(c), 3) * 2 - 1
    bott_r, c16, out_r = so.minkunet_forward(params, c, f, arch)
    net = net.to(dev)
    with torch.no_grad():
        bott, out = net(ME.SparseTensor(f.to(dev), torch.from_numpy(c).to(dev)))
    assert (bott.C.cpu().numpy() == c16).all()
    assert _rel(out.F.cpu(), out_r) < 1e-3 and _rel(bott.F.cpu(), bott_r) < 1e-3


def test_training_mode_batchnorm_matches_oracle(dev):
    from xmask3d_amd import me_compat as ME
    from xmask3d_amd.mink_unet import mink_unet

    torch.manual_seed(2)
    net = mink_unet(3, 32, 3, "MinkUNet14A").train()
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    c = _coords(4000, 5, hi=40, batches=2)
    f = torch.rand(len(c), 3) * 2 - 1
    _, _, out_r = so.minkunet_forward(params, c, f, "MinkUNet14A", training=True)
    net = net.to(dev)
    with torch.no_grad():
        _, out = net(ME.SparseTensor(f.to(dev), torch.from_numpy(c).to(dev)))
    assert _rel(out.F.cpu(), out_r) < 1e-3
    assert int(net.bn0.bn.num_batches_tracked) == 1


def test_reference_style_heads(dev):
    from xmask3d_amd import me_compat as ME
    from xmask3d_amd.pc_processor import PC_Binary_Processor, PC_Processor

    torch.manual_seed(4)
    c = _coords(5000, 8, hi=40, batches=2)
    f = torch.rand(len(c), 3) * 2 - 1
    for cls, fn, arch in ((PC_Processor, so.pc_processor_forward, "MinkUNet34C"), (PC_Binary_Processor, so.pc_binary_forward, "MinkUNet18A")):
        net = cls(arch_3d=arch).eval()
        params = {k: v.detach().clone() for k, v in net.state_dict().items()}
        ref = fn(params, c, f, arch)
        net = net.to(dev)
        with torch.no_grad():
            got = net(ME.SparseTensor(f.to(dev), torch.from_numpy(c).to(dev)))
        if cls is PC_Processor:
            assert _rel(got[0].cpu(), ref[0]) < 1e-3 and _rel(got[1].cpu(), ref[1]) < 1e-3
            assert (got[2].cpu().long() == ref[2]).all()
        else:
            assert _rel(got.cpu(), ref) < 1e-3
