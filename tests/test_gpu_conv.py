"""GPU parity of the fused GroupNorm -> SiLU -> conv3x3 (+ bias, + residual, + output moments) kernel (xm3d_conv3x3_nhwc,
csrc/conv.hip) through the C ABI.

References, all computed by torch in f32 on the same device from the same bf16 inputs:
  * exact: small-integer activations and sparse {-1,0,1} weights (every product and partial sum is exact in bf16 / f32), plain
    convolution mode, asymmetric random data - any indexing / layout / tap-order slip changes integers, so the comparison is
    bit-exact against F.conv2d;
  * fused: F.conv2d(bf16(silu(F.group_norm(x))), w) + bias + residual.  bf16 output: bound 1e-2 of max|out| (output rounding
    2^-9 relative plus single-ulp differences of the bf16-rounded normalised operand; measured ~3e-3);
  * moments of the stored output against sums over the returned tensor.
The reference applies this chain as torch.nn.GroupNorm / SiLU / Conv2d inside ldm's ResnetBlock (stable-diffusion-sdkit, absent
here): numerics of that package are unpinned, the arithmetic definition of the three ops is torch's."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("B,cin,cout,H,W,ups", [(1, 64, 128, 8, 32, False), (2, 128, 256, 16, 64, False), (1, 192, 128, 24, 32, False),
                                                (2, 64, 256, 16, 64, True), (1, 128, 512, 8, 96, False), (1, 64, 128, 16, 32, True)])
def test_conv3x3_plain_is_exact_on_integer_data(dev, B, cin, cout, H, W, ups, waves):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(cin * 7 + cout + H + W)
    hi, wi = (H // 2, W // 2) if ups else (H, W)
    x = torch.randint(-2, 3, (B, cin, hi, wi), generator=g).float()
    w = torch.randint(-1, 2, (cout, cin, 3, 3), generator=g).float() * (torch.rand(cout, cin, 3, 3, generator=g) < 0.06).float()
    bias = torch.randint(-3, 4, (cout,), generator=g).float()
    xd = _nhwc(x.to(dev, torch.bfloat16))
    packed, tile = ops.conv3x3_pack_weight(w.to(dev))
    out = ops.conv3x3(xd, packed, cout, tile, bias=bias.to(dev), upsample=ups, waves=waves)
    xin = x.to(dev)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.to(dev), bias.to(dev), padding=1)
    assert ref.abs().max().item() < 256  # exactly representable in bf16
    assert out.shape == ref.shape and out.dtype == torch.bfloat16 and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out.float(), ref), (out.float() - ref).abs().max().item()


@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("B,cin,cout,H,W,res,per_sample_bias", [(2, 128, 128, 16, 64, True, False), (1, 256, 256, 8, 32, False, True),
                                                                (2, 128, 256, 16, 32, False, False), (1, 512, 512, 8, 64, True, True),
                                                                (3, 320, 640, 8, 32, False, True), (1, 128, 128, 40, 32, True, False), (1, 64, 128, 8, 32, False, False)])
def test_conv3x3_groupnorm_silu_matches_torch(dev, B, cin, cout, H, W, res, per_sample_bias, waves):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(cin + cout * 3 + H)
    G = 32
    x = _nhwc((torch.randn(B, cin, H, W, generator=g) * 1.7 + 0.3 * torch.randn(1, cin, 1, 1, generator=g)).to(dev, torch.bfloat16))
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev, torch.bfloat16)
    gamma = (1 + 0.2 * torch.randn(cin, generator=g)).to(dev)
    beta = (0.2 * torch.randn(cin, generator=g)).to(dev)
    bias = (0.3 * torch.randn((B, cout) if per_sample_bias else (cout,), generator=g)).to(dev)
    residual = _nhwc(torch.randn(B, cout, H, W, generator=g).to(dev, torch.bfloat16)) if res else None
    eps = 1e-6
    packed, tile = ops.conv3x3_pack_weight(w)
    stats = ops.gn_stats_of(x, G)
    gs = 32 if (cout // 32) % 4 == 0 else None
    out = ops.conv3x3(x, packed, cout, tile, bias=bias, gn=(stats, gamma, beta, eps, G), residual=residual, stats_groups=gs, waves=waves)

    xn = F.silu(F.group_norm(x.float(), G, gamma, beta, eps)).to(torch.bfloat16).float()
    ref = F.conv2d(xn, w.float(), None, padding=1) + bias.view(-1 if per_sample_bias else 1, cout, 1, 1)
    if res:
        ref = ref + residual.float()
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-2, err
    if gs:
        st = ops.gn_stats_of(out, gs).view(B, gs, 2)
        o = out.float().view(B, gs, cout // gs, H * W)
        want = torch.stack([o.sum((2, 3)), (o * o).sum((2, 3))], -1).double()
        rel = ((st - want).abs() / (want.abs() + 1.0)).max().item()
        assert rel < 1e-4, rel


def test_conv3x3_chain_is_a_resblock(dev):
    """two fused convolutions = one VAE ResnetBlock (norm1-silu-conv1-norm2-silu-conv2 + x), the second GroupNorm reading the
    moments the first convolution's epilogue accumulated"""
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(11)
    B, C, H, W, G = 2, 128, 16, 64, 32
    x = _nhwc(torch.randn(B, C, H, W, generator=g).to(dev, torch.bfloat16))
    w1, w2 = ((torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev, torch.bfloat16) for _ in range(2))
    g1, b1, g2, b2 = ((1 + 0.1 * torch.randn(C, generator=g)).to(dev) for _ in range(4))
    c1, c2 = (0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    p1, t1 = ops.conv3x3_pack_weight(w1)
    p2, t2 = ops.conv3x3_pack_weight(w2)
    h = ops.conv3x3(x, p1, C, t1, bias=c1, gn=(ops.gn_stats_of(x, G), g1, b1, 1e-6, G), stats_groups=G)
    assert getattr(h, "_xm3d_gn_stats", None) is not None
    out = ops.conv3x3(h, p2, C, t2, bias=c2, gn=(ops.gn_stats_of(h, G), g2, b2, 1e-6, G), residual=x, stats_groups=G)

    hr = F.conv2d(F.silu(F.group_norm(x.float(), G, g1, b1, 1e-6)).to(torch.bfloat16).float(), w1.float(), c1, padding=1).to(torch.bfloat16).float()
    ref = x.float() + F.conv2d(F.silu(F.group_norm(hr, G, g2, b2, 1e-6)).to(torch.bfloat16).float(), w2.float(), c2, padding=1)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1.5e-2, err


def test_conv3x3_rejects_what_it_cannot_run(dev):
    from xmask3d_amd import ops
    from xmask3d_amd._lib import Xm3dError

    x = _nhwc(torch.zeros(1, 64, 8, 16, device=dev, dtype=torch.bfloat16))  # W % 32 != 0
    assert not ops.conv3x3_supported(x, 128)
    assert not ops.conv3x3_supported(_nhwc(torch.zeros(1, 64, 8, 32, device=dev, dtype=torch.bfloat16)), 64)  # half-empty only tile
    with pytest.raises(TypeError):
        ops.conv3x3_pack_weight(torch.zeros(80, 64, 3, 3, device=dev))  # cout % 32 != 0
    x = _nhwc(torch.zeros(1, 64, 8, 32, device=dev, dtype=torch.bfloat16))
    packed, tile = ops.conv3x3_pack_weight(torch.zeros(128, 64, 3, 3, device=dev))
    with pytest.raises(Xm3dError):  # an activation without statistics
        from xmask3d_amd.ops import _ptr, _stream, lib
        from xmask3d_amd._lib import check
        out = torch.empty_like(x)
        check(lib().xm3d_conv3x3_nhwc(_ptr(x), 1, 8, 32, 64, _ptr(packed), 128, tile, None, None, None, None, 0, 0.0, 0, 1, None, 0, None, _ptr(out), None, 0, 0,
                                      0, None, _stream()), "xm3d_conv3x3_nhwc")


def _module_pair(mod_fn, dev):
    """the same frozen module twice: bf16 channels-last (HIP convolution path) and f32 (torch ops)"""
    torch.manual_seed(5)
    ref = mod_fn().to(dev).eval()
    for prm in ref.parameters():
        prm.data.copy_(prm.data.to(torch.bfloat16).float())  # both copies hold bf16-representable weights
    fast = mod_fn().to(dev).eval()
    fast.load_state_dict(ref.state_dict())
    return ref, fast.to(torch.bfloat16).to(memory_format=torch.channels_last)


@pytest.mark.parametrize("cin,cout,H,W,with_pend", [(128, 128, 16, 64, True), (128, 256, 16, 32, True), (256, 256, 8, 64, False), (512, 512, 8, 32, False)])
def test_vae_resblock_on_hip_convolutions(dev, cin, cout, H, W, with_pend):
    from xmask3d_amd import sd_model

    ref, fast = _module_pair(lambda: sd_model.VaeResBlock(cin, cout), dev)
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, H, W, generator=g).to(dev, torch.bfloat16)
    pend = (0.3 * torch.randn(cin, generator=g)).to(dev, torch.bfloat16) if with_pend else None
    with torch.no_grad():
        xf = _nhwc(x)
        assert sd_model.fused_conv_ok(xf, fast.conv1)
        out = fast(xf, pend=pend)
        want = ref(x.float(), pend=None if pend is None else pend.float())
    assert getattr(out, "_xm3d_gn_stats", None) is not None  # moments for the next block's norm1
    err = (out.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 2e-2, err


def test_unet_resblock_and_upsample_on_hip_convolutions(dev):
    from xmask3d_amd import sd_model

    ref, fast = _module_pair(lambda: sd_model.UNetResBlock(1280, 1280, 640), dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1280, 32, 32, generator=g).to(dev, torch.bfloat16)
    emb = torch.randn(2, 1280, generator=g).to(dev, torch.bfloat16)
    with torch.no_grad():
        assert sd_model.fused_conv_ok(_nhwc(x), fast.in_layers[2])
        out = fast(_nhwc(x), emb)
        want = ref(x.float(), emb.float())
    err = (out.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 2e-2, err

    ref, fast = _module_pair(lambda: sd_model.VaeUpsample(256), dev)
    x = torch.randn(2, 256, 16, 32, generator=g).to(dev, torch.bfloat16)
    with torch.no_grad():
        out, pend = fast(_nhwc(x), defer_bias=True)
        want = ref(x.float())
    assert pend is None and tuple(out.shape) == (2, 256, 32, 64)
    err = (out.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-2, err


def test_conv3x3_groupnorm_relu_and_projection_bottleneck(dev):
    """act = ReLU behind the GroupNorm (detectron2's GroupNorm BottleneckBlock, backbone/feature_extractor.py:20-60) and the projection
    bottleneck that uses it: conv1 (1x1) -> [GN + ReLU fused into the 3x3 conv's staging] -> conv2 -> GN + ReLU -> conv3 (1x1) -> GN + skip -> ReLU"""
    from xmask3d_amd import ops
    from xmask3d_amd.image_branch import GNBottleneck

    g = torch.Generator().manual_seed(21)
    B, C, H, W, G = 2, 128, 16, 64, 32
    x = _nhwc((torch.randn(B, C, H, W, generator=g) * 1.3).to(dev, torch.bfloat16))
    w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev, torch.bfloat16)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    packed, tile = ops.conv3x3_pack_weight(w)
    out = ops.conv3x3(x, packed, C, tile, gn=(ops.gn_stats_of(x, G), gamma, beta, 1e-5, G, "relu"), stats_groups=G)
    ref = F.conv2d(F.relu(F.group_norm(x.float(), G, gamma, beta, 1e-5)).to(torch.bfloat16).float(), w.float(), None, padding=1)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-2, err

    ref_m, fast_m = _module_pair(lambda: GNBottleneck(512, 128, 512), dev)
    xb = torch.randn(2, 512, 32, 32, generator=g).to(dev, torch.bfloat16)
    with torch.no_grad():
        got = fast_m(_nhwc(xb))
        want = ref_m(xb.float())
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 3e-2, err


# every (cin, cout, H, W) the bench forward sends to the kernel (tools: /tmp shape trace of sd_model on meta tensors),
# at the TRUE spatial size, one view: VAE encoder / decoder ResnetBlocks, UNet ResBlocks at 64^2 / 32^2, projection bottleneck
PATH_SHAPES = [(128, 128, 512, 512), (128, 256, 256, 256), (256, 256, 256, 256), (256, 512, 128, 128), (512, 512, 128, 128),
               (512, 512, 64, 64), (640, 640, 64, 64), (320, 640, 32, 32), (640, 640, 32, 32), (1280, 1280, 32, 32),
               (1920, 640, 32, 32), (1280, 640, 32, 32), (960, 640, 32, 32), (128, 128, 128, 128),
               # UNet level 0 (64^2): cout = 320 = 2.5 tiles of 128, the last one half empty
               (320, 320, 64, 64), (640, 320, 64, 64), (960, 320, 64, 64)]


@pytest.mark.parametrize("cin,cout,H,W", PATH_SHAPES)
def test_conv3x3_every_shape_of_the_path_matches_torch_fp32(dev, cin, cout, H, W):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(cin + 3 * cout + H)
    G = 32
    x = _nhwc((torch.randn(1, cin, H, W, generator=g) * 1.5 + 0.2 * torch.randn(1, cin, 1, 1, generator=g)).to(dev, torch.bfloat16))
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev, torch.bfloat16)
    gamma, beta = (1 + 0.2 * torch.randn(cin, generator=g)).to(dev), (0.2 * torch.randn(cin, generator=g)).to(dev)
    bias = (0.3 * torch.randn(cout, generator=g)).to(dev)
    residual = _nhwc(torch.randn(1, cout, H, W, generator=g).to(dev, torch.bfloat16))
    packed, tile = ops.conv3x3_pack_weight(w)
    gs = 32 if (cout // 32) % 4 == 0 else None
    out = ops.conv3x3(x, packed, cout, tile, bias=bias, gn=(ops.gn_stats_of(x, G), gamma, beta, 1e-5, G), residual=residual, stats_groups=gs)
    xn = F.silu(F.group_norm(x.float(), G, gamma, beta, 1e-5)).to(torch.bfloat16).float()
    ref = F.conv2d(xn, w.float(), bias, padding=1) + residual.float()
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-2, err
    if gs:
        st = ops.gn_stats_of(out, gs).view(1, gs, 2)
        o = out.float().view(1, gs, cout // gs, H * W)
        want = torch.stack([o.sum((2, 3)), (o * o).sum((2, 3))], -1).double()
        assert ((st - want).abs() / (want.abs() + 1.0)).max().item() < 1e-4


@pytest.mark.parametrize("waves", [4, 8])
def test_conv3x3_cout_not_a_multiple_of_the_tile(dev, waves):
    """cout = 320 / 160 / 352: zero-padded last tile, nothing is written past the real channels (the output tensor has exactly cout
    channels per pixel, so a stray store would land in the next pixel and show up as an error there)"""
    from xmask3d_amd import ops

    for cin, cout in ((64, 320), (128, 160), (64, 352)):
        g = torch.Generator().manual_seed(cout + waves)
        x = _nhwc(torch.randn(2, cin, 16, 32, generator=g).to(dev, torch.bfloat16))
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev, torch.bfloat16)
        bias = torch.randn(cout, generator=g).to(dev)
        res = _nhwc(torch.randn(2, cout, 16, 32, generator=g).to(dev, torch.bfloat16))
        packed, tile = ops.conv3x3_pack_weight(w)
        assert tile == 128 and packed.numel() == -(-cout // 128) * 128 * 9 * cin
        out = ops.conv3x3(x, packed, cout, tile, bias=bias, residual=res, waves=waves)
        ref = F.conv2d(x.float(), w.float(), bias, padding=1) + res.float()
        assert out.shape == ref.shape and (out.float() - ref).abs().max().item() < 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("waves,terms", [(8, 2), (4, 3), (8, 3), (8, "f16"), (4, "f16")])
@pytest.mark.parametrize("B,cin,cout,H,W,mode", [(2, 128, 128, 16, 64, "gn_res"), (1, 256, 512, 8, 32, "plain"), (2, 64, 256, 16, 64, "ups"),
                                                  (1, 320, 320, 8, 32, "gn_res"), (1, 512, 512, 16, 32, "gn_bias2")])
def test_conv3x3_f32_accurate_form_matches_f64(dev, B, cin, cout, H, W, mode, waves, terms):
    """ops.conv3x3_f32 (the fp32 configuration's convolution: split operands, accumulating matrix-core passes) against an f64 evaluation
    of the same layer.  "f16" (the default): two terms in IEEE halves, three passes, bound 4e-6 - the level of an f32 library
    convolution's own rounding at K = 9 * 512; three bf16 terms (six passes): the same bound; two bf16 terms: 2e-5."""
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(cin + 5 * cout + H + waves)
    G = 32
    hi_, wi_ = (H // 2, W // 2) if mode == "ups" else (H, W)
    x = _nhwc((torch.randn(B, cin, hi_, wi_, generator=g) * 1.3 + 0.4).to(dev))
    if mode == "plain":  # un-normalised operands of very different magnitudes: the half's exponent range must not cost bits
        x = _nhwc(x * torch.logspace(-4, 3, cin, device=dev).view(1, cin, 1, 1))
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
    per_sample = mode == "gn_bias2"
    bias = (0.3 * torch.randn((B, cout) if per_sample else (cout,), generator=g)).to(dev)
    packs, tile = ops.conv3x3_pack_weight_split(w, terms)
    assert ops.conv3x3_f32_supported(x, cout, mode == "ups")
    xd = x.double()
    if mode in ("gn_res", "gn_bias2"):
        gamma, beta = (1 + 0.2 * torch.randn(cin, generator=g)).to(dev), (0.2 * torch.randn(cin, generator=g)).to(dev)
        res = _nhwc(torch.randn(B, cout, H, W, generator=g).to(dev)) if mode == "gn_res" else None
        gs = 32 if (cout // 32) % 4 == 0 else None
        out = ops.conv3x3_f32(x, packs, cout, tile, bias=bias, gn=(ops.gn_stats_of(x, G), gamma, beta, 1e-6, G), residual=res, stats_groups=gs, waves=waves)
        xin = F.silu(F.group_norm(xd, G, gamma.double(), beta.double(), 1e-6))
        ref = F.conv2d(xin, w.double(), None, padding=1) + bias.double().view(-1 if per_sample else 1, cout, 1, 1)
        if res is not None:
            ref = ref + res.double()
        if gs:
            st = ops.gn_stats_of(out, gs).view(B, gs, 2)
            o = out.double().view(B, gs, cout // gs, H * W)
            want = torch.stack([o.sum((2, 3)), (o * o).sum((2, 3))], -1)
            assert ((st - want).abs() / (want.abs() + 1.0)).max().item() < 1e-5
    else:
        out = ops.conv3x3_f32(x, packs, cout, tile, bias=bias, upsample=mode == "ups", waves=waves)
        xin = F.interpolate(xd, scale_factor=2.0, mode="nearest") if mode == "ups" else xd
        ref = F.conv2d(xin, w.double(), bias.double(), padding=1)
    assert out.dtype == torch.float32 and out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    # two terms: dropped lo*lo and split residuals, 2^-18 each; three terms: the f32 accumulation of the partial sums is what is left
    assert err < (2e-5 if terms == 2 else 4e-6), err  # measured 3e-7 .. 1.1e-6 (K = 4608); an f32 library convolution: 1e-6 .. 5e-6
    from xmask3d_amd._lib import lib

    assert lib().xm3d_check_flag() == 0  # no operand left the half's range


def test_f16_split_flags_operands_beyond_the_half_range(dev):
    from xmask3d_amd import ops
    from xmask3d_amd._lib import lib

    x = _nhwc(torch.randn(1, 64, 8, 32, device=dev))
    x[0, 3, 2, 5] = 1e7   # * 2^-6 > 65504
    w = torch.randn(128, 64, 3, 3, device=dev) / 24
    packs, tile = ops.conv3x3_pack_weight_split(w, "f16")
    assert lib().xm3d_check_flag() == 0
    ops.conv3x3_f32(x, packs, 128, tile)
    torch.cuda.synchronize()
    assert lib().xm3d_check_flag() != 0   # sticky range flag raised (and cleared by the query)
    assert lib().xm3d_check_flag() == 0
