"""GPU parity of the f32-accurate GEMM / implicit-GEMM convolution (xm3d_gemm_f32acc: three matrix-core passes over operands split in IEEE
halves) - the fp32 configuration's Linear / 1x1 / strided / small-map convolutions, torch.nn.Linear / Conv2d in f32 in the reference
(run/train.py:178: no autocast; models/modeling/meta_arch/clip.py:239-270, ldm.py:386-490).  Reference: an f64 evaluation of the same layer;
bound 4e-6 of max|out| - the level of an f32 library GEMM's own rounding (measured beside it).  Every case runs twice: identical bits."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("M,K,N,act,res", [(5140, 1024, 4096, "quick_gelu", False), (5140, 4096, 1024, None, True), (1000, 320, 960, None, False),
                                           (77 * 3, 768, 320, None, False), (4096, 1280, 1280, "gelu", True), (300, 256, 2048, None, False),
                                           (130, 64, 32, None, True)])
def test_gemm_f32_matches_f64(dev, M, K, N, act, res):
    from xmask3d_amd import ops
    from xmask3d_amd._lib import lib

    g = torch.Generator().manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g) * torch.logspace(-3, 2, K).view(1, K)).to(dev)  # columns of very different magnitudes
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if res else None
    packs, tile, n32 = ops.gemm_pack_weight_f16(w)
    assert n32 == N
    outs = [ops.gemm_f32(x, packs, N, tile, bias=bias, act=act, residual=r) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])
    ref = x.double() @ w.double().t() + bias.double()
    if act == "gelu":
        ref = F.gelu(ref)
    elif act == "quick_gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    if res:
        ref = ref + r.double()
    lib_err = _rel(F.linear(x, w, bias) if act is None and not res else outs[0], ref)
    err = _rel(outs[0], ref)
    assert err < 4e-6, (err, lib_err)
    # the one-launch form (both planes at one scale, three MFMAs per k-step into one accumulator): same bound
    packs1, tile1, n1, sw = ops.gemm_pack_weight_f16(w, one_scale=True)
    fused = [ops.gemm_f32_fused(x, packs1, N, sw, bias=bias, act=act, residual=r) for _ in range(2)]
    assert torch.equal(fused[0], fused[1]) and tile1 == 128 and n1 == N
    assert _rel(fused[0], ref) < 4e-6, _rel(fused[0], ref)
    assert lib().xm3d_check_flag() == 0


@pytest.mark.parametrize("B,cin,H,W,cout,k,stride,pad", [(2, 128, 64, 64, 128, 3, 2, (0, 0, 1, 1)), (2, 320, 32, 32, 320, 3, 2, (1, 1, 1, 1)),
                                                         (3, 1280, 8, 8, 1280, 3, 1, (1, 1, 1, 1)), (2, 512, 32, 32, 512, 1, 1, (0, 0, 0, 0)),
                                                         (2, 512, 16, 16, 8, 3, 1, (1, 1, 1, 1))])
def test_conv_gemm_f32_matches_f64(dev, B, cin, H, W, cout, k, stride, pad):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(cin + cout + H)
    x = (torch.randn(B, cin, H, W, generator=g) * 1.5 + 0.3).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    packs, tile, n32 = ops.gemm_pack_weight_f16(w)
    bpad = torch.zeros(n32, device=dev)
    bpad[:cout] = bias
    pt, pl, pb, pr = pad
    ref = F.conv2d(F.pad(x.double(), (pl, pr, pt, pb)), w.double(), bias.double(), stride=stride)
    res = torch.randn(ref.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last) if n32 == cout else None
    if res is not None:
        ref = ref + res.double()
    outs = [ops.conv_gemm_f32(x, packs, tile, n32, cout, k, stride, pad, bias=bpad, residual=res) for _ in range(2)]
    assert torch.equal(outs[0], outs[1]) and outs[0].dtype == torch.float32 and outs[0].shape == ref.shape
    assert _rel(outs[0], ref) < 4e-6
    packs1, _, n1, sw = ops.gemm_pack_weight_f16(w, one_scale=True)
    fused = [ops.conv_gemm_f32_fused(x, packs1, n1, sw, cout, k, stride, pad, bias=bpad, residual=res) for _ in range(2)]
    assert torch.equal(fused[0], fused[1]) and fused[0].shape == ref.shape
    assert _rel(fused[0], ref) < 4e-6, _rel(fused[0], ref)


def test_one_scale_split_flags_operands_beyond_the_half_range(dev):
    from xmask3d_amd import ops
    from xmask3d_amd._lib import lib

    w = torch.randn(64, 128, device=dev) / 11
    packs, _, n, sw = ops.gemm_pack_weight_f16(w, one_scale=True)
    x = torch.randn(300, 128, device=dev)
    assert lib().xm3d_check_flag() == 0
    ops.gemm_f32_fused(x, packs, n, sw)
    torch.cuda.synchronize()
    assert lib().xm3d_check_flag() == 0
    x[7, 3] = 5000.0  # * 16 > 65504
    ops.gemm_f32_fused(x, packs, n, sw)
    torch.cuda.synchronize()
    assert lib().xm3d_check_flag() != 0 and lib().xm3d_check_flag() == 0


def test_in_kernel_operand_split_equals_the_split_pass_bit_for_bit(dev, monkeypatch):
    """xm3d_gemm_f32x (the f32 activation split into its half planes while it is staged) against xm3d_split_f16t_nhwc + xm3d_gemm_f32: the same
    bits, GEMM (with epilogues) and convolution form (strided, zero padding, 1x1), incl. rows past M and inputs in the denormal-lo range"""
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(11)
    for M, K, N, act, res in [(5140, 1024, 1024, "quick_gelu", False), (1000, 320, 960, None, True), (130, 64, 32, "gelu", True), (257, 2048, 256, "relu", False)]:
        x = (torch.randn(M, K, generator=g) * torch.logspace(-6, 2, K).view(1, K)).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev) if res else None
        packs, _, n, sw = ops.gemm_pack_weight_f16(w, one_scale=True)
        monkeypatch.setattr(ops, "_F32_SPLIT_IN_KERNEL", True)
        a = ops.gemm_f32_fused(x, packs, n, sw, bias=bias, act=act, residual=r)
        monkeypatch.setattr(ops, "_F32_SPLIT_IN_KERNEL", False)
        b = ops.gemm_f32_fused(x, packs, n, sw, bias=bias, act=act, residual=r)
        assert torch.equal(a, b), (M, K, N, act, float((a - b).abs().max()))
    for B, cin, H, W, cout, k, stride, pad in [(2, 128, 64, 64, 128, 3, 2, (0, 0, 1, 1)), (3, 1280, 8, 8, 1280, 3, 1, (1, 1, 1, 1)),
                                               (2, 512, 32, 32, 512, 1, 1, (0, 0, 0, 0)), (2, 512, 16, 16, 8, 3, 1, (1, 1, 1, 1))]:
        x = (torch.randn(B, cin, H, W, generator=g) * 1.5 + 0.3).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        packs, _, n32, sw = ops.gemm_pack_weight_f16(w, one_scale=True)
        bpad = torch.randn(n32, generator=g).to(dev)
        monkeypatch.setattr(ops, "_F32_SPLIT_IN_KERNEL", True)
        a = ops.conv_gemm_f32_fused(x, packs, n32, sw, cout, k, stride, pad, bias=bpad)
        monkeypatch.setattr(ops, "_F32_SPLIT_IN_KERNEL", False)
        b = ops.conv_gemm_f32_fused(x, packs, n32, sw, cout, k, stride, pad, bias=bpad)
        assert torch.equal(a, b), (B, cin, H, W, cout, k, stride)
