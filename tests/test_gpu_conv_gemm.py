"""GPU parity of the implicit-GEMM convolution (xm3d_conv_gemm_bf16, csrc/gemm.hip GF_CONV): the strided Downsample convolutions,
the 3x3 convolutions of the 16^2 / 8^2 UNet levels and large-K 1x1 convolutions of the frozen SD nets
(/root/reference/models/modeling/meta_arch/ldm.py:386-490 -> torch.nn.Conv2d).  References: an fp32 torch convolution of the same
bf16-rounded operands (tolerance = bf16 output rounding + f32 summation order), and a BIT-EXACT comparison on small-integer data, where
every product and partial sum is exact - that catches tap-order / padding / stride / split-K slips.  Every case is also run twice:
the kernel (incl. its split-K) has no atomics, so the two results must be identical bits."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # name,                     B, cin, H,  W,  cout, k, stride, pad (t, l, b, r)
    ("vae_down_128",            2, 128, 64, 64, 128, 3, 2, (0, 0, 1, 1)),
    ("vae_down_512",            3, 512, 32, 32, 512, 3, 2, (0, 0, 1, 1)),
    ("unet_op_320",             2, 320, 32, 32, 320, 3, 2, (1, 1, 1, 1)),
    ("unet_op_1280_splitk",     3, 1280, 16, 16, 1280, 3, 2, (1, 1, 1, 1)),
    ("unet_res_1280_8x8",       3, 1280, 8, 8, 1280, 3, 1, (1, 1, 1, 1)),
    ("unet_res_2560_16x16",     2, 2560, 16, 16, 1280, 3, 1, (1, 1, 1, 1)),
    ("proj_in_1x1_1280",        3, 1280, 16, 16, 1280, 1, 1, (0, 0, 0, 0)),
    ("skip_1x1_1920_640",       2, 1920, 32, 32, 640, 1, 1, (0, 0, 0, 0)),
    ("vae_conv_out_8",          2, 512, 32, 32, 8, 3, 1, (1, 1, 1, 1)),
    ("ragged_rows",             1, 64, 10, 14, 96, 3, 1, (1, 1, 1, 1)),
    ("two_by_two_kernel",       2, 128, 12, 12, 64, 2, 2, (0, 0, 0, 0)),
]


def _run(dev, B, cin, H, W, cout, k, stride, pad, integer=False, with_res=False, seed=0):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(seed)
    if integer:
        x = torch.randint(-3, 4, (B, cin, H, W), generator=g).float()
        w = (torch.randint(-1, 2, (cout, cin, k, k), generator=g) * (torch.rand(cout, cin, k, k, generator=g) < min(0.2, 64.0 / (cin * k * k)))).float()
        bias = torch.randint(-4, 5, (cout,), generator=g).float()
    else:
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        bias = torch.randn(cout, generator=g)
    xb = x.to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wb = w.to(dev, torch.bfloat16)
    packed, tile, n32 = ops.conv_gemm_pack_weight(wb)
    bpad = torch.zeros(n32, device=dev)
    bpad[:cout] = bias.to(dev)
    pt, pl, pb, pr = pad
    ref = F.conv2d(F.pad(xb.float(), (pl, pr, pt, pb)), wb.float(), bias.to(dev), stride=stride)
    res = None
    if with_res and n32 == cout:
        res = (torch.randint(-2, 3, ref.shape, generator=g).float() if integer else torch.randn(ref.shape, generator=g)).to(dev, torch.bfloat16) \
            .contiguous(memory_format=torch.channels_last)
        ref = ref + res.float()
    outs = [ops.conv_gemm(xb, packed, tile, n32, cout, k, stride, pad, bias=bpad, residual=res) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])  # bit-reproducible
    assert outs[0].shape == ref.shape and ops.is_nhwc(outs[0]) or outs[0].shape[2] == 1
    return outs[0].float(), ref


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_gemm_matches_fp32_convolution(dev, case):
    out, ref = _run(dev, *case[1:], with_res=True)
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < 1.2e-2, err  # bf16 output rounding (2^-8 relative) + f32 summation order


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_gemm_is_exact_on_integer_data(dev, case):
    out, ref = _run(dev, *case[1:], integer=True, with_res=True, seed=1)
    assert float(ref.abs().max()) < 256  # exactly representable in bf16
    assert torch.equal(out, ref)


def test_conv_gemm_workspace_query_and_argument_checks(dev):
    from xmask3d_amd import ops
    from xmask3d_amd._lib import Xm3dError, lib

    assert lib().xm3d_conv_gemm_ws_bytes(20 * 512 * 512, 128, 1152, 128) == 0            # large grid: no split
    assert lib().xm3d_conv_gemm_ws_bytes(3 * 64, 1280, 11520, 256) > 0                    # 8^2 level: split-K slabs
    x = torch.zeros(1, 96, 8, 8, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(64, 96, 3, 3, device=dev, dtype=torch.bfloat16)
    with pytest.raises((TypeError, Xm3dError)):
        ops.conv_gemm(x, *ops.conv_gemm_pack_weight(w), 64, 3, 1, (1, 1, 1, 1))          # cin % 64 != 0
