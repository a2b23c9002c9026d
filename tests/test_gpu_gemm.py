"""xm3d_gemm_bf16 (csrc/gemm.hip) against an f32 torch reference of the same bf16-rounded operands: the linear layers / 1x1
convolutions of ldm's SpatialTransformer (models/modeling/meta_arch/ldm.py:425-446), the VAE AttnBlock and the mask-CLIP ViT
(models/modeling/meta_arch/clip.py:239-270).  Tolerance: the output is rounded to bf16 (2^-9 relative) after an f32 accumulation whose
order differs from the reference's -> |diff| <= 1e-2 * max|ref| per tensor, mean |diff| <= 2e-3 * mean|ref|."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, bias, act, residual):
    y = x.float() @ w.to(torch.bfloat16).float().t()
    if bias is not None:
        y = y + bias
    if act == "gelu":
        y = F.gelu(y)
    elif act == "quick_gelu":
        y = y * torch.sigmoid(1.702 * y)
    elif act == "geglu":
        v, g = y.chunk(2, dim=-1)
        y = v * F.gelu(g)
    elif act == "relu":
        y = F.relu(y)
    if residual is not None:
        y = y + residual.float()
    return y


def _check(out, ref):
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    d = (out.float() - ref).abs()
    assert float(d.max()) <= 1e-2 * float(ref.abs().max()) + 1e-3, (float(d.max()), float(ref.abs().max()))
    assert float(d.mean()) <= 2e-3 * float(ref.abs().mean()) + 1e-4, (float(d.mean()), float(ref.abs().mean()))


# (M, K, N rows of W, act, bias, residual): every (K, N) pair of the bench forward's transformer blocks + CLIP + VAE attention
SHAPES = [
    (4096, 320, 320, None, False, False),        # UNet level 0 to_q / to_out (N padded 320 -> 384, half-empty last tile)
    (4096 + 37, 320, 320, None, True, True),     # ragged M, bias + residual (to_out, proj_out)
    (1000, 320, 2560, "geglu", True, False),     # GEGLU 320 -> 2 x 1280
    (1000, 1280, 320, None, True, True),         # feed-forward out + residual
    (77 * 3, 768, 320, None, False, False),      # to_k / to_v of the cross attention (context 768)
    (2048, 640, 640, None, True, False),
    (1024, 640, 5120, "geglu", True, True),
    (512, 1280, 1280, None, False, False),
    (300, 1280, 10240, "geglu", True, False),
    (257 * 3, 1024, 4096, "quick_gelu", True, False),  # CLIP ViT-L c_fc + QuickGELU
    (257 * 3, 4096, 1024, None, True, True),           # CLIP c_proj + residual
    (257 * 3, 1024, 3072, None, True, False),          # CLIP in_proj
    (640, 512, 512, None, True, False),                # VAE AttnBlock q / k / v (1x1 conv)
    (333, 256, 2048, "gelu", True, False),             # exact-GELU epilogue
    (1, 64, 32, None, True, False),                    # smallest
    (255, 64, 96, None, False, True),
    (513, 960, 960, None, True, False),                # 960: 7.5 column tiles of 128
    (5376, 256, 1024, "relu", True, False),            # deformable-attention encoder FFN: ReLU in the epilogue
    (1000, 256, 2048, "relu", True, True),             # transformer-decoder FFN; residual added AFTER the ReLU
]


@pytest.mark.parametrize("M,K,N,act,has_bias,has_res", SHAPES)
def test_gemm_matches_reference(M, K, N, act, has_bias, has_res):
    from xmask3d_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(M * 7 + K + N)
    x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev) if has_bias else None
    nout = N // 2 if act == "geglu" else N
    res = torch.randn(M, nout, generator=g).to(dev, torch.bfloat16) if has_res else None
    packed, tile = ops.gemm_pack_weight(w, act)
    ref = _ref(x, w, bias, act, res)
    out = ops.gemm(x, packed, N, tile, bias=bias, act=act, residual=res)
    _check(out, ref)
    # both workgroup geometries give the same result (same products, same summation order per output)
    o8, o4 = (ops.gemm(x, packed, N, tile, bias=bias, act=act, residual=res, waves=w_) for w_ in (8, 4))
    _check(o8, ref)
    assert torch.equal(o8, o4)


def test_gemm_bf16_weight_strided_rows_and_batch_dims():
    """bf16 weights pack to the same image as their f32 copy; x may be a column slice of a wider tensor (q / k / v out of one fused
    projection) and carry batch dimensions; residual may be strided as well"""
    from xmask3d_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    big = torch.randn(3, 130, 3 * 128, generator=g).to(dev, torch.bfloat16)
    x = big[..., 128:256]
    w = (torch.randn(160, 128, generator=g) / 11).to(dev)
    p32, tile = ops.gemm_pack_weight(w)
    p16, _ = ops.gemm_pack_weight(w.to(torch.bfloat16))
    assert torch.equal(p32, p16)
    resbig = torch.randn(3, 130, 320, generator=g).to(dev, torch.bfloat16)
    res = resbig[..., :160]
    out = ops.gemm(x, p32, 160, tile, residual=res)
    assert out.shape == (3, 130, 160)
    _check(out, _ref(x, w, None, None, res))


def test_gemm_rejects_bad_arguments():
    from xmask3d_amd import ops

    dev = torch.device("cuda:0")
    w = torch.randn(64, 64, device=dev)
    packed, tile = ops.gemm_pack_weight(w)
    with pytest.raises(TypeError):
        ops.gemm(torch.randn(8, 64, device=dev), packed, 64, tile)  # f32 rows
    with pytest.raises(TypeError):
        ops.gemm(torch.randn(8, 128, device=dev).bfloat16()[:, ::2], packed, 64, tile)  # element stride 2
    with pytest.raises(TypeError):
        ops.gemm_pack_weight(torch.randn(48, 64, device=dev))  # N % 32
    with pytest.raises(TypeError):
        ops.gemm_pack_weight(torch.randn(64, 80, device=dev))  # K % 64
    with pytest.raises(TypeError):
        ops.gemm(torch.randn(8, 64, device=dev).bfloat16(), packed, 64, tile, bias=torch.zeros(32, device=dev))


@pytest.mark.parametrize("c,heads,hw", [(320, 8, 32), (640, 8, 16)])
def test_spatial_transformer_on_hip_gemm_matches_library_projections(c, heads, hw, monkeypatch):
    """ldm's SpatialTransformer (models/modeling/meta_arch/ldm.py:425-446) with its projections on k_gemm (fused q/k/v, GEGLU and
    residual epilogues, 1x1 convolutions as token GEMMs) against the same module with every projection on torch: same bf16 weights,
    differences = accumulation order and the points at which intermediates are rounded to bf16"""
    from xmask3d_amd import sd_model

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    blk = sd_model.SpatialTransformer(c, heads, c // heads, context_dim=768).to(dev, torch.bfloat16).eval()
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(2, c, hw, hw, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ctx = torch.randn(2, 77, 768, generator=g).to(dev, torch.bfloat16)
    with torch.no_grad():
        assert sd_model.gemm_ok(sd_model.tokens_of(x), c)
        own = blk(x, ctx)
        monkeypatch.setattr(sd_model, "_GEMM_LIBRARY", True)
        assert not sd_model.gemm_ok(sd_model.tokens_of(x), c)
        lib = blk(x, ctx)
    assert own.shape == lib.shape and own.is_contiguous(memory_format=torch.channels_last)
    d = (own.float() - lib.float()).abs()
    assert float(d.max()) <= 4e-2 * float(lib.float().abs().max()), (float(d.max()), float(lib.float().abs().max()))
    assert float(d.mean()) <= 4e-3 * float(lib.float().abs().mean()) + 1e-4
