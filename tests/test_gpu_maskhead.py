"""Fused prediction heads (csrc/maskhead.hip) against the op chain they replace - forward_prediction_heads + mask handling +
MaskPooling of the masked transformer decoder (/root/reference/models/modeling/meta_arch/odise.py:395,445-491,509-547):
  * logits: einsum("bqc,bchw->bqhw") of the same bf16 operands in f32 (tolerance: bf16 rounding of the stored logit);
  * attention bias: xm3d_attn_mask_bias (itself bit-identical to the torch chain, tests/test_gpu_msda_fuse.py) applied to THE
    KERNEL'S OWN logits - bit-equal, including the "empty mask attends everywhere" rule, for every shrink factor of the path;
    and with logits not requested (pruned layers) the bias is the same;
  * pooling: f32 torch evaluation of sum(mask * features) / (count + 1e-8) with the hard mask taken from the same logits."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():  # the fused heads are the inference path (ops.mask_heads_supported refuses under autograd)
        yield


def _inputs(B, Q, H, W, seed, dead_query=None):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    feat = torch.randn(B, 256, H, W, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    emb = (torch.randn(B, Q, 256, generator=g) / 16).to(dev)
    if dead_query is not None:  # a query whose logits are negative everywhere: mean feature direction, flipped, scaled up
        b, q = dead_query
        emb[b, q] = 0
        feat[b, 0] = feat[b, 0].abs() + 1
        emb[b, q, 0] = -1.0
    return emb, feat


@pytest.mark.parametrize("B,Q,H,W,size", [(2, 50, 128, 128, (16, 16)), (2, 50, 128, 128, (32, 32)), (3, 50, 128, 128, (64, 64)),
                                          (1, 64, 64, 96, (16, 24)), (2, 7, 32, 64, (8, 16))])
def test_logits_and_bias(B, Q, H, W, size):
    from xmask3d_amd import ops

    emb, feat = _inputs(B, Q, H, W, seed=H + W + size[0], dead_query=(B - 1, min(3, Q - 1)))
    assert ops.mask_heads_supported(emb, feat, size)
    logits, bias = ops.mask_logits_bias(emb, feat, size, want_logits=True, bias_dtype=torch.bfloat16)
    ref = torch.einsum("bqc,bchw->bqhw", emb.to(torch.bfloat16).float(), feat.float())
    d = (logits.float() - ref).abs()
    assert float(d.max()) <= 2 ** -8 * float(ref.abs().max()) + 1e-3, float(d.max())
    # bias == the stand-alone kernel on the same logits, bit for bit; the dead query's map is all zeros (attends everywhere)
    want = ops.attn_mask_bias(logits, size, torch.bfloat16)
    assert torch.equal(bias, want)
    assert float(bias[B - 1, min(3, Q - 1)].abs().max()) == 0.0
    assert bool(torch.isinf(bias).any())
    # f32 bias, logits not wanted: same mask
    none, bias32 = ops.mask_logits_bias(emb, feat, size, want_logits=False, bias_dtype=torch.float32)
    assert none is None and bias32.dtype == torch.float32 and torch.equal(bias32, want.float())


@pytest.mark.parametrize("B,Q,H,W", [(2, 50, 128, 128), (1, 64, 32, 48), (3, 5, 16, 16)])
def test_mask_pool(B, Q, H, W):
    from xmask3d_amd import ops

    emb, feat = _inputs(B, Q, H, W, seed=B + Q, dead_query=(0, 1))
    logits, _ = ops.mask_logits_bias(emb, feat, (H // 2, W // 2), want_logits=True) if ops.mask_heads_supported(emb, feat, (H // 2, W // 2)) \
        else (torch.einsum("bqc,bchw->bqhw", emb.to(torch.bfloat16), feat).contiguous(), None)
    pooled = ops.mask_pool(logits, feat)
    m = (logits.float() > 0).float()
    ref = torch.einsum("bchw,bqhw->bqc", feat.float(), m) / (m.sum(dim=(-1, -2)).unsqueeze(-1) + 1e-8)
    assert pooled.dtype == torch.float32 and pooled.shape == (B, Q, 256)
    assert float((pooled - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert float(pooled[0, 1].abs().max()) == 0.0  # empty mask -> zeros


def test_decoder_heads_use_the_fused_path_and_match_the_op_chain(monkeypatch):
    """MaskPooling through the kernel == the module's own torch chain on the same bf16 logits (mean of a hard mask; the chain rounds
    mask / count to bf16, the kernel divides in f32: compare against the f32 evaluation, and loosely against the chain)"""
    from xmask3d_amd import mask_head

    emb, feat = _inputs(2, 50, 64, 64, seed=9)
    logits = torch.einsum("bqc,bchw->bqhw", emb.to(torch.bfloat16), feat).contiguous()
    pool = mask_head.MaskPooling()
    with torch.no_grad():
        own = pool(feat, logits)["mask_pooled_features"]
        with torch.enable_grad():  # the torch branch
            chain = pool(feat, logits.clone())["mask_pooled_features"]
    assert own.dtype == torch.float32
    # the chain's own error (bf16 mask / count, bf16 product) on these zero-mean features is ~5 % of the largest mean: measured 0.009 of 0.205
    assert float((own - chain.float()).abs().max()) <= 5e-2 * float(chain.float().abs().max()) + 5e-3


@pytest.mark.parametrize("B,Q,h,w", [(2, 50, 128, 128), (1, 7, 64, 96), (3, 5, 32, 32)])
def test_clip_mask_blocked_matches_the_torch_chain(B, Q, h, w):
    """xm3d_clip_mask_blocked == (max_pool2d(sigmoid(interpolate(logits, 224, bilinear)), 14, 14) < 0.5) of mask-CLIP (meta_arch/clip.py:272-310).
    The kernel evaluates torch's bilinear arithmetic term for term; a product fused differently by the compiler can move a value that sits within
    one ulp of the threshold, so the comparison allows 1e-4 of the patches to differ (measured: 0)."""
    import torch.nn.functional as F
    from xmask3d_amd import ops

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(B * Q + h)
    base = torch.randn(B, Q, h // 8, w // 8, generator=g)
    logits = (F.interpolate(base, size=(h, w), mode="bicubic") * 3 + torch.randn(B, Q, h, w, generator=g) * 0.3).to(dev).contiguous()
    want = (F.max_pool2d(F.interpolate(logits, size=(224, 224), mode="bilinear", align_corners=False).sigmoid(), 14, 14) < 0.5).reshape(B, Q, -1)
    got = ops.clip_mask_blocked(logits, 224, 14)
    assert got.shape == want.shape and got.dtype == torch.bool
    assert (got != want).float().mean().item() <= 1e-4
    assert 0.05 < want.float().mean().item() < 0.95  # both classes present: the test means something
