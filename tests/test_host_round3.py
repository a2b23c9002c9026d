"""Host-side logic of the round-3 operator wrappers, on the CPU (no compute calls): which tensors the HIP paths accept, how row
strides are recognised, which module paths are taken without a device, the switches.  The kernels themselves are covered by the
`-m gpu` tests (tests/test_gpu_gemm.py, test_gpu_maskhead.py, test_gpu_pointclass.py, test_gpu_conv.py)."""
import torch

from xmask3d_amd import mask_head, ops, sd_model


def test_rows_of_recognises_evenly_strided_row_stacks():
    x = torch.zeros(2, 10, 64, dtype=torch.bfloat16)
    assert ops._rows_of(x) == (20, 64)
    assert ops._rows_of(x[..., :32]) == (20, 64)                       # a column slice keeps the row stride
    wide = torch.zeros(20, 192, dtype=torch.bfloat16)
    assert ops._rows_of(wide[:, 64:128]) == (20, 192)
    assert ops._rows_of(x[:, ::2]) == (10, 128)                        # every other row: still one constant stride
    assert ops._rows_of(x.transpose(1, 2)) is None                     # element stride != 1
    assert ops._rows_of(torch.zeros(3, 4, 68, dtype=torch.bfloat16)[..., :64]) is None  # row stride not a multiple of 8
    ragged = torch.zeros(2, 10, 64, dtype=torch.bfloat16)[:, :7]       # batch stride 640 != 7 * 64
    assert ops._rows_of(ragged) is None


def test_hip_paths_refuse_cpu_tensors_and_autograd():
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    assert not ops.gemm_supported(x, 64)
    assert not sd_model.gemm_ok(x, 64)
    feat = torch.zeros(1, 256, 32, 32, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert not ops.mask_heads_supported(torch.zeros(1, 50, 256), feat, (16, 16))        # not on the device
        assert not ops.point_class_supported(torch.zeros(8, 768), torch.zeros(19, 768))
    assert not ops.conv3x3_supported(feat, 256) and not ops.conv3x3_f32_supported(feat.float(), 256)  # is_nhwc wants a device tensor
    conv = torch.nn.Conv2d(256, 256, 3, padding=1)
    assert not sd_model.fused_conv_ok(feat, conv)


def test_f32_convolution_switch(monkeypatch):
    monkeypatch.delenv("XM3D_CONV_F32", raising=False)
    assert sd_model.conv_f32_terms() == "f16"      # default: the two-term split in IEEE halves, three passes, f32-exact to ~1e-6
    for val, terms in (("f16", "f16"), ("hip", 2), ("hip3", 3), ("library", 0), ("something", 0)):
        monkeypatch.setenv("XM3D_CONV_F32", val)
        assert sd_model.conv_f32_terms() == terms


def test_modules_take_the_torch_branch_on_the_cpu():
    """the CPU oracle runs these modules: the fast paths must not trigger and the results must be the plain torch ones"""
    torch.manual_seed(0)
    blk = sd_model.SpatialTransformer(64, 2, 32, context_dim=48).eval()
    x, ctx = torch.randn(1, 64, 8, 8), torch.randn(1, 5, 48)
    with torch.no_grad():
        y = blk(x, ctx)
        t = blk.transformer_blocks[0]
        tok = blk.proj_in(blk.norm(x)).flatten(2).transpose(1, 2)
        tok = t.attn1(t.norm1(tok)) + tok
        tok = t.attn2(t.norm2(tok), ctx) + tok
        tok = t.ff(t.norm3(tok)) + tok
        ref = x + blk.proj_out(tok.transpose(1, 2).reshape(1, 64, 8, 8))
    assert torch.allclose(y, ref, atol=1e-5)
    pool = mask_head.MaskPooling()
    feat, logits = torch.randn(1, 256, 4, 4), torch.randn(1, 3, 4, 4)
    m = (logits.sigmoid() > 0.5).float()
    want = torch.einsum("bchw,bqhw->bqc", feat, m / (m.sum((-1, -2), keepdim=True) + 1e-8))
    assert torch.allclose(pool(feat, logits)["mask_pooled_features"], want, atol=1e-6)


def test_feed_forward_residual_argument_equals_the_add():
    torch.manual_seed(1)
    ff = sd_model.FeedForward(32).eval()
    x, r = torch.randn(2, 7, 32), torch.randn(2, 7, 32)
    with torch.no_grad():
        assert torch.allclose(ff(x, residual=r), ff(x) + r, atol=1e-6)
