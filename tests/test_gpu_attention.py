"""GPU parity of the HIP flash-attention kernel (xm3d_attention_fwd) against an fp32 softmax(q k^T / sqrt(d) + bias) v torch
reference at every shape of the path (SD UNet self / cross attention, mask-CLIP ViT-L with its additive mask, the masked
cross-attention of the Mask2Former decoder), on strided views as the call sites pass them.  bf16 in / bf16 out: the bound is
the bf16 output rounding plus the bf16 rounding of P in the second product, 2e-2 of max|out| (measured ~4e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, bias, scale):
    q, k, v = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))  # (B, H, N, D)
    s = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias.float()
    return torch.softmax(s, -1).nan_to_num(0.0) @ v


def _check(out, ref, tol=2e-2):
    ref = ref.permute(0, 2, 1, 3)
    err = (out.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-20)
    assert err < tol, err
    return err


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (3, 8, 256, 256, 160), (3, 8, 64, 64, 160),
                                         (2, 8, 4096, 77, 40), (2, 8, 1024, 77, 80), (2, 8, 256, 77, 160), (2, 16, 307, 307, 64),
                                         (2, 8, 50, 256, 32), (2, 8, 50, 4096, 32), (1, 2, 33, 65, 8), (1, 1, 1, 1, 16),
                                         (1, 3, 130, 191, 96), (1, 2, 95, 64, 128)])
def test_attention_matches_fp32_reference(dev, B, H, Nq, Nk, D):
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk + D)
    # projections as the call sites hold them: (B, N, H*D) buffers viewed as (B, N, H, D)
    q = torch.randn(B, Nq, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nq, H, D)
    k = torch.randn(B, Nk, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nk, H, D)
    v = torch.randn(B, Nk, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nk, H, D)
    assert ops.attention_supported(q, k, v)
    out = ops.attention(q, k, v)
    assert out.shape == (B, Nq, H, D) and out.dtype == torch.bfloat16
    _check(out, _ref(q, k, v, None, D ** -0.5))
    # a sharp row: one key dominating (exercises the running-maximum rescale across tiles)
    if Nk > 64:
        k2 = k.clone()
        k2[:, Nk - 3] = q[:, 0:1].expand(-1, 1, -1, -1).reshape(B, H, D) * 4
        _check(ops.attention(q, k2, v), _ref(q, k2, v, None, D ** -0.5))


def test_packed_qkv_and_additive_mask_like_mask_clip(dev):
    """q / k / v as strided views of one packed (B, T, 3, H, D) buffer, additive bf16 mask (B, 1, T, T) with -inf entries"""
    from xmask3d_amd import ops
    from xmask3d_amd.clip_model import additive_mask

    B, T, H, D, Q = 3, 307, 16, 64, 50
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(B, T, 3, H, D, generator=g).to(dev, torch.bfloat16)
    allow = torch.ones(B, T, T, dtype=torch.bool)
    allow[:, :, :Q] = False
    allow[:, :Q, Q + 1:] = torch.rand(B, Q, T - Q - 1, generator=g) < 0.3
    bias = additive_mask(allow[:, None].to(dev), torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    assert ops.attention_supported(q, k, v, bias)
    _check(ops.attention(q, k, v, bias=bias), _ref(q, k, v, bias, D ** -0.5))


def test_sequence_first_layout_shared_bias_and_output_view(dev):
    """(L, B, E) projections of nn.MultiheadAttention viewed as (B, L, H, d); f32 bias (B, 1, Q, K) shared by the heads with a
    fully masked row (-> zeros, no NaN); output written through a view of an (Lq, B, E) buffer"""
    from xmask3d_amd import ops

    Lq, Lk, B, H, d = 50, 1024, 4, 8, 32
    E = H * d
    g = torch.Generator().manual_seed(11)
    q = torch.randn(Lq, B, E, generator=g).to(dev, torch.bfloat16)
    k = torch.randn(Lk, B, E, generator=g).to(dev, torch.bfloat16)
    v = torch.randn(Lk, B, E, generator=g).to(dev, torch.bfloat16)
    bias = torch.zeros(B, Lq, Lk).masked_fill_(torch.rand(B, Lq, Lk, generator=g) < 0.6, float("-inf"))
    bias[1, 7] = float("-inf")  # one fully masked row
    bias = bias.to(dev)
    q4, k4, v4 = (t.view(t.shape[0], B, H, d).transpose(0, 1) for t in (q, k, v))
    o = torch.empty(Lq, B, E, dtype=torch.bfloat16, device=dev)
    ops.attention(q4, k4, v4, bias=bias.view(B, 1, Lq, Lk), out=o.view(Lq, B, H, d).transpose(0, 1))
    ref = _ref(q4, k4, v4, bias.view(B, 1, Lq, Lk), d ** -0.5).permute(0, 2, 1, 3)  # (B, Lq, H, d)
    got = o.view(Lq, B, H, d).transpose(0, 1).float()
    assert torch.isfinite(got).all() and float(got[1, 7].abs().max()) == 0.0
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 2e-2


def test_unsupported_inputs_are_refused(dev):
    from xmask3d_amd import ops

    q = torch.randn(1, 8, 2, 40, device=dev)
    assert not ops.attention_supported(q, q, q)                                # f32
    qb = q.bfloat16()
    assert ops.attention_supported(qb, qb, qb)
    assert not ops.attention_supported(qb.cpu(), qb.cpu(), qb.cpu())          # CPU tensors stay on the library path
    qd = torch.randn(1, 8, 2, 512, device=dev).bfloat16()
    assert not ops.attention_supported(qd, qd, qd)                            # VAE single head of 512 channels: library path
    with torch.enable_grad():
        qg = qb.clone().requires_grad_(True)
        assert not ops.attention_supported(qg, qb, qb)                        # training: autograd path


def test_softmax_rows_matches_torch(dev):
    from xmask3d_amd import ops

    torch.manual_seed(0)
    for rows, cols in ((7, 4096), (33, 1024), (5, 8192), (3, 12)):
        s = torch.randn(rows, cols, device=dev) * 30
        s[0, : cols // 2] = -1e30                                   # masked-out half a row
        got = ops.softmax_rows(s, 0.044).float()
        want = torch.softmax(s.double() * 0.044, -1)
        assert (got.double() - want).abs().max().item() < 4e-3 * want.max().item() + 1e-6   # bf16 output rounding
        assert torch.allclose(got.sum(-1), torch.ones(rows, device=dev), atol=2e-2)
    with pytest.raises(RuntimeError):
        ops.softmax_rows(torch.zeros(2, 10, device=dev), 1.0)       # cols % 4
    with pytest.raises(RuntimeError):
        ops.softmax_rows(torch.zeros(2, 16, device=dev), -1.0)


def test_vae_attention_block_unfused_path_matches_fp32(dev):
    """VaeAttnBlock in bf16 channels-last (two GEMMs around xm3d_softmax_rows_f32_bf16) against the same block in fp32"""
    from xmask3d_amd.sd_model import VaeAttnBlock

    torch.manual_seed(1)
    blk = VaeAttnBlock(512).to(dev).eval()
    x = torch.randn(2, 512, 32, 32, device=dev)
    with torch.no_grad():
        want = blk(x)
        half = VaeAttnBlock(512).to(dev).eval()
        half.load_state_dict(blk.state_dict())
        half = half.to(torch.bfloat16).to(memory_format=torch.channels_last)
        got = half(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)).float()
    rel = (got - want).abs().max().item() / want.abs().max().item()
    assert rel < 2e-2, rel


def test_quick_gelu_matches_expression(dev):
    from xmask3d_amd import ops

    torch.manual_seed(2)
    x = torch.randn(3, 307, 4096, device=dev) * 4
    assert torch.allclose(ops.quick_gelu(x), x * torch.sigmoid(1.702 * x), atol=1e-6, rtol=1e-5)
    xb = x.to(torch.bfloat16)
    want = (xb.float() * torch.sigmoid(1.702 * xb.float()))
    assert (ops.quick_gelu(xb).float() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item()
    with pytest.raises(TypeError):
        ops.quick_gelu(x.transpose(0, 2))


def test_pad_nhwc_matches_f_pad(dev):
    import torch.nn.functional as F
    from xmask3d_amd import ops

    torch.manual_seed(3)
    for dtype, shape in ((torch.bfloat16, (2, 128, 30, 22)), (torch.float32, (1, 8, 5, 7))):
        x = torch.randn(shape, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        y = ops.pad_bottom_right_nhwc(x, 1, 1)
        assert y.is_contiguous(memory_format=torch.channels_last) and torch.equal(y, F.pad(x, (0, 1, 0, 1)))
        assert torch.equal(ops.pad_bottom_right_nhwc(x, 0, 3), F.pad(x, (0, 3, 0, 0)))
    with pytest.raises(TypeError):
        ops.pad_bottom_right_nhwc(torch.zeros(1, 8, 4, 4, device=dev), 1, 1)   # NCHW


@pytest.mark.parametrize("dtype,rows,C", [(torch.bfloat16, 4096 * 2 + 3, 320), (torch.bfloat16, 307 * 3, 1024), (torch.float32, 1001, 256),
                                           (torch.bfloat16, 17, 4096), (torch.bfloat16, 5, 8)])
def test_layer_norm_kernel_matches_torch(dev, dtype, rows, C):
    import torch.nn.functional as F
    from xmask3d_amd import ops

    torch.manual_seed(C)
    x = (torch.randn(rows, C, device=dev) * 3 + 1).to(dtype)
    d = torch.randn(rows, C, device=dev).to(dtype)
    w, b = (torch.rand(C, device=dev) + 0.5).to(dtype), torch.randn(C, device=dev).to(dtype)
    tol = 1e-5 if dtype == torch.float32 else 2 ** -7
    want = F.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-5)
    got = ops.layer_norm(x, w, b, 1e-5)
    assert got.dtype == dtype and (got.float() - want).abs().max().item() <= tol * max(want.abs().max().item(), 1.0)
    s_want = (x + d)                                                    # the residual stream in the storage dtype
    y, s = ops.layer_norm(x, w, b, 1e-5, delta=d, want_sum=True)
    assert torch.equal(s, s_want)
    want = F.layer_norm(s_want.float(), (C,), w.float(), b.float(), 1e-5)
    assert (y.float() - want).abs().max().item() <= tol * max(want.abs().max().item(), 1.0)
    y2 = ops.layer_norm(x, None, None, 1e-5, delta=d)                   # no affine, sum not requested
    want = F.layer_norm((x.float() + d.float()), (C,), None, None, 1e-5)
    assert (y2.float() - want).abs().max().item() <= 2 * tol * max(want.abs().max().item(), 1.0)
    with pytest.raises(TypeError):
        ops.layer_norm(x[:, : C - 4] if C > 8 else x.t(), None, None)


def _ref_grads(q, k, v, bias, scale, go):
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    o = _ref(qf, kf, vf, bias, scale).permute(0, 2, 1, 3)  # (B, Nq, H, D)
    o.backward(go.float())
    return o.detach(), qf.grad, kf.grad, vf.grad


@pytest.mark.parametrize("B,H,Nq,Nk,D,with_bias", [(2, 8, 1024, 1024, 80, False), (1, 8, 4096, 4096, 40, False), (2, 8, 4096, 77, 40, False),
                                                   (2, 8, 256, 77, 160, False), (3, 8, 64, 64, 160, False), (2, 8, 50, 256, 32, True),
                                                   (1, 2, 95, 130, 64, True), (1, 1, 33, 65, 8, False), (1, 3, 130, 191, 96, True)])
def test_attention_backward_matches_fp32_autograd(dev, B, H, Nq, Nk, D, with_bias):
    """dQ, dK, dV of xm3d_attention_bwd (flash-style recomputation from the forward's log-sum-exp) against autograd through an fp32
    softmax(q k^T scale + bias) v on the same bf16 operands, at the UNet's self / cross attention shapes and ragged / masked cases.
    bf16 gradients out: bound 3e-2 of the gradient's max magnitude (bf16 P, dS and output rounding; measured <= 1.2e-2)."""
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(B * 100 + Nq + 7 * Nk + D)
    q = torch.randn(B, Nq, H * D, generator=g).to(dev, torch.bfloat16).view(B, Nq, H, D).requires_grad_(True)
    kv = torch.randn(B, Nk, 2 * H * D, generator=g).to(dev, torch.bfloat16)  # k / v as two halves of one projection buffer (strided views)
    k = kv[..., : H * D].view(B, Nk, H, D).detach().requires_grad_(True)
    v = kv[..., H * D:].view(B, Nk, H, D).detach().requires_grad_(True)
    bias = None
    if with_bias:
        bias = torch.zeros(B, 1, Nq, Nk)
        bias.masked_fill_(torch.rand(B, 1, Nq, Nk, generator=g) < 0.4, float("-inf"))
        bias[:, :, :, 0] = 0.0          # no fully masked row in general ...
        bias[0, 0, 3, :] = float("-inf")  # ... but one: its output and all its gradient contributions are zero
        bias = bias.to(dev)
    go = torch.randn(B, Nq, H, D, generator=g).to(dev, torch.bfloat16)
    scale = D ** -0.5
    assert ops.attention_train_supported(q, k, v, bias)
    out = ops.attention_train(q, k, v, bias=bias, scale=scale)
    out.backward(go)
    bias_r, go_r = bias, go
    if with_bias:  # the reference softmax of an all -inf row is NaN: give it an unmasked row that receives no gradient instead
        bias_r, go_r = bias.clone(), go.clone()
        bias_r[0, 0, 3, :] = 0.0
        go_r[0, 3] = 0
    ro, rq, rk, rv = _ref_grads(q, k, v, bias_r, scale, go_r)
    if with_bias:
        assert float(out.detach()[0, 3].abs().max()) == 0.0
        ro[0, 3] = 0
    assert (out.float() - ro).abs().max().item() / ro.abs().max().item() < 2e-2
    for name, got, want in (("dq", q.grad, rq), ("dk", k.grad, rk), ("dv", v.grad, rv)):
        assert got.shape == want.shape and torch.isfinite(got.float()).all()
        err = (got.float() - want).abs().max().item() / max(want.abs().max().item(), 1e-20)
        assert err < 3e-2, (name, err)
    if with_bias:
        assert float(q.grad[0, 3].abs().max()) == 0.0  # the fully masked query row


def test_attention_backward_is_reproducible_and_used_by_the_unet(dev):
    from xmask3d_amd import ops, sd_model

    g = torch.Generator().manual_seed(5)
    q = torch.randn(2, 300, 8 * 40, generator=g).to(dev, torch.bfloat16).view(2, 300, 8, 40).requires_grad_(True)
    k = torch.randn(2, 77, 8 * 40, generator=g).to(dev, torch.bfloat16).view(2, 77, 8, 40).requires_grad_(True)
    v = torch.randn(2, 77, 8 * 40, generator=g).to(dev, torch.bfloat16).view(2, 77, 8, 40).requires_grad_(True)
    go = torch.randn(2, 300, 8, 40, generator=g).to(dev, torch.bfloat16)
    grads = []
    for _ in range(2):
        for t in (q, k, v):
            t.grad = None
        ops.attention_train(q, k, v).backward(go)
        grads.append([t.grad.clone() for t in (q, k, v)])
    assert all(torch.equal(a, b) for a, b in zip(*grads))  # no atomics: bit-identical
    # ldm's CrossAttention under autograd in bf16 takes the HIP path: gradients reach the context
    torch.manual_seed(0)
    att = sd_model.CrossAttention(320, 768, 8, 40).to(dev, torch.bfloat16)
    x = torch.randn(2, 1024, 320, device=dev, dtype=torch.bfloat16)
    ctx = torch.randn(2, 77, 768, device=dev, dtype=torch.bfloat16, requires_grad=True)
    seen = []
    orig = ops.attention_train
    ops.attention_train = lambda *a, **kw: (seen.append(1), orig(*a, **kw))[1]
    try:
        att(x, ctx).float().square().mean().backward()
    finally:
        ops.attention_train = orig
    assert seen and ctx.grad is not None and float(ctx.grad.abs().sum()) > 0


# ---------------------------------------------------------------- f32-accurate attention (attention_f32.hip)
def _ref64(q, k, v, bias, scale):
    q, k, v = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
    s = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias.double()
    return (torch.softmax(s, -1).nan_to_num(0.0) @ v).permute(0, 2, 1, 3)


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 8, 4096, 4096, 40), (2, 8, 4096, 77, 40), (2, 16, 257, 257, 64), (2, 8, 50, 1024, 32),
                                         (1, 2, 33, 65, 8), (1, 1, 1, 1, 16), (1, 3, 130, 191, 48), (2, 4, 100, 64, 24)])
def test_attention_f32_matches_f64_reference(dev, B, H, Nq, Nk, D):
    """xm3d_attention_fwd_f32 (operands split in IEEE halves, three MFMAs per product, f32 softmax) against an f64 softmax attention:
    bound 2e-6 of max|out| - what torch's own f32 MATH attention reaches on the same inputs (measured beside it)."""
    from xmask3d_amd import ops

    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk + D)
    q = (torch.randn(B, Nq, H * D, generator=g) * 1.5).to(dev).view(B, Nq, H, D)
    k = (torch.randn(B, Nk, H * D, generator=g) * 1.5).to(dev).view(B, Nk, H, D)
    v = (torch.randn(B, Nk, H * D, generator=g) * torch.logspace(-2, 1.5, H * D)).to(dev).view(B, Nk, H, D)
    assert ops.attention_f32_supported(q, k, v)
    out = ops.attention_f32(q, k, v)
    assert out.shape == (B, Nq, H, D) and out.dtype == torch.float32 and torch.equal(out, ops.attention_f32(q, k, v))
    ref = _ref64(q, k, v, None, D ** -0.5)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    lib_err = float((_ref(q, k, v, None, D ** -0.5).permute(0, 2, 1, 3).double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, (err, lib_err)
    if Nk > 64:  # a key that dominates late: the running-maximum rescale across tiles, scores of magnitude ~100
        k2 = k.clone()
        k2[:, Nk - 3] = q[:, 0:1].expand(-1, 1, -1, -1).reshape(B, H, D) * 4
        ref2 = _ref64(q, k2, v, None, D ** -0.5)
        assert float((ops.attention_f32(q, k2, v).double() - ref2).abs().max() / ref2.abs().max()) < 2e-6


def test_attention_f32_masks_strides_and_output_views(dev):
    """packed (B, T, 3, H, D) qkv views, additive f32 mask with -inf entries and a fully masked row (-> zeros), broadcast over heads,
    output written into a transposed (L, B, E) buffer as the decoder layers do"""
    from xmask3d_amd import ops

    B, T, H, D, Qn = 3, 257, 16, 64, 50
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, T, 3, H, D, generator=g).to(dev)
    bias = torch.zeros(B, 1, T, T)
    bias[:, :, :, :Qn] = float("-inf")
    bias[:, :, :Qn, Qn + 1:].masked_fill_(torch.rand(B, 1, Qn, T - Qn - 1, generator=g) < 0.7, float("-inf"))
    bias[1, 0, 5, :] = float("-inf")  # a fully masked row
    bias = bias.to(dev)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    assert ops.attention_f32_supported(q, k, v, bias)
    out = ops.attention_f32(q, k, v, bias=bias)
    ref = _ref64(q, k, v, bias, D ** -0.5)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float(out[1, 5].abs().max()) == 0.0
    # decoder layout: (L, B, E) rows viewed as (B, L, H, d), bias (B, 1, Lq, Lk) shared by the heads
    Lq, Lk, E, Hd = 50, 1024, 256, 8
    qd, kd, vd = (torch.randn(n, B, E, generator=g).to(dev).view(n, B, Hd, E // Hd).transpose(0, 1) for n in (Lq, Lk, Lk))
    mb = torch.where(torch.rand(B, 1, Lq, Lk, generator=g) < 0.5, 0.0, float("-inf")).to(dev)
    o = torch.empty(Lq, B, E, device=dev)
    ops.attention_f32(qd, kd, vd, bias=mb, out=o.view(Lq, B, Hd, E // Hd).transpose(0, 1))
    refd = _ref64(qd, kd, vd, mb, (E // Hd) ** -0.5)
    assert float((o.view(Lq, B, Hd, E // Hd).transpose(0, 1).double() - refd).abs().max() / refd.abs().max()) < 2e-6
    assert not ops.attention_f32_supported(torch.zeros(1, 4, 2, 80, device=dev), torch.zeros(1, 4, 2, 80, device=dev), torch.zeros(1, 4, 2, 80, device=dev))


@pytest.mark.parametrize("rows,C,prow,ddt,pdt", [(1000, 256, 1000, torch.bfloat16, torch.float32), (5376 * 2, 256, 5376, torch.float32, torch.bfloat16),
                                                 (7, 1024, 7, torch.bfloat16, torch.bfloat16), (33, 64, 11, None, torch.float32)])
def test_add_layer_norm_matches_the_torch_chain(dev, rows, C, prow, ddt, pdt):
    """xm3d_add_layer_norm == LayerNorm(x + delta.float()) in f32, its bf16 rounding, and bf16(y + pos) with pos rows repeating along the stream"""
    import torch.nn.functional as F
    from xmask3d_amd import ops

    g = torch.Generator(device="cpu").manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(dev)
    d = None if ddt is None else torch.randn(rows, C, generator=g).to(dev, ddt)
    pos = torch.randn(prow, C, generator=g).to(dev, pdt)
    w, b = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    s = x if d is None else x + d.float()
    want = F.layer_norm(s, (C,), w, b, 1e-5)
    y, yb, yp = ops.add_layer_norm(x, d, w, b, 1e-5, pos=pos, want=("f32", "bf16", "pos"))
    assert (y - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    assert torch.equal(yb, y.to(torch.bfloat16))
    assert torch.equal(yp, (y.view(rows // prow, prow, C) + pos.float()).view(rows, C).to(torch.bfloat16))
    only = ops.add_layer_norm(x, d, w, b, 1e-5, want=("bf16",))
    assert torch.equal(only, yb)
    # the fp32 configuration's form: y and y + pos in f32
    y32, yp32 = ops.add_layer_norm(x, d, w, b, 1e-5, pos=pos, want=("f32", "pos"), out_dtype=torch.float32)
    assert torch.equal(y32, y) and yp32.dtype == torch.float32
    assert torch.equal(yp32, (y.view(rows // prow, prow, C) + pos.float()).view(rows, C))
