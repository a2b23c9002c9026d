"""GPU parity: deformable attention (vs golden vectors of the reference's CPU path and the oracle) and the
mask->point fusion kernel (vs golden vectors of the reference's mask_mapper)."""
import os

import numpy as np
import pytest
import torch

from oracle import msda_oracle as mo

pytestmark = pytest.mark.gpu


def _load(golden_dir, name, dev):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    t = {k: torch.from_numpy(g[k]).to(dev) for k in g.files}
    return g, t


@pytest.mark.parametrize("name", ["msda_toy_f32", "msda_d32_f32", "msda_toy_f64", "msda_d32_f64"])
def test_msda_golden(dev, golden_dir, name):
    from xmask3d_amd import msda

    g, t = _load(golden_dir, name, dev)
    # the reference's own float tolerance: ops/test.py:59 rtol=1e-2, atol=1e-3 (we are far inside it)
    out = msda.ms_deform_attn_forward(t["value"], t["shapes"], t["level_start"], t["loc"], t["w"], 64)
    assert out.dtype == t["value"].dtype
    f64 = name.endswith("f64")  # double has its own kernel instantiation (like the reference's dispatch): held to 1e-10
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-10 if f64 else 1e-4, atol=1e-13 if f64 else 1e-6)
    gv, gl, gw = msda.ms_deform_attn_backward(t["value"], t["shapes"], t["level_start"], t["loc"], t["w"], t["grad_out"], 64)
    for mine, ref in ((gv, g["g_value"]), (gl, g["g_loc"]), (gw, g["g_w"])):
        assert mine.dtype == t["value"].dtype
        np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=1e-9 if f64 else 2e-3, atol=(1e-12 if f64 else 2e-5) * np.abs(ref).max())


def test_msda_autograd_function_and_module(dev, golden_dir):
    from xmask3d_amd import msda

    g, t = _load(golden_dir, "msda_d32_f32", dev)
    v, l, w = (t[k].clone().requires_grad_(True) for k in ("value", "loc", "w"))
    out = msda.MSDeformAttnFunction.apply(v, t["shapes"], t["level_start"], l, w, 128)
    out.backward(t["grad_out"])
    np.testing.assert_allclose(v.grad.cpu().numpy(), g["g_value"], rtol=2e-3, atol=2e-5 * np.abs(g["g_value"]).max())
    np.testing.assert_allclose(w.grad.cpu().numpy(), g["g_w"], rtol=2e-3, atol=2e-5 * np.abs(g["g_w"]).max())
    torch.manual_seed(0)
    mod = msda.MSDeformAttn(256, 3, 8, 4).to(dev)
    shapes = torch.tensor([[8, 8], [4, 4], [2, 2]], device=dev)
    lsi = torch.tensor([0, 64, 80], device=dev)
    q = torch.randn(2, 84, 256, device=dev)
    ref_pts = torch.rand(2, 84, 3, 2, device=dev)
    y = mod(q, ref_pts, q, shapes, lsi)
    assert y.shape == (2, 84, 256) and torch.isfinite(y).all()


def test_msda_real_shape_vs_oracle(dev):
    """B=1, levels 16^2/32^2/64^2 (5376 tokens), H=8, D=32, P=4: the pixel decoder's shape."""
    from xmask3d_amd import ops

    torch.manual_seed(5)
    shapes = torch.tensor([[16, 16], [32, 32], [64, 64]])
    lsi = torch.tensor([0, 256, 1280])
    S = 5376
    value = torch.randn(1, S, 8, 32)
    loc = torch.rand(1, S, 8, 3, 4, 2) * 1.2 - 0.1
    w = torch.softmax(torch.randn(1, S, 8, 12), -1).view(1, S, 8, 3, 4)
    out = ops.msda_forward(value.to(dev), shapes.to(dev), lsi.to(dev), loc.to(dev), w.to(dev))
    ref = mo.forward(value.numpy().astype(np.float64), shapes.numpy(), lsi.numpy(), loc.numpy().astype(np.float64),
                     w.numpy().astype(np.float64))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    go = torch.randn(1, S, 256)
    gv, gl, gw = ops.msda_backward(value.to(dev), shapes.to(dev), lsi.to(dev), loc.to(dev), w.to(dev), go.to(dev))
    rv, rl, rw = mo.backward(value.numpy().astype(np.float64), shapes.numpy(), lsi.numpy(), loc.numpy().astype(np.float64),
                             w.numpy().astype(np.float64), go.numpy().astype(np.float64))
    np.testing.assert_allclose(gv.cpu().numpy(), rv, rtol=1e-3, atol=1e-4 * np.abs(rv).max())
    np.testing.assert_allclose(gl.cpu().numpy(), rl, rtol=1e-3, atol=1e-4 * np.abs(rl).max())
    np.testing.assert_allclose(gw.cpu().numpy(), rw, rtol=1e-3, atol=1e-4 * np.abs(rw).max())


def test_msda_errors_like_reference(dev):
    from xmask3d_amd import msda

    v = torch.zeros(3, 4, 1, 4, device=dev)
    sh, ls = torch.tensor([[2, 2]], device=dev), torch.tensor([0], device=dev)
    loc, w = torch.zeros(3, 1, 1, 1, 1, 2, device=dev), torch.zeros(3, 1, 1, 1, 1, device=dev)
    with pytest.raises(RuntimeError, match="contiguous"):
        msda.ms_deform_attn_forward(torch.zeros(3, 8, 1, 4, device=dev)[:, ::2], sh, ls, loc, w, 64)
    with pytest.raises(RuntimeError, match="must divide"):
        msda.ms_deform_attn_forward(v, sh, ls, loc, w, 2)


def test_mask_point_fuse_matches_reference_mask_mapper(dev, golden_dir):
    from xmask3d_amd import ops

    g = np.load(os.path.join(golden_dir, "fuser.npz"))
    W, b = torch.from_numpy(g["W"]), torch.from_numpy(g["b"])
    for i in range(2):
        mask = torch.from_numpy(g[f"mask{i}"])
        x, y = torch.from_numpy(g[f"x{i}"]), torch.from_numpy(g[f"y{i}"])
        emb, p3d = torch.from_numpy(g[f"emb{i}"]), torch.from_numpy(g[f"p3d{i}"])
        m8 = (mask >= 0.5).to(torch.uint8)
        feat, cnt = ops.mask_point_fuse(m8.to(dev), x.to(dev), y.to(dev), emb.to(dev).contiguous())
        feat, cnt = feat.cpu(), cnt.cpu()
        np.testing.assert_allclose(feat.numpy(), g[f"f2d{i}"], rtol=1e-5, atol=1e-5 * np.abs(g[f"f2d{i}"]).max())
        covered = cnt >= 1
        fused = p3d.clone()
        fused[covered] = torch.cat([feat[covered], p3d[covered]], 1) @ W.T + b
        np.testing.assert_allclose(fused.numpy(), g[f"fused{i}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype,shape,act", [(torch.float32, (2, 64, 16, 24), 1), (torch.bfloat16, (3, 128, 32, 32), 1),
                                            (torch.bfloat16, (1, 320, 64, 64), 0), (torch.float32, (2, 512, 8, 8), 2),
                                            (torch.bfloat16, (2, 2560, 8, 8), 1), (torch.bfloat16, (1, 128, 256, 256), 1)])
def test_fused_group_norm_matches_torch(dev, dtype, shape, act):
    from xmask3d_amd import ops

    torch.manual_seed(sum(shape))
    x = (torch.randn(shape) * 2 + 0.3)
    w, b = torch.rand(shape[1]) + 0.5, torch.randn(shape[1])
    ref = torch.nn.functional.group_norm(x.to(dtype).float(), 32, w.to(dtype).float(), b.to(dtype).float(), 1e-6)
    ref = ref * torch.sigmoid(ref) if act == 1 else (torch.relu(ref) if act == 2 else ref)
    y = ops.group_norm(x.to(dev).to(dtype), 32, w.to(dev).to(dtype), b.to(dev).to(dtype), 1e-6, act)
    assert y.dtype == dtype and y.shape == x.shape
    tol = 1e-5 if dtype == torch.float32 else 2e-2  # bf16 output rounding
    assert (y.float().cpu() - ref).abs().max().item() <= tol * max(ref.abs().max().item(), 1.0)
    # channels-last input -> NHWC kernel, channels-last output, same values
    xc = x.to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    yc = ops.group_norm(xc, 32, w.to(dev).to(dtype), b.to(dev).to(dtype), 1e-6, act)
    assert yc.is_contiguous(memory_format=torch.channels_last) and yc.shape == x.shape
    assert (yc.float().cpu() - ref).abs().max().item() <= max(tol, 2e-5) * max(ref.abs().max().item(), 1.0)
    # residual added after the affine, before the activation (bottleneck tail): NHWC kernel and the NCHW fallback
    r = torch.randn(shape)
    pre = torch.nn.functional.group_norm(x.to(dtype).float(), 32, w.to(dtype).float(), b.to(dtype).float(), 1e-6) + r.to(dtype).float()
    ref_r = pre * torch.sigmoid(pre) if act == 1 else (torch.relu(pre) if act == 2 else pre)
    rc = r.to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    for xin, rin in ((xc, rc), (x.to(dev).to(dtype), r.to(dev).to(dtype))):
        yr = ops.group_norm(xin, 32, w.to(dev).to(dtype), b.to(dev).to(dtype), 1e-6, act, residual=rin)
        assert (yr.float().cpu() - ref_r).abs().max().item() <= max(2 * tol, 2e-5) * max(ref_r.abs().max().item(), 1.0)


@pytest.mark.parametrize("dtype,shape", [(torch.bfloat16, (3, 128, 32, 24)), (torch.float32, (2, 64, 16, 16)), (torch.bfloat16, (2, 320, 64, 64))])
def test_bias_residual_carries_group_norm_statistics(dev, dtype, shape):
    """xm3d_bias_residual_stats_nhwc: same sum as the plain kernel, and the GroupNorm that follows (apply pass only, on the
    statistics taken on the way) equals the GroupNorm that computes its own"""
    from xmask3d_amd import ops

    torch.manual_seed(shape[1])
    a, b = (torch.randn(shape, device=dev).to(dtype).contiguous(memory_format=torch.channels_last) for _ in range(2))
    bias = torch.randn(shape[1], device=dev).to(dtype)
    w, g = (torch.rand(shape[1], device=dev) + 0.5).to(dtype), torch.randn(shape[1], device=dev).to(dtype)
    plain = ops.bias_residual(a, b, bias)
    carried = ops.bias_residual(a, b, bias, stats_groups=32)
    assert torch.equal(plain, carried) and getattr(carried, "_xm3d_gn_stats", None) is not None and not hasattr(plain, "_xm3d_gn_stats")
    for act in (0, 1):
        want = ops.group_norm(plain, 32, w, g, 1e-6, act)
        got = ops.group_norm(carried, 32, w, g, 1e-6, act)
        tol = 1e-5 if dtype == torch.float32 else 2 ** -7     # f32 partial sums in a different order: a last-digit effect
        assert (got.float() - want.float()).abs().max().item() <= tol * max(want.float().abs().max().item(), 1.0)
    # a shift (folded conv bias) or another group count invalidates the carried statistics: the full kernel runs
    # (the statistics are atomically accumulated partial sums: equal up to the last digit, not bit for bit)
    sh = torch.randn(shape[1], device=dev).to(dtype)
    close = lambda p, q: (p.float() - q.float()).abs().max().item() <= tol * max(q.float().abs().max().item(), 1.0)
    assert close(ops.group_norm(carried, 32, w, g, 1e-6, 1, sh), ops.group_norm(plain, 32, w, g, 1e-6, 1, sh))
    assert close(ops.group_norm(carried, 16, w, g, 1e-6, 1), ops.group_norm(plain, 16, w, g, 1e-6, 1))
    none = ops.bias_residual(None, b, bias, stats_groups=32)
    assert torch.equal(none, ops.bias_residual(None, b, bias))


def test_nearest_index_is_exact(dev):
    from xmask3d_amd import ops

    torch.manual_seed(0)
    ref = torch.rand(5000, 3) * 4
    q = torch.rand(3001, 3) * 4
    q[:10] = ref[100:110]  # exact hits
    ref[4000] = ref[7]  # duplicate reference point: the lower index must win
    q[10] = ref[7]
    got = ops.nearest_index(q.to(dev), ref.to(dev)).cpu()
    d = torch.cdist(q.double(), ref.double())
    assert (d.gather(1, got[:, None])[:, 0] <= d.min(1).values + 1e-9).all()
    assert got[:10].tolist() == list(range(100, 110)) and int(got[10]) == 7
    assert ops.nearest_index(q[:0].to(dev), ref.to(dev)).shape == (0,)
    one = ops.nearest_index(q.to(dev), ref[:1].to(dev))
    assert (one == 0).all()
    valid = torch.rand(5000) < 0.3
    got = ops.nearest_index(q.to(dev), ref.to(dev), valid.to(dev)).cpu()
    dm = d.clone()
    dm[:, ~valid] = float("inf")
    assert valid[got].all() and (dm.gather(1, got[:, None])[:, 0] <= dm.min(1).values + 1e-9).all()


def test_nearest_valid_fill_matches_bruteforce(dev):
    from xmask3d_amd import pipeline

    torch.manual_seed(1)
    xyz = torch.rand(7000, 3) * 3
    valid = torch.rand(7000) < 0.35
    fill = pipeline.nearest_valid_fill(xyz.to(dev), valid.to(dev)).cpu()
    d = torch.cdist(xyz.double(), xyz[valid].double())
    src = torch.nonzero(valid)[:, 0]
    want = src[d.argmin(1)]
    assert (fill[valid] == torch.nonzero(valid)[:, 0]).all()
    got_d = (xyz[~valid].double() - xyz[fill[~valid]].double()).norm(dim=1)
    assert valid[fill].all() and torch.allclose(got_d, d[~valid].min(1).values, atol=1e-6)
    # degenerate masks: everything valid -> identity; nothing valid -> every index is a legal row
    assert (pipeline.nearest_valid_fill(xyz.to(dev), torch.ones(7000, dtype=torch.bool, device=dev)).cpu() == torch.arange(7000)).all()
    none = pipeline.nearest_valid_fill(xyz.to(dev), torch.zeros(7000, dtype=torch.bool, device=dev)).cpu()
    assert int(none.min()) >= 0 and int(none.max()) < 7000


@pytest.mark.gpu
def test_segmented_nearest_fill_equals_per_segment_fill(dev):
    """xm3d_nearest_index_segmented (all views in one launch) == nearest_valid_fill on each view's points"""
    from xmask3d_amd import pipeline

    torch.manual_seed(2)
    sizes = [3000, 1, 4500, 700, 2048]
    xyz = (torch.rand(sum(sizes), 3) * 4).to(dev)
    valid = (torch.rand(sum(sizes)) < 0.3).to(dev)
    off = np.cumsum([0] + sizes)
    valid[off[1]:off[2]] = False            # a one-point segment without any reference: maps to itself
    valid[off[3]:off[4]] = True             # a segment with nothing to fill
    seg = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes)]).to(dev)
    got = pipeline.nearest_valid_fill_segmented(xyz, valid, seg, len(sizes), max(sizes)).cpu()
    for i in range(len(sizes)):
        lo, hi = int(off[i]), int(off[i + 1])
        if i == 1:
            assert got[lo:hi].tolist() == [lo]
            continue
        want = pipeline.nearest_valid_fill(xyz[lo:hi], valid[lo:hi]).cpu() + lo
        assert torch.equal(got[lo:hi], want), f"segment {i}"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bilinear_down_bit_identical_on_device(dev, dtype):
    import torch.nn.functional as F
    from xmask3d_amd.mask_head import bilinear_down

    torch.manual_seed(0)
    x = torch.randn(5, 50, 128, 128, device=dev).to(dtype)
    with torch.no_grad():
        for t in (16, 32, 64):
            assert torch.equal(bilinear_down(x, (t, t)), F.interpolate(x, size=(t, t), mode="bilinear", align_corners=False))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pointwise_fusions_match_torch(dev, dtype):
    """xm3d_bias_residual_nhwc, xm3d_geglu, xm3d_group_norm_nhwc(shift) against the torch op chains they replace"""
    import torch.nn.functional as F
    from xmask3d_amd import ops

    torch.manual_seed(0)
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    a = torch.randn(3, 64, 24, 40, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    b = torch.randn(3, 64, 24, 40, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(64, device=dev).to(dtype)
    ref = a.float() + b.float() + bias.float().view(1, -1, 1, 1)
    out = ops.bias_residual(a, b, bias)
    assert out.is_contiguous(memory_format=torch.channels_last) and out.dtype == dtype
    assert (out.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    out = ops.bias_residual(None, b, bias)
    assert (out.float() - (b.float() + bias.float().view(1, -1, 1, 1))).abs().max().item() <= tol * ref.abs().max().item()
    x = torch.randn(4, 77, 256, device=dev).to(dtype)
    u, g = x.float().chunk(2, -1)
    ref = u * F.gelu(g)
    out = ops.geglu(x)
    assert out.shape == (4, 77, 128) and (out.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    # GroupNorm with a per-sample shift == GroupNorm(x + shift)
    w, bb = torch.randn(64, device=dev).to(dtype), torch.randn(64, device=dev).to(dtype)
    for shift in (torch.randn(64, device=dev).to(dtype), torch.randn(3, 64, device=dev).to(dtype)):
        xs = a.float() + shift.float().view(-1, 64, 1, 1)
        ref = F.silu(F.group_norm(xs, 32, w.float(), bb.float(), 1e-5))
        out = ops.group_norm(a, 32, w, bb, 1e-5, 1, shift)
        assert out.is_contiguous(memory_format=torch.channels_last)
        assert (out.float() - ref).abs().max().item() <= max(tol, 2e-5) * max(ref.abs().max().item(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_mask_bias_equals_the_op_chain(dev, dtype):
    """xm3d_attn_mask_bias == bilinear shrink -> sigmoid -> < 0.5 -> all/and-not -> -inf fill (odise.py:395,445-491), bit for bit"""
    import torch.nn.functional as F
    from xmask3d_amd import ops

    torch.manual_seed(4)
    logits = (torch.randn(5, 50, 128, 128, device=dev) * 3).to(dtype)
    logits[1, 7] = -5.0          # a query whose mask is empty everywhere: attends to everything
    logits[2, 3] = 0.0           # sigmoid == 0.5 exactly: not masked
    for t in (16, 32, 64):
        with torch.no_grad():
            m = F.interpolate(logits, size=(t, t), mode="bilinear", align_corners=False).sigmoid().flatten(2) < 0.5
            m = m & ~m.all(dim=-1, keepdim=True)
            want = torch.zeros(m.shape, dtype=dtype, device=dev).masked_fill(m, float("-inf"))
            got = ops.attn_mask_bias(logits, (t, t), dtype)
        assert got.shape == (5, 50, t * t) and torch.equal(got, want)
        assert float(got[1, 7].abs().max()) == 0.0 and bool(torch.isinf(got).any())


@pytest.mark.gpu
def test_cross_attention_bias_path_equals_multihead_attention(dev):
    """CrossAttentionLayer(memory_bias=...) (manual projections + SDPA, bias broadcast over heads) == nn.MultiheadAttention with
    the replicated boolean mask"""
    from xmask3d_amd.mask_head import CrossAttentionLayer

    torch.manual_seed(5)
    layer = CrossAttentionLayer(256, 8).to(dev).eval()
    tgt, qpos = torch.randn(50, 3, 256, device=dev), torch.randn(50, 3, 256, device=dev)
    mem, pos = torch.randn(1024, 3, 256, device=dev), torch.randn(1024, 3, 256, device=dev)
    mask = torch.rand(3, 50, 1024, device=dev) < 0.6
    mask[0, 0] = False
    bias = torch.zeros(3, 50, 1024, device=dev).masked_fill(mask, float("-inf"))
    with torch.no_grad():
        ref = layer(tgt, mem, memory_mask=mask[:, None].repeat(1, 8, 1, 1).flatten(0, 1), pos=pos, query_pos=qpos)
        out = layer(tgt, mem, memory_bias=bias, pos=pos, query_pos=qpos)
    assert (out - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.gpu
def test_mask_owner_equals_the_op_chain(dev):
    """xm3d_mask_owner == sigmoid / (kept ? score : -1) product / argmax / (>= 0.5) & keep chain of models/xmask3d.py:372-392"""
    from xmask3d_amd import ops

    torch.manual_seed(6)
    B, Q, H, W = 3, 50, 60, 80
    logits = (torch.randn(B, Q, H, W, device=dev) * 2)
    logits[0, :, :5] = -9.0                     # pixels no query claims
    logits[1, 3] = logits[1, 7]                 # exact ties: the lower query id wins
    scores = torch.rand(B, Q, device=dev)
    scores[1, 3] = scores[1, 7]
    keep = torch.rand(B, Q, device=dev) < 0.7
    keep[2] = False                             # nothing kept: no owner anywhere
    mask_pred = logits.sigmoid()
    prob = torch.where(keep, scores, torch.full_like(scores, -1.0)).view(B, Q, 1, 1) * mask_pred
    ids = prob.argmax(1)
    final = (ids[:, None] == torch.arange(Q, device=dev).view(1, -1, 1, 1)) & (mask_pred >= 0.5) & keep.view(B, Q, 1, 1)
    want = torch.where(final.any(1), ids, torch.full_like(ids, -1)).int()
    got = ops.mask_owner(logits, scores, keep)
    assert torch.equal(got, want)
    assert int((got[2] >= 0).sum()) == 0 and int((got[0, :5] >= 0).sum()) == 0
