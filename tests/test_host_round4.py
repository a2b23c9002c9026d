"""Host-side logic of the round-4 wrappers on the CPU (no compute calls): which tensors the new HIP paths accept, that the modules fall back to
torch without a device, the switches.  The kernels are covered by the `-m gpu` tests (test_gpu_gemm.py, test_gpu_attention.py, test_gpu_train.py,
test_gpu_maskhead.py, test_gpu_conv_gemm.py, test_gpu_f32acc.py)."""
import torch
import torch.nn.functional as F

from xmask3d_amd import mask_head, norm_train, ops, sd_model, train_graph  # noqa: F401  (train_graph: importable without a device)


def test_training_norm_functions_refuse_cpu_tensors_and_the_modules_fall_back_to_torch():
    torch.manual_seed(0)
    x = torch.randn(6, 64, requires_grad=True)
    ln = sd_model.LayerNorm(64)
    assert not norm_train.layer_norm_ok(x, ln.weight, ln.bias)
    assert torch.allclose(ln(x), F.layer_norm(x, (64,), ln.weight, ln.bias, ln.eps))
    img = torch.randn(2, 64, 8, 8, requires_grad=True)
    gn = sd_model.GroupNorm(32, 64)
    assert not norm_train.group_norm_ok(img, gn)
    assert torch.allclose(gn(img), F.group_norm(img, 32, gn.weight, gn.bias, gn.eps))
    assert torch.allclose(sd_model.gn_act(gn, img, sd_model.ACT_RELU), F.relu(F.group_norm(img, 32, gn.weight, gn.bias, gn.eps)))
    lin = sd_model.Linear(64, 32)
    assert not norm_train.linear_ok(x, lin.weight)
    y = sd_model.flinear(x, lin.weight, lin.bias, act="relu")
    assert torch.allclose(y, F.relu(F.linear(x, lin.weight, lin.bias)))
    y.sum().backward()                                                   # plain autograd on the CPU
    assert lin.weight.grad is not None and x.grad is not None


def test_stream_paths_and_fused_layer_norm_are_off_without_a_device():
    x = torch.zeros(4, 2, 256)
    assert not mask_head._bf16_stream_ok(x)
    assert not ops.add_layer_norm_supported(x, 256)
    assert ops.GEMM_ACTS["relu"] == 4 and ops.GEMM_ACTS["geglu"] == 3
    mha = torch.nn.MultiheadAttention(64, 4)
    assert not mask_head._mha_train_ok(mha, torch.zeros(3, 2, 64))


def test_mha_train_equals_the_module_on_the_cpu_arithmetic():
    """_mha_train restates nn.MultiheadAttention's forward (projections, heads, boolean mask convention, output projection); on the CPU both are
    plain torch, so they must agree to rounding - including the module's `True = may not attend` mask"""
    torch.manual_seed(1)
    mha = torch.nn.MultiheadAttention(64, 4).eval()
    q, k, v = torch.randn(5, 2, 64), torch.randn(7, 2, 64), torch.randn(7, 2, 64)
    mask = torch.rand(2 * 4, 5, 7) < 0.3
    mask[:, :, 0] = False                                                # every query keeps one key
    want = mha(q, k, v, attn_mask=mask, need_weights=False)[0]
    got = mask_head._mha_train(mha, q, k, v, mask)
    assert torch.allclose(got, want, atol=1e-5)
    assert torch.allclose(mask_head._mha_train(mha, q, k, v), mha(q, k, v, need_weights=False)[0], atol=1e-5)


def test_positional_embedding_cache_is_device_only_and_values_match():
    pe = mask_head.PositionEmbeddingSine(64, normalize=True)
    a = pe(torch.zeros(2, 8, 6, 5))
    b = pe._compute(2, 6, 5, torch.device("cpu"), None)
    assert torch.equal(a, b) and "_pe_cache" not in pe.__dict__         # CPU calls never populate the cache


def test_sdpa_backend_switch_is_documented_and_reversible():
    """importing the package switches torch's Triton-built SDPA backends off process-wide (no Triton on this path); a host application
    can take torch's defaults back"""
    import xmask3d_amd

    flash, mem = torch.backends.cuda.flash_sdp_enabled(), torch.backends.cuda.mem_efficient_sdp_enabled()
    try:
        xmask3d_amd._no_triton_attention()
        assert not torch.backends.cuda.flash_sdp_enabled() and not torch.backends.cuda.mem_efficient_sdp_enabled()
        assert torch.backends.cuda.math_sdp_enabled()
        xmask3d_amd.restore_sdpa_defaults()
        assert torch.backends.cuda.flash_sdp_enabled() and torch.backends.cuda.mem_efficient_sdp_enabled()
    finally:
        torch.backends.cuda.enable_flash_sdp(flash)
        torch.backends.cuda.enable_mem_efficient_sdp(mem)
