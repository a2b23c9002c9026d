"""The two operator seams the reference binds against fixtures captured from the reference's own files
(tests/golden/make_golden.py::gen_seams):

* ``import MinkowskiEngine as ME`` -> xmask3d_amd.me_compat: parameter names/shapes of the reference's PC_Processor /
  PC_Binary_Processor built on the product seam (seam_minkunet_keys.json), and the reference's own MinkUNet forward code
  executed on the CPU stand-in of the ME ops (seam_minkunet_forward.npz; pins TOPOLOGY, operator semantics stay
  parity-unpinned, see oracle/me_cpu_stub.py)
* ``import MultiScaleDeformableAttention as MSDA`` -> xmask3d_amd.msda: the positional calls the reference's
  MSDeformAttnFunction makes and the results it expects (seam_msda_call.{json,npz})
"""
import inspect
import json
import os

import numpy as np
import pytest
import torch

from oracle import me_cpu_stub, spconv_oracle


@pytest.fixture(scope="module")
def seam(golden_dir):
    with open(os.path.join(golden_dir, "seam_minkunet_keys.json")) as f:
        keys = json.load(f)
    with open(os.path.join(golden_dir, "seam_msda_call.json")) as f:
        call = json.load(f)
    return keys, np.load(os.path.join(golden_dir, "seam_minkunet_forward.npz")), call, np.load(os.path.join(golden_dir, "seam_msda_call.npz"))


def _own_nets():
    from xmask3d_amd.pc_processor import PC_Binary_Processor, PC_Processor

    return {"pc_decoder": PC_Processor(arch_3d="MinkUNet34C"), "pc_binary_head": PC_Binary_Processor(arch_3d="MinkUNet18A")}


def test_own_nets_have_the_reference_state_dict_layout(seam):
    keys = seam[0]
    for name, net in _own_nets().items():
        mine = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert mine == keys[name], f"{name}: state_dict layout differs from the reference's"
    assert len(keys["pc_decoder"]) == 377 and len(keys["pc_binary_head"]) == 296


def test_oracle_topology_equals_the_reference_forward_code(seam):
    """spconv_oracle's restated MinkUNet == the reference's forward code run on the same CPU ops"""
    keys, fx = seam[0], seam[1]
    coords, feats = fx["coords"], torch.from_numpy(fx["feats"])
    p = me_cpu_stub.closed_form_state(keys["pc_decoder"])
    imp, x, idx = spconv_oracle.pc_processor_forward(p, coords, feats, "MinkUNet34C")
    np.testing.assert_allclose(imp.numpy(), fx["implicit_x"], rtol=1e-4, atol=1e-5 * np.abs(fx["implicit_x"]).max())
    np.testing.assert_allclose(x[::8].numpy(), fx["x_rows"], rtol=1e-4, atol=1e-5 * np.abs(fx["x_rows"]).max())
    assert (idx.numpy() == fx["idx"]).all()
    pb = me_cpu_stub.closed_form_state(keys["pc_binary_head"])
    b = spconv_oracle.pc_binary_forward(pb, coords, feats, "MinkUNet18A")
    np.testing.assert_allclose(b.numpy(), fx["binary"], rtol=1e-4, atol=1e-5 * np.abs(fx["binary"]).max())


def test_msda_module_surface_matches_the_recorded_calls(seam):
    from xmask3d_amd import msda

    call = seam[2]
    fwd = inspect.signature(msda.ms_deform_attn_forward)
    bwd = inspect.signature(msda.ms_deform_attn_backward)
    assert len(fwd.parameters) == len(call["calls"]["ms_deform_attn_forward"]) == 6
    assert len(bwd.parameters) == len(call["calls"]["ms_deform_attn_backward"]) == 7
    assert all(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in list(fwd.parameters.values()) + list(bwd.parameters.values()))
    mod = msda.MSDeformAttn(d_model=256, n_levels=3, n_heads=8, n_points=4)
    assert {k: list(v.shape) for k, v in mod.state_dict().items()} == call["module_state"]
    assert mod.im2col_step == call["module_im2col_step"]
    # CPU tensors: the reference op raises (ms_deform_attn.h:39), no silent fallback
    fx = seam[3]
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        msda.ms_deform_attn_forward(*(torch.from_numpy(fx[k]) for k in ("value", "shapes", "level_start", "loc", "w")), 2)


@pytest.mark.gpu
def test_own_nets_reproduce_the_reference_forward_fixture(dev, seam):
    from xmask3d_amd import me_compat as ME

    keys, fx = seam[0], seam[1]
    coords = torch.from_numpy(fx["coords"]).to(dev)
    feats = torch.from_numpy(fx["feats"]).to(dev)
    nets = _own_nets()
    with torch.no_grad():
        net = nets["pc_decoder"].eval()
        net.load_state_dict(me_cpu_stub.closed_form_state(keys["pc_decoder"]))
        imp, x, idx = net.to(dev)(ME.SparseTensor(feats, coords))
        for mine, ref in ((imp, fx["implicit_x"]), (x[::8], fx["x_rows"])):
            err = np.abs(mine.cpu().numpy() - ref).max() / np.abs(ref).max()
            assert err < 1e-4, err
        assert (idx.cpu().numpy() == fx["idx"]).all()
        netb = nets["pc_binary_head"].eval()
        netb.load_state_dict(me_cpu_stub.closed_form_state(keys["pc_binary_head"]))
        b = netb.to(dev)(ME.SparseTensor(feats, coords))
        assert np.abs(b.cpu().numpy() - fx["binary"]).max() / np.abs(fx["binary"]).max() < 1e-4


@pytest.mark.gpu
def test_msda_recorded_call_on_the_device(dev, seam):
    """the exact positional calls of the reference's MSDeformAttnFunction, answered by the HIP kernels"""
    from xmask3d_amd import msda

    fx = seam[3]
    t = {k: torch.from_numpy(fx[k]).to(dev) for k in fx.files if fx[k].ndim > 0}
    step = int(fx["im2col_step"])
    out = msda.ms_deform_attn_forward(t["value"], t["shapes"], t["level_start"], t["loc"], t["w"], step)
    np.testing.assert_allclose(out.cpu().numpy(), fx["out"], rtol=1e-4, atol=1e-6)
    gv, gl, gw = msda.ms_deform_attn_backward(t["value"], t["shapes"], t["level_start"], t["loc"], t["w"], t["grad_out"], step)
    for mine, ref in ((gv, fx["g_value"]), (gl, fx["g_loc"]), (gw, fx["g_w"])):
        np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * np.abs(ref).max())
