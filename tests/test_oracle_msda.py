"""CPU: the deformable-attention oracle against the reference's own CPU implementation
(ms_deform_attn_core_pytorch outputs + autograd gradients captured in tests/golden)."""
import os

import numpy as np
import pytest

from oracle import msda_oracle as mo

CASES = [("msda_toy_f64", 1e-12), ("msda_toy_f32", 1e-6), ("msda_d32_f32", 2e-5), ("msda_d32_f64", 1e-12)]


@pytest.mark.parametrize("name,tol", CASES)
def test_forward_backward(golden_dir, name, tol):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    out = mo.forward(g["value"], g["shapes"], g["level_start"], g["loc"], g["w"])
    np.testing.assert_allclose(out, g["out"], rtol=tol, atol=tol * np.abs(g["out"]).max())
    gv, gl, gw = mo.backward(g["value"], g["shapes"], g["level_start"], g["loc"], g["w"], g["grad_out"])
    for mine, ref in ((gv, g["g_value"]), (gl, g["g_loc"]), (gw, g["g_w"])):
        np.testing.assert_allclose(mine, ref, rtol=tol, atol=tol * max(np.abs(ref).max(), 1e-30))
