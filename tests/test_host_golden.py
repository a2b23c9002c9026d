"""CPU: host-side restatements against vectors captured from the reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch


def test_position_embedding_sine_matches_reference(golden_dir):
    from xmask3d_amd.mask_head import PositionEmbeddingSine

    g = np.load(os.path.join(golden_dir, "sine_pe.npz"))
    pe = PositionEmbeddingSine(128, normalize=True)(torch.zeros(*g["shape"].tolist()))
    np.testing.assert_allclose(pe.numpy(), g["pe"], rtol=1e-6, atol=1e-6)
    # the explicit-mask path (cumsum) must agree with the closed form used when mask is None
    x = torch.zeros(2, 8, 5, 7)
    a = PositionEmbeddingSine(4, normalize=True)(x)
    b = PositionEmbeddingSine(4, normalize=True)(x, torch.zeros(2, 5, 7, dtype=torch.bool))
    assert torch.allclose(a, b, atol=1e-6)


def test_ensemble_logits_matches_reference(golden_dir):
    from xmask3d_amd.xmask3d import ensemble_logits_with_labels

    g = np.load(os.path.join(golden_dir, "ensemble.npz"))
    labels = [["x"] * int(n) for n in g["lens"]]
    logits = torch.from_numpy(g["logits"])
    np.testing.assert_allclose(ensemble_logits_with_labels(logits, labels, "max").numpy(), g["out_max"], rtol=0, atol=0)
    np.testing.assert_allclose(ensemble_logits_with_labels(logits, labels, "mean").numpy(), g["out_mean"], rtol=1e-6, atol=1e-7)
    single = [["a"]] * 7
    assert torch.equal(ensemble_logits_with_labels(logits, single), logits)


def test_diffusion_constants_match_reference_schedule(golden_dir):
    from xmask3d_amd import image_branch

    g = np.load(os.path.join(golden_dir, "diffusion.npz"))
    assert abs(image_branch.SQRT_AC0 - float(g["sqrt_ac0"])) < 1e-15
    assert abs(image_branch.SQRT_1M_AC0 - float(g["sqrt_1m_ac0"])) < 1e-15
    # "ldm_linear": betas = linspace(sqrt(0.00085), sqrt(0.012), 1000)**2 (gaussian_diffusion.py:61-90)
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    np.testing.assert_allclose(betas[:4], g["betas_head"], rtol=1e-12)


def test_feature_dims_and_strides_of_the_extractor():
    """ldm.py:255-310: dims [512,512,2560,1920,960,640,512,512], strides [4,8,64,32,16,8,8,4]; grouped into s2..s5."""
    from xmask3d_amd.image_branch import FeatureExtractorBackbone, LdmImplicitCaptionerExtractor

    with torch.device("meta"):
        ext = LdmImplicitCaptionerExtractor()
        bb = FeatureExtractorBackbone(ext, ["s2", "s3", "s4", "s5"])
    assert ext.feature_dims == [512, 512, 2560, 1920, 960, 640, 512, 512]
    assert ext.feature_strides == [4, 8, 64, 32, 16, 8, 8, 4]
    assert [(n, s, idx) for n, s, idx in bb._groups] == [("s2", 4, [0, 7]), ("s3", 8, [1, 5, 6]), ("s4", 16, [4]), ("s5", 32, [2, 3])]
    assert bb.output_shape() == {"s2": (512, 4), "s3": (512, 8), "s4": (512, 16), "s5": (512, 32)}
