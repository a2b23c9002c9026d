"""The executable path to the released-checkpoint numbers, end to end on a script-written fixture tree (nothing downloaded, no
released file in the container): Stable-Diffusion checkpoint + HuggingFace CLIP directory + BPE vocabulary in the layouts the
reference reads (ldm.py:105-114, clip.py:69-73,147-149; README.md:28-35), an XMask3D checkpoint, and a ScanNet scene in the on-disk
layout of data_loader_infer.py - then run/infer.py's flow: weights demonstrably come from the files, the scene from the tree, the
score is computed against the tree's labels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_infer_runs_on_local_weights_and_a_scannet_tree(dev, tmp_path):
    from tests.pretrained_fixture import write_pretrained
    from tests.scannet_fixture import write_scene
    from xmask3d_amd import checkpoint as ckpt_io
    from xmask3d_amd import config, driver
    from xmask3d_amd.clip_model import TextTower
    from xmask3d_amd.xmask3d import XMASK3d

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.load_cfg_from_cfg_file(os.path.join(root, "configs", "xmask3d_scannet_B15N4.yaml"))

    # ---- the "released" files: frozen nets of a donor model (seed 123) in the reference's layouts + its trainable checkpoint
    torch.manual_seed(123)
    donor = XMASK3d(cfg)
    assert donor.pretrained_report is None  # nothing on disk yet: seeded random weights, stand-in tokenizer
    paths = write_pretrained(str(tmp_path / "pre"), donor)
    ckpt = str(tmp_path / "model" / "b15n4.pth.tar")
    ckpt_io.save_checkpoint(ckpt, donor, None, 7, 0.0)
    fx = write_scene(str(tmp_path / "scannet"))
    cfg.data_root, cfg.data_root_2d, cfg.caption_path = fx["data_root"], fx["data_root_2d"], fx["caption_path"]
    cfg.sd_checkpoint, cfg.clip_dir = paths["sd_checkpoint"], paths["clip_dir"]

    # ---- a model built under ANOTHER seed comes out with the files' frozen nets, tokenizer and empty-prompt conditioning
    torch.manual_seed(999)
    m = XMASK3d(cfg)
    rep = m.pretrained_report
    assert rep["sd"] and rep["clip"] and rep["tokenizer"] and rep["uncond"] and not rep["refused"], rep
    dl, ml = donor.backbone.feature_extractor.ldm_extractor.ldm, m.backbone.feature_extractor.ldm_extractor.ldm
    for a, b in ((dl.unet_model, ml.unet_model), (dl.first_stage_model, ml.first_stage_model)):
        sa, sb = a.state_dict(), b.state_dict()
        assert set(sa) == set(sb)
        for k in list(sa)[::37]:
            assert torch.allclose(sa[k].float(), sb[k].float(), atol=2e-3, rtol=2e-3), k  # the file holds fp16
    ca, cb = donor.criterion.clip.clip.state_dict(), m.criterion.clip.clip.state_dict()
    assert all(torch.equal(ca[k].float(), cb[k].float()) for k in list(ca)[::23])
    tok = m.criterion.clip.bpe
    assert m.criterion.clip.tokenize(["the wall"])[0, :4].tolist() == [49406] + tok.encode("the wall") + [49407]
    tower = TextTower()
    tower.load_state_dict({k: v.to(torch.float16).float() for k, v in ca.items() if not k.startswith("visual.") and k not in ("logit_scale", "text_projection")})
    with torch.no_grad():
        want = tower(tok([""], pad_id=tok.eot))
    assert torch.allclose(ml.uncond_inputs.float(), want, atol=1e-4) and not torch.allclose(ml.uncond_inputs.float(), dl.uncond_inputs.float(), atol=1e-2)
    # label embeddings were recomputed with the loaded tower and the BPE ids
    with torch.no_grad():
        assert torch.allclose(m.category_head.text_embed, m.criterion.clip.build_text_embed([[l] for l in cfg.label]), atol=1e-5)
    del donor, m, tower

    # ---- run/infer.py's flow on the tree (XMASK3d(cfg) loads the files again, then the trainable checkpoint)
    logs = []
    torch.manual_seed(5)
    out = driver.infer(cfg, scenes=None, resume=ckpt, log=logs.append)
    assert any("ScanNet scene" in l and fx["data_root"] in l for l in logs), logs
    assert any(l.startswith("fused: hIoU") for l in logs)
    for name in ("fused", "2d", "3d"):
        for k, v in out[name].items():
            assert np.isfinite(v) and 0.0 <= v <= 1.0, (name, k, v)
