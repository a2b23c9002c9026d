"""CPU: checkpoint layout / loader semantics and the SD / CLIP key mappings (synthetic state dicts)."""
import os

import pytest
import torch

from xmask3d_amd import checkpoint as ck


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.pc_decoder = torch.nn.Linear(3, 4)
        self.fuser = torch.nn.Linear(4, 2)


def test_roundtrip_with_ddp_prefix_and_optimizer(tmp_path):
    torch.manual_seed(0)
    m = _Tiny()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    m(torch.randn(2, 3)) if False else None
    loss = m.fuser(m.pc_decoder(torch.randn(5, 3))).sum()
    loss.backward()
    opt.step()
    path = os.path.join(tmp_path, "model", "model_last.pth.tar")
    ck.save_checkpoint(path, m, opt, epoch=7, best_iou=0.5)
    # a DDP-saved file has the "module." prefix on every key
    raw = torch.load(path, weights_only=True)
    raw["state_dict"] = {"module." + k: v for k, v in raw["state_dict"].items()}
    torch.save(raw, path)
    m2 = _Tiny()
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    info = ck.load_checkpoint(path, m2, opt2)
    assert info["start_epoch"] == 7 and info["best_iou"] == 0.5
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    assert opt2.state_dict()["state"][0]["step"] == opt.state_dict()["state"][0]["step"]


def test_missing_trainable_keys_raise_and_frozen_keys_do_not(tmp_path):
    m = _Tiny()
    path = os.path.join(tmp_path, "x.pth.tar")
    sd = {k: v for k, v in m.state_dict().items() if not k.startswith("fuser")}
    torch.save({"epoch": 1, "state_dict": sd, "optimizer": None, "best_iou": 0.0}, path)
    with pytest.raises(RuntimeError, match="lacks trainable"):
        ck.load_checkpoint(path, _Tiny(), eval=True)
    sd = dict(m.state_dict())
    sd["pc_decoder.weight"] = torch.zeros(9, 9)
    torch.save({"epoch": 1, "state_dict": sd, "optimizer": None, "best_iou": 0.0}, path)
    with pytest.raises(RuntimeError, match="wrong shape"):
        ck.load_checkpoint(path, _Tiny(), eval=True)


def test_sd_key_mapping_covers_every_parameter_of_the_restated_nets():
    """A synthetic 'sd-v1' state dict with ldm's prefixes maps 1:1 onto the restated VAE / UNet (built on the meta device)."""
    from xmask3d_amd.sd_model import AutoencoderKL, UNetModel

    with torch.device("meta"):
        vae, unet = AutoencoderKL(), UNetModel()
    fake = {"first_stage_model." + k: v for k, v in vae.state_dict().items()}
    fake.update({"model.diffusion_model." + k: v for k, v in unet.state_dict().items()})
    fake["first_stage_model.loss.logvar"] = torch.zeros(1)
    fake["cond_stage_model.transformer.text_model.embeddings.position_ids"] = torch.zeros(1, 77)
    v2, u2 = ck.map_sd_state_dict({"state_dict": fake})
    assert set(v2) == set(vae.state_dict()) and set(u2) == set(unet.state_dict())
    # spot-check names that must exist in a real SD-v1 file
    for k in ("encoder.down.0.block.0.norm1.weight", "encoder.mid.attn_1.q.weight", "decoder.up.3.upsample.conv.weight", "quant_conv.weight"):
        assert k in v2
    for k in ("input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight", "time_embed.0.weight", "output_blocks.11.0.skip_connection.weight",
              "middle_block.1.proj_in.weight", "input_blocks.3.0.op.weight", "output_blocks.2.1.conv.weight", "out.2.weight"):
        assert k in u2


def test_clip_key_names_follow_open_clip():
    from xmask3d_amd.clip_model import CLIP

    with torch.device("meta"):
        clip = CLIP()
    keys = set(clip.state_dict())
    for k in ("visual.conv1.weight", "visual.class_embedding", "visual.positional_embedding", "visual.ln_pre.weight",
              "visual.transformer.resblocks.23.attn.in_proj_weight", "visual.transformer.resblocks.0.mlp.c_fc.weight",
              "visual.ln_post.bias", "visual.proj", "transformer.resblocks.11.mlp.c_proj.bias", "token_embedding.weight",
              "positional_embedding", "ln_final.weight", "text_projection", "logit_scale"):
        assert k in keys
    assert set(ck.map_openclip_state_dict({**{k: 0 for k in keys}, "attn_mask": 0})) == keys


class _NotATensor:  # what a Lightning checkpoint's "callbacks" entry pickles (a ModelCheckpoint object)
    def __init__(self):
        self.best = 1.0


def _stub_model():
    import types

    ldm = types.SimpleNamespace()
    m = types.SimpleNamespace(dense_dtype=torch.float32)
    m.backbone = types.SimpleNamespace(feature_extractor=types.SimpleNamespace(ldm_extractor=types.SimpleNamespace(ldm=ldm)))
    m.criterion = types.SimpleNamespace(clip=types.SimpleNamespace())
    m.set_dense_dtype = lambda d: m
    return m


def test_a_refused_frozen_net_file_is_an_error_not_a_silent_random_net(tmp_path, monkeypatch):
    """sd_model/sd-v1-3.ckpt is a Lightning checkpoint whose `callbacks` entry pickles an object: the tensors-only loader
    refuses it.  That must not end in a silent run on a seeded random SD UNet: load_pretrained raises, or - with the explicit
    opt-out - warns and records the problem in the report the driver prints"""
    import types
    import warnings

    from xmask3d_amd import checkpoint

    (tmp_path / "sd_model").mkdir()
    torch.save({"state_dict": {"w": torch.zeros(2)}, "callbacks": {"ckpt": _NotATensor()}}, tmp_path / "sd_model" / "sd-v1-3.ckpt")
    monkeypatch.setenv("XM3D_PRETRAINED_ROOT", str(tmp_path))
    monkeypatch.delenv("XM3D_ALLOW_PARTIAL_PRETRAINED", raising=False)
    cfg = types.SimpleNamespace()
    assert checkpoint.find_pretrained(cfg)["sd"] is not None
    with pytest.raises(checkpoint.PretrainedSetError, match="refused by the safe"):
        checkpoint.load_pretrained(_stub_model(), cfg)
    cfg.allow_partial_pretrained = True
    lines = []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = _stub_model()
        rep = checkpoint.load_pretrained(m, cfg, log=lines.append)
    assert rep["sd"] is None and len(rep["refused"]) == 1 and rep["problems"] and m.pretrained_report is rep
    assert any(issubclass(x.category, RuntimeWarning) for x in w) and any("WARNING" in l for l in lines)


def test_partial_pretrained_sets_are_named():
    from xmask3d_amd import checkpoint

    ok = {"sd": "a", "clip": "b", "tokenizer": "c", "uncond": "x", "refused": []}
    assert checkpoint.pretrained_problems(ok) == []
    assert checkpoint.pretrained_problems({"sd": None, "clip": None, "tokenizer": None, "uncond": None, "refused": []}) == []
    p = checkpoint.pretrained_problems(dict(ok, tokenizer=None))
    assert len(p) == 1 and "stand-in tokenizer" in p[0] and "tokenizer" in p[0]
    p = checkpoint.pretrained_problems(dict(ok, sd=None))
    assert len(p) == 1 and "'sd'" in p[0]
