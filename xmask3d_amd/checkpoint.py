"""Checkpoint load / save with the reference's file layout (SURVEY.md §5 "Checkpoint / resume", §8f rank 1).

* ``save_checkpoint`` / ``load_checkpoint``: ``{epoch, state_dict, optimizer, best_iou}`` in ``model_last.pth.tar``
  (/root/reference/util/util.py:17-21, run/train.py:354-390) and the loader semantics of
  ``XMask3dCheckpointer.load`` (/root/reference/models/checkpoint/odise_checkpointer.py:132-160): DDP ``module.``
  prefix stripped on either side, optimizer state restored unless ``eval``, keys of the frozen SD / CLIP nets may be
  missing (their ``state_dict()`` is empty by design: helper.py:38-39, clip.py:105-106).
  Files are read with ``torch.load(..., weights_only=True)`` only.
* ``map_sd_state_dict``: key mapping of a Stable-Diffusion v1 checkpoint (``model.diffusion_model.*``,
  ``first_stage_model.*``; ldm.py:112-114 loads ``sd_model/sd-v1-3.ckpt``) onto ``xmask3d_amd.sd_model``.
* ``map_openclip_state_dict``: identity mapping check for OpenAI ViT-L/14 weights in open_clip layout onto
  ``xmask3d_amd.clip_model.CLIP`` (clip.py:69-73).
Neither checkpoint exists offline; the mappings are exercised with synthetic state dicts in tests/test_checkpoint.py.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch


def strip_prefix(state_dict, prefix="module."):
    if state_dict and all(k.startswith(prefix) for k in state_dict):
        return OrderedDict((k[len(prefix):], v) for k, v in state_dict.items())
    return state_dict


def save_checkpoint(path, model, optimizer, epoch, best_iou):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({"epoch": epoch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict() if optimizer else None,
                "best_iou": best_iou}, path)


def load_checkpoint(path, model, optimizer=None, eval=False, map_location="cpu"):
    """-> dict(start_epoch, best_iou, missing, unexpected).  Raises on shape mismatches and on missing trainable keys."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    sd = strip_prefix(ckpt["state_dict"])
    target = model.module if hasattr(model, "module") else model
    own = target.state_dict()
    bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if bad:
        raise RuntimeError(f"checkpoint tensors with the wrong shape: {bad[:5]}")
    res = target.load_state_dict(sd, strict=False)
    frozen = ("ldm_extractor.", ".clip.clip.", "clip_head.", "category_head.clip.")
    missing = [k for k in res.missing_keys if not any(f in k for f in frozen)]
    if missing:
        raise RuntimeError(f"checkpoint lacks trainable parameters: {missing[:8]} (+{max(0, len(missing) - 8)} more)")
    if not eval and optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return {"start_epoch": ckpt["epoch"], "best_iou": ckpt["best_iou"], "missing": res.missing_keys, "unexpected": res.unexpected_keys}


def map_sd_state_dict(sd):
    """SD-v1 checkpoint keys -> (vae_state, unet_state) for sd_model.AutoencoderKL / UNetModel (names already follow ldm)."""
    sd = sd.get("state_dict", sd)
    vae, unet = OrderedDict(), OrderedDict()
    for k, v in sd.items():
        if k.startswith("first_stage_model."):
            k2 = k[len("first_stage_model."):]
            if k2.startswith("loss."):
                continue  # discriminator / lpips of the VAE training setup
            vae[k2] = v
        elif k.startswith("model.diffusion_model."):
            unet[k[len("model.diffusion_model."):]] = v
    return vae, unet


def load_sd_checkpoint(path, ldm):
    """ldm: image_branch.LatentDiffusion.  Strict on the UNet / VAE keys the extractor uses."""
    vae, unet = map_sd_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    r1 = ldm.first_stage_model.load_state_dict(vae, strict=True)
    r2 = ldm.unet_model.load_state_dict(unet, strict=True)
    return r1, r2


def map_openclip_state_dict(sd):
    """open_clip / OpenAI CLIP ViT-L/14 keys are used verbatim by clip_model.CLIP; drop buffers it recomputes."""
    return OrderedDict((k, v) for k, v in sd.items() if k not in ("attn_mask", "input_resolution", "context_length", "vocab_size"))


# ----------------------------------------------------------------------------- pretrained frozen nets (local files only)
def map_hf_clip_state_dict(sd, prefix=""):
    """HuggingFace CLIP layout (``text_model.*`` / ``vision_model.*`` / ``*_projection.weight`` / ``logit_scale``; the
    ``openai/clip-vit-large-patch14`` directory of the reference's README, and - text side only, under the prefix
    ``cond_stage_model.transformer.`` - the Stable-Diffusion checkpoint) -> OpenAI / open_clip layout of ``clip_model.CLIP``.
    q/k/v projections are concatenated into ``in_proj_*``, the two projection Linears are transposed into matrices."""
    out = OrderedDict()

    def blocks(src, dst):
        layers = sorted({int(k[len(prefix + src):].split(".")[0]) for k in sd if k.startswith(prefix + src)})
        for i in layers:
            s, d = f"{prefix}{src}{i}.", f"{dst}{i}."
            for kind in ("weight", "bias"):
                out[d + "attn.in_proj_" + kind] = torch.cat([sd[s + f"self_attn.{p}_proj.{kind}"] for p in "qkv"], 0)
                out[d + "attn.out_proj." + kind] = sd[s + "self_attn.out_proj." + kind]
                out[d + "ln_1." + kind] = sd[s + "layer_norm1." + kind]
                out[d + "ln_2." + kind] = sd[s + "layer_norm2." + kind]
                out[d + "mlp.c_fc." + kind] = sd[s + "mlp.fc1." + kind]
                out[d + "mlp.c_proj." + kind] = sd[s + "mlp.fc2." + kind]

    t = prefix + "text_model."
    if t + "embeddings.token_embedding.weight" in sd:
        out["token_embedding.weight"] = sd[t + "embeddings.token_embedding.weight"]
        out["positional_embedding"] = sd[t + "embeddings.position_embedding.weight"]
        blocks("text_model.encoder.layers.", "transformer.resblocks.")
        out["ln_final.weight"], out["ln_final.bias"] = sd[t + "final_layer_norm.weight"], sd[t + "final_layer_norm.bias"]
    if prefix + "text_projection.weight" in sd:
        out["text_projection"] = sd[prefix + "text_projection.weight"].t().contiguous()
    v = prefix + "vision_model."
    if v + "embeddings.class_embedding" in sd:
        out["visual.class_embedding"] = sd[v + "embeddings.class_embedding"]
        out["visual.conv1.weight"] = sd[v + "embeddings.patch_embedding.weight"]
        out["visual.positional_embedding"] = sd[v + "embeddings.position_embedding.weight"]
        pre = "pre_layrnorm" if v + "pre_layrnorm.weight" in sd else "pre_layernorm"  # (sic) the HF parameter name
        for kind in ("weight", "bias"):
            out["visual.ln_pre." + kind] = sd[v + pre + "." + kind]
            out["visual.ln_post." + kind] = sd[v + "post_layernorm." + kind]
        blocks("vision_model.encoder.layers.", "visual.transformer.resblocks.")
        out["visual.proj"] = sd[prefix + "visual_projection.weight"].t().contiguous()
    if prefix + "logit_scale" in sd:
        out["logit_scale"] = sd[prefix + "logit_scale"]
    return out


def _safe_load(path):
    """tensors only: safetensors, or torch.load(weights_only=True).  A file the safe loader refuses (TorchScript archive,
    pickled custom classes) is NOT opened any other way: the caller is told and continues without it."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu", weights_only=True)


def find_pretrained(cfg):
    """Paths of the frozen nets' files as the reference expects them relative to the working directory (ldm.py:112-114
    ``sd_model/sd-v1-3.ckpt``; README.md:28-35 ``openai/`` = the HuggingFace clip-vit-large-patch14 directory), overridable by
    the config keys ``sd_checkpoint`` / ``clip_dir`` and the root XM3D_PRETRAINED_ROOT.  Entries are None when absent."""
    root = os.environ.get("XM3D_PRETRAINED_ROOT", "")
    sd = os.path.join(root, str(getattr(cfg, "sd_checkpoint", None) or "sd_model/sd-v1-3.ckpt"))
    clip = os.path.join(root, str(getattr(cfg, "clip_dir", None) or "openai"))
    clip_file = None
    if os.path.isdir(clip):
        for d in [clip] + [os.path.join(clip, s) for s in sorted(os.listdir(clip)) if os.path.isdir(os.path.join(clip, s))]:
            for name in ("model.safetensors", "pytorch_model.bin", "open_clip_pytorch_model.bin", "ViT-L-14.pt"):
                if clip_file is None and os.path.isfile(os.path.join(d, name)):
                    clip_file = os.path.join(d, name)
    return {"sd": sd if os.path.isfile(sd) else None, "clip_dir": clip if os.path.isdir(clip) else None, "clip_file": clip_file}


class PretrainedSetError(RuntimeError):
    """the frozen nets' files on disk do not form a usable set (a file was refused by the safe loader, or only part of
    {SD checkpoint, CLIP weights, tokenizer vocabulary} is there): a run would silently mix released weights with seeded random ones"""


def pretrained_problems(rep):
    """-> list of reasons why the loaded set must not be used silently (empty = a complete, consistent set or nothing at all)"""
    probs = [f"refused by the safe (tensors-only) loader: {r}" for r in rep["refused"]]
    have = {k: bool(rep[k]) for k in ("sd", "clip", "tokenizer")}
    if any(have.values()) and not all(have.values()):
        missing = [k for k, v in have.items() if not v]
        loaded = [k for k, v in have.items() if v]
        probs.append(f"partial set: loaded {loaded}, missing {missing} - "
                     + ("real CLIP weights would be fed ids of the stand-in tokenizer; " if have["clip"] and not have["tokenizer"] else "")
                     + "the missing nets keep seeded random weights")
    return probs


def load_pretrained(model, cfg, log=None):
    """Fill the frozen nets of an XMASK3d from local files when they exist (no-op otherwise: seeded random weights and the
    stand-in tokenizer, as every synthetic run uses):
      * Stable-Diffusion checkpoint -> VAE + UNet (strict), as LdmCheckpointer(...).load("sd_model/sd-v1-3.ckpt"), ldm.py:112-114
      * CLIP ViT-L/14 weights (HuggingFace or OpenAI / open_clip state-dict layout) -> the shared ClipAdapter, clip.py:69-73
      * vocab.json + merges.txt (or open_clip's bpe_simple_vocab_16e6.txt.gz) -> the BPE tokenizer, clip.py:147-149
      * uncond_inputs = text encoder hidden states of "" (ldm.py:105), from the checkpoint's own ``cond_stage_model`` weights when
        it has them, else from the CLIP text tower (Stable Diffusion v1 uses exactly that frozen encoder)
    -> report dict (what was found and loaded; ``problems`` lists why the set is unusable).  A refused file or a partial set
    raises PretrainedSetError unless cfg.allow_partial_pretrained / XM3D_ALLOW_PARTIAL_PRETRAINED=1 (then: RuntimeWarning)."""
    from . import bpe as bpe_mod
    from .clip_model import TextTower

    log = log or (lambda *_: None)
    found = find_pretrained(cfg)
    rep = {"sd": None, "clip": None, "tokenizer": None, "uncond": None, "refused": []}
    ldm = model.backbone.feature_extractor.ldm_extractor.ldm
    adapter = model.criterion.clip
    tok = bpe_mod.ClipBPE.from_dir(found["clip_dir"]) if found["clip_dir"] else None
    if tok is not None:
        adapter.set_tokenizer(tok)
        rep["tokenizer"] = found["clip_dir"]
    text_sd = None
    if found["clip_file"]:
        try:
            sd = _safe_load(found["clip_file"])
            sd = sd.get("state_dict", sd)
            sd = map_hf_clip_state_dict(sd) if any(k.startswith("text_model.") or k.startswith("vision_model.") for k in sd) else map_openclip_state_dict(sd)
            res = adapter.clip.load_state_dict(sd, strict=False)
            bad = [k for k in res.missing_keys if k != "attn_mask"]
            if bad or res.unexpected_keys:
                raise RuntimeError(f"CLIP weights do not match ViT-L/14: missing {bad[:4]}, unexpected {res.unexpected_keys[:4]}")
            text_sd = {k: v for k, v in sd.items() if not k.startswith("visual.") and k != "logit_scale"}
            rep["clip"] = found["clip_file"]
        except Exception as e:  # noqa: BLE001  (unsafe / foreign file: say so, continue with what we have)
            if isinstance(e, RuntimeError) and "do not match" in str(e):
                raise
            rep["refused"].append(f"{found['clip_file']}: {type(e).__name__}: {str(e)[:200]}")
    if found["sd"]:
        try:
            raw = _safe_load(found["sd"])
        except Exception as e:  # noqa: BLE001
            raw = None
            rep["refused"].append(f"{found['sd']}: {type(e).__name__}: {str(e)[:200]}")
        if raw is not None:
            raw = raw.get("state_dict", raw)
            vae, unet = map_sd_state_dict(raw)
            ldm.first_stage_model.load_state_dict(vae, strict=True)
            ldm.unet_model.load_state_dict(unet, strict=True)
            rep["sd"] = found["sd"]
            cond = map_hf_clip_state_dict(raw, prefix="cond_stage_model.transformer.")
            if cond:
                text_sd = cond
    if text_sd is not None and tok is not None:
        tower = TextTower()
        tower.load_state_dict({k: v.float() for k, v in text_sd.items() if k != "text_projection"}, strict=True)
        with torch.no_grad():
            hidden = tower(tok([""], pad_id=tok.eot))  # HuggingFace padding = end-of-text id (FrozenCLIPEmbedder)
        ldm.uncond_inputs.copy_(hidden.to(ldm.uncond_inputs))
        rep["uncond"] = "text encoder hidden states of the empty prompt"
    if rep["clip"] or rep["tokenizer"]:
        model.category_head.refresh()  # label / null embeddings were computed by the constructor with the old tower
    model.set_dense_dtype(model.dense_dtype)  # freshly loaded fp32 tensors -> the dense compute dtype
    if any(rep[k] for k in ("sd", "clip", "tokenizer")):
        log(f"pretrained: SD {rep['sd']}, CLIP {rep['clip']}, tokenizer {rep['tokenizer']}, uncond_inputs {rep['uncond']}")
    model.pretrained_report = rep
    probs = pretrained_problems(rep)
    rep["problems"] = probs
    if probs:
        # refusing an unsafe file is right, carrying on silently is not: scores of released CLIP weights against a random SD UNet
        # look like a model failure.  Opt out explicitly (cfg.allow_partial_pretrained / XM3D_ALLOW_PARTIAL_PRETRAINED=1) to continue.
        msg = "pretrained files found but unusable as a set: " + "; ".join(probs)
        if getattr(cfg, "allow_partial_pretrained", False) or os.environ.get("XM3D_ALLOW_PARTIAL_PRETRAINED", "") == "1":
            import warnings

            warnings.warn(msg + " (continuing: allow_partial_pretrained)", RuntimeWarning, stacklevel=2)
            log("pretrained: WARNING " + msg)
        else:
            raise PretrainedSetError(msg + ".  Provide the whole set (tensors-only files: a plain state_dict .ckpt / .safetensors), remove the "
                                     "files, or set cfg.allow_partial_pretrained / XM3D_ALLOW_PARTIAL_PRETRAINED=1")
    return rep
