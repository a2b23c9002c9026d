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
