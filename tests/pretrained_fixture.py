"""Script-written stand-ins for the frozen nets' files, in the layouts the reference expects on disk (nothing here is a copy of a
released file; values are seeded random numbers):

    <root>/sd_model/sd-v1-3.ckpt                       torch.save({"state_dict": {...}, "global_step": 0}) with the Stable-Diffusion v1
                                                       key layout: model.diffusion_model.* (UNet), first_stage_model.* (VAE, plus the
                                                       loss.* keys a real checkpoint carries), cond_stage_model.transformer.text_model.*
                                                       (HuggingFace CLIP text encoder), model_ema.* noise; fp16 like the released file
    <root>/openai/clip-vit-large-patch14/pytorch_model.bin   HuggingFace CLIPModel layout (text_model.*, vision_model.*, *_projection, logit_scale)
    <root>/openai/clip-vit-large-patch14/vocab.json, merges.txt   a BPE vocabulary with CLIP's structure (256 byte symbols, 256 word-final
                                                       byte symbols, merges, <|startoftext|> = 49406, <|endoftext|> = 49407)
"""
import json
import os
from collections import OrderedDict

import torch

from xmask3d_amd import bpe as bpe_mod


def openai_to_hf(sd, prefix=""):
    """inverse of checkpoint.map_hf_clip_state_dict (test-side only)"""
    out = OrderedDict()

    def blocks(src, dst):
        n = len({k[len(src):].split(".")[0] for k in sd if k.startswith(src)})
        for i in range(n):
            s, d = f"{src}{i}.", f"{prefix}{dst}{i}."
            for kind in ("weight", "bias"):
                q, k, v = sd[s + "attn.in_proj_" + kind].chunk(3, 0)
                out[d + f"self_attn.q_proj.{kind}"], out[d + f"self_attn.k_proj.{kind}"], out[d + f"self_attn.v_proj.{kind}"] = q.clone(), k.clone(), v.clone()
                out[d + "self_attn.out_proj." + kind] = sd[s + "attn.out_proj." + kind]
                out[d + "layer_norm1." + kind] = sd[s + "ln_1." + kind]
                out[d + "layer_norm2." + kind] = sd[s + "ln_2." + kind]
                out[d + "mlp.fc1." + kind] = sd[s + "mlp.c_fc." + kind]
                out[d + "mlp.fc2." + kind] = sd[s + "mlp.c_proj." + kind]

    out[prefix + "text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out[prefix + "text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out[prefix + "text_model.embeddings.position_ids"] = torch.arange(77)[None]
    blocks("transformer.resblocks.", "text_model.encoder.layers.")
    out[prefix + "text_model.final_layer_norm.weight"], out[prefix + "text_model.final_layer_norm.bias"] = sd["ln_final.weight"], sd["ln_final.bias"]
    if "visual.conv1.weight" in sd:
        out[prefix + "text_projection.weight"] = sd["text_projection"].t().contiguous()
        out[prefix + "vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
        out[prefix + "vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
        out[prefix + "vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
        for kind in ("weight", "bias"):
            out[prefix + "vision_model.pre_layrnorm." + kind] = sd["visual.ln_pre." + kind]
            out[prefix + "vision_model.post_layernorm." + kind] = sd["visual.ln_post." + kind]
        blocks("visual.transformer.resblocks.", "vision_model.encoder.layers.")
        out[prefix + "visual_projection.weight"] = sd["visual.proj"].t().contiguous()
        out[prefix + "logit_scale"] = sd["logit_scale"]
    return out


def write_vocab(d, n_merges=600):
    """vocab.json + merges.txt with CLIP's structure; merges of common English letter pairs, the rest of the 49408 ids unused"""
    os.makedirs(d, exist_ok=True)
    b2u = list(bpe_mod.bytes_to_unicode().values())
    vocab = b2u + [c + "</w>" for c in b2u]
    merges = []
    words = "the of and a in to is room wall floor chair table door window bed sofa cabinet picture counter desk curtain sink toilet bathtub shower refrigerator bookshelf seen from frame other furniture".split()
    have = set(vocab)
    for w in words:  # merge each word left to right, so that it ends up as one token
        sym = list(w[:-1]) + [w[-1] + "</w>"]
        while len(sym) > 1 and len(merges) < n_merges:
            a, b = sym[0], sym[1]
            if a + b not in have:
                merges.append((a, b))
                have.add(a + b)
                vocab.append(a + b)
            sym = [a + b] + sym[2:]
    vocab += [f"<|unused{i}|>" for i in range(49406 - len(vocab))] + [bpe_mod.SOT_TOKEN, bpe_mod.EOT_TOKEN]
    assert len(vocab) == 49408
    with open(os.path.join(d, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump({t: i for i, t in enumerate(vocab)}, f, ensure_ascii=False)
    with open(os.path.join(d, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    return merges


def write_pretrained(root, model, dtype=torch.float16):
    """dump the frozen nets of `model` (an XMASK3d with seeded random weights) in the reference's file layouts"""
    ldm = model.backbone.feature_extractor.ldm_extractor.ldm
    clip = model.criterion.clip.clip
    sd = OrderedDict()
    for k, v in ldm.unet_model.state_dict().items():
        sd["model.diffusion_model." + k] = v.detach().to("cpu", dtype)
        if k.endswith("time_embed.0.weight"):
            sd["model_ema.diffusion_model" + k.replace(".", "")] = v.detach().to("cpu", dtype)  # EMA copy: ignored by the loader
    for k, v in ldm.first_stage_model.state_dict().items():
        sd["first_stage_model." + k] = v.detach().to("cpu", dtype)
    sd["first_stage_model.loss.logvar"] = torch.zeros(())
    text = {k: v.detach().cpu().float() for k, v in clip.state_dict().items() if not k.startswith("visual.") and k not in ("logit_scale", "text_projection")}
    for k, v in openai_to_hf(text, "cond_stage_model.transformer.").items():
        sd[k] = v if v.dtype == torch.long else v.to(dtype)
    os.makedirs(os.path.join(root, "sd_model"), exist_ok=True)
    torch.save({"state_dict": sd, "global_step": 0}, os.path.join(root, "sd_model", "sd-v1-3.ckpt"))
    d = os.path.join(root, "openai", "clip-vit-large-patch14")
    os.makedirs(d, exist_ok=True)
    full = {k: v.detach().cpu().float() for k, v in clip.state_dict().items()}
    torch.save(openai_to_hf(full), os.path.join(d, "pytorch_model.bin"))
    write_vocab(d)
    return dict(sd_checkpoint=os.path.join(root, "sd_model", "sd-v1-3.ckpt"), clip_dir=os.path.join(root, "openai"))
