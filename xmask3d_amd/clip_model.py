"""CLIP ViT-L/14 (OpenAI) restated in plain torch + the XMask3D adapters around it
(SURVEY.md §8 rows a15-a17).

The reference reaches CLIP through ``open-clip-torch==2.0.2`` (/root/reference/setup.py:30), which is
absent here; the public ViT-L/14 architecture (visual: 24 layers, width 1024, 16 heads, patch 14,
QuickGELU, proj 1024->768; text: 12 layers, width 768, 12 heads, 77 tokens, causal mask) is written out
again with open_clip's parameter names (``visual.transformer.resblocks.N.attn.in_proj_weight`` ...).
Adapters mirror /root/reference/models/modeling/meta_arch/clip.py:
  ClipAdapter.embed_text :132-161, MaskCLIP.encode_image_with_mask :272-310, _mask_clip_forward :239-270,
  get_mask_embed / forward :312-348; build_clip_text_embed :21-63; CategoryEmbed: odise.py:600-700.
PARITY UNPINNED for CLIP numerics (no weights / tokenizer offline).  The BPE tokenizer is replaced by a
deterministic stand-in (byte hash -> ids) that keeps the sequence structure [SOT, ..., EOT, 0...] and so
the compute graph; with real weights plug the real tokenizer into ``ClipAdapter.tokenize``.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .sd_model import flinear, gemm_ok

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
SOT, EOT, VOCAB, CONTEXT = 49406, 49407, 49408, 77


class QuickGELU(nn.Module):
    def forward(self, x):
        if x.is_cuda and not torch.is_grad_enabled() and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() \
                and x.numel() % 8 == 0:
            return ops.quick_gelu(x)  # one pass (xm3d_quick_gelu) instead of three elementwise kernels
        return x * torch.sigmoid(1.702 * x)


def additive_mask(allow, dtype):
    """bool 'may attend' mask -> additive mask (0 / -inf) of `dtype`, contiguous"""
    return torch.zeros(allow.shape, dtype=dtype, device=allow.device).masked_fill_(~allow, float("-inf"))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads, mlp_ratio=4.0):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, int(width * mlp_ratio))), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(int(width * mlp_ratio), width))]))
        self.heads = heads

    def attention(self, x, allow):
        """x (B,T,C); allow: None | bool (T,T) | bool (B,1,T,T), True = may attend."""
        b, t, c = x.shape
        qkv = flinear(x, self.attn.in_proj_weight, self.attn.in_proj_bias).view(b, t, 3, self.heads, c // self.heads)
        if ops.attention_supported(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]):
            # bf16 inference: HIP flash attention straight on the packed qkv buffer; `allow` as an additive mask (built once
            # per forward by the caller through additive_mask(), shared by the 24 layers)
            bias = allow if (allow is None or allow.dtype != torch.bool) else additive_mask(allow, x.dtype)
            if bias is not None and bias.dim() == 2:
                bias = bias[None, None]
            o = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], bias=bias)
            return flinear(o.view(b, t, c), self.attn.out_proj.weight, self.attn.out_proj.bias)
        if ops.attention_f32_supported(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]):
            # fp32 configuration: the f32-accurate flash attention on the packed qkv buffer
            bias = allow if (allow is None or allow.dtype != torch.bool) else additive_mask(allow, x.dtype)
            if bias is not None and bias.dim() == 2:
                bias = bias[None, None]
            o = ops.attention_f32(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], bias=bias)
            return flinear(o.view(b, t, c), self.attn.out_proj.weight, self.attn.out_proj.bias)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        if allow is not None and allow.dtype != torch.bool:
            allow = allow.to(q.dtype)
            if allow.dim() == 2:
                allow = allow[None, None]
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=allow)
        return flinear(o.transpose(1, 2).reshape(b, t, c), self.attn.out_proj.weight, self.attn.out_proj.bias)

    def forward(self, x, allow=None):
        n1, n2 = self.ln_1, self.ln_2
        if ops.layer_norm_supported(x, x.shape[-1]) and n1.weight.dtype == x.dtype:
            # inference: HIP LayerNorm (xm3d_layer_norm); the residual add in front of ln_2 rides in its kernel
            a = self.attention(ops.layer_norm(x, n1.weight, n1.bias, n1.eps), allow)
            h, x = ops.layer_norm(x, n2.weight, n2.bias, n2.eps, delta=a.contiguous(), want_sum=True)
            return self._mlp_res(h, x)
        x = x + self.attention(self.ln_1(x), allow)
        return self._mlp_res(self.ln_2(x), x)

    def _mlp_res(self, h, x):
        """x + c_proj(QuickGELU(c_fc(h))): on the own GEMM kernels - activation and residual in the epilogues - where gemm_ok() says so"""
        fc, pj = self.mlp.c_fc, self.mlp.c_proj
        if gemm_ok(h, fc.out_features, "quick_gelu") and gemm_ok(h, pj.out_features, None, True) and fc.weight.dtype == h.dtype:
            return flinear(flinear(h, fc.weight, fc.bias, act="quick_gelu"), pj.weight, pj.bias, residual=x)
        return x + self.mlp(h)


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList(ResidualAttentionBlock(width, heads) for _ in range(layers))

    def forward(self, x, allow=None):
        if allow is not None and allow.dtype == torch.bool and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and not torch.is_grad_enabled():
            allow = additive_mask(allow, x.dtype)  # once for all layers (the HIP attention takes the additive form)
        for blk in self.resblocks:
            x = blk(x, allow)
        return x


class VisualTransformer(nn.Module):
    def __init__(self, image_size=224, patch_size=14, width=1024, layers=24, heads=16, output_dim=768):
        super().__init__()
        self.image_size = (image_size, image_size)
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((image_size // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class CLIP(nn.Module):
    def __init__(self, embed_dim=768, vision_layers=24, vision_width=1024, vision_heads=16, text_layers=12, text_width=768,
                 text_heads=12):
        super().__init__()
        self.context_length = CONTEXT
        self.visual = VisualTransformer(224, 14, vision_width, vision_layers, vision_heads, embed_dim)
        self.transformer = Transformer(text_width, text_layers, text_heads)
        self.token_embedding = nn.Embedding(VOCAB, text_width)
        self.positional_embedding = nn.Parameter(torch.empty(CONTEXT, text_width))
        self.ln_final = nn.LayerNorm(text_width)
        self.text_projection = nn.Parameter(torch.empty(text_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.register_buffer("attn_mask", torch.ones(CONTEXT, CONTEXT, dtype=torch.bool).tril(), persistent=False)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=text_width ** -0.5)


class TextTower(nn.Module):
    """the text side of CLIP ViT-L/14 alone (same parameter names): what ldm's FrozenCLIPEmbedder runs for Stable Diffusion's
    conditioning - token + positional embedding, 12 causal blocks, final LayerNorm; forward(tokens) -> hidden states (n, 77, 768)
    (HuggingFace ``last_hidden_state``).  Used once, on the CPU, to compute ``uncond_inputs`` (ldm.py:105)."""

    def __init__(self, width=768, layers=12, heads=12):
        super().__init__()
        self.transformer = Transformer(width, layers, heads)
        self.token_embedding = nn.Embedding(VOCAB, width)
        self.positional_embedding = nn.Parameter(torch.zeros(CONTEXT, width))
        self.ln_final = nn.LayerNorm(width)
        self.register_buffer("attn_mask", torch.ones(CONTEXT, CONTEXT, dtype=torch.bool).tril(), persistent=False)

    def forward(self, tokens):
        x = self.token_embedding(tokens) + self.positional_embedding
        return self.ln_final(self.transformer(x, self.attn_mask))


def tokenize_standin(texts, context_length=CONTEXT):
    """Deterministic stand-in for open_clip.tokenize: one id per whitespace token (crc32 into the vocab)."""
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [SOT] + [1 + zlib.crc32(w.encode()) % (SOT - 1) for w in t.lower().split()][: context_length - 2] + [EOT]
        out[i, : len(ids)] = torch.tensor(ids)
    return out


class ClipAdapter(nn.Module):
    def __init__(self, name="ViT-L-14", normalize=True):
        super().__init__()
        if name != "ViT-L-14":
            raise NotImplementedError("every XMask3D config uses clip_name ViT-L-14")
        self.clip = CLIP()
        self.name, self.normalize = name, normalize
        self.tokenize = tokenize_standin
        self.register_buffer("_mean", torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), False)
        self.register_buffer("_std", torch.tensor(CLIP_STD).view(1, 3, 1, 1), False)
        self._freeze()

    def _freeze(self):
        self.clip.eval()
        for p in self.clip.parameters():
            p.requires_grad = False

    def set_tokenizer(self, bpe):
        """a bpe.ClipBPE built from local vocabulary files replaces the stand-in (open_clip.tokenize semantics: pad with 0)"""
        self.tokenize = lambda texts, context_length=CONTEXT: bpe(texts, context_length=context_length, pad_id=0)
        self.bpe = bpe

    def train(self, mode=True):
        super().train(mode)
        self._freeze()
        return self

    def state_dict(self, *a, **k):  # frozen: contributes no checkpoint keys (clip.py:105-106)
        return OrderedDict()

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dim_latent(self):
        return self.clip.text_projection.shape[-1]

    @property
    def image_size(self):
        return self.clip.visual.image_size

    def clip_preprocess(self, image):
        """Resize(224, bicubic) + CenterCrop(224) are identities on the 224x224 input MaskCLIP feeds; normalise."""
        return (image - self._mean.to(image.dtype)) / self._std.to(image.dtype)

    def _encode_text(self, text):
        c = self.clip
        x = c.token_embedding(text) + c.positional_embedding
        x = c.ln_final(c.transformer(x, c.attn_mask))
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ c.text_projection, x

    @torch.no_grad()
    def embed_text(self, captions):
        text = self.tokenize(list(captions) if not isinstance(captions, str) else [captions]).to(self.device)
        emb, enc = self._encode_text(text)
        emb = emb.float()
        return F.normalize(emb, dim=-1) if self.normalize else emb

    @torch.no_grad()
    def build_text_embed(self, labels):
        """labels: list of synonym lists; rows are L2-normalised text embeddings (clip.py:21-63)."""
        if isinstance(labels, str):
            labels = [[labels]]
        elif labels and isinstance(labels[0], str):
            labels = [[t] for t in labels]
        flat = [t for syn in labels for t in syn]
        emb, _ = self._encode_text(self.tokenize(flat).to(self.device))
        return F.normalize(emb.float(), dim=-1)


class MaskCLIP(ClipAdapter):
    def __init__(self, name="ViT-L-14"):
        super().__init__(name=name, normalize=False)

    @property
    def logit_scale(self):
        return torch.clamp(self.clip.logit_scale.exp(), max=100)

    def encode_image_with_mask(self, image, mask, blocked=None):
        """blocked (b, q, patches) bool: the patch mask when the caller has it already (ops.clip_mask_blocked); `mask` is then only read for its shape"""
        v = self.clip.visual
        image = self.clip_preprocess(image.float())
        b, q = mask.shape[:2]
        patch = v.conv1.kernel_size
        if blocked is None:
            blocked = (F.max_pool2d(mask.float().sigmoid(), kernel_size=patch, stride=v.conv1.stride) < 0.5).reshape(b, q, -1)
        n_img = v.positional_embedding.shape[0] - 1
        total = q + 1 + n_img
        # allow[b, row, col]: nobody attends to mask tokens; a mask token sees the class token and its own patches
        allow = torch.ones(b, total, total, dtype=torch.bool, device=image.device)
        allow[:, :, :q] = False
        allow[:, :q, q + 1:] = ~blocked
        x = v.conv1(image.to(v.conv1.weight.dtype)).flatten(2).transpose(1, 2)
        x = torch.cat([v.class_embedding.to(x.dtype).expand(b, 1, -1), x], dim=1) + v.positional_embedding.to(x.dtype)
        x = v.ln_pre(x)
        x = torch.cat([x[:, :1].expand(-1, q, -1), x], dim=1)
        x = v.transformer(x, allow[:, None])
        x = v.ln_post(x[:, :q])
        return x @ v.proj

    def get_mask_embed(self, image, mask):
        image = F.interpolate(image, size=self.image_size, mode="bilinear", align_corners=False)
        v = self.clip.visual
        S, P = image.shape[-1], v.conv1.kernel_size[0]
        if mask.is_cuda and not torch.is_grad_enabled() and mask.dtype == torch.float32 and mask.is_contiguous() and image.shape[-2] == S \
                and tuple(v.conv1.kernel_size) == tuple(v.conv1.stride) == (P, P) and S % P == 0 and os.environ.get("XM3D_CLIP_MASK", "hip") != "library":
            # inference: the patch mask straight from the mask logits (resize + sigmoid + max-pool + compare in one pass, no (B, Q, S, S) tensor)
            from . import ops

            return self.encode_image_with_mask(image, mask, blocked=ops.clip_mask_blocked(mask, S, P))
        mask = F.interpolate(mask, size=image.shape[-2:], mode="bilinear", align_corners=False)
        return self.encode_image_with_mask(image, mask)

    def forward(self, image, mask):
        return {"mask_embed_clip": self.get_mask_embed(image, mask)}


class CategoryEmbed(nn.Module):
    def __init__(self, labels, test_labels, projection_dim, clip_model_name="ViT-L-14", prompt=None):
        super().__init__()
        self.labels, self.test_labels = labels, test_labels
        self.clip = ClipAdapter(clip_model_name, normalize=False) if isinstance(clip_model_name, str) else clip_model_name
        self.text_proj = nn.Identity() if projection_dim < 0 else nn.Linear(self.clip.dim_latent, projection_dim)
        self.register_buffer("text_embed", self.clip.build_text_embed(labels), False)
        self.null_embed = nn.Parameter(self.clip.build_text_embed(""))
        self._test_cache = {}

    @torch.no_grad()
    def refresh(self):
        """recompute what the constructor derived from the CLIP text tower (after its weights / tokenizer were replaced)"""
        self.text_embed.copy_(self.clip.build_text_embed(self.labels).to(self.text_embed))
        self.null_embed.copy_(self.clip.build_text_embed("").to(self.null_embed))
        self._test_cache = {}

    def forward(self, outputs=None, targets=None):
        if self.training:
            return {"text_embed": self.text_proj(self.text_embed), "null_embed": self.text_proj(self.null_embed),
                    "labels": self.labels}
        dev = self.null_embed.device
        key = (tuple(tuple(l) for l in self.test_labels), str(dev))  # per device: a deep copy moved to another device must not
        if key not in self._test_cache:                              # upload a cached CPU tensor inside a HIP-graph capture
            self._test_cache[key] = self.clip.build_text_embed(self.test_labels).to(dev)
        te = self._test_cache[key]
        return {"text_embed": self.text_proj(te), "null_embed": self.text_proj(self.null_embed), "labels": self.test_labels}
