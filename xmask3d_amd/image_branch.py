"""3D-conditioned Stable-Diffusion feature extractor + projection backbone.

Mirrors, with the same attribute / parameter names so released checkpoints map 1:1:
  * LdmExtractor                     /root/reference/models/modeling/meta_arch/ldm.py:209-571
  * PositionalLinear, LdmImplicitCaptionerExtractor            ldm.py:574-676
  * FeatureExtractorBackbone         /root/reference/models/modeling/backbone/feature_extractor.py:20-234
    (its GN bottlenecks are detectron2 ``BottleneckBlock(norm="GN")``: conv{1,2,3}.weight + conv*.norm.*, shortcut.*)

Differences on purpose (documented in DESIGN.md):
  * ``prune_dead_compute`` (default True): the VAE decoder stops at the last tapped block and the UNet
    stops before output_blocks[11]/out - the reference computes and discards them.
    ``prune_dead_compute=False`` runs everything the reference runs.
  * frozen SD weights never get requires_grad flipped on: only gradients w.r.t. the
    conditioning inputs flow.
  * the frozen text encoder's embedding of "" (``uncond_inputs``) is a registered buffer; without the
    SD checkpoint it is filled deterministically (seed 0), like the random-init weights.
"""
from __future__ import annotations

import math
from collections import defaultdict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .sd_model import ACT_NONE, ACT_RELU, AutoencoderKL, GroupNorm, UNetModel, _gn_f32, _packed, _packed_split, fused_conv_ok, gn_act, own_conv

SCALE_FACTOR = 0.18215
# sqrt(alphas_cumprod[0]) and sqrt(1 - alphas_cumprod[0]) of the "ldm_linear" schedule
# (models/modeling/diffusion/gaussian_diffusion.py:61-90,190-199; pinned by tests/golden/diffusion.npz)
SQRT_AC0 = 0.9995749096490968
SQRT_1M_AC0 = 0.029154759474226803


class LatentDiffusion(nn.Module):
    """Holder with the reference's attribute names: .encoder / .unet / .decoder, pixel_mean/std, uncond_inputs."""

    def __init__(self):
        super().__init__()
        self.first_stage_model = AutoencoderKL()
        self.unet_model = UNetModel()
        self.image_size, self.latent_image_size, self.latent_dim = (512, 512), (64, 64), 4
        g = torch.Generator().manual_seed(0)
        self.register_buffer("uncond_inputs", torch.randn(1, 77, 768, generator=g, device="cpu") * 0.5)
        self.register_buffer("pixel_mean", torch.tensor([0.5, 0.5, 0.5]).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor([0.5, 0.5, 0.5]).view(-1, 1, 1), False)

    @property
    def encoder(self):
        return self.first_stage_model.encoder

    @property
    def decoder(self):
        return self.first_stage_model.decoder

    @property
    def unet(self):
        return self.unet_model


class _GraphedTaps:
    """fn(latent, ctx, emb) -> tuple of tensors, differentiable w.r.t. ctx and emb, replayed as ONE forward and ONE backward HIP
    graph.  Same idea as torch.cuda.make_graphed_callables, but every tensor autograd sees during warm-up and capture - the
    static input leaves included - is CREATED ON THE PRIVATE CAPTURE STREAM: with make_graphed_callables the sample leaves were
    made on the caller's (default) stream, so their AccumulateGrad nodes belonged to that stream while the capture ran on
    another one ("The AccumulateGrad node's stream does not match ... may break CUDA graph capture" in every log of round 2):
    the engine then orders the capture stream against the legacy stream inside the capture.  One backward replay of such a
    pair ended in a segmentation fault in hipGraphLaunch (round-2 GPU test log).  Here nothing in either capture refers to
    another stream, the private memory pool and both graphs live as long as this object, and the gradients handed back are
    copies (the static ones are overwritten by the next replay)."""

    def __init__(self, fn, latent, ctx, emb, warmup=3):
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.lat = latent.detach().clone()
            self.ctx = ctx.detach().clone().requires_grad_(True)
            self.emb = emb.detach().clone().requires_grad_(True)
            for _ in range(warmup):  # library algorithm selection and allocator warm-up, outside any capture
                outs = fn(self.lat, self.ctx, self.emb)
                torch.autograd.grad(outs, (self.ctx, self.emb), [torch.ones_like(o) for o in outs])
            del outs
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.pool = torch.cuda.graph_pool_handle()
        self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=self.pool, stream=self.stream):
            self.outs = tuple(fn(self.lat, self.ctx, self.emb))
        with torch.cuda.stream(self.stream):
            self.gouts = [torch.zeros_like(o) for o in self.outs]
        self.stream.synchronize()
        with torch.cuda.graph(self.bwd, pool=self.pool, stream=self.stream):
            self.gins = torch.autograd.grad(self.outs, (self.ctx, self.emb), self.gouts, only_inputs=True)
        torch.cuda.synchronize()

    def __call__(self, latent, ctx, emb):
        return _GraphedTapsFn.apply(self, latent, ctx, emb)


class _GraphedTapsFn(torch.autograd.Function):
    @staticmethod
    def forward(c, g, latent, ctx, emb):
        with torch.no_grad():
            g.lat.copy_(latent)
            g.ctx.copy_(ctx)
            g.emb.copy_(emb)
        g.fwd.replay()
        c.g = g
        g.generation = getattr(g, "generation", 0) + 1
        c.generation = g.generation  # the backward graph reads the activations THIS replay left in the pool
        return tuple(o.detach().clone() for o in g.outs)  # the static outputs are overwritten by the next replay

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(c, *grads):
        g = c.g
        if c.generation != g.generation:
            raise RuntimeError("graphed UNet: backward of a forward whose activations a later forward of the same shape has overwritten "
                               "(two forwards before their backwards: gradient accumulation over views needs enable_train_graph(False))")
        for dst, src in zip(g.gouts, grads):
            if src is None:
                dst.zero_()
            else:
                dst.copy_(src)
        g.bwd.replay()
        return None, None, g.gins[0].detach().clone(), g.gins[1].detach().clone()


class LdmExtractor(nn.Module):
    def __init__(self, encoder_block_indices=(5, 7), unet_block_indices=(2, 5, 8, 11), decoder_block_indices=(2, 5),
                 steps=(0,), prune_dead_compute=True):
        super().__init__()
        assert tuple(steps) == (0,), "XMask3D uses the single step t=0"
        self.encoder_block_indices, self.unet_block_indices = tuple(encoder_block_indices), tuple(unet_block_indices)
        self.decoder_block_indices, self.steps = tuple(decoder_block_indices), tuple(steps)
        self.prune_dead_compute = prune_dead_compute
        self.ldm = LatentDiffusion()
        rng = torch.Generator().manual_seed(42)
        self.register_buffer("shared_noise", torch.randn(1, 4, 64, 64, generator=rng, device="cpu"))
        for p in self.parameters():
            p.requires_grad = False

    def train(self, mode=True):  # frozen: always eval (helper.py:41-49)
        return super().train(False)

    def state_dict(self, *a, **k):  # frozen nets contribute no checkpoint keys (helper.py:38-39)
        from collections import OrderedDict

        return OrderedDict()

    @property
    def feature_dims(self):
        enc = self.ldm.encoder
        blocks = [b for lvl in enc.down for b in lvl.block]
        dims = [blocks[i].in_channels for i in self.encoder_block_indices]
        dims += [self.ldm.unet.output_blocks[i][0].channels for i in self.unet_block_indices]
        dblocks = [b for i in reversed(range(self.ldm.decoder.num_resolutions)) for b in self.ldm.decoder.up[i].block]
        dims += [dblocks[i].in_channels for i in self.decoder_block_indices]
        return dims

    @property
    def feature_strides(self):
        s = [2 ** ((i + 2) // 2 - 1) for i in self.encoder_block_indices]
        s += [64 // (2 ** ((i + 3) // 3 - 1)) for i in self.unet_block_indices]
        s += [8 // (2 ** ((i + 3) // 3 - 1)) for i in self.decoder_block_indices]
        return s

    @property
    def grouped_indices(self):
        n = len(self.encoder_block_indices) + len(self.unet_block_indices) + len(self.decoder_block_indices)
        return [[i] for i in range(n)]

    @torch.no_grad()  # frozen weights, image-only input: nothing upstream needs a gradient (fused inference kernels apply)
    def encode(self, img):
        """VAE encoder stage: img (B,3,H,W) in [0,1] -> (latent (B,4,H/8,W/8), tapped encoder features).  Independent of the
        3D conditioning, so it can run concurrently with the sparse 3D branch."""
        if self._vae_graph_ok(img):
            return self._vae_graphed("enc", self._encode, img)
        return self._encode(img)

    def _encode(self, img):
        ldm = self.ldm
        x = (img - ldm.pixel_mean.to(img.dtype)) / ldm.pixel_std.to(img.dtype)
        if self._vae_train_nhwc:
            x = x.contiguous(memory_format=torch.channels_last)
        moments, enc_feats = ldm.encoder(x, taps=self.encoder_block_indices)
        moments = ldm.first_stage_model.quant_conv(moments)
        if self._vae_train_nhwc:  # the UNet and the trainable projections (library kernels under autograd) keep NCHW
            moments, enc_feats = moments.contiguous(), [f.contiguous() for f in enc_feats]
        return SCALE_FACTOR * moments[:, :4], enc_feats  # posterior mean

    @torch.no_grad()  # depends on the image only (through the latent): no gradient path to any trainable parameter
    def decode_taps(self, latent):
        if self._vae_graph_ok(latent):
            return self._vae_graphed("dec", self._decode_taps, latent)
        return self._decode_taps(latent)

    def _decode_taps(self, latent):
        z = latent / SCALE_FACTOR
        if self._vae_train_nhwc:
            z = z.contiguous(memory_format=torch.channels_last)
        z = self.ldm.first_stage_model.post_quant_conv(z)
        taps = self.ldm.decoder(z, taps=self.decoder_block_indices, stop_after_taps=self.prune_dead_compute)[1]
        return [f.contiguous() for f in taps] if self._vae_train_nhwc else taps

    _vae_train_nhwc = False
    _vae_graphs = None

    def enable_vae_train_path(self, on=True):
        """Training: the frozen, gradient-free VAE stages (encoder, decoder taps) take the inference kernels - channels-last weights, the
        f32-accurate split-operand convolutions / GEMMs on the matrix cores instead of the library's f32 Winograd kernels - and replay
        as one HIP graph each (train_graph.GraphedNoGrad).  Their taps go back to NCHW for the UNet and the trainable projections."""
        self._vae_train_nhwc = on
        self._vae_graphs = {} if on else None
        self.ldm.first_stage_model.to(memory_format=torch.channels_last if on else torch.contiguous_format)
        return self

    def _vae_graph_ok(self, x):
        return self._vae_graphs is not None and x.is_cuda and not torch.cuda.is_current_stream_capturing()

    def _vae_graphed(self, which, fn, x):
        from .train_graph import GraphedNoGrad

        key = (which, tuple(x.shape), x.dtype)
        g = self._vae_graphs.get(key)
        if g is None:
            g = self._vae_graphs[key] = GraphedNoGrad(fn, [x])
        return g(x)

    def unet_taps(self, latent, cond_inputs, cond_emb):
        noise = self.shared_noise.to(latent.dtype)
        if noise.shape[2:] != latent.shape[2:]:
            noise = F.interpolate(noise, size=latent.shape[2:], mode="bicubic", align_corners=False)
        noisy = SQRT_AC0 * latent + SQRT_1M_AC0 * noise.expand_as(latent)
        t = torch.zeros(latent.shape[0], dtype=torch.long, device=latent.device)
        return self.ldm.unet(noisy, t, cond_inputs, cond_emb=cond_emb[:, 0], taps=self.unet_block_indices,
                             stop_after_taps=self.prune_dead_compute)[1]

    def enable_train_graph(self, on=True):
        """Training: replay the frozen UNet's forward AND backward as a pair of HIP graphs (_GraphedTaps), one pair per input
        shape.  The UNet is the only frozen net a gradient passes through (to the 3D conditioning); eagerly its ~1500 forward
        and ~3000 backward launches are host-bound at one view per GPU."""
        self._train_graphs = {} if on else None
        return self

    def _unet_taps_graphed(self, latent, cond_inputs, cond_emb):
        key = (tuple(latent.shape), latent.dtype, tuple(cond_inputs.shape), latent.is_contiguous(memory_format=torch.channels_last))
        fn = self._train_graphs.get(key)
        if fn is None:
            ext = self

            class _UNetTaps(nn.Module):  # no registered parameters: the frozen weights are constants of the graph
                def forward(self, lat, ctx, emb):
                    return tuple(ext.unet_taps(lat, ctx, emb))

            fn = self._train_graphs[key] = _GraphedTaps(_UNetTaps().eval(), latent, cond_inputs, cond_emb)
        return list(fn(latent, cond_inputs, cond_emb))

    def from_latent(self, latent, enc_feats, cond_inputs, cond_emb, fork_stream=None):
        """UNet taps and VAE-decoder taps from the latent.  The two are independent: with `fork_stream` (graph capture)
        the decoder is enqueued on that stream and joined afterwards, so the captured graph runs them side by side."""
        if fork_stream is not None:
            cur = torch.cuda.current_stream()
            fork_stream.wait_stream(cur)
            with torch.cuda.stream(fork_stream):
                dec_feats = self.decode_taps(latent)
            unet_feats = self.unet_taps(latent, cond_inputs, cond_emb)
            cur.wait_stream(fork_stream)
        else:
            if getattr(self, "_train_graphs", None) is not None and torch.is_grad_enabled() and latent.is_cuda \
                    and cond_inputs.requires_grad and cond_emb.requires_grad:
                unet_feats = self._unet_taps_graphed(latent, cond_inputs, cond_emb)
            else:
                unet_feats = self.unet_taps(latent, cond_inputs, cond_emb)
            dec_feats = self.decode_taps(latent)
        return [*enc_feats, *unet_feats, *dec_feats]

    def forward(self, img, cond_inputs, cond_emb):
        """img (B,3,512,512) in [0,1]; cond_inputs (B,77,768); cond_emb (B,1,1280) -> list of 8 feature maps."""
        latent, enc_feats = self.encode(img)
        return self.from_latent(latent, enc_feats, cond_inputs, cond_emb)


class PositionalLinear(nn.Module):
    def __init__(self, in_features, out_features, seq_len=77, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        self.positional_embedding = nn.Parameter(torch.zeros(1, seq_len, out_features))
        nn.init.trunc_normal_(self.positional_embedding, std=0.02)

    def forward(self, x):
        x = self.linear(x)
        if x.dim() == 2:
            x = x.unsqueeze(1) + self.positional_embedding
        return x


class LdmImplicitCaptionerExtractor(nn.Module):
    """cond = uncond + tanh(alpha) * (Linear(prefix) + pos);  cond_emb = tanh(alpha_t) * (Linear_t(prefix) + pos_t)."""

    def __init__(self, learnable_time_embed=True, num_timesteps=1, dim_latent=768, clip=None, **kwargs):
        super().__init__()
        assert clip is None
        self.ldm_extractor = LdmExtractor(**kwargs)
        self.clip_project = PositionalLinear(dim_latent, 768, 77)
        self.alpha_cond = nn.Parameter(torch.zeros(1, 77, 768))
        self.learnable_time_embed = learnable_time_embed
        if learnable_time_embed:
            self.time_embed_project = PositionalLinear(dim_latent, 1280, num_timesteps)
            self.alpha_cond_time_embed = nn.Parameter(torch.zeros(1280))

    feature_size = (512, 512)

    @property
    def feature_dims(self):
        return self.ldm_extractor.feature_dims

    @property
    def feature_strides(self):
        return self.ldm_extractor.feature_strides

    @property
    def grouped_indices(self):
        return self.ldm_extractor.grouped_indices

    def conditioning(self, prefix):
        cond = self.ldm_extractor.ldm.uncond_inputs + torch.tanh(self.alpha_cond) * self.clip_project(prefix)
        cond_emb = None
        if self.learnable_time_embed:
            cond_emb = torch.tanh(self.alpha_cond_time_embed) * self.time_embed_project(prefix)
        return cond, cond_emb

    def forward(self, batched_inputs, prefix):
        cond, cond_emb = self.conditioning(prefix)
        img = batched_inputs["img"]
        if "latent" in batched_inputs:  # VAE encoder already run (XMASK3d overlaps it with the 3D branch)
            dt = batched_inputs["latent"].dtype
            return self.ldm_extractor.from_latent(batched_inputs["latent"], batched_inputs["enc_feats"], cond.to(dt), cond_emb.to(dt),
                                                  batched_inputs.get("fork_stream"))
        return self.ldm_extractor(img, cond.to(img.dtype), cond_emb.to(img.dtype))


# ----------------------------------------------------------------------------- projection backbone
class _ConvGN(nn.Conv2d):
    """detectron2 Conv2d(bias=False, norm=GroupNorm(32)) with optional ReLU; parameter names weight / norm.*"""

    def __init__(self, cin, cout, k, padding=0, relu=False):
        super().__init__(cin, cout, k, padding=padding, bias=False)
        self.norm = GroupNorm(32, cout)
        self._relu = relu

    def forward(self, x, residual=None, relu=None):
        """residual / relu: the tail of the bottleneck, relu(norm(conv(x)) + residual), in the GroupNorm's apply pass"""
        relu = self._relu if relu is None else relu
        return gn_act(self.norm, self.conv_only(x), ACT_RELU if relu else ACT_NONE, residual=residual)

    def conv_only(self, x):
        """the convolution without its norm: the implicit-GEMM kernel on channels-last bf16 inference (bit-reproducible; the
        library's 1x1 convolutions at K >= 512 were not), torch otherwise"""
        out = own_conv(self, x)
        return out if out is not None else nn.Conv2d.forward(self, x)


class GNBottleneck(nn.Module):
    def __init__(self, cin, bottleneck, cout):
        super().__init__()
        self.shortcut = _ConvGN(cin, cout, 1) if cin != cout else None
        self.conv1 = _ConvGN(cin, bottleneck, 1, relu=True)
        self.conv2 = _ConvGN(bottleneck, bottleneck, 3, padding=1, relu=True)
        self.conv3 = _ConvGN(bottleneck, cout, 1)

    def forward(self, x):
        # detectron2 BottleneckBlock: relu(conv3(...) + shortcut); add + ReLU ride in conv3's GroupNorm apply pass
        res = self.shortcut(x) if self.shortcut is not None else x
        c1 = self.conv1.conv_only(x)
        if fused_conv_ok(c1, self.conv2):
            # the 3x3 convolution on the HIP kernel: conv1's GroupNorm + ReLU are applied while its input tile is staged (one
            # statistics pass over conv1's output instead of statistics + apply), and the moments for conv2's own GroupNorm come
            # out of the epilogue (its apply pass, which conv3 - a library 1x1 convolution - needs materialised, skips the statistics)
            gamma, beta = _gn_f32(self.conv1.norm)
            n1 = self.conv1.norm
            gn = (ops.gn_stats_of(c1, n1.num_groups), gamma, beta, n1.eps, n1.num_groups, "relu")
            n2 = self.conv2.norm  # the epilogue's moments need whole 4-channel groups per lane (xm3d_conv3x3_nhwc); else conv2's GroupNorm takes its own pass
            sg = n2.num_groups if (self.conv2.out_channels // n2.num_groups) % 4 == 0 else None
            if c1.dtype == torch.float32:  # fp32 configuration: the f32-accurate form on split bf16 operands
                packs, tile, _ = _packed_split(self.conv2)
                c2 = ops.conv3x3_f32(c1, packs, self.conv2.out_channels, tile, gn=gn, stats_groups=sg)
            else:
                packed, tile, _ = _packed(self.conv2)
                c2 = ops.conv3x3(c1, packed, self.conv2.out_channels, tile, gn=gn, stats_groups=sg)
            return self.conv3(gn_act(self.conv2.norm, c2, ACT_RELU), residual=res, relu=True)
        return self.conv3(self.conv2(gn_act(self.conv1.norm, c1, ACT_RELU)), residual=res, relu=True)


class FeatureExtractorBackbone(nn.Module):
    def __init__(self, feature_extractor, out_features, backbone_in_size=(512, 512), min_stride=4, max_stride=32,
                 projection_dim=512, num_res_blocks=1, use_checkpoint=False, slide_training=False):
        super().__init__()
        assert num_res_blocks == 1
        self.feature_extractor = feature_extractor
        self.use_checkpoint = use_checkpoint
        self.feature_projections = nn.ModuleList(
            nn.Sequential(GNBottleneck(d, projection_dim // 4, projection_dim)) for d in feature_extractor.feature_dims)
        self.backbone_in_size = tuple(backbone_in_size)
        stride_to_indices = defaultdict(list)
        for indices in feature_extractor.grouped_indices:
            for idx in indices:
                s = min(max(feature_extractor.feature_strides[idx], min_stride), max_stride)
                stride_to_indices[s].append(idx)
        self._groups = []
        self._out_feature_strides, self._out_feature_channels = {}, {}
        for s in sorted(stride_to_indices):
            name = f"s{int(math.log2(s))}"
            if name in out_features:
                self._groups.append((name, s, stride_to_indices[s]))
                self._out_feature_strides[name], self._out_feature_channels[name] = s, projection_dim
        self._out_features = [g[0] for g in self._groups]

    size_divisibility = 64

    def output_shape(self):
        return {n: (self._out_feature_channels[n], self._out_feature_strides[n]) for n in self._out_features}

    def forward_features(self, features, input_image_size):
        out = {}
        for name, stride, indices in self._groups:
            acc = None
            size = (input_image_size[-2] // stride, input_image_size[-1] // stride)
            for idx in indices:
                f = features[idx]
                if tuple(f.shape[-2:]) != size:  # (same size: nearest resampling is the identity - no launch, no copy)
                    # outside autocast: there the resampling would run in f32 (a cast before and one after, 0.35 ms each at
                    # 20 x 512 x 128^2); nearest neighbour only moves values, bf16 in / bf16 out is exact
                    with torch.autocast(device_type=f.device.type, enabled=False):
                        f = F.interpolate(f, size=size)
                p = self.feature_projections[idx](f)
                acc = p if acc is None else acc + p
            out[name] = acc
        return out

    def prepare(self, img):
        h, w = img.shape[-2:]
        if (h, w) != self.backbone_in_size:
            img = F.interpolate(img, size=self.backbone_in_size, mode="bicubic", align_corners=False)
        return img

    def extract(self, img, imp_condition, encoded=None, fork_stream=None):
        """the frozen extractor's 8 feature maps (no trainable parameter below this point except the conditioning projections)"""
        img = self.prepare(img)
        # the frozen extractor runs natively in img.dtype (bf16 weights: no autocast casts, GroupNorm stays bf16 I/O)
        inputs = dict(img=img)
        if encoded is not None:
            inputs.update(latent=encoded[0], enc_feats=encoded[1], fork_stream=fork_stream)
        return self.feature_extractor(inputs, imp_condition)

    def project(self, feats, size, low):
        """the trainable fp32 projections; under autocast when the features are bf16"""
        dev = feats[0].device.type
        with torch.autocast(device_type=dev, dtype=feats[0].dtype if low else torch.bfloat16, enabled=low):
            if self.use_checkpoint and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing():
                from torch.utils.checkpoint import checkpoint

                return checkpoint(self.forward_features, feats, size, use_reentrant=False)
            return self.forward_features(feats, size)

    def forward(self, img, imp_condition, encoded=None, fork_stream=None):
        """img (B,3,H,W) in [0,1] (H=W=512 in every XMask3D config -> one 1x1 sliding window, feature_extractor.py:169-226).
        encoded = (latent, enc_feats) from ``feature_extractor.ldm_extractor.encode(prepare(img))`` skips the VAE encoder."""
        h, w = img.shape[-2:]
        feats = self.extract(img, imp_condition, encoded, fork_stream)
        return self.project(feats, (h, w), img.dtype != torch.float32)
