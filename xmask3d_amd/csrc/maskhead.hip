// Prediction heads of the masked transformer decoder in two launches over `mask_features`:
//   forward_prediction_heads of XMask3D's Mask2Former decoder (/root/reference/models/modeling/meta_arch/odise.py:445-491, :462
//   `outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)`) + the mask handling of its forward (:395) + MaskPooling
//   (:509-547, `einsum("bchw,bqhw->bqc", x, hard_mask / denorm)`).
//
// k_mask_logits_bias - per decoder layer: the mask logits on the matrix cores AND, from the logits while they are still on chip, the
//   additive attention bias of the next masked cross-attention (bilinear shrink by an even factor s = mean of the centre 2 x 2 of
//   every s x s block -> sigmoid -> < 0.5 -> "a query whose mask is empty everywhere attends to everything" -> 0 / -inf).  The 9
//   intermediate layers of an evaluation forward need nothing else from their logits: for them the (B, Q, H, W) tensor is never
//   written and only the two centre rows of every s-row band are computed (1/4 of the pixels at 32^2, 1/16 at 16^2 targets).  The
//   last layer also stores the logits (bf16) - they are the `pred_masks` output.
//   Workgroup = 4 waves = one s-row band of one image, a wave one 32-pixel segment of its rows.  mask_features is channels-last
//   (B, H*W, C = 256) bf16: the 32 x 16 A fragments of v_mfma_f32_32x32x16_bf16 (rows = pixels) are 16-byte global loads; the
//   mask embedding (B, Q <= 64, C) sits in LDS in B-fragment order (columns = queries).  Accumulator: query on the lane, 16 pixels
//   in its registers, four consecutive per register quad = 8-byte logit stores.  The bias is computed from a wave-private LDS
//   copy of the two centre rows with k_attn_mask's arithmetic (attnmask.hip: same summation order, same 16-bit roundings), so
//   given equal logits the two produce the same bits.  The empty-mask rule needs all bands of a map: k_mask_bias_fix (one workgroup
//   per map, 8 MB at most) rewrites the maps without an open position.  (A first version kept it inside the launch - open flags +
//   an atomic ticket, the last workgroup of an image fixing up - and spent 50 us per launch in the device-scope fences that needs:
//   on this chip a release fence writes an XCD's L2 back.)
// k_mask_pool - last layer only: pooled[b, q, :] = sum over pixels with logit > 0 of mask_features[b, pixel, :], count[b, q] = number
//   of such pixels (sigmoid(x) > 0.5 <=> x > 0), again on the matrix cores: A = features transposed (rows = channels; eight
//   2-byte loads per fragment, L1-resident rows), B = the hard mask taken from the stored logits (16-byte loads, thresholded in
//   registers).  Every wave writes its partial sums for its pixel chunk; the caller adds the chunks and divides by count + 1e-8
//   (device-scope f32 atomics into (B, Q, C) were the whole run time of a first version).  The sigmoid / threshold / cast / sum / divide
//   passes over (B, Q, H, W) of the op chain do not exist.
// Bound: both are small (0.84 GFLOP per image and product); HBM: mask_features read once per launch (8 MB per image).
#include <hip/hip_bf16.h>

#include <algorithm>

#include "common.h"

namespace xm3d {

typedef float mh_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mh_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mh_bf16x4 __attribute__((ext_vector_type(4)));

constexpr int MH_C = 256;            // mask_dim
constexpr int MH_KS = MH_C / 16;     // MFMA k-steps
constexpr int MH_QB = 2;             // 32-query blocks (Q <= 64)
constexpr int MH_ELDS = MH_QB * MH_KS * 1024;         // mask embedding in fragment order
constexpr int MH_SROW = 40;                           // bf16 per (row, query) line of the scratch: 32 pixels + 8 pad (banks)
constexpr int MH_SCR = 2 * 64 * MH_SROW * 2;          // per wave: two centre rows x 64 queries x 32 pixels, bf16
constexpr int MH_LDS = MH_ELDS + 4 * MH_SCR;

__device__ __forceinline__ float mh_round_bf16(float v) { return float((__bf16)v); }

struct MaskHeadArgs {
    const __bf16* embed;     // (B, Q, C)
    const __bf16* feat;      // (B, H*W, C) channels-last
    __bf16* logits;          // (B, Q, H, W) or null
    void* bias;              // (B, Q, h*w) f32 / bf16 or null
    int B, Q, H, W, h, w;
    int bias_is_bf16;
};

template <bool STORE>
__global__ __launch_bounds__(256) void k_mask_logits_bias(const MaskHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const elds = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y;
    const int H = a.H, W = a.W, Q = a.Q;
    const int s = H / a.h, sw = W / a.w;  // shrink factors (even); bands of s rows
    const int oy0 = s / 2 - 1, ox0 = sw / 2 - 1;

    // mask embedding -> LDS, B-fragment order [qb][ks][lane][8]: column q = 32 qb + (lane & 31), k = 16 ks + 8 (lane >> 5) + j.
    // Once per workgroup, which then walks over its bands (blockIdx.x, + gridDim.x, ...)
    for (int i = tid; i < MH_QB * MH_KS * 64; i += 256) {
        const int l = i & 63, ks = (i >> 6) % MH_KS, qb = (i >> 6) / MH_KS;
        const int q = qb * 32 + (l & 31);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q < Q) v = *reinterpret_cast<const uint4*>(a.embed + (int64_t(b) * Q + q) * MH_C + ks * 16 + (l >> 5) * 8);
        *reinterpret_cast<uint4*>(elds + i * 16) = v;
    }
    __syncthreads();

    __bf16* const scr = reinterpret_cast<__bf16*>(smem + MH_ELDS + wave * MH_SCR);  // [row 0/1][q 64][pixel 32 + pad]
    const __bf16* const fb = a.feat + int64_t(b) * H * W * MH_C + l31 * MH_C + hh * 8;
    const int nseg = W / 32;
    const int r_first = STORE ? 0 : oy0, n_rows = STORE ? s : 2;
    const int nsegw = (nseg - wave + 3) / 4;                                 // segments wave, wave + 4, ...
    const int nband = (a.h - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);  // bands blockIdx.x, + gridDim.x, ...
    const int ntile = nband * nsegw * n_rows;                                // this wave's tiles: (band, segment, row), row fastest
    auto tile_of = [&](int t, int& band, int& seg, int& rr) __attribute__((always_inline)) {
        rr = r_first + t % n_rows;
        const int u = t / n_rows;
        seg = wave + 4 * (u % nsegw);
        band = blockIdx.x + gridDim.x * (u / nsegw);
    };
    auto load = [&](mh_bf16x8 (&af)[MH_KS], int t) __attribute__((always_inline)) {
        int band, seg, rr;
        tile_of(t < ntile ? t : ntile - 1, band, seg, rr);
        const __bf16* const fp = fb + (int64_t(band * s + rr) * W + seg * 32) * MH_C;
#pragma unroll
        for (int ks = 0; ks < MH_KS; ++ks) af[ks] = *reinterpret_cast<const mh_bf16x8*>(fp + ks * 16);
    };
    // one tile = 32 pixels of one row x 64 queries.  Accumulator register i of lane (query l31 of block qb, half hh) = pixel
    // (i & 3) + 8 (i >> 2) + 4 hh of the segment
    auto process = [&](const mh_bf16x8 (&af)[MH_KS], int t) __attribute__((always_inline)) {
        int band, seg, rr;
        tile_of(t, band, seg, rr);
        const int row = band * s + rr;
        mh_f32x16 acc[MH_QB];
#pragma unroll
        for (int qb = 0; qb < MH_QB; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[qb][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < MH_KS; ++ks)
#pragma unroll
            for (int qb = 0; qb < MH_QB; ++qb) {
                const mh_bf16x8 bf = *reinterpret_cast<const mh_bf16x8*>(elds + ((qb * MH_KS + ks) * 64 + lane) * 16);
                acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bf, acc[qb], 0, 0, 0);
            }
        const int centre = rr - oy0;  // 0 / 1: one of the band's two centre rows
#pragma unroll
        for (int qb = 0; qb < MH_QB; ++qb) {
            const int q = qb * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mh_bf16x4 pk;
                pk[0] = (__bf16)acc[qb][4 * g], pk[1] = (__bf16)acc[qb][4 * g + 1], pk[2] = (__bf16)acc[qb][4 * g + 2], pk[3] = (__bf16)acc[qb][4 * g + 3];
                const int px = 8 * g + 4 * hh;
                if (STORE && q < Q) *reinterpret_cast<mh_bf16x4*>(a.logits + ((int64_t(b) * Q + q) * H + row) * W + seg * 32 + px) = pk;
                if (centre == 0 || centre == 1) *reinterpret_cast<mh_bf16x4*>(scr + (centre * 64 + q) * MH_SROW + px) = pk;
            }
        }
        if (rr != r_first + n_rows - 1 || !a.bias) return;
        // last row of this (band, segment): the bias of its 32 / sw target pixels from the centre 2 x 2 of every s x sw block
        __builtin_amdgcn_wave_barrier();  // the wave's own LDS stores are read back by other lanes (LDS is in order per wave)
        const int per = 32 / sw;
        for (int i = lane; i < 64 * per; i += 64) {
            const int q = i / per, oxl = i - q * per;
            if (q >= Q) continue;
            const __bf16* r0 = scr + q * MH_SROW + oxl * sw + ox0;
            const __bf16* r1 = r0 + 64 * MH_SROW;
            const float v = mh_round_bf16(((float(r0[0]) + float(r0[1])) + (float(r1[0]) + float(r1[1]))) * 0.25f);
            const float sg = mh_round_bf16(1.0f / (1.0f + expf(-v)));
            const bool masked = sg < 0.5f;
            const int64_t o = (int64_t(b) * Q + q) * (a.h * a.w) + band * a.w + seg * per + oxl;
            const float val = masked ? -INFINITY : 0.0f;
            if (a.bias_is_bf16) static_cast<__hip_bfloat16*>(a.bias)[o] = __float2bfloat16(val);
            else static_cast<float*>(a.bias)[o] = val;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // two fragment sets: the loads of tile t + 1 are in flight during the MFMAs of tile t (one wave per SIMD here)
    mh_bf16x8 af0[MH_KS], af1[MH_KS];
    if (ntile > 0) load(af0, 0);
    for (int t = 0; t < ntile; t += 2) {
        load(af1, t + 1);
        process(af0, t);
        if (t + 1 < ntile) {
            load(af0, t + 2);
            process(af1, t + 1);
        }
    }
}

// "a query whose mask is empty everywhere attends to everything" (odise.py:395): a map without a single open (0) position becomes
// all zeros.  One workgroup per (image, query) map.
template <typename U>
__global__ __launch_bounds__(256) void k_mask_bias_fix(U* __restrict__ bias, int n) {
    U* const m = bias + int64_t(blockIdx.x) * n;
    int open = 0;
    for (int i = threadIdx.x; i < n; i += 256) open |= int(float(m[i]) == 0.0f);
    if (__syncthreads_or(open)) return;
    for (int i = threadIdx.x; i < n; i += 256) m[i] = U(0.0f);
}

// one wave = (image, 64-channel slice, pixel chunk): acc[cb][qb] += F^T (32 channels x 16 pixels) . M (16 pixels x 32 queries)
__global__ __launch_bounds__(64) void k_mask_pool(const __bf16* __restrict__ logits, const __bf16* __restrict__ feat, int B, int Q, int HW,
                                                  int chunk, float* __restrict__ pooled, float* __restrict__ count) {
    // pooled (chunks, B, Q, C), count (chunks, B, Q): this wave's partial sums over its pixel chunk
    const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
    const int pc = blockIdx.x, cs = blockIdx.y, b = blockIdx.z;
    const int p_begin = pc * chunk, p_end = min(HW, p_begin + chunk);
    const unsigned short* const fb = reinterpret_cast<const unsigned short*>(feat) + int64_t(b) * HW * MH_C + cs * 64 + l31;
    const __bf16* const lb = logits + int64_t(b) * Q * HW;
    mh_f32x16 acc[2][MH_QB];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int qb = 0; qb < MH_QB; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][qb][i] = 0.f;
    float cnt[MH_QB] = {0.f, 0.f};
    for (int p0 = p_begin; p0 < p_end; p0 += 16) {  // HW and chunk are multiples of 16
        const int pk0 = p0 + 8 * hh;                // k slot (hh, j) <-> pixel p0 + 8 hh + j
        mh_bf16x8 m[MH_QB];
#pragma unroll
        for (int qb = 0; qb < MH_QB; ++qb) {
            const int q = qb * 32 + l31;
            uint4 raw = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);  // -0.0: not > 0
            if (q < Q) raw = *reinterpret_cast<const uint4*>(lb + int64_t(q) * HW + pk0);
            const mh_bf16x8 x = __builtin_bit_cast(mh_bf16x8, raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool on = float(x[j]) > 0.f;
                m[qb][j] = on ? (__bf16)1.f : (__bf16)0.f;
                cnt[qb] += on ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = fb[int64_t(pk0 + j) * MH_C + cb * 32];
            uint4 raw;
            raw.x = e[0] | (unsigned(e[1]) << 16), raw.y = e[2] | (unsigned(e[3]) << 16), raw.z = e[4] | (unsigned(e[5]) << 16),
            raw.w = e[6] | (unsigned(e[7]) << 16);
            const mh_bf16x8 af = __builtin_bit_cast(mh_bf16x8, raw);
#pragma unroll
            for (int qb = 0; qb < MH_QB; ++qb) acc[cb][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, m[qb], acc[cb][qb], 0, 0, 0);
        }
    }
    // D: column = query l31 of block qb, row (i & 3) + 8 (i >> 2) + 4 hh = channel of block cb
#pragma unroll
    for (int qb = 0; qb < MH_QB; ++qb) {
        const int q = qb * 32 + l31;
        if (q < Q) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    pooled[((int64_t(pc) * B + b) * Q + q) * MH_C + cs * 64 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh] = acc[cb][qb][i];
            const float other = __shfl_xor(cnt[qb], 32);  // the two halves of the wave counted different pixels
            if (cs == 0 && hh == 0) count[(int64_t(pc) * B + b) * Q + q] = cnt[qb] + other;
        }
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_mask_logits_bias(const void* mask_embed, const void* mask_features, int64_t B, int32_t Q, int32_t C, int32_t H, int32_t W,
                                     void* logits, int32_t h, int32_t w, void* bias, int32_t bias_dtype, void* stream) {
    XM3D_REQUIRE(mask_embed && mask_features, "mask_logits_bias: null pointer");
    XM3D_REQUIRE(B > 0 && B < 65536 && Q > 0 && Q <= 32 * MH_QB, "mask_logits_bias: B %lld / Q %d out of range (Q <= %d)", (long long)B, Q, 32 * MH_QB);
    XM3D_REQUIRE(C == MH_C, "mask_logits_bias: mask_dim must be %d", MH_C);
    XM3D_REQUIRE(logits || bias, "mask_logits_bias: nothing to compute");
    XM3D_REQUIRE(bias_dtype == 0 || bias_dtype == 1, "mask_logits_bias: bias dtype must be 0 (f32) or 1 (bf16)");
    XM3D_REQUIRE(h >= 1 && w >= 1 && H >= 2 * h && W >= 2 * w && H % h == 0 && W % w == 0 && (H / h) % 2 == 0 && (W / w) % 2 == 0 && W % 32 == 0 &&
                     32 % (W / w) == 0,
                 "mask_logits_bias: (%d,%d) -> (%d,%d) is not a shrink by an even factor dividing the 32-pixel segment", H, W, h, w);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(mask_embed) | reinterpret_cast<uintptr_t>(mask_features) | reinterpret_cast<uintptr_t>(logits)) & 15) == 0,
                 "mask_logits_bias: tensors must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    MaskHeadArgs a;
    a.embed = static_cast<const __bf16*>(mask_embed);
    a.feat = static_cast<const __bf16*>(mask_features);
    a.logits = static_cast<__bf16*>(logits);
    a.bias = bias;
    a.B = int(B), a.Q = Q, a.H = H, a.W = W, a.h = h, a.w = w;
    a.bias_is_bf16 = bias_dtype;
    static DeviceOnce configured;  // the attribute is per device
    if (configured.first()) {
        XM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mask_logits_bias<true>), hipFuncAttributeMaxDynamicSharedMemorySize, MH_LDS));
        XM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mask_logits_bias<false>), hipFuncAttributeMaxDynamicSharedMemorySize, MH_LDS));
    }
    // workgroups per image: enough to fill the chip twice over, at most one per band; each stages the embedding once
    const int per_image = int(std::min<int64_t>(h, std::max<int64_t>(1, (512 + B - 1) / B)));
    const dim3 grid{unsigned(per_image), unsigned(B)}, block{256};
    if (logits) hipLaunchKernelGGL(k_mask_logits_bias<true>, grid, block, MH_LDS, s, a);
    else hipLaunchKernelGGL(k_mask_logits_bias<false>, grid, block, MH_LDS, s, a);
    XM3D_LAUNCH_CHECK();
    if (bias) {
        if (bias_dtype) hipLaunchKernelGGL(k_mask_bias_fix<__hip_bfloat16>, dim3(unsigned(B * Q)), dim3(256), 0, s, static_cast<__hip_bfloat16*>(bias), h * w);
        else hipLaunchKernelGGL(k_mask_bias_fix<float>, dim3(unsigned(B * Q)), dim3(256), 0, s, static_cast<float*>(bias), h * w);
        XM3D_LAUNCH_CHECK();
    }
    return XM3D_OK;
}

extern "C" int32_t xm3d_mask_pool_chunks(int64_t HW) { return int32_t((HW + 1023) / 1024); }

extern "C" int xm3d_mask_pool(const void* logits, const void* mask_features, int64_t B, int32_t Q, int32_t C, int64_t HW, float* pooled_partial,
                              float* count_partial, void* stream) {
    XM3D_REQUIRE(logits && mask_features && pooled_partial && count_partial, "mask_pool: null pointer");
    XM3D_REQUIRE(B > 0 && B < 65536 && Q > 0 && Q <= 32 * MH_QB && C == MH_C && HW > 0 && HW % 16 == 0 && HW < (int64_t(1) << 24),
                 "mask_pool: unsupported shape (Q <= %d, mask_dim %d, H*W a multiple of 16)", 32 * MH_QB, MH_C);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(mask_features)) & 15) == 0, "mask_pool: tensors must be 16-byte aligned");
    const int chunk = 1024;  // pixels per wave: 16 x 4 x B waves at 128^2 (latency hiding comes from waves per SIMD)
    const dim3 grid{unsigned(xm3d_mask_pool_chunks(HW)), unsigned(MH_C / 64), unsigned(B)};
    hipLaunchKernelGGL(k_mask_pool, grid, dim3(64), 0, as_stream(stream), static_cast<const __bf16*>(logits), static_cast<const __bf16*>(mask_features),
                       int(B), Q, int(HW), chunk, pooled_partial, count_partial);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
