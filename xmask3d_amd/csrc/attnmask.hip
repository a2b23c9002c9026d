// Attention bias of Mask2Former's masked cross-attention, straight from the mask logits.
// Replaces, per decoder layer, the chain of third_party/Mask2Former .../mask2former_transformer_decoder.py
// forward_prediction_heads + the mask handling in forward (odise.py:395,445-491 in XMask3D's copy):
//   F.interpolate(outputs_mask, size=(h,w), bilinear) -> sigmoid -> < 0.5 -> repeat over heads -> bool
//   -> "a query whose mask is empty everywhere attends to everything" (all / and-not)
//   -> nn.MultiheadAttention's conversion of the bool mask to an additive float mask (masked_fill with -inf)
// (~12 launches and a heads-times replicated mask) by ONE launch that writes the additive bias (B*Q, h*w) once; the
// caller broadcasts it over the heads.  One workgroup per (batch, query) map; the shrink by an even integer factor is
// the exact 2x2 centre mean (see mask_head.bilinear_down) in the library kernel's summation order, the sigmoid and the
// 16-bit roundings are the library's, so the resulting mask is bit-identical to the chain it replaces.
#include <hip/hip_bf16.h>

#include "common.h"

namespace xm3d {

template <typename T>
__device__ inline float ld(const T* p);
template <>
__device__ inline float ld<float>(const float* p) { return *p; }
template <>
__device__ inline float ld<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }

template <typename T>
__device__ inline float round_to(float v);
template <>
__device__ inline float round_to<float>(float v) { return v; }
template <>
__device__ inline float round_to<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

template <typename U>
__device__ inline void st(U* p, float v);
template <>
__device__ inline void st<float>(float* p, float v) { *p = v; }
template <>
__device__ inline void st<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

constexpr int AM_MAX_PER_THREAD = 32;  // h*w <= 256 * 32

template <typename T, typename U>
__global__ __launch_bounds__(256) void k_attn_mask(const T* __restrict__ logits, int H, int W, int h, int w, U* __restrict__ out) {
    const int64_t map = blockIdx.x;
    const T* L = logits + map * int64_t(H) * W;
    U* O = out + map * int64_t(h) * w;
    const int sh = H / h, sw = W / w;
    const int oy0 = sh / 2 - 1, ox0 = sw / 2 - 1;
    const int n = h * w;
    unsigned masked_bits = 0;
    int any_open = 0;
    int k = 0;
    for (int i = threadIdx.x; i < n; i += 256, ++k) {
        const int oy = i / w, ox = i - oy * w;
        const T* p = L + int64_t(oy * sh + oy0) * W + ox * sw + ox0;
        const float a = ld<T>(p), b = ld<T>(p + 1), c = ld<T>(p + W), d = ld<T>(p + W + 1);
        const float v = round_to<T>(((a + b) + (c + d)) * 0.25f);
        const float s = round_to<T>(1.0f / (1.0f + expf(-v)));
        const bool masked = s < 0.5f;
        masked_bits |= unsigned(masked) << k;
        any_open |= int(!masked);
    }
    const int open = __syncthreads_or(any_open);  // no position open for this query -> it attends to everything
    k = 0;
    for (int i = threadIdx.x; i < n; i += 256, ++k) {
        const bool masked = open && ((masked_bits >> k) & 1u);
        st<U>(O + i, masked ? -INFINITY : 0.0f);
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_attn_mask_bias(const void* logits, int32_t in_dtype, int64_t maps, int32_t H, int32_t W, int32_t h, int32_t w,
                                   void* out, int32_t out_dtype, void* stream) {
    XM3D_REQUIRE((in_dtype == 0 || in_dtype == 1) && (out_dtype == 0 || out_dtype == 1), "attn_mask_bias: dtypes must be 0 (f32) or 1 (bf16)");
    XM3D_REQUIRE(maps >= 0 && h >= 1 && w >= 1 && H >= 2 * h && W >= 2 * w && H % h == 0 && W % w == 0 && (H / h) % 2 == 0 && (W / w) % 2 == 0,
                 "attn_mask_bias: (%d,%d) -> (%d,%d) is not a shrink by an even integer factor", H, W, h, w);
    XM3D_REQUIRE(int64_t(h) * w <= 256 * AM_MAX_PER_THREAD, "attn_mask_bias: target map too large (%d x %d)", h, w);
    if (maps == 0) return XM3D_OK;
    XM3D_REQUIRE(logits && out, "attn_mask_bias: null pointer");
    hipStream_t s = as_stream(stream);
    const dim3 grid{unsigned(maps)}, block{256};
    if (in_dtype == 0 && out_dtype == 0)
        hipLaunchKernelGGL((k_attn_mask<float, float>), grid, block, 0, s, static_cast<const float*>(logits), H, W, h, w, static_cast<float*>(out));
    else if (in_dtype == 0)
        hipLaunchKernelGGL((k_attn_mask<float, __hip_bfloat16>), grid, block, 0, s, static_cast<const float*>(logits), H, W, h, w,
                           static_cast<__hip_bfloat16*>(out));
    else if (out_dtype == 0)
        hipLaunchKernelGGL((k_attn_mask<__hip_bfloat16, float>), grid, block, 0, s, static_cast<const __hip_bfloat16*>(logits), H, W, h, w,
                           static_cast<float*>(out));
    else
        hipLaunchKernelGGL((k_attn_mask<__hip_bfloat16, __hip_bfloat16>), grid, block, 0, s, static_cast<const __hip_bfloat16*>(logits), H, W, h,
                           w, static_cast<__hip_bfloat16*>(out));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

// ---- mask-CLIP's patch mask (XMask3D copy of ODISE's MaskCLIP.encode_image_with_mask, /root/reference/models/modeling/meta_arch/clip.py:272-310):
//     blocked[b, q, patch] = max_pool2d(sigmoid(interpolate(mask_logits, (S, S), bilinear)), P, stride P) < 0.5
// without the (B, Q, S, S) f32 intermediate (S = 224: 200 MB per 20-view forward written and read twice).  One thread per (map, patch): the
// bilinear samples of its P x P output pixels (torch's align_corners = False arithmetic, term for term), their maximum (sigmoid is monotone: the
// maximum of the sigmoids is the sigmoid of the maximum), ONE sigmoid and the comparison.
namespace xm3d {
__global__ __launch_bounds__(256) void k_clip_mask_blocked(const float* __restrict__ logits, int64_t maps, int h, int w, int S, int P, int np,
                                                           uint8_t* __restrict__ blocked) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= maps * np * np) return;
    const int px = int(i % np), py = int((i / np) % np);
    const int64_t m = i / (int64_t(np) * np);
    const float* p = logits + m * int64_t(h) * w;
    const float rh = float(h) / float(S), rw = float(w) / float(S);
    float best = -INFINITY;
    for (int oy = py * P; oy < py * P + P; ++oy) {
        float sy = rh * (float(oy) + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = int(sy), y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly = sy - float(y0), hy = 1.f - ly;
        for (int ox = px * P; ox < px * P + P; ++ox) {
            float sx = rw * (float(ox) + 0.5f) - 0.5f;
            sx = sx < 0.f ? 0.f : sx;
            const int x0 = int(sx), x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float lx = sx - float(x0), hx = 1.f - lx;
            const float v = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
            best = fmaxf(best, v);
        }
    }
    const float sg = 1.f / (1.f + expf(-best));
    blocked[i] = sg < 0.5f ? 1 : 0;
}
}  // namespace xm3d

extern "C" int xm3d_clip_mask_blocked(const float* logits, int64_t maps, int32_t h, int32_t w, int32_t S, int32_t P, uint8_t* blocked, void* stream) {
    XM3D_REQUIRE(maps >= 0 && h > 0 && w > 0 && S > 0 && P > 0 && S % P == 0, "clip_mask_blocked: maps=%lld h=%d w=%d S=%d P=%d (S %% P == 0)", (long long)maps, h, w,
                 S, P);
    if (maps == 0) return XM3D_OK;
    XM3D_REQUIRE(logits && blocked, "clip_mask_blocked: null pointer");
    const int np = S / P;
    const int64_t total = maps * np * np;
    XM3D_REQUIRE((total + 255) / 256 < (int64_t(1) << 31), "clip_mask_blocked: grid too large");
    hipLaunchKernelGGL(xm3d::k_clip_mask_blocked, dim3(unsigned((total + 255) / 256)), dim3(256), 0, xm3d::as_stream(stream), logits, maps, h, w, S, P, np, blocked);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
