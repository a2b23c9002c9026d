// 2D mask -> 3D point feature fusion: for every point, the mean CLIP-space embedding of the mask
// queries whose (binary) mask covers the point's pixel.  One wave-sized lane group streams the C
// channels of a point; replaces the <=50-iteration boolean-index scatter loops of
// models/xmask3d.py:421-451 and models/utils/fuser.py:24-35 with one pass.
#include "common.h"

namespace xm3d {

// one workgroup row = one point; 64 lanes stride over C/4 channel quads
__global__ void k_mask_point_fuse(const uint8_t* __restrict__ masks, int Q, int Hm, int Wm, const int64_t* __restrict__ xr,
                                  const int64_t* __restrict__ yc, int64_t n, const float* __restrict__ embed, int C,
                                  float* __restrict__ feat2d, int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t p = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= n) return;
    const int64_t r = xr[p], c = yc[p];
    const bool ok = r >= 0 && r < Hm && c >= 0 && c < Wm;
    // lanes test queries lane, lane+64, ... ; ballots (all 64 lanes active) give the covering set; a lane owns the
    // channel quads lane, lane+64, ... (C <= 1024) and adds the embeddings in ascending q = the reference's order
    constexpr int MAXJ = 4;
    float4 acc[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    const int C4 = C / 4;
    for (int q0 = 0; q0 < Q; q0 += 64) {
        const int q = q0 + lane;
        const bool hit = ok && q < Q && masks[(int64_t(q) * Hm + r) * Wm + c] != 0;
        unsigned long long m = __ballot(hit);
        cnt += __popcll(m);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int c4 = lane + 64 * j;
                if (c4 < C4) {
                    const float4 e = *reinterpret_cast<const float4*>(embed + int64_t(q0 + b) * C + c4 * 4);
                    acc[j].x += e.x; acc[j].y += e.y; acc[j].z += e.z; acc[j].w += e.w;
                }
            }
        }
    }
    const float d = cnt > 0 ? float(cnt) : 1e-5f;  // the reference divides by 1e-5 where no mask covers
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < C4) {
            float4 v = acc[j];
            v.x /= d; v.y /= d; v.z /= d; v.w /= d;
            *reinterpret_cast<float4*>(feat2d + p * C + c4 * 4) = v;
        }
    }
    if (lane == 0) count[p] = cnt;
}

// Pixel ownership among the mask queries (models/xmask3d.py:372-392 / criterion.py:245-328, the panoptic-style merge):
//   prob[q] = (keep[q] ? score[q] : -1) * sigmoid(logit[q]);  id = first arg-max over q;
//   owner   = id if sigmoid(logit[id]) >= 0.5 and keep[id], else -1
// One thread per pixel walks the Q logits (coalesced across the wave for every q): one read of the (B,Q,H,W) logits
// instead of the sigmoid / multiply / arg-max / three compares / two ands / cast chain over that tensor (ten passes).
// Same arithmetic as the chain (f32 sigmoid 1/(1+exp(-x)), f32 product, first maximum wins), so the ownership map is
// identical; the per-query boolean masks the reference materialises are `owner == q`.
__global__ __launch_bounds__(256) void k_mask_owner(const float* __restrict__ logits, const float* __restrict__ score,
                                                    const uint8_t* __restrict__ keep, int Q, int64_t hw, int32_t* __restrict__ owner) {
    const int b = blockIdx.y;
    const int64_t pix = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (pix >= hw) return;
    const float* L = logits + int64_t(b) * Q * hw + pix;
    float best = -INFINITY, best_s = 0.f;
    int id = 0;
    for (int q = 0; q < Q; ++q) {
        const float s = 1.0f / (1.0f + expf(-L[int64_t(q) * hw]));
        const float p = (keep[b * Q + q] ? score[b * Q + q] : -1.0f) * s;
        if (p > best) {  // strict: the first maximum wins, like torch.argmax
            best = p;
            best_s = s;
            id = q;
        }
    }
    owner[int64_t(b) * hw + pix] = (best_s >= 0.5f && keep[b * Q + id]) ? id : -1;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_mask_owner(const float* logits, const float* score, const uint8_t* keep, int32_t B, int32_t Q, int64_t hw,
                               int32_t* owner, void* stream) {
    XM3D_REQUIRE(B >= 0 && Q >= 1 && hw >= 0 && B <= 65535, "mask_owner: bad sizes B=%d Q=%d hw=%lld", B, Q, (long long)hw);
    if (B == 0 || hw == 0) return XM3D_OK;
    XM3D_REQUIRE(logits && score && keep && owner, "mask_owner: null pointer");
    hipLaunchKernelGGL(k_mask_owner, dim3(unsigned((hw + 255) / 256), unsigned(B)), dim3(256), 0, as_stream(stream), logits, score, keep, Q, hw,
                       owner);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_mask_point_fuse(const uint8_t* masks, int32_t Q, int32_t Hm, int32_t Wm, const int64_t* x,
                                    const int64_t* y, int64_t n, const float* embed, int32_t C, float* feat2d,
                                    int32_t* count, void* stream) {
    XM3D_REQUIRE(Q >= 0 && Hm >= 1 && Wm >= 1 && n >= 0 && C >= 4 && C % 4 == 0 && C <= 1024,
                 "mask_point_fuse: bad sizes (C %% 4 == 0, C <= 1024)");
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE((Q == 0 || (masks && embed)) && x && y && feat2d && count, "mask_point_fuse: null pointer");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(embed) | reinterpret_cast<uintptr_t>(feat2d)) & 15) == 0,
                 "mask_point_fuse: embed/feat2d must be 16-byte aligned");
    hipLaunchKernelGGL(k_mask_point_fuse, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), masks, Q, Hm, Wm, x, y, n, embed,
                       C, feat2d, count);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
