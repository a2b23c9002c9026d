// LayerNorm over the last dimension of (rows, C) activations, with an optional residual add in front:
//     s = x (+ delta);   y = (s - mean(s)) * rsqrt(var(s) + eps) * gamma + beta;   optionally also writes s.
// The LayerNorms of the frozen transformer blocks the dense branch runs in bf16 - ldm's BasicTransformerBlock (pre-norm:
// x = attn(norm(x)) + x, three per block; reached from models/modeling/meta_arch/ldm.py:425-446) and open_clip's
// ResidualAttentionBlock (meta_arch/clip.py) - whose library kernel takes 40 us on (20 x 4096, 320) bf16 where the bytes
// (one read + one write) allow 17.  With `delta` the residual add that precedes a pre-norm LayerNorm rides along: one read of
// x and delta, one write of the new residual stream s and one of y, instead of add (2 reads + 1 write) + norm (1 + 1).
// One wave per row, the row in registers (C <= 64 * 8 * LN_VPL values), statistics in f32 by two passes over the registers
// (mean first, then centred squares: no cancellation); HBM-bound streaming, 16 bytes per lane per access.
#include "common.h"
#include "vecio.h"

namespace xm3d {

constexpr int LN_VPL = 8;  // 16-byte vectors per lane: C up to 64 * 8 * 8 = 4096 (bf16) / 2048 (f32)

template <typename T>
__global__ __launch_bounds__(256) void k_layer_norm(const T* __restrict__ x, const T* __restrict__ delta, const T* __restrict__ gamma,
                                                    const T* __restrict__ beta, int64_t rows, int C, float eps, T* __restrict__ sum_out,
                                                    T* __restrict__ y) {
    constexpr int N = VecIO<T>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C / N;
    float v[LN_VPL][N];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_VPL; ++i) {
        const int e = lane + 64 * i;
        if (e < nv) {
            VecIO<T>::load(x + row * C + e * N, v[i]);
            if (delta) {
                float d[N];
                VecIO<T>::load(delta + row * C + e * N, d);
#pragma unroll
                for (int j = 0; j < N; ++j) v[i][j] += d[j];
                if (sum_out) {  // the residual stream continues in the storage dtype: normalise what was stored
                    VecIO<T>::store(sum_out + row * C + e * N, v[i]);
                    if (sizeof(T) == 2) {
#pragma unroll
                        for (int j = 0; j < N; ++j) v[i][j] = __bfloat162float(__float2bfloat16(v[i][j]));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < N; ++j) s += v[i][j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / float(C);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < LN_VPL; ++i) {
        const int e = lane + 64 * i;
        if (e < nv) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float d = v[i][j] - mean;
                ss = fmaf(d, d, ss);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rstd = rsqrtf(ss / float(C) + eps);
#pragma unroll
    for (int i = 0; i < LN_VPL; ++i) {
        const int e = lane + 64 * i;
        if (e < nv) {
            float g[N], b[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                g[j] = 1.f;
                b[j] = 0.f;
            }
            if (gamma) VecIO<T>::load(gamma + e * N, g);
            if (beta) VecIO<T>::load(beta + e * N, b);
#pragma unroll
            for (int j = 0; j < N; ++j) v[i][j] = fmaf((v[i][j] - mean) * rstd, g[j], b[j]);
            VecIO<T>::store(y + row * C + e * N, v[i]);
        }
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_layer_norm(const void* x, const void* delta, int32_t dtype, int64_t rows, int32_t C, const void* gamma, const void* beta,
                               float eps, void* sum_out, void* y, void* stream) {
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "layer_norm: dtype must be 0 (f32) or 1 (bf16)");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(rows >= 0 && C >= N && C % N == 0 && C <= 64 * LN_VPL * N, "layer_norm: rows=%lld, C=%d (multiple of %d, <= %d) expected",
                 (long long)rows, C, N, 64 * LN_VPL * N);
    if (rows == 0) return XM3D_OK;
    XM3D_REQUIRE(x && y, "layer_norm: null pointer");
    XM3D_REQUIRE(!sum_out || delta, "layer_norm: sum_out without delta");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(delta) |
                   reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(sum_out)) & 15) == 0,
                 "layer_norm: tensors must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    const unsigned blocks = unsigned((rows + 3) / 4);
    if (dtype == 0)
        hipLaunchKernelGGL(k_layer_norm<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float*>(x), static_cast<const float*>(delta),
                           static_cast<const float*>(gamma), static_cast<const float*>(beta), rows, C, eps, static_cast<float*>(sum_out),
                           static_cast<float*>(y));
    else
        hipLaunchKernelGGL(k_layer_norm<__hip_bfloat16>, dim3(blocks), dim3(256), 0, s, static_cast<const __hip_bfloat16*>(x),
                           static_cast<const __hip_bfloat16*>(delta), static_cast<const __hip_bfloat16*>(gamma),
                           static_cast<const __hip_bfloat16*>(beta), rows, C, eps, static_cast<__hip_bfloat16*>(sum_out),
                           static_cast<__hip_bfloat16*>(y));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
