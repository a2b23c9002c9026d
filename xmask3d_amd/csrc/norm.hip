// Row-wise batch-norm helpers over an (n, c) feature matrix: per-channel statistics and the fused
// affine (+residual)(+ReLU) stream.  Pure HBM streaming kernels, 16 bytes per lane.
// Replaces ME.MinkowskiBatchNorm / MinkowskiReLU (mink_unet.py:51-116) outside the conv epilogue.
#include "common.h"

namespace xm3d {

// block: 256 threads = RL row lanes x Q channel quads (Q = c/4, power-of-two friendly but not required)
__global__ void k_bn_stats(const float* __restrict__ x, int64_t n, int c, double* __restrict__ sums) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);  // [RL][c][2]
    const int Q = c / 4;
    const int RL = blockDim.x / Q;
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int64_t rows_per_block = 256;
    const int64_t r0 = int64_t(blockIdx.x) * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (rl < RL) {
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * c + q * 4);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            ss[0] += double(v.x) * v.x; ss[1] += double(v.y) * v.y; ss[2] += double(v.z) * v.z; ss[3] += double(v.w) * v.w;
        }
        for (int i = 0; i < 4; ++i) {
            sm[(rl * c + q * 4 + i) * 2] = s[i];
            sm[(rl * c + q * 4 + i) * 2 + 1] = ss[i];
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        double a = 0, b = 0;
        for (int r = 0; r < RL; ++r) {
            a += sm[(r * c + ch) * 2];
            b += sm[(r * c + ch) * 2 + 1];
        }
        atomicAdd(&sums[ch], a);
        atomicAdd(&sums[c + ch], b);
    }
}

__global__ void k_affine_act(const float* __restrict__ x, int64_t n4, int c, const float* __restrict__ scale,
                             const float* __restrict__ shift, const float* __restrict__ residual, int relu,
                             float* __restrict__ out) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n4; e += stride) {
        const int ch = int((e * 4) % c);
        float4 v = reinterpret_cast<const float4*>(x)[e];
        if (scale) {
            const float4 s = *reinterpret_cast<const float4*>(scale + ch);
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        }
        if (shift) {
            const float4 s = *reinterpret_cast<const float4*>(shift + ch);
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
        if (residual) {
            const float4 r = reinterpret_cast<const float4*>(residual)[e];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        reinterpret_cast<float4*>(out)[e] = v;
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_bn_stats(const float* x, int64_t n, int32_t c, double* sum_sumsq, void* stream) {
    XM3D_REQUIRE(n >= 0 && c >= 4 && c % 4 == 0 && c <= 1024, "bn_stats: c=%d must be a multiple of 4 in [4,1024]", c);
    XM3D_REQUIRE(sum_sumsq, "bn_stats: null output");
    hipStream_t s = as_stream(stream);
    XM3D_HIP(hipMemsetAsync(sum_sumsq, 0, 2 * size_t(c) * sizeof(double), s));
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(x && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "bn_stats: x null or misaligned");
    const int Q = c / 4;
    const int RL = 256 / Q > 0 ? 256 / Q : 1;
    const int threads = Q > 256 ? Q : 256;
    const size_t smem = size_t(RL) * c * 2 * sizeof(double);
    hipLaunchKernelGGL(k_bn_stats, dim3((n + 255) / 256), dim3(threads), smem, s, x, n, c, sum_sumsq);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_affine_act(const float* x, int64_t n, int32_t c, const float* scale, const float* shift,
                               const float* residual, int32_t relu, float* out, void* stream) {
    XM3D_REQUIRE(n >= 0 && c >= 4 && c % 4 == 0, "affine_act: c=%d must be a multiple of 4", c);
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(x && out, "affine_act: null pointer");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(residual) |
                   reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0,
                 "affine_act: tensors must be 16-byte aligned");
    const int64_t n4 = n * c / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_affine_act, dim3(blocks), dim3(256), 0, as_stream(stream), x, n4, c, scale, shift, residual, relu, out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
