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

// ---- training-mode BatchNorm: the per-channel arithmetic between the statistics pass and the affine pass, and the
// backward, as single kernels (the torch op chains they replace cost ~20 launches per layer and dominated the sparse
// nets' training step on the host: 110 BatchNorm layers per iteration).
// packed = [sum(c), sumsq(c), (count)] f64 as written by k_bn_stats (and all-reduced for SyncBatchNorm).
__global__ void k_bn_finalize(const double* __restrict__ packed, int c, double total_in, const float* __restrict__ weight,
                              const float* __restrict__ bias, float eps, float momentum, float* __restrict__ running_mean,
                              float* __restrict__ running_var, int64_t* __restrict__ num_batches, float* __restrict__ mean_out,
                              float* __restrict__ invstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out,
                              float* __restrict__ total_out) {
    const double total = total_in >= 0.0 ? total_in : packed[2 * c];
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        const double mean = packed[ch] / total;
        double var = packed[c + ch] / total - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double invstd = 1.0 / sqrt(var + double(eps));
        const double w = weight ? double(weight[ch]) : 1.0;
        mean_out[ch] = float(mean);
        invstd_out[ch] = float(invstd);
        scale_out[ch] = float(invstd * w);
        float sh = float(-mean * invstd * w);
        if (bias) sh += bias[ch];
        shift_out[ch] = sh;
        if (momentum >= 0.f && running_mean && running_var) {  // torch semantics: unbiased variance into the running buffer
            const double denom = total - 1.0 > 1.0 ? total - 1.0 : 1.0;
            const float unbiased = float(var * (total / denom));
            running_mean[ch] = running_mean[ch] * (1.f - momentum) + momentum * float(mean);
            running_var[ch] = running_var[ch] * (1.f - momentum) + momentum * unbiased;
        }
    }
    if (threadIdx.x == 0) {
        *total_out = float(total);
        if (num_batches && momentum >= 0.f) *num_batches += 1;
    }
}

// sums[ch] = sum_r gy[r,ch], sums[c+ch] = sum_r gy[r,ch] * xhat[r,ch]   (same tiling as k_bn_stats)
__global__ void k_bn_bwd_reduce(const float* __restrict__ gy, const float* __restrict__ x, int64_t n, int c,
                                const float* __restrict__ mean, const float* __restrict__ invstd, double* __restrict__ sums) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);
    const int Q = c / 4;
    const int RL = blockDim.x / Q;
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int64_t r0 = int64_t(blockIdx.x) * 256;
    const int64_t r1 = (r0 + 256 < n) ? r0 + 256 : n;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (rl < RL) {
        const float4 m = *reinterpret_cast<const float4*>(mean + q * 4);
        const float4 is = *reinterpret_cast<const float4*>(invstd + q * 4);
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            const float4 g = *reinterpret_cast<const float4*>(gy + r * c + q * 4);
            const float4 v = *reinterpret_cast<const float4*>(x + r * c + q * 4);
            a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
            b[0] += double(g.x) * ((v.x - m.x) * is.x); b[1] += double(g.y) * ((v.y - m.y) * is.y);
            b[2] += double(g.z) * ((v.z - m.z) * is.z); b[3] += double(g.w) * ((v.w - m.w) * is.w);
        }
        for (int i = 0; i < 4; ++i) {
            sm[(rl * c + q * 4 + i) * 2] = a[i];
            sm[(rl * c + q * 4 + i) * 2 + 1] = b[i];
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        double sa = 0, sb = 0;
        for (int r = 0; r < RL; ++r) {
            sa += sm[(r * c + ch) * 2];
            sb += sm[(r * c + ch) * 2 + 1];
        }
        atomicAdd(&sums[ch], sa);
        atomicAdd(&sums[c + ch], sb);
    }
}

// gx = w * invstd * (gy - sum_dy / total - xhat * sum_dy_xhat / total); workgroup 0 also writes gw = sum_dy_xhat, gb = sum_dy
__global__ void k_bn_bwd_apply(const float* __restrict__ gy, const float* __restrict__ x, int64_t n4, int c,
                               const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ weight,
                               const double* __restrict__ sums, const float* __restrict__ total, float* __restrict__ gx,
                               float* __restrict__ gw, float* __restrict__ gb) {
    const float inv_total = 1.f / *total;
    if (blockIdx.x == 0) {
        for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
            if (gb) gb[ch] = float(sums[ch]);
            if (gw) gw[ch] = float(sums[c + ch]);
        }
    }
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n4; e += stride) {
        const int ch = int((e * 4) % c);
        const float4 g = reinterpret_cast<const float4*>(gy)[e];
        const float4 v = reinterpret_cast<const float4*>(x)[e];
        const float gi[4] = {g.x, g.y, g.z, g.w}, vi[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float is = invstd[ch + i];
            const float xhat = (vi[i] - mean[ch + i]) * is;
            const float w = weight ? weight[ch + i] : 1.f;
            o[i] = (w * is) * (gi[i] - float(sums[ch + i]) * inv_total - xhat * (float(sums[c + ch + i]) * inv_total));
        }
        reinterpret_cast<float4*>(gx)[e] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_bn_finalize(const double* packed, int32_t c, double total, const float* weight, const float* bias, float eps,
                                float momentum, float* running_mean, float* running_var, int64_t* num_batches, float* mean,
                                float* invstd, float* scale, float* shift, float* total_out, void* stream) {
    XM3D_REQUIRE(c >= 1 && packed && mean && invstd && scale && shift && total_out, "bn_finalize: bad arguments (c=%d)", c);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(256), 0, as_stream(stream), packed, c, total, weight, bias, eps, momentum,
                       running_mean, running_var, num_batches, mean, invstd, scale, shift, total_out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_bn_bwd_reduce(const float* gy, const float* x, int64_t n, int32_t c, const float* mean, const float* invstd,
                                  double* sums, void* stream) {
    XM3D_REQUIRE(n >= 0 && c >= 4 && c % 4 == 0 && c <= 1024, "bn_bwd_reduce: c=%d must be a multiple of 4 in [4,1024]", c);
    XM3D_REQUIRE(sums && mean && invstd, "bn_bwd_reduce: null pointer");
    hipStream_t s = as_stream(stream);
    XM3D_HIP(hipMemsetAsync(sums, 0, 2 * size_t(c) * sizeof(double), s));
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(gy && x && ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mean) |
                              reinterpret_cast<uintptr_t>(invstd)) & 15) == 0, "bn_bwd_reduce: null or misaligned tensor");
    const int Q = c / 4;
    const int RL = 256 / Q > 0 ? 256 / Q : 1;
    const int threads = Q > 256 ? Q : 256;
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3((n + 255) / 256), dim3(threads), size_t(RL) * c * 2 * sizeof(double), s, gy, x, n, c, mean,
                       invstd, sums);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_bn_bwd_apply(const float* gy, const float* x, int64_t n, int32_t c, const float* mean, const float* invstd,
                                 const float* weight, const double* sums, const float* total, float* gx, float* gw, float* gb,
                                 void* stream) {
    XM3D_REQUIRE(n >= 0 && c >= 4 && c % 4 == 0, "bn_bwd_apply: c=%d must be a multiple of 4", c);
    XM3D_REQUIRE(mean && invstd && sums && total, "bn_bwd_apply: null pointer");
    XM3D_REQUIRE(n == 0 || (gy && x && gx && ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(x) |
                                               reinterpret_cast<uintptr_t>(gx)) & 15) == 0), "bn_bwd_apply: null or misaligned tensor");
    const int64_t n4 = n * c / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(blocks), dim3(256), 0, as_stream(stream), gy, x, n4, c, mean, invstd, weight, sums, total, gx,
                       gw, gb);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

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
