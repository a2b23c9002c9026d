// Streaming pointwise fusions for the frozen SD nets in channels-last layout (bf16 or f32 I/O, f32 arithmetic).  They
// replace chains of library elementwise kernels around the MIOpen convolutions / hipBLASLt GEMMs (ldm ResnetBlock /
// ResBlock / SpatialTransformer / GEGLU, reached from models/modeling/meta_arch/ldm.py:386-490):
//   k_bias_residual : out = a + b + bias[c]   - conv bias folded into the residual add (PyTorch-ROCm adds a conv bias in a
//                     separate broadcast kernel: one full read+write pass per convolution saved)
//   k_geglu         : out[r, d] = x[r, d] * gelu(x[r, D + d])   - one pass instead of gelu + strided multiply
// Both are HBM-bound: algorithmic bytes = (inputs + output) * sizeof(T), 16-byte accesses per lane.
#include "common.h"
#include "vecio.h"

namespace xm3d {

template <typename T>
__global__ __launch_bounds__(256) void k_bias_residual(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ bias,
                                                       int64_t nvec, int vpp, T* __restrict__ out) {
    constexpr int N = VecIO<T>::N;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < nvec; e += stride) {
        float vb[N], vs[N];
        VecIO<T>::load(b + e * N, vb);
        VecIO<T>::load(bias + (e % vpp) * N, vs);
#pragma unroll
        for (int j = 0; j < N; ++j) vb[j] += vs[j];
        if (a) {
            float va[N];
            VecIO<T>::load(a + e * N, va);
#pragma unroll
            for (int j = 0; j < N; ++j) vb[j] += va[j];
        }
        VecIO<T>::store(out + e * N, vb);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_geglu(const T* __restrict__ x, int64_t rows, int vpr, T* __restrict__ out) {
    constexpr int N = VecIO<T>::N;
    const int64_t nvec = rows * vpr;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < nvec; e += stride) {
        const int64_t r = e / vpr;
        const int v = int(e % vpr);
        float val[N], gate[N];
        VecIO<T>::load(x + (r * 2 * vpr + v) * N, val);
        VecIO<T>::load(x + (r * 2 * vpr + vpr + v) * N, gate);
#pragma unroll
        for (int j = 0; j < N; ++j) val[j] *= 0.5f * gate[j] * (1.f + erff(gate[j] * 0.70710678118654752f));
        VecIO<T>::store(out + e * N, val);
    }
}

// Zero padding at the bottom / right of a channels-last map: out (B, H + pb, W + pr, C) <- x (B, H, W, C).  The VAE encoder's
// Downsample pads (0, 1, 0, 1) before its stride-2 convolution (ldm Downsample, reached from meta_arch/ldm.py:386-414); as
// F.pad that is a fill pass plus a strided copy pass (1.12 ms at 20 x 512^2 x 128 bf16), here one streaming pass.
template <typename T>
__global__ __launch_bounds__(256) void k_pad_nhwc(const T* __restrict__ x, int H, int W, int vpp, int Ho, int Wo, int64_t nvec_out,
                                                  T* __restrict__ out) {
    constexpr int N = VecIO<T>::N;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < nvec_out; e += stride) {
        const int v = int(e % vpp);
        const int64_t pix = e / vpp;
        const int wo = int(pix % Wo);
        const int64_t t = pix / Wo;
        const int ho = int(t % Ho);
        const int64_t b = t / Ho;
        float val[N];
#pragma unroll
        for (int j = 0; j < N; ++j) val[j] = 0.f;
        if (ho < H && wo < W) VecIO<T>::load(x + (((b * H + ho) * W + wo) * vpp + v) * N, val);
        VecIO<T>::store(out + e * N, val);
    }
}

// QuickGELU of CLIP's MLPs (meta_arch/clip.py -> open_clip's QuickGELU: x * sigmoid(1.702 x)): one pass instead of the three
// elementwise kernels of the expression (scalar multiply, sigmoid, multiply), evaluated in f32 and rounded once.
template <typename T>
__global__ __launch_bounds__(256) void k_quick_gelu(const T* __restrict__ x, int64_t nvec, T* __restrict__ out) {
    constexpr int N = VecIO<T>::N;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < nvec; e += stride) {
        float v[N];
        VecIO<T>::load(x + e * N, v);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
        VecIO<T>::store(out + e * N, v);
    }
}

// Row softmax of f32 scores into bf16 probabilities: P[r, :] = softmax(scale * S[r, :]).  The middle step of the VAE's
// single-head attention over 4096 positions with 512 channels (ldm's AttnBlock, reached from models/modeling/meta_arch/ldm.py:
// 448-482): at that head width the two products are plain large GEMMs (hipBLASLt, 0.4 ms each at 20 views) and the unfused
// form - f32 scores written once, read once here - beats the fused library kernel (2.6 ms) by ~1 ms per call; the scores
// stay f32 up to the exponential like in a flash kernel, only P is rounded to bf16 (as a flash kernel does before P V).
// One workgroup per row, the row held in registers (cols <= 256 * 4 * SM_VPT): one read + one write, HBM-bound.
constexpr int SM_VPT = 8;  // float4 per thread: rows up to 8192 columns
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ S, int32_t cols, float scale_log2e,
                                                      __hip_bfloat16* __restrict__ P) {
    __shared__ float red[4];
    const float* row = S + int64_t(blockIdx.x) * cols;
    __hip_bfloat16* out = P + int64_t(blockIdx.x) * cols;
    const int nv = cols >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 v[SM_VPT];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_VPT; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < nv) {
            v[i] = reinterpret_cast<const float4*>(row)[e];
            m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2e;  // scale > 0: max commutes with it
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_VPT; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < nv) {
            v[i].x = exp2f(fmaf(v[i].x, scale_log2e, -m));
            v[i].y = exp2f(fmaf(v[i].y, scale_log2e, -m));
            v[i].z = exp2f(fmaf(v[i].z, scale_log2e, -m));
            v[i].w = exp2f(fmaf(v[i].w, scale_log2e, -m));
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
    for (int i = 0; i < SM_VPT; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < nv) {
            __hip_bfloat16 o[4] = {__float2bfloat16(v[i].x * inv), __float2bfloat16(v[i].y * inv), __float2bfloat16(v[i].z * inv),
                                   __float2bfloat16(v[i].w * inv)};
            reinterpret_cast<uint2*>(out)[e] = *reinterpret_cast<const uint2*>(o);
        }
    }
}

static unsigned grid_for(int64_t nvec) {
    int64_t blocks = (nvec + 255) / 256;
    return unsigned(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_bias_residual_nhwc(const void* a, const void* b, const void* bias, int32_t dtype, int64_t pixels, int32_t C, void* out,
                                       void* stream) {
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "bias_residual: dtype must be 0 (f32) or 1 (bf16)");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(pixels >= 0 && C >= 1 && C % N == 0, "bias_residual: bad shape pixels=%lld C=%d", (long long)pixels, C);
    if (pixels == 0) return XM3D_OK;
    XM3D_REQUIRE(b && bias && out, "bias_residual: null pointer");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(bias) |
                   reinterpret_cast<uintptr_t>(out)) & 15) == 0, "bias_residual: tensors must be 16-byte aligned");
    const int64_t nvec = pixels * (C / N);
    hipStream_t s = as_stream(stream);
    if (dtype == 0)
        hipLaunchKernelGGL(k_bias_residual<float>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const float*>(a),
                           static_cast<const float*>(b), static_cast<const float*>(bias), nvec, C / N, static_cast<float*>(out));
    else
        hipLaunchKernelGGL(k_bias_residual<__hip_bfloat16>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const __hip_bfloat16*>(a),
                           static_cast<const __hip_bfloat16*>(b), static_cast<const __hip_bfloat16*>(bias), nvec, C / N,
                           static_cast<__hip_bfloat16*>(out));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_geglu(const void* x, int32_t dtype, int64_t rows, int32_t D, void* out, void* stream) {
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "geglu: dtype must be 0 (f32) or 1 (bf16)");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(rows >= 0 && D >= 1 && D % N == 0, "geglu: bad shape rows=%lld D=%d", (long long)rows, D);
    if (rows == 0) return XM3D_OK;
    XM3D_REQUIRE(x && out, "geglu: null pointer");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "geglu: tensors must be 16-byte aligned");
    const int64_t nvec = rows * (D / N);
    hipStream_t s = as_stream(stream);
    if (dtype == 0)
        hipLaunchKernelGGL(k_geglu<float>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const float*>(x), rows, D / N,
                           static_cast<float*>(out));
    else
        hipLaunchKernelGGL(k_geglu<__hip_bfloat16>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const __hip_bfloat16*>(x), rows,
                           D / N, static_cast<__hip_bfloat16*>(out));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_softmax_rows_f32_bf16(const float* scores, int64_t rows, int32_t cols, float scale, void* probs, void* stream) {
    XM3D_REQUIRE(rows >= 0 && rows < (1ll << 31) && cols >= 4 && cols % 4 == 0 && cols <= 256 * 4 * SM_VPT && scale > 0.f,
                 "softmax_rows: rows=%lld, cols=%d (multiple of 4, <= %d), scale=%g > 0 expected", (long long)rows, cols, 256 * 4 * SM_VPT,
                 double(scale));
    if (rows == 0) return XM3D_OK;
    XM3D_REQUIRE(scores && probs, "softmax_rows: null pointer");
    hipLaunchKernelGGL(k_softmax_rows, dim3(unsigned(rows)), dim3(256), 0, as_stream(stream), scores, cols, scale * 1.4426950408889634f,
                       static_cast<__hip_bfloat16*>(probs));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_quick_gelu(const void* x, int32_t dtype, int64_t numel, void* out, void* stream) {
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "quick_gelu: dtype must be 0 (f32) or 1 (bf16)");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(numel >= 0 && numel % N == 0, "quick_gelu: numel=%lld must be a multiple of %d", (long long)numel, N);
    if (numel == 0) return XM3D_OK;
    XM3D_REQUIRE(x && out, "quick_gelu: null pointer");
    const int64_t nvec = numel / N;
    hipStream_t s = as_stream(stream);
    if (dtype == 0)
        hipLaunchKernelGGL(k_quick_gelu<float>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const float*>(x), nvec,
                           static_cast<float*>(out));
    else
        hipLaunchKernelGGL(k_quick_gelu<__hip_bfloat16>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const __hip_bfloat16*>(x), nvec,
                           static_cast<__hip_bfloat16*>(out));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_pad_nhwc(const void* x, int32_t dtype, int64_t B, int32_t H, int32_t W, int32_t C, int32_t pad_bottom, int32_t pad_right,
                             void* out, void* stream) {
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "pad_nhwc: dtype must be 0 (f32) or 1 (bf16)");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(B >= 0 && H >= 1 && W >= 1 && C >= 1 && C % N == 0 && pad_bottom >= 0 && pad_right >= 0,
                 "pad_nhwc: bad shape B=%lld H=%d W=%d C=%d pad=(%d,%d)", (long long)B, H, W, C, pad_bottom, pad_right);
    if (B == 0) return XM3D_OK;
    XM3D_REQUIRE(x && out, "pad_nhwc: null pointer");
    const int Ho = H + pad_bottom, Wo = W + pad_right, vpp = C / N;
    const int64_t nvec = B * int64_t(Ho) * Wo * vpp;
    hipStream_t s = as_stream(stream);
    if (dtype == 0)
        hipLaunchKernelGGL(k_pad_nhwc<float>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const float*>(x), H, W, vpp, Ho, Wo, nvec,
                           static_cast<float*>(out));
    else
        hipLaunchKernelGGL(k_pad_nhwc<__hip_bfloat16>, dim3(grid_for(nvec)), dim3(256), 0, s, static_cast<const __hip_bfloat16*>(x), H, W, vpp,
                           Ho, Wo, nvec, static_cast<__hip_bfloat16*>(out));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
