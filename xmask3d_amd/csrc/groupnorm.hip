// Fused GroupNorm (+ optional SiLU) over NCHW activations, bf16 or f32 I/O, f32/f64 statistics.
// Replaces, inside the frozen SD VAE / UNet and the projection bottlenecks, the library sequence
// {row-wise moments, fused-params, elementwise affine, sigmoid, multiply} (GroupNorm(32,C) followed by swish:
// ldm's ResnetBlock / ResBlock / SpatialTransformer, reached from models/modeling/meta_arch/ldm.py:386-490) by two
// HBM-streaming passes:
//   k_gn_stats : each workgroup reduces one <=16 Ki-element slice of one (sample, group) - contiguous in NCHW -
//                with 16-byte loads, f32 lanes -> f64 wave/LDS reduction -> one partial pair per workgroup in its own slot;
//                k_gn_reduce sums the slots of a (sample, group) in FIXED order (no floating-point atomics anywhere: two runs
//                on the same input give the same bits, which the sharded == single-process inference test relies on)
//   k_gn_apply : y = act((x - mean) * rstd * gamma[c] + beta[c]), act = none | SiLU (1) | ReLU (2), 16 bytes per lane, the second read of x
//                mostly hits L2 / Infinity Cache for all but the 512x512 VAE maps
// Algorithmic bytes: 3 * numel * sizeof(T) (two reads, one write).
#include <hip/hip_bf16.h>

#include "common.h"
#include "vecio.h"

namespace xm3d {

constexpr int GN_SLICE = 16384;  // elements per workgroup in the statistics pass

// Moments of a (sample, group) from the per-workgroup partial pairs: one wave per (sample, group), lane i takes partials
// i, i + 64, ... in f64, then a fixed shuffle tree.  part is [(b * G + g)][n_part] float2 (sum, sum of squares).
__global__ __launch_bounds__(64) void k_gn_reduce(const float2* __restrict__ part, int n_part, double* __restrict__ stats) {
    const int64_t bg = blockIdx.x;
    const float2* p = part + bg * n_part;
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 64) {
        const float2 v = p[i];
        s += double(v.x);
        ss += double(v.y);
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    if (threadIdx.x == 0) {
        stats[bg * 2] = s;
        stats[bg * 2 + 1] = ss;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_stats(const T* __restrict__ x, int64_t group_elems, int slices,
                                                  float2* __restrict__ part) {
    constexpr int N = VecIO<T>::N;
    const int64_t bg = blockIdx.x / slices;
    const int slice = blockIdx.x % slices;
    const int64_t lo = int64_t(slice) * GN_SLICE;
    const int64_t hi = (lo + GN_SLICE < group_elems) ? lo + GN_SLICE : group_elems;
    const T* base = x + bg * group_elems;
    float s = 0.f, ss = 0.f;
    for (int64_t i = lo + int64_t(threadIdx.x) * N; i < hi; i += 256 * N) {
        float v[N];
        VecIO<T>::load(base + i, v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            s += v[j];
            ss = fmaf(v[j], v[j], ss);
        }
    }
    double ds = s, dss = ss;
    for (int off = 32; off > 0; off >>= 1) {
        ds += __shfl_xor(ds, off);
        dss += __shfl_xor(dss, off);
    }
    __shared__ double sm[8];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[wave * 2] = ds;
        sm[wave * 2 + 1] = dss;
    }
    __syncthreads();
    if (threadIdx.x == 0) part[bg * slices + slice] = make_float2(float(sm[0] + sm[2] + sm[4] + sm[6]), float(sm[1] + sm[3] + sm[5] + sm[7]));
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_apply(const T* __restrict__ x, const T* __restrict__ gamma, const T* __restrict__ beta,
                                                  const double* __restrict__ stats, int64_t nvec, int C, int hw, int cg,
                                                  float inv_elems, float eps, int silu, T* __restrict__ y) {
    constexpr int N = VecIO<T>::N;
    const int G = C / cg;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < nvec; e += stride) {
        const int64_t elem = e * N;
        const int64_t bc = elem / hw;  // (b*C + c): a vector never straddles channels (hw % N == 0)
        const int c = int(bc % C);
        const int64_t bg = (bc / C) * G + c / cg;
        const double m = stats[bg * 2] * inv_elems;
        const double var = stats[bg * 2 + 1] * inv_elems - m * m;
        const float mean = float(m), rstd = rsqrtf(fmaxf(float(var), 0.f) + eps);
        const float ga = gamma ? VecIO<T>::scalar(gamma + c) : 1.f, be = beta ? VecIO<T>::scalar(beta + c) : 0.f;
        const float a = rstd * ga, b = be - mean * a;
        float v[N];
        VecIO<T>::load(x + elem, v);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float t = fmaf(v[j], a, b);
            if (silu == 1) t = t / (1.f + __expf(-t));
            else if (silu == 2) t = fmaxf(t, 0.f);
            v[j] = t;
        }
        VecIO<T>::store(y + elem, v);
    }
}

// ---------------------------------------------------------------- NHWC (channels-last) variants
// x is (B, HW, C): a pixel's C channels are contiguous, so a 16-byte vector holds N consecutive channels of one pixel.
// Statistics: a workgroup owns a slab of pixels of one sample; thread t owns vector column (t % VPP) and walks pixels with
// stride (256 / VPP); the per-channel f32 sums of every thread go to their own LDS slot, are folded per group in a fixed order
// (gn_fold_store) and leave as ONE partial pair per (workgroup, group) in a slot of its own; k_gn_reduce sums a group's slots
// in fixed order.  Apply: per element group lookup (a vector may straddle two groups when C/G < N).

// red: [pixel lane][C] (sum, sum of squares) of this workgroup.  Thread t serves group t / hpg as helper t % hpg (hpg = a power of
// two <= 64 helpers per group, hpg * G <= 256): items helper, helper + hpg, ... of the group's lanes * cg slots, then a shuffle
// tree over the helpers - the order of the additions depends on nothing but the shape.
__device__ __forceinline__ void gn_fold_store(const float2* red, int lanes, int C, int cg, int G, int hpg, float2* __restrict__ part,
                                               int64_t b, int slabs, int slab) {
    const int g = threadIdx.x / hpg, hl = threadIdx.x % hpg;
    float s = 0.f, ss = 0.f;
    if (g < G) {
        const int items = lanes * cg;
        for (int i = hl; i < items; i += hpg) {
            const int pl = i / cg;
            const float2 v = red[pl * C + g * cg + (i - pl * cg)];
            s += v.x;
            ss += v.y;
        }
    }
    for (int off = hpg >> 1; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    if (g < G && hl == 0) part[(b * G + g) * slabs + slab] = make_float2(s, ss);
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_stats_nhwc(const T* __restrict__ x, const T* __restrict__ shift, int shift_bstride, int hw, int C, int cg, int G, int slabs, int pix,
                                                       int hpg, float2* __restrict__ part) {
    constexpr int N = VecIO<T>::N;
    extern __shared__ __attribute__((aligned(16))) float2 gn_red[];  // [pixel lane][C]
    const int b = blockIdx.x / slabs, slab = blockIdx.x % slabs;
    const int vpp = C / N;                         // vectors per pixel
    const int p0 = slab * pix;
    const int p1 = (p0 + pix < hw) ? p0 + pix : hw;
    const T* base = x + int64_t(b) * hw * C;
    for (int v = threadIdx.x % (vpp < 256 ? vpp : 256); v < vpp; v += 256) {  // vpp > 256 only for C > 2048 (bf16)
        const int lanes = vpp < 256 ? 256 / vpp : 1;  // pixel lanes sharing this vector column
        const int pl = vpp < 256 ? threadIdx.x / vpp : 0;
        if (pl >= lanes) continue;
        float s[N], ss[N], sh[N];
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] = ss[j] = sh[j] = 0.f;
        if (shift) VecIO<T>::load(shift + int64_t(b) * shift_bstride + v * N, sh);
        int p = p0 + pl;
        for (; p + 3 * lanes < p1; p += 4 * lanes) {  // four independent 16-byte loads in flight per lane
            float val[4][N];
#pragma unroll
            for (int u = 0; u < 4; ++u) VecIO<T>::load(base + int64_t(p + u * lanes) * C + v * N, val[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float t = val[u][j] + sh[j];
                    s[j] += t;
                    ss[j] = fmaf(t, t, ss[j]);
                }
        }
        for (; p < p1; p += lanes) {
            float val[N];
            VecIO<T>::load(base + int64_t(p) * C + v * N, val);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                val[j] += sh[j];
                s[j] += val[j];
                ss[j] = fmaf(val[j], val[j], ss[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) gn_red[pl * C + v * N + j] = make_float2(s[j], ss[j]);
    }
    __syncthreads();
    gn_fold_store(gn_red, vpp < 256 ? 256 / vpp : 1, C, cg, G, hpg, part, b, slabs, slab);
}

// out = a + b + bias[c] (the residual add that closes a ResBlock, k_bias_residual of pointwise.hip) with the GroupNorm
// statistics of `out` accumulated on the way: the next block's first GroupNorm then skips its statistics pass (one read of the
// activation saved).  Same thread mapping as k_gn_stats_nhwc; the sums are taken over the values AS STORED (rounded to T).
template <typename T>
__global__ __launch_bounds__(256) void k_bias_residual_stats(const T* __restrict__ a, const T* __restrict__ bsrc, const T* __restrict__ bias, int hw,
                                                             int C, int cg, int G, int slabs, int pix, int hpg, T* __restrict__ out,
                                                             float2* __restrict__ part) {
    constexpr int N = VecIO<T>::N;
    extern __shared__ __attribute__((aligned(16))) float2 gn_red[];  // [pixel lane][C]
    const int b = blockIdx.x / slabs, slab = blockIdx.x % slabs;
    const int vpp = C / N;
    const int p0 = slab * pix;
    const int p1 = (p0 + pix < hw) ? p0 + pix : hw;
    const int64_t base = int64_t(b) * hw * C;
    for (int v = threadIdx.x % (vpp < 256 ? vpp : 256); v < vpp; v += 256) {
        const int lanes = vpp < 256 ? 256 / vpp : 1;
        const int pl = vpp < 256 ? threadIdx.x / vpp : 0;
        if (pl >= lanes) continue;
        float s[N], ss[N], bi[N];
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] = ss[j] = bi[j] = 0.f;
        if (bias) VecIO<T>::load(bias + v * N, bi);
        for (int p = p0 + pl; p < p1; p += lanes) {
            const int64_t off = base + int64_t(p) * C + v * N;
            float val[N], sk[N];
            VecIO<T>::load(bsrc + off, val);
            if (a) VecIO<T>::load(a + off, sk);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float t = val[j] + bi[j];
                if (a) t += sk[j];
                val[j] = t;
            }
            VecIO<T>::store(out + off, val);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float t = sizeof(T) == 2 ? __bfloat162float(__float2bfloat16(val[j])) : val[j];
                s[j] += t;
                ss[j] = fmaf(t, t, ss[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) gn_red[pl * C + v * N + j] = make_float2(s[j], ss[j]);
    }
    __syncthreads();
    gn_fold_store(gn_red, vpp < 256 ? 256 / vpp : 1, C, cg, G, hpg, part, b, slabs, slab);
}

// Apply pass, same thread mapping as the statistics pass: a thread owns one 16-byte channel vector column, so the
// per-channel terms fold into y = act(x * a + c) with a = rstd * gamma, c = (shift - mean) * rstd * gamma + beta computed
// ONCE per thread; the pixel loop is load - fma - (exp) - store with no index arithmetic (a flat grid-stride loop paid two
// 64-bit divisions and two f64 statistics loads per vector: 1.8 TB/s instead of HBM speed).
template <typename T>
__global__ __launch_bounds__(256) void k_gn_apply_nhwc(const T* __restrict__ x, const T* __restrict__ shift, int shift_bstride,
                                                       const T* __restrict__ gamma, const T* __restrict__ beta, const double* __restrict__ stats,
                                                       int hw, int C, int cg, int G, int slabs, int pix, float inv_elems, float eps, int silu,
                                                       const T* __restrict__ residual, T* __restrict__ y) {
    constexpr int N = VecIO<T>::N;
    const int b = blockIdx.x / slabs, slab = blockIdx.x % slabs;
    const int vpp = C / N;
    const int p0 = slab * pix;
    const int p1 = (p0 + pix < hw) ? p0 + pix : hw;
    const T* xb = x + int64_t(b) * hw * C;
    const T* rb = residual ? residual + int64_t(b) * hw * C : nullptr;  // added after the affine, before the activation
    T* yb = y + int64_t(b) * hw * C;
    for (int v = threadIdx.x % (vpp < 256 ? vpp : 256); v < vpp; v += 256) {
        const int lanes = vpp < 256 ? 256 / vpp : 1;
        const int pl = vpp < 256 ? threadIdx.x / vpp : 0;
        if (pl >= lanes) continue;
        float a[N], c[N], ga[N], be[N], sh[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            ga[j] = 1.f;
            be[j] = sh[j] = 0.f;
        }
        if (gamma) VecIO<T>::load(gamma + v * N, ga);
        if (beta) VecIO<T>::load(beta + v * N, be);
        if (shift) VecIO<T>::load(shift + int64_t(b) * shift_bstride + v * N, sh);
        int gprev = -1;
        float mean = 0.f, rstd = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int g = (v * N + j) / cg;
            if (g != gprev) {
                const double m = stats[(int64_t(b) * G + g) * 2] * inv_elems;
                const double var = stats[(int64_t(b) * G + g) * 2 + 1] * inv_elems - m * m;
                mean = float(m);
                rstd = rsqrtf(fmaxf(float(var), 0.f) + eps);
                gprev = g;
            }
            a[j] = rstd * ga[j];
            c[j] = fmaf(sh[j] - mean, a[j], be[j]);
        }
        int p = p0 + pl;
        if (!rb) {
            for (; p + 3 * lanes < p1; p += 4 * lanes) {  // four independent 16-byte loads in flight per lane
                float val[4][N];
#pragma unroll
                for (int u = 0; u < 4; ++u) VecIO<T>::load(xb + int64_t(p + u * lanes) * C + v * N, val[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        float t = fmaf(val[u][j], a[j], c[j]);
                        if (silu == 1) t = t / (1.f + __expf(-t));
                        else if (silu == 2) t = fmaxf(t, 0.f);
                        val[u][j] = t;
                    }
                    VecIO<T>::store(yb + int64_t(p + u * lanes) * C + v * N, val[u]);
                }
            }
        }
        for (; p < p1; p += lanes) {
            float val[N], res[N];
            VecIO<T>::load(xb + int64_t(p) * C + v * N, val);
            if (rb) VecIO<T>::load(rb + int64_t(p) * C + v * N, res);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float t = fmaf(val[j], a[j], c[j]);
                if (rb) t += res[j];
                if (silu == 1) t = t / (1.f + __expf(-t));
                else if (silu == 2) t = fmaxf(t, 0.f);
                val[j] = t;
            }
            VecIO<T>::store(yb + int64_t(p) * C + v * N, val);
        }
    }
}

// Launch plan of the channels-last statistics kernels: pixels per workgroup (aim at ~1024 workgroups, at least one pixel per
// pixel lane, at most 256), slabs per image, helpers per group of the fold, LDS bytes.  The partial pairs live behind the B*G*2
// moments in the caller's statistics buffer (xm3d_gn_stats_doubles_nhwc doubles in all).
struct GnPlan {
    int pix, slabs, hpg, lds;
};
static GnPlan gn_plan_nhwc(int64_t B, int C, int hw, int G, int nvec) {
    GnPlan p;
    const int vpp = C / nvec;
    const int lanes = vpp < 256 ? 256 / vpp : 1;
    int64_t pix = (B * int64_t(hw) + 1023) / 1024;
    if (pix < lanes) pix = lanes;
    if (pix > 256) pix = 256;
    if (pix > hw) pix = hw;
    p.pix = int(pix);
    p.slabs = int((hw + pix - 1) / pix);
    p.hpg = 1;
    while (p.hpg * 2 * G <= 256 && p.hpg < 64) p.hpg *= 2;
    p.lds = lanes * C * int(sizeof(float2));
    return p;
}
static float2* gn_partials(double* stats, int64_t B, int G) { return reinterpret_cast<float2*>(stats + B * G * 2); }

template <typename T>
static int gn_launch_nhwc(const void* x, const void* shift, int shift_bstride, int64_t B, int C, int hw, int G, const void* gamma, const void* beta, float eps, int silu,
                          const void* residual, void* y, double* stats, hipStream_t s, bool have_stats = false) {
    const int cg = C / G;
    const GnPlan p = gn_plan_nhwc(B, C, hw, G, VecIO<T>::N);
    if (!have_stats) {
        float2* part = gn_partials(stats, B, G);
        hipLaunchKernelGGL(k_gn_stats_nhwc<T>, dim3(unsigned(B * p.slabs)), dim3(256), p.lds, s, static_cast<const T*>(x), static_cast<const T*>(shift), shift_bstride, hw, C, cg, G, p.slabs, p.pix, p.hpg, part);
        hipLaunchKernelGGL(k_gn_reduce, dim3(unsigned(B * G)), dim3(64), 0, s, part, p.slabs, stats);
    }
    hipLaunchKernelGGL(k_gn_apply_nhwc<T>, dim3(unsigned(B * p.slabs)), dim3(256), 0, s, static_cast<const T*>(x), static_cast<const T*>(shift),
                       shift_bstride, static_cast<const T*>(gamma), static_cast<const T*>(beta), stats, hw, C, cg, G, p.slabs, p.pix,
                       1.0f / (float(cg) * float(hw)), eps, silu, static_cast<const T*>(residual), static_cast<T*>(y));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

template <typename T>
static int bias_residual_stats_launch(const void* a, const void* b, const void* bias, int64_t B, int C, int hw, int G, void* out, double* stats,
                                      hipStream_t s) {
    const int cg = C / G;
    const GnPlan p = gn_plan_nhwc(B, C, hw, G, VecIO<T>::N);
    float2* part = gn_partials(stats, B, G);
    hipLaunchKernelGGL(k_bias_residual_stats<T>, dim3(unsigned(B * p.slabs)), dim3(256), p.lds, s, static_cast<const T*>(a), static_cast<const T*>(b),
                       static_cast<const T*>(bias), hw, C, cg, G, p.slabs, p.pix, p.hpg, static_cast<T*>(out), part);
    hipLaunchKernelGGL(k_gn_reduce, dim3(unsigned(B * G)), dim3(64), 0, s, part, p.slabs, stats);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

template <typename T>
static int gn_launch(const void* x, int64_t B, int C, int hw, int G, const void* gamma, const void* beta, float eps, int silu,
                     void* y, double* stats, hipStream_t s) {
    const int cg = C / G;
    const int64_t group_elems = int64_t(cg) * hw;
    const int slices = int((group_elems + GN_SLICE - 1) / GN_SLICE);
    float2* part = gn_partials(stats, B, G);
    hipLaunchKernelGGL(k_gn_stats<T>, dim3(unsigned(B * G * slices)), dim3(256), 0, s, static_cast<const T*>(x), group_elems, slices, part);
    hipLaunchKernelGGL(k_gn_reduce, dim3(unsigned(B * G)), dim3(64), 0, s, part, slices, stats);
    const int64_t nvec = B * C * int64_t(hw) / VecIO<T>::N;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gn_apply<T>, dim3(unsigned(blocks)), dim3(256), 0, s, static_cast<const T*>(x),
                       static_cast<const T*>(gamma), static_cast<const T*>(beta), stats, nvec, C, hw, cg, 1.0f / float(group_elems), eps, silu, static_cast<T*>(y));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_group_norm(const void* x, int32_t dtype, int64_t B, int32_t C, int32_t hw, int32_t G, const void* gamma,
                               const void* beta, float eps, int32_t silu, void* y, double* stats_ws, void* stream) {
    XM3D_REQUIRE(B >= 0 && C >= 1 && hw >= 1 && G >= 1 && C % G == 0, "group_norm: bad shape B=%lld C=%d hw=%d G=%d", (long long)B, C, hw, G);
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "group_norm: dtype must be 0 (f32) or 1 (bf16)");
    if (B == 0) return XM3D_OK;
    XM3D_REQUIRE(x && y && stats_ws, "group_norm: null pointer");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(hw % N == 0, "group_norm: H*W=%d must be a multiple of %d", hw, N);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "group_norm: x/y must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    if (dtype == 0) return gn_launch<float>(x, B, C, hw, G, gamma, beta, eps, silu, y, stats_ws, s);
    return gn_launch<__hip_bfloat16>(x, B, C, hw, G, gamma, beta, eps, silu, y, stats_ws, s);
}

template <typename T>
static int gn_stats_only_nhwc(const void* x, const void* shift, int shift_bstride, int64_t B, int C, int hw, int G, double* stats, hipStream_t s) {
    const int cg = C / G;
    const GnPlan p = gn_plan_nhwc(B, C, hw, G, VecIO<T>::N);
    float2* part = gn_partials(stats, B, G);
    hipLaunchKernelGGL(k_gn_stats_nhwc<T>, dim3(unsigned(B * p.slabs)), dim3(256), p.lds, s, static_cast<const T*>(x), static_cast<const T*>(shift), shift_bstride, hw, C, cg, G, p.slabs, p.pix, p.hpg, part);
    hipLaunchKernelGGL(k_gn_reduce, dim3(unsigned(B * G)), dim3(64), 0, s, part, p.slabs, stats);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int64_t xm3d_gn_stats_doubles_nhwc(int64_t B, int32_t C, int32_t hw, int32_t G, int32_t dtype) {
    if (B <= 0 || C <= 0 || hw <= 0 || G <= 0 || C % G != 0 || C % (dtype == 0 ? 4 : 8) != 0) return B > 0 && G > 0 ? B * G * 2 : 0;
    return B * G * 2 + B * G * int64_t(gn_plan_nhwc(B, C, hw, G, dtype == 0 ? 4 : 8).slabs);
}

extern "C" int64_t xm3d_gn_stats_doubles_nchw(int64_t B, int32_t C, int32_t hw, int32_t G) {
    if (B <= 0 || C <= 0 || hw <= 0 || G <= 0 || C % G != 0) return 0;
    return B * G * 2 + B * G * ((int64_t(C / G) * hw + GN_SLICE - 1) / GN_SLICE);
}

extern "C" int xm3d_group_norm_nhwc_stats(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C, int32_t hw,
                                          int32_t G, double* stats, void* stream) {
    XM3D_REQUIRE(B >= 0 && C >= 1 && C <= 8192 && hw >= 1 && G >= 1 && G <= 64 && C % G == 0, "group_norm_nhwc_stats: bad shape B=%lld C=%d hw=%d G=%d",
                 (long long)B, C, hw, G);
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "group_norm_nhwc_stats: dtype must be 0 (f32) or 1 (bf16)");
    if (B == 0) return XM3D_OK;
    XM3D_REQUIRE(x && stats, "group_norm_nhwc_stats: null pointer");
    XM3D_REQUIRE(C % (dtype == 0 ? 4 : 8) == 0, "group_norm_nhwc_stats: C=%d must be a multiple of %d", C, dtype == 0 ? 4 : 8);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0, "group_norm_nhwc_stats: tensors must be 16-byte aligned");
    XM3D_REQUIRE(shift_bstride == 0 || shift_bstride == C, "group_norm_nhwc_stats: shift_bstride must be 0 (shared) or C (per sample)");
    hipStream_t s = as_stream(stream);
    if (dtype == 0) return gn_stats_only_nhwc<float>(x, shift, shift_bstride, B, C, hw, G, stats, s);
    return gn_stats_only_nhwc<__hip_bfloat16>(x, shift, shift_bstride, B, C, hw, G, stats, s);
}

extern "C" int xm3d_group_norm_nhwc_res(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C,
                                        int32_t hw, int32_t G, const void* gamma, const void* beta, float eps, int32_t silu,
                                        const void* residual, void* y, double* stats_ws, void* stream) {
    XM3D_REQUIRE(B >= 0 && C >= 1 && C <= 8192 && hw >= 1 && G >= 1 && G <= 64 && C % G == 0, "group_norm_nhwc: bad shape B=%lld C=%d hw=%d G=%d",
                 (long long)B, C, hw, G);
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "group_norm_nhwc: dtype must be 0 (f32) or 1 (bf16)");
    if (B == 0) return XM3D_OK;
    XM3D_REQUIRE(x && y && stats_ws, "group_norm_nhwc: null pointer");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(C % N == 0, "group_norm_nhwc: C=%d must be a multiple of %d", C, N);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
                   reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0,
                 "group_norm_nhwc: tensors must be 16-byte aligned");
    XM3D_REQUIRE(shift_bstride == 0 || shift_bstride == C, "group_norm_nhwc: shift_bstride must be 0 (shared) or C (per sample)");
    hipStream_t s = as_stream(stream);
    if (dtype == 0) return gn_launch_nhwc<float>(x, shift, shift_bstride, B, C, hw, G, gamma, beta, eps, silu, residual, y, stats_ws, s);
    return gn_launch_nhwc<__hip_bfloat16>(x, shift, shift_bstride, B, C, hw, G, gamma, beta, eps, silu, residual, y, stats_ws, s);
}

extern "C" int xm3d_group_norm_nhwc(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C, int32_t hw,
                                    int32_t G, const void* gamma, const void* beta, float eps, int32_t silu, void* y, double* stats_ws,
                                    void* stream) {
    return xm3d_group_norm_nhwc_res(x, shift, shift_bstride, dtype, B, C, hw, G, gamma, beta, eps, silu, nullptr, y, stats_ws, stream);
}

extern "C" int xm3d_bias_residual_stats_nhwc(const void* a, const void* b, const void* bias, int32_t dtype, int64_t B, int32_t C, int32_t hw,
                                             int32_t G, void* out, double* stats, void* stream) {
    XM3D_REQUIRE(B >= 0 && C >= 1 && C <= 8192 && hw >= 1 && G >= 1 && G <= 64 && C % G == 0, "bias_residual_stats: bad shape B=%lld C=%d hw=%d G=%d",
                 (long long)B, C, hw, G);
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "bias_residual_stats: dtype must be 0 (f32) or 1 (bf16)");
    if (B == 0) return XM3D_OK;
    XM3D_REQUIRE(b && out && stats, "bias_residual_stats: null pointer");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(C % N == 0, "bias_residual_stats: C=%d must be a multiple of %d", C, N);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(bias) |
                   reinterpret_cast<uintptr_t>(out)) & 15) == 0, "bias_residual_stats: tensors must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    if (dtype == 0) return bias_residual_stats_launch<float>(a, b, bias, B, C, hw, G, out, stats, s);
    return bias_residual_stats_launch<__hip_bfloat16>(a, b, bias, B, C, hw, G, out, stats, s);
}

extern "C" int xm3d_group_norm_nhwc_apply(const void* x, const void* shift, int32_t shift_bstride, int32_t dtype, int64_t B, int32_t C,
                                          int32_t hw, int32_t G, const void* gamma, const void* beta, float eps, int32_t silu,
                                          const void* residual, void* y, const double* stats, void* stream) {
    XM3D_REQUIRE(B >= 0 && C >= 1 && C <= 8192 && hw >= 1 && G >= 1 && G <= 64 && C % G == 0, "group_norm_nhwc_apply: bad shape B=%lld C=%d hw=%d G=%d",
                 (long long)B, C, hw, G);
    XM3D_REQUIRE(dtype == 0 || dtype == 1, "group_norm_nhwc_apply: dtype must be 0 (f32) or 1 (bf16)");
    if (B == 0) return XM3D_OK;
    XM3D_REQUIRE(x && y && stats, "group_norm_nhwc_apply: null pointer");
    const int N = dtype == 0 ? 4 : 8;
    XM3D_REQUIRE(C % N == 0, "group_norm_nhwc_apply: C=%d must be a multiple of %d", C, N);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) |
                   reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0,
                 "group_norm_nhwc_apply: tensors must be 16-byte aligned");
    XM3D_REQUIRE(shift_bstride == 0 || shift_bstride == C, "group_norm_nhwc_apply: shift_bstride must be 0 (shared) or C (per sample)");
    hipStream_t s = as_stream(stream);
    double* st = const_cast<double*>(stats);
    if (dtype == 0) return gn_launch_nhwc<float>(x, shift, shift_bstride, B, C, hw, G, gamma, beta, eps, silu, residual, y, st, s, true);
    return gn_launch_nhwc<__hip_bfloat16>(x, shift, shift_bstride, B, C, hw, G, gamma, beta, eps, silu, residual, y, st, s, true);
}
