// Multi-scale deformable attention sampling, forward and backward, for gfx950.
//
// Semantics follow the reference device code (ops/src/cuda/ms_deform_im2col_cuda.cuh:38-89 tap
// rule, :92-157 backward of one sample, :242-304 accumulation over levels x points) but the work
// decomposition is different: a group of D/4 adjacent lanes owns one (batch, query, head) and each
// lane owns FOUR consecutive channels (one channel per lane when D is not a multiple of 4), so every bilinear tap is one coalesced 16-byte load per lane
// (a 128-byte row segment per head for D = 32) instead of 32 scalar loads, the sampling location /
// weight of a sample is read once per lane group, and the backward reduces grad_loc / grad_attn
// over channels with wave shuffles instead of a serial shared-memory loop.  grad_value is
// accumulated with f32 atomics exactly like the reference (order-dependent in the last bits).
#include "common.h"

namespace xm3d {

// T = float or double: the reference op is instantiated for both (AT_DISPATCH_FLOATING_TYPES,
// ops/src/cuda/ms_deform_attn_cuda.cu:64,134; its gradcheck runs in double, ops/test.py:66-81).
template <typename T, int V>
struct Vec {
    T v[V];
    __device__ T& operator[](int i) { return v[i]; }
    __device__ const T& operator[](int i) const { return v[i]; }
};
template <typename T, int V>
__device__ inline Vec<T, V> vzero() {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = T(0);
    return r;
}
template <typename T, int V>
__device__ inline Vec<T, V> vload(const T* p) {
    Vec<T, V> r;
    if constexpr (V == 4 && sizeof(T) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else if constexpr (V == 4 && sizeof(T) == 8) {
        const double2 a = reinterpret_cast<const double2*>(p)[0], b = reinterpret_cast<const double2*>(p)[1];
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) r.v[i] = p[i];
    }
    return r;
}
template <typename T, int V>
__device__ inline void vstore(T* p, const Vec<T, V>& r) {
    if constexpr (V == 4 && sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    } else if constexpr (V == 4 && sizeof(T) == 8) {
        reinterpret_cast<double2*>(p)[0] = make_double2(r.v[0], r.v[1]);
        reinterpret_cast<double2*>(p)[1] = make_double2(r.v[2], r.v[3]);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = r.v[i];
    }
}

template <typename T>
struct Tap {
    int64_t off[4];  // element offset of the 4 corners (channel 0 of this head), -1 if outside
    T w[4];
    T lh, lw, hh, hw;
};

template <typename T>
__device__ inline bool make_tap(T loc_x, T loc_y, int H, int W, int64_t level_start, int row_stride, Tap<T>& t) {
    const T h_im = loc_y * H - T(0.5);
    const T w_im = loc_x * W - T(0.5);
    if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) return false;
    const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    t.lh = h_im - h_low;
    t.lw = w_im - w_low;
    t.hh = 1 - t.lh;
    t.hw = 1 - t.lw;
    t.w[0] = t.hh * t.hw;
    t.w[1] = t.hh * t.lw;
    t.w[2] = t.lh * t.hw;
    t.w[3] = t.lh * t.lw;
    const bool hl = h_low >= 0, hh_ok = h_high <= H - 1, wl = w_low >= 0, wh = w_high <= W - 1;
    t.off[0] = (hl && wl) ? (level_start + int64_t(h_low) * W + w_low) * row_stride : -1;
    t.off[1] = (hl && wh) ? (level_start + int64_t(h_low) * W + w_high) * row_stride : -1;
    t.off[2] = (hh_ok && wl) ? (level_start + int64_t(h_high) * W + w_low) * row_stride : -1;
    t.off[3] = (hh_ok && wh) ? (level_start + int64_t(h_high) * W + w_high) * row_stride : -1;
    return true;
}

// one thread = (b, q, h, group of V channels); V = 4 when D % 4 == 0 (16-byte taps), else 1
template <typename T, int V>
__global__ void k_msda_fwd(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                           const int64_t* __restrict__ lstart, const T* __restrict__ loc,
                           const T* __restrict__ attn, int B, int S, int H, int D, int L, int Lq, int P,
                           T* __restrict__ out) {
    const int QD = D / V;
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t total = int64_t(B) * Lq * H * QD;
    if (e >= total) return;
    const int cq = int(e % QD);
    const int64_t bqh = e / QD;  // (b*Lq + q)*H + h
    const int h = int(bqh % H);
    const int64_t bq = bqh / H;
    const int b = int(bq / Lq);
    const int row_stride = H * D;
    const T* vbase = value + int64_t(b) * S * row_stride + h * D + cq * V;
    const T* lp = loc + bqh * (int64_t(L) * P * 2);
    const T* ap = attn + bqh * (int64_t(L) * P);
    Vec<T, V> acc = vzero<T, V>();
    for (int l = 0; l < L; ++l) {
        const int Hl = int(shapes[2 * l]), Wl = int(shapes[2 * l + 1]);
        const int64_t ls = lstart[l];
        for (int p = 0; p < P; ++p) {
            const T lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
            const T aw = ap[l * P + p];
            Tap<T> t;
            if (!make_tap<T>(lx, ly, Hl, Wl, ls, row_stride, t)) continue;
            Vec<T, V> sv = vzero<T, V>();
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (t.off[c] >= 0) {
                    const Vec<T, V> x = vload<T, V>(vbase + t.off[c]);
#pragma unroll
                    for (int i = 0; i < V; ++i) sv[i] += t.w[c] * x[i];
                }
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] += aw * sv[i];
        }
    }
    vstore<T, V>(out + bqh * D + cq * V, acc);
}

template <typename T, int V, bool SHUFFLE>
__global__ void k_msda_bwd(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                           const int64_t* __restrict__ lstart, const T* __restrict__ loc,
                           const T* __restrict__ attn, const T* __restrict__ gout, int B, int S, int H, int D,
                           int L, int Lq, int P, T* __restrict__ gvalue, T* __restrict__ gloc,
                           T* __restrict__ gattn) {
    const int QD = D / V;
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t total = int64_t(B) * Lq * H * QD;
    const bool live = e < total;
    const int64_t ee = live ? e : total - 1;  // keep every lane in the shuffles
    const int cq = int(ee % QD);
    const int64_t bqh = ee / QD;
    const int h = int(bqh % H);
    const int64_t bq = bqh / H;
    const int b = int(bq / Lq);
    const int row_stride = H * D;
    const int64_t voff = int64_t(b) * S * row_stride + h * D + cq * V;
    const T* lp = loc + bqh * (int64_t(L) * P * 2);
    const T* ap = attn + bqh * (int64_t(L) * P);
    Vec<T, V> go = vload<T, V>(gout + bqh * D + cq * V);
    if (!live) go = vzero<T, V>();
    for (int l = 0; l < L; ++l) {
        const int Hl = int(shapes[2 * l]), Wl = int(shapes[2 * l + 1]);
        const int64_t ls = lstart[l];
        for (int p = 0; p < P; ++p) {
            const T lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
            const T aw = ap[l * P + p];
            Tap<T> t;
            const bool inside = make_tap<T>(lx, ly, Hl, Wl, ls, row_stride, t);  // uniform within a lane group
            T g_attn = 0, g_x = 0, g_y = 0;
            if (inside) {
                Vec<T, V> v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = vzero<T, V>();
                    if (t.off[c] >= 0) {
                        v[c] = vload<T, V>(value + voff + t.off[c]);
                        if (live) {
                            T* dst = gvalue + voff + t.off[c];
                            const T wa = t.w[c] * aw;
#pragma unroll
                            for (int i = 0; i < V; ++i) atomicAdd(dst + i, wa * go[i]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const T sampled = t.w[0] * v[0][i] + t.w[1] * v[1][i] + t.w[2] * v[2][i] + t.w[3] * v[3][i];
                    const T gh = -t.hw * v[0][i] - t.lw * v[1][i] + t.hw * v[2][i] + t.lw * v[3][i];
                    const T gw = -t.hh * v[0][i] + t.hh * v[1][i] - t.lh * v[2][i] + t.lh * v[3][i];
                    const T top = aw * go[i];
                    g_attn += go[i] * sampled;
                    g_x += gw * top;
                    g_y += gh * top;
                }
                g_x *= Wl;
                g_y *= Hl;
            }
            const int64_t sidx = bqh * (int64_t(L) * P) + l * P + p;
            if (SHUFFLE) {
                for (int off = QD >> 1; off > 0; off >>= 1) {
                    g_attn += __shfl_xor(g_attn, off);
                    g_x += __shfl_xor(g_x, off);
                    g_y += __shfl_xor(g_y, off);
                }
                if (live && cq == 0) {
                    gattn[sidx] = g_attn;
                    gloc[sidx * 2] = g_x;
                    gloc[sidx * 2 + 1] = g_y;
                }
            } else if (live && inside) {
                atomicAdd(&gattn[sidx], g_attn);
                atomicAdd(&gloc[sidx * 2], g_x);
                atomicAdd(&gloc[sidx * 2 + 1], g_y);
            }
        }
    }
}

static int check_msda(int B, int S, int H, int D, int L, int Lq, int P) {
    XM3D_REQUIRE(B >= 0 && S >= 1 && H >= 1 && D >= 1 && L >= 1 && Lq >= 0 && P >= 1, "msda: bad sizes");
    return XM3D_OK;
}

}  // namespace xm3d

using namespace xm3d;

template <typename T>
static int msda_forward_t(const T* value, const int64_t* spatial_shapes, const int64_t* level_start, const T* loc, const T* attn,
                          int32_t B, int32_t S, int32_t H, int32_t D, int32_t L, int32_t Lq, int32_t P, T* out, void* stream) {
    int rc = check_msda(B, S, H, D, L, Lq, P);
    if (rc) return rc;
    if (int64_t(B) * Lq == 0) return XM3D_OK;
    XM3D_REQUIRE(value && spatial_shapes && level_start && loc && attn && out, "msda_forward: null pointer");
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int64_t total = int64_t(B) * Lq * H * (vec4 ? D / 4 : D);
    if (vec4)
        hipLaunchKernelGGL((k_msda_fwd<T, 4>), dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), value, spatial_shapes,
                           level_start, loc, attn, B, S, H, D, L, Lq, P, out);
    else
        hipLaunchKernelGGL((k_msda_fwd<T, 1>), dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), value, spatial_shapes,
                           level_start, loc, attn, B, S, H, D, L, Lq, P, out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

template <typename T>
static int msda_backward_t(const T* value, const int64_t* spatial_shapes, const int64_t* level_start, const T* loc, const T* attn,
                           const T* grad_out, int32_t B, int32_t S, int32_t H, int32_t D, int32_t L, int32_t Lq, int32_t P,
                           T* grad_value, T* grad_loc, T* grad_attn, void* stream) {
    int rc = check_msda(B, S, H, D, L, Lq, P);
    if (rc) return rc;
    if (int64_t(B) * Lq == 0) return XM3D_OK;
    XM3D_REQUIRE(value && spatial_shapes && level_start && loc && attn && grad_out && grad_value && grad_loc && grad_attn,
                 "msda_backward: null pointer");
    const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(grad_out) |
                                        reinterpret_cast<uintptr_t>(grad_value)) & 15) == 0;
    const int QD = vec4 ? D / 4 : D;
    const int64_t total = int64_t(B) * Lq * H * QD;
    const bool pow2 = (QD & (QD - 1)) == 0 && QD <= 64;
    dim3 grid((total + 255) / 256), blk(256);
#define XM3D_BWD(V, SH)                                                                                                 \
    hipLaunchKernelGGL((k_msda_bwd<T, V, SH>), grid, blk, 0, as_stream(stream), value, spatial_shapes, level_start, loc, \
                       attn, grad_out, B, S, H, D, L, Lq, P, grad_value, grad_loc, grad_attn)
    if (vec4 && pow2) XM3D_BWD(4, true);
    else if (vec4) XM3D_BWD(4, false);
    else if (pow2) XM3D_BWD(1, true);
    else XM3D_BWD(1, false);
#undef XM3D_BWD
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_msda_forward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                 const float* loc, const float* attn, int32_t B, int32_t S, int32_t H, int32_t D,
                                 int32_t L, int32_t Lq, int32_t P, float* out, void* stream) {
    return msda_forward_t<float>(value, spatial_shapes, level_start, loc, attn, B, S, H, D, L, Lq, P, out, stream);
}

extern "C" int xm3d_msda_backward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                  const float* loc, const float* attn, const float* grad_out, int32_t B, int32_t S,
                                  int32_t H, int32_t D, int32_t L, int32_t Lq, int32_t P, float* grad_value,
                                  float* grad_loc, float* grad_attn, void* stream) {
    return msda_backward_t<float>(value, spatial_shapes, level_start, loc, attn, grad_out, B, S, H, D, L, Lq, P, grad_value, grad_loc,
                                  grad_attn, stream);
}

extern "C" int xm3d_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                     const double* loc, const double* attn, int32_t B, int32_t S, int32_t H, int32_t D,
                                     int32_t L, int32_t Lq, int32_t P, double* out, void* stream) {
    return msda_forward_t<double>(value, spatial_shapes, level_start, loc, attn, B, S, H, D, L, Lq, P, out, stream);
}

extern "C" int xm3d_msda_backward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                      const double* loc, const double* attn, const double* grad_out, int32_t B, int32_t S,
                                      int32_t H, int32_t D, int32_t L, int32_t Lq, int32_t P, double* grad_value,
                                      double* grad_loc, double* grad_attn, void* stream) {
    return msda_backward_t<double>(value, spatial_shapes, level_start, loc, attn, grad_out, B, S, H, D, L, Lq, P, grad_value,
                                   grad_loc, grad_attn, stream);
}
