// Backward of LayerNorm and GroupNorm(+SiLU / ReLU) for the TRAINING iteration (the reference trains these through
// torch's autograd: /root/reference/run/train.py:504-540 over nn.LayerNorm / nn.GroupNorm of
// /root/reference/models/modeling/transformer_decoder/mask2former_transformer_decoder.py:17-178, .../pixel_decoder/msdeformattn.py:35-60,
// .../backbone/feature_extractor.py:40-47 and the frozen UNet's blocks, meta_arch/ldm.py:425-446).  f32, as the reference trains.
//
// LayerNorm (rows, C):   dx = rstd (g - mean_C(g) - xhat mean_C(g xhat)),  g = dy gamma;   dgamma = sum_rows dy xhat;  dbeta = sum_rows dy
//   one wave per row, the row of x and dy in registers, the row's statistics RECOMPUTED (two passes over the registers, like the forward):
//   nothing but x is kept from the forward.  dgamma / dbeta: per-lane register accumulators over a wave's rows, the four waves of a
//   workgroup added through LDS, one partial row per workgroup, a second launch adds the partial rows in index order - no atomics, the
//   result is bit-identical from run to run.  Library: 3 launches (grad_input, two-stage gamma/beta) with saved mean / rstd.
// GroupNorm + activation, NCHW (B, C, hw), groups of C/G channels, moments (sum, sum of squares per (sample, group), f64) from the forward:
//   y = xhat gamma + beta;  g = dy act'(y);  dx = rstd (gamma g - mean_grp(gamma g) - xhat mean_grp(gamma g xhat))
//   k_gn_bwd_rows:  per (b, c) row: s1 = sum g, s2 = sum g xhat                              (reads x, dy)
//   k_gn_bwd_group: per (b, group): the two group means;  k_gn_bwd_affine: dgamma_c = sum_b s2, dbeta_c = sum_b s1 (fixed order)
//   k_gn_bwd_apply: dx                                                                        (reads x, dy, writes dx)
//   The activation's backward rides along (y is recomputed from x): the library chain is GroupNorm backward (5 launches) + activation
//   backward (1-2) + their forward counterparts keeping y AND act(y) alive.
// Bound: HBM streaming (2 reads + 1 write of the tensor for dx, 2 reads for the sums).
#include "common.h"

namespace xm3d {

template <int VPL>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma, int64_t rows,
                                                int C, float eps, float* __restrict__ dx, float* __restrict__ part) {
    extern __shared__ float ln_red[];  // [4 waves][2][C], only with part
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C / 4;
    float ag[VPL][4], ab[VPL][4];
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ag[i][j] = ab[i][j] = 0.f;
    const float invC = 1.f / float(C);
    for (int64_t row = int64_t(blockIdx.x) * 4 + wave; row < rows; row += int64_t(gridDim.x) * 4) {
        float v[VPL][4], d[VPL][4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int e = lane + 64 * i;
            if (e < nv) {
                const float4 t = *reinterpret_cast<const float4*>(x + row * C + e * 4);
                const float4 u = *reinterpret_cast<const float4*>(dy + row * C + e * 4);
                v[i][0] = t.x, v[i][1] = t.y, v[i][2] = t.z, v[i][3] = t.w;
                d[i][0] = u.x, d[i][1] = u.y, d[i][2] = u.z, d[i][3] = u.w;
                s += (t.x + t.y) + (t.z + t.w);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s * invC;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            if (lane + 64 * i < nv)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float c = v[i][j] - mean;
                    ss = fmaf(c, c, ss);
                }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float rstd = rsqrtf(ss * invC + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int e = lane + 64 * i;
            if (e < nv) {
                const float4 g4 = gamma ? *reinterpret_cast<const float4*>(gamma + e * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (v[i][j] - mean) * rstd;
                    ag[i][j] = fmaf(d[i][j], xh, ag[i][j]);
                    ab[i][j] += d[i][j];
                    v[i][j] = xh;          // x is no longer needed: keep xhat
                    d[i][j] *= g[j];       // and g = dy gamma
                    s1 += d[i][j];
                    s2 = fmaf(d[i][j], xh, s2);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
        s1 *= invC, s2 *= invC;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int e = lane + 64 * i;
            if (e < nv)
                *reinterpret_cast<float4*>(dx + row * C + e * 4) =
                    make_float4(rstd * (d[i][0] - s1 - v[i][0] * s2), rstd * (d[i][1] - s1 - v[i][1] * s2), rstd * (d[i][2] - s1 - v[i][2] * s2),
                                rstd * (d[i][3] - s1 - v[i][3] * s2));
        }
    }
    if (part) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int e = lane + 64 * i;
            if (e < nv) {
                *reinterpret_cast<float4*>(ln_red + (wave * 2 + 0) * C + e * 4) = make_float4(ag[i][0], ag[i][1], ag[i][2], ag[i][3]);
                *reinterpret_cast<float4*>(ln_red + (wave * 2 + 1) * C + e * 4) = make_float4(ab[i][0], ab[i][1], ab[i][2], ab[i][3]);
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < 2 * C; j += 256) {
            const int which = j / C, c = j - which * C;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += ln_red[(w * 2 + which) * C + c];
            part[int64_t(blockIdx.x) * 2 * C + j] = t;
        }
    }
}

// out[j] = sum over p (in order) of part[p][j], j < n: the fixed-order second stage of the parameter gradients
__global__ void k_colsum(const float* __restrict__ part, int n_part, int n, float* __restrict__ o0, float* __restrict__ o1, int split) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float t = 0.f;
    for (int p = 0; p < n_part; ++p) t += part[int64_t(p) * n + j];
    if (j < split) {
        if (o0) o0[j] = t;
    } else if (o1) {
        o1[j - split] = t;
    }
}

__device__ __forceinline__ float gn_act_grad(float y, int act) {
    if (act == 1) {  // SiLU: d/dy (y sigma(y)) = sigma (1 + y (1 - sigma))
        const float sg = 1.f / (1.f + __expf(-y));
        return sg * (1.f + y * (1.f - sg));
    }
    if (act == 2) return y > 0.f ? 1.f : 0.f;
    return 1.f;
}

// one workgroup per (b, c) row of hw values: rowsum[(b C + c) 2 + {0, 1}] = sum g, sum g xhat
__global__ __launch_bounds__(256) void k_gn_bwd_rows(const float* __restrict__ x, const float* __restrict__ dy, const double* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, int C, int hw, int G, float eps, int act,
                                                     float* __restrict__ rowsum) {
    __shared__ float red[2][4];
    const int bc = blockIdx.x, b = bc / C, c = bc - b * C;
    const int cpg = C / G, g = c / cpg;
    const double n = double(cpg) * hw;
    const double m = stats[(b * G + g) * 2] / n;
    const float mean = float(m), rstd = rsqrtf(float(fmax(stats[(b * G + g) * 2 + 1] / n - m * m, 0.0)) + eps);
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float* xr = x + int64_t(bc) * hw;
    const float* dr = dy + int64_t(bc) * hw;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x * 4; i < hw; i += 1024) {
        const float4 t = *reinterpret_cast<const float4*>(xr + i);
        const float4 u = *reinterpret_cast<const float4*>(dr + i);
        const float xv[4] = {t.x, t.y, t.z, t.w}, dv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xv[j] - mean) * rstd;
            const float gg = dv[j] * gn_act_grad(fmaf(xh, ga, be), act);
            s1 += gg;
            s2 = fmaf(gg, xh, s2);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[0][wave] = s1, red[1][wave] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
        rowsum[int64_t(bc) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        rowsum[int64_t(bc) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// one thread per (b, group): coef[(b G + g) 2 + {0, 1}] = mean over the group of gamma g, of gamma g xhat (channels added in order)
__global__ void k_gn_bwd_group(const float* __restrict__ rowsum, const float* __restrict__ gamma, int B, int C, int hw, int G, float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G, cpg = C / G;
    float a = 0.f, q = 0.f;
    for (int k = 0; k < cpg; ++k) {
        const int c = g * cpg + k;
        const float ga = gamma ? gamma[c] : 1.f;
        a = fmaf(ga, rowsum[(int64_t(b) * C + c) * 2], a);
        q = fmaf(ga, rowsum[(int64_t(b) * C + c) * 2 + 1], q);
    }
    const float inv = 1.f / (float(cpg) * float(hw));
    coef[i * 2] = a * inv;
    coef[i * 2 + 1] = q * inv;
}

// one thread per channel: dgamma_c = sum_b rowsum[b, c, 1], dbeta_c = sum_b rowsum[b, c, 0] (samples added in order)
__global__ void k_gn_bwd_affine(const float* __restrict__ rowsum, int B, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, q = 0.f;
    for (int b = 0; b < B; ++b) {
        a += rowsum[(int64_t(b) * C + c) * 2];
        q += rowsum[(int64_t(b) * C + c) * 2 + 1];
    }
    if (dbeta) dbeta[c] = a;
    if (dgamma) dgamma[c] = q;
}

__global__ __launch_bounds__(256) void k_gn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy, const double* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ coef, int C, int hw,
                                                      int G, float eps, int act, float* __restrict__ dx) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const int cpg = C / G, g = c / cpg;
    const double n = double(cpg) * hw;
    const double m = stats[(b * G + g) * 2] / n;
    const float mean = float(m), rstd = rsqrtf(float(fmax(stats[(b * G + g) * 2 + 1] / n - m * m, 0.0)) + eps);
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float ca = coef[(b * G + g) * 2], cq = coef[(b * G + g) * 2 + 1];
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= hw) return;
    const int64_t o = int64_t(bc) * hw + i;
    const float4 t = *reinterpret_cast<const float4*>(x + o);
    const float4 u = *reinterpret_cast<const float4*>(dy + o);
    const float xv[4] = {t.x, t.y, t.z, t.w}, dv[4] = {u.x, u.y, u.z, u.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float xh = (xv[j] - mean) * rstd;
        const float gg = dv[j] * gn_act_grad(fmaf(xh, ga, be), act);
        r[j] = rstd * (ga * gg - ca - xh * cq);
    }
    *reinterpret_cast<float4*>(dx + o) = make_float4(r[0], r[1], r[2], r[3]);
}

}  // namespace xm3d

using namespace xm3d;

static int ln_bwd_blocks(int64_t rows) { return int(rows < 4 * 256 ? (rows + 3) / 4 : 256); }

/* floats of workspace xm3d_layer_norm_bwd needs when dgamma or dbeta is asked for */
extern "C" int64_t xm3d_layer_norm_bwd_ws_floats(int64_t rows, int32_t C) { return int64_t(ln_bwd_blocks(rows)) * 2 * C; }

extern "C" int xm3d_layer_norm_bwd(const float* x, const float* dy, const float* gamma, int64_t rows, int32_t C, float eps, float* dx, float* dgamma, float* dbeta,
                                   float* ws, void* stream) {
    XM3D_REQUIRE(rows >= 0 && C >= 4 && C % 4 == 0 && C <= 2048, "layer_norm_bwd: rows=%lld, C=%d (multiple of 4, <= 2048) expected", (long long)rows, C);
    if (rows == 0) return XM3D_OK;
    XM3D_REQUIRE(x && dy && dx, "layer_norm_bwd: null pointer");
    const bool affine = dgamma || dbeta;
    XM3D_REQUIRE(!affine || (ws && C <= 1024), "layer_norm_bwd: parameter gradients need the workspace (xm3d_layer_norm_bwd_ws_floats) and C <= 1024");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(gamma) |
                   reinterpret_cast<uintptr_t>(ws)) & 15) == 0, "layer_norm_bwd: tensors must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    const int blocks = ln_bwd_blocks(rows);
    const size_t lds = affine ? size_t(4) * 2 * C * sizeof(float) : 0;
    if (C <= 1024) hipLaunchKernelGGL(k_ln_bwd<4>, dim3(blocks), dim3(256), lds, s, x, dy, gamma, rows, C, eps, dx, affine ? ws : nullptr);
    else hipLaunchKernelGGL(k_ln_bwd<8>, dim3(blocks), dim3(256), lds, s, x, dy, gamma, rows, C, eps, dx, affine ? ws : nullptr);
    XM3D_LAUNCH_CHECK();
    if (affine) {
        hipLaunchKernelGGL(k_colsum, dim3((2 * C + 255) / 256), dim3(256), 0, s, ws, blocks, 2 * C, dgamma, dbeta, C);
        XM3D_LAUNCH_CHECK();
    }
    return XM3D_OK;
}

/* floats of workspace of xm3d_group_norm_bwd: the (B, C, 2) row sums and the (B, G, 2) group coefficients */
extern "C" int64_t xm3d_group_norm_bwd_ws_floats(int64_t B, int32_t C, int32_t G) { return B * C * 2 + B * G * 2; }

/* x, dy, dx: (B, C, hw) contiguous f32 (NCHW), hw % 4 == 0; stats: the forward's moments (xm3d_group_norm's stats_ws: sum and sum of squares
 * per (sample, group), f64); gamma / beta f32 (C) or null; act 0 none / 1 SiLU / 2 ReLU = what xm3d_group_norm applied behind the affine.
 * dgamma / dbeta (C) may be null (frozen norms: the UNet's). */
extern "C" int xm3d_group_norm_bwd(const float* x, const float* dy, const double* stats, const float* gamma, const float* beta, int64_t B, int32_t C, int32_t hw,
                                   int32_t G, float eps, int32_t act, float* dx, float* dgamma, float* dbeta, float* ws, void* stream) {
    XM3D_REQUIRE(B >= 0 && C > 0 && G > 0 && C % G == 0 && hw > 0 && hw % 4 == 0, "group_norm_bwd: B=%lld C=%d hw=%d G=%d (hw %% 4 == 0, C %% G == 0)",
                 (long long)B, C, hw, G);
    XM3D_REQUIRE(act >= 0 && act <= 2, "group_norm_bwd: act must be 0 (none), 1 (SiLU) or 2 (ReLU)");
    if (B == 0) return XM3D_OK;
    XM3D_REQUIRE(x && dy && stats && dx && ws, "group_norm_bwd: null pointer");
    XM3D_REQUIRE(B * C <= 65535, "group_norm_bwd: B * C = %lld exceeds the grid's y range", (long long)(B * C));
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
                 "group_norm_bwd: tensors must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    float* rowsum = ws;
    float* coef = ws + B * C * 2;
    hipLaunchKernelGGL(k_gn_bwd_rows, dim3(unsigned(B * C)), dim3(256), 0, s, x, dy, stats, gamma, beta, C, hw, G, eps, act, rowsum);
    XM3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gn_bwd_group, dim3(unsigned((B * G + 63) / 64)), dim3(64), 0, s, rowsum, gamma, int(B), C, hw, G, coef);
    XM3D_LAUNCH_CHECK();
    if (dgamma || dbeta) {
        hipLaunchKernelGGL(k_gn_bwd_affine, dim3((C + 63) / 64), dim3(64), 0, s, rowsum, int(B), C, dgamma, dbeta);
        XM3D_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_gn_bwd_apply, dim3((hw / 4 + 255) / 256, unsigned(B * C)), dim3(256), 0, s, x, dy, stats, gamma, beta, coef, C, hw, G, eps, act, dx);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

namespace xm3d {
// per-workgroup partial column sums of a (rows, n) f32 matrix: workgroup w adds rows w, w + gridDim.x, ... (fixed order) for 256 columns
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ x, int64_t rows, int n, int64_t ld, float* __restrict__ part) {
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= n) return;
    float t = 0.f;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) t += x[r * ld + j];
    part[int64_t(blockIdx.x) * n + j] = t;
}
}  // namespace xm3d

static int colsum_parts(int64_t rows) { return int(rows < 256 ? (rows < 1 ? 1 : rows) : 256); }

/* floats of workspace of xm3d_column_sum */
extern "C" int64_t xm3d_column_sum_ws_floats(int64_t rows, int32_t n) { return int64_t(colsum_parts(rows)) * n; }

/* out[j] = sum over rows of x[r, j], x (rows, n) f32 with row stride ld: the bias gradient of a linear layer.  Two launches, every sum in a
 * fixed order: no atomics, no semaphores - bit-reproducible, and safe to replay from a HIP graph (torch's multi-block column reduction is not
 * on this stack: tools/graph_reduce_probe.py). */
extern "C" int xm3d_column_sum(const float* x, int64_t rows, int32_t n, int64_t ld, float* out, float* ws, void* stream) {
    XM3D_REQUIRE(rows >= 0 && n > 0 && ld >= n, "column_sum: rows=%lld n=%d ld=%lld", (long long)rows, n, (long long)ld);
    XM3D_REQUIRE(x && out && ws, "column_sum: null pointer");
    hipStream_t s = as_stream(stream);
    const int parts = colsum_parts(rows);
    hipLaunchKernelGGL(k_colsum_part, dim3(parts, (n + 255) / 256), dim3(256), 0, s, x, rows, n, ld, ws);
    XM3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_colsum, dim3((n + 255) / 256), dim3(256), 0, s, ws, parts, n, out, static_cast<float*>(nullptr), n);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
