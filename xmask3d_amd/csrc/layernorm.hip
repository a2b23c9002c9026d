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

// ---- residual add + LayerNorm over an f32 residual stream whose readers are bf16 GEMMs (the pixel decoder's deformable-attention encoder
// and the masked-attention transformer decoder under bf16 inference: /root/reference/models/modeling/pixel_decoder/msdeformattn.py:35-60,
// .../transformer_decoder/mask2former_transformer_decoder.py:17-178):
//     y = LayerNorm(x + delta) * gamma + beta         x f32 (the stream), delta f32 / bf16 / none, statistics and y in f32
// written as any of: y (f32, the new stream), bf16(y) (what the next linear layer's input cast would produce), bf16(y + pos) (the
// query / key input "with positional embedding"; pos f32 or bf16, row r of the stream reads pos row r % pos_rows: a batch-broadcast
// embedding).  One launch instead of cast + add + LayerNorm + add + two casts; the arithmetic per element is the chain's (f32 add, f32
// LayerNorm, f32 add, one rounding to bf16).  One wave per row, C <= 1024, 16-byte accesses.
template <typename TO>
__device__ __forceinline__ void aln_store(TO* dst, const float (&o)[4]) {
    if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        __hip_bfloat16 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = __float2bfloat16(o[j]);
        *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(q);
    }
}

// TO: type of the second and third output (bf16: the bf16 configuration's GEMM inputs; float: the fp32 configuration, where only `stream + pos`
// is a new tensor and y_bf is not asked for)
template <typename TD, typename TP, typename TO>
__global__ __launch_bounds__(256) void k_add_layer_norm(const float* __restrict__ x, const TD* __restrict__ delta, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t rows, int C, float eps, const TP* __restrict__ pos,
                                                        int64_t pos_rows, float* __restrict__ y, TO* __restrict__ y_bf, TO* __restrict__ ypos_bf) {
    constexpr int VPL = 4;  // float4 vectors per lane: C <= 64 * 4 * 4
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C / 4;
    float v[VPL][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int e = lane + 64 * i;
        if (e < nv) {
            const float4 t = *reinterpret_cast<const float4*>(x + row * C + e * 4);
            v[i][0] = t.x, v[i][1] = t.y, v[i][2] = t.z, v[i][3] = t.w;
            if (delta) {
                if constexpr (sizeof(TD) == 4) {
                    const float4 d = *reinterpret_cast<const float4*>(delta + row * C + e * 4);
                    v[i][0] += d.x, v[i][1] += d.y, v[i][2] += d.z, v[i][3] += d.w;
                } else {
                    const uint2 d = *reinterpret_cast<const uint2*>(delta + row * C + e * 4);
                    v[i][0] += __uint_as_float(d.x << 16), v[i][1] += __uint_as_float(d.x & 0xFFFF0000u);
                    v[i][2] += __uint_as_float(d.y << 16), v[i][3] += __uint_as_float(d.y & 0xFFFF0000u);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s += v[i][j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / float(C);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        if (lane + 64 * i < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                ss = fmaf(d, d, ss);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rstd = rsqrtf(ss / float(C) + eps);
    const int64_t prow = pos ? row % pos_rows : 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int e = lane + 64 * i;
        if (e < nv) {
            const float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + e * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 b = beta ? *reinterpret_cast<const float4*>(beta + e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float o[4] = {fmaf((v[i][0] - mean) * rstd, g.x, b.x), fmaf((v[i][1] - mean) * rstd, g.y, b.y), fmaf((v[i][2] - mean) * rstd, g.z, b.z),
                          fmaf((v[i][3] - mean) * rstd, g.w, b.w)};
            if (y) *reinterpret_cast<float4*>(y + row * C + e * 4) = make_float4(o[0], o[1], o[2], o[3]);
            if (y_bf) aln_store<TO>(y_bf + row * C + e * 4, o);
            if (ypos_bf) {
                float p[4];
                if constexpr (sizeof(TP) == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(pos + prow * C + e * 4);
                    p[0] = t.x, p[1] = t.y, p[2] = t.z, p[3] = t.w;
                } else {
                    const uint2 t = *reinterpret_cast<const uint2*>(pos + prow * C + e * 4);
                    p[0] = __uint_as_float(t.x << 16), p[1] = __uint_as_float(t.x & 0xFFFF0000u);
                    p[2] = __uint_as_float(t.y << 16), p[3] = __uint_as_float(t.y & 0xFFFF0000u);
                }
                const float op[4] = {o[0] + p[0], o[1] + p[1], o[2] + p[2], o[3] + p[3]};
                aln_store<TO>(ypos_bf + row * C + e * 4, op);
            }
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

// delta_dtype / pos_dtype / out_dtype: 0 = f32, 1 = bf16 (ignored when the pointer is null); out_dtype is the type of y_bf and ypos_bf.  Any of
// y / y_bf / ypos_bf may be null (not all); ypos_bf needs pos.
extern "C" int xm3d_add_layer_norm(const float* x, const void* delta, int32_t delta_dtype, int64_t rows, int32_t C, const float* gamma, const float* beta,
                                   float eps, const void* pos, int32_t pos_dtype, int64_t pos_rows, float* y, void* y_bf, void* ypos_bf, int32_t out_dtype,
                                   void* stream) {
    XM3D_REQUIRE(rows >= 0 && C >= 4 && C % 4 == 0 && C <= 1024, "add_layer_norm: rows=%lld, C=%d (multiple of 4, <= 1024) expected", (long long)rows, C);
    if (rows == 0) return XM3D_OK;
    XM3D_REQUIRE(x && (y || y_bf || ypos_bf), "add_layer_norm: null pointer");
    XM3D_REQUIRE((delta_dtype == 0 || delta_dtype == 1) && (pos_dtype == 0 || pos_dtype == 1) && (out_dtype == 0 || out_dtype == 1),
                 "add_layer_norm: dtype codes are 0 (f32) / 1 (bf16)");
    XM3D_REQUIRE(out_dtype == 1 || ((reinterpret_cast<uintptr_t>(y_bf) | reinterpret_cast<uintptr_t>(ypos_bf)) & 15) == 0, "add_layer_norm: f32 outputs must be 16-byte aligned");
    XM3D_REQUIRE(!ypos_bf || (pos && pos_rows > 0), "add_layer_norm: ypos_bf needs pos and pos_rows > 0");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0 &&
                     ((reinterpret_cast<uintptr_t>(delta) | reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(y_bf) | reinterpret_cast<uintptr_t>(ypos_bf)) & 7) == 0,
                 "add_layer_norm: f32 tensors must be 16-byte aligned, bf16 tensors 8-byte aligned");
    hipStream_t s = as_stream(stream);
    const unsigned blocks = unsigned((rows + 3) / 4);
    typedef __hip_bfloat16 bf;
#define XM3D_ALN(TD, TP, TO)                                                                                                                        \
    hipLaunchKernelGGL((k_add_layer_norm<TD, TP, TO>), dim3(blocks), dim3(256), 0, s, x, static_cast<const TD*>(delta), gamma, beta, rows, C, eps, \
                       static_cast<const TP*>(pos), pos_rows, y, static_cast<TO*>(y_bf), static_cast<TO*>(ypos_bf))
#define XM3D_ALN_O(TD, TP)            \
    do {                              \
        if (out_dtype == 0) XM3D_ALN(TD, TP, float); \
        else XM3D_ALN(TD, TP, bf);    \
    } while (0)
    if (delta_dtype == 0) {
        if (pos_dtype == 0) XM3D_ALN_O(float, float);
        else XM3D_ALN_O(float, bf);
    } else {
        if (pos_dtype == 0) XM3D_ALN_O(bf, float);
        else XM3D_ALN_O(bf, bf);
    }
#undef XM3D_ALN_O
#undef XM3D_ALN
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
