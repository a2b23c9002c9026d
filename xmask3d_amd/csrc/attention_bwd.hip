// Softmax attention BACKWARD for gfx950: dQ, dK, dV of  O = softmax(Q K^T * scale + bias) V , bf16 in / f32 accumulate / bf16 out.
//
// Training reaches it through the frozen Stable-Diffusion UNet: the gradient of the losses w.r.t. the 3D conditioning passes through
// every self- and cross-attention of the UNet (reference: loss.backward() in run/train.py:537 through ldm's CrossAttention, reached from
// models/modeling/meta_arch/ldm.py:425-446,670-676).  torch's fused backward on ROCm is AOTriton (Triton-built code, excluded from
// this path); the unfused one materialises the (heads, Nq, Nk) score matrix three times.  Flash-style: P is recomputed per tile from
// Q, K and the forward's log-sum-exp (xm3d_attention_fwd_lse), nothing of size Nq x Nk touches memory.
//
// Two kernels with the forward kernel's structure (4 waves x 32 rows per workgroup, 64-row tiles of the other side staged in LDS,
// register-staged double buffering, one barrier pair per tile), no atomics, bit-reproducible:
//   k_attn_bwd_dq : a wave owns 32 QUERIES (query on the lane).  Per key tile: S^T = K Q^T and dP^T = V dO^T (A operand = rows of the K / V
//                   tile, B operand = the wave's Q / dO fragments in registers), P^T = exp2(S^T - lse), dS^T = P^T (dP^T - delta) scale in
//                   the accumulator registers, which are - converted to bf16 - directly the B operand of dQ^T += K^T dS^T; K^T comes
//                   from transposing LDS reads (ds_read_b64_tr_b16) of the row-major K tile.  Also writes delta = rowsum(dO . O).
//   k_attn_bwd_dkv: a wave owns 32 KEYS (key on the lane).  Per query tile: S = Q K^T and dP = dO V^T (A = rows of the Q / dO tile, B = the
//                   wave's K / V fragments), P and dS in registers with the per-query lse / delta read from LDS, then
//                   dV^T += dO^T P and dK^T += Q^T dS with dO^T / Q^T from transposing reads of the same tiles.
// Recomputing S in both kernels costs 7 MFMA products instead of the minimal 5, and buys: no cross-workgroup sums at all.
#include "common.h"

namespace xm3d {

typedef float ab_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ab_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ab_bf16x4 __attribute__((ext_vector_type(4)));

constexpr int AB_NW = 4;     // waves per workgroup
constexpr int AB_T = 64;     // rows of the streamed side per tile
constexpr float AB_NEG = -1e30f;
constexpr float AB_LOG2E = 1.4426950408889634f;

__device__ __forceinline__ unsigned ab_lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>(reinterpret_cast<const __attribute__((address_space(3))) char*>(
        reinterpret_cast<uintptr_t>(p))));
}
// transposing reads of a row-major bf16 tile (see attention.hip): issue / one wait for everything issued
__device__ __forceinline__ void ab_tr_issue(uint2& lo, uint2& hi, unsigned addr, unsigned addr2) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(addr), "v"(addr2) : "memory");
}
template <int N>
__device__ __forceinline__ void ab_tr_wait(uint2 (&lo)[N], uint2 (&hi)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(lo[i]), "+v"(hi[i]));
}

struct AttnBwdArgs {
    const __bf16 *q, *k, *v, *o, *dout;
    __bf16 *dq, *dk, *dv;       // contiguous (B, N, H, D)
    const float* lse2;          // (B, H, Nq) log2-domain log-sum-exp of the forward
    float* delta;               // (B, H, Nq) rowsum(dO . O): written by the dq kernel, read by the dkv kernel
    const void* bias;
    int Nq, Nk, D, H;
    int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, g_sb, g_sn, g_sh, b_sb, b_sh, b_sq;
    float scale, scale_log2e;
};

// 8-channel fragment (16 bytes) of row `row` of a strided (rows, D) bf16 matrix, zero outside
__device__ __forceinline__ ab_bf16x8 ab_row_frag(const __bf16* base, int64_t row_stride, int row, int nrows, int c, int D) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nrows && c < D) v = *reinterpret_cast<const uint4*>(base + int64_t(row) * row_stride + c);
    return __builtin_bit_cast(ab_bf16x8, v);
}

template <int BIAS>
__device__ __forceinline__ float ab_score(float s, const AttnBwdArgs& a, int b, int head, int q, int key) {
    float sc = s * a.scale_log2e;
    if (BIAS != 0) {
        if (q < a.Nq && key < a.Nk) {
            const int64_t bo = b * a.b_sb + head * a.b_sh + int64_t(q) * a.b_sq + key;
            const float bv = BIAS == 1 ? static_cast<const float*>(a.bias)[bo] : float(static_cast<const __bf16*>(a.bias)[bo]);
            sc = bv < -1e29f ? AB_NEG : sc + bv * AB_LOG2E;
        }
    }
    return key < a.Nk ? sc : AB_NEG;
}

// ------------------------------------------------------------------------------------------------------------------ dQ (+ delta)
template <int DQ, int DV, int BIAS>
__global__ __launch_bounds__(64 * AB_NW) void k_attn_bwd_dq(const AttnBwdArgs a) {
    constexpr int SQ = DQ / 16, TV = DV / 32, LD = DV + 8;
    __shared__ __attribute__((aligned(16))) __bf16 lk[2][AB_T][LD];
    __shared__ __attribute__((aligned(16))) __bf16 lv[2][AB_T][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, q0 = (blockIdx.x * AB_NW + wave) * 32, q = q0 + l31;
    const int D = a.D, Nq = a.Nq, Nk = a.Nk, dchunks = D / 8;
    const __bf16* Kb = a.k + b * a.k_sb + head * a.k_sh;
    const __bf16* Vb = a.v + b * a.v_sb + head * a.v_sh;

    for (int i = tid; i < int(sizeof(lk) / 16); i += 64 * AB_NW) reinterpret_cast<uint4*>(&lk[0][0][0])[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < int(sizeof(lv) / 16); i += 64 * AB_NW) reinterpret_cast<uint4*>(&lv[0][0][0])[i] = make_uint4(0, 0, 0, 0);

    // this lane's query row: Q, dO fragments (B operands) and delta = sum_d dO O (both lane halves together cover all channels)
    ab_bf16x8 qf[SQ], dof[SQ];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < SQ; ++s) {
        const int c = 16 * s + 8 * h;
        qf[s] = ab_row_frag(a.q + b * a.q_sb + head * a.q_sh, a.q_sn, q, Nq, c, D);
        dof[s] = ab_row_frag(a.dout + b * a.g_sb + head * a.g_sh, a.g_sn, q, Nq, c, D);
        const ab_bf16x8 of = ab_row_frag(a.o + b * a.o_sb + head * a.o_sh, a.o_sn, q, Nq, c, D);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl = fmaf(float(dof[s][j]), float(of[j]), dl);
    }
    dl += __shfl_xor(dl, 32);
    const int64_t rowi = (int64_t(b) * a.H + head) * Nq + q;
    const float lse = q < Nq ? a.lse2[rowi] : 1e30f;
    if (q < Nq && h == 0) a.delta[rowi] = dl;

    constexpr int CH = (AB_T * (DV / 8) + 64 * AB_NW - 1) / (64 * AB_NW);
    uint4 kst[CH], vst[CH];
    auto tile_load = [&](int t) __attribute__((always_inline)) {
        const int k0 = t * AB_T;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = i * 64 * AB_NW + tid, row = c / dchunks, col = c - row * dchunks;
            kst[i] = vst[i] = make_uint4(0, 0, 0, 0);
            if (row < AB_T && k0 + row < Nk) {
                kst[i] = *reinterpret_cast<const uint4*>(Kb + int64_t(k0 + row) * a.k_sn + col * 8);
                vst[i] = *reinterpret_cast<const uint4*>(Vb + int64_t(k0 + row) * a.v_sn + col * 8);
            }
        }
    };
    auto tile_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = i * 64 * AB_NW + tid, row = c / dchunks, col = c - row * dchunks;
            if (row < AB_T) {
                *reinterpret_cast<uint4*>(&lk[buf][row][col * 8]) = kst[i];
                *reinterpret_cast<uint4*>(&lv[buf][row][col * 8]) = vst[i];
            }
        }
    };

    ab_f32x16 dq[TV];
#pragma unroll
    for (int t = 0; t < TV; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

    const int ntiles = (Nk + AB_T - 1) / AB_T;
    tile_load(0);
    __syncthreads();
    tile_store(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) tile_load(t + 1);
        ab_bf16x8 dsf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            ab_f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = dpacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < SQ; ++s) {
                const ab_bf16x8 kf = *reinterpret_cast<const ab_bf16x8*>(&lk[buf][32 * kb + l31][16 * s + 8 * h]);
                const ab_bf16x8 vf = *reinterpret_cast<const ab_bf16x8*>(&lv[buf][32 * kb + l31][16 * s + 8 * h]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dpacc, 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * s2 + j;
                    const int key = t * AB_T + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float sc = ab_score<BIAS>(sacc[r], a, b, head, q, key);
                    const float p = sc <= AB_NEG ? 0.f : __builtin_amdgcn_exp2f(sc - lse);
                    dsf[kb][s2][j] = (__bf16)(p * (dpacc[r] - dl) * a.scale);
                }
        }
        // dQ^T += K^T dS^T: A = K^T through transposing reads of the row-major K tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint2 lo[2 * TV], hi[2 * TV];
            const int li = lane & 15;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int krow = 32 * kb + 16 * s2 + 4 * h;
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    const unsigned addr = ab_lds_addr(&lk[buf][krow + (li >> 2)][32 * tv + 16 * ((lane >> 4) & 1) + 4 * (li & 3)]);
                    ab_tr_issue(lo[s2 * TV + tv], hi[s2 * TV + tv], addr, addr + 8 * LD * 2);
                }
            }
            ab_tr_wait(lo, hi);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    const uint4 r = make_uint4(lo[s2 * TV + tv].x, lo[s2 * TV + tv].y, hi[s2 * TV + tv].x, hi[s2 * TV + tv].y);
                    dq[tv] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ab_bf16x8, r), dsf[kb][s2], dq[tv], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles) tile_store(buf ^ 1);
        __syncthreads();
    }
    if (q < Nq) {
        __bf16* out = a.dq + ((int64_t(b) * Nq + q) * a.H + head) * D;
#pragma unroll
        for (int tv = 0; tv < TV; ++tv)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * tv + 8 * g + 4 * h;
                if (c < D) {
                    ab_bf16x4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = (__bf16)dq[tv][4 * g + r];
                    *reinterpret_cast<ab_bf16x4*>(out + c) = w;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------ dK, dV
template <int DQ, int DV, int BIAS>
__global__ __launch_bounds__(64 * AB_NW) void k_attn_bwd_dkv(const AttnBwdArgs a) {
    constexpr int SQ = DQ / 16, TV = DV / 32, LD = DV + 8;
    __shared__ __attribute__((aligned(16))) __bf16 lq[2][AB_T][LD];
    __shared__ __attribute__((aligned(16))) __bf16 lg[2][AB_T][LD];   // dO
    __shared__ float lrow[2][2][AB_T];                                 // [buf][lse2 | delta][query]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, k0w = (blockIdx.x * AB_NW + wave) * 32, key = k0w + l31;
    const int D = a.D, Nq = a.Nq, Nk = a.Nk, dchunks = D / 8;
    const __bf16* Qb = a.q + b * a.q_sb + head * a.q_sh;
    const __bf16* Gb = a.dout + b * a.g_sb + head * a.g_sh;
    const float* lse_b = a.lse2 + (int64_t(b) * a.H + head) * Nq;
    const float* dl_b = a.delta + (int64_t(b) * a.H + head) * Nq;

    for (int i = tid; i < int(sizeof(lq) / 16); i += 64 * AB_NW) reinterpret_cast<uint4*>(&lq[0][0][0])[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < int(sizeof(lg) / 16); i += 64 * AB_NW) reinterpret_cast<uint4*>(&lg[0][0][0])[i] = make_uint4(0, 0, 0, 0);

    ab_bf16x8 kf[SQ], vf[SQ];  // this lane's key row: K, V fragments (B operands)
#pragma unroll
    for (int s = 0; s < SQ; ++s) {
        kf[s] = ab_row_frag(a.k + b * a.k_sb + head * a.k_sh, a.k_sn, key, Nk, 16 * s + 8 * h, D);
        vf[s] = ab_row_frag(a.v + b * a.v_sb + head * a.v_sh, a.v_sn, key, Nk, 16 * s + 8 * h, D);
    }

    constexpr int CH = (AB_T * (DV / 8) + 64 * AB_NW - 1) / (64 * AB_NW);
    uint4 qst[CH], gst[CH];
    float rst[2] = {1e30f, 0.f};  // threads 0..63: lse2 / delta of query row tid of the tile
    auto tile_load = [&](int t) __attribute__((always_inline)) {
        const int r0 = t * AB_T;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = i * 64 * AB_NW + tid, row = c / dchunks, col = c - row * dchunks;
            qst[i] = gst[i] = make_uint4(0, 0, 0, 0);
            if (row < AB_T && r0 + row < Nq) {
                qst[i] = *reinterpret_cast<const uint4*>(Qb + int64_t(r0 + row) * a.q_sn + col * 8);
                gst[i] = *reinterpret_cast<const uint4*>(Gb + int64_t(r0 + row) * a.g_sn + col * 8);
            }
        }
        if (tid < AB_T) {
            const bool ok = r0 + tid < Nq;
            rst[0] = ok ? lse_b[r0 + tid] : 1e30f;  // a row past the end: P = exp2(s - 1e30) = 0
            rst[1] = ok ? dl_b[r0 + tid] : 0.f;
        }
    };
    auto tile_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = i * 64 * AB_NW + tid, row = c / dchunks, col = c - row * dchunks;
            if (row < AB_T) {
                *reinterpret_cast<uint4*>(&lq[buf][row][col * 8]) = qst[i];
                *reinterpret_cast<uint4*>(&lg[buf][row][col * 8]) = gst[i];
            }
        }
        if (tid < AB_T) {
            lrow[buf][0][tid] = rst[0];
            lrow[buf][1][tid] = rst[1];
        }
    };

    ab_f32x16 dk[TV], dv[TV];
#pragma unroll
    for (int t = 0; t < TV; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.f;

    const int ntiles = (Nq + AB_T - 1) / AB_T;
    tile_load(0);
    __syncthreads();
    tile_store(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) tile_load(t + 1);
        ab_bf16x8 pf[2][2], dsf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            ab_f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = dpacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < SQ; ++s) {
                const ab_bf16x8 qr = *reinterpret_cast<const ab_bf16x8*>(&lq[buf][32 * qb + l31][16 * s + 8 * h]);
                const ab_bf16x8 gr = *reinterpret_cast<const ab_bf16x8*>(&lg[buf][32 * qb + l31][16 * s + 8 * h]);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr, kf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gr, vf[s], dpacc, 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * s2 + j;
                    const int ql = 32 * qb + (r & 3) + 8 * (r >> 2) + 4 * h;  // query row inside the tile
                    const float sc = ab_score<BIAS>(sacc[r], a, b, head, t * AB_T + ql, key);
                    const float p = sc <= AB_NEG ? 0.f : __builtin_amdgcn_exp2f(sc - lrow[buf][0][ql]);
                    pf[qb][s2][j] = (__bf16)p;
                    dsf[qb][s2][j] = (__bf16)(p * (dpacc[r] - lrow[buf][1][ql]) * a.scale);
                }
        }
        // dV^T += dO^T P and dK^T += Q^T dS: A operands through transposing reads of the dO / Q tiles
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            uint2 glo[2 * TV], ghi[2 * TV], qlo[2 * TV], qhi[2 * TV];
            const int li = lane & 15;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int row = 32 * qb + 16 * s2 + 4 * h;
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    const int col = 32 * tv + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
                    const unsigned ag = ab_lds_addr(&lg[buf][row + (li >> 2)][col]), aq = ab_lds_addr(&lq[buf][row + (li >> 2)][col]);
                    ab_tr_issue(glo[s2 * TV + tv], ghi[s2 * TV + tv], ag, ag + 8 * LD * 2);
                    ab_tr_issue(qlo[s2 * TV + tv], qhi[s2 * TV + tv], aq, aq + 8 * LD * 2);
                }
            }
            ab_tr_wait(glo, ghi);
            ab_tr_wait(qlo, qhi);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    const int i = s2 * TV + tv;
                    const uint4 rg = make_uint4(glo[i].x, glo[i].y, ghi[i].x, ghi[i].y), rq = make_uint4(qlo[i].x, qlo[i].y, qhi[i].x, qhi[i].y);
                    dv[tv] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ab_bf16x8, rg), pf[qb][s2], dv[tv], 0, 0, 0);
                    dk[tv] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ab_bf16x8, rq), dsf[qb][s2], dk[tv], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles) tile_store(buf ^ 1);
        __syncthreads();
    }
    if (key < Nk) {
        __bf16* ok_ = a.dk + ((int64_t(b) * Nk + key) * a.H + head) * D;
        __bf16* ov_ = a.dv + ((int64_t(b) * Nk + key) * a.H + head) * D;
#pragma unroll
        for (int tv = 0; tv < TV; ++tv)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * tv + 8 * g + 4 * h;
                if (c < D) {
                    ab_bf16x4 wk, wv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        wk[r] = (__bf16)dk[tv][4 * g + r];
                        wv[r] = (__bf16)dv[tv][4 * g + r];
                    }
                    *reinterpret_cast<ab_bf16x4*>(ok_ + c) = wk;
                    *reinterpret_cast<ab_bf16x4*>(ov_ + c) = wv;
                }
            }
    }
}

template <int DQ, int DV, int BIAS>
static void attn_bwd_launch(const AttnBwdArgs& a, int B, hipStream_t s) {
    hipLaunchKernelGGL((k_attn_bwd_dq<DQ, DV, BIAS>), dim3((a.Nq + 32 * AB_NW - 1) / (32 * AB_NW), a.H, B), dim3(64 * AB_NW), 0, s, a);
    hipLaunchKernelGGL((k_attn_bwd_dkv<DQ, DV, BIAS>), dim3((a.Nk + 32 * AB_NW - 1) / (32 * AB_NW), a.H, B), dim3(64 * AB_NW), 0, s, a);
}

template <int DQ, int DV>
static void attn_bwd_bias(const AttnBwdArgs& a, int B, int bias_dtype, hipStream_t s) {
    if (bias_dtype == 0) attn_bwd_launch<DQ, DV, 0>(a, B, s);
    else if (bias_dtype == 1) attn_bwd_launch<DQ, DV, 1>(a, B, s);
    else attn_bwd_launch<DQ, DV, 2>(a, B, s);
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_attention_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse2,
                                  int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t D, const int64_t* q_strides, const int64_t* k_strides,
                                  const int64_t* v_strides, const int64_t* o_strides, const int64_t* do_strides, const void* bias,
                                  int32_t bias_dtype, const int64_t* bias_strides, float scale, void* dq, void* dk, void* dv, float* delta_ws,
                                  void* stream) {
    XM3D_REQUIRE(B >= 0 && H >= 1 && Nq >= 0 && Nk >= 1 && D >= 8, "attention_bwd: bad sizes B=%d H=%d Nq=%d Nk=%d D=%d", B, H, Nq, Nk, D);
    XM3D_REQUIRE(D % 8 == 0 && D <= 160, "attention_bwd: head channels must be a multiple of 8 and <= 160 (got %d)", D);
    if (B == 0 || Nq == 0) return XM3D_OK;
    XM3D_REQUIRE(q && k && v && out && dout && lse2 && dq && dk && dv && delta_ws && q_strides && k_strides && v_strides && o_strides && do_strides,
                 "attention_bwd: null pointer");
    XM3D_REQUIRE(bias_dtype >= 0 && bias_dtype <= 2 && (bias_dtype == 0 || (bias && bias_strides)), "attention_bwd: bad bias arguments");
    for (const int64_t* s : {q_strides, k_strides, v_strides, o_strides, do_strides})
        XM3D_REQUIRE(s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0, "attention_bwd: strides must be multiples of 8 elements (16-byte rows)");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(dout)) & 15) == 0 &&
                     ((reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) & 7) == 0,
                 "attention_bwd: inputs must be 16-byte aligned, gradients 8-byte aligned");
    AttnBwdArgs a;
    a.q = static_cast<const __bf16*>(q), a.k = static_cast<const __bf16*>(k), a.v = static_cast<const __bf16*>(v);
    a.o = static_cast<const __bf16*>(out), a.dout = static_cast<const __bf16*>(dout);
    a.dq = static_cast<__bf16*>(dq), a.dk = static_cast<__bf16*>(dk), a.dv = static_cast<__bf16*>(dv);
    a.lse2 = lse2, a.delta = delta_ws, a.bias = bias;
    a.Nq = Nq, a.Nk = Nk, a.D = D, a.H = H;
    a.q_sb = q_strides[0], a.q_sn = q_strides[1], a.q_sh = q_strides[2];
    a.k_sb = k_strides[0], a.k_sn = k_strides[1], a.k_sh = k_strides[2];
    a.v_sb = v_strides[0], a.v_sn = v_strides[1], a.v_sh = v_strides[2];
    a.o_sb = o_strides[0], a.o_sn = o_strides[1], a.o_sh = o_strides[2];
    a.g_sb = do_strides[0], a.g_sn = do_strides[1], a.g_sh = do_strides[2];
    a.b_sb = bias_dtype ? bias_strides[0] : 0, a.b_sh = bias_dtype ? bias_strides[1] : 0, a.b_sq = bias_dtype ? bias_strides[2] : 0;
    a.scale = scale, a.scale_log2e = scale * AB_LOG2E;
    hipStream_t s = as_stream(stream);
    if (D <= 32) attn_bwd_bias<32, 32>(a, B, bias_dtype, s);
    else if (D <= 48) attn_bwd_bias<48, 64>(a, B, bias_dtype, s);
    else if (D <= 64) attn_bwd_bias<64, 64>(a, B, bias_dtype, s);
    else if (D <= 80) attn_bwd_bias<80, 96>(a, B, bias_dtype, s);
    else if (D <= 96) attn_bwd_bias<96, 96>(a, B, bias_dtype, s);
    else if (D <= 128) attn_bwd_bias<128, 128>(a, B, bias_dtype, s);
    else attn_bwd_bias<160, 160>(a, B, bias_dtype, s);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
