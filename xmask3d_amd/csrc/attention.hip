// Fused attention forward  O = softmax(Q K^T * scale + bias) V  for gfx950, bf16 in / f32 softmax + accumulate / bf16 out.
//
// Replaces the library attention (AOTriton `attn_fwd` behind torch's scaled_dot_product_attention) on every softmax
// attention of the dense 2D branch:
//   * SD-v1 UNet self-attention, 4096 / 1024 / 256 / 64 tokens x 8 heads of 40 / 80 / 160 channels, and cross-attention to
//     the 77 context tokens (reference call sites: models/modeling/meta_arch/ldm.py:425-446 -> ldm's CrossAttention)
//   * mask-CLIP ViT-L/14: 307 tokens x 16 heads of 64 channels with the per-image additive mask (clip.py:239-270)
//   * Mask2Former masked cross-attention: 50 queries x 256 / 1024 / 4096 keys x 8 heads of 32 channels with the additive
//     mask of xm3d_attn_mask_bias shared by the heads (mask2former_transformer_decoder.py:17-80, odise.py:395)
// Layout: Q (B, Nq, H, D), K / V (B, Nk, H, D) with arbitrary element strides (batch, row, head; channels contiguous), so the
// (B, N, H*D) projections are consumed in place and O is written as (B, Nq, H*D): no transposes around the call.
//
// Decomposition (one workgroup = NW waves x 32 query rows, all on one (batch, head)):
//   * K / V tiles of 64 keys are staged in LDS once per workgroup (register-staged, next tile's global loads issued before
//     the current tile's MFMAs, written after them: one barrier pair per tile) and shared by the waves
//   * scores are computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16_bf16: the accumulator then has the query on the
//     lane and 16 of a block's 32 keys in its registers, so the row maximum / sum of the online softmax are in-lane
//     reductions plus ONE exchange with lane ^ 32
//   * the f32 accumulator registers 8s..8s+7 of a 32-key block are, converted to bf16, directly the B operand of the
//     second product O^T += V^T P^T for k-step s (same k-slot permutation on both operands); the A operand V^T comes out
//     of the row-major V tile through the hardware transposing LDS read ds_read_b64_tr_b16
//   * exp2 with the scale folded into one fma per score; fully masked rows give zeros
#include <type_traits>

#include "common.h"

namespace xm3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef float f32x2a __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8a __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4a __attribute__((ext_vector_type(4)));

constexpr int ATT_NW = 4;     // waves per workgroup
constexpr int ATT_KV = 64;    // keys per tile
constexpr float ATT_NEG = -1e30f;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>(reinterpret_cast<const __attribute__((address_space(3))) char*>(
        reinterpret_cast<uintptr_t>(p))));
}

// Transposing reads of a row-major bf16 tile: per 16-lane group a block of 4 rows x 16 columns, delivered column-major.  One
// fragment of the A operand (8 k-slot elements) = rows base .. base+3 and base+8 .. base+11.  EXEC must be all ones (it is: no
// divergence around the calls).  Issue (no wait) / collect (one lgkmcnt(0) for everything issued) are separate statements so
// that all the fragments of a 32-key block are in flight together; the destinations are tied to the wait statement.
__device__ __forceinline__ void tr_issue(uint2& lo, uint2& hi, unsigned addr, unsigned addr2) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(addr), "v"(addr2) : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait(uint2 (&lo)[N], uint2 (&hi)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // nothing that reads the fragments may move above the wait
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(lo[i]), "+v"(hi[i]));  // ... nor any compiler copy of them: redefined here
}

// DQ / DV: head channels padded to a multiple of 16 / 32 (the LDS row lengths); D: real channel count.
// BIAS: 0 none, 1 additive f32, 2 additive bf16; bias element strides: batch, head, query row (key contiguous).
template <int DQ, int DV, int BIAS>
__global__ __launch_bounds__(64 * ATT_NW) void k_attn_fwd(
    const __bf16* __restrict__ Q, const __bf16* __restrict__ K, const __bf16* __restrict__ V, __bf16* __restrict__ O, int Nq,
    int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
    int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, const void* __restrict__ bias, int64_t b_sb, int64_t b_sh, int64_t b_sq,
    float scale_log2e, float* __restrict__ lse2) {
    constexpr int SQ = DQ / 16;   // k-steps of the score product
    constexpr int TV = DV / 32;   // 32-channel output tiles
    constexpr int KLD = DQ + 8;   // padded LDS rows (bf16): breaks the power-of-two stride of the fragment reads
    constexpr int VLD = DV + 8;
    __shared__ __attribute__((aligned(16))) __bf16 lk[2][ATT_KV][KLD];
    __shared__ __attribute__((aligned(16))) __bf16 lv[2][ATT_KV][VLD];
    // additive bias: the accumulator layout has the QUERY on the lane and the keys in registers, so reading bias[q][key] at the point
    // of use made every load instruction touch 64 rows (64 cache lines for 128 bytes of payload: mask-CLIP's 307 x 307 attention ran
    // 3.3 x slower with its mask than without).  Each wave instead fetches its 32 x 64 block row by row with the KEY on the lane
    // (one or two lines per instruction) into a wave-private LDS tile and reads it back transposed, four consecutive keys per access.
    using BiasT = std::conditional_t<BIAS == 1, float, __bf16>;
    constexpr int BLD = ATT_KV + (BIAS == 1 ? 4 : 4);  // padded row: the 32 row-strided reads of a half-wave spread over the banks
    __shared__ __attribute__((aligned(16))) BiasT lb[BIAS != 0 ? ATT_NW : 1][BIAS != 0 ? 32 : 1][BIAS != 0 ? BLD : 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = (blockIdx.x * ATT_NW + wave) * 32;
    const __bf16* Qb = Q + b * q_sb + head * q_sh;
    const __bf16* Kb = K + b * k_sb + head * k_sh;
    const __bf16* Vb = V + b * v_sb + head * v_sh;
    const int dchunks = D / 8;  // 16-byte chunks per row of real data

    // zero the LDS tiles once: padding columns must stay finite (0 * NaN would poison the products)
    for (int i = tid; i < int(sizeof(lk) / 16); i += 64 * ATT_NW) reinterpret_cast<uint4*>(&lk[0][0][0])[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < int(sizeof(lv) / 16); i += 64 * ATT_NW) reinterpret_cast<uint4*>(&lv[0][0][0])[i] = make_uint4(0, 0, 0, 0);

    // Q^T fragments (B operand): lane (query l31, half h) holds Q[q][16 s + 8 h .. + 8]
    bf16x8a qf[SQ];
    {
        const int q = q0 + l31;
#pragma unroll
        for (int s = 0; s < SQ; ++s) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int c = 16 * s + 8 * h;
            if (q < Nq && c < D) v = *reinterpret_cast<const uint4*>(Qb + int64_t(q) * q_sn + c);
            qf[s] = __builtin_bit_cast(bf16x8a, v);
        }
    }

    // staging registers: chunks of the K / V tile owned by this thread
    constexpr int KCH = (ATT_KV * (DQ / 8) + 64 * ATT_NW - 1) / (64 * ATT_NW);
    constexpr int VCH = (ATT_KV * (DV / 8) + 64 * ATT_NW - 1) / (64 * ATT_NW);
    uint4 kst[KCH], vst[VCH];
    auto tile_load = [&](int t) __attribute__((always_inline)) {
        const int k0 = t * ATT_KV;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = i * 64 * ATT_NW + tid, row = c / dchunks, col = c - row * dchunks;
            kst[i] = make_uint4(0, 0, 0, 0);
            if (row < ATT_KV && k0 + row < Nk) kst[i] = *reinterpret_cast<const uint4*>(Kb + int64_t(k0 + row) * k_sn + col * 8);
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = i * 64 * ATT_NW + tid, row = c / dchunks, col = c - row * dchunks;
            vst[i] = make_uint4(0, 0, 0, 0);
            if (row < ATT_KV && k0 + row < Nk) vst[i] = *reinterpret_cast<const uint4*>(Vb + int64_t(k0 + row) * v_sn + col * 8);
        }
    };
    auto tile_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = i * 64 * ATT_NW + tid, row = c / dchunks, col = c - row * dchunks;
            if (row < ATT_KV) *reinterpret_cast<uint4*>(&lk[buf][row][col * 8]) = kst[i];
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = i * 64 * ATT_NW + tid, row = c / dchunks, col = c - row * dchunks;
            if (row < ATT_KV) *reinterpret_cast<uint4*>(&lv[buf][row][col * 8]) = vst[i];
        }
    };

    f32x16 o[TV];
#pragma unroll
    for (int t = 0; t < TV; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = ATT_NEG, l_run = 0.f;  // running maximum (log2 domain) and sum of this lane's query (both halves agree)

    const int ntiles = (Nk + ATT_KV - 1) / ATT_KV;
    tile_load(0);
    __syncthreads();  // zero fill done
    tile_store(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) tile_load(t + 1);
        // bias block of this wave and tile, KEY on the lane: requested before the score products, transposed through LDS behind them
        BiasT braw[BIAS != 0 ? 32 : 1];
        if constexpr (BIAS != 0) {
            // row addresses are wave-uniform (scalar registers), the lane adds its key; rows past Nq / keys past Nk are clamped to the
            // last valid one: their values are never used (the key bound below, no store for rows past Nq)
            const int q0s = __builtin_amdgcn_readfirstlane(q0);
            const BiasT* const bp = static_cast<const BiasT*>(bias) + b * b_sb + head * b_sh;
            const int kl = min(t * ATT_KV + lane, Nk - 1);
#pragma unroll
            for (int i = 0; i < 32; ++i) braw[i] = bp[int64_t(min(q0s + i, Nq - 1)) * b_sq + kl];
        }
        // ---- S^T = K Q^T for the two 32-key blocks of the tile
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < SQ; ++s) {
                const bf16x8a kf = *reinterpret_cast<const bf16x8a*>(&lk[buf][32 * kb + l31][16 * s + 8 * h]);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kb], 0, 0, 0);
            }
        }
        // ---- scale, bias, key bound; running maximum (log2 domain).  Unmasked interior tiles (no bias, all 64 keys in range)
        //      take the short path: max on the raw scores, one fma + one exp2 per score
        const int kbase = t * ATT_KV;
        const bool plain = BIAS == 0 && kbase + ATT_KV <= Nk;  // wave-uniform
        if constexpr (BIAS != 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) lb[wave][i][lane] = braw[i];
            __builtin_amdgcn_wave_barrier();  // wave-private tile; the LDS executes one wave's accesses in issue order
        }
        float mx = ATT_NEG;
        float bq4[4] = {0.f, 0.f, 0.f, 0.f};
        if (plain) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            mx *= scale_log2e;  // scale > 0
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float sc = sacc[kb][r] * scale_log2e;
                    if constexpr (BIAS != 0) {  // (rows past Nq / keys past Nk hold clamped copies: the key bound follows)
                        if ((r & 3) == 0) {     // the lane's four consecutive keys of this group: one 8- / 16-byte LDS read
                            const BiasT* const src = &lb[wave][l31][32 * kb + 2 * r + 4 * h];
                            if constexpr (BIAS == 1) {
                                const float4 v4 = *reinterpret_cast<const float4*>(src);
                                bq4[0] = v4.x, bq4[1] = v4.y, bq4[2] = v4.z, bq4[3] = v4.w;
                            } else {
                                const uint2 v2 = *reinterpret_cast<const uint2*>(src);
                                bq4[0] = __uint_as_float(v2.x << 16), bq4[1] = __uint_as_float(v2.x & 0xFFFF0000u);
                                bq4[2] = __uint_as_float(v2.y << 16), bq4[3] = __uint_as_float(v2.y & 0xFFFF0000u);
                            }
                        }
                        const float bv = bq4[r & 3];
                        sc = bv < -1e29f ? ATT_NEG : fmaf(bv, 1.4426950408889634f, sc);
                    }
                    if (key >= Nk) sc = ATT_NEG;
                    sacc[kb][r] = sc;
                    mx = fmaxf(mx, sc);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 1 when nothing changed, 0 on the first tile
        float psum = 0.f;
        bf16x8a pf[2][2];
        if (plain) {
            // score pairs through the packed f32 pipe (v_pk_fma_f32 / v_pk_add_f32: two scores per issue slot) - this loop is
            // VALU-bound for the 40-channel heads (32 scores per lane and tile against 14 MFMAs per wave)
            const f32x2a sc2 = {scale_log2e, scale_log2e}, nm2 = {-m_new, -m_new};
            f32x2a ps2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x2a sv = {sacc[kb][8 * s2 + j], sacc[kb][8 * s2 + j + 1]};
                        const f32x2a e = __builtin_elementwise_fma(sv, sc2, nm2);
                        f32x2a p;
                        p[0] = __builtin_amdgcn_exp2f(e[0]);  // bare v_exp_f32: the argument is <= 0, a denormal result may flush
                        p[1] = __builtin_amdgcn_exp2f(e[1]);
                        ps2 += p;
                        pf[kb][s2][j] = (__bf16)p[0];
                        pf[kb][s2][j + 1] = (__bf16)p[1];
                    }
            psum = ps2[0] + ps2[1];
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float sc = sacc[kb][8 * s2 + j];
                        const float p = sc <= ATT_NEG ? 0.f : __builtin_amdgcn_exp2f(sc - m_new);  // masked scores contribute nothing, also when the row is all masked
                        psum += p;
                        pf[kb][s2][j] = (__bf16)p;
                    }
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__any(alpha != 1.f)) {  // the maximum of some row of this wave moved: rescale (wave-uniform branch)
#pragma unroll
            for (int tv = 0; tv < TV; ++tv)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[tv][r] *= alpha;
        }
        // ---- O^T += V^T P^T: A = V^T through transposing reads of the row-major V tile (all fragments of a 32-key block
        //      requested together, one wait)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint2 vlo[2 * TV], vhi[2 * TV];
            const int li = lane & 15;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int krow = 32 * kb + 16 * s2 + 4 * h;  // first key row of this lane half's 4 + 4 keys
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    // 16-lane group g reads the 4 x 16 block at rows krow.., columns 32 tv + 16 (g & 1); lane 4 q' + p of the
                    // group addresses row q', columns 4 p .. 4 p + 3
                    const unsigned addr = lds_addr(&lv[buf][krow + (li >> 2)][32 * tv + 16 * ((lane >> 4) & 1) + 4 * (li & 3)]);
                    tr_issue(vlo[s2 * TV + tv], vhi[s2 * TV + tv], addr, addr + 8 * VLD * 2);
                }
            }
            tr_wait(vlo, vhi);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    const uint4 r = make_uint4(vlo[s2 * TV + tv].x, vlo[s2 * TV + tv].y, vhi[s2 * TV + tv].x, vhi[s2 * TV + tv].y);
                    o[tv] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8a, r), pf[kb][s2], o[tv], 0, 0, 0);
                }
        }
        // buffer buf ^ 1 was last read in iteration t - 1, which every wave left through the barrier below: safe to refill now
        if (t + 1 < ntiles) tile_store(buf ^ 1);
        __syncthreads();
    }
    // ---- normalise and store O[q][32 tv + (r & 3) + 8 (r >> 2) + 4 h]: four runs of four consecutive channels per tile
    const int q = q0 + l31;
    if (q < Nq && lse2 != nullptr && h == 0)  // log2-domain log-sum-exp of the row, for the backward kernels (attention_bwd.hip);
        lse2[(int64_t(b) * gridDim.y + head) * Nq + q] = l_run > 0.f ? m_run + log2f(l_run) : 1e30f;  // a fully masked row: P = 0 there
    if (q < Nq) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        __bf16* Ob = O + b * o_sb + head * o_sh + int64_t(q) * o_sn;
#pragma unroll
        for (int tv = 0; tv < TV; ++tv)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * tv + 8 * g + 4 * h;
                if (c < D) {
                    bf16x4a w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = (__bf16)(o[tv][4 * g + r] * inv);
                    *reinterpret_cast<bf16x4a*>(Ob + c) = w;
                }
            }
    }
}

}  // namespace xm3d

using namespace xm3d;

static int attention_fwd_impl(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t Nq, int32_t Nk,
                              int32_t D, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                              const int64_t* o_strides, const void* bias, int32_t bias_dtype, const int64_t* bias_strides,
                              float scale, float* lse2, void* stream) {
    XM3D_REQUIRE(B >= 0 && H >= 1 && Nq >= 0 && Nk >= 1 && D >= 8, "attention_fwd: bad sizes B=%d H=%d Nq=%d Nk=%d D=%d", B, H, Nq, Nk, D);
    XM3D_REQUIRE(D % 8 == 0 && D <= 160, "attention_fwd: head channels must be a multiple of 8 and <= 160 (got %d)", D);
    if (B == 0 || Nq == 0) return XM3D_OK;
    XM3D_REQUIRE(q && k && v && out && q_strides && k_strides && v_strides && o_strides, "attention_fwd: null pointer");
    XM3D_REQUIRE(bias_dtype >= 0 && bias_dtype <= 2 && (bias_dtype == 0 || (bias && bias_strides)), "attention_fwd: bad bias arguments");
    for (const int64_t* s : {q_strides, k_strides, v_strides})
        XM3D_REQUIRE(s[0] % 8 == 0 && s[1] % 8 == 0 && s[2] % 8 == 0, "attention_fwd: q/k/v strides must be multiples of 8 elements (16-byte rows)");
    XM3D_REQUIRE(o_strides[0] % 4 == 0 && o_strides[1] % 4 == 0 && o_strides[2] % 4 == 0, "attention_fwd: output strides must be multiples of 4");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 7) == 0,
                 "attention_fwd: q/k/v must be 16-byte aligned, out 8-byte aligned");
    const float sl2 = scale * 1.4426950408889634f;
    dim3 grid((Nq + 32 * ATT_NW - 1) / (32 * ATT_NW), H, B), blk(64 * ATT_NW);
    hipStream_t s = as_stream(stream);
    const int64_t z3[3] = {0, 0, 0};
    const int64_t* bs = bias_dtype ? bias_strides : z3;
#define XM3D_ATT(DQ_, DV_, BI_)                                                                                                   \
    hipLaunchKernelGGL((k_attn_fwd<DQ_, DV_, BI_>), grid, blk, 0, s, static_cast<const __bf16*>(q), static_cast<const __bf16*>(k),  \
                       static_cast<const __bf16*>(v), static_cast<__bf16*>(out), Nq, Nk, D, q_strides[0], q_strides[1], q_strides[2], \
                       k_strides[0], k_strides[1], k_strides[2], v_strides[0], v_strides[1], v_strides[2], o_strides[0], o_strides[1], \
                       o_strides[2], bias, bs[0], bs[1], bs[2], sl2, lse2)
#define XM3D_ATT_D(DQ_, DV_)                       \
    do {                                           \
        if (bias_dtype == 0) XM3D_ATT(DQ_, DV_, 0); \
        else if (bias_dtype == 1) XM3D_ATT(DQ_, DV_, 1); \
        else XM3D_ATT(DQ_, DV_, 2);                \
    } while (0)
    if (D <= 32) XM3D_ATT_D(32, 32);
    else if (D <= 48) XM3D_ATT_D(48, 64);
    else if (D <= 64) XM3D_ATT_D(64, 64);
    else if (D <= 80) XM3D_ATT_D(80, 96);
    else if (D <= 96) XM3D_ATT_D(96, 96);
    else if (D <= 128) XM3D_ATT_D(128, 128);
    else XM3D_ATT_D(160, 160);
#undef XM3D_ATT_D
#undef XM3D_ATT
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_attention_fwd(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t Nq, int32_t Nk,
                                  int32_t D, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                  const int64_t* o_strides, const void* bias, int32_t bias_dtype, const int64_t* bias_strides,
                                  float scale, void* stream) {
    return attention_fwd_impl(q, k, v, out, B, H, Nq, Nk, D, q_strides, k_strides, v_strides, o_strides, bias, bias_dtype, bias_strides, scale,
                              nullptr, stream);
}

extern "C" int xm3d_attention_fwd_lse(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t H, int32_t Nq, int32_t Nk,
                                      int32_t D, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                      const int64_t* o_strides, const void* bias, int32_t bias_dtype, const int64_t* bias_strides,
                                      float scale, float* lse2, void* stream) {
    XM3D_REQUIRE(lse2 != nullptr, "attention_fwd_lse: null lse2");
    return attention_fwd_impl(q, k, v, out, B, H, Nq, Nk, D, q_strides, k_strides, v_strides, o_strides, bias, bias_dtype, bias_strides, scale,
                              lse2, stream);
}
