// Fused attention forward to f32 ACCURACY on the 16-bit matrix cores:  O = softmax(Q K^T * scale + bias) V,  f32 in / f32 out.
//
// The fp32 configuration (the reference's arithmetic: run/train.py:178, no autocast) ran its attentions through torch's MATH backend:
// a (B, H, Nq, Nk) f32 score tensor in HBM (10.7 GB for one 64^2 self attention of 20 views), two library f32 GEMMs, softmax, and the
// isneginf / all / where passes of _safe_softmax - ~11 ms per 64^2 self attention, 15 % of that configuration's forward
// (profiles/r04_prof_lines_fp32.log).  This kernel is attention.hip's flash structure with every product computed from operands split
// in IEEE halves, like the f32-accurate convolution / GEMM (conv.hip, gemm.hip):
//     x = xh / s + xl / (2048 s)   (xh = half(x s), xl = half((x - xh / s) 2048 s): 22 mantissa bits, both terms at the magnitude of x s)
//     S^T = K Q^T       = [Kh Qh] + 2^-11 [Kh Ql + Kl Qh]           (+ 2^-22 |K||Q|), two f32 accumulators (leading / small terms)
//     O^T += V^T P^T    = [Vh Ph] + 2^-11 [Vh Pl + Vl Ph]           P = softmax numerators in f32, split the same way
// f32 softmax state exactly as in attention.hip.  Call sites: the SD UNet's 64^2 self / cross attention (head dim 40:
// models/modeling/meta_arch/ldm.py:425-446), mask-CLIP ViT-L (64: clip.py:239-270), the Mask2Former decoder (32:
// mask2former_transformer_decoder.py:17-178).  Head dims <= 64; the 80 / 160-wide heads of the UNet's inner levels (<= 1024 tokens,
// < 1 ms of scores traffic) stay on torch - two f32 output accumulators of 96 / 160 channels do not fit the register file.
// One 32-key block at a time (scores, softmax update, P V) keeps the live set at ~200 VGPRs.
#include "common.h"

namespace xm3d {

typedef float af_f32x16 __attribute__((ext_vector_type(16)));
typedef float af_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 af_f16x8 __attribute__((ext_vector_type(8)));

constexpr int AF_NW = 4;      // waves per workgroup (32 query rows each)
constexpr int AF_KV = 64;     // keys per staged tile (two 32-key blocks)
constexpr float AF_NEG = -1e30f;
constexpr float AF_SX = 0.0625f;          // operand scale s = 2^-4: |x| up to 1e6 before the half overflows
constexpr float AF_LO = 2048.f;           // the small term is stored times 2^11
constexpr float AF_INV_LO = 1.f / 2048.f;

__device__ __forceinline__ unsigned af_lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>(reinterpret_cast<const __attribute__((address_space(3))) char*>(
        reinterpret_cast<uintptr_t>(p))));
}
__device__ __forceinline__ void af_tr_issue(uint2& lo, uint2& hi, unsigned addr, unsigned addr2) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(addr), "v"(addr2) : "memory");
}
template <int N>
__device__ __forceinline__ void af_tr_wait(uint2 (&lo)[N], uint2 (&hi)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(lo[i]), "+v"(hi[i]));
}
// x (already times the operand scale) -> leading / small half terms
#define AF_SPLIT(x, hi, lo)                            \
    do {                                               \
        const float _x = (x);                          \
        const _Float16 _h = (_Float16)_x;              \
        (hi) = _h;                                     \
        (lo) = (_Float16)((_x - float(_h)) * AF_LO);   \
    } while (0)

// DQ / DV: head channels padded to a multiple of 16 / 32; D: real channel count (multiple of 4).  BIAS: 0 none, 1 additive f32.
template <int DQ, int DV, int BIAS>
__global__ __launch_bounds__(64 * AF_NW) void k_attn_fwd_f32acc(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, float* __restrict__ O, int Nq, int Nk, int D,
    int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh,
    int64_t o_sb, int64_t o_sn, int64_t o_sh, const float* __restrict__ bias, int64_t b_sb, int64_t b_sh, int64_t b_sq, float scale_log2e) {
    constexpr int SQ = DQ / 16, TV = DV / 32;
    constexpr int KLD = DQ + 8, VLD = DV + 8;  // padded LDS rows (halves)
    __shared__ __attribute__((aligned(16))) _Float16 lk[2][2][AF_KV][KLD];  // [buffer][hi | lo][key][channel]
    __shared__ __attribute__((aligned(16))) _Float16 lv[2][2][AF_KV][VLD];
    // additive bias: the score accumulator has the QUERY on the lane and the keys in registers - read at the point of use, every load
    // instruction touched 64 rows of the bias.  Each wave fetches its block with 8 keys x 8 rows per instruction (8 lines instead of
    // 64) and transposes it through a wave-private LDS strip, 8 keys at a time (the K / V tiles leave 6 KB for two workgroups per CU).
    __shared__ __attribute__((aligned(16))) float lb[BIAS != 0 ? AF_NW : 1][BIAS != 0 ? 32 : 1][BIAS != 0 ? 12 : 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = (blockIdx.x * AF_NW + wave) * 32;
    const float* Qb = Q + b * q_sb + head * q_sh;
    const float* Kb = K + b * k_sb + head * k_sh;
    const float* Vb = V + b * v_sb + head * v_sh;
    const int dchunks = D / 4;  // 16-byte (4-float) chunks per row of real data

    for (int i = tid; i < int(sizeof(lk) / 16); i += 64 * AF_NW) reinterpret_cast<uint4*>(&lk[0][0][0][0])[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < int(sizeof(lv) / 16); i += 64 * AF_NW) reinterpret_cast<uint4*>(&lv[0][0][0][0])[i] = make_uint4(0, 0, 0, 0);

    // Q^T fragments (B operand), split: lane (query l31, half h) holds Q[q][16 s + 8 h .. + 8]
    af_f16x8 qh[SQ], ql[SQ];
    {
        const int q = q0 + l31;
#pragma unroll
        for (int s = 0; s < SQ; ++s) {
            const int c = 16 * s + 8 * h;
            float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (q < Nq && c < D) {
                const float4 a = *reinterpret_cast<const float4*>(Qb + int64_t(q) * q_sn + c);
                x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w;
                if (c + 4 < D) {
                    const float4 a2 = *reinterpret_cast<const float4*>(Qb + int64_t(q) * q_sn + c + 4);
                    x[4] = a2.x, x[5] = a2.y, x[6] = a2.z, x[7] = a2.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) AF_SPLIT(x[j] * AF_SX, qh[s][j], ql[s][j]);
        }
    }

    // staging registers: 4-float chunks of the K / V tile owned by this thread
    constexpr int KCH = (AF_KV * (DQ / 4) + 64 * AF_NW - 1) / (64 * AF_NW);
    constexpr int VCH = (AF_KV * (DV / 4) + 64 * AF_NW - 1) / (64 * AF_NW);
    float4 kst[KCH], vst[VCH];
    auto tile_load = [&](int t) __attribute__((always_inline)) {
        const int k0 = t * AF_KV;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = i * 64 * AF_NW + tid, row = c / dchunks, col = c - row * dchunks;
            kst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < AF_KV && k0 + row < Nk) kst[i] = *reinterpret_cast<const float4*>(Kb + int64_t(k0 + row) * k_sn + col * 4);
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = i * 64 * AF_NW + tid, row = c / dchunks, col = c - row * dchunks;
            vst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < AF_KV && k0 + row < Nk) vst[i] = *reinterpret_cast<const float4*>(Vb + int64_t(k0 + row) * v_sn + col * 4);
        }
    };
    auto put4 = [&](_Float16* hi, _Float16* lo, float4 v) __attribute__((always_inline)) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 a, c;
        AF_SPLIT(v.x * AF_SX, a[0], c[0]);
        AF_SPLIT(v.y * AF_SX, a[1], c[1]);
        AF_SPLIT(v.z * AF_SX, a[2], c[2]);
        AF_SPLIT(v.w * AF_SX, a[3], c[3]);
        *reinterpret_cast<f16x4*>(hi) = a;
        *reinterpret_cast<f16x4*>(lo) = c;
    };
    auto tile_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = i * 64 * AF_NW + tid, row = c / dchunks, col = c - row * dchunks;
            if (row < AF_KV) put4(&lk[buf][0][row][col * 4], &lk[buf][1][row][col * 4], kst[i]);
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = i * 64 * AF_NW + tid, row = c / dchunks, col = c - row * dchunks;
            if (row < AF_KV) put4(&lv[buf][0][row][col * 4], &lv[buf][1][row][col * 4], vst[i]);
        }
    };

    af_f32x16 ob[TV], os[TV];  // leading / small-term accumulators of O^T
#pragma unroll
    for (int t = 0; t < TV; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[t][r] = os[t][r] = 0.f;
    float m_run = AF_NEG, l_run = 0.f;
    const float sc2 = scale_log2e * (1.f / (AF_SX * AF_SX));  // undoes the operand scales of Q and K

    int64_t brow[BIAS != 0 ? 4 : 1];  // bias rows this lane fetches: 8 j + (lane >> 3), clamped to the last query (never used past it)
    if constexpr (BIAS != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) brow[j] = b * b_sb + head * b_sh + int64_t(min(q0 + 8 * j + (lane >> 3), Nq - 1)) * b_sq;
    }
    const int ntiles = (Nk + AF_KV - 1) / AF_KV;
    tile_load(0);
    __syncthreads();  // zero fill done
    tile_store(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) tile_load(t + 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int kbase = t * AF_KV + 32 * kb;
            if (kbase >= Nk) continue;  // wave-uniform: a block past the end contributes nothing
            float braw[BIAS != 0 ? 16 : 1];  // requested before the score products: group g = keys 8 g .. 8 g + 7 of the block
            if constexpr (BIAS != 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kl = min(kbase + 8 * g + (lane & 7), Nk - 1);  // keys past Nk: clamped copies, masked by the key bound
#pragma unroll
                    for (int j = 0; j < 4; ++j) braw[4 * g + j] = bias[brow[j] + kl];
                }
            }
            // ---- S^T = K Q^T for this 32-key block: leading and small terms
            af_f32x16 sb, ss;
#pragma unroll
            for (int r = 0; r < 16; ++r) sb[r] = ss[r] = 0.f;
#pragma unroll
            for (int s = 0; s < SQ; ++s) {
                const af_f16x8 kfh = *reinterpret_cast<const af_f16x8*>(&lk[buf][0][32 * kb + l31][16 * s + 8 * h]);
                const af_f16x8 kfl = *reinterpret_cast<const af_f16x8*>(&lk[buf][1][32 * kb + l31][16 * s + 8 * h]);
                ss = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, ql[s], ss, 0, 0, 0);
                ss = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl, qh[s], ss, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, qh[s], sb, 0, 0, 0);
            }
            // ---- scale, bias, key bound; running maximum (log2 domain)
            // whole unmasked blocks take the short path: pairs of scores through the packed f32 pipe (v_pk_fma_f32 / v_pk_add_f32),
            // the maximum on the unscaled scores (scale > 0), no key-bound / mask tests
            const bool plain = BIAS == 0 && kbase + 32 <= Nk;  // wave-uniform
            float mx = AF_NEG;
            float bq4[4] = {0.f, 0.f, 0.f, 0.f};
            if (plain) {
                const af_f32x2 il2 = {AF_INV_LO, AF_INV_LO};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const af_f32x2 s2v = {ss[r], ss[r + 1]}, sbv = {sb[r], sb[r + 1]};
                    const af_f32x2 tv2 = __builtin_elementwise_fma(s2v, il2, sbv);
                    sb[r] = tv2[0];
                    sb[r + 1] = tv2[1];
                    mx = fmaxf(mx, fmaxf(tv2[0], tv2[1]));
                }
                mx *= sc2;
            } else
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * h;
                float sc = fmaf(ss[r], AF_INV_LO, sb[r]) * sc2;
                if constexpr (BIAS != 0) {
                    if ((r & 3) == 0) {  // group r / 4 through the strip: rows on the lanes -> this lane's four consecutive keys
#pragma unroll
                        for (int j = 0; j < 4; ++j) lb[wave][8 * j + (lane >> 3)][lane & 7] = braw[r + j];
                        __builtin_amdgcn_wave_barrier();  // wave-private strip; the LDS executes one wave's accesses in issue order
                        const float4 v4 = *reinterpret_cast<const float4*>(&lb[wave][l31][4 * h]);
                        bq4[0] = v4.x, bq4[1] = v4.y, bq4[2] = v4.z, bq4[3] = v4.w;
                        __builtin_amdgcn_wave_barrier();
                    }
                    const float bv = bq4[r & 3];
                    sc = bv < -1e29f ? AF_NEG : fmaf(bv, 1.4426950408889634f, sc);
                }
                if (key >= Nk) sc = AF_NEG;
                sb[r] = sc;
                mx = fmaxf(mx, sc);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
            af_f16x8 ph[2], pl[2];
            if (plain) {
                const af_f32x2 scv = {sc2, sc2}, nm2 = {-m_new, -m_new};
                af_f32x2 ps2 = {0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const af_f32x2 sv = {sb[8 * s2 + j], sb[8 * s2 + j + 1]};
                        const af_f32x2 e = __builtin_elementwise_fma(sv, scv, nm2);
                        af_f32x2 p;
                        p[0] = __builtin_amdgcn_exp2f(e[0]);
                        p[1] = __builtin_amdgcn_exp2f(e[1]);
                        ps2 += p;
                        AF_SPLIT(p[0], ph[s2][j], pl[s2][j]);
                        AF_SPLIT(p[1], ph[s2][j + 1], pl[s2][j + 1]);
                    }
                psum = ps2[0] + ps2[1];
            } else
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float sc = sb[8 * s2 + j];
                    const float p = sc <= AF_NEG ? 0.f : __builtin_amdgcn_exp2f(sc - m_new);
                    psum += p;
                    AF_SPLIT(p, ph[s2][j], pl[s2][j]);
                }
            psum += __shfl_xor(psum, 32);
            l_run = l_run * alpha + psum;
            m_run = m_new;
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int tv = 0; tv < TV; ++tv)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        ob[tv][r] *= alpha;
                        os[tv][r] *= alpha;
                    }
            }
            // ---- O^T += V^T P^T: V^T (both planes) through transposing reads of the row-major tiles
            uint2 vhl[2 * TV], vhh[2 * TV], vll[2 * TV], vlh[2 * TV];  // hi plane (lo / hi row groups), lo plane
            const int li = lane & 15;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int krow = 32 * kb + 16 * s2 + 4 * h;
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    const int col = 32 * tv + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
                    const unsigned a0 = af_lds_addr(&lv[buf][0][krow + (li >> 2)][col]);
                    const unsigned a1 = af_lds_addr(&lv[buf][1][krow + (li >> 2)][col]);
                    af_tr_issue(vhl[s2 * TV + tv], vhh[s2 * TV + tv], a0, a0 + 8 * VLD * 2);
                    af_tr_issue(vll[s2 * TV + tv], vlh[s2 * TV + tv], a1, a1 + 8 * VLD * 2);
                }
            }
            af_tr_wait(vhl, vhh);
            af_tr_wait(vll, vlh);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int tv = 0; tv < TV; ++tv) {
                    const int i = s2 * TV + tv;
                    const af_f16x8 vfh = __builtin_bit_cast(af_f16x8, make_uint4(vhl[i].x, vhl[i].y, vhh[i].x, vhh[i].y));
                    const af_f16x8 vfl = __builtin_bit_cast(af_f16x8, make_uint4(vll[i].x, vll[i].y, vlh[i].x, vlh[i].y));
                    os[tv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh, pl[s2], os[tv], 0, 0, 0);
                    os[tv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl, ph[s2], os[tv], 0, 0, 0);
                    ob[tv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh, ph[s2], ob[tv], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles) tile_store(buf ^ 1);
        __syncthreads();
    }
    // ---- normalise and store O[q][32 tv + (r & 3) + 8 (r >> 2) + 4 h]
    const int q = q0 + l31;
    if (q < Nq) {
        const float inv = l_run > 0.f ? (1.f / AF_SX) / l_run : 0.f;  // undoes V's operand scale
        float* Ob = O + b * o_sb + head * o_sh + int64_t(q) * o_sn;
#pragma unroll
        for (int tv = 0; tv < TV; ++tv)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * tv + 8 * g + 4 * h;
                if (c < D) {
                    float4 w;
                    w.x = fmaf(os[tv][4 * g], AF_INV_LO, ob[tv][4 * g]) * inv;
                    w.y = fmaf(os[tv][4 * g + 1], AF_INV_LO, ob[tv][4 * g + 1]) * inv;
                    w.z = fmaf(os[tv][4 * g + 2], AF_INV_LO, ob[tv][4 * g + 2]) * inv;
                    w.w = fmaf(os[tv][4 * g + 3], AF_INV_LO, ob[tv][4 * g + 3]) * inv;
                    *reinterpret_cast<float4*>(Ob + c) = w;
                }
            }
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_attention_fwd_f32(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t H, int32_t Nq, int32_t Nk,
                                      int32_t D, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                                      const int64_t* o_strides, const float* bias, const int64_t* bias_strides, float scale, void* stream) {
    XM3D_REQUIRE(B >= 0 && H >= 1 && Nq >= 0 && Nk >= 1 && D >= 8, "attention_fwd_f32: bad sizes B=%d H=%d Nq=%d Nk=%d D=%d", B, H, Nq, Nk, D);
    XM3D_REQUIRE(D % 8 == 0 && D <= 64, "attention_fwd_f32: head channels must be a multiple of 8 and <= 64 (got %d)", D);
    if (B == 0 || Nq == 0) return XM3D_OK;
    XM3D_REQUIRE(q && k && v && out && q_strides && k_strides && v_strides && o_strides, "attention_fwd_f32: null pointer");
    XM3D_REQUIRE(!bias || bias_strides, "attention_fwd_f32: bias needs its strides");
    for (const int64_t* s : {q_strides, k_strides, v_strides, o_strides})
        XM3D_REQUIRE(s[0] % 4 == 0 && s[1] % 4 == 0 && s[2] % 4 == 0, "attention_fwd_f32: strides must be multiples of 4 elements (16-byte rows)");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "attention_fwd_f32: q / k / v / out must be 16-byte aligned");
    const float sl2 = scale * 1.4426950408889634f;
    dim3 grid((Nq + 32 * AF_NW - 1) / (32 * AF_NW), H, B), blk(64 * AF_NW);
    hipStream_t s = as_stream(stream);
    const int64_t z3[3] = {0, 0, 0};
    const int64_t* bs = bias ? bias_strides : z3;
#define XM3D_AF(DQ_, DV_, BI_)                                                                                                            \
    hipLaunchKernelGGL((k_attn_fwd_f32acc<DQ_, DV_, BI_>), grid, blk, 0, s, q, k, v, out, Nq, Nk, D, q_strides[0], q_strides[1], q_strides[2], \
                       k_strides[0], k_strides[1], k_strides[2], v_strides[0], v_strides[1], v_strides[2], o_strides[0], o_strides[1],       \
                       o_strides[2], bias, bs[0], bs[1], bs[2], sl2)
#define XM3D_AF_D(DQ_, DV_)                \
    do {                                   \
        if (bias) XM3D_AF(DQ_, DV_, 1);    \
        else XM3D_AF(DQ_, DV_, 0);         \
    } while (0)
    if (D <= 32) XM3D_AF_D(32, 32);
    else if (D <= 48) XM3D_AF_D(48, 64);
    else XM3D_AF_D(64, 64);
#undef XM3D_AF_D
#undef XM3D_AF
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
