// Linear layer / 1x1 convolution over bf16 token rows on the gfx950 matrix cores, with what follows it folded into the epilogue:
//
//     out = act( x @ W^T + bias ) (+ residual)            x (M, K) bf16 rows, W (N, K), out (M, N) bf16      act: none | GELU | QuickGELU
//     out = (x @ Wv^T + bv) * GELU(x @ Wg^T + bg)          GEGLU: W = [Wv; Wg] (2 N_out, K), out (M, N_out)
//
// These are the dense projections around the attentions of the 3D-conditioned Stable-Diffusion UNet - to_q / to_k / to_v / to_out,
// the GEGLU feed-forward and the 1x1 proj_in / proj_out of ldm's SpatialTransformer, reached from
// /root/reference/models/modeling/meta_arch/ldm.py:425-446 - the 1x1 q / k / v / proj_out of the VAE's AttnBlock (:386-414,
// :448-490) and the c_fc / c_proj / in_proj / out_proj of the mask-CLIP ViT-L (/root/reference/models/modeling/meta_arch/clip.py:
// 239-270): north_star's "MFMA on the dense QKV / ResBlock / CLIP GEMMs".  The ResBlock 3x3 convolutions are conv.hip; this file
// is the same machine with ONE filter tap and a flat token list instead of an image tile.
//
// Decomposition (one workgroup = 8 waves = 256 token rows x CT output columns, CT = 256 or 128):
//   * K loop = chunks of 64 input channels x 4 MFMA k-steps of 16.
//   * TOKENS (MFMA B operand) go through LDS: per chunk 256 rows x 128 bytes, 144-byte row stride (the 32-row fragment reads
//     ds_read_b128 are bank-conflict free), two buffers, one workgroup barrier per chunk.  A thread owns four 16-byte pieces per
//     chunk; they are REQUESTED two chunks ahead (register double buffer: a chunk lasts ~2 k cycles per SIMD, an HBM miss longer)
//     and written to LDS one piece per k-step of the chunk before they are needed.
//   * WEIGHTS (MFMA A operand) never touch LDS: every wave owns 32 output columns and streams exactly its own fragments from a
//     pre-packed, fragment-ordered image (xm3d_gemm_pack_weight) through L2 into an 8-deep register ring, one 16-byte load per
//     lane and k-step.  No wave waits for another wave's loads.
//   * v_mfma_f32_32x32x16_bf16, weights as A (rows = output columns), tokens as B: a lane's accumulator registers are 16
//     CONSECUTIVE output columns of one token (the packing permutes the rows so), i.e. two 16-byte stores per 32 x 32 tile, and for
//     GEGLU the value / gate halves of a column pair sit in lanes l and l + 32 of one wave (the packing interleaves 16 value
//     columns with their 16 gate columns per row block): one cross-half exchange, no second pass over a (M, 2 N_out) tensor.
// Two chunks are unrolled per loop iteration so that ring slots and piece registers are static.
// Bound: MFMA for K >= 1024 (CLIP, UNet mid levels); HBM for the 320-wide level (2 M K N flop over 2 M (K + N) bytes = 160 flop/B).
// Algorithmic FLOP = 2 M K N; bytes = M (K + N_out [+ N_out residual]) * 2 + N K * 2.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace xm3d {

typedef float gm_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));

constexpr int GM_MT = 256;               // token rows per workgroup
constexpr int GM_KC = 64;                // K granularity (chunks are 64 or, when K allows, 128 channels: template KC)
constexpr int GM_D = 8;                  // weight ring depth in k-steps (= two chunks)

enum { GM_ACT_NONE = 0, GM_ACT_GELU = 1, GM_ACT_QUICK_GELU = 2, GM_ACT_GEGLU = 3, GM_ACT_RELU = 4 };  // RELU: the NONE instantiation with floor = 0

struct GemmArgs {
    const __bf16* x;         // (M, K), row stride ldx elements; GF_CONV: the (B, Hin, Win, Cin) channels-last image
    const __bf16* wp;        // packed weights (xm3d_gemm_pack_weight)
    const float* bias;       // (N) or null; GEGLU: (2 N_out), value half first
    const __bf16* residual;  // (M, N_out), row stride ldr, or null (GF_OUT32: f32)
    __bf16* out;             // (M, N_out), row stride ldo (GF_OUT32: f32; split-K: slab blockIdx.y at + blockIdx.y * M * ldo)
    const __bf16* x2;        // GF_SPLIT3: the small-term plane of x (same layout as x)
    const __bf16* wp2;       // GF_SPLIT3: the packed small-term image of W
    const float* accin;      // GF_OUT32: (M, N_out) f32, row stride ldo, added BEFORE the activation (accumulating passes), or null
    float alpha;             // GF_OUT32: the product is scaled by alpha (the power-of-two scale of a split term pair) before anything is added
    float floor = -INFINITY;  // lower clamp after the activation, before the residual: 0 = ReLU (act code 4), -inf = none (one v_max per output)
    int M, K, N;             // N = rows of W (GEGLU: 2 N_out)
    int ldx, ldr, ldo;
    int nct;                 // column tiles
    int chunks_per_split;    // K chunks per blockIdx.y slice (split-K; = K / KC without)
    // GF_CONV: row m = output pixel (b, oy, ox) of a (ksz x ksz, stride, zero padding) convolution, K = ksz * ksz * Cin in (ky, kx, c) order
    int Hin, Win, Cin, Ho, Wo, stride, pad_t, pad_l, ksz;
    float xscale = 1.f;      // GF_XF32: the power-of-two operand scale s of x
    int* flag = nullptr;     // GF_XF32: sticky device flag, set when |x s| leaves the half range (xm3d_check_flag)
};

// FL bits.  GF_CONV: implicit-GEMM convolution - the token rows are gathered from the image while they are staged (no im2col tensor):
// strided / small-map / 1x1 convolutions that conv.hip's 32-pixel-wide halo tiles do not take.  GF_OUT32: f32 output (accumulating
// passes of the f32-accurate GEMM, partial slabs of the deterministic split-K).  GF_F16: operands are IEEE half (v_mfma_f32_32x32x16_f16):
// the two-term split of an f32 operand in halves carries 22 mantissa bits (bf16: 16)
// GF_SPLIT3 (with GF_F16 | GF_OUT32): the whole f32-accurate product in ONE launch - both planes of x (x, x2) staged, both weight images
// (wp, wp2) streamed, three MFMAs per k-step into ONE accumulator: x_hi w_lo + x_lo w_hi + x_hi w_hi.  Needs the small terms at their TRUE
// scale (lo = half(x s - hi), not times 2^11): fine while |x s| >= 2^-3 keeps lo a normal half - smaller |x| lose only absolute precision
// (2^-24 / s), which is invisible next to 1e-6 of max|out|.  A third of the token / output traffic of the three accumulating passes.
// GF_XF32 (with GF_SPLIT3): x is the F32 tensor itself; a piece is 8 floats (two 16-byte loads) and is split into its two half planes
// (hi = half(x s), lo = half(x s - hi): the arithmetic of k_split_nhwc, bit for bit) on its way into LDS - the separate split pass in front of
// every f32-accurate GEMM (one read + one write of the activation, 9.9 % of the fp32 configuration's device time) is gone.
enum { GF_CONV = 1, GF_OUT32 = 2, GF_F16 = 4, GF_SPLIT3 = 8, GF_XF32 = 16 };
typedef _Float16 gm_f16x8 __attribute__((ext_vector_type(8)));

// GELU(x) = x/2 (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, two hardware transcendentals)
// instead of erff's branchy polynomial: the epilogue of a GEGLU tile evaluates it 128 times per lane
__device__ __forceinline__ float gm_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    const float p = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = 1.f - p * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);  // erf(|x| / sqrt 2)
    return 0.5f * x * (1.f + copysignf(e, x));
}

// The same GELU for a PAIR of gates, arranged for the packed f32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two values per issue slot) - the GEGLU
// epilogue evaluates 64 of them per lane and, for the K = 320 / 640 layers of the UNet, costs more vector time than the K loop has MFMA
// time.  With p(t) the A&S polynomial and q = p(t) 2^(-x^2 log2(e) / 2) / 2:   GELU(x) = x q + max(x, 0) (1 - 2 q)   (x >= 0: x - x q,
// x < 0: x q), the same function as gm_gelu up to the rounding of the rearranged products.
typedef float gm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gm_f32x2 gm_gelu2(gm_f32x2 x) {
    const gm_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const gm_f32x2 d = __builtin_elementwise_fma(ax, gm_f32x2{0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f}, gm_f32x2{1.f, 1.f});
    const gm_f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    gm_f32x2 p = __builtin_elementwise_fma(t, gm_f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, gm_f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
    p = __builtin_elementwise_fma(t, p, gm_f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
    p = __builtin_elementwise_fma(t, p, gm_f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
    p = __builtin_elementwise_fma(t, p, gm_f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
    p = p * t;
    const gm_f32x2 arg = (x * x) * gm_f32x2{-0.5f * 1.4426950408889634f, -0.5f * 1.4426950408889634f};
    const gm_f32x2 ex = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
    const gm_f32x2 q = p * ex;
    const gm_f32x2 xq = x * q;
    const gm_f32x2 rx = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    const gm_f32x2 w = __builtin_elementwise_fma(q, gm_f32x2{-2.f, -2.f}, gm_f32x2{1.f, 1.f});
    return __builtin_elementwise_fma(rx, w, xq);
}

// NW = 8: 512 threads, 256 token rows (one workgroup per CU).  NW = 4 (CT = 128 only): 256 threads, 128 token rows, a wave = one
// 32-column block x all 128 rows; two to three workgroups per CU - the geometry for small M (M = 5 k rows x 256-row tiles leaves
// the chip two thirds empty) and for overlapping one workgroup's prologue / epilogue with another's K loop
template <int CT, int ACT, int KC, int NW, int FL>
__global__ __launch_bounds__(NW * 64) void k_gemm(const GemmArgs a) {
    static_assert(NW == 8 || (NW == 4 && CT == 128), "geometries");
    constexpr bool CONV = (FL & GF_CONV) != 0, OUT32 = (FL & GF_OUT32) != 0, F16 = (FL & GF_F16) != 0, SP3 = (FL & GF_SPLIT3) != 0;
    constexpr bool XF32 = (FL & GF_XF32) != 0;
    static_assert(!XF32 || SP3, "f32 rows are split on the fly only in the one-launch form");
    static_assert(!OUT32 || ACT != GM_ACT_GEGLU, "GEGLU has no f32-output form");
    static_assert(!SP3 || (F16 && OUT32 && CT == 128 && KC == 64), "the one-launch split form: halves, f32 out, 128-column tiles, 64-channel chunks");
    constexpr int NPLN = SP3 ? 2 : 1;      // operand planes staged per buffer
    constexpr int MT = NW * 32;            // token rows per workgroup
    constexpr int NTH = NW * 64;
    constexpr int NT = CT == 256 ? 8 : 4;  // 32-token tiles per wave
    constexpr int NG = NT / 4;             // groups of 4 MFMAs per k-step
    constexpr int KS = KC / 16;            // k-steps per chunk = 16-byte pieces per thread and chunk (4 or 8)
    constexpr int PSTR = KC * 2 + 16;      // bytes per staged row (144 / 272: the 32-row fragment reads stay bank-conflict free)
    constexpr int ASZ = MT * PSTR;         // bytes per LDS buffer
    constexpr int RPR = NTH / (KC / 8);    // rows staged per round
    constexpr int D = ((CT == 256 && KC == 128) || SP3) ? 4 : GM_D;  // weight ring depth in k-steps (register budget: 128 accumulators + 32 pieces there; two rings with SP3)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = (CT == 256 || NW == 4) ? wave : wave >> 1;         // 32-column row block of this wave inside the tile
    const int nbase = (CT == 128 && NW == 8) ? 4 * (wave & 1) : 0;    // first 32-token tile of this wave
    const int l31 = lane & 31, h = lane >> 5;

    int bid;  // XCD-aware order (speed only): an XCD's L2 sees a contiguous run of tiles; column tiles of one row tile are adjacent
    {
        const int n = gridDim.x, i = blockIdx.x, xcd = i & 7, qd = n >> 3, r = n & 7;
        bid = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (i >> 3);
    }
    const int ct = bid % a.nct, mt = bid / a.nct;
    const int row0 = mt * MT, M = a.M;
    // split-K: slice blockIdx.y owns chunks [cfirst, cfirst + nch) of the K loop (the host makes every slice non-empty)
    const int cfirst = blockIdx.y * a.chunks_per_split;
    const int nch = min(a.chunks_per_split, a.K / KC - cfirst), nk = nch * KS;
    // N that is not a multiple of the column tile (UNet: 320, 960 with CT = 128): the row blocks past N in the last tile belong to
    // waves that only stage tokens - no weight stream, no fragment reads, no MFMAs (their SIMD's other wave runs alone)
    const int gblk = ct * (CT / 32) + wb;  // global 32-row block of W
    const bool active = gblk * 32 < a.N;

    // ---- weight stream of this wave: 1 KiB (64 lanes x 16 bytes = one A fragment) per k-step, contiguous
    const char* const wstream = reinterpret_cast<const char*>(a.wp) + ((int64_t(ct) * (CT / 32) + wb) * (a.K / 16) + int64_t(cfirst) * KS) * 1024 + lane * 16;
    const char* const wstream2 = SP3 ? reinterpret_cast<const char*>(a.wp2) + (wstream - reinterpret_cast<const char*>(a.wp)) : nullptr;
    gm_bf16x8 wr[D], wr2[SP3 ? D : 1];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        wr[i] = *reinterpret_cast<const gm_bf16x8*>(wstream + int64_t(i < nk ? i : nk - 1) * 1024);
        if constexpr (SP3) wr2[i] = *reinterpret_cast<const gm_bf16x8*>(wstream2 + int64_t(i < nk ? i : nk - 1) * 1024);
    }

    // ---- token staging: thread owns the 16-byte piece (row prow + RPR r, channels 8 kc .. 8 kc + 7) of every chunk, r = 0 .. KS - 1.
    // Rows past M read row M - 1 (their results are never stored).  ONE register per piece: piece r of chunk c + 1 is written to LDS
    // in k-step r of chunk c and the register is re-requested right there for chunk c + 2 - every load has a whole chunk to land
    const int kc = tid % (KC / 8), prow = tid / (KC / 8);
    int aoff[KS];       // element offsets (32 bit: the host checks the extent); GF_CONV: of the pixel under tap (0, 0), may be negative
    unsigned vmask[KS];  // GF_CONV: bit ky: input row oy * stride - pad_t + ky is inside the image; bit 4 + kx: the same for columns
#pragma unroll
    for (int r = 0; r < KS; ++r) {
        const int row = row0 + prow + RPR * r;
        const int rc = row < M ? row : M - 1;
        if constexpr (CONV) {
            const int ox = rc % a.Wo, t = rc / a.Wo, oy = t % a.Ho, b = t / a.Ho;
            const int iy0 = oy * a.stride - a.pad_t, ix0 = ox * a.stride - a.pad_l;
            aoff[r] = ((b * a.Hin + iy0) * a.Win + ix0) * a.Cin + kc * 8;
            unsigned m = 0;
            for (int k = 0; k < a.ksz; ++k) {
                m |= unsigned(iy0 + k >= 0 && iy0 + k < a.Hin) << k;
                m |= unsigned(ix0 + k >= 0 && ix0 + k < a.Win) << (4 + k);
            }
            vmask[r] = m;
        } else {
            aoff[r] = int(unsigned(rc) * unsigned(a.ldx)) + kc * 8;
            vmask[r] = 0;
        }
    }
    char* const a_wr = smem + prow * PSTR + kc * 16;  // + r * RPR * PSTR + buffer
    gm_u32x4 raw[KS], raw2[SP3 ? KS : 1];
    // GF_CONV: chunk -> (filter tap, channel slice), walked incrementally (wave-uniform): `ld` is the chunk being requested, `wr_need`
    // the validity bits of the chunk whose pieces are being written to LDS (requested one chunk earlier)
    const int cpt = CONV ? a.Cin / KC : 1;  // chunks per tap
    struct Tap {
        int sl, kx, ky, off;
        unsigned need;
    };
    Tap ld;
    unsigned wr_need = 0;
    auto tap_set = [&](Tap& t, int cabs) __attribute__((always_inline)) {
        const int tap = cabs / cpt;
        t.sl = cabs - tap * cpt;
        t.ky = tap / a.ksz;
        t.kx = tap - t.ky * a.ksz;
        t.off = (t.ky * a.Win + t.kx) * a.Cin + t.sl * KC;
        t.need = (1u << t.ky) | (16u << t.kx);
    };
    auto tap_next = [&](Tap& t) __attribute__((always_inline)) {
        if (++t.sl == cpt) {
            t.sl = 0;
            if (++t.kx == a.ksz) t.kx = 0, ++t.ky;
        }
        t.off = (t.ky * a.Win + t.kx) * a.Cin + t.sl * KC;
        t.need = (1u << t.ky) | (16u << t.kx);
    };
    // request piece r of the chunk `ld` points at (plain: of local chunk c, clamped to the slice's last)
    auto a_load = [&](int r, int c) __attribute__((always_inline)) {
        int off;
        if constexpr (CONV) {
            const bool ok = (vmask[r] & ld.need) == ld.need;
            off = ok ? aoff[r] + ld.off : kc * 8;
        } else {
            off = aoff[r] + (cfirst + (c < nch ? c : nch - 1)) * KC;
        }
        if constexpr (XF32) {  // 8 consecutive floats of the row
            const float* const xf = reinterpret_cast<const float*>(a.x) + off;
            raw[r] = *reinterpret_cast<const gm_u32x4*>(xf);
            raw2[r] = *reinterpret_cast<const gm_u32x4*>(xf + 4);
        } else {
            raw[r] = *reinterpret_cast<const gm_u32x4*>(a.x + off);
            if constexpr (SP3) raw2[r] = *reinterpret_cast<const gm_u32x4*>(a.x2 + off);
        }
    };
    // GF_XF32: piece r (8 floats in raw / raw2) -> its two half planes.  t = x s is exact (s a power of two), hi = half(t) to nearest,
    // t - hi is exact in f32, lo = half(t - hi): the values k_split_nhwc writes.  |t| beyond the half range raises the sticky flag.
    bool xbad = false;
    auto a_split = [&](int r, gm_u32x4& ph, gm_u32x4& pl) __attribute__((always_inline)) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const f2 s2 = {a.xscale, a.xscale};
        float tmax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // pairs on the packed pipe: v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_add_f32
            const f2 x2 = {__uint_as_float(j < 2 ? raw[r][2 * j] : raw2[r][2 * j - 4]), __uint_as_float(j < 2 ? raw[r][2 * j + 1] : raw2[r][2 * j - 3])};
            const f2 t2 = x2 * s2;
            const h2 hh = __builtin_convertvector(t2, h2);
            const f2 d2 = t2 - __builtin_convertvector(hh, f2);
            const h2 ll = __builtin_convertvector(d2, h2);
            ph[j] = __builtin_bit_cast(unsigned, hh);
            pl[j] = __builtin_bit_cast(unsigned, ll);
            tmax = fmaxf(tmax, fmaxf(fabsf(t2[0]), fabsf(t2[1])));
        }
        xbad |= !(tmax <= 65504.f);  // also true for a NaN input
    };
    // both planes of piece r into the staging buffer at `dst` (plane 1 at + ASZ); GF_CONV: zero padding under a tap outside the image
    auto a_store = [&](int r, char* dst) __attribute__((always_inline)) {
        if constexpr (XF32) {
            gm_u32x4 ph, pl;
            a_split(r, ph, pl);
            if constexpr (CONV) {
                if ((vmask[r] & wr_need) != wr_need) ph = pl = gm_u32x4{0u, 0u, 0u, 0u};
            }
            *reinterpret_cast<gm_u32x4*>(dst) = ph;
            *reinterpret_cast<gm_u32x4*>(dst + ASZ) = pl;
        }
    };
    auto a_piece = [&](int r) __attribute__((always_inline)) -> gm_u32x4 {  // zero padding: a piece under a tap outside the image
        if constexpr (CONV) return (vmask[r] & wr_need) == wr_need ? raw[r] : gm_u32x4{0u, 0u, 0u, 0u};
        else return raw[r];
    };
    auto a_piece2 = [&](int r) __attribute__((always_inline)) -> gm_u32x4 {
        if constexpr (CONV) return (vmask[r] & wr_need) == wr_need ? raw2[SP3 ? r : 0] : gm_u32x4{0u, 0u, 0u, 0u};
        else return raw2[SP3 ? r : 0];
    };

    // tokens as B operand: row (nbase + n) * 32 + l31 of the tile, 16-byte granule 2 ks + h
    const char* const xbase = smem + (nbase * 32 + l31) * PSTR + h * 16;

    gm_f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

    // ---- prologue: chunk 0 into buffer 0, chunk 1 requested
    if constexpr (CONV) tap_set(ld, cfirst);
#pragma unroll
    for (int r = 0; r < KS; ++r) a_load(r, 0);
    wr_need = ld.need;
#pragma unroll
    for (int r = 0; r < KS; ++r) {
        if constexpr (XF32) {
            a_store(r, a_wr + r * RPR * PSTR);
        } else {
            *reinterpret_cast<gm_u32x4*>(a_wr + r * RPR * PSTR) = a_piece(r);
            if constexpr (SP3) *reinterpret_cast<gm_u32x4*>(a_wr + ASZ + r * RPR * PSTR) = a_piece2(r);
        }
    }
    if constexpr (CONV) {
        if (nch > 1) tap_next(ld);
    }
#pragma unroll
    for (int r = 0; r < KS; ++r) a_load(r, 1);
    __syncthreads();

    // chunk c (parity P = c & 1): MFMAs on LDS buffer P; in k-step ks piece ks of chunk c + 1 goes to buffer 1 - P and is re-requested
    // for chunk c + 2; ring slot of k-step ks = (P KS + ks) mod D
    auto chunk = [&](int c, auto par_tag, auto act_tag) __attribute__((always_inline)) {
        constexpr int P = decltype(par_tag)::value;
        constexpr bool ACTV = decltype(act_tag)::value;  // a wave past N only stages (one branch per chunk, straight-line bodies)
        const char* const xl = xbase + P * NPLN * ASZ;  // buffer P: plane 0 (leading terms), plane 1 at + ASZ (GF_SPLIT3)
        char* const anext = a_wr + (1 - P) * NPLN * ASZ;
        if constexpr (CONV) {  // the pieces in flight belong to chunk c + 1 (written in this chunk); requests go to chunk c + 2
            wr_need = ld.need;
            if (c + 2 < nch) tap_next(ld);
        }
        gm_bf16x8 xf[2][4], xf2[SP3 ? 2 : 1][4];
        constexpr int XS = KS * NG;  // x-sets (4 token tiles each) per chunk
        auto x_load = [&](int xs, int s) __attribute__((always_inline)) {
            const int ks = xs / NG, half = xs % NG;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                xf[s][n] = *reinterpret_cast<const gm_bf16x8*>(xl + (half * 4 + n) * 32 * PSTR + ks * 32);
                if constexpr (SP3) xf2[s][n] = *reinterpret_cast<const gm_bf16x8*>(xl + ASZ + (half * 4 + n) * 32 * PSTR + ks * 32);
            }
        };
        if (ACTV) x_load(0, 0);
#pragma unroll
        for (int g = 0; g < XS; ++g) {
            const int ks = g / NG, half = g % NG;
            const int slot = (P * KS + ks) % D;
            __builtin_amdgcn_sched_barrier(0);
            if (ACTV && g + 1 < XS) x_load(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (half == 0) {
                if constexpr (XF32) {
                    a_store(ks, anext + ks * RPR * PSTR);
                } else {
                    *reinterpret_cast<gm_u32x4*>(anext + ks * RPR * PSTR) = a_piece(ks);
                    if constexpr (SP3) *reinterpret_cast<gm_u32x4*>(anext + ASZ + ks * RPR * PSTR) = a_piece2(ks);
                }
                a_load(ks, c + 2);
            }
            if (ACTV) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if constexpr (SP3) {  // small products first, the leading one last
                        acc[half * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gm_f16x8, wr2[slot]), __builtin_bit_cast(gm_f16x8, xf[g & 1][n]),
                                                                                   acc[half * 4 + n], 0, 0, 0);
                        acc[half * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gm_f16x8, wr[slot]), __builtin_bit_cast(gm_f16x8, xf2[g & 1][n]),
                                                                                   acc[half * 4 + n], 0, 0, 0);
                    }
                    if constexpr (F16)
                        acc[half * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gm_f16x8, wr[slot]), __builtin_bit_cast(gm_f16x8, xf[g & 1][n]),
                                                                                   acc[half * 4 + n], 0, 0, 0);
                    else acc[half * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[slot], xf[g & 1][n], acc[half * 4 + n], 0, 0, 0);
                }
                if (half == NG - 1) {  // the ring slot is free: request the fragment of k-step + D
                    __builtin_amdgcn_sched_barrier(0);
                    const int jn = c * KS + ks + D;
                    wr[slot] = *reinterpret_cast<const gm_bf16x8*>(wstream + int64_t(jn < nk ? jn : nk - 1) * 1024);
                    if constexpr (SP3) wr2[slot] = *reinterpret_cast<const gm_bf16x8*>(wstream2 + int64_t(jn < nk ? jn : nk - 1) * 1024);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // buffer hand-over: my LDS stores are done, nobody reads buffer P any more.  Raw barrier: __syncthreads() would also drain
        // vmcnt, i.e. the weight ring and the requested pieces
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int c = 0;
    if (active) {
        for (; c + 1 < nch; c += 2) {
            chunk(c, P0{}, std::true_type{});
            chunk(c + 1, P1{}, std::true_type{});
        }
        if (c < nch) chunk(c, P0{}, std::true_type{});
    } else {
        for (; c + 1 < nch; c += 2) {
            chunk(c, P0{}, std::false_type{});
            chunk(c + 1, P1{}, std::false_type{});
        }
        if (c < nch) chunk(c, P0{}, std::false_type{});
    }

    if constexpr (XF32) {
        if (xbad && a.flag) *a.flag = XM3D_ERANGE;
    }
    // ---- epilogue.  Accumulator register i of lane (token l31 of tile n, half h) = column 16 h + i of the wave's row block
    if constexpr (ACT == GM_ACT_GEGLU) {
        // rows 0 .. 15 of the block are value columns 16 gblk + i, rows 16 .. 31 their gate columns: half 0 holds values, half 1
        // gates.  Exchange so that half 0 finishes columns 0 .. 7 and half 1 columns 8 .. 15 of the 16 (one 16-byte store each)
        const int nout = a.N >> 1;
        const int col = gblk * 16 + 8 * h;
        if (active) {
            float bv[8], bg[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bv[i] = a.bias ? a.bias[col + i] : 0.f;
                bg[i] = a.bias ? a.bias[nout + col + i] : 0.f;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int row = row0 + (nbase + n) * 32 + l31;
                gm_bf16x8 pk;
                uint4 rr = make_uint4(0, 0, 0, 0);
                if (a.residual && row < M) rr = *reinterpret_cast<const uint4*>(a.residual + int64_t(row) * a.ldr + col);
                const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    gm_f32x2 val, gate;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // two STATIC register reads, laundered: `h ? acc[i] : acc[8 + i]` is otherwise canonicalised into one extract with a
                        // run-time index, which hipcc lowers to a 16-way compare / select cascade (~30 vector instructions per element:
                        // 1900 of them per tile row block, more than the whole K loop of the 320-channel layers)
                        float lo = acc[n][i + j], hi = acc[n][8 + i + j];
                        asm volatile("" : "+v"(lo), "+v"(hi));
                        const float send = h ? lo : hi;  // h = 1 sends gates 0..7, h = 0 sends values 8..15
                        const float recv = __shfl_xor(send, 32);
                        val[j] = (h ? recv : lo) + bv[i + j];
                        gate[j] = (h ? hi : recv) + bg[i + j];
                    }
                    const gm_f32x2 res = {__uint_as_float(rw[i >> 1] << 16), __uint_as_float(rw[i >> 1] & 0xFFFF0000u)};
                    const gm_f32x2 o = __builtin_elementwise_fma(val, gm_gelu2(gate), res);
                    pk[i] = (__bf16)o[0];
                    pk[i + 1] = (__bf16)o[1];
                }
                if (row < M) *reinterpret_cast<gm_bf16x8*>(a.out + int64_t(row) * a.ldo + col) = pk;
            }
        }
    } else if constexpr (OUT32) {
        // f32 rows: out = act(acc + bias + accin) + residual; split-K slices write their own slab (no bias / accin / residual there)
        const int col = gblk * 32 + 16 * h;
        if (active) {
            float* const outf = reinterpret_cast<float*>(a.out) + int64_t(blockIdx.y) * a.M * a.ldo;
            const float* const resf = reinterpret_cast<const float*>(a.residual);
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = a.bias ? *reinterpret_cast<const float4*>(a.bias + col + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int row = row0 + (nbase + n) * 32 + l31;
                if (row >= M) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {fmaf(acc[n][4 * q], a.alpha, bq[q].x), fmaf(acc[n][4 * q + 1], a.alpha, bq[q].y), fmaf(acc[n][4 * q + 2], a.alpha, bq[q].z),
                                  fmaf(acc[n][4 * q + 3], a.alpha, bq[q].w)};
                    if (a.accin) {
                        const float4 t = *reinterpret_cast<const float4*>(a.accin + int64_t(row) * a.ldo + col + 4 * q);
                        v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (ACT == GM_ACT_GELU) v[j] = gm_gelu(v[j]);
                        if constexpr (ACT == GM_ACT_QUICK_GELU)
                            v[j] = v[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v[j]));
                        if constexpr (ACT == GM_ACT_NONE) v[j] = fmaxf(v[j], a.floor);
                    }
                    if (resf) {
                        const float4 t = *reinterpret_cast<const float4*>(resf + int64_t(row) * a.ldr + col + 4 * q);
                        v[0] += t.x, v[1] += t.y, v[2] += t.z, v[3] += t.w;
                    }
                    *reinterpret_cast<float4*>(outf + int64_t(row) * a.ldo + col + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    } else {
        const int col = gblk * 32 + 16 * h;
        if (active) {
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = a.bias ? *reinterpret_cast<const float4*>(a.bias + col + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int row = row0 + (nbase + n) * 32 + l31;
                uint4 rr[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
                if (a.residual && row < M) {
                    rr[0] = *reinterpret_cast<const uint4*>(a.residual + int64_t(row) * a.ldr + col);
                    rr[1] = *reinterpret_cast<const uint4*>(a.residual + int64_t(row) * a.ldr + col + 8);
                }
                const unsigned rw[8] = {rr[0].x, rr[0].y, rr[0].z, rr[0].w, rr[1].x, rr[1].y, rr[1].z, rr[1].w};
                gm_bf16x8 pk[2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {acc[n][4 * q] + bq[q].x, acc[n][4 * q + 1] + bq[q].y, acc[n][4 * q + 2] + bq[q].z, acc[n][4 * q + 3] + bq[q].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (ACT == GM_ACT_GELU) v[j] = gm_gelu(v[j]);
                        if constexpr (ACT == GM_ACT_QUICK_GELU)
                            v[j] = v[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v[j]));
                        if constexpr (ACT == GM_ACT_NONE) v[j] = fmaxf(v[j], a.floor);
                    }
                    v[0] += __uint_as_float(rw[2 * q] << 16);
                    v[1] += __uint_as_float(rw[2 * q] & 0xFFFF0000u);
                    v[2] += __uint_as_float(rw[2 * q + 1] << 16);
                    v[3] += __uint_as_float(rw[2 * q + 1] & 0xFFFF0000u);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[q >> 1][(q & 1) * 4 + j] = (__bf16)v[j];
                }
                if (row < M) {
                    *reinterpret_cast<gm_bf16x8*>(a.out + int64_t(row) * a.ldo + col) = pk[0];
                    *reinterpret_cast<gm_bf16x8*>(a.out + int64_t(row) * a.ldo + col + 8) = pk[1];
                }
            }
        }
    }
}

// W (N, K) row-major, bf16 or f32 -> fragment order: [32-row block][k-step][lane][8 bf16]; MFMA row rho of a block = block row
// 16 ((rho >> 2) & 1) + (rho & 3) + 4 (rho >> 3) (so that a lane's accumulator registers are 16 consecutive block rows); block row
// -> row of W: plain: 32 blk + r; GEGLU: r < 16: value row 16 blk + r, else gate row N/2 + 16 blk + (r - 16).  Rows past N
// (padding of the last column tile) are zero.
template <typename T>
__global__ void k_gemm_pack(const T* __restrict__ w, int N, int K, int nblk, int geglu, __bf16* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one 16-byte output vector per thread
    const int nk = K / 16;
    if (i >= int64_t(nblk) * nk * 64) return;
    const int l = int(i & 63);
    const int j = int((i >> 6) % nk);
    const int blk = int((i >> 6) / nk);
    const int rho = l & 31;
    const int r = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
    int row;
    bool valid;
    if (geglu) {
        const int half = N >> 1;
        row = r < 16 ? 16 * blk + r : half + 16 * blk + (r - 16);
        valid = 16 * blk + (r & 15) < half;
    } else {
        row = 32 * blk + r;
        valid = row < N;
    }
    const int k = j * 16 + (l >> 5) * 8;
    gm_bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = valid ? (__bf16)float(w[int64_t(row) * K + k + e]) : (__bf16)0.f;
    reinterpret_cast<gm_bf16x8*>(out)[i] = o;
}

// out = act(sum_z slab[z] + bias) + residual, bf16 rows: the deterministic split-K's second launch (slices summed in index order)
__global__ void k_gemm_splitk_finish(const float* __restrict__ slab, int ksplit, int64_t M, int N, const float* __restrict__ bias, int act,
                                     const __bf16* __restrict__ residual, int64_t ldr, __bf16* __restrict__ out, int64_t ldo) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one 8-column vector
    const int n8 = N / 8;
    if (i >= M * n8) return;
    const int64_t row = i / n8;
    const int col = int(i - row * n8) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bias ? bias[col + j] : 0.f;
    for (int z = 0; z < ksplit; ++z) {
        const float* p = slab + (int64_t(z) * M + row) * N + col;
        const float4 t0 = *reinterpret_cast<const float4*>(p), t1 = *reinterpret_cast<const float4*>(p + 4);
        v[0] += t0.x, v[1] += t0.y, v[2] += t0.z, v[3] += t0.w, v[4] += t1.x, v[5] += t1.y, v[6] += t1.z, v[7] += t1.w;
    }
    gm_bf16x8 r8;
    if (residual) r8 = *reinterpret_cast<const gm_bf16x8*>(residual + row * ldr + col);
    gm_bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = v[j];
        if (act == GM_ACT_GELU) t = gm_gelu(t);
        else if (act == GM_ACT_QUICK_GELU) t = t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * t));
        else if (act == GM_ACT_RELU) t = fmaxf(t, 0.f);
        if (residual) t += float(r8[j]);
        o[j] = (__bf16)t;
    }
    *reinterpret_cast<gm_bf16x8*>(out + row * ldo + col) = o;
}

template <int CT, int ACT, int KC, int NW, int FL>
static int launch_gemm_kc(const GemmArgs& a0, int ksplit, hipStream_t s) {
    constexpr int MT = NW * 32;
    constexpr int LDS = 2 * ((FL & GF_SPLIT3) ? 2 : 1) * MT * (KC * 2 + 16);
    static DeviceOnce configured;  // the attribute is per device
    if (configured.first()) {
        XM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<CT, ACT, KC, NW, FL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    GemmArgs a = a0;
    const int nch = a.K / KC;
    a.chunks_per_split = (nch + ksplit - 1) / ksplit;
    const int slices = (nch + a.chunks_per_split - 1) / a.chunks_per_split;  // every slice non-empty
    const int grid = ((a.M + MT - 1) / MT) * a.nct;
    hipLaunchKernelGGL((k_gemm<CT, ACT, KC, NW, FL>), dim3(grid, slices), dim3(NW * 64), LDS, s, a);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

// 128-channel chunks (one barrier per 8 k-steps) whenever K (GF_CONV: Cin) allows; K = 320 and other odd multiples of 64 run 64-channel
// chunks.  waves: 8 / 4 (4: CT = 128 only)
template <int CT, int ACT, int FL>
static int launch_gemm(const GemmArgs& a, int waves, int ksplit, hipStream_t s) {
    if constexpr ((FL & GF_SPLIT3) != 0) {  // 128-column tiles, 64-channel chunks only
        if constexpr (CT == 128) return waves == 4 ? launch_gemm_kc<128, ACT, 64, 4, FL>(a, ksplit, s) : launch_gemm_kc<128, ACT, 64, 8, FL>(a, ksplit, s);
        else return XM3D_EINVAL;
    } else {
        const bool kc128 = ((FL & GF_CONV) ? a.Cin : a.K) % 128 == 0;
        if constexpr (CT == 128) {
            if (waves == 4) return kc128 ? launch_gemm_kc<CT, ACT, 128, 4, FL>(a, ksplit, s) : launch_gemm_kc<CT, ACT, 64, 4, FL>(a, ksplit, s);
        }
        return kc128 ? launch_gemm_kc<CT, ACT, 128, 8, FL>(a, ksplit, s) : launch_gemm_kc<CT, ACT, 64, 8, FL>(a, ksplit, s);
    }
}

template <int CT>
static int dispatch_gemm(const GemmArgs& a0, int act, int waves, hipStream_t s) {
    GemmArgs a = a0;
    if (act == GM_ACT_RELU) a.floor = 0.f, act = GM_ACT_NONE;
    switch (act) {
        case GM_ACT_NONE: return launch_gemm<CT, GM_ACT_NONE, 0>(a, waves, 1, s);
        case GM_ACT_GELU: return launch_gemm<CT, GM_ACT_GELU, 0>(a, waves, 1, s);
        case GM_ACT_QUICK_GELU: return launch_gemm<CT, GM_ACT_QUICK_GELU, 0>(a, waves, 1, s);
        default: return launch_gemm<CT, GM_ACT_GEGLU, 0>(a, waves, 1, s);
    }
}

// number of K slices of the deterministic split-K.  Measured (tools/conv_gemm_bench.py SWEEP=1, 20 views, profiles/r04_conv_gemm_bench.log):
// the slab round trip + second launch only pay when the grid has far fewer tiles than the chip has CUs AND K is long - the 8^2 UNet
// level (100 tiles of 128 x 128, K = 11520 / 23040: 63 vs 89 us, 110 vs 176 us with 2 - 4 slices); at 200 - 400 tiles one slice wins
// (unet res 1280 @16: 177 vs 205 us; 1x1 1280 @16: 28 vs 59 us)
static int choose_ksplit(int64_t tiles, int K) {
    if (tiles > 128 || K < 4096) return 1;
    return tiles > 64 ? 2 : 4;
}

}  // namespace xm3d

using namespace xm3d;

// column tile for N rows of W: 256 when that wastes nothing (or N is large), else 128; the packed image is padded to whole tiles
extern "C" int xm3d_gemm_col_tile(int n_rows) { return n_rows % 256 == 0 ? 256 : 128; }

// the packed image does not depend on the column tile (32-row blocks, padded to a multiple of 256 rows for either): a weight packed
// for tile 256 may be run with tile 128 (the 4-wave geometry needs 128)

extern "C" int64_t xm3d_gemm_packed_elems(int n_rows, int K, int col_tile) {
    const int64_t npad = (int64_t(n_rows) + col_tile - 1) / col_tile * col_tile;
    return npad * K;
}

extern "C" int xm3d_gemm_pack_weight(const void* w, int w_is_f32, int N, int K, int act, int col_tile, void* packed, void* stream) {
    XM3D_REQUIRE(w && packed, "gemm_pack_weight: null pointer");
    XM3D_REQUIRE(K > 0 && K % GM_KC == 0, "gemm_pack_weight: K %d is not a multiple of %d", K, GM_KC);
    XM3D_REQUIRE(col_tile == 128 || col_tile == 256, "gemm_pack_weight: column tile %d unsupported", col_tile);
    XM3D_REQUIRE(N > 0 && N % 32 == 0, "gemm_pack_weight: N %d is not a multiple of 32", N);
    XM3D_REQUIRE(act >= 0 && act <= 3, "gemm_pack_weight: unknown epilogue %d", act);
    const int geglu = act == GM_ACT_GEGLU;
    const int nblk = int((int64_t(N) + col_tile - 1) / col_tile) * (col_tile / 32);
    const int64_t total = int64_t(nblk) * (K / 16) * 64;
    if (w_is_f32)
        hipLaunchKernelGGL(k_gemm_pack<float>, dim3(unsigned((total + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const float*>(w), N, K,
                           nblk, geglu, static_cast<__bf16*>(packed));
    else
        hipLaunchKernelGGL(k_gemm_pack<__bf16>, dim3(unsigned((total + 255) / 256)), dim3(256), 0, as_stream(stream), static_cast<const __bf16*>(w), N,
                           K, nblk, geglu, static_cast<__bf16*>(packed));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

// workgroup geometry when the caller does not choose (waves = 0): 128-row, 4-wave workgroups (column tile 128 only) when the 256-row
// grid would not give every CU two workgroups' worth of tiles
extern "C" int xm3d_gemm_default_waves(int64_t M, int N, int col_tile) {
    if (col_tile != 128) return 8;
    const int64_t grid8 = ((M + 255) / 256) * ((N + col_tile - 1) / col_tile);
    return grid8 < 512 ? 4 : 8;
}

extern "C" int xm3d_gemm_bf16(const void* x, int64_t M, int K, int64_t ldx, const void* wpacked, int N, int col_tile, const float* bias, int act,
                              const void* residual, int64_t ldr, void* out, int64_t ldo, int waves, void* stream) {
    XM3D_REQUIRE(x && wpacked && out, "gemm_bf16: null pointer");
    XM3D_REQUIRE(M > 0 && M < (int64_t(1) << 31) - GM_MT && M * ldx < (int64_t(1) << 31), "gemm_bf16: M (x ldx) out of range (32-bit offsets)");
    XM3D_REQUIRE(K > 0 && K % GM_KC == 0, "gemm_bf16: K %d is not a multiple of %d", K, GM_KC);
    XM3D_REQUIRE(col_tile == 128 || col_tile == 256, "gemm_bf16: column tile %d unsupported", col_tile);
    XM3D_REQUIRE(N > 0 && N % 32 == 0, "gemm_bf16: N %d is not a multiple of 32", N);
    XM3D_REQUIRE(act >= 0 && act <= 4, "gemm_bf16: unknown epilogue %d", act);
    const int nout = act == GM_ACT_GEGLU ? N / 2 : N;
    XM3D_REQUIRE(ldx >= K && ldo >= nout && (!residual || ldr >= nout) && ldx % 8 == 0 && ldo % 8 == 0 && ldr % 8 == 0 &&
                     ldx < (int64_t(1) << 31) && ldo < (int64_t(1) << 31) && ldr < (int64_t(1) << 31),
                 "gemm_bf16: row strides must cover the row and be multiples of 8 elements");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                 "gemm_bf16: tensors must be 16-byte aligned");
    GemmArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.wp = static_cast<const __bf16*>(wpacked);
    a.bias = bias;
    a.residual = static_cast<const __bf16*>(residual);
    a.out = static_cast<__bf16*>(out);
    a.accin = nullptr;
    a.x2 = a.wp2 = nullptr;
    a.alpha = 1.f;
    a.M = int(M), a.K = K, a.N = N;
    a.ldx = int(ldx), a.ldr = int(ldr), a.ldo = int(ldo);
    a.nct = (N + col_tile - 1) / col_tile;
    a.Hin = a.Win = a.Cin = a.Ho = a.Wo = a.stride = a.pad_t = a.pad_l = a.ksz = 0;
    XM3D_REQUIRE(((M + 127) / 128) * a.nct < (int64_t(1) << 31), "gemm_bf16: grid too large");
    XM3D_REQUIRE(waves == 0 || waves == 8 || (waves == 4 && col_tile == 128), "gemm_bf16: waves must be 0 (choose), 8, or 4 with column tile 128");
    if (waves == 0) waves = xm3d_gemm_default_waves(M, N, col_tile);
    return col_tile == 256 ? dispatch_gemm<256>(a, act, waves, as_stream(stream)) : dispatch_gemm<128>(a, act, waves, as_stream(stream));
}

// ---- implicit-GEMM convolution (GF_CONV): out (B, Ho, Wo, N) = conv(x (B, Hin, Win, Cin), W) + bias (+ residual), channels-last bf16, for the
// convolutions conv.hip's 32-pixel-wide halo tiles do not take - strided Downsample, the 16^2 / 8^2 UNet levels, 1x1 with large K.
// Small grids run a DETERMINISTIC split-K: K slices write f32 partial slabs, a second launch adds them in slice order (the library's
// split-K convolutions add with atomics: the only irreproducible kernels the forward had, tools/find_nondeterminism.py).
static void conv_gemm_plan(int64_t M, int N, int K, int col_tile, int* tile, int* waves, int* ksplit) {
    int t = col_tile;
    if (t == 256 && ((M + 255) / 256) * ((N + 255) / 256) < 200) t = 128;
    const int w = t == 128 ? xm3d_gemm_default_waves(M, N, 128) : 8;
    const int64_t tiles = ((M + w * 32 - 1) / (w * 32)) * ((N + t - 1) / t);
    *tile = t, *waves = w, *ksplit = choose_ksplit(tiles, K);
    if (const char* e = getenv("XM3D_CONV_GEMM_KSPLIT")) {  // tuning aid (tools/conv_gemm_bench.py): force the number of K slices
        const int f = atoi(e);
        if (f >= 1 && f <= 64 && K / f >= 64) *ksplit = f;
    }
}

extern "C" int64_t xm3d_conv_gemm_ws_bytes(int64_t M, int32_t N, int32_t K, int32_t col_tile) {
    if (M <= 0 || N <= 0 || K <= 0 || (col_tile != 128 && col_tile != 256)) return 0;
    int t, w, ks;
    conv_gemm_plan(M, N, K, col_tile, &t, &w, &ks);
    return ks > 1 ? int64_t(ks) * M * N * int64_t(sizeof(float)) : 0;
}

extern "C" int xm3d_conv_gemm_bf16(const void* x, int64_t B, int32_t Hin, int32_t Win, int32_t Cin, const void* wpacked, int32_t N, int32_t col_tile,
                                   int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, const float* bias,
                                   const void* residual, void* out, void* ws, void* stream) {
    XM3D_REQUIRE(x && wpacked && out, "conv_gemm_bf16: null pointer");
    XM3D_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cin % GM_KC == 0, "conv_gemm_bf16: bad shape (Cin %d must be a multiple of %d)", Cin, GM_KC);
    XM3D_REQUIRE(ksize >= 1 && ksize <= 3 && stride >= 1 && pad_t >= 0 && pad_l >= 0 && pad_t < ksize && pad_l < ksize, "conv_gemm_bf16: kernel size 1..3, padding < kernel size");
    XM3D_REQUIRE((Ho - 1) * stride - pad_t < Hin && (Wo - 1) * stride - pad_l < Win, "conv_gemm_bf16: output %dx%d reaches outside the input", Ho, Wo);
    XM3D_REQUIRE(col_tile == 128 || col_tile == 256, "conv_gemm_bf16: column tile %d unsupported", col_tile);
    XM3D_REQUIRE(N > 0 && N % 32 == 0, "conv_gemm_bf16: N %d is not a multiple of 32", N);
    const int64_t M = B * Ho * Wo;
    XM3D_REQUIRE(M < (int64_t(1) << 31) - GM_MT && B * Hin * Win * Cin < (int64_t(1) << 31) && M * N < (int64_t(1) << 31), "conv_gemm_bf16: tensor too large for 32-bit offsets");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
                 "conv_gemm_bf16: tensors must be 16-byte aligned");
    GemmArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.wp = static_cast<const __bf16*>(wpacked);
    a.accin = nullptr;
    a.x2 = a.wp2 = nullptr;
    a.alpha = 1.f;
    a.M = int(M), a.K = ksize * ksize * Cin, a.N = N;
    a.ldx = 0, a.ldr = N, a.ldo = N;
    a.Hin = Hin, a.Win = Win, a.Cin = Cin, a.Ho = Ho, a.Wo = Wo, a.stride = stride, a.pad_t = pad_t, a.pad_l = pad_l, a.ksz = ksize;
    int tile, waves, ksplit;
    conv_gemm_plan(M, N, a.K, col_tile, &tile, &waves, &ksplit);
    a.nct = (N + tile - 1) / tile;
    hipStream_t s = as_stream(stream);
    if (ksplit == 1) {
        a.bias = bias;
        a.residual = static_cast<const __bf16*>(residual);
        a.out = static_cast<__bf16*>(out);
        return tile == 256 ? launch_gemm<256, GM_ACT_NONE, GF_CONV>(a, waves, 1, s) : launch_gemm<128, GM_ACT_NONE, GF_CONV>(a, waves, 1, s);
    }
    XM3D_REQUIRE(ws, "conv_gemm_bf16: this shape runs split-K and needs xm3d_conv_gemm_ws_bytes(M, N, K, col_tile) bytes of workspace");
    a.bias = nullptr;
    a.residual = nullptr;
    a.out = static_cast<__bf16*>(ws);
    const int rc = tile == 256 ? launch_gemm<256, GM_ACT_NONE, GF_CONV | GF_OUT32>(a, waves, ksplit, s) : launch_gemm<128, GM_ACT_NONE, GF_CONV | GF_OUT32>(a, waves, ksplit, s);
    if (rc != XM3D_OK) return rc;
    const int KC = a.Cin % 128 == 0 ? 128 : 64, nch = a.K / KC, cps = (nch + ksplit - 1) / ksplit, slices = (nch + cps - 1) / cps;
    const int64_t nvec = M * (N / 8);
    hipLaunchKernelGGL(k_gemm_splitk_finish, dim3(unsigned((nvec + 255) / 256)), dim3(256), 0, s, static_cast<const float*>(ws), slices, M, N, bias, 0,
                       static_cast<const __bf16*>(residual), int64_t(N), static_cast<__bf16*>(out), int64_t(N));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

// ---- f32-accurate GEMM / implicit-GEMM convolution from matrix-core passes over operands split in IEEE halves (GF_F16 | GF_OUT32):
//   x = xhi / s + xlo / (2048 s), w = whi + wlo / 2048   ->   x w = [xhi whi] / s + [xhi wlo + xlo whi] / (2048 s)  (+ 2^-22 |x w|)
// ONE pass: out (f32) = act(alpha * (x_term @ W_term^T) + bias + accin) + residual.  The caller runs the three term pairs with
// accin = out (xmask3d_amd.ops.gemm_f32 / conv_gemm_f32).  x: (M, K) halves, row stride ldx (conv = 0), or the channels-last image
// (B, Hin, Win, Cin) in halves (conv = 1: geometry as xm3d_conv_gemm_bf16, M = B * Ho * Wo, no split-K).
static int gemm_f32acc_impl(const void* x, const void* x_lo, int64_t M, int32_t K, int64_t ldx, const void* wpacked, const void* wpacked_lo, int32_t N,
                           int32_t col_tile, const float* bias, int32_t act, float alpha, const float* accin, const float* residual, int64_t ldr, float* out,
                           int64_t ldo, int32_t waves, int32_t conv, int64_t B, int32_t Hin, int32_t Win, int32_t Cin, int32_t ksize, int32_t stride,
                           int32_t pad_t, int32_t pad_l, int32_t Ho, int32_t Wo, void* stream, float x_f32_scale = 0.f) {
    XM3D_REQUIRE(x && wpacked && out, "gemm_f32acc: null pointer");
    const bool xf32 = x_f32_scale > 0.f;  // x is the f32 tensor, split while it is staged (GF_XF32)
    if (xf32) x_lo = x;                   // (the one-launch form; there is no second plane in memory)
    const bool fused = x_lo != nullptr;
    XM3D_REQUIRE(!fused || (wpacked_lo && !accin && ((reinterpret_cast<uintptr_t>(x_lo) | reinterpret_cast<uintptr_t>(wpacked_lo)) & 15) == 0),
                 "gemm_f32: the one-launch form needs both planes of x and both weight images (16-byte aligned), and takes no accin");
    XM3D_REQUIRE(col_tile == 128 || col_tile == 256, "gemm_f32acc: column tile %d unsupported", col_tile);
    XM3D_REQUIRE(N > 0 && N % 32 == 0, "gemm_f32acc: N %d is not a multiple of 32", N);
    XM3D_REQUIRE((act >= 0 && act <= 2) || act == GM_ACT_RELU, "gemm_f32acc: epilogue %d unsupported (none, GELU, QuickGELU, ReLU)", act);
    GemmArgs a;
    if (act == GM_ACT_RELU) a.floor = 0.f, act = GM_ACT_NONE;
    if (conv) {
        XM3D_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cin % GM_KC == 0, "gemm_f32acc: bad convolution shape (Cin %d)", Cin);
        XM3D_REQUIRE(ksize >= 1 && ksize <= 3 && stride >= 1 && pad_t >= 0 && pad_l >= 0 && pad_t < ksize && pad_l < ksize, "gemm_f32acc: kernel size 1..3, padding < kernel size");
        XM3D_REQUIRE((Ho - 1) * stride - pad_t < Hin && (Wo - 1) * stride - pad_l < Win, "gemm_f32acc: output %dx%d reaches outside the input", Ho, Wo);
        XM3D_REQUIRE(M == B * Ho * Wo && K == ksize * ksize * Cin && B * Hin * Win * Cin < (int64_t(1) << 31), "gemm_f32acc: M / K do not match the convolution");
        XM3D_REQUIRE(act == 0, "gemm_f32acc: the convolution form has no activation");
    } else {
        XM3D_REQUIRE(K > 0 && K % GM_KC == 0 && ldx >= K && ldx % 8 == 0 && M * ldx < (int64_t(1) << 31), "gemm_f32acc: K %d / ldx unsupported", K);
    }
    XM3D_REQUIRE(M > 0 && M < (int64_t(1) << 31) - GM_MT && M * ldo < (int64_t(1) << 31), "gemm_f32acc: M out of range (32-bit offsets)");
    XM3D_REQUIRE(ldo >= N && ldo % 4 == 0 && (!residual || (ldr >= N && ldr % 4 == 0)), "gemm_f32acc: row strides must cover the row (multiples of 4)");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(accin) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                 "gemm_f32acc: tensors must be 16-byte aligned");
    a.x = static_cast<const __bf16*>(x);
    a.wp = static_cast<const __bf16*>(wpacked);
    a.bias = bias;
    a.residual = reinterpret_cast<const __bf16*>(residual);
    a.out = reinterpret_cast<__bf16*>(out);
    a.accin = accin;
    a.x2 = static_cast<const __bf16*>(x_lo);
    a.wp2 = static_cast<const __bf16*>(wpacked_lo);
    a.alpha = alpha;
    a.M = int(M), a.K = K, a.N = N;
    a.ldx = int(ldx), a.ldr = int(ldr), a.ldo = int(ldo);
    a.Hin = Hin, a.Win = Win, a.Cin = Cin, a.Ho = Ho, a.Wo = Wo, a.stride = stride, a.pad_t = pad_t, a.pad_l = pad_l, a.ksz = ksize;
    int tile = col_tile;
    if (fused) tile = 128;
    if (tile == 256 && ((M + 255) / 256) * ((N + 255) / 256) < 200) tile = 128;
    XM3D_REQUIRE(waves == 0 || waves == 8 || (waves == 4 && tile == 128), "gemm_f32acc: waves must be 0 (choose), 8, or 4 with column tile 128");
    if (waves == 0) waves = tile == 128 ? xm3d_gemm_default_waves(M, N, 128) : 8;
    a.nct = (N + tile - 1) / tile;
    hipStream_t s = as_stream(stream);
    constexpr int FG = GF_F16 | GF_OUT32, FC = GF_F16 | GF_OUT32 | GF_CONV;
    if (xf32) {
        constexpr int XG = FG | GF_SPLIT3 | GF_XF32, XC = FC | GF_SPLIT3 | GF_XF32;
        a.xscale = x_f32_scale;
        a.flag = device_flag();
        if (conv) return launch_gemm<128, GM_ACT_NONE, XC>(a, waves, 1, s);
        if (act == GM_ACT_GELU) return launch_gemm<128, GM_ACT_GELU, XG>(a, waves, 1, s);
        if (act == GM_ACT_QUICK_GELU) return launch_gemm<128, GM_ACT_QUICK_GELU, XG>(a, waves, 1, s);
        return launch_gemm<128, GM_ACT_NONE, XG>(a, waves, 1, s);
    }
    if (fused) {
        constexpr int SG = FG | GF_SPLIT3, SC = FC | GF_SPLIT3;
        if (conv) return launch_gemm<128, GM_ACT_NONE, SC>(a, waves, 1, s);
        if (act == GM_ACT_GELU) return launch_gemm<128, GM_ACT_GELU, SG>(a, waves, 1, s);
        if (act == GM_ACT_QUICK_GELU) return launch_gemm<128, GM_ACT_QUICK_GELU, SG>(a, waves, 1, s);
        return launch_gemm<128, GM_ACT_NONE, SG>(a, waves, 1, s);
    }
    if (conv) return tile == 256 ? launch_gemm<256, GM_ACT_NONE, FC>(a, waves, 1, s) : launch_gemm<128, GM_ACT_NONE, FC>(a, waves, 1, s);
    if (tile == 256) {
        if (act == GM_ACT_GELU) return launch_gemm<256, GM_ACT_GELU, FG>(a, waves, 1, s);
        if (act == GM_ACT_QUICK_GELU) return launch_gemm<256, GM_ACT_QUICK_GELU, FG>(a, waves, 1, s);
        return launch_gemm<256, GM_ACT_NONE, FG>(a, waves, 1, s);
    }
    if (act == GM_ACT_GELU) return launch_gemm<128, GM_ACT_GELU, FG>(a, waves, 1, s);
    if (act == GM_ACT_QUICK_GELU) return launch_gemm<128, GM_ACT_QUICK_GELU, FG>(a, waves, 1, s);
    return launch_gemm<128, GM_ACT_NONE, FG>(a, waves, 1, s);
}

extern "C" int xm3d_gemm_f32acc(const void* x, int64_t M, int32_t K, int64_t ldx, const void* wpacked, int32_t N, int32_t col_tile, const float* bias,
                                int32_t act, float alpha, const float* accin, const float* residual, int64_t ldr, float* out, int64_t ldo, int32_t waves,
                                int32_t conv, int64_t B, int32_t Hin, int32_t Win, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t,
                                int32_t pad_l, int32_t Ho, int32_t Wo, void* stream) {
    return gemm_f32acc_impl(x, nullptr, M, K, ldx, wpacked, nullptr, N, col_tile, bias, act, alpha, accin, residual, ldr, out, ldo, waves, conv, B, Hin, Win, Cin,
                            ksize, stride, pad_t, pad_l, Ho, Wo, stream);
}

// The whole f32-accurate product in ONE launch (GF_SPLIT3): x_hi / x_lo = the two half planes of x AT ONE SCALE (x s = hi + lo: the split with
// lo_mul = 1), wp_hi / wp_lo the packed images of the two half terms of W t; alpha = 1 / (s t).
//     out (f32) = act( alpha * (x_hi W_hi^T + x_hi W_lo^T + x_lo W_hi^T) + bias ) + residual
// Same shapes / convolution form as xm3d_gemm_f32acc; 128-column tiles.
extern "C" int xm3d_gemm_f32(const void* x_hi, const void* x_lo, int64_t M, int32_t K, int64_t ldx, const void* wp_hi, const void* wp_lo, int32_t N,
                             const float* bias, int32_t act, float alpha, const float* residual, int64_t ldr, float* out, int64_t ldo, int32_t waves, int32_t conv,
                             int64_t B, int32_t Hin, int32_t Win, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l, int32_t Ho,
                             int32_t Wo, void* stream) {
    XM3D_REQUIRE(x_lo && wp_lo, "gemm_f32: null pointer");
    return gemm_f32acc_impl(x_hi, x_lo, M, K, ldx, wp_hi, wp_lo, N, 128, bias, act, alpha, nullptr, residual, ldr, out, ldo, waves, conv, B, Hin, Win, Cin, ksize,
                            stride, pad_t, pad_l, Ho, Wo, stream);
}

// xm3d_gemm_f32 straight from the F32 activation: x (M, K) f32 rows (conv = 1: the channels-last f32 image) is split into its half planes
// x x_scale = hi + lo while it is staged (GF_XF32) - no xm3d_split_f16t_nhwc pass, no half planes in memory; results are bit-identical to
// that pass followed by xm3d_gemm_f32.  x_scale: a power of two; |x| x_scale beyond 65504 raises the sticky range flag (xm3d_check_flag).
// alpha = 1 / (x_scale t) with t the weight scale of the packed images.  ldx in floats.
extern "C" int xm3d_gemm_f32x(const float* x, float x_scale, int64_t M, int32_t K, int64_t ldx, const void* wp_hi, const void* wp_lo, int32_t N,
                              const float* bias, int32_t act, float alpha, const float* residual, int64_t ldr, float* out, int64_t ldo, int32_t waves,
                              int32_t conv, int64_t B, int32_t Hin, int32_t Win, int32_t Cin, int32_t ksize, int32_t stride, int32_t pad_t, int32_t pad_l,
                              int32_t Ho, int32_t Wo, void* stream) {
    XM3D_REQUIRE(wp_lo, "gemm_f32x: null pointer");
    XM3D_REQUIRE(x_scale > 0.f, "gemm_f32x: x_scale must be a positive power of two");
    return gemm_f32acc_impl(x, nullptr, M, K, ldx, wp_hi, wp_lo, N, 128, bias, act, alpha, nullptr, residual, ldr, out, ldo, waves, conv, B, Hin, Win, Cin, ksize,
                            stride, pad_t, pad_l, Ho, Wo, stream, x_scale);
}
