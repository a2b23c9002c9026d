// 3x3 convolution (stride 1, zero padding 1) over channels-last bf16 activations on the gfx950 matrix cores, with the
// GroupNorm that precedes it and the GroupNorm that follows it folded in:
//
//     out = conv3x3( act( GN(x) ) ) + bias (+ residual)          statistics of `out` accumulated on the way
//
// This is the ResnetBlock body of ldm's VAE / UNet (GroupNorm(32) -> SiLU -> Conv2d(3x3), twice, + skip), reached from
// /root/reference/models/modeling/meta_arch/ldm.py:386-414 (VAE encoder), :425-446 (UNet), :448-490 (VAE decoder): the FLOP
// majority of the whole scene.  It replaces, per convolution, {GroupNorm statistics pass, GroupNorm apply + SiLU pass, library
// implicit-GEMM convolution, bias / residual pass} by ONE launch that reads x once (+ 33 % halo) and writes out once.
//
// Decomposition (one workgroup = 8 waves = one 8 x 32 pixel tile of one image x CT output channels):
//   * K loop = input-channel chunks of 64 (outer) x the 9 filter taps (inner) x 4 MFMA k-steps of 16 channels.
//   * ACTIVATIONS: per chunk the (8+2) x (32+2) HALO tile of the input is staged ONCE in LDS - raw bf16 from global memory into
//     registers, normalised with the per-(image, channel) affine derived from the f64 GroupNorm moments, SiLU, rounded to bf16,
//     written with a 144-byte pixel stride - and all 9 taps read it with the same per-lane address plus an immediate: a tap is
//     a shift of the pixel index, and the padded stride makes the 32-pixel fragment reads (ds_read_b128) bank-conflict free at
//     ANY shift.  The normalisation is therefore done once per element (x 1.33 halo), not 9 times, and the activation operand
//     costs 1/9 of an im2col-style staging.  Two halo buffers: the tile of chunk c+1 is staged during the taps of chunk c (one
//     16-byte piece per thread and stage, requested one stage before it is normalised), ONE workgroup barrier per chunk.
//   * WEIGHTS never touch LDS: every wave owns 32 output channels (rows) of the tile and streams exactly its own MFMA A
//     fragments from a pre-packed, fragment-ordered image (xm3d_conv3x3_pack_weight) through L2 straight into registers, one
//     global_load_dwordx4 per k-step, six k-steps ahead (register ring).  No wave ever waits for another wave's loads, so there
//     is no per-stage barrier and no lockstep: the two waves of a SIMD drift apart and one's normalisation VALU work and
//     fragment reads run under the other's MFMAs.  (The first version shared a CT x 64 weight tile per stage through LDS-DMA
//     with a barrier per stage: 47 % of the time the matrix pipe sat idle behind {barrier, DMA issue, first fragment reads}.)
//   * MFMA v_mfma_f32_32x32x16_bf16 with the WEIGHTS as A operand (rows = output channels) and the pixels as B operand:
//     the accumulator has the pixel on the lane and 4 consecutive output channels per register quad = 8-byte bf16 stores,
//     and the GroupNorm statistics of the result are in-lane sums + one 32-lane reduction per channel quad.
//   * wave tile: CT = 256: 32 channels x all 256 pixels (8 accumulator tiles); CT = 128: 32 channels x 128 pixels (waves 4 x 2).
// Bound: MFMA (bf16).  Algorithmic FLOP = 2 * B*H*W * 9*Cin * Cout; bytes = B*H*W*(Cin + Cout [+ Cout residual])*2 + weights.
#include <type_traits>

#include "common.h"

namespace xm3d {

typedef float cv_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cv_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cv_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cv_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned cv_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 cv_f16x8 __attribute__((ext_vector_type(8)));

constexpr int CV_TW = 32;                            // output pixel tile: CV_TW columns x 8 or 4 rows of one image (ConvGeom)
constexpr int CV_HW = CV_TW + 2;                     // halo tile width
constexpr int CV_PSTR = 144;                         // bytes per halo pixel: 64 channels bf16 + 16 pad
constexpr int CV_KC = 64;                            // input channels per chunk
constexpr float CV_LOG2E = 1.4426950408889634f;
#ifndef CV_ABL
#define CV_ABL 0  // timing-only ablations (tools/conv_ablate.sh): 1 no weight ring refill, 2 no MFMA, 4 no fragment reads, 8 no chunk barrier, 16 no halo staging
#endif
#ifndef CV_WRING
#define CV_WRING 6  // weight fragments in flight per wave (k-steps); must divide 36
#endif
#ifndef CV_WRITE_KS
#define CV_WRITE_KS 1  // k-step of a stage in which the halo piece requested one stage earlier is normalised and stored
#endif

struct ConvArgs {
    const __bf16* x;          // (B, H>>ups, W>>ups, cin)
    const __bf16* wp;         // packed weights (xm3d_conv3x3_pack_weight)
    const float* affine;      // (B, cin / 8, 2, 8) GroupNorm scale / shift per (image, channel), or null (k_gn_affine)
    const float* bias;        // (cout) or (B, cout) with bias_bstride = cout, or null
    const __bf16* residual;   // (B, H, W, cout) or null
    __bf16* out;              // (B, H, W, cout)
    float2* stats_part;       // per-workgroup (sum, sum of squares) partials [b][cout tile][slot][pixel tile], or null (k_conv_stats_reduce)
    int B, H, W, cin, cout;
    int bias_bstride;
    float alpha;              // F32OUT: out = residual + alpha * conv + bias (the power-of-two scale of a split term; 1 otherwise)
    int groups_out, cg_out, stat_slots;
    int tiles_x, tiles_y, nct;
};

// Geometry of a workgroup.  NW = 8: 512 threads, 8 x 32 pixel tile, one workgroup per CU (98 KB of LDS), each wave one 32-channel
// row block.  NW = 4: 256 threads, 4 x 32 pixel tile, TWO workgroups per CU (59 KB each): the prologue (first halo tile) and the
// epilogue (residual read, output write - HBM-bound phases without matrix work) of one overlap the main loop of the other; each
// wave owns CT / 128 row blocks x all 4 pixel rows.
template <int CT, int NW>
struct ConvGeom {
    static constexpr int TH = NW == 8 ? 8 : 4;                         // pixel rows of the tile
    static constexpr int HPIX = (TH + 2) * CV_HW;                      // halo pixels
    static constexpr int ASZ = HPIX * CV_PSTR;                         // bytes per halo buffer
    static constexpr int NTH = NW * 64;                                // threads
    static constexpr int PR = NTH / 8;                                 // pixels staged per round
    static constexpr int ROUNDS = (HPIX + PR - 1) / PR;                // 16-byte pieces per thread and chunk (last round partial)
    static constexpr int MB = NW == 8 ? 1 : CT / 128;                  // 32-channel row blocks per wave
    static constexpr int NT = NW == 8 ? (CT == 256 ? 8 : 4) : 4;       // 32-pixel tiles (rows of the pixel tile) per wave
    static constexpr int NG = MB * NT / 4;                             // groups of 4 MFMAs per k-step
    static constexpr int D = MB == 2 ? 4 : CV_WRING;                   // weight ring depth (k-steps in flight per row block)
    static constexpr int LDS = 2 * ASZ + 16;
    static_assert(ROUNDS <= 8 && 36 % D == 0, "staging schedule / ring depth");
};

// MODE 0: x is the operand as it stands (plain convolution); MODE 2: operand = SiLU(GroupNorm(x)); MODE 1: ReLU(GroupNorm(x)) (the
// 3x3 convolution of detectron2's GroupNorm BottleneckBlock in the projection backbone, backbone/feature_extractor.py:20-60).
// UPS: x has half the resolution, the operand is its nearest-neighbour 2x upsampling (ldm's Upsample -> conv).
// F32OUT (plain convolutions only): `out` and `residual` are f32 - the accumulating form used by the f32-accurate convolution
// (passes over split operands, xm3d_conv3x3_nhwc_f32acc).  F16 (F32OUT only): the operands are IEEE halves (v_mfma_f32_32x32x16_f16):
// a two-term split in halves carries 22 mantissa bits, the same as three bf16 terms, in half the passes
template <int CT, int MODE, bool UPS, int NW, bool F32OUT, bool F16 = false>
__device__ __forceinline__ void conv3x3_body(const ConvArgs& a) {
    static_assert(!F16 || (F32OUT && MODE == 0), "half operands: accumulating plain convolution only");
    using G = ConvGeom<CT, NW>;
    constexpr int TH = G::TH, HPIX = G::HPIX, ASZ = G::ASZ, PR = G::PR, ROUNDS = G::ROUNDS, MB = G::MB, NT = G::NT, NG = G::NG, D = G::D;
    constexpr int NGRP = 36 * NG;          // groups per chunk
    constexpr int ROWB = CV_HW * CV_PSTR;  // bytes per halo row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const abuf = smem;  // 2 halo buffers, then a 16-byte dump slot

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb0 = NW == 8 ? (CT == 256 ? wave : wave >> 1) : wave * MB;  // first 32-channel row block of this wave
    const int nbase = (NW == 8 && CT == 128) ? 4 * (wave & 1) : 0;          // first pixel row of this wave
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order (speed only): blocks i and i + 8 share an XCD's L2 - give each XCD a contiguous run of tiles
    int bid;
    {
        const int n = gridDim.x, i = blockIdx.x, xcd = i & 7, qd = n >> 3, r = n & 7;
        bid = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (i >> 3);
    }
    const int ct = bid % a.nct;
    int t_ = bid / a.nct;
    const int tx = t_ % a.tiles_x;
    t_ /= a.tiles_x;
    const int ty = t_ % a.tiles_y;
    const int b = t_ / a.tiles_y;

    const int H = a.H, W = a.W, cin = a.cin;
    const int Hi = UPS ? H >> 1 : H, Wi = UPS ? W >> 1 : W;
    const __bf16* const xb = a.x + int64_t(b) * Hi * Wi * cin;
    const int nch = cin / CV_KC;
    const int nk = nch * 36;  // k-steps

    // ---- weight streams of this wave: per row block 1 KiB (64 lanes x 16 bytes = one A fragment) per k-step, contiguous
    const char* const wstream = reinterpret_cast<const char*>(a.wp) + (int64_t(ct) * (CT / 32) + wb0) * nk * 1024 + lane * 16;
    cv_bf16x8 wr[D][MB];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int m = 0; m < MB; ++m)
            wr[i][m] = *reinterpret_cast<const cv_bf16x8*>(wstream + (int64_t(m) * nk + (i < nk ? i : nk - 1)) * 1024);

    // ---- halo staging: thread owns 16-byte piece (pixel prow + PR r, channels 8 kc .. 8 kc + 7) of every round r
    const int kc = tid & 7, prow = tid >> 3;
    int aoff[ROUNDS];  // element offset of the piece's source inside the image; -1: outside the image (zero padding)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int p = prow + PR * r;
        const int hy = p / CV_HW, hx = p - hy * CV_HW;
        const int gy = ty * TH - 1 + hy, gx = tx * CV_TW - 1 + hx;
        const bool inb = p < HPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int sy = UPS ? gy >> 1 : gy, sx = UPS ? gx >> 1 : gx;
        aoff[r] = inb ? (sy * Wi + sx) * cin + kc * 8 : -1;
    }
    const unsigned a_wr = unsigned(prow) * CV_PSTR + kc * 16;  // + r * PR * CV_PSTR

    // GroupNorm affine of the chunk being staged, y = x * sc + sh: per-(image, channel) table written by k_gn_affine just before
    // this launch.  Loaded for the chunk after next in stage 8, when the last piece of the next one has been normalised.
    float sc[8], sh[8];
    const float* const aff = MODE != 0 ? a.affine + (int64_t(b) * cin + kc * 8) * 2 : nullptr;
    auto gn_coeffs = [&](int c0) __attribute__((always_inline)) {
        if constexpr (MODE != 0) {
            const float4 s0 = *reinterpret_cast<const float4*>(aff + c0 * 2), s1 = *reinterpret_cast<const float4*>(aff + c0 * 2 + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(aff + c0 * 2 + 8), h1 = *reinterpret_cast<const float4*>(aff + c0 * 2 + 12);
            sc[0] = s0.x, sc[1] = s0.y, sc[2] = s0.z, sc[3] = s0.w, sc[4] = s1.x, sc[5] = s1.y, sc[6] = s1.z, sc[7] = s1.w;
            sh[0] = h0.x, sh[1] = h0.y, sh[2] = h0.z, sh[3] = h0.w, sh[4] = h1.x, sh[5] = h1.y, sh[6] = h1.z, sh[7] = h1.w;
        }
    };
    // branch free (a chunk's 36 k-steps are one basic block): a piece outside the image reads the image's first pixel and is
    // zeroed on the way to LDS; the threads without a piece in the last, partial round write to a dump slot
    auto a_load = [&](int r, int c0) __attribute__((always_inline)) -> cv_u32x4 {
        return *reinterpret_cast<const cv_u32x4*>(xb + (aoff[r] >= 0 ? aoff[r] : kc * 8) + c0);
    };
    // normalise + SiLU channel pair i of a piece in place (two bf16 in -> two bf16 out in the same register)
    auto a_norm_pair = [&](cv_u32x4& raw, int i) __attribute__((always_inline)) {
        if constexpr (MODE != 0) {
            float y0 = fmaf(__uint_as_float(raw[i] << 16), sc[2 * i], sh[2 * i]);
            float y1 = fmaf(__uint_as_float(raw[i] & 0xFFFF0000u), sc[2 * i + 1], sh[2 * i + 1]);
            if constexpr (MODE == 2) {  // SiLU
                y0 = y0 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-CV_LOG2E * y0));
                y1 = y1 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-CV_LOG2E * y1));
            } else {  // ReLU
                y0 = fmaxf(y0, 0.f);
                y1 = fmaxf(y1, 0.f);
            }
            cv_bf16x2 pk;
            pk[0] = (__bf16)y0;
            pk[1] = (__bf16)y1;
            raw[i] = __builtin_bit_cast(unsigned, pk);
        }
    };
    auto a_store = [&](int r, cv_u32x4 o, char* dst) __attribute__((always_inline)) {
        if (aoff[r] < 0) o = cv_u32x4{0u, 0u, 0u, 0u};  // zero padding applies to the activated operand
        char* p = dst + a_wr + r * PR * CV_PSTR;
        if (r == ROUNDS - 1 && prow + PR * r >= HPIX) p = smem + 2 * ASZ;  // dump slot (16 bytes)
        *reinterpret_cast<cv_u32x4*>(p) = o;
    };
    auto a_write = [&](int r, cv_u32x4 raw, char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a_norm_pair(raw, i);
        a_store(r, raw, dst);
    };

    // pixels (B operand): halo pixel (nbase + n + ky) * 34 + l31 + kx, 16-byte granule 2 ks + h
    const unsigned xbase = unsigned(nbase * CV_HW + l31) * CV_PSTR + h * 16;

    cv_f32x16 acc[MB][NT];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // ---- prologue: chunk 0 halo
    gn_coeffs(0);
    if (!(CV_ABL & 64))
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) a_write(r, a_load(r, 0), abuf);
    gn_coeffs((nch > 1 ? 1 : 0) * CV_KC);  // chunk 1 is staged during chunk 0
    __syncthreads();

    unsigned acur = 0;  // byte offset of the current halo buffer
    // one chunk = 36 k-steps, fully unrolled (tap offsets are immediates, ring slots static).  STAGE: the next chunk's halo tile is
    // staged beside the MFMAs (every chunk but the last)
    auto chunk = [&](int c, auto stage_tag) __attribute__((always_inline)) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        const int c1 = (c + 1) * CV_KC;
        const int c2 = (c + 2 < nch ? c + 2 : nch - 1) * CV_KC;
        char* const anext = abuf + (acur ^ unsigned(ASZ));
        const char* const xl = abuf + acur + xbase;
        const int kbase = c * 36;
        cv_u32x4 rawq[2];
        cv_bf16x8 xf[2][4];
        // The B fragments ("x-sets" of 4 pixel tiles) of set s+1 are requested before the MFMAs of set s (register double buffer); the
        // order is pinned with sched_barrier: hipcc's own schedule issues each read right in front of its consumer and waits
        // lgkmcnt(0) every 4 MFMAs.  NT = 8: a k-step has two x-sets (pixel rows 0-3 / 4-7); MB = 2: both row blocks share one.
        constexpr int XS = NT == 8 ? 72 : 36;  // x-sets per chunk
        auto x_load = [&](int xs, int s) __attribute__((always_inline)) {
            const int J = NT == 8 ? xs / 2 : xs, half = NT == 8 ? xs % 2 : 0, t = J / 4, ks = J % 4, ky = t / 3, kx = t % 3;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                xf[s][n] = *reinterpret_cast<const cv_bf16x8*>(xl + (half * 4 + n + ky) * ROWB + kx * CV_PSTR + ks * 32);
        };
        x_load(0, 0);
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int J = g / NG, sub = g % NG, t = J / 4, ks = J % 4;
            const int xs = NT == 8 ? g : J, mb = MB == 2 ? sub : 0, half = NT == 8 ? sub : 0;
            const bool first_of_set = NT == 8 || sub == 0;
            __builtin_amdgcn_sched_barrier(0);
            // piece t of the next chunk's halo: requested at the top of stage t, normalised and written during stage t + 1
            if (STAGE && ks == 0 && sub == 0 && t < ROUNDS && !(CV_ABL & 16)) rawq[t & 1] = a_load(t, c1);
            if (first_of_set && xs + 1 < XS && !(CV_ABL & 4)) x_load(xs + 1, (xs + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (STAGE && t >= 1 && t <= ROUNDS && !(CV_ABL & 16)) {
                // the piece requested in the previous stage: one channel pair per quarter of the stage's groups, beside that
                // group's MFMAs, the store with the last one - ~18 vector instructions per 4 MFMAs instead of ~75 in one place.
                // The asm ties the arithmetic (no side effects) to this point: it would otherwise be scheduled directly behind
                // the load, in front of a wait for it
                constexpr int SG = 4 * NG;  // groups per stage
                const int sg = ks * NG + sub;
                cv_u32x4& raw = rawq[(t - 1) & 1];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (sg == (i * SG) / 4 + (SG > 4 ? 1 : 0)) {
                        asm volatile("" : "+v"(raw));
                        a_norm_pair(raw, i);
                        if (i == 3) a_store(t - 1, raw, anext);
                    }
            }
            if (STAGE && t == 8 && ks == 0 && sub == 0) gn_coeffs(c2);  // affine of the chunk staged during the NEXT chunk
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (CV_ABL & 2) asm volatile("" ::"v"(wr[J % D][mb]), "v"(xf[xs & 1][n]));
                else if constexpr (F16)
                    acc[mb][half * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cv_f16x8, wr[J % D][mb]), __builtin_bit_cast(cv_f16x8, xf[xs & 1][n]),
                                                                                   acc[mb][half * 4 + n], 0, 0, 0);
                else acc[mb][half * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[J % D][mb], xf[xs & 1][n], acc[mb][half * 4 + n], 0, 0, 0);
            }
            if (sub == NG - 1 && !(CV_ABL & 1)) {  // the ring slot is free: request the fragments of k-step J + D
                __builtin_amdgcn_sched_barrier(0);
                const int jn = kbase + J + D;
#pragma unroll
                for (int m = 0; m < MB; ++m)
                    wr[J % D][m] = *reinterpret_cast<const cv_bf16x8*>(wstream + (int64_t(m) * nk + (jn < nk ? jn : nk - 1)) * 1024);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // halo buffer hand-over: my stores are done, nobody reads the old buffer any more.  Raw barrier: __syncthreads() would
        // also drain vmcnt, i.e. the weight ring
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(CV_ABL & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        acur ^= unsigned(ASZ);
    };
    for (int c = 0; c + 1 < nch; ++c) chunk(c, std::true_type{});
    chunk(nch - 1, std::false_type{});

    // ---- epilogue: + bias (+ residual) -> bf16, GroupNorm statistics of the stored values.  No floating-point atomics: every
    // (wave, channel quad) sum has its own LDS slot, a thread per group adds the group's slots in a fixed order and the workgroup's
    // partial pair goes to a slot of its own in global memory (k_conv_stats_reduce adds the pixel tiles in fixed order)
    constexpr int NPH = (NW == 8 && CT == 128) ? 2 : 1;  // waves that share a channel quad (pixel halves of the tile)
    float2* const sq = reinterpret_cast<float2*>(smem);  // [NPH][CT / 4]; the halo buffers are free: every wave passed the last chunk's barrier
    const bool want_stats = a.stats_part != nullptr;
    const int cg_out = a.cg_out;
    const int g_first = want_stats ? (ct * CT) / cg_out : 0;
    const int g_last = want_stats ? (min(ct * CT + CT, a.cout) - 1) / cg_out : 0;
    const float* const bias = a.bias ? a.bias + int64_t(b) * a.bias_bstride + ct * CT : nullptr;
    if ((CV_ABL & 32) && a.B > 0) {  // timing only: no epilogue (one store keeps the accumulators alive)
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) t += acc[m][n][0];
        if (t == 123.456f) a.out[0] = (__bf16)t;
    } else {
        // the packed weights put channel 16 h + i of a row block's 32 on MFMA row (i & 3) + 8 (i >> 2) + 4 h, i.e. in accumulator
        // register i of lane (pixel, h): a lane owns 16 CONSECUTIVE channels of one pixel = two 16-byte accesses.  Pixel row outer,
        // row block inner: the wave's MB x 64 bytes of a pixel (a whole 128-byte line for MB = 2) are stored back to back - with the
        // row block outer the two halves of a line left L2 separately (WRITE_SIZE 1.22 x the output, profiles/r03_roofline_pmc.txt)
        float gs[MB][4], gq[MB][4];
        float4 bq[MB][4];
        bool live[MB];  // cout that is no multiple of the tile (UNet: 320 = 2.5 x 128): the row blocks past cout are zero weights, not stored
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            live[m] = ct * CT + (wb0 + m) * 32 < a.cout;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gs[m][q] = gq[m][q] = 0.f;
                bq[m][q] = bias && live[m] ? *reinterpret_cast<const float4*>(bias + (wb0 + m) * 32 + 16 * h + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (F32OUT) {
            float* const outf = reinterpret_cast<float*>(a.out);
            const float* const resf = reinterpret_cast<const float*>(a.residual);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int y = ty * TH + nbase + n, px = tx * CV_TW + l31;
                const int64_t opix = ((int64_t(b) * H + y) * W + px) * a.cout + ct * CT;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    if (!live[m]) continue;
                    const int64_t o = opix + (wb0 + m) * 32 + 16 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 r = resf ? *reinterpret_cast<const float4*>(resf + o + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                        r.x += fmaf(acc[m][n][4 * q], a.alpha, bq[m][q].x), r.y += fmaf(acc[m][n][4 * q + 1], a.alpha, bq[m][q].y);
                        r.z += fmaf(acc[m][n][4 * q + 2], a.alpha, bq[m][q].z), r.w += fmaf(acc[m][n][4 * q + 3], a.alpha, bq[m][q].w);
                        *reinterpret_cast<float4*>(outf + o + 4 * q) = r;
                        gs[m][q] += (r.x + r.y) + (r.z + r.w);
                        gq[m][q] = fmaf(r.x, r.x, fmaf(r.y, r.y, fmaf(r.z, r.z, fmaf(r.w, r.w, gq[m][q]))));
                    }
                }
            }
        } else
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int y = ty * TH + nbase + n, px = tx * CV_TW + l31;
            const int64_t opix = ((int64_t(b) * H + y) * W + px) * a.cout + ct * CT;
            uint4 rr[MB][2];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                rr[m][0] = rr[m][1] = make_uint4(0, 0, 0, 0);
                if (a.residual && live[m]) {
                    const int64_t o = opix + (wb0 + m) * 32 + 16 * h;
                    rr[m][0] = *reinterpret_cast<const uint4*>(a.residual + o);
                    rr[m][1] = *reinterpret_cast<const uint4*>(a.residual + o + 8);
                }
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int64_t o = opix + (wb0 + m) * 32 + 16 * h;
                const unsigned rw[8] = {rr[m][0].x, rr[m][0].y, rr[m][0].z, rr[m][0].w, rr[m][1].x, rr[m][1].y, rr[m][1].z, rr[m][1].w};
                cv_bf16x8 pk[2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v[4] = {acc[m][n][4 * q] + bq[m][q].x + __uint_as_float(rw[2 * q] << 16),
                                        acc[m][n][4 * q + 1] + bq[m][q].y + __uint_as_float(rw[2 * q] & 0xFFFF0000u),
                                        acc[m][n][4 * q + 2] + bq[m][q].z + __uint_as_float(rw[2 * q + 1] << 16),
                                        acc[m][n][4 * q + 3] + bq[m][q].w + __uint_as_float(rw[2 * q + 1] & 0xFFFF0000u)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const __bf16 r = (__bf16)v[j];
                        pk[q >> 1][(q & 1) * 4 + j] = r;
                        const float vr = (float)r;
                        gs[m][q] += vr;
                        gq[m][q] = fmaf(vr, vr, gq[m][q]);
                    }
                }
                if (live[m]) {
                    *reinterpret_cast<cv_bf16x8*>(a.out + o) = pk[0];
                    *reinterpret_cast<cv_bf16x8*>(a.out + o + 8) = pk[1];
                }
            }
        }
        if (want_stats) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float s = gs[m][q], ss = gq[m][q];
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) {
                        s += __shfl_xor(s, off);
                        ss += __shfl_xor(ss, off);
                    }
                    if (l31 == 0 && live[m]) sq[(NPH == 2 ? (wave & 1) * (CT / 4) : 0) + (wb0 + m) * 8 + 4 * h + q] = make_float2(s, ss);
                }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid <= g_last - g_first) {
            const int g = g_first + tid;
            const int q_lo = (max(g * cg_out, ct * CT) - ct * CT) >> 2, q_hi = (min(min((g + 1) * cg_out, ct * CT + CT), a.cout) - ct * CT) >> 2;
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int p = 0; p < NPH; ++p)
                for (int q = q_lo; q < q_hi; ++q) {
                    const float2 v = sq[p * (CT / 4) + q];
                    s += v.x;
                    ss += v.y;
                }
            a.stats_part[((int64_t(b) * a.nct + ct) * a.stat_slots + tid) * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx] = make_float2(s, ss);
        }
    }
}

// Moments of (image, group) from the per-workgroup partials of the epilogue above: one wave per (image, group); lane i adds pixel
// tiles i, i + 64, ... of the one or two output-channel tiles the group lies in, in f64, then a fixed shuffle tree
__global__ __launch_bounds__(64) void k_conv_stats_reduce(const float2* __restrict__ part, int nct, int CT, int slots, int ntiles, int cg, int G,
                                                          double* __restrict__ stats) {
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int ct0 = (g * cg) / CT, ct1 = ((g + 1) * cg - 1) / CT;
    double s = 0.0, ss = 0.0;
    for (int ct = ct0; ct <= ct1; ++ct) {
        const float2* p = part + ((int64_t(b) * nct + ct) * slots + (g - (ct * CT) / cg)) * ntiles;
        for (int i = threadIdx.x; i < ntiles; i += 64) {
            const float2 v = p[i];
            s += double(v.x);
            ss += double(v.y);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    if (threadIdx.x == 0) {
        stats[int64_t(blockIdx.x) * 2] = s;
        stats[int64_t(blockIdx.x) * 2 + 1] = ss;
    }
}

template <int CT, int MODE, bool UPS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_conv3x3(const ConvArgs a) {
    conv3x3_body<CT, MODE, UPS, NW, false>(a);
}

template <int CT, bool UPS, int NW, bool F16>
__global__ __launch_bounds__(NW * 64, 2) void k_conv3x3_f32acc(const ConvArgs a) {
    conv3x3_body<CT, 0, UPS, NW, true, F16>(a);
}

// f32 (B, H*W, C) -> bf16 hi / lo with x = hi + lo (+ <= 2^-17 |x|), optionally through y = act(x * scale + shift) with the
// per-(image, channel) GroupNorm affine of k_gn_affine: the operand split of the f32-accurate convolution.
// f16 (scale_hi > 0): hi = half(y * scale_hi), lo = half((y - hi / scale_hi) * scale_hi * 2^11): y = hi / s + lo / (s 2^11) to 2^-22 |y|; both terms
// live at the magnitude of y * s, so neither loses bits to the half's narrow exponent range; values beyond 65504 / s set the range flag
// Mapping: a thread owns TWO groups of 4 consecutive channels, float4 index q and q + 256 of its image (blockIdx.y): every load / store
// instruction of a wave is one contiguous run (lane stride 16 bytes in, 8 bytes out).  (One thread = 8 consecutive channels made each 16-byte
// load a 32-byte-strided gather and spent two 64-bit divisions per thread on the indices: 2.2 TB/s.)  One 32-bit remainder per thread.
__global__ __launch_bounds__(256) void k_split_nhwc(const float* __restrict__ x, const float* __restrict__ affine, int act, unsigned per_img4, int C,
                                                    __bf16* __restrict__ hi, __bf16* __restrict__ lo, __bf16* __restrict__ lo2, float scale_hi,
                                                    float lo_mul, int* __restrict__ flag) {
    typedef _Float16 cv_f16x4 __attribute__((ext_vector_type(4)));
    const unsigned cg = unsigned(C) >> 2;  // float4 groups per pixel
    const int b = blockIdx.y;
    const unsigned q0 = blockIdx.x * 512u + threadIdx.x;
    const int64_t ibase = int64_t(b) * per_img4 * 4;
    unsigned c4 = q0 % cg;
    const unsigned step = 256u % cg;
    float4 vin[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned q = q0 + 256u * u;
        vin[u] = q < per_img4 ? *reinterpret_cast<const float4*>(x + ibase + int64_t(q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned q = q0 + 256u * u;
        const int c = int(c4) * 4;  // first of this group's 4 channels
        c4 += step;
        if (c4 >= cg) c4 -= cg;
        if (q >= per_img4) continue;
        const int64_t e = ibase + int64_t(q) * 4;
        float v[4] = {vin[u].x, vin[u].y, vin[u].z, vin[u].w};
        if (affine) {
            // [8 scales][8 shifts] per 8-channel block (k_gn_affine); c & 7 is 0 or 4: two aligned 16-byte loads, not eight 4-byte ones (the
            // table loads, not the data, were most of this pass's vector-memory instructions)
            const float* t = affine + (int64_t(b) * C + (c & ~7)) * 2 + (c & 7);
            const float4 sc4 = *reinterpret_cast<const float4*>(t), sh4 = *reinterpret_cast<const float4*>(t + 8);
            const float scj[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shj[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = fmaf(v[j], scj[j], shj[j]);
                // SiLU on the two hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each) like the bf16 kernel's staging
                if (act == 1) y = y * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-CV_LOG2E * y));
                else if (act == 2) y = fmaxf(y, 0.f);
                v[j] = y;
            }
        }
        if (scale_hi > 0.f) {
            cv_f16x4 h4, l4;
            const float inv = 1.f / scale_hi, sl = scale_hi * lo_mul;  // powers of two: exact (lo_mul 2048: the scaled small term; 1: at hi's scale)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = v[j] * scale_hi;
                bad |= !(fabsf(t) <= 65504.f);
                h4[j] = (_Float16)t;
                l4[j] = (_Float16)((v[j] - float(h4[j]) * inv) * sl);
            }
            *reinterpret_cast<cv_f16x4*>(hi + e) = h4;
            *reinterpret_cast<cv_f16x4*>(lo + e) = l4;
        } else {
            cv_bf16x4 h4, l4, m4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h4[j] = (__bf16)v[j];
                const float r1 = v[j] - float(h4[j]);  // exact in f32
                l4[j] = (__bf16)r1;
                m4[j] = (__bf16)(r1 - float(l4[j]));   // third term: x = hi + lo + lo2 to 2^-25 |x|
            }
            *reinterpret_cast<cv_bf16x4*>(hi + e) = h4;
            *reinterpret_cast<cv_bf16x4*>(lo + e) = l4;
            if (lo2) *reinterpret_cast<cv_bf16x4*>(lo2 + e) = m4;
        }
    }
    if (bad) *flag = XM3D_ERANGE;
}

// OHWI (cout, 9, cin) bf16 -> fragment-ordered weight streams [cout tile][32-row block][chunk][tap][k-step][lane][8]: lane l of
// the wave that owns a row block loads, per k-step, the 8 channels 16 ks + 8 (l >> 5) .. + 7 of row (l & 31) = its MFMA A
// fragment, as ONE coalesced 16-byte-per-lane load; a wave's k-steps are contiguous (1 KiB each)
__global__ void k_conv3x3_pack(const __bf16* __restrict__ w, int cout, int cin, int CT, __bf16* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one 16-byte granule of the output
    const int64_t total = int64_t((cout + CT - 1) / CT) * CT * 9 * cin / 8;  // rows past cout (padding of the last tile): zeros
    if (i >= total) return;
    const int nch = cin / CV_KC;
    const int l = int(i & 63);
    int64_t r_ = i >> 6;
    const int ks = int(r_ & 3);
    r_ >>= 2;
    const int t = int(r_ % 9);
    r_ /= 9;
    const int c = int(r_ % nch);
    r_ /= nch;
    const int wb = int(r_ % (CT / 32));
    const int ct = int(r_ / (CT / 32));
    const int rho = l & 31;  // MFMA row -> channel 16 h + i with rho = (i & 3) + 8 (i >> 2) + 4 h (see the kernel's epilogue)
    const int row = ct * CT + wb * 32 + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
    const int k = c * CV_KC + ks * 16 + (l >> 5) * 8;
    reinterpret_cast<uint4*>(out)[i] = row < cout ? *reinterpret_cast<const uint4*>(w + (int64_t(row) * 9 + t) * cin + k) : make_uint4(0, 0, 0, 0);
}

// GroupNorm moments -> per-(image, channel) affine y = x * scale + shift, laid out [image][channel / 8][scale 8 | shift 8] so that
// a staging thread of k_conv3x3 fetches the 16 coefficients of its 8 channels with four 16-byte loads
// in_shift (C) or (B, C) with bstride = C, or null: the moments are those of x + in_shift (a bias the producer of x left to its
// consumer), so y = (x + in_shift) * scale + shift
__global__ void k_gn_affine(const double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                            const float* __restrict__ in_shift, int in_shift_bstride, int B, int C, int G, double inv_cnt, float eps,
                            float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C, g = c / (C / G);
    const double m = stats[(int64_t(b) * G + g) * 2] * inv_cnt;
    const float var = float(stats[(int64_t(b) * G + g) * 2 + 1] * inv_cnt - m * m);
    const float scale = gamma[c] * rsqrtf(fmaxf(var, 0.f) + eps);
    float* o = out + (int64_t(b) * C + (c & ~7)) * 2 + (c & 7);
    o[0] = scale;
    o[8] = beta[c] + ((in_shift ? in_shift[int64_t(b) * in_shift_bstride + c] : 0.f) - float(m)) * scale;
}

template <int CT, int MODE, bool UPS, int NW>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
    using G = ConvGeom<CT, NW>;
    static DeviceOnce configured;  // the attribute is per device
    if (configured.first()) {
        XM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<CT, MODE, UPS, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    }
    const int grid = a.B * a.tiles_y * a.tiles_x * a.nct;
    hipLaunchKernelGGL((k_conv3x3<CT, MODE, UPS, NW>), dim3(grid), dim3(G::NTH), G::LDS, s, a);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

template <int CT, bool UPS, int NW, bool F16>
static int launch_conv_f32acc_t(const ConvArgs& a, hipStream_t s) {
    using G = ConvGeom<CT, NW>;
    static DeviceOnce configured;  // the attribute is per device
    if (configured.first()) {
        XM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_f32acc<CT, UPS, NW, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    }
    const int grid = a.B * a.tiles_y * a.tiles_x * a.nct;
    hipLaunchKernelGGL((k_conv3x3_f32acc<CT, UPS, NW, F16>), dim3(grid), dim3(G::NTH), G::LDS, s, a);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

template <int CT, bool UPS, int NW>
static int launch_conv_f32acc(const ConvArgs& a, bool f16, hipStream_t s) {
    return f16 ? launch_conv_f32acc_t<CT, UPS, NW, true>(a, s) : launch_conv_f32acc_t<CT, UPS, NW, false>(a, s);
}

template <int CT, int NW>
static int dispatch_conv(const ConvArgs& a, int gn_act, bool upsample, hipStream_t s) {
    if (gn_act == 1) return launch_conv<CT, 2, false, NW>(a, s);
    if (gn_act == 2) return launch_conv<CT, 1, false, NW>(a, s);
    return upsample ? launch_conv<CT, 0, true, NW>(a, s) : launch_conv<CT, 0, false, NW>(a, s);
}

// the partial pairs of the statistics epilogue live behind the B * groups * 2 moments in the caller's buffer
static void conv_stats_setup(ConvArgs& a, double* stats_out, int64_t B, int cout, int cout_tile, int groups_out) {
    a.stats_part = stats_out ? reinterpret_cast<float2*>(stats_out + B * groups_out * 2) : nullptr;
    a.groups_out = stats_out ? groups_out : 1;
    a.cg_out = stats_out ? cout / groups_out : cout;
    a.stat_slots = stats_out ? cout_tile / a.cg_out + 2 : 0;
}
static int conv_stats_finish(const ConvArgs& a, double* stats_out, int cout_tile, hipStream_t s) {
    if (!stats_out) return XM3D_OK;
    hipLaunchKernelGGL(k_conv_stats_reduce, dim3(unsigned(a.B * a.groups_out)), dim3(64), 0, s, a.stats_part, a.nct, cout_tile, a.stat_slots,
                       a.tiles_x * a.tiles_y, a.cg_out, a.groups_out, stats_out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

}  // namespace xm3d

using namespace xm3d;

// doubles the `stats_out` buffer of xm3d_conv3x3_nhwc / _f32acc must hold: the B * groups_out * 2 moments (first) + the per-workgroup
// partial pairs they are reduced from in fixed order (no floating-point atomics: bit-reproducible statistics)
extern "C" int64_t xm3d_conv3x3_stats_doubles(int64_t B, int32_t H, int32_t W, int32_t cout, int32_t cout_tile, int32_t groups_out, int32_t waves) {
    if (B <= 0 || H <= 0 || W <= 0 || cout <= 0 || groups_out <= 0 || cout % groups_out != 0 || (cout_tile != 128 && cout_tile != 256)) return 0;
    if (waves == 0) waves = xm3d_conv3x3_default_waves(H, W, 0, cout);
    const int TH = waves == 8 ? 8 : 4;
    const int64_t ntiles = int64_t((H + TH - 1) / TH) * ((W + CV_TW - 1) / CV_TW), nct = (cout + cout_tile - 1) / cout_tile;
    return B * groups_out * 2 + B * nct * (cout_tile / (cout / groups_out) + 2) * ntiles;
}

// 256 / 128 when cout is a multiple; other multiples of 32 (UNet: 320) run on 128-channel tiles with a zero-padded last tile
extern "C" int xm3d_conv3x3_cout_tile(int cout) { return cout <= 0 ? 0 : (cout % 256 == 0 ? 256 : (cout % 32 == 0 ? 128 : 0)); }

extern "C" int64_t xm3d_conv3x3_packed_elems(int cout, int cin, int cout_tile) {
    return cout_tile > 0 ? int64_t((cout + cout_tile - 1) / cout_tile) * cout_tile * 9 * cin : 0;
}

extern "C" int xm3d_conv3x3_pack_weight(const void* w_ohwi, int cout, int cin, int cout_tile, void* packed, void* stream) {
    XM3D_REQUIRE(w_ohwi && packed, "conv3x3_pack_weight: null pointer");
    XM3D_REQUIRE(cin > 0 && cin % CV_KC == 0, "conv3x3_pack_weight: cin %d is not a multiple of %d", cin, CV_KC);
    XM3D_REQUIRE((cout_tile == 128 || cout_tile == 256) && cout > 0 && cout % 32 == 0, "conv3x3_pack_weight: cout %d / tile %d unsupported", cout,
                 cout_tile);
    const int64_t total = xm3d_conv3x3_packed_elems(cout, cin, cout_tile) / 8;
    hipLaunchKernelGGL(k_conv3x3_pack, dim3(unsigned((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16*>(w_ohwi), cout, cin, cout_tile, static_cast<__bf16*>(packed));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

// workgroup geometry when the caller does not choose (waves = 0); see ConvGeom.  Measured on MI355X, 20 views (tools/conv_bench.py)
extern "C" int xm3d_conv3x3_default_waves(int H, int W, int cin, int cout) {
    (void)W;
    (void)cout;
    if (H % 8 != 0) return 4;
    // Alone, two 4-wave workgroups per CU are 2 - 14 % faster on the VAE's layers (prologue and epilogue overlap the other
    // workgroup's MFMAs) and one 8-wave workgroup 3 - 4 % on the UNet's cin >= 640 at 32^2 (profiles/r03_conv_bench_v4.log).  Inside
    // the forward, where the UNet, the VAE decoder and the sparse 3D branch share the chip on three streams, the 8-wave geometry
    // wins on every layer it was tried on: it moves 1.25 x fewer bytes (halo 1.33 x instead of 1.59 x; 1.40 vs 1.76 GB on
    // 512 ch @ 128^2, profiles/roofline_pmc.json).  End to end, scenes/s: 8-wave everywhere 36.4 / 36.0, 4-wave for cin <= 512
    // below 128^2 35.6, 4-wave for cin <= 128 35.4, 4-wave everywhere 35.4 (profiles/r03_bench_geometry_ab.log).  The 4-wave
    // geometry stays for heights that are a multiple of 4 only, and as a caller's explicit choice.
    (void)cin;
    return 8;
}

extern "C" int64_t xm3d_conv3x3_ws_bytes(int64_t B, int32_t cin) { return B > 0 && cin > 0 ? B * int64_t(cin) * 2 * int64_t(sizeof(float)) : 0; }

extern "C" int xm3d_conv3x3_nhwc(const void* x, int64_t B, int H, int W, int cin, const void* wpacked, int cout, int cout_tile,
                                 const double* gn_stats, const float* gamma, const float* beta, const float* in_shift,
                                 int in_shift_bstride, float eps, int groups, int act, const float* bias, int bias_bstride, const void* residual, void* out, double* stats_out,
                                 int groups_out, int upsample, int waves, void* ws, void* stream) {
    XM3D_REQUIRE(x && wpacked && out, "conv3x3_nhwc: null pointer");
    XM3D_REQUIRE(waves == 0 || waves == 4 || waves == 8, "conv3x3_nhwc: waves must be 0 (auto), 4 or 8");
    if (waves == 0) waves = xm3d_conv3x3_default_waves(H, W, cin, cout);
    const int TH = waves == 8 ? 8 : 4;
    XM3D_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && H % TH == 0 && W % CV_TW == 0,
                 "conv3x3_nhwc: output %dx%d is not a multiple of the %dx%d pixel tile", H, W, TH, CV_TW);
    XM3D_REQUIRE(cin > 0 && cin % CV_KC == 0, "conv3x3_nhwc: cin %d is not a multiple of %d", cin, CV_KC);
    XM3D_REQUIRE((cout_tile == 128 || cout_tile == 256) && cout > 0 && cout % 32 == 0, "conv3x3_nhwc: cout %d / tile %d unsupported", cout,
                 cout_tile);
    XM3D_REQUIRE(int64_t(H) * W * (cin > cout ? cin : cout) < (int64_t(1) << 31), "conv3x3_nhwc: image too large for 32-bit offsets");
    XM3D_REQUIRE(bias_bstride == 0 || bias_bstride == cout, "conv3x3_nhwc: bias_bstride must be 0 or cout");
    XM3D_REQUIRE(in_shift_bstride == 0 || in_shift_bstride == cin, "conv3x3_nhwc: in_shift_bstride must be 0 or cin");
    XM3D_REQUIRE(!in_shift || gn_stats, "conv3x3_nhwc: in_shift only applies to the GroupNorm input");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gn_stats) |
                   reinterpret_cast<uintptr_t>(stats_out) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
                 "conv3x3_nhwc: tensors must be 16-byte aligned");
    const bool gn = gn_stats != nullptr;
    if (gn) {
        XM3D_REQUIRE(gamma && beta && groups > 0 && cin % groups == 0, "conv3x3_nhwc: GroupNorm needs gamma, beta and groups dividing cin");
        XM3D_REQUIRE(ws, "conv3x3_nhwc: GroupNorm needs a workspace of xm3d_conv3x3_ws_bytes(B, cin) bytes");
        XM3D_REQUIRE(act == 1 || act == 2, "conv3x3_nhwc: the fused GroupNorm is followed by SiLU (act 1) or ReLU (act 2)");
        XM3D_REQUIRE(!upsample, "conv3x3_nhwc: upsample and GroupNorm cannot be combined");
    } else {
        XM3D_REQUIRE(act == 0, "conv3x3_nhwc: an activation needs the GroupNorm statistics");
    }
    if (upsample) XM3D_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_nhwc: upsample needs even output dims");
    if (stats_out)
        XM3D_REQUIRE(groups_out > 0 && cout % groups_out == 0 && (cout / groups_out) % 4 == 0 && cout_tile / (cout / groups_out) + 2 <= 128,
                     "conv3x3_nhwc: output statistics need a multiple of 4 channels per group (cout %d, groups %d)", cout, groups_out);
    hipStream_t s = as_stream(stream);
    ConvArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.wp = static_cast<const __bf16*>(wpacked);
    a.affine = nullptr;
    if (gn) {
        const int n = int(B) * cin;
        hipLaunchKernelGGL(k_gn_affine, dim3((n + 255) / 256), dim3(256), 0, s, gn_stats, gamma, beta, in_shift, in_shift_bstride, int(B), cin, groups,
                           1.0 / (double(H) * W * (cin / groups)), eps, static_cast<float*>(ws));
        a.affine = static_cast<const float*>(ws);
    }
    a.bias = bias;
    a.residual = static_cast<const __bf16*>(residual);
    a.out = static_cast<__bf16*>(out);
    conv_stats_setup(a, stats_out, B, cout, cout_tile, groups_out);
    a.B = int(B);
    a.H = H;
    a.W = W;
    a.cin = cin;
    a.cout = cout;
    a.bias_bstride = bias_bstride;
    a.alpha = 1.f;
    a.tiles_x = W / CV_TW;
    a.tiles_y = H / TH;
    a.nct = (cout + cout_tile - 1) / cout_tile;
    const int gn_act = gn ? act : 0;
    int rc;
    if (cout_tile == 256) rc = waves == 8 ? dispatch_conv<256, 8>(a, gn_act, upsample != 0, s) : dispatch_conv<256, 4>(a, gn_act, upsample != 0, s);
    else rc = waves == 8 ? dispatch_conv<128, 8>(a, gn_act, upsample != 0, s) : dispatch_conv<128, 4>(a, gn_act, upsample != 0, s);
    return rc != XM3D_OK ? rc : conv_stats_finish(a, stats_out, cout_tile, s);
}

// ---- f32-accurate convolution from three bf16 passes over split operands:  x = x_hi + x_lo, w = w_hi + w_lo (bf16 each),
//      conv(x, w) ~= conv(x_hi, w_hi) + conv(x_hi, w_lo) + conv(x_lo, w_hi)   (the dropped x_lo * w_lo and the split residuals are
//      <= 2^-16 |x w| each), f32 accumulation across the passes in the f32 output tensor.
static int split_nhwc_impl(const float* x, int64_t B, int64_t HW, int32_t C, const double* gn_stats, const float* gamma, const float* beta,
                           const float* in_shift, int32_t in_shift_bstride, float eps, int32_t groups, int32_t act, void* hi, void* lo,
                           void* lo2, float scale_hi, float lo_mul, void* ws, void* stream) {
    XM3D_REQUIRE(x && hi && lo, "split_bf16_nhwc: null pointer");
    XM3D_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0, "split_bf16_nhwc: C %d must be a multiple of 8", C);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15) == 0,
                 "split_bf16_nhwc: tensors must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    const float* affine = nullptr;
    if (gn_stats) {
        XM3D_REQUIRE(gamma && beta && groups > 0 && C % groups == 0 && ws && (act == 0 || act == 1 || act == 2), "split_bf16_nhwc: bad GroupNorm arguments");
        XM3D_REQUIRE(in_shift_bstride == 0 || in_shift_bstride == C, "split_bf16_nhwc: in_shift_bstride must be 0 or C");
        XM3D_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "split_bf16_nhwc: the affine workspace must be 16-byte aligned");
        const int n = int(B) * C;
        hipLaunchKernelGGL(k_gn_affine, dim3((n + 255) / 256), dim3(256), 0, s, gn_stats, gamma, beta, in_shift, in_shift_bstride, int(B), C, groups,
                           1.0 / (double(HW) * (C / groups)), eps, static_cast<float*>(ws));
        affine = static_cast<const float*>(ws);
    } else {
        XM3D_REQUIRE(act == 0 && !in_shift, "split_bf16_nhwc: activation / shift need the GroupNorm statistics");
    }
    const int64_t per_img4 = HW * (C / 4);  // float4 groups per image
    XM3D_REQUIRE(per_img4 < (int64_t(1) << 31) && B <= 65535, "split_bf16_nhwc: image of %lld x %d elements / batch %lld too large", (long long)HW, C, (long long)B);
    hipLaunchKernelGGL(k_split_nhwc, dim3(unsigned((per_img4 + 511) / 512), unsigned(B)), dim3(256), 0, s, x, affine, act, unsigned(per_img4), C,
                       static_cast<__bf16*>(hi), static_cast<__bf16*>(lo), static_cast<__bf16*>(lo2), scale_hi, lo_mul, device_flag());
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_split_bf16_nhwc(const float* x, int64_t B, int64_t HW, int32_t C, const double* gn_stats, const float* gamma, const float* beta,
                                    const float* in_shift, int32_t in_shift_bstride, float eps, int32_t groups, int32_t act, void* hi, void* lo,
                                    void* lo2, void* ws, void* stream) {
    return split_nhwc_impl(x, B, HW, C, gn_stats, gamma, beta, in_shift, in_shift_bstride, eps, groups, act, hi, lo, lo2, 0.f, 0.f, ws, stream);
}

// the two-term split in IEEE halves: y = hi / scale_hi + lo / (scale_hi * 2048), |y - (..)| <= 2^-22 |y| (see k_split_nhwc); scale_hi a power
// of two such that |y| * scale_hi <= 65504 (larger values set the sticky range flag, xm3d_check_flag)
extern "C" int xm3d_split_f16_nhwc(const float* x, int64_t B, int64_t HW, int32_t C, const double* gn_stats, const float* gamma, const float* beta,
                                   const float* in_shift, int32_t in_shift_bstride, float eps, int32_t groups, int32_t act, float scale_hi, void* hi,
                                   void* lo, void* ws, void* stream) {
    XM3D_REQUIRE(scale_hi > 0.f, "split_f16_nhwc: scale_hi must be positive");
    return split_nhwc_impl(x, B, HW, C, gn_stats, gamma, beta, in_shift, in_shift_bstride, eps, groups, act, hi, lo, nullptr, scale_hi, 2048.f, ws, stream);
}

// ... with both terms at ONE scale: y * scale_hi = hi + lo (lo = half(y scale_hi - hi), not times 2^11) - the operand form of the one-launch
// f32-accurate GEMM (xm3d_gemm_f32), whose three products share one accumulator.  lo is a normal half while |y| scale_hi >= 2^-3; smaller
// values keep an absolute precision of 2^-24 / scale_hi.  Range: |y| scale_hi <= 65504 (beyond: the sticky range flag).
extern "C" int xm3d_split_f16t_nhwc(const float* x, int64_t B, int64_t HW, int32_t C, const double* gn_stats, const float* gamma, const float* beta,
                                    const float* in_shift, int32_t in_shift_bstride, float eps, int32_t groups, int32_t act, float scale_hi, void* hi,
                                    void* lo, void* ws, void* stream) {
    XM3D_REQUIRE(scale_hi > 0.f, "split_f16t_nhwc: scale_hi must be positive");
    return split_nhwc_impl(x, B, HW, C, gn_stats, gamma, beta, in_shift, in_shift_bstride, eps, groups, act, hi, lo, nullptr, scale_hi, 1.f, ws, stream);
}

// out (f32) = conv3x3(x bf16, w bf16 packed) + bias + residual (f32; may be `out` itself: accumulate in place)
extern "C" int xm3d_conv3x3_nhwc_f32acc2(const void* x, int64_t B, int32_t H, int32_t W, int32_t cin, const void* wpacked, int32_t cout, int32_t cout_tile,
                                         const float* bias, int32_t bias_bstride, const float* residual, float* out, double* stats_out,
                                         int32_t groups_out, int32_t upsample, int32_t waves, int32_t f16, float alpha, void* stream);

extern "C" int xm3d_conv3x3_nhwc_f32acc(const void* x, int64_t B, int32_t H, int32_t W, int32_t cin, const void* wpacked, int32_t cout, int32_t cout_tile,
                                        const float* bias, int32_t bias_bstride, const float* residual, float* out, double* stats_out,
                                        int32_t groups_out, int32_t upsample, int32_t waves, void* stream) {
    return xm3d_conv3x3_nhwc_f32acc2(x, B, H, W, cin, wpacked, cout, cout_tile, bias, bias_bstride, residual, out, stats_out, groups_out, upsample, waves, 0, 1.f, stream);
}

// ... with the operands in IEEE halves (f16 = 1) and the product scaled by alpha before it is added: out = residual + alpha * conv(x, w) + bias
extern "C" int xm3d_conv3x3_nhwc_f32acc2(const void* x, int64_t B, int32_t H, int32_t W, int32_t cin, const void* wpacked, int32_t cout, int32_t cout_tile,
                                         const float* bias, int32_t bias_bstride, const float* residual, float* out, double* stats_out,
                                         int32_t groups_out, int32_t upsample, int32_t waves, int32_t f16, float alpha, void* stream) {
    XM3D_REQUIRE(x && wpacked && out, "conv3x3_nhwc_f32acc: null pointer");
    XM3D_REQUIRE(waves == 0 || waves == 4 || waves == 8, "conv3x3_nhwc_f32acc: waves must be 0 (auto), 4 or 8");
    if (waves == 0) waves = xm3d_conv3x3_default_waves(H, W, cin, cout);
    const int TH = waves == 8 ? 8 : 4;
    XM3D_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && H % TH == 0 && W % CV_TW == 0,
                 "conv3x3_nhwc_f32acc: output %dx%d is not a multiple of the %dx%d pixel tile", H, W, TH, CV_TW);
    XM3D_REQUIRE(cin > 0 && cin % CV_KC == 0 && (cout_tile == 128 || cout_tile == 256) && cout > 0 && cout % 32 == 0, "conv3x3_nhwc_f32acc: channels unsupported");
    XM3D_REQUIRE(int64_t(H) * W * (cin > cout ? cin : cout) < (int64_t(1) << 31), "conv3x3_nhwc_f32acc: image too large for 32-bit offsets");
    XM3D_REQUIRE(bias_bstride == 0 || bias_bstride == cout, "conv3x3_nhwc_f32acc: bias_bstride must be 0 or cout");
    if (upsample) XM3D_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_nhwc_f32acc: upsample needs even output dims");
    if (stats_out)
        XM3D_REQUIRE(groups_out > 0 && cout % groups_out == 0 && (cout / groups_out) % 4 == 0 && cout_tile / (cout / groups_out) + 2 <= 128,
                     "conv3x3_nhwc_f32acc: output statistics need a multiple of 4 channels per group (cout %d, groups %d)", cout, groups_out);
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(stats_out)) & 15) == 0,
                 "conv3x3_nhwc_f32acc: tensors must be 16-byte aligned");
    ConvArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.wp = static_cast<const __bf16*>(wpacked);
    a.affine = nullptr;
    a.bias = bias;
    a.residual = reinterpret_cast<const __bf16*>(residual);  // f32 in this variant (the kernel casts back)
    a.out = reinterpret_cast<__bf16*>(out);
    conv_stats_setup(a, stats_out, B, cout, cout_tile, groups_out);
    a.B = int(B), a.H = H, a.W = W, a.cin = cin, a.cout = cout;
    a.bias_bstride = bias_bstride;
    a.alpha = alpha;
    a.tiles_x = W / CV_TW;
    a.tiles_y = H / TH;
    a.nct = (cout + cout_tile - 1) / cout_tile;
    hipStream_t s = as_stream(stream);
    const bool h = f16 != 0;
    int rc;
    if (cout_tile == 256) {
        if (waves == 8) rc = upsample ? launch_conv_f32acc<256, true, 8>(a, h, s) : launch_conv_f32acc<256, false, 8>(a, h, s);
        else rc = upsample ? launch_conv_f32acc<256, true, 4>(a, h, s) : launch_conv_f32acc<256, false, 4>(a, h, s);
    } else if (waves == 8) rc = upsample ? launch_conv_f32acc<128, true, 8>(a, h, s) : launch_conv_f32acc<128, false, 8>(a, h, s);
    else rc = upsample ? launch_conv_f32acc<128, true, 4>(a, h, s) : launch_conv_f32acc<128, false, 4>(a, h, s);
    return rc != XM3D_OK ? rc : conv_stats_finish(a, stats_out, cout_tile, s);
}
