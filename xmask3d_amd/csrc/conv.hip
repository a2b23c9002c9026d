// 3x3 convolution (stride 1, zero padding 1) over channels-last bf16 activations on the gfx950 matrix cores, with the
// GroupNorm that precedes it and the GroupNorm that follows it folded in:
//
//     out = conv3x3( act( GN(x) ) ) + bias (+ residual)          statistics of `out` accumulated on the way
//
// This is the ResnetBlock body of ldm's VAE / UNet (GroupNorm(32) -> SiLU -> Conv2d(3x3), twice, + skip), reached from
// /root/reference/models/modeling/meta_arch/ldm.py:386-414 (VAE encoder), :425-446 (UNet), :448-490 (VAE decoder): the FLOP
// majority of the whole scene.  It replaces, per convolution, {GroupNorm statistics pass, GroupNorm apply + SiLU pass, library
// implicit-GEMM convolution, bias / residual pass} by ONE launch that reads x once (+ 27 % halo) and writes out once.
//
// Decomposition (one workgroup = 8 waves = one 8 x 32 pixel tile of one image x CT output channels):
//   * K loop = input-channel chunks of 64 (outer) x the 9 filter taps (inner).  Per chunk the (8+2) x (32+2) HALO tile of the
//     input is staged ONCE in LDS - raw bf16 from global memory into registers, normalised with the per-(image, channel)
//     affine derived from the f64 GroupNorm moments, SiLU, rounded to bf16, written with a 144-byte pixel stride - and all 9
//     taps read it with the same per-lane address plus an immediate: a tap is a shift of the pixel index, and the padded
//     stride makes the 32-pixel fragment reads bank-conflict free at ANY shift.  The normalisation work is therefore done once
//     per element (x 1.33 halo), not 9 times, and the activation operand costs 1/9 of an im2col-style staging.
//   * per (chunk, tap) stage the CT x 64 weight tile arrives by LDS-DMA (global_load_lds_dwordx4) from a pre-packed image
//     (xm3d_conv3x3_pack_weight: stage-major, rows XOR-swizzled so that the fragment reads are conflict free); two weight
//     buffers, the DMA of stage s+1 is issued at the top of stage s and has a whole stage (>= 1024 MFMA cycles) to land;
//     the halo tile of the NEXT chunk is staged during taps 0..5 of the current one (one 16-byte piece per thread and tap),
//     so the normalisation VALU work runs beside the MFMAs of the other wave on the SIMD.  One barrier per stage.
//   * MFMA v_mfma_f32_32x32x16_bf16 with the WEIGHTS as A operand (rows = output channels) and the pixels as B operand:
//     the accumulator has the pixel on the lane and 4 consecutive output channels per register quad = 8-byte bf16 stores,
//     and the GroupNorm statistics of the result are in-lane sums + one 32-lane reduction per channel quad.
//   * waves 2 (output channels) x 4 (pixel rows): wave tile (CT/2) channels x 64 pixels.
// Bound: MFMA (bf16).  Algorithmic FLOP = 2 * B*H*W * 9*Cin * Cout; bytes = B*H*W*(Cin + Cout [+ Cout residual])*2 + weights.
#include "common.h"

namespace xm3d {

typedef float cv_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cv_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cv_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned cv_u32x4 __attribute__((ext_vector_type(4)));

constexpr int CV_TW = 32, CV_TH = 8;                 // output pixel tile (one image)
constexpr int CV_HW = CV_TW + 2, CV_HH = CV_TH + 2;  // halo tile
constexpr int CV_HPIX = CV_HW * CV_HH;               // 340 halo pixels
constexpr int CV_PSTR = 144;                         // bytes per halo pixel: 64 channels bf16 + 16 pad
constexpr int CV_ASZ = CV_HPIX * CV_PSTR;            // 48,960 bytes per halo buffer
constexpr int CV_KC = 64;                            // input channels per chunk
constexpr int CV_ROUNDS = (CV_HPIX * 8 + 511) / 512; // 16-byte pieces per thread and chunk (6; the last round is partial)
constexpr float CV_LOG2E = 1.4426950408889634f;
#ifndef CV_ABL
#define CV_ABL 0  // timing-only ablations (tools/conv_ablate.sh): 1 no weight DMA, 2 no MFMA, 4 no fragment reads, 8 no stage barrier, 16 no halo staging
#endif

struct ConvArgs {
    const __bf16* x;          // (B, H>>ups, W>>ups, cin)
    const __bf16* wp;         // packed weights (xm3d_conv3x3_pack_weight)
    const double* gn_stats;   // (B, groups_in, 2) sum / sum of squares of x, or null
    const float* gamma;       // (cin)
    const float* beta;        // (cin)
    const float* bias;        // (cout) or (B, cout) with bias_bstride = cout, or null
    const __bf16* residual;   // (B, H, W, cout) or null
    __bf16* out;              // (B, H, W, cout)
    double* stats_out;        // (B, groups_out, 2) accumulated (+=), or null
    int B, H, W, cin, cout;
    int groups_in, cg_in;
    double inv_cnt_in;
    float eps;
    int bias_bstride;
    int groups_out, cg_out;
    int tiles_x, tiles_y, nct;
};

__device__ __forceinline__ unsigned cv_lds_addr(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>(reinterpret_cast<const __attribute__((address_space(3))) char*>(reinterpret_cast<uintptr_t>(p))));
}

__device__ __forceinline__ void cv_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(gsrc)),
                                     reinterpret_cast<__attribute__((address_space(3))) void*>(static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds_wave_base))),
                                     16, 0, 0);
}

// MODE 0: x is the operand as it stands (plain convolution); MODE 2: operand = SiLU(GroupNorm(x)).
// UPS: x has half the resolution, the operand is its nearest-neighbour 2x upsampling (ldm's Upsample -> conv).
template <int CT, int MODE, bool UPS>
__global__ __launch_bounds__(512, 2) void k_conv3x3(const ConvArgs a) {
    constexpr int MT = CT / 64;        // 32-channel MFMA row tiles per wave
    constexpr int BSZ = CT * 128;      // bytes per weight stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const abuf = smem;                 // 2 halo buffers
    char* const bbuf = smem + 2 * CV_ASZ;    // 2 weight buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_m = wave >> 2, w_n = wave & 3;
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

    // ---- halo staging: thread owns 16-byte piece (pixel prow + 64 r, channels 8 kc .. 8 kc + 7) of every round r
    const int kc = tid & 7, prow = tid >> 3;
    int aoff[CV_ROUNDS];  // element offset of the piece's source inside the image; -1: outside the image (zero padding)
#pragma unroll
    for (int r = 0; r < CV_ROUNDS; ++r) {
        const int p = prow + 64 * r;
        const int hy = p / CV_HW, hx = p - hy * CV_HW;
        const int gy = ty * CV_TH - 1 + hy, gx = tx * CV_TW - 1 + hx;
        const bool inb = p < CV_HPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int sy = UPS ? gy >> 1 : gy, sx = UPS ? gx >> 1 : gx;
        aoff[r] = inb ? (sy * Wi + sx) * cin + kc * 8 : -1;
    }
    const unsigned a_wr = unsigned(prow) * CV_PSTR + kc * 16;  // + r * 64 * CV_PSTR

    float sc[8], sh[8];  // GroupNorm affine of the chunk being staged: y = x * sc + sh
    auto gn_coeffs = [&](int c0) __attribute__((always_inline)) {
        if constexpr (MODE != 0) {
            const int ch0 = c0 + kc * 8;
            const int g0 = ch0 / a.cg_in, g1 = (ch0 + 7) / a.cg_in;  // <= 2 groups per 8 channels (cg_in >= 4)
            const double* st = a.gn_stats + (int64_t(b) * a.groups_in + g0) * 2;
            const double s0 = st[0], q0 = st[1], s1 = st[(g1 - g0) * 2], q1 = st[(g1 - g0) * 2 + 1];
            const double m0 = s0 * a.inv_cnt_in, m1 = s1 * a.inv_cnt_in;
            const float v0 = float(q0 * a.inv_cnt_in - m0 * m0), v1 = float(q1 * a.inv_cnt_in - m1 * m1);
            const float r0 = rsqrtf(fmaxf(v0, 0.f) + a.eps), r1 = rsqrtf(fmaxf(v1, 0.f) + a.eps);
            const float fm0 = float(m0), fm1 = float(m1);
            const int split = (g0 + 1) * a.cg_in - ch0;
            const float4 ga0 = *reinterpret_cast<const float4*>(a.gamma + ch0), ga1 = *reinterpret_cast<const float4*>(a.gamma + ch0 + 4);
            const float4 be0 = *reinterpret_cast<const float4*>(a.beta + ch0), be1 = *reinterpret_cast<const float4*>(a.beta + ch0 + 4);
            const float ga[8] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w};
            const float be[8] = {be0.x, be0.y, be0.z, be0.w, be1.x, be1.y, be1.z, be1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float mean = j < split ? fm0 : fm1, rstd = j < split ? r0 : r1;
                sc[j] = ga[j] * rstd;
                sh[j] = be[j] - mean * sc[j];
            }
        }
    };
    // branch free (the whole tap loop is one basic block between barriers): a piece outside the image reads the image's first
    // pixel and is zeroed on the way to LDS; the threads without a piece in the last, partial round write to a dump slot
    auto a_load = [&](int r, int c0) __attribute__((always_inline)) -> uint4 {
        return *reinterpret_cast<const uint4*>(xb + (aoff[r] >= 0 ? aoff[r] : kc * 8) + c0);
    };
    // the same load hidden from hipcc's wait bookkeeping (main loop): beside LDS-DMA loads hipcc waits vmcnt(0) at the first use
    // of an ordinary load - here that would be mid-stage, a few hundred cycles behind the request.  Nothing waits for it
    // explicitly: the value is first used in the NEXT stage, behind the vmcnt(0) + barrier that ends this one.
    auto a_load_async = [&](int r, int c0) __attribute__((always_inline)) -> cv_u32x4 {
        cv_u32x4 v;
        const __bf16* p = xb + (aoff[r] >= 0 ? aoff[r] : kc * 8) + c0;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
        return v;
    };
    auto a_write = [&](int r, uint4 raw, char* dst, bool hidden) __attribute__((always_inline)) {
        uint4 o = raw;
        if constexpr (MODE != 0) {
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
            float y[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[2 * i] = fmaf(__uint_as_float(w[i] << 16), sc[2 * i], sh[2 * i]);
                y[2 * i + 1] = fmaf(__uint_as_float(w[i] & 0xFFFF0000u), sc[2 * i + 1], sh[2 * i + 1]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = y[i] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-CV_LOG2E * y[i]));
            cv_bf16x8 pk;
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] = (__bf16)y[i];
            o = __builtin_bit_cast(uint4, pk);
        }
        if (aoff[r] < 0) o = make_uint4(0, 0, 0, 0);  // zero padding applies to the activated operand
        char* p = dst + a_wr + r * 64 * CV_PSTR;
        if (r == CV_ROUNDS - 1 && prow + 64 * r >= CV_HPIX) p = smem + 2 * CV_ASZ + 2 * BSZ;  // dump slot (16 bytes)
        if (hidden) {
            // main loop: an LDS store hipcc knows of makes it wait vmcnt(0) first while an LDS-DMA is in flight (it cannot tell
            // that the two never overlap) - mid-stage, behind requests issued a few hundred cycles earlier.  The store is
            // retired by the lgkmcnt(0) in front of the barrier that ends the stage.
            const cv_u32x4 ov = {o.x, o.y, o.z, o.w};
            asm volatile("ds_write_b128 %0, %1" ::"v"(cv_lds_addr(p)), "v"(ov) : "memory");
        } else {
            *reinterpret_cast<uint4*>(p) = o;
        }
    };

    // ---- weight stages: (ct, chunk, tap) images of BSZ bytes, 512 threads x 16 bytes per round
    const char* const wsrc = reinterpret_cast<const char*>(a.wp) + int64_t(ct) * nch * 9 * BSZ + tid * 16;
    auto b_issue = [&](int stage, char* dst) __attribute__((always_inline)) {
        const char* src = wsrc + int64_t(stage) * BSZ;
#pragma unroll
        for (int r = 0; r < BSZ / 8192; ++r) cv_glds16(src + r * 8192, dst + r * 8192 + wave * 1024);
    };

    // ---- fragment addresses
    // weights (A operand): row = w_m * MT*32 + m*32 + l31, 16-byte granule (2 ks + h) ^ ((row >> 1) & 7)
    const unsigned wbase = unsigned(w_m * MT * 32 + l31) * 128 + (((unsigned(h) ^ ((unsigned(lane) >> 1) & 7u)) & 7u) << 4);
    // pixels (B operand): halo pixel (2 w_n + n + ky) * 34 + l31 + kx, granule 2 ks + h
    const unsigned xbase = unsigned((2 * w_n) * CV_HW + l31) * CV_PSTR + h * 16;

    cv_f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // ---- prologue: chunk 0 halo + stage 0 weights
    b_issue(0, bbuf);
    gn_coeffs(0);
#pragma unroll
    for (int r = 0; r < CV_ROUNDS; ++r) a_write(r, a_load(r, 0), abuf, false);
    __syncthreads();

    unsigned bcur = 0, acur = 0;  // byte offsets of the current weight / halo buffer
    cv_bf16x8 wf[2][MT], xf[2][2];
    const int nstage = nch * 9;
    for (int c = 0; c < nch; ++c) {
        // the chunk staged during this one; past the end the last chunk is staged again into the buffer nobody reads any
        // more (keeps the loop free of branches)
        const int c1 = (c + 1 < nch ? c + 1 : c) * CV_KC;
        gn_coeffs(c1);
        char* const anext = abuf + (acur ^ unsigned(CV_ASZ));
        cv_u32x4 rawq[2];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            // piece t of the next chunk's halo: requested in stage t, normalised and written in stage t + 1 (the barrier that
            // ends a stage drains vmcnt, so the request has the whole stage to come back from HBM)
            if (t < CV_ROUNDS && !(CV_ABL & 16)) rawq[t & 1] = a_load_async(t, c1);
            if (!(CV_ABL & 1)) {
                const int s1 = c * 9 + t + 1;
                b_issue(s1 < nstage ? s1 : nstage - 1, bbuf + (bcur ^ unsigned(BSZ)));
            }
            const char* const xl = abuf + acur + xbase + (ky * CV_HW + kx) * CV_PSTR;
            // fragments of k-step ks+1 are requested before the MFMAs of k-step ks (register double buffer); the order is
            // pinned with sched_barrier: hipcc's own schedule issues each read right in front of its consumer and waits
            // lgkmcnt(0) every 4 MFMAs
            auto frag_load = [&](int ks, int s) __attribute__((always_inline)) {
                if (CV_ABL & 4) {
                    if (c == 0 && t == 0 && ks == 0) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) wf[0][m] = wf[1][m] = *reinterpret_cast<const cv_bf16x8*>(bbuf + wbase + m * 4096);
#pragma unroll
                        for (int n = 0; n < 2; ++n) xf[0][n] = xf[1][n] = *reinterpret_cast<const cv_bf16x8*>(abuf + xbase + n * 64);
                    }
                    return;
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    wf[s][m] = *reinterpret_cast<const cv_bf16x8*>(bbuf + bcur + ((wbase ^ unsigned(ks << 5)) + m * 4096));
#pragma unroll
                for (int n = 0; n < 2; ++n) xf[s][n] = *reinterpret_cast<const cv_bf16x8*>(xl + n * CV_HW * CV_PSTR + ks * 32);
            };
            frag_load(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                __builtin_amdgcn_sched_barrier(0);
                if (ks < 3) frag_load(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 1 && t >= 1 && t <= CV_ROUNDS && !(CV_ABL & 16)) {  // normalisation VALU work beside this k-step's MFMAs
                    // tie the piece to this point: without it the arithmetic (no side effects) is scheduled at the top of the
                    // stage, directly behind its global load and a vmcnt(0)
                    cv_u32x4& raw = rawq[(t - 1) & 1];
                    asm volatile("" : "+v"(raw));
                    a_write(t - 1, make_uint4(raw.x, raw.y, raw.z, raw.w), anext, true);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        if (CV_ABL & 2) asm volatile("" ::"v"(wf[ks & 1][m]), "v"(xf[ks & 1][n]));
                        else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][m], xf[ks & 1][n], acc[m][n], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this stage's LDS-DMA, halo request and halo store are done
            if (!(CV_ABL & 8)) __syncthreads();
            bcur ^= unsigned(BSZ);
        }
        acur ^= unsigned(CV_ASZ);
    }

    // ---- epilogue: + bias (+ residual) -> bf16, GroupNorm statistics of the stored values
    float* const sred = reinterpret_cast<float*>(smem);  // per-group (sum, sumsq) of this tile
    const bool want_stats = a.stats_out != nullptr;
    const int cg_out = a.cg_out;
    const int g_first = want_stats ? (ct * CT) / cg_out : 0;
    const int g_last = want_stats ? (ct * CT + CT - 1) / cg_out : 0;
    if (want_stats) {
        if (tid < 2 * (g_last - g_first + 1)) sred[tid] = 0.f;
        __syncthreads();
    }
    const float* const bias = a.bias ? a.bias + int64_t(b) * a.bias_bstride + ct * CT : nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int chl = w_m * MT * 32 + m * 32 + 4 * h;  // + 8 q + j : channel inside the tile
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
        float4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = bias ? *reinterpret_cast<const float4*>(bias + chl + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int y = ty * CV_TH + 2 * w_n + n, px = tx * CV_TW + l31;
            const int64_t o = ((int64_t(b) * H + y) * W + px) * a.cout + ct * CT + chl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[m][n][4 * q] + bq[q].x, acc[m][n][4 * q + 1] + bq[q].y, acc[m][n][4 * q + 2] + bq[q].z,
                              acc[m][n][4 * q + 3] + bq[q].w};
                if (a.residual) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(a.residual + o + 8 * q);
                    v[0] += __uint_as_float(rr.x << 16);
                    v[1] += __uint_as_float(rr.x & 0xFFFF0000u);
                    v[2] += __uint_as_float(rr.y << 16);
                    v[3] += __uint_as_float(rr.y & 0xFFFF0000u);
                }
                cv_bf16x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pk[j] = (__bf16)v[j];
                    const float vr = (float)pk[j];
                    gs[q] += vr;
                    gq[q] = fmaf(vr, vr, gq[q]);
                }
                *reinterpret_cast<cv_bf16x4*>(a.out + o + 8 * q) = pk;
            }
        }
        if (want_stats) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = gs[q], ss = gq[q];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) {
                    s += __shfl_xor(s, off);
                    ss += __shfl_xor(ss, off);
                }
                if (l31 == 0) {
                    const int gl = (ct * CT + chl + 8 * q) / cg_out - g_first;
                    atomicAdd(&sred[2 * gl], s);
                    atomicAdd(&sred[2 * gl + 1], ss);
                }
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid < 2 * (g_last - g_first + 1))
            atomicAdd(a.stats_out + (int64_t(b) * a.groups_out + g_first) * 2 + tid, double(sred[tid]));
    }
}

// OHWI (cout, 9, cin) bf16 -> stage-major images [cout tile][chunk][tap][row][8 granules of 8 channels], granule g of row r
// stored at position g ^ ((r >> 1) & 7): the 32-row fragment reads (ds_read_b128) of the kernel are then conflict free
__global__ void k_conv3x3_pack(const __bf16* __restrict__ w, int cout, int cin, int CT, __bf16* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one 16-byte granule of the output
    const int64_t total = int64_t(cout) * 9 * cin / 8;
    if (i >= total) return;
    const int nch = cin / CV_KC;
    const int gp = int(i & 7);
    int64_t r_ = i >> 3;
    const int row = int(r_ % CT);
    r_ /= CT;
    const int t = int(r_ % 9);
    r_ /= 9;
    const int c = int(r_ % nch);
    const int ct = int(r_ / nch);
    const int g = gp ^ ((row >> 1) & 7);
    const uint4 v = *reinterpret_cast<const uint4*>(w + (int64_t(ct * CT + row) * 9 + t) * cin + c * CV_KC + g * 8);
    reinterpret_cast<uint4*>(out)[i] = v;
}

template <int CT, int MODE, bool UPS>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
    constexpr int LDS = 2 * CV_ASZ + 2 * CT * 128 + 16;
    static bool configured = false;
    if (!configured) {
        XM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<CT, MODE, UPS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        configured = true;
    }
    const int grid = a.B * a.tiles_y * a.tiles_x * a.nct;
    hipLaunchKernelGGL((k_conv3x3<CT, MODE, UPS>), dim3(grid), dim3(512), LDS, s, a);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_conv3x3_cout_tile(int cout) { return cout % 256 == 0 ? 256 : (cout % 128 == 0 ? 128 : 0); }

extern "C" int xm3d_conv3x3_pack_weight(const void* w_ohwi, int cout, int cin, int cout_tile, void* packed, void* stream) {
    XM3D_REQUIRE(w_ohwi && packed, "conv3x3_pack_weight: null pointer");
    XM3D_REQUIRE(cin > 0 && cin % CV_KC == 0, "conv3x3_pack_weight: cin %d is not a multiple of %d", cin, CV_KC);
    XM3D_REQUIRE((cout_tile == 128 || cout_tile == 256) && cout > 0 && cout % cout_tile == 0,
                 "conv3x3_pack_weight: cout %d / tile %d unsupported", cout, cout_tile);
    const int64_t total = int64_t(cout) * 9 * cin / 8;
    hipLaunchKernelGGL(k_conv3x3_pack, dim3(unsigned((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16*>(w_ohwi), cout, cin, cout_tile, static_cast<__bf16*>(packed));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_conv3x3_nhwc(const void* x, int64_t B, int H, int W, int cin, const void* wpacked, int cout, int cout_tile,
                                 const double* gn_stats, const float* gamma, const float* beta, float eps, int groups, int act,
                                 const float* bias, int bias_bstride, const void* residual, void* out, double* stats_out,
                                 int groups_out, int upsample, void* stream) {
    XM3D_REQUIRE(x && wpacked && out, "conv3x3_nhwc: null pointer");
    XM3D_REQUIRE(B > 0 && B < 65536 && H > 0 && W > 0 && H % CV_TH == 0 && W % CV_TW == 0,
                 "conv3x3_nhwc: output %dx%d is not a multiple of the %dx%d pixel tile", H, W, CV_TH, CV_TW);
    XM3D_REQUIRE(cin > 0 && cin % CV_KC == 0, "conv3x3_nhwc: cin %d is not a multiple of %d", cin, CV_KC);
    XM3D_REQUIRE((cout_tile == 128 || cout_tile == 256) && cout > 0 && cout % cout_tile == 0, "conv3x3_nhwc: cout %d / tile %d unsupported",
                 cout, cout_tile);
    XM3D_REQUIRE(int64_t(H) * W * (cin > cout ? cin : cout) < (int64_t(1) << 31), "conv3x3_nhwc: image too large for 32-bit offsets");
    XM3D_REQUIRE(bias_bstride == 0 || bias_bstride == cout, "conv3x3_nhwc: bias_bstride must be 0 or cout");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                   reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gn_stats) | reinterpret_cast<uintptr_t>(stats_out)) & 15) == 0,
                 "conv3x3_nhwc: tensors must be 16-byte aligned");
    const bool gn = gn_stats != nullptr;
    if (gn) {
        XM3D_REQUIRE(gamma && beta && groups > 0 && cin % groups == 0 && (cin / groups) >= 4, "conv3x3_nhwc: GroupNorm needs gamma, beta and >= 4 channels per group");
        XM3D_REQUIRE(act == 1, "conv3x3_nhwc: the fused GroupNorm is followed by SiLU (act 1)");
        XM3D_REQUIRE(!upsample, "conv3x3_nhwc: upsample and GroupNorm cannot be combined");
    } else {
        XM3D_REQUIRE(act == 0, "conv3x3_nhwc: an activation needs the GroupNorm statistics");
    }
    if (upsample) XM3D_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_nhwc: upsample needs even output dims");
    if (stats_out)
        XM3D_REQUIRE(groups_out > 0 && cout % groups_out == 0 && (cout / groups_out) % 4 == 0 && cout_tile / (cout / groups_out) + 2 <= 128,
                     "conv3x3_nhwc: output statistics need a multiple of 4 channels per group (cout %d, groups %d)", cout, groups_out);
    ConvArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.wp = static_cast<const __bf16*>(wpacked);
    a.gn_stats = gn_stats;
    a.gamma = gamma;
    a.beta = beta;
    a.bias = bias;
    a.residual = static_cast<const __bf16*>(residual);
    a.out = static_cast<__bf16*>(out);
    a.stats_out = stats_out;
    a.B = int(B);
    a.H = H;
    a.W = W;
    a.cin = cin;
    a.cout = cout;
    a.groups_in = gn ? groups : 1;
    a.cg_in = gn ? cin / groups : cin;
    const int Hi = upsample ? H / 2 : H, Wi = upsample ? W / 2 : W;
    a.inv_cnt_in = gn ? 1.0 / (double(Hi) * Wi * (cin / groups)) : 0.0;
    a.eps = eps;
    a.bias_bstride = bias_bstride;
    a.groups_out = stats_out ? groups_out : 1;
    a.cg_out = stats_out ? cout / groups_out : cout;
    a.tiles_x = W / CV_TW;
    a.tiles_y = H / CV_TH;
    a.nct = cout / cout_tile;
    hipStream_t s = as_stream(stream);
    if (cout_tile == 256) {
        if (gn) return launch_conv<256, 2, false>(a, s);
        return upsample ? launch_conv<256, 0, true>(a, s) : launch_conv<256, 0, false>(a, s);
    }
    if (gn) return launch_conv<128, 2, false>(a, s);
    return upsample ? launch_conv<128, 0, true>(a, s) : launch_conv<128, 0, false>(a, s);
}
