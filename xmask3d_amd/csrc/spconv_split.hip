// Sparse convolution, producer / consumer form on the bf16 matrix cores with f32-split operands (algo 4).
//
//   out[o,:] = epi( sum_k in[nbr[k,o],:] @ W[k] )          same contract and tiled rulebook as k_spconv_tiles (spconv.hip)
//
// Why a second kernel: k_spconv_tiles (exact f32 MFMA) sat at 19 % of the f32 matrix peak with 55 % of its wave cycles
// parked - one workgroup barrier per (offset, channel chunk) step, 16-pair tiles dealt unevenly to four symmetric waves,
// every input row gathered once per 32 output channels - and f32 MFMA itself runs at 1/16 of the bf16 rate.  Here:
//   * every f32 operand is split into two bf16 terms, x = x_hi + x_lo (+ <= 2^-17 |x|), and a product is three bf16 MFMAs
//     with f32 accumulation: x_hi*w_hi + x_hi*w_lo + x_lo*w_hi (the dropped x_lo*w_lo and the split residuals are <= 2^-16
//     |x*w| each, random in sign; measured against the f64 oracle: 4.8e-6 of max|out| on the S1-full 96->96 layer, inside
//     the 2e-5 bound of the f32 kernel's own test).  Weights are split once when they are packed, activations on the fly.
//   * workgroup = one 256-row output tile x CTT = 16*NT output channels (96 on the dominant layers: every input row is
//     gathered ONCE, not 3x); CONSUMER waves each own 16-channel slices of the LDS accumulator (wave-private columns: no
//     atomics, fixed summation order -> bitwise reproducible); PRODUCER waves gather the packed (input row, output row)
//     pairs, split them and publish them in MFMA-fragment order in a two-slot LDS ring that all consumers read.
//   * work = CHUNKS of up to IPC 16-pair items of ONE (offset, channel chunk) group, one chunk per barrier interval: the
//     pairs of a chunk have distinct output rows, so a consumer reads all its accumulator rows up front and accumulates
//     straight onto them in the MFMA's C operand (D^T = W^T X^T, v_mfma_f32_16x16x32_bf16: a lane ends with 4 consecutive
//     output channels of one pair = one 16-byte LDS store); the next item's fragments are requested before the MFMAs.
//   * consumers stream weight fragments from L2 straight into registers (A operand, fragment-ordered by the packer), three
//     register sets, the set of chunk n + 2 requested at the top of interval n with hand-counted vmcnt waits;
//     producers get their memory-level parallelism from staggered wave groups (one register set per wave, plain waits).
// Measured (tools/spconv_stamps.py, s_memtime stamps): after these steps the kernel is instruction-issue bound (12 waves x
// ~200 instructions per interval); next lever = activations stored pre-split by the producing conv.
// Replaces ME.MinkowskiConvolution(+Transpose) + the BN/ReLU/residual tail of ME's BasicBlock like algo 3
// (models/modeling/meta_arch/mink_unet.py:47-109,118-178, resnet_base.py:64-96).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace xm3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SROWS = 256;  // output rows per workgroup (= the tiled rulebook's tile)

// Diagnostic build only (make EXTRA=-DXM3D_SPLIT_STAMPS, tools/spconv_stamps.py): per-wave s_memtime stamps of the first
// workgroups - when a wave finished its share of an interval and when the barrier released it.  The stamps go to a buffer
// of their own; no output value depends on them; the product build contains none of this.
#ifdef XM3D_SPLIT_STAMPS
constexpr int STAMP_WGS = 8, STAMP_WAVES = 16, STAMP_SLOTS = 256;
__device__ long long g_stamps[STAMP_WGS * STAMP_WAVES * STAMP_SLOTS];
#define XM3D_STAMP(idx)                                                                                         \
    do {                                                                                                        \
        if (lane == 0 && blockIdx.x < STAMP_WGS && blockIdx.y == 0 && blockIdx.z == 0 && (idx) < STAMP_SLOTS)   \
            g_stamps[(blockIdx.x * STAMP_WAVES + wave) * STAMP_SLOTS + (idx)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#else
#define XM3D_STAMP(idx) do { } while (0)
#endif
#define STAMP_LAST 255

// ------------------------------------------------------------------ weight packing (split to bf16 hi / lo)
// Wq as uint4[(((k * (cin/32) + sg) * (cout/16) + ng) * 2 + h) * 64 + lane] = 8 bf16: element j of lane (n16 = lane & 15,
// q = lane >> 4) is the hi (h = 0) / lo (h = 1) part of W[k][32 sg + 8 q + j][16 ng + n16] (k-slot (q, j) = channel 8 q + j of
// the 32-channel step, the same order in which the producers fetch a row: 8 consecutive channels per lane).
__global__ void k_pack_weight_split(const float* __restrict__ W, int K, int cin, int cout, __bf16* __restrict__ Wq) {
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one thread per (block, h = both, lane, j)
    const int64_t total = int64_t(K) * cin * cout;
    if (e >= total) return;
    const int j = int(e & 7);
    const int lane = int((e >> 3) & 63);
    const int64_t blk = e >> 9;  // (k, sg, ng)
    const int NG = cout / 16, SG = cin / 32;
    const int ng = int(blk % NG);
    const int sg = int((blk / NG) % SG);
    const int k = int(blk / (int64_t(NG) * SG));
    const int ci = 32 * sg + 8 * (lane >> 4) + j;
    const int co = 16 * ng + (lane & 15);
    const float w = W[(int64_t(k) * cin + ci) * cout + co];
    const __bf16 hi = (__bf16)w;
    const __bf16 lo = (__bf16)(w - (float)hi);
    const int64_t base = (blk * 2) * 512 + lane * 8 + j;
    Wq[base] = hi;
    Wq[base + 512] = lo;
}

template <int NT, int CC, int IPC, int NPL = 2>
struct __attribute__((aligned(16))) SplitLds {
    float acc[SROWS][16 * NT + 4];        // padded rows: 16-byte aligned, breaks the bank stride
    uint4 stage[2][IPC][CC / 32][NPL][64];  // [slot][item][k-step][hi/lo][lane] bf16x8 fragments (B operand); NPL = 1: bf16 form, hi only
    int dst[2][IPC][16];                  // local output row of each pair slot (-1 = padding)
    int nzk[128], nzc[128];               // non-empty offsets of this tile (compacted): offset slot, pair count
    int nnz;
};

struct ChunkIt {
    int kk, c, sub;  // compacted offset entry, channel chunk, sub-chunk (IPC items each) of that group
};

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// one 1-KiB weight fragment block: lane-linear 16-byte loads, scalar base + 32-bit lane offset + immediate.  Inline asm on
// purpose: hipcc neither counts nor waits for these loads, the consumer loop counts them itself (see wait_weights).
template <int IMM>
__device__ __forceinline__ void load_frag(uint4& dst, const uint4* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}

// NT = 16-channel output tiles per workgroup, NTW of them per consumer wave (NT / NTW consumers), CC = input channels per
// item, IPC = items per chunk (= barrier interval), NG x NPW producer waves (NG staggered groups of NPW waves).
// A CHUNK is up to IPC 16-pair items of ONE (offset, channel chunk) group: all its pairs have distinct output rows (one
// offset), so the accumulator updates inside a chunk are independent, and its weights are one fragment set.
// PRE: the input also exists pre-split (in_hi / in_lo bf16 planes, written by the epilogue of the conv that produced it): the
// producers then only move fragments (two 16-byte loads + two LDS stores per 32-channel step instead of 24 conversion VALU ops).
// BF (needs PRE): the plain-bf16 form of the bf16 configuration - activations exist ONLY as one bf16 plane (in_hi / out_hi; residual is a
// bf16 plane too), a product is ONE MFMA on the hi parts: a third of the matrix work, half the gathered, staged and written bytes.
// G: chunks per barrier interval.  A chunk stays "up to IPC items of ONE offset" (one weight set, distinct output rows), but the ring slot
// holds G of them and the workgroup barrier comes once per G chunks: at the rulebook densities of the path an offset has 2 - 3 items per
// 256-row tile, so with one chunk per interval a tile was ~27 latency-bound intervals of ~3 k cycles with ~430 cycles of matrix work each.
template <int NT, int NTW, int CC, int IPC, int NG, int NPW, bool PRE, bool BF = false, int G = 1>
__global__ __launch_bounds__(64 * (NT / NTW + NG * NPW)) void k_spconv_split(
    const float* __restrict__ in, const __bf16* __restrict__ in_hi, const __bf16* __restrict__ in_lo, __bf16* __restrict__ out_hi,
    __bf16* __restrict__ out_lo, int cin, const uint4* __restrict__ Wq, int K, int cout, const int32_t* __restrict__ tsrc,
    const uint8_t* __restrict__ tdst, const int32_t* __restrict__ tcnt, const int32_t* __restrict__ order, int64_t n_out,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ residual, int relu,
    float* __restrict__ out, int ksplit, float* __restrict__ slab, int ntiles) {
    static_assert(!BF || PRE, "the bf16 form reads a bf16 plane");
    constexpr int NPL = BF ? 1 : 2;            // operand planes (hi / lo)
    __shared__ SplitLds<NT, CC, IPC * G, NPL> lds;
    constexpr int S = CC / 32;                 // k-steps per item
    constexpr int CTT = 16 * NT, ACCLD = CTT + 4;
    constexpr int NCONS = NT / NTW;
    constexpr int NP = NG * NPW;
    constexpr int NTHREADS = 64 * (NCONS + NP);
    constexpr int IPP = IPC / NPW;             // items per producer wave per chunk
    constexpr int WD = 3;                      // weight fragment sets per consumer (loaded two chunks ahead)
    constexpr int WLOADS = NTW * S * NPL;      // 1-KiB loads per weight set
    static_assert(IPC % NPW == 0 && NT % NTW == 0 && NG >= 2, "IPC must be a multiple of NPW, NT of NTW");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, q = lane >> 4;
    // XCD-aware tile order (speed only): blocks b and b + 8 share an XCD's L2, give each XCD a contiguous run of row tiles
    // (spatial neighbours gather the same input rows).  Bijective for any ntiles.
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, qd = ntiles >> 3, r = ntiles & 7;
        tile = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (b >> 3);
    }
    const int ct0 = blockIdx.y * CTT;
    const int kz = blockIdx.z;  // split-K: this workgroup handles offsets kz, kz + ksplit, ...
    const int nk = (K - kz + ksplit - 1) / ksplit;
    const int nchunk = cin / CC;
    const int SG = cin / 32, NGC = cout / 16;  // 32-channel k-steps / 16-channel output tiles of the layer
    const int64_t tbase = int64_t(tile) * K;

    XM3D_STAMP(0);
    for (int c = tid; c < SROWS * ACCLD / 4; c += NTHREADS) reinterpret_cast<f32x4*>(&lds.acc[0][0])[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The non-empty offsets of this tile, compacted once (wave 0) and then held by every wave in two registers (lane r:
    // offset slot and pair count of the r-th non-empty offset, r < 64; the next 64 in the second pair): the chunk iterator
    // below is branch-free scalar code on v_readlane values.
    if (wave == 0) {
        int base = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = 64 * h + lane;
            const int cn = kk < nk ? tcnt[tbase + kz + kk * ksplit] : 0;
            const unsigned long long m = __ballot(cn > 0);
            if (cn > 0) {
                const int r = base + __popcll(m & ((1ull << lane) - 1ull));
                lds.nzk[r] = kk;
                lds.nzc[r] = cn;
            }
            base += __popcll(m);
        }
        if (lane == 0) lds.nnz = base;
    }
    __syncthreads();  // accumulator zeroed, offset list published
    const int nnz = __builtin_amdgcn_readfirstlane(lds.nnz);
    const int zk0 = lane < nnz ? lds.nzk[lane] : 0, zc0 = lane < nnz ? lds.nzc[lane] : 0;
    const int zk1 = 64 + lane < nnz ? lds.nzk[64 + lane] : 0, zc1 = 64 + lane < nnz ? lds.nzc[64 + lane] : 0;
    // kk indexes the compacted list from here on; entries past its end read as 0 pairs of offset slot 0
    auto cntk = [&](int kk) __attribute__((always_inline)) {
        return kk < 64 ? __builtin_amdgcn_readlane(zc0, kk) : __builtin_amdgcn_readlane(zc1, (kk - 64) & 63);
    };
    auto offk = [&](int kk) __attribute__((always_inline)) {  // offset index (into K) of compacted entry kk
        return kz + (kk < 64 ? __builtin_amdgcn_readlane(zk0, kk) : __builtin_amdgcn_readlane(zk1, (kk - 64) & 63)) * ksplit;
    };
    auto nsub = [&](int kk) __attribute__((always_inline)) { return (((cntk(kk) + 15) >> 4) + IPC - 1) / IPC; };
    auto next = [&](ChunkIt r) __attribute__((always_inline)) {  // branch-free; past the end kk keeps growing (empty chunks)
        const int kk = r.kk < 127 ? r.kk : 127;
        const bool last_s = r.sub + 1 >= nsub(kk);
        const bool last_c = r.c + 1 == nchunk;
        ChunkIt o;
        o.sub = last_s ? 0 : r.sub + 1;
        o.c = last_s ? (last_c ? 0 : r.c + 1) : r.c;
        o.kk = (last_s && last_c) ? kk + 1 : kk;
        return o;
    };
    // number of chunks = barrier intervals (uniform over the workgroup), padded to a multiple of the weight rotation
    int nch = 0;
    for (int kk = 0; kk < nnz; ++kk) nch += nsub(kk) * nchunk;
    nch = (nch + WD * G - 1) / (WD * G) * (WD * G);  // whole weight rotations and whole barrier intervals
    const ChunkIt it0{0, 0, 0};
    XM3D_STAMP(1);

    if (wave >= NCONS) {
        // ============================================================ producer
        // Memory-level parallelism comes from WAVES, not from a deep register pipeline inside one wave (hipcc rotates
        // loop-carried load registers with copies at the loop latch, and a copy of an in-flight register drains vmcnt(0)):
        // NG groups of NPW waves; chunk n is gathered, split and published by group n % NG, each of its waves doing
        // IPC / NPW items with ONE register set and plain full waits.  Per group and period of NG intervals: publish my
        // chunk + issue the pair-index loads of my next one (phase 0), issue its row gathers (phase 1), then NG - 2 idle
        // intervals while they land.
        const int p = wave - NCONS;
        const int g = __builtin_amdgcn_readfirstlane(p / NPW), m = __builtin_amdgcn_readfirstlane(p % NPW);
        int a_src[G * IPP], a_dst[G * IPP], a_c[G];
        bool a_ok[G * IPP];
        f32x4 rows[G * IPP][NPL * S];
        ChunkIt it_mine = it0;  // my group's next interval (G chunks)
        for (int i = 0; i < g * G; ++i) it_mine = next(it_mine);

        auto stage_idx = [&]() __attribute__((always_inline)) {  // my items of the G chunks at it_mine; it_mine += NG intervals
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                const ChunkIt t = it_mine;
                const int kk = t.kk < 127 ? t.kk : 127;
                const int cn = cntk(kk);  // 0 past the end
                a_c[gi] = t.c;
                const int64_t obase = (tbase + offk(kk)) * SROWS;
#pragma unroll
                for (int u = 0; u < IPP; ++u) {
                    const int pp = ((t.sub * IPC + m * IPP + u) << 4) + p16;
                    a_ok[gi * IPP + u] = pp < cn;
                    // unconditional loads (clamped to a valid slot): no divergent branch, the predicate is applied on use
                    const int64_t o = obase + (a_ok[gi * IPP + u] ? pp : 0);
                    a_src[gi * IPP + u] = tsrc[o];
                    a_dst[gi * IPP + u] = int(tdst[o]);
                }
                it_mine = next(it_mine);
            }
            for (int i = 0; i < (NG - 1) * G; ++i) it_mine = next(it_mine);
        };
        auto stage_rows = [&]() __attribute__((always_inline)) {  // indices -> row gathers in flight
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int u = 0; u < IPP; ++u) {
                    const int x = gi * IPP + u;
                    // padding slots read row 0 (valid memory); their MFMA columns are never accumulated (dst = -1)
                    const int64_t eo = int64_t(a_ok[x] ? a_src[x] : 0) * cin + a_c[gi] * CC + 8 * q;  // this lane's 8 channels per step
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if constexpr (BF) {  // the one bf16 plane
                            rows[x][s] = *reinterpret_cast<const f32x4*>(in_hi + eo + 32 * s);
                        } else if constexpr (PRE) {  // ready-made bf16 hi / lo fragments
                            rows[x][2 * s] = *reinterpret_cast<const f32x4*>(in_hi + eo + 32 * s);
                            rows[x][2 * s + 1] = *reinterpret_cast<const f32x4*>(in_lo + eo + 32 * s);
                        } else {
                            rows[x][2 * s] = *reinterpret_cast<const f32x4*>(in + eo + 32 * s);
                            rows[x][2 * s + 1] = *reinterpret_cast<const f32x4*>(in + eo + 32 * s + 4);
                        }
                    }
                }
        };
        auto stage_write = [&](int slot) __attribute__((always_inline)) {  // rows -> bf16 hi / lo fragments in the ring slot
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int u = 0; u < IPP; ++u) {
                    const int x = gi * IPP + u;
                    const int e = gi * IPC + m * IPP + u;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if constexpr (BF) {
                            lds.stage[slot][e][s][0][lane] = __builtin_bit_cast(uint4, rows[x][s]);
                        } else if constexpr (PRE) {
                            lds.stage[slot][e][s][0][lane] = __builtin_bit_cast(uint4, rows[x][2 * s]);
                            lds.stage[slot][e][s][1][lane] = __builtin_bit_cast(uint4, rows[x][2 * s + 1]);
                        } else {
                            bf16x8 hi, lo;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float xv = rows[x][2 * s + (j >> 2)][j & 3];
                                hi[j] = (__bf16)xv;
                                lo[j] = (__bf16)(xv - (float)hi[j]);
                            }
                            lds.stage[slot][e][s][0][lane] = __builtin_bit_cast(uint4, hi);
                            lds.stage[slot][e][s][1][lane] = __builtin_bit_cast(uint4, lo);
                        }
                    }
                    if (q == 0) lds.dst[slot][e][p16] = a_ok[x] ? a_dst[x] : -1;
                }
        };
        // interval 0 is published here by group 0 (its latency is paid once); every group then has its first loop interval
        // (interval NG for group 0, interval g for the others) in flight
        stage_idx();
        stage_rows();
        if (g == 0) {
            stage_write(0);
            stage_idx();
            stage_rows();
        }
        XM3D_STAMP(2);
        __syncthreads();
        XM3D_STAMP(3);
        int ph = (NG + 1 - g) % NG;  // phase of interval 0: interval iv publishes interval iv + 1 = group (iv + 1) % NG
        for (int iv = 0; iv < nch / G; ++iv) {
            if (ph == 0) {
                stage_write((iv + 1) & 1);  // (past the end: all-padding chunks nobody reads)
                stage_idx();                // my next interval: iv + 1 + NG
            } else if (ph == 1 % NG) {
                stage_rows();
            }
            ph = ph + 1 == NG ? 0 : ph + 1;
            XM3D_STAMP(4 + 2 * iv);
            __syncthreads();
            XM3D_STAMP(5 + 2 * iv);
        }
    } else {
        // ============================================================ consumer: output channels ct0 + 16 NTW wave .. + 16 NTW
        // Weight fragments: WD = 3 register sets; the set of chunk ch + 2 is requested at the top of interval ch with
        // hand-written loads hipcc does not track, so nothing drains them early; before a set is used the wave waits with a
        // COUNTED vmcnt that leaves the two younger sets in flight (loads retire in order, these are the wave's only
        // vector-memory operations inside the loop).
        const int ng0 = ct0 / 16 + wave * NTW;
        uint4 W[WD][NTW][S][NPL];
        const unsigned voff = lane * 16;
        auto load_w = [&](auto SETc, ChunkIt t) __attribute__((always_inline)) {
            constexpr int SET = decltype(SETc)::value;
            const int kk = t.kk < 127 ? t.kk : 127;  // past the end: entry reads as offset slot 0 -> a valid (unused) set
            const int k = offk(kk);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint4* base = Wq + (((int64_t(k) * SG + (t.c * S + s)) * NGC + ng0) * 2) * 64;
                // wave-uniform by construction; readfirstlane makes that provable so the base can live in SGPRs.  (Widen each
                // half as UNSIGNED: an int half or-ed into a 64-bit value sign-extends and corrupts the address.)
                const uint64_t a64 = reinterpret_cast<uintptr_t>(base);
                const uint32_t alo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(a64))));
                const uint32_t ahi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(a64 >> 32))));
                const uint4* sb = reinterpret_cast<const uint4*>((uint64_t(ahi) << 32) | uint64_t(alo));
                load_frag<0>(W[SET][0][s][0], sb, voff);
                if constexpr (!BF) load_frag<1024>(W[SET][0][s][NPL - 1], sb, voff);
                if constexpr (NTW > 1) {
                    load_frag<2048>(W[SET][NTW > 1 ? 1 : 0][s][0], sb, voff);
                    if constexpr (!BF) load_frag<3072>(W[SET][NTW > 1 ? 1 : 0][s][NPL - 1], sb, voff);
                }
                static_assert(NTW <= 2, "immediate offsets cover two 16-channel tiles per consumer");
            }
        };
        auto wait_weights = [&](auto SETc) __attribute__((always_inline)) {  // set SET landed; two younger sets stay in flight
            constexpr int SET = decltype(SETc)::value;
            (void)SET;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WD - 1) * WLOADS) : "memory");
            __builtin_amdgcn_sched_barrier(0);  // no MFMA (a register-only instruction) may be scheduled above the wait
        };
        ChunkIt it = it0, it_w = it0;  // current chunk; chunk whose weights are requested next
        load_w(std::integral_constant<int, 0>{}, it_w);
        it_w = next(it_w);
        load_w(std::integral_constant<int, 1>{}, it_w);
        it_w = next(it_w);
        XM3D_STAMP(2);
        __syncthreads();  // chunk 0 published
        XM3D_STAMP(3);
        for (int ch0 = 0; ch0 < nch; ch0 += WD) {
            static_for<WD>([&](auto PHc) __attribute__((always_inline)) {
                constexpr int PH = decltype(PHc)::value;
                const int ch = ch0 + PH;
                const int slot = (ch / G) & 1, ib = (ch % G) * IPC;  // ring slot of the interval, first item of this chunk in it
                load_w(std::integral_constant<int, (PH + 2) % WD>{}, it_w);  // chunk ch + 2 (its set was last used by chunk ch - 1)
                it_w = next(it_w);
                const int kk = it.kk < 127 ? it.kk : 127;
                const int nitems = min(IPC, ((cntk(kk) + 15) >> 4) - it.sub * IPC);  // <= 0 past the end
                it = next(it);
                // every accumulator row of the chunk is distinct: read them all up front, together with the B fragments of
                // the first item; then per item: request the next item's fragments, MFMAs, add + store
                int drow[IPC];
                f32x4 av[IPC][NTW];
#pragma unroll
                for (int e = 0; e < IPC; ++e) {
                    drow[e] = lds.dst[slot][ib + e][p16];
                    if (e >= nitems) drow[e] = -1;
                }
                uint4 B[2][S][NPL];
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) B[0][s][pl] = lds.stage[slot][ib][s][pl][lane];
#pragma unroll
                for (int e = 0; e < IPC; ++e)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        av[e][t] = *reinterpret_cast<const f32x4*>(&lds.acc[drow[e] < 0 ? 0 : drow[e]][16 * (wave * NTW + t) + 4 * q]);
                wait_weights(PHc);
#pragma unroll
                for (int e = 0; e < IPC; ++e) {
                    if (e + 1 < IPC) {
#pragma unroll
                        for (int s = 0; s < S; ++s)
#pragma unroll
                            for (int pl = 0; pl < NPL; ++pl) B[(e + 1) & 1][s][pl] = lds.stage[slot][ib + e + 1][s][pl][lane];
                    }
                    if (e < nitems) {
                        f32x4 d[NTW];
#pragma unroll
                        for (int t = 0; t < NTW; ++t) d[t] = av[e][t];  // accumulate straight onto the row's current value
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const bf16x8 bhi = __builtin_bit_cast(bf16x8, B[e & 1][s][0]);
                            const bf16x8 blo = __builtin_bit_cast(bf16x8, B[e & 1][s][NPL - 1]);
#pragma unroll
                            for (int t = 0; t < NTW; ++t) {
                                const bf16x8 ahi = __builtin_bit_cast(bf16x8, W[PH][t][s][0]);
                                if constexpr (!BF) {
                                    const bf16x8 alo = __builtin_bit_cast(bf16x8, W[PH][t][s][NPL - 1]);
                                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bhi, d[t], 0, 0, 0);
                                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, blo, d[t], 0, 0, 0);
                                }
                                d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bhi, d[t], 0, 0, 0);
                            }
                        }
                        if (drow[e] >= 0) {  // d[r] = pair p16, channel 16 (wave NTW + t) + 4 q + r
#pragma unroll
                            for (int t = 0; t < NTW; ++t)
                                *reinterpret_cast<f32x4*>(&lds.acc[drow[e]][16 * (wave * NTW + t) + 4 * q]) = d[t];
                        }
                    }
                }
                if (G == 1 || ch % G == G - 1) {  // the interval's last chunk: hand the ring slot back (workgroup-uniform)
                    XM3D_STAMP(4 + 2 * (ch / G));
                    __syncthreads();
                    XM3D_STAMP(5 + 2 * (ch / G));
                }
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two weight sets requested past the end
    }
    // ================================================================ epilogue (all waves): CTT / 4 lanes x float4 per row
    constexpr int LPR = CTT / 4;
    for (int e = tid; e < SROWS * LPR; e += NTHREADS) {
        const int lr = e / LPR, c4 = (e % LPR) * 4;
        const int64_t sl = int64_t(tile) * SROWS + lr;
        if (sl >= n_out) continue;
        const int64_t grow = order ? order[sl] : sl;
        f32x4 v = *reinterpret_cast<const f32x4*>(&lds.acc[lr][c4]);
        const int c = ct0 + c4;
        if (ksplit > 1) {  // raw partial sum; k_slab_reduce applies the epilogue
            *reinterpret_cast<f32x4*>(slab + (int64_t(kz) * n_out + grow) * cout + c) = v;
            continue;
        }
        if (scale) v *= *reinterpret_cast<const f32x4*>(scale + c);
        if (shift) v += *reinterpret_cast<const f32x4*>(shift + c);
        if constexpr (BF) {  // the residual is a bf16 plane, the result one bf16 row segment (8 bytes)
            typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
            if (residual) {
                const bf16x4e r = *reinterpret_cast<const bf16x4e*>(reinterpret_cast<const __bf16*>(residual) + grow * cout + c);
                v[0] += float(r[0]), v[1] += float(r[1]), v[2] += float(r[2]), v[3] += float(r[3]);
            }
            bf16x4e o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)(relu ? fmaxf(v[i], 0.f) : v[i]);
            *reinterpret_cast<bf16x4e*>(out_hi + grow * cout + c) = o;
            continue;
        }
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + grow * cout + c);
        if (relu) {
            v[0] = fmaxf(v[0], 0.f);
            v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f);
            v[3] = fmaxf(v[3], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + grow * cout + c) = v;
        if (out_hi) store_split4(out_hi + grow * cout + c, out_lo + grow * cout + c, v);  // pre-split copy for the next conv
    }
    XM3D_STAMP(STAMP_LAST);
}

}  // namespace xm3d

using namespace xm3d;

#ifdef XM3D_SPLIT_STAMPS
extern "C" int xm3d_debug_read_stamps(long long* host, int64_t n) {  // diagnostic build only
    XM3D_HIP(hipDeviceSynchronize());
    XM3D_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), size_t(n) * sizeof(long long)));
    return XM3D_OK;
}
extern "C" int xm3d_debug_clear_stamps() {
    static long long zeros[STAMP_WGS * STAMP_WAVES * STAMP_SLOTS];
    XM3D_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros)));
    return XM3D_OK;
}
#endif

extern "C" int xm3d_spconv_pack_weight_split(const float* W, int32_t K, int32_t cin, int32_t cout, void* Wq, void* stream) {
    XM3D_REQUIRE(W && Wq && K >= 1, "pack_weight_split: bad args");
    XM3D_REQUIRE(cin % 32 == 0 && cout % 16 == 0, "pack_weight_split: cin=%d must be a multiple of 32, cout=%d of 16", cin, cout);
    const int64_t total = int64_t(K) * cin * cout;
    hipLaunchKernelGGL(k_pack_weight_split, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), W, K, cin, cout,
                       static_cast<__bf16*>(Wq));
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

// output channels per workgroup of the split kernel for a layer with `cout` channels
extern "C" int xm3d_spconv_split_channels(int32_t cout) { return cout % 96 == 0 ? 96 : (cout % 64 == 0 ? 64 : 32); }

static int spconv_fwd_split_impl(const float* in, const void* in_split, int64_t n_in, int32_t cin, const void* Wq, int32_t K, int32_t cout,
                                 const int32_t* tsrc, const uint8_t* tdst, const int32_t* tcnt, const int32_t* order, int64_t n_out,
                                 const float* scale, const float* shift, const float* residual, int32_t relu, float* out,
                                 void* out_split, int32_t ksplit, float* slab, void* stream) {
    XM3D_REQUIRE(n_in >= 0 && n_out >= 0 && K >= 1, "spconv_fwd_split: bad sizes");
    XM3D_REQUIRE(cin % 32 == 0 && cout % 32 == 0 && cin >= 32, "spconv_fwd_split: cin=%d cout=%d must be multiples of 32", cin, cout);
    if (n_out == 0) return XM3D_OK;
    XM3D_REQUIRE(in && Wq && tsrc && tdst && tcnt && out, "spconv_fwd_split: null pointer");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(Wq) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
                   reinterpret_cast<uintptr_t>(in_split) | reinterpret_cast<uintptr_t>(out_split)) & 15) == 0,
                 "spconv_fwd_split: tensors must be 16-byte aligned");
    XM3D_REQUIRE(ksplit >= 1 && ksplit <= K && (ksplit == 1 || slab), "spconv_fwd_split: ksplit=%d needs 1..K and a slab", ksplit);
    XM3D_REQUIRE(K <= 128, "spconv_fwd_split: K=%d > 128", K);
    XM3D_REQUIRE((!in_split || (n_in * int64_t(cin)) % 8 == 0) && (!out_split || (n_out * int64_t(cout)) % 8 == 0),
                 "spconv_fwd_split: split planes must keep 16-byte alignment (rows x channels a multiple of 8)");
    hipStream_t s = as_stream(stream);
    const int ntiles = int((n_out + SROWS - 1) / SROWS);
    const int ctt = xm3d_spconv_split_channels(cout);
    const int cc = (cin % 64 == 0) ? 64 : (cin % 96 == 0 ? 96 : 32);
    dim3 grid(ntiles, cout / ctt, ksplit);
    const uint4* wq = static_cast<const uint4*>(Wq);
    // split planes: (2, N, C) bf16 = hi plane followed by lo plane
    const __bf16* in_hi = static_cast<const __bf16*>(in_split);
    const __bf16* in_lo = in_hi ? in_hi + n_in * int64_t(cin) : nullptr;
    __bf16* o_hi = static_cast<__bf16*>(out_split);
    __bf16* o_lo = o_hi ? o_hi + n_out * int64_t(cout) : nullptr;
    __bf16* k_hi = ksplit > 1 ? nullptr : o_hi;  // with split-K the reducer writes the split copy
    __bf16* k_lo = ksplit > 1 ? nullptr : o_lo;
#define XM3D_SPLIT_P(NT, NTW, CC, IPC, NG, NPW, PRE)                                                                                    \
    hipLaunchKernelGGL((k_spconv_split<NT, NTW, CC, IPC, NG, NPW, PRE>), grid, dim3(64 * (NT / NTW + NG * NPW)), 0, s, in, in_hi, in_lo, \
                       k_hi, k_lo, cin, wq, K, cout, tsrc, tdst, tcnt, order, n_out, scale, shift, residual, relu, out, ksplit, slab,    \
                       ntiles)
#define XM3D_SPLIT(NT, NTW, CC, IPC, NG, NPW)                     \
    do {                                                          \
        if (in_hi) XM3D_SPLIT_P(NT, NTW, CC, IPC, NG, NPW, true); \
        else XM3D_SPLIT_P(NT, NTW, CC, IPC, NG, NPW, false);      \
    } while (0)
    // (NT, CC) -> items per interval sized so that accumulator + two ring slots fit the 160 KiB LDS
    // wave split per shape (tools/spconv_bench.py, tools/spconv_stamps.py): one 16-channel tile per consumer wave, three
    // staggered groups of two producer waves; XM3D_SPLIT_VARIANT=2 selects two tiles per consumer (experiments)
    static const int variant = [] {
        const char* e = getenv("XM3D_SPLIT_VARIANT");
        return e ? atoi(e) : 0;
    }();
    if (ctt == 96) {
        if (cc == 96) {
            if (variant == 2) XM3D_SPLIT(6, 2, 96, 4, 2, 2);
            else XM3D_SPLIT(6, 1, 96, 4, 3, 2);
        } else if (cc == 64) XM3D_SPLIT(6, 1, 64, 4, 3, 2);
        else XM3D_SPLIT(6, 1, 32, 8, 3, 2);
    } else if (ctt == 64) {
        if (cc == 96) XM3D_SPLIT(4, 1, 96, 4, 3, 2);
        else if (cc == 64) XM3D_SPLIT(4, 1, 64, 8, 3, 2);
        else XM3D_SPLIT(4, 1, 32, 8, 3, 2);
    } else {
        if (cc == 96) XM3D_SPLIT(2, 1, 96, 2, 3, 1);  // <= 80 KiB of LDS: two workgroups per CU
        else if (cc == 64) XM3D_SPLIT(2, 1, 64, 4, 3, 1);
        else XM3D_SPLIT(2, 1, 32, 8, 3, 1);
    }
#undef XM3D_SPLIT
#undef XM3D_SPLIT_P
    if (ksplit > 1) launch_slab_reduce(slab, ksplit, n_out, cout, scale, shift, residual, relu, out, s, o_hi, o_lo);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_spconv_fwd_split(const float* in, int64_t n_in, int32_t cin, const void* Wq, int32_t K, int32_t cout,
                                     const int32_t* tsrc, const uint8_t* tdst, const int32_t* tcnt, const int32_t* order,
                                     int64_t n_out, const float* scale, const float* shift, const float* residual,
                                     int32_t relu, float* out, int32_t ksplit, float* slab, void* stream) {
    return spconv_fwd_split_impl(in, nullptr, n_in, cin, Wq, K, cout, tsrc, tdst, tcnt, order, n_out, scale, shift, residual, relu, out,
                                 nullptr, ksplit, slab, stream);
}

extern "C" int xm3d_spconv_fwd_split2(const float* in, const void* in_split, int64_t n_in, int32_t cin, const void* Wq, int32_t K,
                                      int32_t cout, const int32_t* tsrc, const uint8_t* tdst, const int32_t* tcnt, const int32_t* order,
                                      int64_t n_out, const float* scale, const float* shift, const float* residual, int32_t relu,
                                      float* out, void* out_split, int32_t ksplit, float* slab, void* stream) {
    return spconv_fwd_split_impl(in, in_split, n_in, cin, Wq, K, cout, tsrc, tdst, tcnt, order, n_out, scale, shift, residual, relu, out,
                                 out_split, ksplit, slab, stream);
}

// ---- the plain-bf16 form (BF): the bf16 configuration's sparse convolution.  Activations are ONE bf16 plane (n, C), the product one
// MFMA per k-step on bf16(w) (the hi plane of the split weight image), f32 accumulation in LDS as above, bf16 rows out:
//     out = bf16( relu( scale * sum_k in[nbr[k]] @ bf16(W[k]) + shift + residual ) )
// a third of the matrix work and half the gathered / staged / written bytes of the split form; same rulebook, same order of the
// additions (bit-reproducible).  Accuracy: bf16 operands and bf16 activations between the layers, ~1e-2 at the end of MinkUNet34C -
// the level of the bf16 dense branch it feeds, NOT north_star's 1e-3 (that is the f32 / split form, the fp32 configuration's).
extern "C" int xm3d_spconv_fwd_bf16(const void* in, int64_t n_in, int32_t cin, const void* Wq, int32_t K, int32_t cout, const int32_t* tsrc,
                                    const uint8_t* tdst, const int32_t* tcnt, const int32_t* order, int64_t n_out, const float* scale,
                                    const float* shift, const void* residual, int32_t relu, void* out, int32_t ksplit, float* slab, void* stream) {
    XM3D_REQUIRE(n_in >= 0 && n_out >= 0 && K >= 1, "spconv_fwd_bf16: bad sizes");
    XM3D_REQUIRE(cin % 32 == 0 && cout % 32 == 0 && cin >= 32, "spconv_fwd_bf16: cin=%d cout=%d must be multiples of 32", cin, cout);
    if (n_out == 0) return XM3D_OK;
    XM3D_REQUIRE(in && Wq && tsrc && tdst && tcnt && out, "spconv_fwd_bf16: null pointer");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(Wq) | reinterpret_cast<uintptr_t>(residual) |
                   reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0,
                 "spconv_fwd_bf16: tensors must be 16-byte aligned");
    XM3D_REQUIRE(ksplit >= 1 && ksplit <= K && (ksplit == 1 || slab), "spconv_fwd_bf16: ksplit=%d needs 1..K and a slab", ksplit);
    XM3D_REQUIRE(K <= 128, "spconv_fwd_bf16: K=%d > 128", K);
    hipStream_t s = as_stream(stream);
    const int ntiles = int((n_out + SROWS - 1) / SROWS);
    const int ctt = xm3d_spconv_split_channels(cout);
    const int cc = (cin % 64 == 0) ? 64 : (cin % 96 == 0 ? 96 : 32);
    dim3 grid(ntiles, cout / ctt, ksplit);
    const uint4* wq = static_cast<const uint4*>(Wq);
    const __bf16* in_b = static_cast<const __bf16*>(in);
    __bf16* o_b = ksplit > 1 ? nullptr : static_cast<__bf16*>(out);
    const float* res = static_cast<const float*>(residual);  // a bf16 plane (the kernel casts back)
#define XM3D_BF(NT, NTW, CC, IPC, NG, NPW, G_)                                                                                                   \
    hipLaunchKernelGGL((k_spconv_split<NT, NTW, CC, IPC, NG, NPW, true, true, G_>), grid, dim3(64 * (NT / NTW + NG * NPW)), 0, s, nullptr, in_b, \
                       nullptr, o_b, nullptr, cin, wq, K, cout, tsrc, tdst, tcnt, order, n_out, scale, shift, res, relu, nullptr, ksplit, slab,     \
                       ntiles)
    // One chunk per barrier interval (G = 1).  G = 2 - 4 chunks per interval were measured and REJECTED (profiles/r04_spconv_bench.log: S1-full
    // 96 -> 96 90.7 us against 84.2 us, every layer 0 - 8 % slower): the barrier is not what an interval waits for - the chain {pair indices ->
    // row gather -> LDS publish} of the producer groups and the consumers' {row ids -> accumulator rows -> MFMA chain -> store} are
    // per-chunk latencies that more chunks per interval do not shorten.  (The template keeps G for the next attempt.)
    if (ctt == 96) {
        if (cc == 96) XM3D_BF(6, 1, 96, 4, 3, 2, 1);
        else if (cc == 64) XM3D_BF(6, 1, 64, 4, 3, 2, 1);
        else XM3D_BF(6, 1, 32, 8, 3, 2, 1);
    } else if (ctt == 64) {
        if (cc == 96) XM3D_BF(4, 1, 96, 4, 3, 2, 1);
        else if (cc == 64) XM3D_BF(4, 1, 64, 8, 3, 2, 1);
        else XM3D_BF(4, 1, 32, 8, 3, 2, 1);
    } else {
        if (cc == 96) XM3D_BF(2, 1, 96, 2, 3, 1, 1);
        else if (cc == 64) XM3D_BF(2, 1, 64, 4, 3, 1, 1);
        else XM3D_BF(2, 1, 32, 8, 3, 1, 1);
    }
#undef XM3D_BF
    if (ksplit > 1) launch_slab_reduce(slab, ksplit, n_out, cout, scale, shift, nullptr, relu, nullptr, s, out, nullptr, residual);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
