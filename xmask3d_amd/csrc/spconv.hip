// Sparse convolution over a neighbour table (output-stationary hashed rulebook).
//
//   out[o,:] = epi( sum_k in[nbr[k,o],:] @ W[k] ),   W (K,Cin,Cout) f32, nbr (K,n_out) i32 (-1 = hole)
//
// algo 1  k_spconv_scalar : one thread per (row, cout); reference-quality kernel, any Cin/Cout.
// algo 2  k_spconv_mfma   : f32 MFMA (v_mfma_f32_16x16x4_f32, exact f32 fma chains).
//   * workgroup = 4 waves, 256 output rows (64 per wave, one ballot wide) x 32 output channels
//   * per kernel offset each wave PACKS its valid (row, neighbour) pairs into dense 16-pair MFMA
//     tiles, so holes in the rulebook cost padding to 16, not whole zero tiles
//   * the 32-deep weight chunk W[k][c0:c0+32][ct0:ct0+32] is staged once per workgroup in LDS in
//     MFMA-fragment order (pre-packed by xm3d_spconv_pack_weight, so the stage is a linear copy)
//     and shared by the four waves; neighbour rows are gathered straight from global/L2 as
//     64-byte segments (4 lanes x float4 per row per 16 channels)
//   * products are computed transposed (D^T = W^T X^T) so a lane ends up with 4 CONSECUTIVE output
//     channels of one pair and folds them into the wave-private LDS accumulator row of that pair's
//     output row with one 16-byte read-modify-write (no atomics; deterministic summation order:
//     offsets ascending, 32-channel chunks ascending, fma chain inside)
//   * epilogue: scale/shift (folded BatchNorm), residual add, ReLU, coalesced row stores
//
// Replaces ME.MinkowskiConvolution(+Transpose) and the BN/ReLU/residual tail of ME's BasicBlock
// (call sites: models/modeling/meta_arch/mink_unet.py:47-109,118-178, resnet_base.py:64-96).
#include <cstdlib>

#include "common.h"

namespace xm3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ scalar kernel
__global__ void k_spconv_scalar(const float* __restrict__ in, int cin, const float* __restrict__ W, int K, int cout,
                                const int32_t* __restrict__ nbr, int64_t n_out, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ residual, int relu,
                                float* __restrict__ out) {
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n_out * cout) return;
    const int64_t o = e / cout;
    const int c = int(e % cout);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int i = nbr ? nbr[int64_t(k) * n_out + o] : int(o);
        if (i < 0) continue;
        const float* x = in + int64_t(i) * cin;
        const float* w = W + (int64_t(k) * cin) * cout + c;
        float part = 0.f;
        for (int ci = 0; ci < cin; ++ci) part = fmaf(x[ci], w[int64_t(ci) * cout], part);
        acc += part;
    }
    if (scale) acc *= scale[c];
    if (shift) acc += shift[c];
    if (residual) acc += residual[e];
    if (relu) acc = fmaxf(acc, 0.f);
    out[e] = acc;
}

// ------------------------------------------------------------------ weight packing
// Wp[((k*CS + cs)*NS + ns)*256 + lane*4 + jj] = W[k][16cs + 4(lane>>4) + jj][16ns + (lane&15)]
__global__ void k_pack_weight(const float* __restrict__ W, int K, int cin, int cout, float* __restrict__ Wp) {
    const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t total = int64_t(K) * cin * cout;
    if (e >= total) return;
    const int jj = int(e & 3);
    const int lane = int((e >> 2) & 63);
    const int64_t blk = e >> 8;
    const int NS = cout / 16, CS = cin / 16;
    const int ns = int(blk % NS);
    const int cs = int((blk / NS) % CS);
    const int k = int(blk / (int64_t(NS) * CS));
    const int ci = 16 * cs + 4 * (lane >> 4) + jj;
    const int co = 16 * ns + (lane & 15);
    Wp[e] = W[(int64_t(k) * cin + ci) * cout + co];
}

// ------------------------------------------------------------------ MFMA kernel
constexpr int RW = 64;        // rows per wave
constexpr int WAVES = 4;      // waves per workgroup
constexpr int CT = 32;        // output channels per workgroup
constexpr int NT = CT / 16;   // 16-wide channel tiles
constexpr int CK = 32;        // input-channel chunk staged per step
constexpr int ST = CK / 16;   // 16-deep k-steps per chunk
constexpr int ACC_LD = CT + 4;  // padded accumulator row (floats): 16-byte aligned, breaks bank stride

struct __attribute__((aligned(16))) SpconvLds {
    float acc[WAVES][RW][ACC_LD];     // wave-private accumulators
    float w[ST * NT * 256];           // staged weight chunk, fragment order
    int src[WAVES][RW];               // packed pair -> input row
    int dst[WAVES][RW];               // packed pair -> local output row
};

__global__ __launch_bounds__(256, 2) void k_spconv_mfma(const float* __restrict__ in, int cin, const float* __restrict__ Wp,
                                                         int K, int cout, const int32_t* __restrict__ nbr,
                                                         const int32_t* __restrict__ order, int64_t n_out,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ residual, int relu,
                                                         float* __restrict__ out) {
    __shared__ SpconvLds lds;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int p16 = lane & 15;  // pair slot inside an MFMA tile / cout inside a tile (A side)
    const int kq = lane >> 4;   // k quarter (operands) / cout quarter (result)
    const int ct0 = blockIdx.y * CT;
    const int NS = cout / 16, CS = cin / 16;

    // this lane's output row
    const int64_t slot = int64_t(blockIdx.x) * (RW * WAVES) + wave * RW + lane;
    int64_t row = -1;
    if (slot < n_out) row = order ? order[slot] : slot;

    for (int c = lane; c < RW * ACC_LD; c += 64) (&lds.acc[wave][0][0])[c] = 0.f;

    for (int k = 0; k < K; ++k) {
        int v = -1;
        if (row >= 0) v = nbr ? nbr[int64_t(k) * n_out + row] : int(row);
        const unsigned long long m = __ballot(v >= 0);
        const int cnt = __popcll(m);
        if (__syncthreads_or(cnt) == 0) continue;  // nobody in the workgroup uses this offset
        if (v >= 0) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            lds.src[wave][rank] = v;
            lds.dst[wave][rank] = lane;
        }
        const int ntile = (cnt + 15) >> 4;
        for (int cc = 0; cc < cin / CK; ++cc) {
            // stage W[k][cc*CK .. +CK][ct0 .. +CT] (already in fragment order): ST*NT blocks of 1 KiB
            __syncthreads();  // previous chunk fully consumed (and src/dst visible)
            {
                const int b = tid >> 6;  // one 1-KiB block per wave when ST*NT == 4
                for (int blk = b; blk < ST * NT; blk += WAVES) {
                    const int s = blk / NT, n = blk % NT;
                    const int64_t g = ((int64_t(k) * CS + (cc * ST + s)) * NS + (ct0 / 16 + n)) * 256 + lane * 4;
                    *reinterpret_cast<f32x4*>(&lds.w[blk * 256 + lane * 4]) = *reinterpret_cast<const f32x4*>(Wp + g);
                }
            }
            __syncthreads();
            for (int t = 0; t < ntile; ++t) {
                const int p = t * 16 + p16;
                const int srow = (p < cnt) ? lds.src[wave][p] : -1;
                f32x4 x[ST];
#pragma unroll
                for (int s = 0; s < ST; ++s) {
                    x[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (srow >= 0)
                        x[s] = *reinterpret_cast<const f32x4*>(in + int64_t(srow) * cin + cc * CK + s * 16 + kq * 4);
                }
                const int drow = (p < cnt) ? lds.dst[wave][p] : -1;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < ST; ++s) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(&lds.w[(s * NT + n) * 256 + lane * 4]);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x[s][0], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], x[s][1], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], x[s][2], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], x[s][3], d, 0, 0, 0);
                    }
                    // d[j] = result for pair p16, channel ct0 + 16n + 4kq + j
                    if (drow >= 0) {
                        f32x4* a = reinterpret_cast<f32x4*>(&lds.acc[wave][drow][n * 16 + kq * 4]);
                        f32x4 cur = *a;
                        cur += d;
                        *a = cur;
                    }
                }
            }
        }
    }
    // epilogue: each wave writes its own 64 rows; 8 lanes x float4 cover one 32-channel row
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): own LDS RMWs done (wave-private rows)
    const int rsub = lane >> 3;          // 0..7 row inside an 8-row group
    const int c4 = (lane & 7) * 4;       // channel offset
    for (int r0 = 0; r0 < RW; r0 += 8) {
        const int lr = r0 + rsub;
        const int64_t sl = int64_t(blockIdx.x) * (RW * WAVES) + wave * RW + lr;
        if (sl >= n_out) continue;
        const int64_t grow = order ? order[sl] : sl;
        f32x4 vacc = *reinterpret_cast<const f32x4*>(&lds.acc[wave][lr][c4]);
        const int c = ct0 + c4;
        if (scale) vacc *= *reinterpret_cast<const f32x4*>(scale + c);
        if (shift) vacc += *reinterpret_cast<const f32x4*>(shift + c);
        if (residual) vacc += *reinterpret_cast<const f32x4*>(residual + grow * cout + c);
        if (relu) {
            vacc[0] = fmaxf(vacc[0], 0.f);
            vacc[1] = fmaxf(vacc[1], 0.f);
            vacc[2] = fmaxf(vacc[2], 0.f);
            vacc[3] = fmaxf(vacc[3], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + grow * cout + c) = vacc;
    }
}

// ------------------------------------------------------------------ tiled rulebook + MFMA kernel v2 (algo 3)
// The neighbour table is compacted ONCE per kernel map into per-workgroup pair lists (shared by every conv
// that uses the map, 8-22 layers per level in the MinkUNets): for row tile b (256 rows in processing order)
// and offset k, tsrc/tdst[(b*K + k)*256 + j] hold the input row / local output row of the j-th valid pair and
// tcnt[b*K + k] their number.  The conv kernel then has no ballots and knows all its work up front:
//   * pairs are packed at WORKGROUP level (ceil(cnt/16) MFMA tiles dealt round-robin to the 4 waves)
//   * weights for step s+1 are fetched into registers while step s computes, written to the other LDS
//     buffer after the MFMAs: one barrier per (offset, channel-chunk) step
//   * channel chunk = 16*ST input channels (ST = 2, 4 or 6 -> 32/64/96-deep steps)
constexpr int TROWS = 256;

__global__ __launch_bounds__(256) void k_build_tiles(const int32_t* __restrict__ nbr, const int32_t* __restrict__ order,
                                                     int64_t n_out, int K, int32_t* __restrict__ tsrc,
                                                     uint8_t* __restrict__ tdst, int32_t* __restrict__ tcnt) {
    __shared__ int wcnt[4];
    const int tile = blockIdx.x, k = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t slot = int64_t(tile) * TROWS + threadIdx.x;
    int v = -1;
    if (slot < n_out) {
        const int64_t row = order ? order[slot] : slot;
        v = nbr ? nbr[int64_t(k) * n_out + row] : int(row);
    }
    const unsigned long long m = __ballot(v >= 0);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wcnt[w];
    const int64_t off = (int64_t(tile) * K + k) * TROWS;
    if (v >= 0) {
        const int j = base + __popcll(m & ((1ull << lane) - 1ull));
        tsrc[off + j] = v;
        tdst[off + j] = uint8_t(threadIdx.x);
    }
    if (threadIdx.x == 0) tcnt[int64_t(tile) * K + k] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// NTT = 16-wide output-channel tiles per workgroup (2 -> 32 channels, 3 -> 48): a wider tile gathers every input row
// fewer times (cout / (16*NTT) workgroups share a row tile) at the price of a larger LDS accumulator.
template <int ST_, int NTT>
struct __attribute__((aligned(16))) TileLds {
    float acc[TROWS][16 * NTT + 4];
    float w[2][ST_ * NTT * 256];
    int cnt[128];  // pairs per offset of this tile (K <= 125)
};

// Per-wave software pipeline over the wave's own tile sequence (which spans steps):
//   stage A: pair indices (tsrc/tdst) of tile i+3      stage B: row gathers of tile i+2      stage C: MFMAs of tile i
// so the count -> index -> gather chain of dependent L2 round trips is off the critical path.
template <int ST_, int NTT = NT>
__global__ __launch_bounds__(256, (NTT <= 2 ? 2 : 1)) void k_spconv_tiles(const float* __restrict__ in, int cin, const float* __restrict__ Wp,
                                                          int K, int cout, const int32_t* __restrict__ tsrc,
                                                          const uint8_t* __restrict__ tdst, const int32_t* __restrict__ tcnt,
                                                          const int32_t* __restrict__ order, int64_t n_out,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ residual, int relu,
                                                          float* __restrict__ out, int ksplit, float* __restrict__ slab) {
    __shared__ TileLds<ST_, NTT> lds;
    constexpr int CTT = 16 * NTT, ACCLD = CTT + 4;
    constexpr int TOT4 = ST_ * NTT * 64;          // float4 of weights per step
    constexpr int NW = (TOT4 + 255) / 256;        // float4 weight loads per thread per step
    const int kz = blockIdx.z;                              // split-K: this workgroup handles offsets kz, kz+ksplit, ...
    const int nk = (K - kz + ksplit - 1) / ksplit;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p16 = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x;
    const int ct0 = blockIdx.y * CTT;
    const int NS = cout / 16, CS = cin / 16;
    const int nchunk = cin / (16 * ST_);
    const int nsteps = nk * nchunk;
    const int64_t tbase = int64_t(tile) * K;

    for (int c = tid; c < TROWS * ACCLD; c += 256) (&lds.acc[0][0])[c] = 0.f;
    if (tid < nk) lds.cnt[tid] = tcnt[tbase + kz + tid * ksplit];

    f32x4 wreg[NW];
    auto load_w = [&](int step) {
        const int kk = step / nchunk, cc = step - kk * nchunk;
        const int k = kz + kk * ksplit;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int f = j * 256 + tid;  // float4 index inside the chunk: ST segments of NTT*64 float4
            if (TOT4 % 256 != 0 && f >= TOT4) continue;
            const int s = f / (NTT * 64), within = f % (NTT * 64);
            const int64_t g = ((int64_t(k) * CS + cc * ST_ + s) * NS + ct0 / 16) * 256 + within * 4;
            wreg[j] = *reinterpret_cast<const f32x4*>(Wp + g);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NW; ++j)
            if (TOT4 % 256 == 0 || j * 256 + tid < TOT4) *reinterpret_cast<f32x4*>(&lds.w[buf][(j * 256 + tid) * 4]) = wreg[j];
    };
    if (nsteps > 0) {
        load_w(0);
        store_w(0);
    }
    __syncthreads();

    // tile iterator of this wave: (step, t) with t = wave, wave+4, ... < ceil(cnt[k]/16); step == nsteps means "none"
    struct Ref {
        int step, t;
    };
    auto advance = [&](Ref r) {
        r.t += WAVES;
        while (r.step < nsteps) {
            const int kk = r.step / nchunk;
            if (r.t * 16 < lds.cnt[kk]) break;
            ++r.step;
            r.t = wave;
        }
        return r;
    };
    auto load_idx = [&](Ref r, int& srow, int& drow) {
        srow = -1;
        drow = -1;
        if (r.step < nsteps) {
            const int kk = r.step / nchunk;
            const int p = r.t * 16 + p16;
            if (p < lds.cnt[kk]) {
                const int64_t o = (tbase + kz + kk * ksplit) * TROWS + p;
                srow = tsrc[o];
                drow = int(tdst[o]);
            }
        }
    };
    auto gather = [&](Ref r, int srow, f32x4 (&x)[ST_]) {
        const int cc = (r.step < nsteps) ? r.step % nchunk : 0;
#pragma unroll
        for (int s = 0; s < ST_; ++s) {
            x[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (srow >= 0) x[s] = *reinterpret_cast<const f32x4*>(in + int64_t(srow) * cin + cc * (16 * ST_) + s * 16 + kq * 4);
        }
    };

    // pipeline registers: indices three tiles ahead (s3,d3), gathers two tiles ahead (x2), MFMAs on x0
    Ref r0 = advance(Ref{0, wave - WAVES});
    Ref r1 = advance(r0);
    Ref r2 = advance(r1);
    int s0, d0, s1, d1, s2, d2;
    load_idx(r0, s0, d0);
    load_idx(r1, s1, d1);
    load_idx(r2, s2, d2);
    f32x4 x0[ST_], x1[ST_], x2[ST_];
    gather(r0, s0, x0);
    gather(r1, s1, x1);

    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) load_w(step + 1);
        while (r0.step == step) {
            const Ref r3 = advance(r2);
            int s3, d3;
            load_idx(r3, s3, d3);   // stage A for tile i+3
            gather(r2, s2, x2);     // stage B for tile i+2
#pragma unroll
            for (int n = 0; n < NTT; ++n) {  // stage C
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < ST_; ++s) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(&lds.w[cur][(s * NTT + n) * 256 + lane * 4]);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x0[s][0], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], x0[s][1], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], x0[s][2], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], x0[s][3], d, 0, 0, 0);
                }
                if (d0 >= 0) {
                    f32x4* a = reinterpret_cast<f32x4*>(&lds.acc[d0][n * 16 + kq * 4]);
                    f32x4 c = *a;
                    c += d;
                    *a = c;
                }
            }
#pragma unroll
            for (int s = 0; s < ST_; ++s) {
                x0[s] = x1[s];
                x1[s] = x2[s];
            }
            d0 = d1;
            d1 = d2;
            s2 = s3;
            d2 = d3;
            r0 = r1;
            r1 = r2;
            r2 = r3;
        }
        if (step + 1 < nsteps) store_w(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // epilogue: CTT/4 lanes x float4 per row, 256/(CTT/4) rows per pass
    constexpr int LPR = CTT / 4, RPP = 256 / LPR;
    const int rsub = tid / LPR, c4 = (tid % LPR) * 4;
    for (int r = 0; r < TROWS; r += RPP) {
        const int lr = r + rsub;
        const int64_t sl = int64_t(tile) * TROWS + lr;
        if (rsub >= RPP || lr >= TROWS || sl >= n_out) continue;
        const int64_t grow = order ? order[sl] : sl;
        f32x4 v = *reinterpret_cast<const f32x4*>(&lds.acc[lr][c4]);
        const int c = ct0 + c4;
        if (ksplit > 1) {  // raw partial sum; k_slab_reduce applies the epilogue
            *reinterpret_cast<f32x4*>(slab + (int64_t(kz) * n_out + grow) * cout + c) = v;
            continue;
        }
        if (scale) v *= *reinterpret_cast<const f32x4*>(scale + c);
        if (shift) v += *reinterpret_cast<const f32x4*>(shift + c);
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + grow * cout + c);
        if (relu) {
            v[0] = fmaxf(v[0], 0.f);
            v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f);
            v[3] = fmaxf(v[3], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + grow * cout + c) = v;
    }
}

// out = epi(sum_z slab[z]) in fixed z order (bitwise reproducible)
__global__ void k_slab_reduce(const float* __restrict__ slab, int ksplit, int64_t n4, int64_t stride4, int c,
                              const float* __restrict__ scale, const float* __restrict__ shift,
                              const float* __restrict__ residual, int relu, float* __restrict__ out,
                              __bf16* __restrict__ out_hi, __bf16* __restrict__ out_lo, const __bf16* __restrict__ residual_bf) {
    const int64_t gs = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n4; e += gs) {
        f32x4 v = reinterpret_cast<const f32x4*>(slab)[e];
        for (int z = 1; z < ksplit; ++z) v += reinterpret_cast<const f32x4*>(slab)[z * stride4 + e];
        const int ch = int((e * 4) % c);
        if (scale) v *= *reinterpret_cast<const f32x4*>(scale + ch);
        if (shift) v += *reinterpret_cast<const f32x4*>(shift + ch);
        if (residual) v += reinterpret_cast<const f32x4*>(residual)[e];
        if (residual_bf) {  // the bf16 form: residual and result are single bf16 planes
            typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
            const bf16x4e r = reinterpret_cast<const bf16x4e*>(residual_bf)[e];
            v[0] += float(r[0]), v[1] += float(r[1]), v[2] += float(r[2]), v[3] += float(r[3]);
        }
        if (relu) {
            v[0] = fmaxf(v[0], 0.f);
            v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f);
            v[3] = fmaxf(v[3], 0.f);
        }
        if (out) reinterpret_cast<f32x4*>(out)[e] = v;
        if (out_hi && out_lo) store_split4(out_hi + e * 4, out_lo + e * 4, v);
        else if (out_hi) {
            typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
            bf16x4e o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
            reinterpret_cast<bf16x4e*>(out_hi)[e] = o;
        }
    }
}

// host-side launcher shared with spconv_split.hip
int launch_slab_reduce(const float* slab, int ksplit, int64_t n_out, int cout, const float* scale, const float* shift,
                       const float* residual, int relu, float* out, hipStream_t s, void* out_hi, void* out_lo, const void* residual_bf) {
    const int64_t n4 = n_out * cout / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_slab_reduce, dim3(blocks), dim3(256), 0, s, slab, ksplit, n4, n4, cout, scale, shift, residual, relu, out,
                       static_cast<__bf16*>(out_hi), static_cast<__bf16*>(out_lo), static_cast<const __bf16*>(residual_bf));
    return 0;
}

// ------------------------------------------------------------------ weight gradient
// gW[k][ci][co] = sum_o in[nbr[k,o]][ci] * gout[o][co].
// grid = (row chunks, (cin/32)*(cout/32) tiles, K); each wave owns one 32x32 tile of gW[k] and a slice of
// the chunk's rows: valid (in,out) pairs are compacted with a ballot, consumed 4 at a time by
// v_mfma_f32_16x16x4_f32 (A = X^T: lane (ci, pair), B = G: lane (pair, co), both 64-byte row segments
// straight from global/L2), and the tile is folded into gW with f32 atomics (order-dependent last bits,
// like every atomics-based wgrad; K*cin*cout*4 bytes per chunk, far below the 1.3 TB/s atomic ceiling).
constexpr int WG_ROWS = 2048;  // rows per workgroup chunk (512 per wave)

__global__ __launch_bounds__(256) void k_spconv_wgrad_mfma(const float* __restrict__ in, int cin,
                                                            const float* __restrict__ gout, int cout,
                                                            const int32_t* __restrict__ nbr, int64_t n_out,
                                                            float* __restrict__ gW) {
    __shared__ int s_src[WAVES][64];
    __shared__ int s_dst[WAVES][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.z;
    const int tiles_co = cout / 32;
    const int ci0 = (blockIdx.y / tiles_co) * 32, co0 = (blockIdx.y % tiles_co) * 32;
    const int l16 = lane & 15, pq = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t base = int64_t(blockIdx.x) * WG_ROWS + wave * (WG_ROWS / WAVES);
    for (int64_t r0 = base; r0 < base + WG_ROWS / WAVES && r0 < n_out; r0 += 64) {
        const int64_t o = r0 + lane;
        int v = -1;
        if (o < n_out) v = nbr ? nbr[int64_t(k) * n_out + o] : int(o);
        const unsigned long long m = __ballot(v >= 0);
        const int cnt = __popcll(m);
        if (cnt == 0) continue;
        if (v >= 0) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            s_src[wave][rank] = v;
            s_dst[wave][rank] = int(o - r0);
        }
        for (int p0 = 0; p0 < cnt; p0 += 4) {
            const int p = p0 + pq;
            const bool ok = p < cnt;
            const int64_t si = ok ? s_src[wave][p] : 0;
            const int64_t di = ok ? r0 + s_dst[wave][p] : 0;
            float x[2], g[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                x[t] = ok ? in[si * cin + ci0 + 16 * t + l16] : 0.f;
                g[t] = ok ? gout[di * cout + co0 + 16 * t + l16] : 0.f;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[a], g[b], acc[a][b], 0, 0, 0);
        }
    }
    // D[i = ci][j = co]: lane holds co = l16, ci = 4*pq + reg
    float* dst = gW + (int64_t(k) * cin) * cout;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float val = acc[a][b][j];
                if (val != 0.f) atomicAdd(dst + int64_t(ci0 + 16 * a + 4 * pq + j) * cout + co0 + 16 * b + l16, val);
            }
}

// any channel count: one workgroup per (row chunk, k); threads stride over the cin*cout elements
__global__ void k_spconv_wgrad_scalar(const float* __restrict__ in, int cin, const float* __restrict__ gout, int cout,
                                      const int32_t* __restrict__ nbr, int64_t n_out, float* __restrict__ gW) {
    const int k = blockIdx.y;
    const int64_t r0 = int64_t(blockIdx.x) * WG_ROWS;
    const int64_t r1 = (r0 + WG_ROWS < n_out) ? r0 + WG_ROWS : n_out;
    for (int e = threadIdx.x; e < cin * cout; e += blockDim.x) {
        const int ci = e / cout, co = e % cout;
        float acc = 0.f;
        for (int64_t o = r0; o < r1; ++o) {
            const int i = nbr ? nbr[int64_t(k) * n_out + o] : int(o);
            if (i >= 0) acc = fmaf(in[int64_t(i) * cin + ci], gout[o * cout + co], acc);
        }
        if (acc != 0.f) atomicAdd(gW + (int64_t(k) * cin + ci) * cout + co, acc);
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_spconv_pack_weight(const float* W, int32_t K, int32_t cin, int32_t cout, float* Wp, void* stream) {
    XM3D_REQUIRE(W && Wp && K >= 1, "pack_weight: bad args");
    XM3D_REQUIRE(cin % 16 == 0 && cout % 16 == 0, "pack_weight: cin=%d cout=%d must be multiples of 16", cin, cout);
    const int64_t total = int64_t(K) * cin * cout;
    hipLaunchKernelGGL(k_pack_weight, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), W, K, cin, cout, Wp);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_spconv_fwd(const float* in, int64_t n_in, int32_t cin, const float* W, int32_t K, int32_t cout,
                               const int32_t* nbr, const int32_t* order, int64_t n_out, const float* scale,
                               const float* shift, const float* residual, int32_t relu, float* out, int32_t algo,
                               void* stream) {
    XM3D_REQUIRE(n_in >= 0 && n_out >= 0 && cin >= 1 && cout >= 1 && K >= 1, "spconv_fwd: bad sizes");
    XM3D_REQUIRE(n_out * int64_t(cout) < (int64_t(1) << 40), "spconv_fwd: output too large");
    if (n_out == 0) return XM3D_OK;
    XM3D_REQUIRE(in && W && out, "spconv_fwd: null pointer");
    XM3D_REQUIRE(nbr || (K == 1 && n_in == n_out), "spconv_fwd: nbr may be NULL only for K=1 identity maps");
    const bool mfma_ok = (cin % CK == 0) && (cout % CT == 0);
    if (algo == 0) algo = mfma_ok ? 2 : 1;
    hipStream_t s = as_stream(stream);
    if (algo == 1) {
        const int64_t total = n_out * cout;
        hipLaunchKernelGGL(k_spconv_scalar, dim3((total + 255) / 256), dim3(256), 0, s, in, cin, W, K, cout, nbr, n_out,
                           scale, shift, residual, relu, out);
    } else if (algo == 2) {
        XM3D_REQUIRE(mfma_ok, "spconv_fwd: algo 2 needs cin %% %d == 0 and cout %% %d == 0 (got %d, %d)", CK, CT, cin, cout);
        XM3D_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(W) & 15) == 0,
                     "spconv_fwd: algo 2 needs 16-byte aligned tensors");
        // for algo 2, W must be the packed layout produced by xm3d_spconv_pack_weight
        dim3 grid((n_out + RW * WAVES - 1) / (RW * WAVES), cout / CT);
        hipLaunchKernelGGL(k_spconv_mfma, grid, dim3(256), 0, s, in, cin, W, K, cout, nbr, order, n_out, scale, shift,
                           residual, relu, out);
    } else {
        XM3D_REQUIRE(false, "spconv_fwd: unknown algo %d", algo);
    }
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_spconv_bwd_data(const float* gout, int64_t n_out, int32_t cout, const float* Wt, int32_t K,
                                    int32_t cin, const int32_t* nbr_t, const int32_t* order, int64_t n_in, float* gin,
                                    int32_t algo, void* stream) {
    // gin = conv(gout) over the inverse map with per-offset transposed kernels Wt[k] = W[k]^T  (K, Cout, Cin)
    return xm3d_spconv_fwd(gout, n_out, cout, Wt, K, cin, nbr_t, order, n_in, nullptr, nullptr, nullptr, 0, gin, algo,
                           stream);
}

extern "C" int xm3d_spconv_bwd_weight(const float* in, int64_t n_in, int32_t cin, const float* gout, int64_t n_out,
                                      int32_t cout, const int32_t* nbr, int32_t K, float* gW, void* stream) {
    XM3D_REQUIRE(n_in >= 0 && n_out >= 0 && cin >= 1 && cout >= 1 && K >= 1 && gW, "spconv_bwd_weight: bad args");
    XM3D_REQUIRE(nbr || (K == 1 && n_in == n_out), "spconv_bwd_weight: nbr may be NULL only for K=1 identity maps");
    hipStream_t s = as_stream(stream);
    XM3D_HIP(hipMemsetAsync(gW, 0, size_t(K) * cin * cout * sizeof(float), s));
    if (n_out == 0 || n_in == 0) return XM3D_OK;
    XM3D_REQUIRE(in && gout, "spconv_bwd_weight: null pointer");
    const int chunks = int((n_out + WG_ROWS - 1) / WG_ROWS);
    if (cin % 32 == 0 && cout % 32 == 0) {
        hipLaunchKernelGGL(k_spconv_wgrad_mfma, dim3(chunks, (cin / 32) * (cout / 32), K), dim3(256), 0, s, in, cin, gout, cout,
                           nbr, n_out, gW);
    } else {
        hipLaunchKernelGGL(k_spconv_wgrad_scalar, dim3(chunks, K), dim3(256), 0, s, in, cin, gout, cout, nbr, n_out, gW);
    }
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_rulebook_tiles(const int32_t* nbr, const int32_t* order, int64_t n_out, int32_t K, int32_t* tsrc,
                                   uint8_t* tdst, int32_t* tcnt, void* stream) {
    XM3D_REQUIRE(n_out >= 0 && K >= 1, "rulebook_tiles: bad sizes");
    XM3D_REQUIRE(nbr || K == 1, "rulebook_tiles: nbr may be NULL only for K=1 identity maps");
    if (n_out == 0) return XM3D_OK;
    XM3D_REQUIRE(tsrc && tdst && tcnt, "rulebook_tiles: null output");
    const int ntiles = int((n_out + TROWS - 1) / TROWS);
    hipLaunchKernelGGL(k_build_tiles, dim3(ntiles, K), dim3(256), 0, as_stream(stream), nbr, order, n_out, K, tsrc, tdst, tcnt);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

// output channels per workgroup of xm3d_spconv_fwd_tiles for a (cin, cout) layer.  32 everywhere: the 48-wide variant
// (k_spconv_tiles<6, 3>, one gather per 48 instead of 32 output channels) measured 343 us against 255 us on the 96 -> 96
// layer of the roofline bench - one workgroup per CU instead of two costs more than the saved gathers (and 48-deep steps
// with two workgroups: 366 us).  XM3D_SPCONV_CT=48 selects it for experiments (cin % 96 == 0, cout % 48 == 0 layers).
extern "C" int xm3d_spconv_tile_channels(int32_t cin, int32_t cout) {
    static const int forced = [] {
        const char* e = getenv("XM3D_SPCONV_CT");
        return e ? atoi(e) : 0;
    }();
    return (forced == 48 && cin % 96 == 0 && cout % 48 == 0 && cout % 32 == 0) ? 48 : 32;
}

extern "C" int xm3d_spconv_fwd_tiles(const float* in, int64_t n_in, int32_t cin, const float* Wp, int32_t K, int32_t cout,
                                     const int32_t* tsrc, const uint8_t* tdst, const int32_t* tcnt, const int32_t* order,
                                     int64_t n_out, const float* scale, const float* shift, const float* residual,
                                     int32_t relu, float* out, int32_t ksplit, float* slab, void* stream) {
    XM3D_REQUIRE(n_in >= 0 && n_out >= 0 && K >= 1, "spconv_fwd_tiles: bad sizes");
    XM3D_REQUIRE(cin % 32 == 0 && cout % CT == 0 && cin >= 32, "spconv_fwd_tiles: cin=%d cout=%d must be multiples of 32", cin, cout);
    if (n_out == 0) return XM3D_OK;
    XM3D_REQUIRE(in && Wp && tsrc && tdst && tcnt && out, "spconv_fwd_tiles: null pointer");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(Wp) |
                   reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0,
                 "spconv_fwd_tiles: tensors must be 16-byte aligned");
    XM3D_REQUIRE(ksplit >= 1 && ksplit <= K && (ksplit == 1 || slab), "spconv_fwd_tiles: ksplit=%d needs 1..K and a slab", ksplit);
    hipStream_t s = as_stream(stream);
    XM3D_REQUIRE(K <= 128, "spconv_fwd_tiles: K=%d > 128", K);
    if (xm3d_spconv_tile_channels(cin, cout) == 48) {  // 48-channel output tiles: every input row gathered cout/48 times
        dim3 grid48((n_out + TROWS - 1) / TROWS, cout / 48, ksplit);
        hipLaunchKernelGGL((k_spconv_tiles<6, 3>), grid48, dim3(256), 0, s, in, cin, Wp, K, cout, tsrc, tdst, tcnt, order, n_out, scale,
                           shift, residual, relu, out, ksplit, slab);
        if (ksplit > 1) {
            const int64_t n4 = n_out * cout / 4;
            int64_t blocks = (n4 + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(k_slab_reduce, dim3(blocks), dim3(256), 0, s, slab, ksplit, n4, n4, cout, scale, shift, residual, relu, out,
                               static_cast<__bf16*>(nullptr), static_cast<__bf16*>(nullptr), static_cast<const __bf16*>(nullptr));
        }
        XM3D_LAUNCH_CHECK();
        return XM3D_OK;
    }
    dim3 grid((n_out + TROWS - 1) / TROWS, cout / CT, ksplit);
#define XM3D_TILES(ST_)                                                                                                 \
    hipLaunchKernelGGL(k_spconv_tiles<ST_>, grid, dim3(256), 0, s, in, cin, Wp, K, cout, tsrc, tdst, tcnt, order, n_out, \
                       scale, shift, residual, relu, out, ksplit, slab)
    if (cin % 128 == 0) XM3D_TILES(8);
    else if (cin % 96 == 0) XM3D_TILES(6);
    else if (cin % 64 == 0) XM3D_TILES(4);
    else XM3D_TILES(2);
#undef XM3D_TILES
    if (ksplit > 1) {
        const int64_t n4 = n_out * cout / 4;
        int64_t blocks = (n4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_slab_reduce, dim3(blocks), dim3(256), 0, s, slab, ksplit, n4, n4, cout, scale, shift, residual, relu, out,
                               static_cast<__bf16*>(nullptr), static_cast<__bf16*>(nullptr), static_cast<const __bf16*>(nullptr));
    }
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
