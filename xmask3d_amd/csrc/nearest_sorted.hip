// Exact nearest-valid-point fill, Morton-sorted and tile-pruned: out[i] = i for valid points, else the index of the nearest
// valid point - same f32 distance chain and lowest-index tie rule as k_nearest / xm3d_nearest_valid_fill (identity when
// nothing is valid).  Third formulation of the KD-tree query of run/infer.py:682-694, for the case the other two are bad at:
// MANY queries FAR from the references (S1's vote fill: two thirds of the room never seen).  The scan (k_nearest) tests every
// pair; the octree (nearest_grid.hip) chases pointers through every leaf inside each query's ball, one thread per query.
// Here the broadcast scan is kept - it is the efficient inner loop - but both sides are sorted along a Morton curve so that
//   * the 64 queries of a WAVE are spatial neighbours with a small bounding box,
//   * the references come in 64-point tiles with bounding boxes,
// and a wave only scans the tiles whose box is not farther from its query box than the worst best-distance among its queries
// (strictly farther: an equal distance may carry a lower index), and of those only the ones some query's own point-to-box
// bound cannot exclude.  Waves are autonomous - own bound, own wave-private LDS staging slot, no workgroup barrier in the
// loop; the bounds of 64 tiles are evaluated lane-parallel and the survivors visited through a ballot mask.  The tile nearest
// to the query box goes first, so the bound is tight from the start.  (A first
// version with 256-query workgroups and 256-point tiles pruned almost nothing on S1: one far query per workgroup keeps the
// workgroup bound at metres.)
// Pipeline (device-resident sizes, no host synchronisation): bounding box -> keys (valid bit | 30-bit Morton) -> rocPRIM radix
// sort (31 bits) -> gather + tile boxes -> query.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace xm3d {

constexpr int NS_TILE = 64;          // queries per wave = references per tile
constexpr int NS_P_MIN = 0, NS_P_MAX = 3, NS_P_NVALID = 6, NS_P_WORDS = 16;
constexpr int NS_P_SCANNED = 8, NS_P_SCANNED_MAX = 9, NS_P_WAVES = 10;  // statistics: tiles scanned (sum / max per wave), live waves

__device__ __forceinline__ uint32_t ns_ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ns_unordered(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__device__ __forceinline__ uint32_t ns_spread10(uint32_t v) {  // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_ns_init(uint32_t* __restrict__ params) {
    if (threadIdx.x < NS_P_WORDS) params[threadIdx.x] = (threadIdx.x < NS_P_MAX) ? 0xFFFFFFFFu : 0u;
}

// bounding box of ALL points and the number of valid ones
__global__ __launch_bounds__(256) void k_ns_bbox(const float* __restrict__ xyz, int64_t n, const uint8_t* __restrict__ valid,
                                                 uint32_t* __restrict__ params) {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u}, cnt = 0;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        cnt += valid[i] != 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t o = ns_ordered(xyz[3 * i + a]);
            lo[a] = min(lo[a], o);
            hi[a] = max(hi[a], o);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], uint32_t(__shfl_xor(int(lo[a]), off)));
            hi[a] = max(hi[a], uint32_t(__shfl_xor(int(hi[a]), off)));
        }
        cnt += uint32_t(__shfl_xor(int(cnt), off));
    }
    __shared__ uint32_t red[4][7];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[wave][a] = lo[a];
            red[wave][3 + a] = hi[a];
        }
        red[wave][6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int a = threadIdx.x;
        uint32_t v = red[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? min(v, red[w][a]) : (a < 6 ? max(v, red[w][a]) : v + red[w][a]);
        if (a < 3) atomicMin(&params[NS_P_MIN + a], v);
        else if (a < 6) atomicMax(&params[NS_P_MAX + a - 3], v);
        else if (v) atomicAdd(&params[NS_P_NVALID], v);
    }
}

// key = valid bit (30) | Morton code (0..29): queries sort first, each side along the curve; out[i] = i for every point
__global__ __launch_bounds__(256) void k_ns_keys(const float* __restrict__ xyz, int64_t n, const uint8_t* __restrict__ valid,
                                                 const uint32_t* __restrict__ params, uint32_t* __restrict__ keys,
                                                 int32_t* __restrict__ vals, int64_t* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = ns_unordered(params[NS_P_MIN + a]), hi = ns_unordered(params[NS_P_MAX + a]);
        const float ext = hi - lo;
        const float t = ext > 0.f ? (xyz[3 * i + a] - lo) / ext : 0.f;
        const uint32_t c = uint32_t(fminf(fmaxf(t * 1023.f, 0.f), 1023.f));
        code |= ns_spread10(c) << a;
    }
    keys[i] = code | (valid[i] ? (1u << 30) : 0u);
    vals[i] = int32_t(i);
    out[i] = i;
}

// sorted order -> float4 (x, y, z, original index) + the bounding box of every 64-point tile (one wave per tile).  blockIdx.y =
// 0: query tiles (sorted positions [0, nq)), 1: reference tiles ([nq, n)); tiles are aligned to the start of their side.
__global__ __launch_bounds__(256) void k_ns_gather(const float* __restrict__ xyz, int64_t n, const uint32_t* __restrict__ params,
                                                   const int32_t* __restrict__ order, float4* __restrict__ pts,
                                                   float* __restrict__ boxes, int64_t max_tiles) {
    const int64_t nq = n - int64_t(params[NS_P_NVALID]);
    const int side = blockIdx.y;
    const int64_t lo = side ? nq : 0, hi = side ? n : nq;
    const int64_t t = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t j = lo + t * NS_TILE + (threadIdx.x & 63);
    if (lo + t * NS_TILE >= hi) return;  // whole wave
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (j < hi) {
        const int o = order[j];
        const float x = xyz[3 * int64_t(o)], y = xyz[3 * int64_t(o) + 1], z = xyz[3 * int64_t(o) + 2];
        pts[j] = make_float4(x, y, z, __int_as_float(o));
        mn[0] = mx[0] = x;
        mn[1] = mx[1] = y;
        mn[2] = mx[2] = z;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        float* b = boxes + (int64_t(side) * max_tiles + t) * 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            b[a] = mn[a];
            b[3 + a] = mx[a];
        }
    }
}

// One workgroup = one 64-query tile, four waves that share the queries and split the candidate tiles (step % 4 in the outward
// order): four times the waves in flight to hide the global-load latency of a tile fetch, and a four times shorter tail.  Every
// wave keeps its own best per query; the four are merged through LDS at the end (same (distance, index) order).
__global__ __launch_bounds__(256) void k_ns_query(int64_t n, uint32_t* __restrict__ params, const float4* __restrict__ pts,
                                                  const float* __restrict__ boxes, int64_t max_tiles, int64_t* __restrict__ out) {
    __shared__ float4 stage[4][NS_TILE];      // wave-private slots: the reference tile being scanned
    __shared__ float4 tbox[4][NS_TILE][2];    // wave-private: boxes of the 64 tiles under consideration
    __shared__ float mbest[4][NS_TILE];
    __shared__ int mbesti[4][NS_TILE];
    const int64_t nref = int64_t(params[NS_P_NVALID]);
    const int64_t nq = n - nref;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t qt = blockIdx.x;
    const int64_t q0 = qt * NS_TILE;
    if (q0 >= nq || nref == 0) return;  // whole workgroup; nothing valid anywhere: out keeps the identity written by k_ns_keys
    const int ntile = int((nref + NS_TILE - 1) / NS_TILE);
    const bool live = q0 + lane < nq;
    const float4 Q = pts[live ? q0 + lane : q0];
    const float* qb = boxes + qt * 6;
    const float qlo[3] = {qb[0], qb[1], qb[2]}, qhi[3] = {qb[3], qb[4], qb[5]};
    const float* rb = boxes + max_tiles * 6;
    float best = live ? INFINITY : 0.f;  // idle lanes must not hold the wave bound up
    int besti = 0x7fffffff;
    float bmax = INFINITY;
    uint32_t scanned = 0;
    auto scan_tile = [&](int t) {
        ++scanned;
        const int64_t r0 = nq + int64_t(t) * NS_TILE;
        const int cnt = int(nref - int64_t(t) * NS_TILE < NS_TILE ? nref - int64_t(t) * NS_TILE : NS_TILE);
        __builtin_amdgcn_wave_barrier();  // the slot's previous contents are consumed
        stage[wave][lane] = lane < cnt ? pts[r0 + lane] : make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 8
        for (int j = 0; j < NS_TILE; ++j) {
            const float4 P = stage[wave][j];  // same address for the whole wave: LDS broadcast
            const float dx = Q.x - P.x, dy = Q.y - P.y, dz = Q.z - P.z;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));  // the k_nearest chain: bit-identical distances
            const int idx = __float_as_int(P.w);
            if (d < best || (d == best && idx < besti)) {
                best = d;
                besti = idx;
            }
        }
        float wm = best;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off));
        bmax = wm;
    };
    // the tile nearest to the query box first (every wave scans it: a tight bound from the start)
    float sb = INFINITY;
    int si = 0;
    for (int t = lane; t < ntile; t += 64) {
        float d2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g = fmaxf(fmaxf(rb[6 * int64_t(t) + a] - qhi[a], qlo[a] - rb[6 * int64_t(t) + 3 + a]), 0.f);
            d2 = fmaf(g, g, d2);
        }
        if (d2 < sb) {
            sb = d2;
            si = t;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(sb, off);
        const int oi = __shfl_xor(si, off);
        if (ob < sb || (ob == sb && oi < si)) {
            sb = ob;
            si = oi;
        }
    }
    const int seed = si;
    scan_tile(seed);
    // all other tiles outwards from the seed (neighbours on the curve are mostly neighbours in space: the bests shrink fast),
    // 64 of this wave's steps at a time: lane l fetches the box of its step (one load latency for 64 tiles) and drops it if the box is
    // farther from the query box than the wave's worst best; the survivors are then tested per QUERY (each lane its own point
    // against the tile's box, read back from LDS as a broadcast) and scanned iff some query cannot exclude them - so a wave
    // that straddles a jump of the curve pays for two neighbourhoods, not for the whole cloud.
    for (int s0 = 1; s0 < 2 * ntile; s0 += 256) {
        const int step = s0 + 4 * lane + ((wave - 1) & 3);  // s0 = 1 (mod 4): this wave's steps are = wave (mod 4)
        const int t = (step & 1) ? seed - ((step + 1) >> 1) : seed + (step >> 1);
        bool cand = step < 2 * ntile && t >= 0 && t < ntile;
        if (cand) {
            const float* tb = rb + 6 * int64_t(t);
            const float4 lo4 = make_float4(tb[0], tb[1], tb[2], 0.f), hi4 = make_float4(tb[3], tb[4], tb[5], 0.f);
            const float gx = fmaxf(fmaxf(lo4.x - qhi[0], qlo[0] - hi4.x), 0.f), gy = fmaxf(fmaxf(lo4.y - qhi[1], qlo[1] - hi4.y), 0.f),
                        gz = fmaxf(fmaxf(lo4.z - qhi[2], qlo[2] - hi4.z), 0.f);
            cand = !(fmaf(gz, gz, fmaf(gy, gy, gx * gx)) * 0.99999f > bmax);
            tbox[wave][lane][0] = lo4;
            tbox[wave][lane][1] = hi4;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long m = __ballot(cand);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float4 lo4 = tbox[wave][b][0], hi4 = tbox[wave][b][1];
            const float gx = fmaxf(fmaxf(lo4.x - Q.x, Q.x - hi4.x), 0.f), gy = fmaxf(fmaxf(lo4.y - Q.y, Q.y - hi4.y), 0.f),
                        gz = fmaxf(fmaxf(lo4.z - Q.z, Q.z - hi4.z), 0.f);
            const float pb = fmaf(gz, gz, fmaf(gy, gy, gx * gx)) * 0.99999f;  // rounded DOWN: a bound, never an overestimate
            if (__ballot(!(pb > best)) == 0ull) continue;                     // strictly farther than every query's best
            scan_tile(__shfl(t, b));
        }
        __builtin_amdgcn_wave_barrier();  // tbox consumed before the next block overwrites it
    }
    mbest[wave][lane] = best;
    mbesti[wave][lane] = besti;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float d = mbest[w][lane];
            const int idx = mbesti[w][lane];
            if (d < best || (d == best && idx < besti)) {
                best = d;
                besti = idx;
            }
        }
        if (live && besti != 0x7fffffff) out[__float_as_int(Q.w)] = int64_t(besti);
    }
    if (lane == 0) {
        atomicAdd(&params[NS_P_SCANNED], scanned);
        atomicMax(&params[NS_P_SCANNED_MAX], scanned);
        atomicAdd(&params[NS_P_WAVES], 1u);
    }
}

static size_t ns_sort_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t*, uint32_t*, const int32_t*, int32_t*>(
        nullptr, bytes, nullptr, nullptr, nullptr, nullptr, size_t(n > 0 ? n : 1), 0, 31, nullptr, false);
    return align_up(bytes, 256);
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int64_t xm3d_nearest_valid_fill_sorted_workspace_bytes(int64_t n) {
    if (n < 1) n = 1;
    const int64_t tiles = (n + NS_TILE - 1) / NS_TILE + 1;
    return 256 + int64_t(align_up(size_t(n) * 4, 256)) * 4 + int64_t(align_up(size_t(n) * 16, 256)) +
           int64_t(align_up(size_t(tiles) * 2 * 6 * 4, 256)) + int64_t(ns_sort_bytes(n));
}

extern "C" int xm3d_nearest_valid_fill_sorted(const float* xyz, int64_t n, const uint8_t* valid, int64_t* out, void* ws, void* stream) {
    XM3D_REQUIRE(n >= 0 && n < (1ll << 31), "nearest_valid_fill_sorted: 0 <= n < 2^31 points expected (n=%lld)", (long long)n);
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(xyz && valid && out && ws, "nearest_valid_fill_sorted: null pointer");
    XM3D_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "nearest_valid_fill_sorted: workspace must be 256-byte aligned");
    hipStream_t s = as_stream(stream);
    const int64_t tiles = (n + NS_TILE - 1) / NS_TILE + 1;
    Carver c(ws);
    uint32_t* params = c.take<uint32_t>(NS_P_WORDS);
    uint32_t* k_in = c.take<uint32_t>(size_t(n));
    uint32_t* k_out = c.take<uint32_t>(size_t(n));
    int32_t* v_in = c.take<int32_t>(size_t(n));
    int32_t* v_out = c.take<int32_t>(size_t(n));
    float4* pts = c.take<float4>(size_t(n));
    float* boxes = c.take<float>(size_t(tiles) * 2 * 6);
    size_t sort_bytes = ns_sort_bytes(n);
    void* sort_ws = c.take<char>(sort_bytes);
    const unsigned nb = unsigned((n + 255) / 256);
    hipLaunchKernelGGL(k_ns_init, dim3(1), dim3(256), 0, s, params);
    hipLaunchKernelGGL(k_ns_bbox, dim3(nb < 128u ? nb : 128u), dim3(256), 0, s, xyz, n, valid, params);
    hipLaunchKernelGGL(k_ns_keys, dim3(nb), dim3(256), 0, s, xyz, n, valid, params, k_in, v_in, out);
    XM3D_HIP(rocprim::radix_sort_pairs(sort_ws, sort_bytes, (const uint32_t*)k_in, k_out, (const int32_t*)v_in, v_out, size_t(n), 0, 31, s,
                                       false));
    hipLaunchKernelGGL(k_ns_gather, dim3(unsigned((tiles + 3) / 4), 2), dim3(256), 0, s, xyz, n, params, v_out, pts, boxes, tiles);
    hipLaunchKernelGGL(k_ns_query, dim3(unsigned(tiles)), dim3(256), 0, s, n, params, pts, boxes, tiles, out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
