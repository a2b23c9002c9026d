// Exact nearest-valid-point fill on an implicit octree: out[i] = i for valid points, else the index of the nearest valid
// point (squared f32 distance by the same fma chain as k_nearest, lowest index on ties; identity when nothing is valid).
// Replaces the KD-tree of run/infer.py:682-694 (labels of never-seen scene points from the nearest seen point) the way
// k_nearest does, without the n x m scan:
//   * the valid points are binned into a 64^3 grid over their bounding box (cell edge h >= `cell`) by a counting sort on
//     the MORTON code of the cell (histogram -> two-kernel scan -> scatter), so that the points of every octree node - an
//     aligned 2^l cube of cells - are one contiguous range [start[code << 3l], start[(code + 1) << 3l]);
//   * every query descends that octree depth first, nearest child first (child order = Morton order XOR the query's own
//     octant bits), skipping empty nodes and nodes whose box is farther than the best distance so far (strictly farther: a
//     node at exactly the best distance may still hold a lower index).  Holes of any size cost O(log) node visits, where a
//     shell walk over grid cells costs O(r^2) empty-cell lookups per shell (26 ms on the 120 k-point scene, measured).
// Everything stays on the device: counts, bounding box and grid are recomputed from the same device words by every kernel.
// Latency-bound gather kernel (about 1e2 node visits + a few tens of candidate points per query instead of m = 1e4..1e5
// candidates), not a roofline kernel.
#include "common.h"

namespace xm3d {

constexpr int NG_BITS = 6;                        // cells per axis = 2^6
constexpr int NG_SIDE = 1 << NG_BITS;
constexpr int NG_CELLS = 1 << (3 * NG_BITS);      // 2^18 Morton-ordered cells
constexpr int NG_TABLE = NG_CELLS + 1024;         // + the end sentinel, padded to whole 1024-entry scan blocks
constexpr int NG_SCAN_BLOCKS = NG_TABLE / 1024;
constexpr int NG_LEAF = 48;                       // nodes with at most this many points are scanned, not descended
constexpr int NG_P_MIN = 0, NG_P_MAX = 3, NG_P_NVALID = 6, NG_P_WORDS = 8;

__device__ __forceinline__ uint32_t f_ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f_unordered(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 6 bits -> every third bit
    v = (v | (v << 8)) & 0x0000F00Fu;   // only 6 bits are live: 2 + 4
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}

struct Grid {
    float ox, oy, oz, h, inv_h;
    __device__ __forceinline__ uint32_t cell_of(float x, float y, float z, float& fx, float& fy, float& fz, int& cx, int& cy, int& cz) const {
        fx = (x - ox) * inv_h;
        fy = (y - oy) * inv_h;
        fz = (z - oz) * inv_h;
        cx = min(max(int(floorf(fx)), 0), NG_SIDE - 1);
        cy = min(max(int(floorf(fy)), 0), NG_SIDE - 1);
        cz = min(max(int(floorf(fz)), 0), NG_SIDE - 1);
        return spread3(uint32_t(cx)) | (spread3(uint32_t(cy)) << 1) | (spread3(uint32_t(cz)) << 2);
    }
};

// the grid every kernel derives from the bounding box of the valid points (deterministic: same words in, same grid out)
__device__ __forceinline__ Grid grid_from(const uint32_t* __restrict__ params, float h0) {
    Grid g;
    g.ox = f_unordered(params[NG_P_MIN + 0]);
    g.oy = f_unordered(params[NG_P_MIN + 1]);
    g.oz = f_unordered(params[NG_P_MIN + 2]);
    const float ex = f_unordered(params[NG_P_MAX + 0]) - g.ox, ey = f_unordered(params[NG_P_MAX + 1]) - g.oy,
                ez = f_unordered(params[NG_P_MAX + 2]) - g.oz;
    g.h = fmaxf(h0, fmaxf(ex, fmaxf(ey, ez)) * (1.0f / (NG_SIDE - 0.5f)));  // the farthest valid point sits in cell 63
    g.inv_h = 1.0f / g.h;
    return g;
}

__global__ __launch_bounds__(256) void k_ng_init(uint32_t* __restrict__ params, int32_t* __restrict__ cells) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NG_TABLE) cells[i] = 0;
    if (i < NG_P_WORDS) params[i] = (i < NG_P_MAX) ? 0xFFFFFFFFu : 0u;
}

__global__ __launch_bounds__(256) void k_ng_bbox(const float* __restrict__ xyz, int64_t n, const uint8_t* __restrict__ valid,
                                                 uint32_t* __restrict__ params) {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u}, cnt = 0;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        if (!valid[i]) continue;
        ++cnt;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t o = f_ordered(xyz[3 * i + a]);
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
    // one set of atomics per workgroup (per-wave atomics on seven shared words cost 88 us with 1900 waves)
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
        if (a < 3) atomicMin(&params[NG_P_MIN + a], v);
        else if (a < 6) atomicMax(&params[NG_P_MAX + a - 3], v);
        else if (v) atomicAdd(&params[NG_P_NVALID], v);
    }
}

__global__ __launch_bounds__(256) void k_ng_count(const float* __restrict__ xyz, int64_t n, const uint8_t* __restrict__ valid,
                                                  const uint32_t* __restrict__ params, float h0, int32_t* __restrict__ cells) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n || !valid[i]) return;
    const Grid g = grid_from(params, h0);
    float fx, fy, fz;
    int cx, cy, cz;
    atomicAdd(&cells[g.cell_of(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], fx, fy, fz, cx, cy, cz)], 1);
}

// exclusive scan of the cell histogram, pass 1: every workgroup scans its 1024 entries in place and publishes its total
__global__ __launch_bounds__(256) void k_ng_scan_local(int32_t* __restrict__ cells, int32_t* __restrict__ block_sum) {
    __shared__ int32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int4* p = reinterpret_cast<int4*>(cells + blockIdx.x * 1024) + threadIdx.x;
    const int4 v = *p;
    const int tsum = v.x + v.y + v.z + v.w;
    int inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = inc - tsum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    *p = make_int4(run, run + v.x, run + v.x + v.y, run + v.x + v.y + v.z);
    if (threadIdx.x == 255) block_sum[blockIdx.x] = run + tsum;
}

// pass 2: add the total of the preceding workgroups; `cursor` receives a copy for the scatter pass
__global__ __launch_bounds__(256) void k_ng_scan_add(int32_t* __restrict__ cells, const int32_t* __restrict__ block_sum,
                                                     int32_t* __restrict__ cursor) {
    __shared__ int32_t wsum[4];
    int part = 0;
    for (int b = threadIdx.x; b < int(blockIdx.x); b += 256) part += block_sum[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    const int base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int4* p = reinterpret_cast<int4*>(cells + blockIdx.x * 1024) + threadIdx.x;
    int4 v = *p;
    v.x += base;
    v.y += base;
    v.z += base;
    v.w += base;
    *p = v;
    reinterpret_cast<int4*>(cursor + blockIdx.x * 1024)[threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void k_ng_scatter(const float* __restrict__ xyz, int64_t n, const uint8_t* __restrict__ valid,
                                                    const uint32_t* __restrict__ params, float h0, int32_t* __restrict__ cursor,
                                                    float4* __restrict__ sorted) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n || !valid[i]) return;
    const Grid g = grid_from(params, h0);
    float fx, fy, fz;
    int cx, cy, cz;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const int pos = atomicAdd(&cursor[g.cell_of(x, y, z, fx, fy, fz, cx, cy, cz)], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float(int(i)));
}

__global__ __launch_bounds__(256) void k_ng_query(const float* __restrict__ xyz, int64_t n, const uint8_t* __restrict__ valid,
                                                  const uint32_t* __restrict__ params, float h0, const int32_t* __restrict__ start,
                                                  const float4* __restrict__ sorted, int64_t* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    if (valid[i] || params[NG_P_NVALID] == 0) {
        out[i] = i;
        return;
    }
    const Grid g = grid_from(params, h0);
    const float qx = xyz[3 * i], qy = xyz[3 * i + 1], qz = xyz[3 * i + 2];
    float fx, fy, fz;
    int cx, cy, cz;
    g.cell_of(qx, qy, qz, fx, fy, fz, cx, cy, cz);
    const float h2 = g.h * g.h * 0.999f;  // 0.999: rounding of the products below and of the candidates' f32 distances
    float best = INFINITY;
    int bi = 0x7fffffff;
    auto scan = [&](int s, int e) {  // four independent point loads in flight (the tail repeats the last point: harmless)
        for (int p = s; p < e; p += 4) {
            float4 P[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) P[j] = sorted[min(p + j, e - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dx = qx - P[j].x, dy = qy - P[j].y, dz = qz - P[j].z;
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));  // the k_nearest chain: bit-identical distances
                const int idx = __float_as_int(P[j].w);
                if (d < best || (d == best && idx < bi)) {
                    best = d;
                    bi = idx;
                }
            }
        }
    };
    // the nine table entries delimiting the eight children (level l) of node `code` (level l + 1): one round of loads
    auto children = [&](uint32_t code, int l, int (&st)[9]) {
        const int32_t* base = start + (size_t(code) << (3 * (l + 1)));
#pragma unroll
        for (int c = 0; c < 9; ++c) st[c] = base[size_t(c) << (3 * l)];
    };
    auto occupancy = [&](const int (&st)[9]) {
        uint32_t m = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) m |= uint32_t(st[c + 1] > st[c]) << c;
        return m;
    };
    // depth-first over the octree: `lvl` = level of the children being enumerated (0 = single cells), (px,py,pz) / pcode =
    // coordinates / Morton code of their parent at level lvl + 1, `pending` = 8-bit masks of non-empty unvisited children per level
    int lvl = NG_BITS - 1;
    uint32_t pcode = 0;
    int px = 0, py = 0, pz = 0;
    unsigned long long pending = 0;
    {
        int st[9];
        children(0u, lvl, st);
        pending = (unsigned long long)occupancy(st) << (8 * lvl);
    }
    for (;;) {
        uint32_t m = uint32_t(pending >> (8 * lvl)) & 0xffu;
        if (m == 0u) {  // all children of this parent done: back to its own level
            if (++lvl == NG_BITS) break;
            pcode >>= 3;
            px >>= 1;
            py >>= 1;
            pz >>= 1;
            continue;
        }
        // nearest child first: Morton order XOR the octant of the query's own cell at this level
        const uint32_t oct = uint32_t((cx >> lvl) & 1) | (uint32_t((cy >> lvl) & 1) << 1) | (uint32_t((cz >> lvl) & 1) << 2);
        if (oct & 1u) m = ((m & 0x55u) << 1) | ((m & 0xAAu) >> 1);
        if (oct & 2u) m = ((m & 0x33u) << 2) | ((m & 0xCCu) >> 2);
        if (oct & 4u) m = ((m & 0x0Fu) << 4) | ((m & 0xF0u) >> 4);
        const uint32_t child = uint32_t(__ffs(int(m)) - 1) ^ oct;
        pending &= ~(1ull << (8 * lvl + child));
        const uint32_t code = (pcode << 3) | child;
        const int X = (px << 1) | int(child & 1u), Y = (py << 1) | int((child >> 1) & 1u), Z = (pz << 1) | int(child >> 2);
        // gap (in cells) between the query and the node's box per axis, shrunk by more than the rounding of the cell
        // assignment (|error| < 1e-5 cell inside the grid, relative 2e-7 for queries far outside of it)
        const float lox = float(X << lvl), loy = float(Y << lvl), loz = float(Z << lvl), w = float(1 << lvl);
        const float gx = fmaxf(fmaxf(lox - fx, fx - (lox + w)) * 0.99999f - 1e-3f, 0.f);
        const float gy = fmaxf(fmaxf(loy - fy, fy - (loy + w)) * 0.99999f - 1e-3f, 0.f);
        const float gz = fmaxf(fmaxf(loz - fz, fz - (loz + w)) * 0.99999f - 1e-3f, 0.f);
        if ((gx * gx + gy * gy + gz * gz) * h2 > best) continue;  // strictly farther: equal distance may carry a lower index
        if (lvl == 0) {
            scan(start[code], start[code + 1u]);
            continue;
        }
        int st[9];
        children(code, lvl - 1, st);
        if (st[8] - st[0] <= NG_LEAF) {  // few points under this node: test them all instead of descending
            scan(st[0], st[8]);
            continue;
        }
        pcode = code;
        px = X;
        py = Y;
        pz = Z;
        --lvl;
        pending |= (unsigned long long)occupancy(st) << (8 * lvl);
    }
    out[i] = bi == 0x7fffffff ? i : int64_t(bi);
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int64_t xm3d_nearest_valid_fill_workspace_bytes(int64_t n) {
    return int64_t(NG_P_WORDS) * 4 + (2 * int64_t(NG_TABLE) + NG_SCAN_BLOCKS) * 4 + 64 + (n > 0 ? n : 0) * 16;
}

extern "C" int xm3d_nearest_valid_fill(const float* xyz, int64_t n, const uint8_t* valid, float cell, int64_t* out, void* ws,
                                       void* stream) {
    XM3D_REQUIRE(n >= 0 && n < (1ll << 31), "nearest_valid_fill: 0 <= n < 2^31 points expected (n=%lld)", (long long)n);
    XM3D_REQUIRE(cell > 0.f, "nearest_valid_fill: cell edge must be positive (%g)", double(cell));
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(xyz && valid && out && ws, "nearest_valid_fill: null pointer");
    XM3D_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "nearest_valid_fill: workspace must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    uint32_t* params = static_cast<uint32_t*>(ws);                      // 8 words = 32 B: the tables below stay 16 B aligned
    int32_t* cells = reinterpret_cast<int32_t*>(params + NG_P_WORDS);
    int32_t* cursor = cells + NG_TABLE;
    int32_t* block_sum = cursor + NG_TABLE;
    float4* sorted = reinterpret_cast<float4*>((reinterpret_cast<uintptr_t>(block_sum + NG_SCAN_BLOCKS) + 15) & ~uintptr_t(15));
    const unsigned nb = unsigned((n + 255) / 256);
    hipLaunchKernelGGL(k_ng_init, dim3(NG_TABLE / 256), dim3(256), 0, s, params, cells);
    hipLaunchKernelGGL(k_ng_bbox, dim3(nb < 128u ? nb : 128u), dim3(256), 0, s, xyz, n, valid, params);
    hipLaunchKernelGGL(k_ng_count, dim3(nb), dim3(256), 0, s, xyz, n, valid, params, cell, cells);
    hipLaunchKernelGGL(k_ng_scan_local, dim3(NG_SCAN_BLOCKS), dim3(256), 0, s, cells, block_sum);
    hipLaunchKernelGGL(k_ng_scan_add, dim3(NG_SCAN_BLOCKS), dim3(256), 0, s, cells, block_sum, cursor);
    hipLaunchKernelGGL(k_ng_scatter, dim3(nb), dim3(256), 0, s, xyz, n, valid, params, cell, cursor, sorted);
    hipLaunchKernelGGL(k_ng_query, dim3(nb), dim3(256), 0, s, xyz, n, valid, params, cell, cells, sorted, out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
