// Exact nearest reference point for every query point (3-D, f32), brute force with LDS tiling.
// Replaces the sklearn KD-tree queries of run/infer.py:523-553 (2D-feature hole filling per view) and :682-694
// (labels of never-seen scene points): same result as an exact 1-NN query; ties resolve to the lowest reference index.
// One thread per query; reference points stream through LDS in 1024-point tiles (float4, 16 KiB) shared by the 256
// queries of the workgroup: n*m*(3 sub + 3 fma + compare) VALU work, m*16 B of L2 traffic per workgroup.
#include "common.h"

namespace xm3d {

constexpr int NN_TILE = 1024;

// gridDim.y > 1: the reference range is cut into gridDim.y slices (whole tiles each), one workgroup per (query slab,
// slice), partial results merged through a 64-bit atomicMin on (distance bits << 32 | index) in `best64` (pre-filled with
// ~0): the same winner as a sequential scan (smallest distance, lowest index on ties - distances are >= 0, so their bit
// patterns order like the values), and 4x as many workgroups for the device to balance (80 k queries are only 312 slabs
// on 256 CUs: 56 CUs got two, the kernel took 2x the per-CU time).
__global__ __launch_bounds__(256) void k_nearest(const float* __restrict__ q, int64_t n, const float* __restrict__ r, int64_t m,
                                                 const uint8_t* __restrict__ valid, const int64_t* __restrict__ counts,
                                                 int64_t* __restrict__ out, unsigned long long* __restrict__ best64) {
    __shared__ float4 tile[NN_TILE];
    if (counts) {  // device-resident sizes: only the first counts[0] queries / counts[1] references take part
        n = counts[0] < n ? counts[0] : n;
        m = counts[1] < m ? counts[1] : m;
        if (int64_t(blockIdx.x) * blockDim.x >= n) return;  // whole workgroup beyond the live queries
    }
    int64_t m_lo = 0;
    if (gridDim.y > 1) {
        const int64_t tiles = (m + NN_TILE - 1) / NN_TILE;
        const int64_t per = (tiles + gridDim.y - 1) / gridDim.y;
        m_lo = int64_t(blockIdx.y) * per * NN_TILE;
        const int64_t m_hi = m_lo + per * NN_TILE;
        m = m_hi < m ? m_hi : m;
        if (m_lo >= m) return;
    }
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (i < n) {
        qx = q[3 * i];
        qy = q[3 * i + 1];
        qz = q[3 * i + 2];
    }
    float best = INFINITY;
    int64_t besti = m_lo;
    for (int64_t t0 = m_lo; t0 < m; t0 += NN_TILE) {
        const int cnt = int((m - t0 < NN_TILE) ? m - t0 : NN_TILE);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) {
            const float* p = r + 3 * (t0 + j);
            const bool ok = !valid || valid[t0 + j] != 0;  // masked-out reference points sit at +inf distance
            tile[j] = ok ? make_float4(p[0], p[1], p[2], 0.f) : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
        }
        __syncthreads();
        float tb = best;
        int tj = -1;
#pragma unroll 8
        for (int j = 0; j < cnt; ++j) {
            const float4 p = tile[j];  // same address for the whole wave: LDS broadcast
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d < tb) {
                tb = d;
                tj = j;
            }
        }
        if (tj >= 0) {
            best = tb;
            besti = t0 + tj;
        }
    }
    if (i >= n) return;
    if (best64)
        atomicMin(&best64[i], (static_cast<unsigned long long>(__float_as_uint(best)) << 32) | static_cast<unsigned long long>(besti));
    else
        out[i] = besti;
}

__global__ void k_nearest_unpack(const unsigned long long* __restrict__ best64, int64_t n, const int64_t* __restrict__ counts,
                                 int64_t* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (counts && counts[0] < n) n = counts[0];
    // no live reference point (counts[1] == 0): every slice returned early and best64 still holds its 0xFF fill ->
    // index 0, like the unsplit path (never an out-of-range index for the caller's gather)
    if (i < n) out[i] = (best64[i] == ~0ull) ? 0 : int64_t(best64[i] & 0xFFFFFFFFull);
}

// Segmented form for the per-view hole filling of a whole scene batch in one launch: `pts` holds, for every segment
// (view), its query points followed by its reference points; desc[s] = {q_off, q_cnt, r_off, r_cnt} lives on the device
// (the counts come out of a device-side partition: no host round trip).  blockIdx.y = segment, blockIdx.x = 256-query
// slab; out[q_off + i] = r_off + (index of the nearest reference of the same segment); segments without references
// leave their queries untouched (the caller pre-fills out with the identity).
__global__ __launch_bounds__(256) void k_nearest_seg(const float* __restrict__ pts, const int64_t* __restrict__ desc,
                                                     int64_t* __restrict__ out) {
    __shared__ float4 tile[NN_TILE];
    const int64_t* d4 = desc + 4 * int64_t(blockIdx.y);
    const int64_t q_off = d4[0], n = d4[1], r_off = d4[2], m = d4[3];
    if (int64_t(blockIdx.x) * blockDim.x >= n || m <= 0) return;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (i < n) {
        const float* p = pts + 3 * (q_off + i);
        qx = p[0];
        qy = p[1];
        qz = p[2];
    }
    const float* r = pts + 3 * r_off;
    float best = INFINITY;
    int64_t besti = 0;
    for (int64_t t0 = 0; t0 < m; t0 += NN_TILE) {
        const int cnt = int((m - t0 < NN_TILE) ? m - t0 : NN_TILE);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) {
            const float* p = r + 3 * (t0 + j);
            tile[j] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        float tb = best;
        int tj = -1;
#pragma unroll 8
        for (int j = 0; j < cnt; ++j) {
            const float4 p = tile[j];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d < tb) {
                tb = d;
                tj = j;
            }
        }
        if (tj >= 0) {
            best = tb;
            besti = t0 + tj;
        }
    }
    if (i < n) out[q_off + i] = r_off + besti;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_nearest_index_segmented(const float* pts, const int64_t* desc, int32_t n_seg, int64_t max_queries, int64_t* out,
                                            void* stream) {
    XM3D_REQUIRE(n_seg >= 0 && n_seg <= 65535 && max_queries >= 0, "nearest_index_segmented: bad sizes n_seg=%d max_queries=%lld", n_seg,
                 (long long)max_queries);
    if (n_seg == 0 || max_queries == 0) return XM3D_OK;
    XM3D_REQUIRE(pts && desc && out, "nearest_index_segmented: null pointer");
    hipLaunchKernelGGL(k_nearest_seg, dim3(unsigned((max_queries + 255) / 256), unsigned(n_seg)), dim3(256), 0, as_stream(stream), pts, desc,
                       out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_nearest_index(const float* query, int64_t n, const float* ref, int64_t m, const uint8_t* ref_valid,
                                  const int64_t* counts, int64_t* out, void* ws, void* stream) {
    XM3D_REQUIRE(n >= 0 && m >= 1, "nearest_index: need n >= 0 queries and m >= 1 reference points (n=%lld m=%lld)", (long long)n, (long long)m);
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(query && ref && out, "nearest_index: null pointer");
    XM3D_REQUIRE(m < (1ll << 32), "nearest_index: at most 2^32-1 reference points");
    hipStream_t s = as_stream(stream);
    const unsigned slabs = unsigned((n + 255) / 256);
    // few query slabs per CU and many reference tiles: slice the references (needs the 8*n-byte workspace)
    const int64_t tiles = (m + NN_TILE - 1) / NN_TILE;
    const unsigned splits = (ws && slabs < 2048 && tiles >= 8) ? 4u : 1u;
    if (splits == 1) {
        hipLaunchKernelGGL(k_nearest, dim3(slabs), dim3(256), 0, s, query, n, ref, m, ref_valid, counts, out, nullptr);
    } else {
        unsigned long long* best64 = static_cast<unsigned long long*>(ws);
        XM3D_HIP(hipMemsetAsync(best64, 0xFF, size_t(n) * 8, s));
        hipLaunchKernelGGL(k_nearest, dim3(slabs, splits), dim3(256), 0, s, query, n, ref, m, ref_valid, counts, out, best64);
        hipLaunchKernelGGL(k_nearest_unpack, dim3(slabs), dim3(256), 0, s, best64, n, counts, out);
    }
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
