// Exact nearest reference point for every query point (3-D, f32), brute force with LDS tiling.
// Replaces the sklearn KD-tree queries of run/infer.py:523-553 (2D-feature hole filling per view) and :682-694
// (labels of never-seen scene points): same result as an exact 1-NN query; ties resolve to the lowest reference index.
// One thread per query; reference points stream through LDS in 1024-point tiles (float4, 16 KiB) shared by the 256
// queries of the workgroup: n*m*(3 sub + 3 fma + compare) VALU work, m*16 B of L2 traffic per workgroup.
#include "common.h"

namespace xm3d {

constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(256) void k_nearest(const float* __restrict__ q, int64_t n, const float* __restrict__ r, int64_t m,
                                                 const uint8_t* __restrict__ valid, const int64_t* __restrict__ counts,
                                                 int64_t* __restrict__ out) {
    __shared__ float4 tile[NN_TILE];
    if (counts) {  // device-resident sizes: only the first counts[0] queries / counts[1] references take part
        n = counts[0] < n ? counts[0] : n;
        m = counts[1] < m ? counts[1] : m;
        if (int64_t(blockIdx.x) * blockDim.x >= n) return;  // whole workgroup beyond the live queries
    }
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (i < n) {
        qx = q[3 * i];
        qy = q[3 * i + 1];
        qz = q[3 * i + 2];
    }
    float best = INFINITY;
    int64_t besti = 0;
    for (int64_t t0 = 0; t0 < m; t0 += NN_TILE) {
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
    if (i < n) out[i] = besti;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_nearest_index(const float* query, int64_t n, const float* ref, int64_t m, const uint8_t* ref_valid,
                                  const int64_t* counts, int64_t* out, void* stream) {
    XM3D_REQUIRE(n >= 0 && m >= 1, "nearest_index: need n >= 0 queries and m >= 1 reference points (n=%lld m=%lld)", (long long)n, (long long)m);
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(query && ref && out, "nearest_index: null pointer");
    hipLaunchKernelGGL(k_nearest, dim3(unsigned((n + 255) / 256)), dim3(256), 0, as_stream(stream), query, n, ref, m, ref_valid, counts, out);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
