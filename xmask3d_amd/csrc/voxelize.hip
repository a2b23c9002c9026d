// Voxelisation on the GPU: rigid transform + floor + min shift + FNV key + stable sort + unique.
// Mirrors dataset/voxelizer.py:104-122 and dataset/voxelization_utils.py:6-18,93-102 of the reference
// bit for bit: the f64 dot product uses the same fma chain numpy's matmul produces
// (fma(1,t3, fma(z,t2, fma(y,t1, x*t0)))), keys fold as h = h*PRIME ^ v, ties resolve to the
// smallest point index (radix sort is stable and values start ascending).
#include "common.h"

namespace xm3d {

struct Mat34 {
    double t[12];
};

__global__ void k_transform_floor(const double* __restrict__ xyz, int64_t n, Mat34 T, int32_t* __restrict__ g,
                                  int* __restrict__ mins) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    int v[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
    if (i < n) {
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double* t = T.t + 4 * r;
            double a = __dmul_rn(x, t[0]);
            a = fma(y, t[1], a);
            a = fma(z, t[2], a);
            a = fma(1.0, t[3], a);
            v[r] = (int)floor(a);
            g[3 * i + r] = v[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int m = v[r];
        for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0) atomicMin(&mins[r], m);
    }
}

__device__ inline uint64_t fnv3(uint64_t a, uint64_t b, uint64_t c) {
    uint64_t h = 14695981039346656037ULL;
    h *= 1099511628211ULL;
    h ^= a;
    h *= 1099511628211ULL;
    h ^= b;
    h *= 1099511628211ULL;
    h ^= c;
    return h;
}

__global__ void k_shift_key(int32_t* __restrict__ g, int64_t n, const int* __restrict__ mins, uint64_t* __restrict__ keys,
                            int32_t* __restrict__ idx) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int a = g[3 * i] - mins[0], b = g[3 * i + 1] - mins[1], c = g[3 * i + 2] - mins[2];
    g[3 * i] = a;
    g[3 * i + 1] = b;
    g[3 * i + 2] = c;
    keys[i] = fnv3(uint64_t(int64_t(a)), uint64_t(int64_t(b)), uint64_t(int64_t(c)));
    idx[i] = int32_t(i);
}

__global__ void k_fnv_only(const int32_t* __restrict__ g, int64_t n, uint64_t* __restrict__ keys) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = fnv3(uint64_t(int64_t(g[3 * i])), uint64_t(int64_t(g[3 * i + 1])), uint64_t(int64_t(g[3 * i + 2])));
}

__global__ void k_head_flags(const uint64_t* __restrict__ sk, int64_t n, int32_t* __restrict__ flag) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || sk[i] != sk[i - 1]) ? 1 : 0;
}

__global__ void k_emit_voxels(const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                              const int32_t* __restrict__ sidx, const int32_t* __restrict__ gshift, int64_t n,
                              int32_t* __restrict__ grid_out, int64_t* __restrict__ inds, int64_t* __restrict__ inverse,
                              int64_t* __restrict__ n_unique) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = flag[i];
    const int rank = pos[i] + f - 1;
    const int src = sidx[i];
    inverse[src] = rank;
    if (f) {
        inds[rank] = src;
        grid_out[3 * rank] = gshift[3 * src];
        grid_out[3 * rank + 1] = gshift[3 * src + 1];
        grid_out[3 * rank + 2] = gshift[3 * src + 2];
    }
    if (i == n - 1) *n_unique = rank + 1;
}

struct VoxWs {
    int32_t* g;
    uint64_t *k0, *k1;
    int32_t *i0, *i1, *flag, *pos;
    int* mins;
    int64_t* cnt;
    void* prim;
    size_t prim_bytes;
    size_t total;
};

static VoxWs carve_vox(void* ws, int64_t n) {
    Carver c(ws);
    VoxWs w;
    w.g = c.take<int32_t>(3 * n);
    w.k0 = c.take<uint64_t>(n);
    w.k1 = c.take<uint64_t>(n);
    w.i0 = c.take<int32_t>(n);
    w.i1 = c.take<int32_t>(n);
    w.flag = c.take<int32_t>(n);
    w.pos = c.take<int32_t>(n);
    w.mins = c.take<int>(4);
    w.cnt = c.take<int64_t>(1);
    size_t a = sort_pairs_ws_bytes(n), b = scan_ws_bytes(n);
    w.prim_bytes = a > b ? a : b;
    w.prim = c.take<char>(w.prim_bytes);
    w.total = c.off;
    return w;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_voxelize_ws_bytes(int64_t n, size_t* bytes) {
    XM3D_REQUIRE(n >= 0 && bytes, "voxelize_ws_bytes: bad args");
    *bytes = carve_vox(nullptr, n > 0 ? n : 1).total;
    return XM3D_OK;
}

extern "C" int xm3d_fnv_keys(const int32_t* grid, int64_t n, uint64_t* keys, void* stream) {
    XM3D_REQUIRE(n >= 0, "fnv_keys: n < 0");
    if (n == 0) return XM3D_OK;
    hipLaunchKernelGGL(k_fnv_only, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), grid, n, keys);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_voxelize(const double* xyz, int64_t n, const double* T16, int32_t* grid, int64_t* inds,
                             int64_t* inverse, int64_t* n_unique, void* ws, size_t ws_bytes, void* stream) {
    XM3D_REQUIRE(n > 0 && n < (int64_t(1) << 31), "voxelize: n=%lld out of range (reference asserts n>0)", (long long)n);
    XM3D_REQUIRE(xyz && T16 && grid && inds && inverse && n_unique && ws, "voxelize: null pointer");
    VoxWs w = carve_vox(ws, n);
    XM3D_REQUIRE(ws_bytes >= w.total, "voxelize: workspace %zu < %zu", ws_bytes, w.total);
    hipStream_t s = as_stream(stream);
    Mat34 T;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) T.t[4 * r + c] = T16[4 * r + c];
    const int init[4] = {INT32_MAX, INT32_MAX, INT32_MAX, 0};
    XM3D_HIP(hipMemcpyAsync(w.mins, init, sizeof(init), hipMemcpyHostToDevice, s));
    dim3 grd((n + 255) / 256), blk(256);
    hipLaunchKernelGGL(k_transform_floor, grd, blk, 0, s, xyz, n, T, w.g, w.mins);
    hipLaunchKernelGGL(k_shift_key, grd, blk, 0, s, w.g, n, w.mins, w.k0, w.i0);
    XM3D_LAUNCH_CHECK();
    int rc = sort_pairs_u64(w.k0, w.k1, w.i0, w.i1, n, w.prim, w.prim_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_head_flags, grd, blk, 0, s, w.k1, n, w.flag);
    rc = exclusive_scan_i32(w.flag, w.pos, n, w.prim, w.prim_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_emit_voxels, grd, blk, 0, s, w.flag, w.pos, w.i1, w.g, n, grid, inds, inverse, w.cnt);
    XM3D_LAUNCH_CHECK();
    XM3D_HIP(hipMemcpyAsync(n_unique, w.cnt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    XM3D_HIP(hipStreamSynchronize(s));
    return XM3D_OK;
}
