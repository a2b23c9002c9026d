// Coordinate-manager primitives: strided coordinate sets, spatial order, coordinate hash,
// neighbour tables (rulebooks).  Replaces MinkowskiEngine's coordinate manager for the
// call sites listed in include/xm3d.h.  All integer work, HBM/L2-latency bound.
#include "common.h"

namespace xm3d {

__device__ inline int floordiv(int a, int b) {
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

__global__ void k_stride_keys(const int32_t* __restrict__ c, int64_t n, int ts, uint64_t* __restrict__ keys,
                              int32_t* __restrict__ idx, int* __restrict__ flag) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 v = reinterpret_cast<const int4*>(c)[i];
    int x = v.y, y = v.z, z = v.w;
    if (ts > 1) {
        x = floordiv(x, ts) * ts;
        y = floordiv(y, ts) * ts;
        z = floordiv(z, ts) * ts;
    }
    if (!coord_in_range(v.x, x, y, z)) *flag = XM3D_ERANGE;
    keys[i] = pack_coord(v.x, x, y, z);
    idx[i] = int32_t(i);
}

__global__ void k_heads(const uint64_t* __restrict__ sk, int64_t n, int32_t* __restrict__ flag) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || sk[i] != sk[i - 1]) ? 1 : 0;
}

__global__ void k_emit_coords(const uint64_t* __restrict__ sk, const int32_t* __restrict__ flag,
                              const int32_t* __restrict__ pos, int64_t n, int32_t* __restrict__ out,
                              int64_t* __restrict__ cnt) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = flag[i];
    const int rank = pos[i] + f - 1;
    if (f) {
        const uint64_t k = sk[i];
        int4 v;
        v.x = int(k >> 48);
        v.y = int((k >> 32) & 0xFFFF) - COORD_BIAS;
        v.z = int((k >> 16) & 0xFFFF) - COORD_BIAS;
        v.w = int(k & 0xFFFF) - COORD_BIAS;
        reinterpret_cast<int4*>(out)[rank] = v;
    }
    if (i == n - 1) *cnt = rank + 1;
}

__global__ void k_count_heads(const uint64_t* __restrict__ sk, int64_t n, int64_t* __restrict__ cnt) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    int f = (i < n && (i == 0 || sk[i] != sk[i - 1])) ? 1 : 0;
    unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(reinterpret_cast<unsigned long long*>(cnt), (unsigned long long)__popcll(m));
}

__global__ void k_hash_insert(const int32_t* __restrict__ c, int64_t n, uint64_t* __restrict__ tk,
                              int32_t* __restrict__ tv, int64_t cap, int* __restrict__ flag) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 v = reinterpret_cast<const int4*>(c)[i];
    if (!coord_in_range(v.x, v.y, v.z, v.w)) {
        *flag = XM3D_ERANGE;
        return;
    }
    const uint64_t key = pack_coord(v.x, v.y, v.z, v.w);
    const uint64_t mask = uint64_t(cap - 1);
    uint64_t slot = hash_u64(key) & mask;
    for (int64_t probe = 0; probe < cap; ++probe) {
        const unsigned long long prev =
            atomicCAS(reinterpret_cast<unsigned long long*>(&tk[slot]), (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (prev == EMPTY_KEY || prev == key) {
            atomicMin(&tv[slot], int32_t(i));  // duplicates resolve to the smallest row
            return;
        }
        slot = (slot + 1) & mask;
    }
    *flag = XM3D_ENOSPC;
}

__device__ inline int hash_lookup(const uint64_t* __restrict__ tk, const int32_t* __restrict__ tv, uint64_t mask,
                                  uint64_t key) {
    uint64_t slot = hash_u64(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const uint64_t kk = tk[slot];
        if (kk == key) return tv[slot];
        if (kk == EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
    return -1;
}

// grid: (ceil(n_out/256), K).  One thread per (output row, kernel offset); offset-major output so
// both the coordinate read and the nbr write are coalesced along rows.
__global__ void k_kernel_map(const int32_t* __restrict__ oc, int64_t n_out, const uint64_t* __restrict__ tk,
                             const int32_t* __restrict__ tv, int64_t cap, int ks, int ts, int sign,
                             int32_t* __restrict__ nbr) {
    int64_t o = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int k = blockIdx.y;
    const int base = (ks & 1) ? -(ks / 2) : 0;
    const int dx = ((k % ks) + base) * ts * sign;
    const int dy = (((k / ks) % ks) + base) * ts * sign;
    const int dz = ((k / (ks * ks)) + base) * ts * sign;
    const int4 v = reinterpret_cast<const int4*>(oc)[o];
    const int x = v.y + dx, y = v.z + dy, z = v.w + dz;
    int r = -1;
    if (coord_in_range(v.x, x, y, z)) r = hash_lookup(tk, tv, uint64_t(cap - 1), pack_coord(v.x, x, y, z));
    nbr[int64_t(k) * n_out + o] = r;
}

__global__ void k_invert_map(const int32_t* __restrict__ nbr, int64_t n_out, int64_t n_in, int32_t* __restrict__ nbr_t) {
    int64_t o = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int k = blockIdx.y;
    const int i = nbr[int64_t(k) * n_out + o];
    if (i >= 0) nbr_t[int64_t(k) * n_in + i] = int32_t(o);
}

struct CoordWs {
    uint64_t *k0, *k1;
    int32_t *i0, *i1, *flag, *pos;
    int64_t* cnt;
    void* prim;
    size_t prim_bytes, total;
};

static CoordWs carve_coord(void* ws, int64_t n) {
    Carver c(ws);
    CoordWs w;
    w.k0 = c.take<uint64_t>(n);
    w.k1 = c.take<uint64_t>(n);
    w.i0 = c.take<int32_t>(n);
    w.i1 = c.take<int32_t>(n);
    w.flag = c.take<int32_t>(n);
    w.pos = c.take<int32_t>(n);
    w.cnt = c.take<int64_t>(1);
    size_t a = sort_pairs_ws_bytes(n), b = scan_ws_bytes(n);
    w.prim_bytes = a > b ? a : b;
    w.prim = c.take<char>(w.prim_bytes);
    w.total = c.off;
    return w;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_stride_ws_bytes(int64_t n, size_t* bytes) {
    XM3D_REQUIRE(n >= 0 && bytes, "stride_ws_bytes: bad args");
    *bytes = carve_coord(nullptr, n > 0 ? n : 1).total;
    return XM3D_OK;
}

extern "C" int xm3d_coords_stride(const int32_t* coords, int64_t n, int32_t ts_out, int32_t* out_coords,
                                  int64_t* n_out, void* ws, size_t ws_bytes, void* stream) {
    XM3D_REQUIRE(n >= 0 && n < (int64_t(1) << 31) && ts_out >= 1, "coords_stride: bad n/ts");
    XM3D_REQUIRE(n_out, "coords_stride: null n_out");
    if (n == 0) {
        *n_out = 0;
        return XM3D_OK;
    }
    XM3D_REQUIRE(coords && out_coords && ws, "coords_stride: null pointer");
    XM3D_REQUIRE((reinterpret_cast<uintptr_t>(coords) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_coords) & 15) == 0,
                 "coords_stride: coordinate rows must be 16-byte aligned");
    CoordWs w = carve_coord(ws, n);
    XM3D_REQUIRE(ws_bytes >= w.total, "coords_stride: workspace %zu < %zu", ws_bytes, w.total);
    hipStream_t s = as_stream(stream);
    int* flag = device_flag();
    XM3D_REQUIRE(flag, "no device flag");
    dim3 grd((n + 255) / 256), blk(256);
    hipLaunchKernelGGL(k_stride_keys, grd, blk, 0, s, coords, n, ts_out, w.k0, w.i0, flag);
    XM3D_LAUNCH_CHECK();
    int rc = sort_pairs_u64(w.k0, w.k1, w.i0, w.i1, n, w.prim, w.prim_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_heads, grd, blk, 0, s, w.k1, n, w.flag);
    rc = exclusive_scan_i32(w.flag, w.pos, n, w.prim, w.prim_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_emit_coords, grd, blk, 0, s, w.k1, w.flag, w.pos, n, out_coords, w.cnt);
    XM3D_LAUNCH_CHECK();
    XM3D_HIP(hipMemcpyAsync(n_out, w.cnt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    int hflag = 0;
    XM3D_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    XM3D_HIP(hipStreamSynchronize(s));
    if (hflag) {
        XM3D_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
        set_error("coords_stride: coordinate outside packable range");
        return hflag;
    }
    return XM3D_OK;
}

extern "C" int xm3d_coords_order(const int32_t* coords, int64_t n, int32_t* order, int64_t* n_unique, void* ws,
                                 size_t ws_bytes, void* stream) {
    XM3D_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "coords_order: bad n");
    XM3D_REQUIRE(n_unique, "coords_order: null n_unique");
    if (n == 0) {
        *n_unique = 0;
        return XM3D_OK;
    }
    XM3D_REQUIRE(coords && order && ws, "coords_order: null pointer");
    XM3D_REQUIRE((reinterpret_cast<uintptr_t>(coords) & 15) == 0, "coords_order: coords must be 16-byte aligned");
    CoordWs w = carve_coord(ws, n);
    XM3D_REQUIRE(ws_bytes >= w.total, "coords_order: workspace %zu < %zu", ws_bytes, w.total);
    hipStream_t s = as_stream(stream);
    int* flag = device_flag();
    XM3D_REQUIRE(flag, "no device flag");
    dim3 grd((n + 255) / 256), blk(256);
    hipLaunchKernelGGL(k_stride_keys, grd, blk, 0, s, coords, n, 1, w.k0, w.i0, flag);
    XM3D_LAUNCH_CHECK();
    int rc = sort_pairs_u64(w.k0, w.k1, w.i0, order, n, w.prim, w.prim_bytes, s);
    if (rc) return rc;
    XM3D_HIP(hipMemsetAsync(w.cnt, 0, sizeof(int64_t), s));
    hipLaunchKernelGGL(k_count_heads, grd, blk, 0, s, w.k1, n, w.cnt);
    XM3D_LAUNCH_CHECK();
    XM3D_HIP(hipMemcpyAsync(n_unique, w.cnt, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    int hflag = 0;
    XM3D_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    XM3D_HIP(hipStreamSynchronize(s));
    if (hflag) {
        XM3D_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
        set_error("coords_order: coordinate outside packable range");
        return hflag;
    }
    return XM3D_OK;
}

extern "C" int xm3d_hash_build(const int32_t* coords, int64_t n, uint64_t* table_keys, int32_t* table_vals,
                               int64_t cap, void* stream) {
    XM3D_REQUIRE(n >= 0 && cap >= 2 && (cap & (cap - 1)) == 0 && cap >= 2 * n, "hash_build: cap %lld must be pow2 >= 2n (n=%lld)",
                 (long long)cap, (long long)n);
    XM3D_REQUIRE(table_keys && table_vals, "hash_build: null table");
    hipStream_t s = as_stream(stream);
    XM3D_HIP(hipMemsetAsync(table_keys, 0xFF, size_t(cap) * sizeof(uint64_t), s));
    XM3D_HIP(hipMemsetAsync(table_vals, 0x7F, size_t(cap) * sizeof(int32_t), s));
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(coords && (reinterpret_cast<uintptr_t>(coords) & 15) == 0, "hash_build: coords null or misaligned");
    int* flag = device_flag();
    XM3D_REQUIRE(flag, "no device flag");
    hipLaunchKernelGGL(k_hash_insert, dim3((n + 255) / 256), dim3(256), 0, s, coords, n, table_keys, table_vals, cap, flag);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_kernel_map(const int32_t* out_coords, int64_t n_out, const uint64_t* table_keys,
                               const int32_t* table_vals, int64_t cap, int32_t ksize, int32_t ts, int32_t sign,
                               int32_t* nbr, void* stream) {
    XM3D_REQUIRE(n_out >= 0 && ksize >= 1 && ksize <= 7 && ts >= 1 && (sign == 1 || sign == -1), "kernel_map: bad args");
    XM3D_REQUIRE(cap >= 2 && (cap & (cap - 1)) == 0, "kernel_map: cap must be a power of two");
    if (n_out == 0) return XM3D_OK;
    XM3D_REQUIRE(out_coords && table_keys && table_vals && nbr, "kernel_map: null pointer");
    XM3D_REQUIRE((reinterpret_cast<uintptr_t>(out_coords) & 15) == 0, "kernel_map: coords must be 16-byte aligned");
    const int K = ksize * ksize * ksize;
    hipLaunchKernelGGL(k_kernel_map, dim3((n_out + 255) / 256, K), dim3(256), 0, as_stream(stream), out_coords, n_out,
                       table_keys, table_vals, cap, ksize, ts, sign, nbr);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}

extern "C" int xm3d_kernel_map_invert(const int32_t* nbr, int32_t K, int64_t n_out, int64_t n_in, int32_t* nbr_t,
                                      void* stream) {
    XM3D_REQUIRE(K >= 1 && n_out >= 0 && n_in >= 0, "kernel_map_invert: bad args");
    hipStream_t s = as_stream(stream);
    if (n_in > 0) XM3D_HIP(hipMemsetAsync(nbr_t, 0xFF, size_t(K) * n_in * sizeof(int32_t), s));
    if (n_out == 0 || n_in == 0) return XM3D_OK;
    hipLaunchKernelGGL(k_invert_map, dim3((n_out + 255) / 256, K), dim3(256), 0, s, nbr, n_out, n_in, nbr_t);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
