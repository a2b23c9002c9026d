// Shared helpers for libxm3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/xm3d.h"

namespace xm3d {

void set_error(const char* fmt, ...);

// device-side sticky error flag (XM3D_ERANGE / XM3D_ENOSPC), see xm3d_check_flag()
int* device_flag();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define XM3D_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            xm3d::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return XM3D_EHIP;                                                               \
        }                                                                                   \
    } while (0)

#define XM3D_REQUIRE(cond, ...)             \
    do {                                    \
        if (!(cond)) {                      \
            xm3d::set_error(__VA_ARGS__);   \
            return XM3D_EINVAL;             \
        }                                   \
    } while (0)

#define XM3D_LAUNCH_CHECK()                                                                 \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess) {                                                             \
            xm3d::set_error("%s:%d launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return XM3D_EHIP;                                                               \
        }                                                                                   \
    } while (0)

// "first call on this device" flag for per-device function attributes (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per
// device: a process-wide flag would leave a second GPU of the process unconfigured).  A benign race sets the attribute twice.
struct DeviceOnce {
    bool done[64] = {};
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        const bool f = !done[d];
        done[d] = true;
        return f;
    }
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve typed, 256-byte aligned pieces out of a caller-provided workspace
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + off);
        off += align_up(count * sizeof(T), 256);
        return p;
    }
};

constexpr int COORD_BIAS = 1 << 15;

__host__ __device__ inline uint64_t pack_coord(int b, int x, int y, int z) {
    return (uint64_t(uint32_t(b)) << 48) | (uint64_t(uint32_t(x + COORD_BIAS) & 0xFFFFu) << 32) |
           (uint64_t(uint32_t(y + COORD_BIAS) & 0xFFFFu) << 16) | uint64_t(uint32_t(z + COORD_BIAS) & 0xFFFFu);
}
__device__ inline bool coord_in_range(int b, int x, int y, int z) {
    return b >= 0 && b < 32768 && x >= -COORD_BIAS && x < COORD_BIAS && y >= -COORD_BIAS && y < COORD_BIAS &&
           z >= -COORD_BIAS && z < COORD_BIAS;
}
__device__ inline uint64_t hash_u64(uint64_t k) {  // murmur3 finaliser
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFULL;

// stable sort of (u64 key, i32 value) pairs + helpers built on rocPRIM (sort.hip)
size_t sort_pairs_ws_bytes(int64_t n);
int sort_pairs_u64(const uint64_t* kin, uint64_t* kout, const int32_t* vin, int32_t* vout, int64_t n, void* ws,
                   size_t ws_bytes, hipStream_t s);
size_t scan_ws_bytes(int64_t n);
int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes, hipStream_t s);

// out = epi(sum_z slab[z]) in fixed z order (spconv.hip; shared by the split-K paths of both sparse-conv kernels)
int launch_slab_reduce(const float* slab, int ksplit, int64_t n_out, int cout, const float* scale, const float* shift,
                       const float* residual, int relu, float* out, hipStream_t s, void* out_hi = nullptr, void* out_lo = nullptr,
                       const void* residual_bf = nullptr);

// f32 x 4 -> bf16 hi / lo parts (x = hi + lo up to 2^-17 |x|), 8-byte stores: the pre-split activation format of the
// split-operand sparse conv (spconv_split.hip)
#ifdef __HIPCC__
__device__ __forceinline__ void store_split4(__bf16* hi, __bf16* lo, const float __attribute__((ext_vector_type(4))) v) {
    typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
    bf16x4s h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (__bf16)v[i];
        l[i] = (__bf16)(v[i] - (float)h[i]);
    }
    *reinterpret_cast<bf16x4s*>(hi) = h;
    *reinterpret_cast<bf16x4s*>(lo) = l;
}
#endif

}  // namespace xm3d
