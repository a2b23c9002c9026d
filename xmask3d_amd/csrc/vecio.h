// 16-byte vector load/store of f32 x4 / bf16 x8 into f32 registers, shared by the streaming pointwise kernels.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

namespace xm3d {

template <typename T>
struct VecIO;
template <>
struct VecIO<float> {
    static constexpr int N = 4;
    __device__ static void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    __device__ static float scalar(const float* p) { return *p; }
};
template <>
struct VecIO<__hip_bfloat16> {
    static constexpr int N = 8;
    __device__ static void load(const __hip_bfloat16* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    __device__ static void store(__hip_bfloat16* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);
            w[i] = unsigned(*reinterpret_cast<const unsigned short*>(&lo)) | (unsigned(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static float scalar(const __hip_bfloat16* p) { return __bfloat162float(*p); }
};

}  // namespace xm3d
