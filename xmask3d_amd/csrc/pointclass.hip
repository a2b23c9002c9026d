// Per-point class label from a (Np, K) f32 feature table and the normalised text embeddings, one pass over the features:
//
//     label[p] = argmax_c gate_c( value_c(p) ),     logit_c = [scale / |x_p|] * <x_p, t_c>
//
// the three label chains of the inference post-processing (/root/reference/run/infer.py:489-507 gate, :556-612 fused / 2D-only /
// 3D-only predictions; pipeline.postprocess_scene): F.normalize over (Np, 768) -> `@ text.t()` -> * scale -> softmax -> geometric
// ensemble with the open-vocabulary probabilities of the point's mask query -> base / novel gate -> arg-max.  The op chain reads
// and writes the 170 k x 768 features twice before the GEMM and makes ~12 passes over (Np, C); here the features are read ONCE,
// the (32 points x C) logits never leave the chip.
//   * <x_p, t_c> on the matrix cores in EXACT f32 (v_mfma_f32_32x32x2_f32: a bf16 product would move the scale-100 logits by 0.2):
//     points are the rows (A operand), classes the columns (B operand, C <= 32).  The k index of MFMA step (j, i), i = 0..3, half h
//     is channel 8 j + 4 h + i on BOTH operands, so a lane's A values for four consecutive steps are one 16-byte load of its row,
//     and its B values one ds_read_b128 of the text table staged in LDS in that order ([j][lane][4], 96 KiB for K = 768).
//   * |x_p|^2 is accumulated from the same loads (lane (r, h) sees half of row r; one exchange with lane ^ 32).
//   * the accumulator (class on the lane, 16 points in registers) goes through a wave-private LDS tile, then lane p < 32 owns point
//     p: softmax over the C classes, ensemble, gate, first-maximum arg-max (NaN counts as maximal, like torch.argmax).
//   * row_index: label of point p computed from row row_index[p] (the 2D-only labels take the nearest covered point's features).
// Bound: HBM (Np x K x 4 bytes read once) against f32 MFMA time 2 Np K 32 / 157 TFLOP/s - the same order at K = 768.
#include <algorithm>

#include "common.h"

namespace xm3d {

typedef float pc_f32x16 __attribute__((ext_vector_type(16)));

constexpr int PC_TSTR = 33;  // floats per point row of the transposition tile

struct PointClassArgs {
    const float* x;              // (rows, K), row stride ldx
    const int64_t* row_index;    // (Np) or null
    const float* text;           // (C, K) unit rows
    const float* scale;          // device scalar or null (1)
    const int64_t* binary_pred;  // (Np) != 0: base-predicted point
    const uint8_t* base_mask;    // (C)
    const uint8_t* novel_mask;   // (C)
    // ensemble (mode 1) only
    const uint8_t* masks;        // (Np, Q) point-in-mask flags, at most one set per point
    const int64_t* vid;          // (Np) batch entry of the point
    const float* open_p;         // (B, Q, C) open-vocabulary class probabilities per mask query
    const float* overlap;        // (C) 1 base / 0 novel
    int64_t* label;              // (Np)
    int64_t np, ldx;
    int C, K, Q, mode;           // mode 0: arg-max of gated logits; 1: normalise, scale, softmax, ensemble
    float base_ratio, novel_ratio;
};

__global__ __launch_bounds__(256) void k_point_class(const PointClassArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int C = a.C, K = a.K, nj = K / 8;
    float4* const tl = reinterpret_cast<float4*>(smem);  // [j][lane]: text[class lane & 31][8 j + 4 (lane >> 5) .. + 3]
    float* const tile = reinterpret_cast<float*>(smem + size_t(nj) * 64 * 16) + wave * 32 * PC_TSTR;
    for (int i = tid; i < nj * 64; i += 256) {
        const int l = i & 63, j = i >> 6, c = l & 31;
        tl[i] = c < C ? *reinterpret_cast<const float4*>(a.text + int64_t(c) * K + 8 * j + 4 * (l >> 5)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float scale = a.scale ? *a.scale : 1.f;
    const int64_t ntile = (a.np + 31) / 32;
    for (int64_t t = int64_t(blockIdx.x) * 4 + wave; t < ntile; t += int64_t(gridDim.x) * 4) {
        const int64_t p = t * 32 + l31;
        const int64_t pv = p < a.np ? p : a.np - 1;
        const int64_t src = a.row_index ? a.row_index[pv] : pv;
        const float* const xp = a.x + src * a.ldx + 4 * h;
        pc_f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float nrm = 0.f;
        for (int j = 0; j < nj; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(xp + 8 * j);
            const float4 w = tl[j * 64 + lane];
            nrm = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, nrm))));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, w.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, w.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, w.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, w.w, acc, 0, 0, 0);
        }
        nrm += __shfl_xor(nrm, 32);
        // D: column = class l31, register i = point (i & 3) + 8 (i >> 2) + 4 h  ->  tile[point][class]
#pragma unroll
        for (int i = 0; i < 16; ++i) tile[((i & 3) + 8 * (i >> 2) + 4 * h) * PC_TSTR + l31] = acc[i];
        __builtin_amdgcn_wave_barrier();  // LDS is in order per wave: the rows are read back by other lanes below
        if (h == 0 && p < a.np) {
            const float* const row = tile + l31 * PC_TSTR;
            const bool base_pt = a.binary_pred[p] != 0;
            int q = -1;
            if (a.mode == 1) {
                const uint8_t* const m = a.masks + p * a.Q;
                for (int i = a.Q - 1; i >= 0; --i)
                    if (m[i]) q = i;  // first set flag (torch: masks.to(uint8).argmax(1))
            }
            float mx = -INFINITY, sum = 0.f;
            float inv = 1.f;
            if (a.mode == 1) {
                inv = scale / fmaxf(sqrtf(nrm), 1e-12f);  // scale * <x / max(|x|, eps), t>
                for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c] * inv);
                for (int c = 0; c < C; ++c) sum += expf(row[c] * inv - mx);
            }
            const float* const po = (a.mode == 1 && q >= 0) ? a.open_p + (a.vid[p] * a.Q + q) * C : nullptr;
            float best = 0.f;
            int best_c = 0;
            for (int c = 0; c < C; ++c) {
                float v = row[c] * inv;
                if (a.mode == 1) {
                    const float pr = expf(v - mx) / sum;
                    v = pr;
                    if (po) {
                        const float ov = a.overlap[c];
                        const float bb = logf(powf(pr, a.base_ratio) * powf(po[c], 1.f - a.base_ratio)) * ov;
                        const float nn = logf(powf(pr, a.novel_ratio) * powf(po[c], 1.f - a.novel_ratio)) * (1.f - ov);
                        v = bb + nn;
                    }
                }
                if (base_pt ? a.novel_mask[c] : a.base_mask[c]) v = -1e10f;
                if (c == 0 || v > best || (v != v && best == best)) best = v, best_c = c;
            }
            a.label[p] = best_c;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_point_class(const float* x, int64_t ldx, const int64_t* row_index, int64_t np, const float* text, int32_t C, int32_t K,
                                const float* scale, const int64_t* binary_pred, const uint8_t* base_mask, const uint8_t* novel_mask, int32_t mode,
                                const uint8_t* masks, int32_t Q, const int64_t* vid, const float* open_p, const float* overlap, float base_ratio,
                                float novel_ratio, int64_t* label, void* stream) {
    if (np == 0) return XM3D_OK;
    XM3D_REQUIRE(x && text && binary_pred && base_mask && novel_mask && label, "point_class: null pointer");
    XM3D_REQUIRE(np > 0 && C >= 1 && C <= 32 && K >= 8 && K % 8 == 0 && K <= 1024 && ldx >= K && ldx % 4 == 0, "point_class: C <= 32, K % 8 == 0, K <= 1024");
    XM3D_REQUIRE(mode == 0 || (mode == 1 && masks && vid && open_p && overlap && Q >= 1), "point_class: mode 1 needs masks, vid, open_p, overlap");
    XM3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(text)) & 15) == 0, "point_class: x / text must be 16-byte aligned");
    PointClassArgs a;
    a.x = x, a.row_index = row_index, a.text = text, a.scale = scale, a.binary_pred = binary_pred, a.base_mask = base_mask, a.novel_mask = novel_mask;
    a.masks = masks, a.vid = vid, a.open_p = open_p, a.overlap = overlap, a.label = label;
    a.np = np, a.ldx = ldx, a.C = C, a.K = K, a.Q = Q, a.mode = mode, a.base_ratio = base_ratio, a.novel_ratio = novel_ratio;
    const int lds = (K / 8) * 64 * 16 + 4 * 32 * PC_TSTR * 4;
    static DeviceOnce configured;  // the attribute is per device
    if (configured.first()) {
        XM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_point_class), hipFuncAttributeMaxDynamicSharedMemorySize, 1024 / 8 * 64 * 16 + 4 * 32 * PC_TSTR * 4));
    }
    const int64_t ntile = (np + 31) / 32;
    const unsigned grid = unsigned(std::min<int64_t>((ntile + 3) / 4, 512));
    hipLaunchKernelGGL(k_point_class, dim3(grid), dim3(256), lds, as_stream(stream), a);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
