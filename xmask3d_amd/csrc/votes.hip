// Scene votes: per-view class predictions of the visible points -> vote table -> label per scene point.
//
// Replaces the reference's per-view `scene_pred[mask_2d, logits_pred] += 1` (x3: fused / 2D-only / 3D-only), `counter[mask_2d] +=
// 1` and `torch.max(scene_pred, dim=1)` (run/infer.py:642-661,690-694) for ALL views of a group of scenes in two launches.  As torch
// ops that was, per scene and prediction kind, an index_put_(accumulate=True) - a sort, a segmented reduction and four
// bounds-check reductions with device asserts - about 100 launches per scene on the serial tail of the forward.
//   k_vote        one thread per (kind, visible point): votes[kind][row[p]][pred[kind][p]] += 1 (integer atomics: exact and order-free)
//   k_vote_label  one thread per (kind, scene point): first maximal class (torch.max's documented tie rule on the CPU tensors
//                 of the reference), and for kind 0 the "seen by any view" flag (counter != 0)
// HBM streaming: K*R*C*4 B of table zeroed, updated sparsely and read once (K = 3 kinds, R = 480 k rows, C = 19 classes: 109 MB).
#include "common.h"

namespace xm3d {

__global__ __launch_bounds__(256) void k_vote(const int64_t* __restrict__ rows, const int64_t* __restrict__ pred, int64_t np,
                                              int64_t n_rows, int32_t n_cls, int32_t* __restrict__ votes, int32_t* __restrict__ err) {
    const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int kind = blockIdx.y;
    if (p >= np) return;
    const int64_t r = rows[p], c = pred[int64_t(kind) * np + p];
    if (r < 0 || r >= n_rows || c < 0 || c >= n_cls) {  // what index_put_ would have asserted on
        *err = XM3D_ERANGE;  // sticky device flag, read by xm3d_check_flag()
        return;
    }
    atomicAdd(&votes[(int64_t(kind) * n_rows + r) * n_cls + c], 1);
}

__global__ __launch_bounds__(256) void k_vote_label(const int32_t* __restrict__ votes, int64_t n_rows, int32_t n_cls,
                                                    int64_t* __restrict__ label, uint8_t* __restrict__ seen) {
    const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int kind = blockIdx.y;
    if (r >= n_rows) return;
    const int32_t* v = votes + (int64_t(kind) * n_rows + r) * n_cls;
    int best = v[0], bi = 0, total = v[0];
    for (int c = 1; c < n_cls; ++c) {
        const int x = v[c];
        total += x;
        if (x > best) {  // strict: the first maximal class wins
            best = x;
            bi = c;
        }
    }
    label[int64_t(kind) * n_rows + r] = bi;
    if (kind == 0) seen[r] = total > 0;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_scene_votes(const int64_t* rows, const int64_t* pred, int32_t n_kinds, int64_t n_points, int64_t n_rows,
                                int32_t n_cls, int32_t* votes, int64_t* label, uint8_t* seen, void* stream) {
    XM3D_REQUIRE(n_kinds >= 1 && n_kinds <= 65535 && n_points >= 0 && n_rows >= 0 && n_cls >= 1,
                 "scene_votes: bad sizes kinds=%d points=%lld rows=%lld classes=%d", n_kinds, (long long)n_points, (long long)n_rows, n_cls);
    if (n_rows == 0) return XM3D_OK;
    XM3D_REQUIRE(votes && label && seen && (n_points == 0 || (rows && pred)), "scene_votes: null pointer");
    hipStream_t s = as_stream(stream);
    XM3D_HIP(hipMemsetAsync(votes, 0, size_t(n_kinds) * size_t(n_rows) * size_t(n_cls) * 4, s));
    int32_t* err = device_flag();
    XM3D_REQUIRE(err, "scene_votes: no device flag");
    if (n_points > 0)
        hipLaunchKernelGGL(k_vote, dim3(unsigned((n_points + 255) / 256), unsigned(n_kinds)), dim3(256), 0, s, rows, pred, n_points, n_rows,
                           n_cls, votes, err);
    hipLaunchKernelGGL(k_vote_label, dim3(unsigned((n_rows + 255) / 256), unsigned(n_kinds)), dim3(256), 0, s, votes, n_rows, n_cls, label,
                       seen);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
