// Batched linear sum assignment (Hungarian / shortest augmenting path) on the device.
//
// Replaces the host round trip of the reference's matcher - cost matrix on the GPU, `.cpu()`, scipy.optimize.
// linear_sum_assignment, ten times per training iteration (third_party/Mask2Former/mask2former/modeling/matcher.py:95-156,
// called from criterion.py for the main and the nine auxiliary decoder outputs): all matrices of an iteration are solved by
// ONE launch, one wave per matrix, and the matched indices never leave the device.
//
// Problem per matrix: cost (Q x T), Q <= 64 mask queries (50 on this path), T <= 256 masks present in the image: match
// min(Q, T) (query, target) pairs, each query and each target at most once, minimising the summed cost (scipy's rectangular
// semantics: T <= Q matches every target, T > Q - ScanNet200 images with more label values than queries - every query).  Algorithm: the O(T^2 Q) potentials form (Jonker-Volgenant style
// shortest augmenting paths): targets are inserted one at a time; lane j owns query column j (its potential, its slack to the
// alternating tree, its matched target and its predecessor on the path), the argmin over the free columns is a wave
// reduction (lowest column wins ties), target potentials live in LDS.  Arithmetic in f64 like scipy (the optimum of an f32
// matrix is then the same set of total costs; with tied costs several optimal assignments exist - the result is one of them).
// Output order follows scipy: pairs sorted by ascending query index.
#include "common.h"

namespace xm3d {

constexpr int ASG_CPL = 4;                 // columns per lane
constexpr int ASG_MAXC = 64 * ASG_CPL;     // columns (the longer side of the matrix) per problem
constexpr int ASG_MAXR = 64;               // rows (the shorter side)

// rows = the shorter side (n <= 64), columns = the longer side (m <= 256, ASG_CPL per lane: column j-1 lives in slot
// (j-1) >> 6 of lane (j-1) & 63).  `swap` = rows are queries and columns targets (T > Q); otherwise rows are targets.
__global__ __launch_bounds__(64) void k_assign(const float* __restrict__ cost, int64_t mat_stride, int32_t row_stride,
                                               const int32_t* __restrict__ n_targets, int32_t Q, int64_t out_stride,
                                               int64_t* __restrict__ out_q, int64_t* __restrict__ out_t) {
    __shared__ double u[ASG_MAXR + 1];  // row potentials (1-based)
    __shared__ int qmatch[ASG_MAXC];    // swap case: target matched to each query (for the sorted output)
    const int lane = threadIdx.x;
    const int64_t mi = blockIdx.x;
    const int T = n_targets[mi];
    const bool swap = T > Q;
    const int n = swap ? Q : T, m = swap ? T : Q;
    const float* C = cost + mi * mat_stride;  // C[q * row_stride + t]
    auto cost_at = [&](int row, int col) {    // 0-based
        return swap ? double(C[int64_t(row) * row_stride + col]) : double(C[int64_t(col) * row_stride + row]);
    };
    double v[ASG_CPL], minv[ASG_CPL];
    int p[ASG_CPL], way[ASG_CPL];
    bool used[ASG_CPL];
#pragma unroll
    for (int s = 0; s < ASG_CPL; ++s) {
        v[s] = 0.0;
        p[s] = 0;
    }
    auto col_p = [&](int j) {  // p of column j (1-based, uniform j): owner lane broadcasts
        const int c = j - 1, l = c & 63, sl = c >> 6;
        int r = 0;
#pragma unroll
        for (int s = 0; s < ASG_CPL; ++s) {
            const int t = __shfl(p[s], l);
            if (s == sl) r = t;
        }
        return r;
    };
    auto col_way = [&](int j) {
        const int c = j - 1, l = c & 63, sl = c >> 6;
        int r = 0;
#pragma unroll
        for (int s = 0; s < ASG_CPL; ++s) {
            const int t = __shfl(way[s], l);
            if (s == sl) r = t;
        }
        return r;
    };
    u[lane] = 0.0;
    if (lane == 0) u[ASG_MAXR] = 0.0;
    __syncthreads();
    const double INF = 1e300;
    for (int i = 1; i <= n; ++i) {
        const int p0 = i;  // row matched to the virtual root column during this insertion
        int j0 = 0;
        bool used0 = false;
#pragma unroll
        for (int s = 0; s < ASG_CPL; ++s) {
            minv[s] = INF;
            used[s] = false;
            way[s] = 0;
        }
        for (int guard = 0; guard <= m + 1; ++guard) {  // grow the alternating tree until a free column is reached
            if (j0 == 0) used0 = true;
            const int i0 = j0 == 0 ? p0 : col_p(j0);
            const double ui0 = u[i0];
            double best = INF;
            int bestj = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < ASG_CPL; ++s) {
                const int j = 64 * s + lane + 1;
                if (j == j0) used[s] = true;
                if (j <= m && !used[s]) {
                    const double cur = cost_at(i0 - 1, j - 1) - ui0 - v[s];
                    if (cur < minv[s]) {
                        minv[s] = cur;
                        way[s] = j0;
                    }
                    if (minv[s] < best) {  // slots ascend in column index: strict < keeps the lowest column on ties
                        best = minv[s];
                        bestj = j;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {  // wave arg-min, lowest column on ties
                const double ob = __shfl_xor(best, off);
                const int oj = __shfl_xor(bestj, off);
                if (ob < best || (ob == best && oj < bestj)) {
                    best = ob;
                    bestj = oj;
                }
            }
            const double delta = best;
            const int j1 = bestj;
            if (used0 && lane == 0) u[p0] += delta;
#pragma unroll
            for (int s = 0; s < ASG_CPL; ++s) {
                const int j = 64 * s + lane + 1;
                if (j <= m) {
                    if (used[s]) {
                        u[p[s]] += delta;  // distinct rows per tree column: no conflict
                        v[s] -= delta;
                    } else {
                        minv[s] -= delta;
                    }
                }
            }
            __syncthreads();
            if (j1 > m) break;  // no free column left (cannot happen while n <= m)
            j0 = j1;
            if (col_p(j0) == 0) break;  // free column reached
        }
        for (int guard = 0; guard <= m + 1 && j0 != 0; ++guard) {  // augment along the path back to the root
            const int j1 = col_way(j0);
            const int pj1 = j1 == 0 ? p0 : col_p(j1);
#pragma unroll
            for (int s = 0; s < ASG_CPL; ++s)
                if (64 * s + lane + 1 == j0) p[s] = pj1;
            j0 = j1;
        }
    }
    // emit (query, target) pairs sorted by ascending query index (scipy's order)
    if (!swap) {  // columns are queries: rank of a matched column = number of matched lower columns
        int base = 0;
#pragma unroll
        for (int s = 0; s < ASG_CPL; ++s) {
            const bool hit = 64 * s + lane < m && p[s] != 0;
            const unsigned long long mm = __ballot(hit);
            if (hit) {
                const int rank = base + __popcll(mm & ((1ull << lane) - 1ull));
                out_q[mi * out_stride + rank] = 64 * s + lane;
                out_t[mi * out_stride + rank] = p[s] - 1;
            }
            base += __popcll(mm);
        }
    } else {      // columns are targets, rows are queries: every query is matched; gather per query through LDS
#pragma unroll
        for (int s = 0; s < ASG_CPL; ++s)
            if (64 * s + lane < m && p[s] != 0) qmatch[p[s] - 1] = 64 * s + lane;
        __syncthreads();
        if (lane < n) {
            out_q[mi * out_stride + lane] = lane;
            out_t[mi * out_stride + lane] = qmatch[lane];
        }
    }
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_linear_sum_assignment(const float* cost, int64_t n_mat, int32_t Q, int32_t T_max, const int32_t* n_targets,
                                          int64_t* out_q, int64_t* out_t, void* stream) {
    XM3D_REQUIRE(n_mat >= 0 && Q >= 1 && T_max >= 0 && (Q <= ASG_MAXR || T_max <= ASG_MAXR) && Q <= ASG_MAXC && T_max <= ASG_MAXC,
                 "linear_sum_assignment: the shorter side must be <= %d and the longer <= %d (Q=%d T_max=%d)", ASG_MAXR, ASG_MAXC, Q, T_max);
    if (n_mat == 0 || T_max == 0) return XM3D_OK;
    XM3D_REQUIRE(cost && n_targets && out_q && out_t, "linear_sum_assignment: null pointer");
    XM3D_REQUIRE(Q <= ASG_MAXR, "linear_sum_assignment: more than %d queries is not on this path (Q=%d)", ASG_MAXR, Q);
    hipLaunchKernelGGL(k_assign, dim3(unsigned(n_mat)), dim3(64), 0, as_stream(stream), cost, int64_t(Q) * T_max, T_max, n_targets, Q,
                       int64_t(T_max), out_q, out_t);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
