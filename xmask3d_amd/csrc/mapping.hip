// Point cloud -> image pixel mapping with frustum / border / optional depth-occlusion test, on the device.
//
// Replaces PointCloudToImageMapper.compute_mapping (/root/reference/models/utils/fusion_util.py:46-142), which the reference's
// data loaders run in numpy per view (dataset/data_loader_infer.py:161-176, data_loader.py:173-176): the world->camera
// transform, the pinhole projection with the fixed ScanNet intrinsics, round-half-to-even to a pixel, the `cut_bound` border and
// the |depth - z| <= vis_thres * depth visibility test.  All arithmetic in f64 with the reference's operation order (the 4x4
// inverse is taken on the host, like np.linalg.inv); one thread per point, 24 B in + 12 B out per point: HBM streaming.
#include "common.h"

namespace xm3d {

struct MapParams {
    double w2c[12];  // rows 0..2 of world->camera (row-major 3x4)
    double fx, fy, cx, cy;
    int width, height, cut_bound, dh, dw;
    double vis_thres;
};

__global__ void k_compute_mapping(const double* __restrict__ pts, int64_t n, MapParams P, const double* __restrict__ depth,
                                  int32_t* __restrict__ mapping) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    // p = W @ [x, y, z, 1]: the products summed left to right like a row-times-column dot product, no fused multiply-add
    const double px_ = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(P.w2c[0], x), __dmul_rn(P.w2c[1], y)), __dmul_rn(P.w2c[2], z)), P.w2c[3]);
    const double py_ = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(P.w2c[4], x), __dmul_rn(P.w2c[5], y)), __dmul_rn(P.w2c[6], z)), P.w2c[7]);
    const double pz = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(P.w2c[8], x), __dmul_rn(P.w2c[9], y)), __dmul_rn(P.w2c[10], z)), P.w2c[11]);
    const double safe_z = fabs(pz) < 1e-8 ? 1.0 : pz;
    const double u = __dadd_rn(__ddiv_rn(__dmul_rn(px_, P.fx), safe_z), P.cx);
    const double v = __dadd_rn(__ddiv_rn(__dmul_rn(py_, P.fy), safe_z), P.cy);
    // np.round = round half to even; astype(int) of the rounded value
    const double ru = rint(u), rv = rint(v);
    bool inside = pz > 0 && ru >= P.cut_bound && rv >= P.cut_bound && ru < P.width - P.cut_bound && rv < P.height - P.cut_bound;
    const int iu = inside ? int(ru) : 0, iv = inside ? int(rv) : 0;
    if (inside && depth) {
        inside = false;
        if (iv >= 0 && iv < P.dh && iu >= 0 && iu < P.dw) {
            const double d = depth[int64_t(iv) * P.dw + iu];
            inside = fabs(d - pz) <= P.vis_thres * d;
        }
    }
    mapping[3 * i] = inside ? iv : 0;
    mapping[3 * i + 1] = inside ? iu : 0;
    mapping[3 * i + 2] = inside ? 1 : 0;
}

}  // namespace xm3d

using namespace xm3d;

extern "C" int xm3d_compute_mapping(const double* pts, int64_t n, const double* world_to_camera, const double* intrinsic4,
                                    int32_t width, int32_t height, int32_t cut_bound, const double* depth, int32_t depth_h,
                                    int32_t depth_w, double vis_thres, int32_t* mapping, void* stream) {
    XM3D_REQUIRE(n >= 0 && width > 0 && height > 0 && cut_bound >= 0, "compute_mapping: bad sizes");
    if (n == 0) return XM3D_OK;
    XM3D_REQUIRE(pts && world_to_camera && intrinsic4 && mapping, "compute_mapping: null pointer");
    XM3D_REQUIRE(!depth || (depth_h > 0 && depth_w > 0), "compute_mapping: depth map without a shape");
    MapParams P;
    for (int i = 0; i < 12; ++i) P.w2c[i] = world_to_camera[i];  // host pointers: a 4x4 row-major matrix, rows 0..2 used
    P.fx = intrinsic4[0];
    P.cx = intrinsic4[2];
    P.fy = intrinsic4[5];
    P.cy = intrinsic4[6];
    P.width = width;
    P.height = height;
    P.cut_bound = cut_bound;
    P.dh = depth_h;
    P.dw = depth_w;
    P.vis_thres = vis_thres;
    hipLaunchKernelGGL(k_compute_mapping, dim3(unsigned((n + 255) / 256)), dim3(256), 0, as_stream(stream), pts, n, P, depth, mapping);
    XM3D_LAUNCH_CHECK();
    return XM3D_OK;
}
