// Error plumbing + sort/scan primitives (rocPRIM device algorithms) shared by the
// voxeliser and the coordinate manager.
#include <stdarg.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace xm3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int* g_flag[64] = {nullptr};

int* device_flag() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!g_flag[dev]) {
        if (hipMalloc(&g_flag[dev], sizeof(int)) != hipSuccess) return nullptr;
        (void)hipMemset(g_flag[dev], 0, sizeof(int));
    }
    return g_flag[dev];
}

size_t sort_pairs_ws_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const uint64_t*, uint64_t*, const int32_t*, int32_t*>(
        nullptr, bytes, nullptr, nullptr, nullptr, nullptr, size_t(n > 0 ? n : 1), 0, 64, nullptr, false);
    return align_up(bytes, 256);
}

int sort_pairs_u64(const uint64_t* kin, uint64_t* kout, const int32_t* vin, int32_t* vout, int64_t n, void* ws,
                   size_t ws_bytes, hipStream_t s) {
    if (n == 0) return XM3D_OK;
    size_t need = ws_bytes;
    XM3D_HIP(rocprim::radix_sort_pairs(ws, need, kin, kout, vin, vout, size_t(n), 0, 64, s, false));
    return XM3D_OK;
}

size_t scan_ws_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan<rocprim::default_config, const int32_t*, int32_t*, int32_t>(
        nullptr, bytes, nullptr, nullptr, 0, size_t(n > 0 ? n : 1), rocprim::plus<int32_t>(), nullptr, false);
    return align_up(bytes, 256);
}

int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes, hipStream_t s) {
    if (n == 0) return XM3D_OK;
    size_t need = ws_bytes;
    XM3D_HIP(rocprim::exclusive_scan(ws, need, in, out, int32_t(0), size_t(n), rocprim::plus<int32_t>(), s, false));
    return XM3D_OK;
}

}  // namespace xm3d

extern "C" const char* xm3d_last_error(void) { return xm3d::g_err; }
extern "C" int xm3d_version(void) { return 100; }

extern "C" int xm3d_device_info(int dev, int* n_cu, int* wave, char* arch64) {
    hipDeviceProp_t p;
    XM3D_HIP(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (wave) *wave = p.warpSize;
    if (arch64) {
        strncpy(arch64, p.gcnArchName, 63);
        arch64[63] = 0;
    }
    return XM3D_OK;
}

extern "C" int xm3d_check_flag(void) {
    int* f = xm3d::device_flag();
    if (!f) {
        xm3d::set_error("no device flag");
        return XM3D_EHIP;
    }
    int h = 0;
    XM3D_HIP(hipDeviceSynchronize());
    XM3D_HIP(hipMemcpy(&h, f, sizeof(int), hipMemcpyDeviceToHost));
    if (h != 0) {
        XM3D_HIP(hipMemset(f, 0, sizeof(int)));
        xm3d::set_error(h == XM3D_ERANGE ? "coordinate / index outside the valid range" : "hash table overflow");
    }
    return h;
}
