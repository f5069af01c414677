// layout.hip -- library state + layout plumbing: (G,C) genes-major <-> (C,ld) cells-major.
//
// The reference keeps every matrix as a numpy (genes, cells) array (analysis.py:59-61) and its
// kernels gather columns at stride C.  On the device a cell's gene vector is one contiguous row;
// this file holds the tiled transpose (+ dtype change) that converts at the API boundary.
#include "common.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>

namespace vcy {
thread_local char g_err[512] = "";

// Write-once caches behind one mutex (see common.h).  They hold facts about the machine and the environment, never
// anything derived from a caller's data, and an entry is immutable once inserted.
static std::mutex g_once_mutex;
static std::map<int, DevInfo> g_devinfo;
static std::map<std::pair<int, const void *>, size_t> g_dyn_lds;
static std::map<std::string, int> g_env;

int device_info(DevInfo *out)
{
    int dev = 0;
    VCY_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_once_mutex);
    auto it = g_devinfo.find(dev);
    if (it == g_devinfo.end()) {
        hipDeviceProp_t p;
        VCY_CHECK_HIP(hipGetDeviceProperties(&p, dev));
        DevInfo d{p.multiProcessorCount, (int)p.sharedMemPerBlock};   // 64 KiB default; opt-in up to 160 KiB on gfx950
        int opt = 0;
        if (hipDeviceGetAttribute(&opt, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && opt > d.lds_optin) d.lds_optin = opt;
        it = g_devinfo.emplace(dev, d).first;
    }
    *out = it->second;
    return VCY_OK;
}

int ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    int dev = 0;
    VCY_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_once_mutex);
    size_t &have = g_dyn_lds[std::make_pair(dev, kernel)];
    if (bytes > have) {
        VCY_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        have = bytes;
    }
    return VCY_OK;
}

int env_int(const char *name, int dflt)
{
    std::lock_guard<std::mutex> lock(g_once_mutex);
    auto it = g_env.find(name);
    if (it == g_env.end()) {
        const char *ev = getenv(name);
        it = g_env.emplace(name, ev ? atoi(ev) : dflt).first;
    }
    return it->second;
}

// 64 x 64 tile through LDS (+1 padding): reads coalesced along src columns, writes coalesced
// along dst columns.  dst padding columns [rows, ld_dst) are zero-filled by the row's last tile.
template <typename S, typename D>
__global__ __launch_bounds__(256) void k_transpose(const S *__restrict__ src, D *__restrict__ dst, int64_t rows, int64_t cols,
                                                    int64_t ld_src, int64_t ld_dst)
{
    __shared__ D tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    for (int j = ty; j < 64; j += 4) {
        const int64_t r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? (D)src[r * ld_src + c] : D(0);
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int64_t orow = c0 + j, ocol = r0 + tx;   // dst is (cols, ld_dst)
        if (orow < cols && ocol < ld_dst) dst[orow * ld_dst + ocol] = tile[tx][j];
    }
}

template <typename S, typename D>
static int launch_transpose(const void *src, void *dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, hipStream_t st)
{
    // grid.y covers ld_dst (not just rows) so that the padding columns get their zeros
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((ld_dst + 63) / 64));
    hipLaunchKernelGGL((k_transpose<S, D>), grid, dim3(256), 0, st, (const S *)src, (D *)dst, rows, cols, ld_src, ld_dst);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

// Shader-clock probe: one wave per workgroup (workgroup b lands on XCD b % 8) takes `nsamples` readings of the shader-clock counter
// (s_memtime: counts at the clock the CUs run at, which the chip sets to its power budget) against the constant 100 MHz counter
// (s_memrealtime), `interval` real-time ticks apart, sleeping in between.  Launched on a side stream it measures the clock WHILE another
// kernel runs: the wave takes one slot of one SIMD per XCD and issues a handful of scalar instructions per microsecond.
__global__ __launch_bounds__(64) void k_clock_probe(long long *__restrict__ samples, int nsamples, long long interval)
{
    long long *mine = samples + (long long)blockIdx.x * 2 * nsamples;
    const long long r0 = wall_clock64();
    for (int i = 0; i < nsamples; ++i) {
        const long long until = r0 + (long long)i * interval;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
        const long long c = clock64(), r = wall_clock64();
        if (threadIdx.x == 0) { mine[2 * i] = c; mine[2 * i + 1] = r; }
    }
}
}  // namespace vcy

using namespace vcy;

extern "C" const char *vcy_last_error(void) { return g_err; }
// 2: vcy_diffuse_step_factored gained `prepared` (round 3); vcy_gram added and vcy_knn_pool_csr's 4-element minimum stated (round 4)
// 3: vcy_clock_probe added (round 5)
extern "C" int vcy_abi_version(void) { return 4; }

extern "C" int vcy_clock_probe(int64_t *samples, int64_t nblocks, int64_t nsamples, int64_t interval_ticks, vcy_stream stream)
{
    VCY_REQUIRE(samples && nblocks > 0 && nblocks <= 64 && nsamples >= 2 && nsamples <= 4096 && interval_ticks > 0 &&
                nsamples * interval_ticks <= 300000000, "clock_probe: bad arguments (at most 3 s of 100 MHz ticks in all)");
    hipLaunchKernelGGL(k_clock_probe, dim3((unsigned)nblocks), dim3(64), 0, as_stream(stream), (long long *)samples, (int)nsamples, (long long)interval_ticks);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_device_info(int *cu_count, int *lds_bytes_per_block, int64_t *hbm_bytes)
{
    DevInfo d;
    int rc = device_info(&d);
    if (rc) return rc;
    if (cu_count) *cu_count = d.cus;
    if (lds_bytes_per_block) *lds_bytes_per_block = d.lds_optin;
    if (hbm_bytes) {
        size_t free_b = 0, total_b = 0;
        VCY_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
        *hbm_bytes = (int64_t)total_b;
    }
    return VCY_OK;
}

extern "C" int vcy_transpose(const void *src, void *dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst,
                             int src_dtype, int dst_dtype, vcy_stream stream)
{
    VCY_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "transpose: bad arguments");
    hipStream_t st = as_stream(stream);
    if (src_dtype == VCY_F32 && dst_dtype == VCY_F32) return launch_transpose<float, float>(src, dst, rows, cols, ld_src, ld_dst, st);
    if (src_dtype == VCY_F64 && dst_dtype == VCY_F32) return launch_transpose<double, float>(src, dst, rows, cols, ld_src, ld_dst, st);
    if (src_dtype == VCY_F32 && dst_dtype == VCY_F64) return launch_transpose<float, double>(src, dst, rows, cols, ld_src, ld_dst, st);
    if (src_dtype == VCY_F64 && dst_dtype == VCY_F64) return launch_transpose<double, double>(src, dst, rows, cols, ld_src, ld_dst, st);
    if (src_dtype == VCY_U8 && dst_dtype == VCY_U8) return launch_transpose<unsigned char, unsigned char>(src, dst, rows, cols, ld_src, ld_dst, st);
    if (src_dtype == VCY_U16 && dst_dtype == VCY_U16) return launch_transpose<unsigned short, unsigned short>(src, dst, rows, cols, ld_src, ld_dst, st);
    if (src_dtype == VCY_U16 && dst_dtype == VCY_F32) return launch_transpose<unsigned short, float>(src, dst, rows, cols, ld_src, ld_dst, st);
    if (src_dtype == VCY_U16 && dst_dtype == VCY_F64) return launch_transpose<unsigned short, double>(src, dst, rows, cols, ld_src, ld_dst, st);
    return fail(VCY_ERR_INVALID, "%s: bad dtype", "transpose");
}
