// Host-side pieces of libpats_amd.so: error plumbing, version, and the chunk planner.
#include "common.hpp"
#include <stdlib.h>
#include "chunk_plan.hpp"

#include <atomic>
#include <mutex>
#include <string>

namespace pats {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return e == hipErrorNoDevice ? PATS_ERR_NO_DEVICE : PATS_ERR_LAUNCH;
    }
    return PATS_OK;
}

// 16 bytes per thread; base pointers come from the caller's allocator (>= 16-byte aligned), the tail goes byte by byte
__global__ void __launch_bounds__(256) fill_bytes_kernel(unsigned char* p, unsigned word, size_t n) {
    const size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (o >= n) return;
    if (o + 16 <= n && ((reinterpret_cast<uintptr_t>(p) + o) & 15) == 0) {
        *reinterpret_cast<uint4*>(p + o) = make_uint4(word, word, word, word);
    } else {
        for (size_t k = o; k < n && k < o + 16; ++k) p[k] = (unsigned char)word;
    }
}

int fill_bytes(void* p, int value, size_t n, hipStream_t st) {
    if (n == 0) return PATS_OK;
    const unsigned b = (unsigned)value & 0xffu, word = b | (b << 8) | (b << 16) | (b << 24);
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, st,
                       static_cast<unsigned char*>(p), word, n);
    return check_launch("fill_bytes_kernel");
}

static int g_mode = PATS_SINKHORN_AUTO;
int sinkhorn_mode() { return g_mode; }

static unsigned long long* g_fallbacks[64];
static std::mutex g_fallbacks_mu;
unsigned long long* fallback_counter() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lock(g_fallbacks_mu);           // first use may race between host threads
    if (!g_fallbacks[dev]) {
        unsigned long long* p = nullptr;
        if (hipMalloc((void**)&p, sizeof(*p)) != hipSuccess || hipMemset(p, 0, sizeof(*p)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_fallbacks[dev] = p;
    }
    return g_fallbacks[dev];
}

// ---- the GNN layers' overflow protocol (gnn.hip) -----------------------------------------------------------------------------
// The fp16-split kernels of a layer raise a device-side flag when an activation leaves the fp16 range; by default (mode 0,
// "inline") every call queues its fp32 composition behind them, each kernel gated on that flag - correct without a host read,
// at the price of ~16 launches per layer that do nothing (340 per image pair, 21 ms of a 48-pair step).  Mode 1 ("deferred")
// queues none of them: the flag is ONE sticky word per device, the caller reads it where it synchronises anyway
// (pats_gnn_overflows) and, if it is raised, repeats the work in mode 0 - outputs of a call that raised it are not valid.
static std::atomic<int> g_gnn_redo_mode{0};
bool gnn_redo_deferred() { return g_gnn_redo_mode.load(std::memory_order_relaxed) == 1; }
static int* g_gnn_overflow[64];
int* gnn_overflow_flag() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lock(g_fallbacks_mu);
    if (!g_gnn_overflow[dev]) {
        int* p = nullptr;
        if (hipMalloc((void**)&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_gnn_overflow[dev] = p;
    }
    return g_gnn_overflow[dev];
}

}  // namespace pats

using namespace pats;

extern "C" int pats_set_gnn_redo_mode(int mode) {
    if (mode != 0 && mode != 1) return g_gnn_redo_mode.load();
    if (mode == 1 && !gnn_overflow_flag()) return g_gnn_redo_mode.load();       // no flag on this device: stay inline
    return g_gnn_redo_mode.exchange(mode);
}

extern "C" int pats_gnn_overflows(int64_t* raised, int reset) {
    PATS_REQUIRE(raised, "gnn_overflows: null pointer");
    int* p = gnn_overflow_flag();
    PATS_REQUIRE(p, "gnn_overflows: no flag on this device");
    int v = 0;
    if (hipDeviceSynchronize() != hipSuccess) return check_launch("gnn_overflows");
    if (hipMemcpy(&v, p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return check_launch("gnn_overflows");
    if (reset && (hipMemset(p, 0, sizeof(v)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) return check_launch("gnn_overflows");
    *raised = v != 0;
    return PATS_OK;
}

extern "C" const char* pats_version(void) { return "pats_amd 0.3.0 (gfx950)"; }

extern "C" int pats_abi_version(void) { return PATS_ABI_VERSION; }

// A HIP stream whose kernels may only run on the compute units set in `cu_mask` (bit i of word i / 32 = CU i).
// MI355X has 256 CUs; the HBM-bound stages of the path (crop and descriptor gathers) saturate the memory system from a
// fraction of them, the VALU-bound solvers want the rest: two masked streams let consecutive batches share the GPU
// SPATIALLY, where plain streams only time-slice (every kernel of the path fills all CUs' registers on its own).
extern "C" int pats_stream_create_cu_mask(const uint32_t* cu_mask, int words, pats_stream_t* stream) {
    PATS_REQUIRE(cu_mask && words > 0 && stream, "stream_create_cu_mask: bad argument");
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, cu_mask) != hipSuccess) return check_launch("hipExtStreamCreateWithCUMask");
    *stream = reinterpret_cast<pats_stream_t>(st);
    return PATS_OK;
}

extern "C" int pats_stream_destroy(pats_stream_t stream) {
    if (stream && hipStreamDestroy(as_stream(stream)) != hipSuccess) return check_launch("hipStreamDestroy");
    return PATS_OK;
}

extern "C" const char* pats_last_error(void) { return g_last_error.c_str(); }

extern "C" int pats_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int pats_set_sinkhorn_mode(int mode) {
    const int prev = g_mode;
    if (mode == PATS_SINKHORN_AUTO || mode == PATS_SINKHORN_LOG || mode == PATS_SINKHORN_KERNEL)
        g_mode = mode;
    return prev;
}

// fine level (145 x 145, pats_cost_ot_f32 variant 2): 0 = cost_mfma_kernel + sinkhorn_blk145_kernel (default: measured 1 % faster),
// 1 = the fused kernel (sinkhorn_blk.hip FUSED: no score matrix in HBM, two workgroups per CU).  PATS_FINE_FUSED=1 sets the default.
static int g_fine_fused = env_switch("PATS_FINE_FUSED") && atoi(env_switch("PATS_FINE_FUSED")) != 0;
namespace pats { bool cost_f32_only() { static const bool on = [] { const char* e = env_switch("PATS_COST_F32"); return e && atoi(e) != 0; }(); return on; } }
namespace pats { bool fine_fused() { return g_fine_fused != 0; } }
extern "C" int pats_set_fine_fused(int on) {
    const int prev = g_fine_fused;
    g_fine_fused = on != 0;
    return prev;
}

extern "C" int pats_sinkhorn_fallbacks(int64_t* count, int reset) {
    PATS_REQUIRE(count, "sinkhorn_fallbacks: null pointer");
    unsigned long long* p = fallback_counter();
    PATS_REQUIRE(p, "sinkhorn_fallbacks: no counter on this device");
    unsigned long long v = 0;
    // Kernels that bump the counter run on the caller's (possibly non-blocking) streams, which the
    // null-stream copy below does not wait for: drain the device first, so that the read sees every
    // launch issued so far and the reset cannot race with a kernel still in flight.
    if (hipDeviceSynchronize() != hipSuccess) return check_launch("sinkhorn_fallbacks");
    if (hipMemcpy(&v, p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return check_launch("sinkhorn_fallbacks");
    if (reset && (hipMemset(p, 0, sizeof(v)) != hipSuccess || hipDeviceSynchronize() != hipSuccess))
        return check_launch("sinkhorn_fallbacks");
    *count = (int64_t)v;
    return PATS_OK;
}

// split_patches, utils/utils.py:152-181.  Greedy row-aligned chunking of the matched coarse
// patches: a chunk closes at the first grid row whose cumulative match count exceeds
// max_once_used * (chunks so far + 1); chunks overlap by one grid row (the third_layer_set pair
// says how many leading / trailing patches of a chunk belong to the neighbouring chunk).
// Python's negative index sum_cycle[i*width - 1] at i == 0 (last element) is reproduced.
namespace pats {
// one thread per image pair: the plan is ~height comparisons on a cumsum that is already in L2
__global__ void split_patches_kernel(const int32_t* __restrict__ sc, int64_t pairs, int height, int width,
                                     int max_once_used, int64_t* __restrict__ second, int64_t* __restrict__ third,
                                     int32_t* __restrict__ cycle_num) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs) return;
    const int64_t row = 2 * (int64_t)(height + 1);
    for (int q = 0; q < row; ++q) { second[i * row + q] = 0; third[i * row + q] = 0; }
    cycle_num[i] = split_patches_plan(sc + i * (int64_t)height * width, height, width, max_once_used,
                                      second + i * row, third + i * row);
}
}  // namespace pats

extern "C" int pats_split_patches_device(const int32_t* sum_cycle, int64_t pairs, int height, int width,
                                         int max_once_used, int64_t* second, int64_t* third, int32_t* cycle_num,
                                         pats_stream_t stream) {
    PATS_REQUIRE(pairs >= 0 && height > 0 && width > 0 && max_once_used > 0, "split_patches_device: bad argument");
    if (pairs == 0) return PATS_OK;
    PATS_REQUIRE(sum_cycle && second && third && cycle_num, "split_patches_device: null pointer");
    hipLaunchKernelGGL(split_patches_kernel, dim3((unsigned)ceil_div(pairs, 64)), dim3(64), 0, as_stream(stream),
                       sum_cycle, pairs, height, width, max_once_used, second, third, cycle_num);
    return check_launch("split_patches_kernel");
}

extern "C" int pats_split_patches(const int32_t* sc, int height, int width, int max_once_used,
                                  int64_t* second, int64_t* third) {
    if (!sc || !second || !third || height <= 0 || width <= 0 || max_once_used <= 0) {
        set_error("split_patches: bad argument");
        return -PATS_ERR_INVALID;
    }
    return split_patches_plan(sc, height, width, max_once_used, second, third);
}
