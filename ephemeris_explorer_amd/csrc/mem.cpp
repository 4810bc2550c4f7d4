// mem.cpp -- device-memory plumbing shared by the handles of libephemeris_amd.
//
// 1. A cache of LARGE device blocks (>= kPoolMinBytes). A spacecraft batch owns knot slabs of several GB (56 B per knot and
//    craft), and a sweep loop creates and destroys identical batches: taking such blocks from the driver and handing them back
//    costs ~100 ms per batch (five 5.4 GB batches: 790 ms created with hipMalloc each time, 240-265 ms with the cache;
//    profiles/r04_sweep_evidence.md). Blocks of exactly the requested size are reused instead, up to a cap (EPH_POOL_MAX_MB,
//    default a quarter of the device); eph_release_cached_memory() returns them to the driver, and an allocation that fails
//    empties the cache and retries. Contents are NOT cleared: no user of a block this large reads what it did not write (knot
//    rows beyond nknots are unspecified by contract).
// 2. One process-wide staging buffer in pinned, device-mapped host memory that KERNELS read and write (no copy engine, no
//    pinning of short-lived host vectors): the deal's index arrays at batch creation, the reordered knot rows of a dealt batch.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "host.h"

namespace eph {

namespace {
constexpr size_t kPoolMinBytes = (size_t)64 << 20;
std::mutex g_pool_mu;
std::map<std::pair<int, size_t>, std::vector<void *>> g_pool;   // (device, bytes) -> free blocks
size_t g_pool_bytes = 0;

size_t pool_cap(int device) {
    static const long long forced = [] { const char *e = getenv("EPH_POOL_MAX_MB"); return e ? atoll(e) : -1LL; }();
    if (forced >= 0) return (size_t)forced << 20;
    size_t free_b = 0, total = 0;
    int current = device;
    (void)hipGetDevice(&current);
    if (device != current) (void)hipSetDevice(device);
    const hipError_t e = hipMemGetInfo(&free_b, &total);
    if (device != current) (void)hipSetDevice(current);
    return e == hipSuccess ? total / 4 : 0;
}
}  // namespace

int dev_alloc(size_t bytes, void **out) {
    *out = nullptr;
    int device = 0;
    if (bytes >= kPoolMinBytes && hipGetDevice(&device) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool.find({device, bytes});
        if (it != g_pool.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            g_pool_bytes -= bytes;
            return EPH_OK;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory && release_cached_memory() > 0) {      // the cache must never be why an allocation fails
        (void)hipGetLastError();
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) {
        *out = nullptr;
        set_last_error("hipMalloc", e);
        return e == hipErrorOutOfMemory ? EPH_ERR_OUT_OF_MEMORY : EPH_ERR_HIP;
    }
    return EPH_OK;
}

void dev_free(void *p, size_t bytes) {
    if (!p) return;
    int device = 0;
    if (bytes >= kPoolMinBytes && hipGetDevice(&device) == hipSuccess) {
        const int current = device;
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, p) == hipSuccess) device = attr.device;
        // what hipFree would have done before handing the block on: nothing on ITS device may still be using it
        if (device != current) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        if (device != current) (void)hipSetDevice(current);
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool_bytes + bytes <= pool_cap(device)) {
            g_pool[{device, bytes}].push_back(p);
            g_pool_bytes += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

size_t release_cached_memory() {
    std::map<std::pair<int, size_t>, std::vector<void *>> take;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        take.swap(g_pool);
        bytes = g_pool_bytes;
        g_pool_bytes = 0;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : take) {
        (void)hipSetDevice(kv.first.first);
        for (void *p : kv.second) (void)hipFree(p);
    }
    (void)hipSetDevice(cur);
    return bytes;
}

size_t cached_memory_bytes() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    return g_pool_bytes;
}

// ---- pinned, device-mapped staging ------------------------------------------------------------------------------------------
namespace {
std::mutex g_stage_mu;
void *g_stage_host = nullptr, *g_stage_dev = nullptr;
size_t g_stage_bytes = 0;
}  // namespace

PinnedStage::PinnedStage(size_t bytes) : lock_(g_stage_mu) {
    if (bytes > g_stage_bytes) {
        if (g_stage_host) (void)hipHostFree(g_stage_host);
        g_stage_host = g_stage_dev = nullptr;
        g_stage_bytes = 0;
        const size_t want = std::max(bytes + bytes / 4, (size_t)1 << 20);
        void *h = nullptr, *d = nullptr;
        hipError_t e = hipHostMalloc(&h, want, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
            if (h) (void)hipHostFree(h);
            set_last_error("hipHostMalloc (staging)", e);
            status_ = e == hipErrorOutOfMemory ? EPH_ERR_OUT_OF_MEMORY : EPH_ERR_HIP;
            return;
        }
        g_stage_host = h;
        g_stage_dev = d;
        g_stage_bytes = want;
    }
    host_ = g_stage_host;
    dev_ = g_stage_dev;
}

}  // namespace eph
