// mem.cpp -- device-memory plumbing shared by the handles of libephemeris_amd.
//
// 1. A cache of LARGE device blocks (>= kPoolMinBytes). A spacecraft batch owns knot slabs of several GB (56 B per knot and
//    craft), and a sweep loop creates and destroys identical batches: taking such blocks from the driver and handing them back
//    costs ~100 ms per batch (five 5.4 GB batches: 790 ms created with hipMalloc each time, 240-265 ms with the cache;
//    profiles/r04_sweep_evidence.md). Blocks of exactly the requested size are reused instead.
//    The cache is accounted PER DEVICE and bounded per device: EPH_POOL_MAX_MB (0 switches it off), default an eighth of that
//    device's memory, read once per device. It never outlives the library's use of a device: when the LAST device allocation of
//    this library on a device is freed (every handle destroyed), that device's cached blocks go back to the driver, so a process
//    that shares the GPU with another allocator (PyTorch, RCCL, a second library) is not left holding GBs it does not use; while
//    handles are alive such a caller calls eph_release_cached_memory() itself. An allocation of ours that fails empties the cache
//    and retries. A reused block is cleared, synchronously (what a fresh allocation looks like: no caller is handed another batch's rows).
// 2. A pool of staging buffers in pinned, device-mapped host memory that KERNELS read and write (no copy engine, no pinning of
//    short-lived host vectors): seam 1's operands, state read-backs, the deal's index arrays at batch creation, the reordered knot
//    rows of a dealt batch. One buffer per concurrent user: threads driving distinct handles never wait for each other here.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "host.h"

namespace eph {

namespace {
constexpr size_t kPoolMinBytes = (size_t)64 << 20;
constexpr int kMaxDevices = 64;
std::mutex g_pool_mu;
std::map<std::pair<int, size_t>, std::vector<void *>> g_pool;   // (device, bytes) -> free blocks
size_t g_pool_bytes[kMaxDevices] = {};                          // cached bytes per device
long long g_live[kMaxDevices] = {};                             // this library's live device allocations per device (cached ones excluded)
size_t g_cap[kMaxDevices] = {};
bool g_cap_known[kMaxDevices] = {};

// the cap of `device`'s cache, computed once (the caller holds g_pool_mu and `device` is the current device)
size_t pool_cap_locked(int device) {
    if (g_cap_known[device]) return g_cap[device];
    static const long long forced = [] { const char *e = getenv("EPH_POOL_MAX_MB"); return e ? atoll(e) : -1LL; }();
    size_t cap = 0;
    if (forced >= 0) {
        cap = (size_t)forced << 20;
    } else {
        size_t free_b = 0, total = 0;
        if (hipMemGetInfo(&free_b, &total) == hipSuccess) cap = total / 8;
    }
    g_cap[device] = cap;
    g_cap_known[device] = true;
    return cap;
}
// blocks of `device` out of the cache (g_pool_mu held); the caller frees them outside the lock
void take_device_locked(int device, std::vector<void *> *out) {
    for (auto it = g_pool.begin(); it != g_pool.end();) {
        if (it->first.first == device) {
            out->insert(out->end(), it->second.begin(), it->second.end());
            it = g_pool.erase(it);
        } else {
            ++it;
        }
    }
    g_pool_bytes[device] = 0;
}
}  // namespace

int dev_alloc(size_t bytes, void **out) {
    *out = nullptr;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) device = -1;
    if (bytes >= kPoolMinBytes && device >= 0) {
        void *hit = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            auto it = g_pool.find({device, bytes});
            if (it != g_pool.end() && !it->second.empty()) {
                hit = it->second.back();
                it->second.pop_back();
                g_pool_bytes[device] -= bytes;
                g_live[device] += 1;
            }
        }
        if (hit) {
            // what a fresh allocation looks like -- COMPLETE before the block is handed out: the handles' streams are non-blocking,
            // so a clear merely queued on the default stream can land after the new owner's first writes (it did: a batch's knot 0
            // came back with a zeroed component in one run of test_body_order_of_the_acceleration_sum, round 5)
            if (hipMemsetAsync(hit, 0, bytes, nullptr) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess) {
                *out = hit;
                return EPH_OK;
            }
            (void)hipGetLastError();                     // the clear failed: the block is not handed out uncleared -- back to the driver,
            (void)hipFree(hit);                          // and a fresh allocation below
            std::lock_guard<std::mutex> lk(g_pool_mu);
            g_live[device] -= 1;
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
    if (device >= 0) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_live[device] += 1;
    }
    return EPH_OK;
}

void dev_free(void *p, size_t bytes) {
    if (!p) return;
    int current = 0, device = -1;
    if (hipGetDevice(&current) == hipSuccess) {
        device = current;
        // whose block it is: asked of the driver only where it can differ from the current device (seam 1 frees six blocks per call)
        static const int n_devices = [] { int n = 1; return hipGetDeviceCount(&n) == hipSuccess ? n : 1; }();
        if (n_devices > 1) {
            hipPointerAttribute_t attr{};
            if (hipPointerGetAttributes(&attr, p) == hipSuccess) device = attr.device;
            else (void)hipGetLastError();
        }
    }
    if (device < 0 || device >= kMaxDevices) { (void)hipFree(p); return; }
    bool cached = false;
    std::vector<void *> drop;
    if (bytes >= kPoolMinBytes) {
        // what hipFree would have done before handing the block on: nothing on ITS device may still be using it
        if (device != current) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            if (g_pool_bytes[device] + bytes <= pool_cap_locked(device)) {
                g_pool[{device, bytes}].push_back(p);
                g_pool_bytes[device] += bytes;
                cached = true;
            }
        }
        if (device != current) (void)hipSetDevice(current);
    }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_live[device] > 0) g_live[device] -= 1;
        if (g_live[device] == 0) take_device_locked(device, &drop);   // the library's last allocation on this device is gone
    }
    if (!cached) (void)hipFree(p);
    if (!drop.empty()) {
        if (device != current) (void)hipSetDevice(device);
        for (void *q : drop) (void)hipFree(q);          // (p itself is among them when it had just been cached)
        if (device != current) (void)hipSetDevice(current);
    }
}

size_t release_cached_memory() {
    std::map<std::pair<int, size_t>, std::vector<void *>> take;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        take.swap(g_pool);
        for (int d = 0; d < kMaxDevices; ++d) { bytes += g_pool_bytes[d]; g_pool_bytes[d] = 0; }
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
    size_t bytes = 0;
    for (int d = 0; d < kMaxDevices; ++d) bytes += g_pool_bytes[d];
    return bytes;
}

// ---- pinned, device-mapped staging ------------------------------------------------------------------------------------------
// A POOL of buffers, one per concurrent user (round 6): the reference runs its forward and backward propagators and one task per ship
// on a thread pool (ephemeris_explorer/src/prediction.rs:385-391), so two handles' read-backs must not wait for each other -- round 5's
// single buffer was held across a hipStreamSynchronize of the caller's stream, i.e. for as long as that handle's queued steps took.
// g_stage_mu covers the free list only; a PinnedStage owns its buffer until it is destroyed. Buffers are portable (every device may
// map them); at most kStageKeep idle ones (kStageKeepBytes in all) are kept.
namespace {
struct StageBuf { void *host; size_t bytes; };
constexpr size_t kStageKeep = 8;
constexpr size_t kStageKeepBytes = (size_t)384 << 20;      // idle pinned memory kept (one 256 MB knot-slab pass + the small ones)
std::mutex g_stage_mu;
std::vector<StageBuf> g_stage_free;
size_t g_stage_free_bytes = 0;
}  // namespace

PinnedStage::PinnedStage(size_t bytes) {
    StageBuf got{nullptr, 0};
    void *drop = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        size_t best = g_stage_free.size();
        for (size_t i = 0; i < g_stage_free.size(); ++i)               // smallest idle buffer that is large enough
            if (g_stage_free[i].bytes >= bytes && (best == g_stage_free.size() || g_stage_free[i].bytes < g_stage_free[best].bytes)) best = i;
        if (best < g_stage_free.size()) {
            got = g_stage_free[best];
            g_stage_free_bytes -= got.bytes;
            g_stage_free.erase(g_stage_free.begin() + (long)best);
        } else if (!g_stage_free.empty()) {                            // none fits: the largest idle one makes way for a bigger one
            size_t big = 0;
            for (size_t i = 1; i < g_stage_free.size(); ++i) if (g_stage_free[i].bytes > g_stage_free[big].bytes) big = i;
            drop = g_stage_free[big].host;
            g_stage_free_bytes -= g_stage_free[big].bytes;
            g_stage_free.erase(g_stage_free.begin() + (long)big);
        }
    }
    if (drop) (void)hipHostFree(drop);
    if (!got.host) {
        const size_t want = std::max(bytes + bytes / 4, (size_t)1 << 20);
        void *h = nullptr;
        hipError_t e = hipHostMalloc(&h, want, hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess) {
            set_last_error("hipHostMalloc (staging)", e);
            status_ = e == hipErrorOutOfMemory ? EPH_ERR_OUT_OF_MEMORY : EPH_ERR_HIP;
            return;
        }
        got = StageBuf{h, want};
    }
    void *d = nullptr;
    hipError_t e = hipHostGetDevicePointer(&d, got.host, 0);           // as the CURRENT device sees it
    if (e != hipSuccess) {
        (void)hipHostFree(got.host);
        set_last_error("hipHostGetDevicePointer (staging)", e);
        status_ = EPH_ERR_HIP;
        return;
    }
    host_ = got.host;
    dev_ = d;
    bytes_ = got.bytes;
}

PinnedStage::~PinnedStage() {
    if (!host_) return;
    void *drop = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        if (g_stage_free.size() < kStageKeep && g_stage_free_bytes + bytes_ <= kStageKeepBytes) {
            g_stage_free.push_back(StageBuf{host_, bytes_});
            g_stage_free_bytes += bytes_;
        } else {
            drop = host_;
        }
    }
    if (drop) (void)hipHostFree(drop);
}

}  // namespace eph
