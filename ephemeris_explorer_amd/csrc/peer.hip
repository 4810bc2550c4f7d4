// peer.hip -- the exchange step of a target-partitioned run WITHOUT a collective library: every rank writes its
// slice straight into a mailbox in every peer's memory (hipIpc-mapped, xGMI peer access) and raises a flag there;
// the same launch then waits for the peers' flags and copies their slices out of its own mailbox into the gathered
// buffer. SURVEY 5 "fully-connected direct write": xGMI is point-to-point, so world-1 concurrent peer writes use
// world-1 links at once, and for the small systems of the path (N <= 4096: 128 KB of packed positions per step) the
// exchange costs one small kernel launch instead of a ring collective's start-up latency.
//
// Mailbox (one allocation per rank, fine-grained device memory, exported with hipIpcGetMemHandle):
//     [0, 4096)                  header: flags[src] (u64, the last epoch src has delivered), part counters
//     4096 + (parity*world + src) * slot_bytes     slice of rank `src` of the exchange with that epoch parity
// One exchange = one epoch (1, 2, ...; identical on every rank: exchanges are collective and stream-ordered).
// Why two parities suffice: nobody completes epoch e+1 before EVERY rank has pushed epoch e+1, which a rank does
// only after its own epoch-e launch has finished (stream order) -- so when a push of epoch e+2 lands in the slots
// of parity e, every reader of those slots is done.
// Memory model: remote data and flags are written with system-scope stores, the flag after a system-scope release
// fence; the waiting side polls with system-scope acquire loads and reads the mailbox with system-scope loads (cache
// bypass), so neither side depends on kernel-boundary coherence. A wait is bounded (EPH_PEER_TIMEOUT_MS, default
// 20 s): a lost peer turns into EPH_ERR_COMM at the next call, never into a hung device.
#include <cstdlib>
#include <cstring>
#include <string>

#include "host.h"

namespace eph {

constexpr size_t kPeerHeader = 4096;
constexpr int kPeerThreads = 256;

struct PeerArgs {
    char *peer_base[kPeerMaxWorld];   // mapped mailbox of every rank (own entry = local)
    char *buf;                        // the gathered buffer: world slices, mine current
    unsigned long long slice_bytes;   // bytes each rank contributes to the whole exchange (stride between slices in buf)
    unsigned long long off, len;      // the part of every slice this launch moves (len <= slot_bytes, multiples of 8)
    unsigned long long slot_bytes;
    unsigned long long epoch;
    unsigned long long timeout_ticks; // wall_clock64 ticks (100 MHz)
    unsigned *status;                 // host-mapped word: != 0 after a timed-out wait
    int rank, world, parts;           // parts = workgroups per peer
};

__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// grid = (world - 1) * parts workgroups: group g serves one peer, its `parts` workgroups split the slice
__global__ void __launch_bounds__(kPeerThreads) k_peer_exchange(const PeerArgs a) {
    const int g = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const int peer = g >= a.rank ? g + 1 : g;
    const unsigned long long words = a.len / 8;
    const unsigned long long per = (words + a.parts - 1) / a.parts;
    const unsigned long long w0 = min(words, per * part), w1 = min(words, w0 + per);
    const unsigned long long parity = a.epoch & 1ull;
    char *local = a.peer_base[a.rank];
    // ---- push: my slice -> the peer's mailbox
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(a.buf + a.rank * a.slice_bytes + a.off);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.peer_base[peer] + kPeerHeader +
                                                                         (parity * a.world + a.rank) * a.slot_bytes);
        for (unsigned long long w = w0 + threadIdx.x; w < w1; w += kPeerThreads) sys_store(dst + w, src[w]);
    }
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    __atomic_thread_fence(__ATOMIC_RELEASE);            // (HIP: system scope) my stores before the flag
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *counters = reinterpret_cast<unsigned *>(local + 2048);
        const unsigned done = atomicAdd(&counters[peer], 1u);
        if (done == (unsigned)a.parts - 1u) {           // last part of this peer's group: the slice is complete
            counters[peer] = 0u;                        // (next launch is stream-ordered after this one)
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.peer_base[peer]) + a.rank, a.epoch, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // ---- wait for the peer's slice in MY mailbox
        const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(local) + peer;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > a.timeout_ticks) {
                __hip_atomic_store(a.status, 1u + (unsigned)peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                timed_out = 1;
                break;
            }
        }
    }
    __syncthreads();
    // ---- copy out: the peer's slice -> its place in the gathered buffer (not after a timed-out wait: the slot holds an older
    // epoch's data; the gathered buffer keeps what it had and the host sees EPH_ERR_COMM at its next synchronisation point)
    if (!timed_out) {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(local + kPeerHeader +
                                                                                     (parity * a.world + peer) * a.slot_bytes);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(a.buf + peer * a.slice_bytes + a.off);
        for (unsigned long long w = w0 + threadIdx.x; w < w1; w += kPeerThreads) dst[w] = sys_load(src + w);
    }
}

PeerTransport::~PeerTransport() {
    (void)hipSetDevice(device_);
    for (int r = 0; r < world_; ++r)
        if (r != rank_ && base_[r]) (void)hipIpcCloseMemHandle(base_[r]);
    if (base_[rank_]) (void)hipFree(base_[rank_]);
    if (status_) (void)hipHostFree(status_);
    if (order_) (void)hipEventDestroy(order_);
}

// EPH_PEER_FORCE_FAIL=alloc|export|open (tests): makes the first attempt of that step fail, to walk the fallback branches.
static bool peer_force_fail(const char *what) {
    const char *e = getenv("EPH_PEER_FORCE_FAIL");
    return e && std::strcmp(e, what) == 0;
}
// The mailbox: fine-grained device memory when it can be allocated AND exported (coherent with peer writers without relying
// on kernel boundaries), plain hipMalloc memory otherwise -- every mailbox access in the kernel is system-scope either way.
// form: 0 = try fine-grained, fall back to coarse | 1 = fine-grained or fail | 2 = coarse.
int PeerTransport::create(int rank, int world, size_t slot_bytes, int form, std::shared_ptr<PeerTransport> *out) {
    if (world < 1 || world > kPeerMaxWorld || rank < 0 || rank >= world || form < 0 || form > 2) return EPH_ERR_BAD_ARGUMENT;
    slot_bytes = (slot_bytes + 255) / 256 * 256;
    if (slot_bytes == 0) slot_bytes = 256;
    std::shared_ptr<PeerTransport> t(new PeerTransport());
    t->rank_ = rank;
    t->world_ = world;
    t->slot_ = slot_bytes;
    EPH_HIP(hipGetDevice(&t->device_));
    const size_t bytes = kPeerHeader + 2 * (size_t)world * slot_bytes;
    std::string why;
    for (int attempt = (form == 2 ? 2 : 1); attempt <= 2; ++attempt) {      // 1 = fine-grained, 2 = coarse
        void *p = nullptr;
        hipError_t e = attempt == 1 ? (peer_force_fail("alloc") ? hipErrorOutOfMemory : hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained))
                                    : hipMalloc(&p, bytes);
        const char *step = attempt == 1 ? "hipExtMallocWithFlags (fine-grained peer mailbox)" : "hipMalloc (peer mailbox)";
        if (e == hipSuccess) {
            e = hipMemset(p, 0, kPeerHeader);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e == hipSuccess) {
                step = "hipIpcGetMemHandle (peer mailbox)";
                e = attempt == 1 && peer_force_fail("export") ? hipErrorInvalidValue
                                                               : hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(t->handle_), p);
            }
        }
        if (e == hipSuccess) {
            t->base_[rank] = p;
            t->form_ = attempt;
            break;
        }
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        why += std::string(why.empty() ? "" : "; ") + step + ": " + hipGetErrorString(e);
        if (attempt == 2 || form == 1) {
            set_last_error_text("eph_peer_create: " + why);
            return e == hipErrorOutOfMemory ? EPH_ERR_OUT_OF_MEMORY : EPH_ERR_COMM;
        }
    }
    t->fallback_reason_ = why;
    EPH_HIP(hipHostMalloc((void **)&t->status_, sizeof(unsigned), hipHostMallocMapped));
    *t->status_ = 0;
    EPH_HIP(hipHostGetDevicePointer((void **)&t->status_dev_, t->status_, 0));
    EPH_HIP(hipEventCreateWithFlags(&t->order_, hipEventDisableTiming));
    const char *env = getenv("EPH_PEER_TIMEOUT_MS");
    const double ms = env && *env ? atof(env) : 20000.0;
    t->timeout_ticks_ = (unsigned long long)(ms * 1e5);          // wall_clock64: 100 MHz
    *out = std::move(t);
    return EPH_OK;
}

int PeerTransport::connect(const void *handles) {
    if (!handles) return EPH_ERR_BAD_ARGUMENT;
    if (connected_) return EPH_ERR_BAD_ARGUMENT;
    EPH_HIP(hipSetDevice(device_));
    const char *h = static_cast<const char *>(handles);
    if (std::memcmp(h + (size_t)rank_ * kPeerHandleBytes, handle_, kPeerHandleBytes) != 0) {
        set_last_error_text("eph_peer_connect: the handle table's entry for this rank is not this rank's handle");
        return EPH_ERR_BAD_ARGUMENT;
    }
    for (int r = 0; r < world_; ++r) {
        if (r == rank_) continue;
        hipIpcMemHandle_t mh;
        std::memcpy(&mh, h + (size_t)r * kPeerHandleBytes, sizeof(mh));
        void *p = nullptr;
        hipError_t e = peer_force_fail("open") && form_ == 1 ? hipErrorInvalidValue : hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess && !(peer_force_fail("open") && form_ == 1)) {
            // lazy peer enabling did not do it: enable access to every other device explicitly and try once more
            (void)hipGetLastError();
            int ndev = 0;
            (void)hipGetDeviceCount(&ndev);
            for (int d = 0; d < ndev; ++d)
                if (d != device_) { (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }
            e = hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < r; ++q)                  // leave nothing half-mapped: the caller may re-create in the other form
                if (q != rank_ && base_[q]) { (void)hipIpcCloseMemHandle(base_[q]); base_[q] = nullptr; }
            set_last_error_text(std::string("eph_peer_connect: hipIpcOpenMemHandle of rank ") + std::to_string(r) + "'s " +
                                (form_ == 1 ? "fine-grained" : "coarse") + " mailbox: " + hipGetErrorString(e));
            return EPH_ERR_COMM;
        }
        base_[r] = p;
    }
    connected_ = true;
    return EPH_OK;
}

int PeerTransport::poll_error() const {
    if (status_ && *status_) {
        set_last_error_text("peer exchange: no data from rank " + std::to_string((int)*status_ - 1) + " within the time limit");
        return EPH_ERR_COMM;
    }
    return EPH_OK;
}

// buf = world * slice_bytes, this rank's slice current; in place, on stream s
int PeerTransport::all_gather_inplace(void *buf, size_t slice_bytes, hipStream_t s) {
    if (world_ == 1 || slice_bytes == 0) return EPH_OK;
    if (!connected_ || slice_bytes % 8 != 0) return EPH_ERR_BAD_ARGUMENT;
    int st = poll_error();
    if (st) return st;
    PeerArgs a{};
    for (int r = 0; r < world_; ++r) a.peer_base[r] = static_cast<char *>(base_[r]);
    a.buf = static_cast<char *>(buf);
    a.slice_bytes = slice_bytes;
    a.slot_bytes = slot_;
    a.timeout_ticks = timeout_ticks_;
    a.status = status_dev_;
    a.rank = rank_;
    a.world = world_;
    // The two-parity argument (top of file) needs the exchanges of this transport to run in issue order on the device.
    // Same stream: stream order. Another stream (a clone's): wait for the previous exchange's event first.
    if (have_last_ && last_stream_ != s) EPH_HIP(hipStreamWaitEvent(s, order_, 0));
    for (size_t off = 0; off < slice_bytes; off += slot_) {      // a slice larger than a slot goes in several epochs
        a.off = off;
        a.len = std::min(slot_, slice_bytes - off);
        a.epoch = ++epoch_;
        a.parts = (int)std::min<size_t>(16, (a.len + 32767) / 32768);
        hipLaunchKernelGGL(k_peer_exchange, dim3((unsigned)((world_ - 1) * a.parts)), dim3(kPeerThreads), 0, s, a);
        EPH_HIP(hipGetLastError());
    }
    EPH_HIP(hipEventRecord(order_, s));
    last_stream_ = s;
    have_last_ = true;
    return EPH_OK;
}

}  // namespace eph
