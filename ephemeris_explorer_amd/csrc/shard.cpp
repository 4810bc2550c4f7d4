// shard.cpp -- the one exchange step of a target-partitioned N-body run: an in-place all-gather of equal slices.
//
// The massive-body path shards by TARGET body: rank r owns bodies [r*slice, (r+1)*slice), keeps their history
// and computes their ordered all-pairs sums against ALL sources, so every sum keeps the summation order of
// NewtonianGravity::eval (ephemeris/src/propagators/nbody.rs:22-38) and the sharded run is bit-identical to the
// single-device one. What a rank lacks after a step are the other ranks' new packed positions: one all-gather
// of slice*32 bytes per force evaluation. Two transports:
//   * RCCL over xGMI: ncclAllGather on the handle's own stream, no host synchronisation between steps. The
//     library is resolved at run time (dlopen): the RCCL beside the HIP runtime this library is bound to -- NOT whatever
//     RCCL the process already carries (PyTorch bundles one, bound to its own bundled runtime); EPH_RCCL_LIB overrides.
//   * direct peer writes (peer.hip, eph_peer_*): each rank's slice written straight into hipIpc-mapped mailboxes of
//     its peers, one small launch per exchange, no collective library -- the low-latency transport for small systems.
//   * a caller-supplied function (eph_exchange_fn), e.g. MPI or a host-staged gather; it is handed the stream
//     and must order itself after the work enqueued there.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "host.h"

namespace eph {
namespace {
// the slice of the RCCL API used here (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclAllGather,
// ncclCommDestroy, ncclGetErrorString); ncclUniqueId is 128 opaque bytes passed by value
struct UniqueId { char internal[128]; };
typedef void *Comm;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return GetUniqueId && CommInitRank && AllGather && CommDestroy; }
};
constexpr int kNcclInt8 = 0;   // ncclInt8 / ncclChar

// Which RCCL: the one that sits beside the HIP runtime THIS library is bound to. A process that imported PyTorch carries a second,
// bundled HIP runtime and an RCCL bound to it (torch/lib/librccl.so); device pointers and streams of one runtime mean nothing to
// the other, so an RCCL that merely happens to be loaded already is the wrong one unless it lives in our runtime's directory
// (round 5: ncclCommInitRank "no ROCm-capable device is detected" in a test process that had imported torch first). RTLD_DEEPBIND:
// the second RCCL resolves its own symbols before any global-scope copy. EPH_RCCL_LIB overrides the search.
std::string dir_of(const void *symbol) {
    Dl_info di{};
    if (!dladdr(symbol, &di) || !di.dli_fname) return std::string();
    const std::string f(di.dli_fname);
    const size_t k = f.rfind('/');
    return k == std::string::npos ? std::string() : f.substr(0, k);
}
Rccl *rccl() {
    static Rccl r = [] {
        Rccl x;
        const int flags = RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND;
        const char *env = getenv("EPH_RCCL_LIB");
        if (env && *env) x.lib = dlopen(env, flags);
        const std::string hipdir = dir_of((const void *)&hipGetDeviceCount);   // where the runtime this library calls lives
        if (!x.lib && !hipdir.empty())
            for (const char *name : {"/librccl.so.1", "/librccl.so"})
                if ((x.lib = dlopen((hipdir + name).c_str(), flags))) break;
        // no RCCL beside the runtime: one that is loaded already, provided it is bound to the same runtime directory
        const char *loaded[] = {"librccl.so", "librccl.so.1"};
        for (size_t i = 0; !x.lib && i < 2; ++i) {
            void *h = dlopen(loaded[i], RTLD_NOW | RTLD_NOLOAD);
            if (!h) continue;
            void *sym = dlsym(h, "ncclAllGather");
            if (sym && (hipdir.empty() || dir_of(sym) == hipdir)) x.lib = h;
            else dlclose(h);
        }
        const char *fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (size_t i = 0; !x.lib && i < 3; ++i) x.lib = dlopen(fresh[i], flags);
        if (!x.lib) return x;
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.lib, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.lib, "ncclCommInitRank");
        x.AllGather = (decltype(x.AllGather))dlsym(x.lib, "ncclAllGather");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.lib, "ncclCommDestroy");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.lib, "ncclGetErrorString");
        return x;
    }();
    return &r;
}
int rccl_fail(const char *what, int code) {
    Rccl *r = rccl();
    set_last_error_text(std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(code) : "RCCL error") +
                        " (" + std::to_string(code) + ")");
    return EPH_ERR_COMM;
}
}  // namespace

int rccl_unique_id(void *out128) {
    Rccl *r = rccl();
    if (!r->ok()) {
        set_last_error_text("RCCL not found (librccl.so; set EPH_RCCL_LIB)");
        return EPH_ERR_COMM;
    }
    UniqueId id;
    int rc = r->GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(out128, &id, sizeof(id));
    return EPH_OK;
}

Exchange::~Exchange() {
    if (comm_) rccl()->CommDestroy((Comm)comm_);
}

int Exchange::create(int rank, int world, const void *unique_id, eph_exchange_fn fn, void *ctx,
                     std::shared_ptr<Exchange> *out) {
    if (world < 1 || rank < 0 || rank >= world) return EPH_ERR_BAD_ARGUMENT;
    if (world > 1 && !unique_id && !fn) return EPH_ERR_BAD_ARGUMENT;
    std::shared_ptr<Exchange> e(new Exchange());
    e->rank_ = rank;
    e->world_ = world;
    e->fn_ = fn;
    e->ctx_ = ctx;
    if (unique_id && !fn) {
        Rccl *r = rccl();
        if (!r->ok()) {
            set_last_error_text("RCCL not found (librccl.so; set EPH_RCCL_LIB)");
            return EPH_ERR_COMM;
        }
        UniqueId id;
        std::memcpy(&id, unique_id, sizeof(id));
        Comm c = nullptr;
        int rc = r->CommInitRank(&c, world, id, rank);      // collective over the ranks, on the current device
        if (rc) return rccl_fail("ncclCommInitRank", rc);
        e->comm_ = c;
    }
    *out = std::move(e);
    return EPH_OK;
}

int Exchange::create_peer(std::shared_ptr<PeerTransport> t, std::shared_ptr<Exchange> *out) {
    if (!t) return EPH_ERR_BAD_ARGUMENT;
    std::shared_ptr<Exchange> e(new Exchange());
    e->rank_ = t->rank();
    e->world_ = t->world();
    e->peer_ = std::move(t);
    *out = std::move(e);
    return EPH_OK;
}

// buf = world * slice_bytes; this rank's slice (at rank * slice_bytes) is current, the rest is filled in
int Exchange::all_gather_inplace(void *buf, size_t slice_bytes, hipStream_t s) {
    if (slice_bytes == 0 || (world_ == 1 && !comm_)) return EPH_OK;
    gathers_ += 1;
    if (peer_) return peer_->all_gather_inplace(buf, slice_bytes, s);
    if (comm_) {                                             // (a one-rank communicator still goes through RCCL)
        int rc = rccl()->AllGather((const char *)buf + (size_t)rank_ * slice_bytes, buf, slice_bytes, kNcclInt8,
                                   (Comm)comm_, s);
        if (rc) return rccl_fail("ncclAllGather", rc);
        return EPH_OK;
    }
    int32_t rc = fn_(ctx_, buf, (uint64_t)slice_bytes, rank_, world_, (void *)s);
    if (rc) {
        set_last_error_text("exchange callback failed with " + std::to_string(rc));
        return EPH_ERR_COMM;
    }
    return EPH_OK;
}

}  // namespace eph
