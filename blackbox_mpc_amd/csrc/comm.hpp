// The agent-sharded path's one exchange: every rank optimises its own agents with no
// collective on the data path, then the packed per-agent records [A_local, U+S+1] of a control step are all-gathered
// so that every rank (the one that steps the environment, the logger) has all actions.  The reference has no
// counterpart -- its agents are one batch dimension in one process (optimizers/optimizer_base.py:59-94).
//
// RCCL is entered directly from here rather than through torch.distributed: at ~50 us per control step the ~30 us of
// host time c10d spends per asynchronous collective would make the host the bottleneck (tools/gather_overhead.py).
// The collective runs on the handle's own communication stream; the launch stream is never blocked by it, and for
// single-kernel control steps nothing at all is added to it (the kernel publishes a sequence number itself).
//
// librccl is bound at run time (dlopen) so that libbbmpc.so itself has no link-time dependency on it: a process that
// already loaded an RCCL (PyTorch does) shares that copy.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

namespace bbmpc {

// The slice of rccl.h (NCCL-compatible C ABI, RCCL 2.x) this path needs.
struct Rccl {
    struct UniqueId { char internal[128]; };              // ncclUniqueId
    typedef void* Comm;                                   // ncclComm_t
    static constexpr int kFloat32 = 7;                    // ncclFloat32
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*CommUserRank)(Comm, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;

    static const Rccl& get() {
        static const Rccl api = load();
        return api;
    }
    void check(int rc, const char* what) const {
        if (rc != 0) throw std::runtime_error(std::string(what) + ": " + (GetErrorString ? GetErrorString(rc) : "RCCL error"));
    }

    // the same table served by the in-process communicator below (bbmpc_comm_init_local)
    static const Rccl& local();

private:
    static Rccl load() {
        void* lib = nullptr;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy the process already has, if any
            if (lib) break;
        }
        for (int i = 0; !lib && i < 3; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!lib) throw std::runtime_error("librccl.so.1 not found (needed for the multi-GPU record all-gather)");
        Rccl a;
        auto sym = [&](const char* s) {
            void* p = dlsym(lib, s);
            if (!p) throw std::runtime_error(std::string("librccl: missing symbol ") + s);
            return p;
        };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
        a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(sym("ncclCommUserRank"));
        return a;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// In-process communicator: several handles of ONE process on ONE device form a group whose all-gather runs on the callers'
// streams -- the collective contract of ncclAllGather (same count on every rank, calls in the same order on every rank,
// asynchronous on the stream it is given) without RCCL, which refuses two ranks on one GPU.  What it is for: the rank > 0 /
// nranks > 1 paths of the engine (record gather hand-offs, the per-iteration exchanges of a sharded population, the merges in
// rank order) on a one-GPU box, and several handles sharing a device in one process.
//
// Operation q of rank r, ring slot q % kDepth:
//   (a) stream: wait for every rank's `done` event of the slot's previous use, copy the rank's piece into the slot, record
//       `arrive[r][slot]`;
//   (b) HOST: rendezvous of all ranks (each rank is driven by its own host thread, as ranks are processes elsewhere);
//   (c) stream: wait for every other rank's `arrive` event, copy the slot to the destination, record `done[r][slot]`.
// The rendezvous is what makes it safe: every event is recorded (in host order) before anybody waits for it, so the waits
// can never sit in a hardware queue in FRONT of the record they wait for -- the runtime multiplexes streams onto a few
// hardware queues, and the first version (stream wait-value / write-value on signal memory, no host rendezvous) deadlocked
// exactly there: rank 0's "wait for arrive[1]" ahead of rank 1's "arrive[1] = 1" in one queue.
struct LocalGroup {
    static constexpr int kDepth = 4, kMaxRanks = 16;
    static constexpr size_t kMaxBytes = 1 << 20;          // per rank and operation
    int nranks = 0, device = 0, refs = 0;
    uint32_t joined = 0;                                  // bit r: rank r has been taken
    char* stage = nullptr;                                // [kDepth][nranks * kMaxBytes]
    hipEvent_t arrive[kMaxRanks][kDepth] = {};
    hipEvent_t done[kMaxRanks][kDepth] = {};
    std::mutex mu;                                        // host rendezvous
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    bool broken = false;                                  // a rank gave up waiting: every later operation fails instead of hanging
};
struct LocalRank {
    LocalGroup* grp = nullptr;
    int rank = 0;
    uint32_t op = 0;
    uint64_t key = 0;
};
struct LocalRegistry {
    std::mutex mu;
    std::map<uint64_t, LocalGroup*> groups;
    static LocalRegistry& get() { static LocalRegistry r; return r; }
};

inline void local_group_free(LocalGroup* g) {
    (void)hipFree(g->stage);
    for (int r = 0; r < g->nranks; ++r)
        for (int d = 0; d < LocalGroup::kDepth; ++d) {
            if (g->arrive[r][d]) (void)hipEventDestroy(g->arrive[r][d]);
            if (g->done[r][d]) (void)hipEventDestroy(g->done[r][d]);
        }
    delete g;
}

inline int local_comm_join(uint64_t key, int nranks, int rank, int device, Rccl::Comm* out) {
    if (nranks < 1 || nranks > LocalGroup::kMaxRanks || rank < 0 || rank >= nranks) return 4;     // ncclInvalidArgument
    LocalRegistry& reg = LocalRegistry::get();
    std::lock_guard<std::mutex> lk(reg.mu);
    LocalGroup*& g = reg.groups[key];
    if (!g) {
        g = new LocalGroup();
        g->nranks = nranks; g->device = device;
        bool ok = hipMalloc((void**)&g->stage, (size_t)LocalGroup::kDepth * nranks * LocalGroup::kMaxBytes) == hipSuccess;
        for (int r = 0; ok && r < nranks; ++r)
            for (int d = 0; ok && d < LocalGroup::kDepth; ++d)
                ok = hipEventCreateWithFlags(&g->arrive[r][d], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&g->done[r][d], hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); LocalGroup* dead = g; reg.groups.erase(key); local_group_free(dead); return 1; }   // ncclUnhandledCudaError
    }
    bool group_broken;
    { std::lock_guard<std::mutex> gl(g->mu); group_broken = g->broken; }
    if (g->nranks != nranks || g->device != device || (g->joined >> rank) & 1u || group_broken) {   // (a second handle claiming a taken rank; a group a rank has already left)
        if (g->refs == 0) { LocalGroup* dead = g; reg.groups.erase(key); local_group_free(dead); }
        return 4;
    }
    g->joined |= 1u << rank;
    LocalRank* me = new LocalRank();
    me->grp = g; me->rank = rank; me->key = key;
    ++g->refs;
    *out = me;
    return 0;
}

namespace local_api {
// all ranks of the group, or nobody: false after `seconds` without the others (a rank that never calls is a caller's bug --
// e.g. one host thread driving two ranks -- and must show up as an error, not as a hang)
inline bool rendezvous(LocalGroup* g, int seconds) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return false;
    const uint64_t gen = g->generation;
    if (++g->waiting == g->nranks) {
        g->waiting = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    if (!g->cv.wait_for(lk, std::chrono::seconds(seconds), [&] { return g->generation != gen || g->broken; })) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return g->generation != gen;       // the generation advanced: this collective is complete even if a peer has left since
}
// a rank that cannot take part marks the group: its peers fail at once instead of waiting out the rendezvous
inline int give_up(LocalGroup* g, int rc) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;
    g->cv.notify_all();
    return rc;
}
inline int rendezvous_seconds() {
    static const int s = [] {
        const char* e = getenv("BBMPC_LOCAL_COMM_TIMEOUT_S");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 60;
    }();
    return s;
}
inline int all_gather(const void* send, void* recv, size_t count, int dtype, Rccl::Comm comm, hipStream_t st) {
    LocalRank* me = static_cast<LocalRank*>(comm);
    LocalGroup* g = me->grp;
    const size_t bytes = count * (dtype == Rccl::kFloat32 ? 4 : 1);
    if (bytes == 0) return give_up(g, 4);
    if (bytes > LocalGroup::kMaxBytes) return give_up(g, 6);
    const uint32_t q = me->op++;
    const int d = (int)(q % LocalGroup::kDepth);
    char* slot = g->stage + (size_t)d * g->nranks * LocalGroup::kMaxBytes;
    hipError_t e = hipSuccess;
    if (q >= (uint32_t)LocalGroup::kDepth)                              // recorded kDepth operations ago, i.e. before the last rendezvous
        for (int p = 0; p < g->nranks && e == hipSuccess; ++p) e = hipStreamWaitEvent(st, g->done[p][d], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(slot + (size_t)me->rank * bytes, send, bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(g->arrive[me->rank][d], st);
    if (e != hipSuccess) return give_up(g, 1);
    if (!rendezvous(g, rendezvous_seconds())) return 5;
    for (int p = 0; p < g->nranks && e == hipSuccess; ++p)
        if (p != me->rank) e = hipStreamWaitEvent(st, g->arrive[p][d], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(recv, slot, (size_t)g->nranks * bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(g->done[me->rank][d], st);
    return e == hipSuccess ? 0 : 1;
}
inline int comm_destroy(Rccl::Comm comm) {
    LocalRank* me = static_cast<LocalRank*>(comm);
    LocalRegistry& reg = LocalRegistry::get();
    std::lock_guard<std::mutex> lk(reg.mu);
    LocalGroup* g = me->grp;
    {   // a rank that leaves takes the group with it: the others' next collective fails instead of waiting for it
        std::lock_guard<std::mutex> gl(g->mu);
        g->broken = true;
        g->cv.notify_all();
    }
    if (--g->refs == 0) {
        reg.groups.erase(me->key);
        local_group_free(g);
    }
    delete me;
    return 0;
}
inline int comm_count(Rccl::Comm comm, int* n) { *n = static_cast<LocalRank*>(comm)->grp->nranks; return 0; }
inline int comm_user_rank(Rccl::Comm comm, int* r) { *r = static_cast<LocalRank*>(comm)->rank; return 0; }
inline const char* error_string(int rc) {
    return rc == 4 ? "in-process communicator: invalid argument (count, rank, group shape, or a group that a rank has already left)"
         : rc == 5 ? "in-process communicator: the other ranks did not arrive within the rendezvous time (60 s, BBMPC_LOCAL_COMM_TIMEOUT_S; "
                     "every rank needs its own host thread), or one of them has failed or left"
         : rc == 6 ? "in-process communicator: more than 1 MiB per rank in one all-gather (the staging ring's slot size; RCCL has no such limit)"
                   : "in-process communicator: HIP error";
}
}  // namespace local_api

inline const Rccl& Rccl::local() {
    static const Rccl api = [] {
        Rccl a;
        a.AllGather = local_api::all_gather;
        a.CommDestroy = local_api::comm_destroy;
        a.CommCount = local_api::comm_count;
        a.CommUserRank = local_api::comm_user_rank;
        a.GetErrorString = local_api::error_string;
        return a;
    }();
    return api;
}

// Per-handle communication state: communicator, stream, and two slots of (ready, done) events so that the gather of
// control step t overlaps step t+1 while the caller double-buffers its record / gathered arrays.
struct RecordComm {
    static constexpr int kSlots = 2;
    Rccl::Comm comm = nullptr;
    const Rccl* api = nullptr;         // who serves `comm`: librccl (bbmpc_comm_init) or the in-process communicator (bbmpc_comm_init_local)
    int nranks = 0, rank = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ready[kSlots] = {nullptr, nullptr};
    hipEvent_t done[kSlots] = {nullptr, nullptr};
    bool pending[kSlots] = {false, false};
    // "records ready" hand-off without a cross-stream event: the persistent kernel's last workgroup publishes a
    // sequence number in signal memory and the communication stream waits for the value (hipStreamWaitValue32).
    // An event the other stream waits on costs the launch stream 5-9 us per control step, see gather_records().
    uint32_t* flag = nullptr;          // signal memory: sequence number of the newest complete records
    uint32_t* count = nullptr;         // device memory: arrival counter of the publishing kernel's workgroups
    uint32_t* done_flag[kSlots] = {nullptr, nullptr};   // signal memory (host readable): sequence number whose gather finished
    uint32_t done_seq[kSlots] = {0, 0};
    uint32_t seq = 0;
    int sync_mode = 0;                 // 0 events only, 1 flags in signal memory (see gather_records in bbmpc.hip)

    void destroy() {
        if (stream) (void)hipStreamSynchronize(stream);
        if (comm && api) (void)api->CommDestroy(comm);
        comm = nullptr;
        api = nullptr;
        for (int s = 0; s < kSlots; ++s) {
            if (ready[s]) (void)hipEventDestroy(ready[s]);
            if (done[s]) (void)hipEventDestroy(done[s]);
            ready[s] = done[s] = nullptr;
            pending[s] = false;
        }
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        if (flag) (void)hipFree(flag);
        if (count) (void)hipFree(count);
        flag = count = nullptr;
        for (int s = 0; s < kSlots; ++s) {
            if (done_flag[s]) (void)hipFree(done_flag[s]);
            done_flag[s] = nullptr;
            done_seq[s] = 0;
        }
        seq = 0;
    }
};

}  // namespace bbmpc
