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

// Per-handle communication state: communicator, stream, and two slots of (ready, done) events so that the gather of
// control step t overlaps step t+1 while the caller double-buffers its record / gathered arrays.
struct RecordComm {
    static constexpr int kSlots = 2;
    Rccl::Comm comm = nullptr;
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
        if (comm) (void)Rccl::get().CommDestroy(comm);
        comm = nullptr;
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
