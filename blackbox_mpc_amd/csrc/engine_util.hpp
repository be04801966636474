// Host-side helpers shared by the engine's translation units (bbmpc.hip, bbmpc_cma.hip, bbmpc_mlp.hip).
#pragma once
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "engine.hpp"

namespace bbmpc {

inline void upload(DevBuf<float>& b, const std::vector<float>& v) {
    b.alloc(v.size());
    HIP_CHECK(hipMemcpy(b.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (device, kernel)
// -- a process may hold handles on several devices (bbmpc_config.device).
inline void ensure_max_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.find({dev, fn}) == done.end()) {
        HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done.insert({dev, fn});
    }
}

// kernels whose dynamic LDS grows with the population (an agent's rewards, Nst floats): past the 64 KB default they need
// the attribute raised
// static_bytes: what the kernel declares as static __shared__ on top (the two together must fit the CU's 160 KB)
inline void want_lds(const void* fn, size_t bytes, size_t static_bytes = 0) {
    if (bytes > 64 * 1024) {
        REQUIRE(bytes + static_bytes <= 159 * 1024, BBMPC_E_UNSUPPORTED, "population too large for one CU's LDS");
        ensure_max_lds(fn, (int)(159 * 1024 - static_bytes));
    }
}


// Launch the last kernel of a control step.  When the caller asked for a completion event (the record all-gather
// waits on it from its own stream) the event rides on the kernel's own dispatch packet: a separate
// hipEventRecord costs the launch stream ~5 us per control step (tools/gather_overhead.py).
template <class F, class... Args>
inline void launch_with_tail(Engine& e, F fn, dim3 grid, dim3 block, size_t lds, const Args&... args) {
    if (e.tail_event) {
        hipExtLaunchKernelGGL(fn, grid, block, lds, e.stream, nullptr, e.tail_event, 0, args...);
        e.tail_attached = true;
    } else {
        hipLaunchKernelGGL(fn, grid, block, lds, e.stream, args...);
    }
}

// every translation unit has its own copy of the truncated-normal quantile table (rng.hpp, `static __device__`): the
// core uploads all of them when a handle is created on a device for the first time
void bbmpc_tu_cma_upload_tnq(const float2* table);
void bbmpc_tu_mlp_upload_tnq(const float2* table);
void bbmpc_tu_fused_upload_tnq(const float2* table);
void note_resident_handle();          // a handle published a resident kernel's mailbox (stop_foreign_residents counts them)

}  // namespace bbmpc
