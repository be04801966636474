// Cost of a grid-wide barrier between 250 co-resident workgroups (one per CU, all 8 XCDs) with the data exchanged
// through cache-bypassing accesses, i.e. what a rollout kernel that stays resident across optimizer iterations would
// pay twice per iteration (rewards -> refit -> new mean).
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier.bin grid_barrier.hip && ./grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float coh_load(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coh_store(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// mode 0: barriers only.  mode 1: every workgroup publishes 4 floats, barrier, every workgroup reads all 4*G floats
// (the rewards), barrier.  mode 2: as 1 plus a 1000-float row per workgroup (the refit's dot product) + publishes one value
// that everybody reads after the second barrier (the new mean: 180 values).
__global__ __launch_bounds__(256) void k(unsigned* ctr, float* buf, float* rows, float* mean, int rounds, int mode, float* sink) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.0f;
    unsigned target = 0;
    for (int r = 0; r < rounds; ++r) {
        if (mode >= 1 && tid < 4) coh_store(buf + b * 4 + tid, (float)(r + b + tid));
        target += G;
        grid_barrier(ctr, target);
        if (mode >= 1) {
            for (int i = tid; i < 4 * G; i += 256) acc += coh_load(buf + i);
        }
        if (mode >= 2) {
            float s = 0.0f;
            if (b < 180) for (int i = tid; i < 1000; i += 256) s += coh_load(rows + b * 1000 + i);
            if (b < 180 && tid == 0) coh_store(mean + b, s + acc);
        }
        target += G;
        grid_barrier(ctr, target);
        if (mode >= 2 && tid < 180) acc += coh_load(mean + tid);
    }
    if (acc == -1.0f) sink[0] = acc;
}

int main() {
    unsigned* ctr; float *buf, *rows, *mean, *sink;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, 4096 * 4)); CK(hipMalloc(&rows, 180 * 1000 * 4)); CK(hipMalloc(&mean, 1024)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(rows, 0, 180 * 1000 * 4)); CK(hipMemset(buf, 0, 4096 * 4)); CK(hipMemset(mean, 0, 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {64, 250}) for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            int rounds = 200;
            CK(hipMemset(ctr, 0, 4));
            void* args[] = {&ctr, &buf, &rows, &mean, &rounds, &mode, &sink};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*)k, dim3(G), dim3(256), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("G=%3d mode %d: %.2f us per round (two barriers%s)\n", G, mode, ms * 1e3 / rounds,
                            mode == 0 ? "" : mode == 1 ? " + 4 floats per workgroup exchanged" : " + rewards exchange + 180 row sums + mean broadcast");
        }
    }
    return 0;
}
