// What a cluster of W workgroups on ONE XCD pays per optimizer iteration to act as one agent's workgroup in the
// persistent pendulum kernel: every member publishes its PI2 partials
// (beta_r, eta_r, S_r[H U] = 32 floats, or its local top-k for CEM: k (reward, index) pairs), an XCD-local barrier,
// every member reads all W partials and merges, a second barrier before the slots are reused.
//   hipcc --offload-arch=gfx950 -O3 -o cluster_barrier.bin cluster_barrier.hip && ./cluster_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float coh_load(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coh_store(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cluster_barrier(unsigned* ctr, unsigned target) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// grid 8 * W * C: cluster c of XCD x = workgroups {x + 8 (c W + m)}, m < W  (block b is dispatched to XCD b % 8)
__global__ __launch_bounds__(1024) void k(unsigned* ctrs, float* slots, int W, int words, int rounds, unsigned* xcc_seen, float* sink) {
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, c = slot / W, m = slot % W;
    const int cluster = c * 8 + x;
    unsigned* ctr = ctrs + cluster * 32;
    float* my = slots + ((size_t)cluster * W + m) * words;
    const float* all = slots + (size_t)cluster * W * words;
    if (threadIdx.x == 0) atomicOr(xcc_seen + cluster, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u));
    float acc = 0.0f;
    unsigned target = 0;
    for (int r = 0; r < rounds; ++r) {
        if ((int)threadIdx.x < words) coh_store(my + threadIdx.x, (float)(r + m) + acc * 1e-9f);
        target += W;
        cluster_barrier(ctr, target);
        for (int i = threadIdx.x; i < W * words; i += blockDim.x) acc += coh_load(all + i);
        target += W;
        cluster_barrier(ctr, target);
    }
    if (acc == -1.0f) sink[0] = acc;
}

int main() {
    unsigned *ctrs, *xcc; float *slots, *sink;
    CK(hipMalloc(&ctrs, 4096 * 4)); CK(hipMalloc(&xcc, 1024 * 4)); CK(hipMalloc(&slots, 1 << 22)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int words : {32, 100}) for (int W : {1, 2, 4}) for (int C : {1, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int rounds = 400;
            CK(hipMemset(ctrs, 0, 4096 * 4)); CK(hipMemset(xcc, 0, 1024 * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(8 * W * C), dim3(1024), 0, 0, ctrs, slots, W, words, rounds, xcc, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned seen[64]; CK(hipMemcpy(seen, xcc, sizeof(seen), hipMemcpyDeviceToHost));
            int one_xcd = 1; for (int i = 0; i < 8 * C; ++i) one_xcd &= (seen[i] & (seen[i] - 1)) == 0;
            if (rep) printf("W=%d workgroups per agent, %2d agents, %3d words per member: %.2f us per iteration (2 barriers + exchange)%s\n",
                            W, 8 * C, words, ms * 1e3 / rounds, one_xcd ? "" : "   [cluster NOT on one XCD]");
        }
    }
    return 0;
}
