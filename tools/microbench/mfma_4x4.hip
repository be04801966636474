// Microbenchmark: issue cost / dependent latency of v_mfma_f32_4x4x1_16b_f32 vs v_mfma_f32_16x16x4_f32
// (one wave per SIMD, one workgroup of 256 threads per CU).  Build: hipcc --offload-arch=gfx950 -O3 mfma_4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int CHAINS>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (MODE == 0) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u % CHAINS], 0, 0, 0);
            else acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CHAINS], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int CHAINS>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, CHAINS><<<256, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, CHAINS><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64;
    printf("%-28s chains=%d  %.2f ns/instr/wave  (%.1f counter ticks/instr)\n", name, CHAINS, ms * 1e6 / n, (double)c / n);
}
int main() {
    run<0, 1>("4x4x1_16b"); run<0, 2>("4x4x1_16b"); run<0, 4>("4x4x1_16b"); run<0, 8>("4x4x1_16b");
    run<1, 1>("16x16x4"); run<1, 2>("16x16x4"); run<1, 4>("16x16x4");
    return 0;
}
