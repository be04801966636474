// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (dependent chains / independent chains) with 1..2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 pk_fp32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int CHAINS>
__global__ void k(float* out, int iters) {
    f2 v[CHAINS];
    float s[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { v[i] = f2{threadIdx.x * 1e-3f + i, 1.0f}; s[i] = threadIdx.x * 1e-3f + i; }
    const f2 a2 = {1.0001f, 0.9999f}, b2 = {1e-4f, 2e-4f};
    const float a = 1.0001f, b = 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == 0) s[c] = __builtin_fmaf(s[c], a, b);
                else v[c] = __builtin_elementwise_fma(v[c], a2, b2);
            }
        }
    }
    float r = 0;
    for (int i = 0; i < CHAINS; ++i) r += s[i] + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int CHAINS>
void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, CHAINS><<<256, threads>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, CHAINS><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 32 * CHAINS;           // instructions per wave
    printf("%-14s chains=%d waves/SIMD=%d : %.2f ns per instr per wave  (%.2f ns per instr per SIMD)\n", name, CHAINS, threads / 256,
           ms * 1e6 / n, ms * 1e6 / n / (threads / 256));
    hipFree(out);
}
int main() {
    run<0, 1>("v_fma_f32", 256); run<0, 4>("v_fma_f32", 256); run<0, 1>("v_fma_f32", 512); run<0, 4>("v_fma_f32", 512);
    run<1, 1>("v_pk_fma_f32", 256); run<1, 4>("v_pk_fma_f32", 256); run<1, 1>("v_pk_fma_f32", 512); run<1, 4>("v_pk_fma_f32", 512);
    run<0, 4>("v_fma_f32", 1024); run<1, 4>("v_pk_fma_f32", 1024);
    return 0;
}
