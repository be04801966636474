// Microbenchmark: does v_mfma_f32_4x4x1_16b_f32 (and 16x16x4) issue faster from two or four waves of a SIMD than from one?
// Four independent accumulator chains per wave; shader clocks per MFMA per wave and ns per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -w -o mfma_4x4_occupancy.bin mfma_4x4_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (MODE == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 3], 0, 0, 0);
            else acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    k<MODE><<<256, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, cyc, 2000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 2000.0 * 64;
    printf("%-28s waves/SIMD=%d : %.2f clocks per MFMA per wave, %.2f ns per MFMA per SIMD\n", name, threads / 256, (double)c / n, ms * 1e6 / n / (threads / 256));
}
int main() {
    for (int t = 256; t <= 1024; t *= 2) { run<0>("v_mfma_f32_4x4x1_16b_f32", t); run<1>("v_mfma_f32_16x16x4_f32", t); }
    return 0;
}
