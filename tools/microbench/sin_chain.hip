// Microbenchmark: the pendulum recurrence's dependent chain  phi -> v_sin_f32 -> v_fmac -> v_rndne -> v_sub -> phi  alone,
// and with 18 independent filler instructions per step (what the model step has beside the chain), at 1 and 2 waves per
// SIMD; shader clocks per step.  Also v_sin_f32 back to back (dependent) for its latency.
//   hipcc --offload-arch=gfx950 -O3 -w -o sin_chain.bin sin_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* clk, int iters) {
    float phi = threadIdx.x * 1e-3f, sn = 0.1f, a = 0.01f + threadIdx.x * 1e-6f;
    float f[6];
    for (int i = 0; i < 6; ++i) f[i] = threadIdx.x * 1e-3f + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {                                   // v_sin only, dependent
                asm volatile("v_sin_f32 %0, %0" : "+v"(phi));
            } else {
                asm volatile("v_fmac_f32 %0, 0x3bc391d1, %1\n\tv_rndne_f32 %2, %0\n\tv_sub_f32 %0, %0, %2\n\tv_sin_f32 %1, %0" : "+v"(phi), "+v"(sn), "+v"(a));
                if (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 6; ++c) asm volatile("v_fmac_f32 %0, 0x3f000000, %0" : "+v"(f[c]));
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = phi + sn + a;
    for (int i = 0; i < 6; ++i) r += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 8);
    k<MODE><<<256, threads>>>(out, clk, 10);
    hipDeviceSynchronize();
    k<MODE><<<256, threads>>>(out, clk, 2000);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD=%d : %.1f clocks per step\n", name, threads / 256, (double)c / (2000.0 * 8));
    hipFree(out); hipFree(clk);
}
int main() {
    for (int t = 256; t <= 1024; t *= 2) {
        run<0>("v_sin_f32 dependent", t);
        run<1>("fmac, rndne, sub, sin (the chain)", t);
        run<2>("the chain + 18 independent v_fmac", t);
    }
    return 0;
}
