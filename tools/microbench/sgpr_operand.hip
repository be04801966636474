// Microbenchmark: does a VALU instruction pay for reading an SGPR operand?  Eight independent v_fma_f32 chains, the
// multiplier / addend in VGPRs, in SGPRs (one SGPR per instruction -- the constant-bus limit of gfx9), as an inline
// constant, and with the SGPR written by v_readfirstlane inside the loop; 1 and 2 waves per SIMD; shader clocks per
// instruction.     hipcc --offload-arch=gfx950 -O3 -o sgpr_operand.bin sgpr_operand.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(float* out, long long* clk, int iters, float av, float bv) {
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = threadIdx.x * 1e-3f + i;
    float a = av + (MODE == 0 ? threadIdx.x * 1e-9f : 0.0f), b = bv + (MODE == 0 ? threadIdx.x * 1e-9f : 0.0f);   // MODE 0: per-lane -> VGPRs
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {                                   // the SGPR freshly written by the VALU
            a = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s[0] * 1e-9f + av)));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[c]) : "v"(a), "v"(b));
                else if (MODE == 1 || MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[c]) : "s"(a), "v"(b));
                else if (MODE == 2) asm volatile("v_fma_f32 %0, %0, 0.5, %1" : "+v"(s[c]) : "v"(b));
                else if (MODE == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[c]) : "s"(a), "v"(b));
                else if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[c]) : "v"(a), "v"(b));
                else if (MODE == 6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(s[c]) : "s"(a));
                else if (MODE == 7) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(s[c]) : "v"(a));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 8);
    const int iters = 2000;
    k<MODE><<<256, threads>>>(out, clk, 10, 1.0001f, 1e-4f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, clk, iters, 1.0001f, 1e-4f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * 8;
    printf("%-34s waves/SIMD=%d : %.2f clocks per instr per wave, %.3f ns per instr per SIMD\n", name, threads / 256, (double)c / n, ms * 1e6 / n / (threads / 256));
    hipFree(out); hipFree(clk);
}
int main() {
    for (int t = 256; t <= 512; t += 256) {
        run<0>("v_fma_f32 v, v, v, v", t); run<1>("v_fma_f32 v, v, s, v", t); run<2>("v_fma_f32 v, v, 0.5, v", t);
        run<3>("v_fma_f32 v, v, s(readfirstlane), v", t); run<5>("v_fmac_f32 v, v, v", t); run<4>("v_fmac_f32 v, s, v", t);
        run<7>("v_mul_f32 v, v, v", t); run<6>("v_mul_f32 v, s, v", t);
    }
    return 0;
}
