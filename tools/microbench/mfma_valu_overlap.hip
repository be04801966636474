// Microbenchmark: does matrix-core work overlap with VALU work issued by the SAME wave / other waves of the SIMD?
// fp32-input MFMA (v_mfma_f32_16x16x4_f32) vs bf16-input MFMA (v_mfma_f32_16x16x16_bf16), each alone, VALU alone
// (independent v_fma_f32 chains), and both interleaved.  One workgroup of 256 threads (1 wave per SIMD) or 512 (2).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // bit0: fp32 mfma, bit1: bf16 mfma, bit2: valu
__global__ void k(float* out, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    bf16x4 ab = {(short)threadIdx.x, 1, 2, 3}, bb = {3, 2, 1, (short)threadIdx.x};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE & 1) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
            if (MODE & 2) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, acc[u & 3], 0, 0, 0);
            if (MODE & 4) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = __builtin_fmaf(v[c], 1.0001f, 1e-4f);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD=%d : %.1f ns per (1 MFMA%s) group per wave\n", name, threads / 256, ms * 1e6 / (iters * 16.0),
           (MODE & 4) ? " + 8 VALU" : "");
    hipFree(out);
}
int main() {
    run<1>("fp32 MFMA 16x16x4 alone", 256); run<2>("bf16 MFMA 16x16x16 alone", 256); run<4>("8 x v_fma_f32 alone", 256);
    run<5>("fp32 MFMA + 8 VALU", 256); run<6>("bf16 MFMA + 8 VALU", 256);
    run<1>("fp32 MFMA 16x16x4 alone", 512); run<2>("bf16 MFMA 16x16x16 alone", 512); run<4>("8 x v_fma_f32 alone", 512);
    run<5>("fp32 MFMA + 8 VALU", 512); run<6>("bf16 MFMA + 8 VALU", 512);
    return 0;
}
