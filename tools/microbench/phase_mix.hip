// Microbenchmark: one barrier-delimited "phase" of the quad rollout kernels -- a burst of LDS operand reads, NM
// v_mfma_f32_4x4x1_16b_f32 in three accumulator chains, NT tanh evaluations (the 6-instruction form of bb_tanhf) on the
// results, one LDS store, one barrier -- issued by ONE wave per SIMD (256 threads) versus split over TWO waves per SIMD
// (512 threads, each wave half the MFMAs and half the activations).  Answers: does a second wave per SIMD hide the
// per-phase VALU / LDS / barrier time that a lone wave exposes?   Build: hipcc --offload-arch=gfx950 -O3 phase_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tanh6(float x) {
    const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * fabsf(x));
    const float r = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
    return copysignf(r, x);
}

template <int NW, int NM, int NT, int NR>
__global__ __launch_bounds__(NW * 64, 1) void k(float* out, long long* cyc, int iters, const float* wsrc) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 16 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    float w[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) w[i] = wsrc[(i * 64 + lane) & 1023];
    for (int i = tid; i < 64 * 16 * 2; i += NW * 64) lds[i] = 0.001f * i;
    __syncthreads();
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const float* src = lds + (it & 1) * 1024 + (lane & 3) * 4 + (lane >> 4) * 13 * 16;
        f32x4 b[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) b[c] = *reinterpret_cast<const f32x4*>(src + (c % 13) * 16);
        f32x4 a0 = keep, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
#pragma unroll
        for (int i = 0; i + 2 < NM; i += 3) {
            const f32x4 bb = b[(i / 3 / 4) % NR];
            const float bv = bb[(i / 3) & 3];
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i], bv, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i + 1], bv, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i + 2], bv, a2, 0, 0, 0);
        }
        f32x4 o = {a0.x + a1.x + a2.x, a0.y + a1.y + a2.y, a0.z + a1.z + a2.z, a0.w + a1.w + a2.w};
#pragma unroll
        for (int r = 0; r < NT; r += 4) {
            o.x = tanh6(o.x + r); o.y = tanh6(o.y + r); o.z = tanh6(o.z + r); o.w = tanh6(o.w + r);
        }
        keep = o;
        *reinterpret_cast<f32x4*>(lds + ((it + 1) & 1) * 1024 + (tid & 63) * 4 + (tid >> 6) * 64) = o;   // next phase's input
        __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * NW * 64 + tid] = keep.x + keep.y + keep.z + keep.w;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NW, int NM, int NT, int NR>
void run(const char* name, const float* wsrc) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 3000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NW, NM, NT, NR><<<256, NW * 64>>>(out, cyc, 10, wsrc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NW, NM, NT, NR><<<256, NW * 64>>>(out, cyc, iters, wsrc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD=%d  MFMA/wave=%3d tanh/wave=%2d reads=%2d : %7.1f ns/phase  %6.0f cycles/phase  (MFMA floor per SIMD %d cycles)\n",
           name, NW / 4, NM, NT, NR, ms * 1e6 / iters, (double)c / iters, (NW / 4) * NM * 8);
}
int main() {
    float* wsrc; (void)hipMalloc(&wsrc, 4096); (void)hipMemset(wsrc, 0, 4096);
    run<4, 156, 12, 13>("layer-1 jobs, one wave per SIMD", wsrc);
    run<8, 78, 8, 13>("layer-1 jobs, split over two waves", wsrc);
    run<4, 156, 0, 13>("  same without activations", wsrc);
    run<8, 78, 0, 13>("  same without activations", wsrc);
    run<4, 30, 4, 1>("layer 0 (28 MFMAs + tanh), one wave", wsrc);
    run<8, 15, 4, 1>("layer 0, two waves", wsrc);
    run<4, 69, 16, 13>("last layer + epilogue-like VALU, one wave", wsrc);
    run<8, 36, 8, 13>("last layer, two waves", wsrc);
    run<4, 3, 0, 1>("barrier + LDS round trip only, one wave", wsrc);
    run<8, 3, 0, 1>("barrier + LDS round trip only, two waves", wsrc);
    return 0;
}
