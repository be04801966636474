// Can the host store straight into device memory (fine-grained allocation, large BAR)?  And what does a kernel pay for
// reading 80 bytes from pinned host memory instead?   build: hipcc --offload-arch=gfx950 -O3 -o host_write_dev.bin host_write_dev.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_read(const float* src, float* out, long long* clk) {
    const long long t0 = wall_clock64();
    float s = 0.0f;
    if (threadIdx.x < 20) s = src[threadIdx.x];
    s += __shfl_xor(s, 1, 64);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = s; clk[blockIdx.x] = t1 - t0; }
}
int main() {
    int attr = 0;
    float *pinned = nullptr, *pinned_dev = nullptr, *fine = nullptr, *out = nullptr; long long* clk = nullptr;
    CK(hipHostMalloc((void**)&pinned, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&pinned_dev, pinned, 0));
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, 4096));
    hipError_t e = hipExtMallocWithFlags((void**)&fine, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(fine-grained): %s\n", hipGetErrorString(e));
    hipPointerAttribute_t pa;
    if (e == hipSuccess && hipPointerGetAttributes(&pa, fine) == hipSuccess) printf("  type %d hostPointer %p devicePointer %p\n", (int)pa.type, pa.hostPointer, pa.devicePointer);
    (void)attr;
    for (int i = 0; i < 20; ++i) pinned[i] = (float)i;
    long long h[4];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, 0, pinned_dev, out, clk);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost));
        printf("kernel reads 20 floats from pinned host memory: %lld x 10 ns\n", h[0]);
    }
    if (e == hipSuccess) {
        CK(hipMemset(fine, 0, 4096));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, 0, fine, out, clk);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost));
            printf("kernel reads 20 floats from (fine-grained) device memory: %lld x 10 ns\n", h[0]);
        }
        printf("host store into it ...\n"); fflush(stdout);
        volatile float* hf = (volatile float*)fine;
        hf[0] = 42.0f;                    // faults without a host mapping
        float back = 0; CK(hipMemcpy(&back, fine, 4, hipMemcpyDeviceToHost));
        printf("  ok, device sees %.1f\n", back);
    }
    return 0;
}
