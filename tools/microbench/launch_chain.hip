// How long does a chain of dependent tiny kernels take per kernel -- launched one by one on a stream, or as a captured
// hipGraph?  (The control steps of the learned-model and CMA-ES paths are 11-60 such launches)
// build: hipcc --offload-arch=gfx950 -O3 -o launch_chain.bin launch_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tiny(float* p, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + (float)i; }
int main() {
    float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int L = 64, REP = 200;
    for (int grid : {1, 256}) {
        // stream launches
        for (int w = 0; w < 3; ++w) { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_tiny, dim3(grid), dim3(64), 0, s, d, i); CK(hipStreamSynchronize(s)); }
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_tiny, dim3(grid), dim3(64), 0, s, d, i); CK(hipStreamSynchronize(s)); }
        double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REP * L);
        // the same chain as a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < L; ++i) hipLaunchKernelGGL(k_tiny, dim3(grid), dim3(64), 0, s, d, i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REP * L);
        printf("grid %3d x 64 threads, chain of %d dependent launches: %.2f us per kernel on a stream, %.2f us per kernel as a graph\n", grid, L, us_stream, us_graph);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
