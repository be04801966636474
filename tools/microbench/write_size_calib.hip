// Calibration of rocprofv3's WRITE_SIZE / FETCH_SIZE on gfx950 for the store patterns the rollout kernels use
// (MI355X_MICROARCH.md, HBM: "WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel writes (or reads) exactly BYTES bytes once:
//   w16   coalesced 16-byte stores (a wave = 1 KB contiguous)
//   w4    coalesced 4-byte stores  (a wave = 256 B contiguous)
//   w4seg 4-byte stores in 64-byte segments, 8 KB apart (16 particles of one row of the particle-minor sample matrix)
//   r16   coalesced 16-byte loads
// Build: hipcc --offload-arch=gfx950 -O3 write_size_calib.hip -o write_size_calib.bin
// Run:   rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d out -o w -- ./write_size_calib.bin   (and --pmc FETCH_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr size_t BYTES = 64ull << 20;
__global__ void w16(float4* p) { p[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); }
__global__ void w4(float* p) { p[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = 1.0f; }
__global__ void w4seg(float* p, int nst) {
    // thread t of the grid: segment s = t / 16 (one 64-byte piece of a row), lane-in-segment l = t % 16;
    // consecutive segments of a wave land in consecutive ROWS (nst floats apart), as a 16-particle tile's stores do
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t seg = t / 16, l = t % 16;
    const size_t rows = BYTES / 4 / nst;                 // rows of nst floats
    const size_t row = seg % rows, col0 = (seg / rows) * 16;
    p[row * nst + col0 + l] = 1.0f;
}
__global__ void r16(const float4* p, float* out) {
    const float4 v = p[(size_t)blockIdx.x * blockDim.x + threadIdx.x];
    if (v.x == 123.456f) out[0] = v.y;
}
int main() {
    float* d; float* o;
    (void)hipMalloc(&d, BYTES); (void)hipMalloc(&o, 64);
    (void)hipMemset(d, 0, BYTES);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        w16<<<BYTES / 16 / 256, 256>>>((float4*)d);
        w4<<<BYTES / 4 / 256, 256>>>(d);
        w4seg<<<BYTES / 4 / 256, 256>>>(d, 2048);
        r16<<<BYTES / 16 / 256, 256>>>((const float4*)d, o);
    }
    (void)hipDeviceSynchronize();
    printf("each kernel moved %zu bytes\n", BYTES);
    return 0;
}
