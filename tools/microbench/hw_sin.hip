// Accuracy of v_sin_f32 (input in revolutions) against double sin over theta in [-pi, pi], and of the two ways to feed it:
// from an angle carried in radians (one multiply by 1/2pi in front) and from an angle carried in revolutions.
//   hipcc --offload-arch=gfx950 -O3 -w -o hw_sin.bin hw_sin.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* th, float* s_rad, float* s_rev, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s_rad[i] = __builtin_amdgcn_sinf(th[i] * 0.15915494309189535f);     // radians -> revolutions -> v_sin_f32
    s_rev[i] = __builtin_amdgcn_sinf(th[i]);                            // th[i] taken as revolutions
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n), a(n), b(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(-M_PI + 2.0 * M_PI * (i + 0.5) / n);
    float *d, *da, *db;
    hipMalloc(&d, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, da, db, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e1r = 0;
    for (int i = 0; i < n; ++i) { const double r = sin((double)h[i]); const double e = fabs(a[i] - r); if (e > e1) e1 = e; }
    // revolutions: phi in [-0.5, 0.5]
    for (int i = 0; i < n; ++i) h[i] = (float)(-0.5 + (i + 0.5) / n);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, da, db, n);
    hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    double e2 = 0, e2small = 0;
    for (int i = 0; i < n; ++i) {
        const double r = sin(2.0 * M_PI * (double)h[i]); const double e = fabs(b[i] - r);
        if (e > e2) e2 = e;
        if (fabs(h[i]) < 0.01) { const double rel = e / fmax(fabs(r), 1e-30); if (rel > e2small) e2small = rel; }
    }
    printf("v_sin_f32 from radians (x * 1/2pi): max abs error %.3e\n", e1);
    printf("v_sin_f32 from revolutions        : max abs error %.3e   (max relative error for |phi| < 0.01: %.3e)\n", e2, e2small);
    printf("(fp32 ulp at 1: 1.19e-07)\n");
    return 0;
}
