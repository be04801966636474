// Development harness for csrc/kernels_eigh.hpp: runs the direct eigensolver's kernels on covariance matrices from a
// file (int32 n, int32 count, count x n x n float32), checks every stage against double-precision host arithmetic and
// times the launches.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o eigh_probe.bin eigh_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../blackbox_mpc_amd/csrc/kernels_eigh.hpp"
using namespace bbmpc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "_probe/eigh_mats.bin";
    FILE* f = fopen(path, "rb");
    if (!f) { printf("cannot open %s\n", path); return 1; }
    int hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 1;
    const int n = hdr[0], cnt = hdr[1];
    std::vector<float> mats((size_t)cnt * n * n);
    if (fread(mats.data(), 4, mats.size(), f) != mats.size()) return 1;
    fclose(f);
    printf("n = %d, %d matrices\n", n, cnt);
    const int G = cnt;
    EighArgs q{};
    q.n = n; q.G = G;
    float* dC;
    CK(hipMalloc(&dC, mats.size() * 4)); CK(hipMemcpy(dC, mats.data(), mats.size() * 4, hipMemcpyHostToDevice));
    q.C = dC;
    const size_t LD = EIGH_LD, MAT = LD * LD;
    CK(hipMalloc(&q.d, G * LD * 4)); CK(hipMalloc(&q.e, G * LD * 4)); CK(hipMalloc(&q.tau, G * LD * 4)); CK(hipMalloc(&q.alpha, G * 4));
    CK(hipMalloc(&q.Vt, G * MAT * 4));
    CK(hipMemset(q.Vt, 0xff, G * MAT * 4));
    CK(hipFuncSetAttribute((const void*)k_eigh_tridiag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(EighTriLds)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_eigh_tridiag, dim3(G), dim3(EIGH_TRI_THREADS), sizeof(EighTriLds), 0, q);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_eigh_tridiag: %.1f us (%d instances)\n", ms * 1e3, G);
    }
#ifdef EIGH_CLK
    {
        long long clk[16];
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(bbmpc::g_eigh_clk), sizeof(clk)));
        for (int i = 0; i < 10; ++i) printf("  row-class phase %d: %8.0f cycles per step (3 launches, 32 steps each)\n", i, (double)clk[i] / (3.0 * 32));
    }
#endif
    CK(hipGetLastError());
    std::vector<float> d(G * LD), e(G * LD), tau(G * LD), al(G), Vt(G * MAT);
    CK(hipMemcpy(d.data(), q.d, d.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(e.data(), q.e, e.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tau.data(), q.tau, tau.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(al.data(), q.alpha, G * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Vt.data(), q.Vt, Vt.size() * 4, hipMemcpyDeviceToHost));
    for (int g = 0; g < G; ++g) {
        // Q = H_0 ... H_{n-3} in double; R = Q^T E Q - T
        std::vector<double> Q((size_t)n * n, 0.0), E((size_t)n * n);
        for (int i = 0; i < n; ++i) Q[(size_t)i * n + i] = 1.0;
        for (int i = 0; i < n * n; ++i) E[i] = mats[(size_t)g * n * n + i];
        for (int i = 0; i < n; ++i) E[(size_t)i * n + i] -= al[g];
        for (int k = n - 3; k >= 0; --k) {       // Q <- H_k Q
            const float* v = &Vt[g * MAT + k * LD];
            const double t = tau[g * LD + k];
            if (t == 0.0) continue;
            for (int c = 0; c < n; ++c) {
                double s = 0; for (int i = k + 1; i < n; ++i) s += (double)v[i] * Q[(size_t)i * n + c];
                s *= t;
                for (int i = k + 1; i < n; ++i) Q[(size_t)i * n + c] -= (double)v[i] * s;
            }
        }
        std::vector<double> EQ((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) for (int k2 = 0; k2 < n; ++k2) { const double a = E[(size_t)i * n + k2]; for (int j = 0; j < n; ++j) EQ[(size_t)i * n + j] += a * Q[(size_t)k2 * n + j]; }
        double maxr = 0, maxt = 0, vbad = 0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
            double s = 0; for (int k2 = 0; k2 < n; ++k2) s += Q[(size_t)k2 * n + i] * EQ[(size_t)k2 * n + j];
            double tij = 0;
            if (i == j) tij = d[g * LD + i]; else if (j == i + 1) tij = e[g * LD + i]; else if (i == j + 1) tij = e[g * LD + j];
            maxr = fmax(maxr, fabs(s - tij)); maxt = fmax(maxt, fabs(tij));
        }
        for (int k = 0; k < n - 2; ++k) { const float* v = &Vt[g * MAT + k * LD]; for (int i = 0; i <= k; ++i) vbad = fmax(vbad, fabs(v[i])); vbad = fmax(vbad, fabs(v[k + 1] - 1.0)); }
        printf("instance %d: alpha %.6f |T| %.3e  max|Q^T E Q - T| %.3e  reflector layout err %.1e\n", g, al[g], maxt, maxr, vbad);
    }
    return 0;
}
