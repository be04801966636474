// Development harness for csrc/kernels_eigh.hpp: runs the direct eigensolver's kernels on covariance matrices from a
// file (int32 n, int32 count, count x n x n float32), checks every stage against double-precision host arithmetic and
// times the launches.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o eigh_probe.bin eigh_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define BBMPC_TU_CMA
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
    const int off = argc > 3 ? atoi(argv[3]) : 0;
    const int G = argc > 2 ? atoi(argv[2]) : cnt;
    if (off) mats.erase(mats.begin(), mats.begin() + (size_t)off * n * n);
    EighArgs q{};
    q.n = n; q.G = G;
    float* dC;
    CK(hipMalloc(&dC, mats.size() * 4)); CK(hipMemcpy(dC, mats.data(), mats.size() * 4, hipMemcpyHostToDevice));
    q.C = dC;
    const size_t LD = EIGH_LD, MAT = LD * LD;
    CK(hipMalloc(&q.d, G * LD * 4)); CK(hipMalloc(&q.e, G * LD * 4)); CK(hipMalloc(&q.tau, G * LD * 4)); CK(hipMalloc(&q.alpha, G * 4));
    CK(hipMalloc(&q.Vt, G * MAT * 4));
    CK(hipMalloc(&q.lam, G * LD * 4)); CK(hipMalloc(&q.Z, G * MAT * 4)); CK(hipMalloc(&q.Z2, G * MAT * 4)); CK(hipMalloc(&q.P, G * MAT * 4));
    CK(hipMalloc(&q.Tf, G * (LD / 32) * 1024 * 4)); CK(hipMalloc(&q.flags, G * 8 * 4)); CK(hipMemset(q.flags, 0, G * 8 * 4));
    CK(hipMalloc(&q.B, (size_t)G * n * n * 4)); CK(hipMalloc(&q.Dd, G * n * 4));
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
    // ---- stage 2
    const size_t lds2 = sizeof(EighSolveLds) > (32 * (EIGH_LD + 1) + 32 * 33) * 4 ? sizeof(EighSolveLds) : (32 * (EIGH_LD + 1) + 32 * 33) * 4;
    CK(hipFuncSetAttribute((const void*)k_eigh_tri_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_eigh_tri_solve, dim3(EIGH_SLOT_WGS + EIGH_TF_WGS, G), dim3(EIGH_SOLVE_THREADS), lds2, 0, q);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_eigh_tri_solve: %.1f us (%d instances)\n", ms * 1e3, G);
    }
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_eigh_tri_solve, dim3(EIGH_SLOT_WGS, G), dim3(EIGH_SOLVE_THREADS), lds2, 0, q);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_eigh_tri_solve, slot workgroups only: %.1f us\n", ms * 1e3);
    }
#ifdef EIGH_CLK
    {
        long long clk[16];
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(bbmpc::g_eigh_clk), sizeof(clk)));
        const char* nm[7] = {"setup+scans", "multisection", "qd recurrences", "argmin", "z recurrences", "normalise", "write Z"};
        for (int i = 0; i < 7; ++i) printf("  tri_solve %-16s %8lld cycles\n", nm[i], clk[i]);
    }
#endif
    CK(hipGetLastError());
    std::vector<float> lam(G * LD), Z(G * MAT), Tf(G * (LD / 32) * 1024);
    std::vector<unsigned> flags(G * 8);
    CK(hipMemcpy(lam.data(), q.lam, lam.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Z.data(), q.Z, Z.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Tf.data(), q.Tf, Tf.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(flags.data(), q.flags, flags.size() * 4, hipMemcpyDeviceToHost));
    for (int g = 0; g < (argc > 4 ? G : 0); ++g) {
        const float* dg = &d[g * LD]; const float* eg = &e[g * LD]; const float* Zg = &Z[g * MAT];
        double tn = 0; for (int i = 0; i < n; ++i) tn = fmax(tn, fmax(fabs(dg[i]), i < n - 1 ? fabs(eg[i]) : 0.0));
        // residual |T z - lam z| and orthogonality in double
        double maxres = 0, maxorth = 0;
        for (int j = 0; j < n; ++j) {
            double r2 = 0;
            for (int i = 0; i < n; ++i) {
                double tz = dg[i] * (double)Zg[i * LD + j];
                if (i > 0) tz += eg[i - 1] * (double)Zg[(i - 1) * LD + j];
                if (i < n - 1) tz += eg[i] * (double)Zg[(i + 1) * LD + j];
                const double rr = tz - (double)lam[g * LD + j] * Zg[i * LD + j];
                r2 += rr * rr;
            }
            maxres = fmax(maxres, sqrt(r2));
        }
        for (int a = 0; a < n; ++a) for (int b = a; b < n; ++b) {
            double s = 0; for (int i = 0; i < n; ++i) s += (double)Zg[i * LD + a] * Zg[i * LD + b];
            maxorth = fmax(maxorth, fabs(s - (a == b ? 1.0 : 0.0)));
        }
        float fr; memcpy(&fr, &flags[g * 8 + 2], 4);
        printf("instance %d: |T| %.3e  max|Tz - lam z| %.3e (device says %.3e)  max|Z^T Z - I| %.3e\n", g, tn, maxres, fr, maxorth);
        // T factors: I - V T V^T orthogonal?  check block 0 and the last block against the product of reflectors
        for (int b : {0, (n - 3) / 32}) {
            std::vector<double> Qb((size_t)n * n, 0.0), Qr((size_t)n * n, 0.0);
            for (int i = 0; i < n; ++i) Qr[(size_t)i * n + i] = 1.0;
            for (int k = 32 * b + 31; k >= 32 * b; --k) {      // Qr = H_{32b} ... H_{32b+31}
                const float* v = &Vt[g * MAT + k * LD]; const double tk = tau[g * LD + k];
                for (int c = 0; c < n; ++c) { double s = 0; for (int i = 0; i < n; ++i) s += (double)v[i] * Qr[(size_t)i * n + c]; s *= tk; for (int i = 0; i < n; ++i) Qr[(size_t)i * n + c] -= (double)v[i] * s; }
            }
            double maxd = 0;
            const float* T = &Tf[((size_t)g * (LD / 32) + b) * 1024];
            for (int i = 0; i < n; ++i) for (int c = 0; c < n; ++c) {
                double s = (i == c) ? 1.0 : 0.0;
                for (int a2 = 0; a2 < 32; ++a2) { double tv = 0; for (int c2 = 0; c2 < 32; ++c2) tv += (double)T[a2 * 32 + c2] * Vt[g * MAT + (32 * b + c2) * LD + c]; s -= (double)Vt[g * MAT + (32 * b + a2) * LD + i] * tv; }
                maxd = fmax(maxd, fabs(s - Qr[(size_t)i * n + c]));
            }
            printf("   T factor block %d: max|I - V T V^T - H..H| %.3e\n", b, maxd);
        }
    }
    // ---- stages 3, 4
    auto polish_and_back = [&]() {
        const dim3 gg(EIGH_LD / 64, EIGH_LD / 16, G);
        hipLaunchKernelGGL(k_eigh_gemm<0>, gg, dim3(256), 0, 0, q, (const float*)q.Z, (const float*)nullptr, q.P, 1, 0);
        hipLaunchKernelGGL(k_eigh_gemm<1>, gg, dim3(256), 0, 0, q, (const float*)q.Z, (const float*)q.P, q.Z2, 0, 0);
        hipLaunchKernelGGL(k_eigh_gemm<0>, gg, dim3(256), 0, 0, q, (const float*)q.Z2, (const float*)nullptr, q.P, 3, 1);
        hipLaunchKernelGGL(k_eigh_gemm<1>, gg, dim3(256), 0, 0, q, (const float*)q.Z2, (const float*)q.P, q.Z, 0, 1);
        hipLaunchKernelGGL(k_eigh_backtransform, dim3(EIGH_LD / 16, G), dim3(EIGH_BT_THREADS), 0, 0, q, (const float*)q.Z, (const float*)q.Z2);
    };
    for (int rep = 0; rep < 3; ++rep) {
        // (the polish overwrites Z: run the whole pipeline per repetition)
        CK(hipMemset(q.flags, 0, G * 8 * 4));
        float ms[4];
        hipEvent_t ev[5]; for (auto& x : ev) CK(hipEventCreate(&x));
        CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(k_eigh_tridiag, dim3(G), dim3(EIGH_TRI_THREADS), sizeof(EighTriLds), 0, q);
        CK(hipEventRecord(ev[1]));
        hipLaunchKernelGGL(k_eigh_tri_solve, dim3(EIGH_SLOT_WGS + EIGH_TF_WGS, G), dim3(EIGH_SOLVE_THREADS), lds2, 0, q);
        CK(hipEventRecord(ev[2]));
        polish_and_back();
        CK(hipEventRecord(ev[3])); CK(hipEventSynchronize(ev[3]));
        {   // the five launches of stages 3, 4 one by one (results unchanged: the polish of an orthogonal Z is the identity)
            const dim3 gg(EIGH_LD / 64, EIGH_LD / 16, G);
            hipEvent_t f[6]; for (auto& x : f) CK(hipEventCreate(&x));
            CK(hipEventRecord(f[0]));
            hipLaunchKernelGGL(k_eigh_gemm<0>, gg, dim3(256), 0, 0, q, (const float*)q.Z, (const float*)nullptr, q.P, 4, 0);
            CK(hipEventRecord(f[1]));
            hipLaunchKernelGGL(k_eigh_gemm<1>, gg, dim3(256), 0, 0, q, (const float*)q.Z, (const float*)q.P, q.Z2, 0, 0);
            CK(hipEventRecord(f[2]));
            hipLaunchKernelGGL(k_eigh_backtransform, dim3(EIGH_LD / 16, G), dim3(EIGH_BT_THREADS), 0, 0, q, (const float*)q.Z, (const float*)q.Z2);
            CK(hipEventRecord(f[3])); CK(hipEventSynchronize(f[3]));
            float a, b, c; CK(hipEventElapsedTime(&a, f[0], f[1])); CK(hipEventElapsedTime(&b, f[1], f[2])); CK(hipEventElapsedTime(&c, f[2], f[3]));
            if (rep == 2) printf("   single launches: gram %.1f  multiply %.1f  backtransform %.1f us\n", a * 1e3, b * 1e3, c * 1e3);
        }
        CK(hipGetLastError());
        for (int i = 0; i < 3; ++i) CK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
        CK(hipEventElapsedTime(&ms[3], ev[0], ev[3]));
        printf("pipeline: tridiag %.1f  tri_solve %.1f  polish+back %.1f  total %.1f us (%d instances)\n", ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, ms[3] * 1e3, G);
    }
    {
        std::vector<float> B((size_t)G * n * n), Dd((size_t)G * n);
        CK(hipMemcpy(B.data(), q.B, B.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Dd.data(), q.Dd, Dd.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(flags.data(), q.flags, flags.size() * 4, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; ++g) {
            const float* Bg = &B[(size_t)g * n * n]; const float* Cg = &mats[(size_t)g * n * n];
            double rec = 0, orth = 0, sortbad = 0;
            for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) {
                double s = 0, o = 0;
                for (int k2 = 0; k2 < n; ++k2) { const double dd = Dd[(size_t)g * n + k2]; s += (double)Bg[(size_t)i * n + k2] * dd * dd * Bg[(size_t)j * n + k2]; o += (double)Bg[(size_t)k2 * n + i] * Bg[(size_t)k2 * n + j]; }
                if (!(fabs(s) < 1e30) || !(fabs(o) < 1e30)) { rec = orth = 1e30; } rec = fmax(rec, fabs(s - Cg[(size_t)i * n + j])); orth = fmax(orth, fabs(o - (i == j ? 1.0 : 0.0)));
            }
            for (int k2 = 1; k2 < n; ++k2) if (Dd[(size_t)g * n + k2] > Dd[(size_t)g * n + k2 - 1]) sortbad += 1;
            float f1, f3, f2; memcpy(&f1, &flags[g * 8 + 1], 4); memcpy(&f3, &flags[g * 8 + 3], 4); memcpy(&f2, &flags[g * 8 + 2], 4);
            printf("final %d: max|B D^2 B^T - C| %.3e  max|B^T B - I| %.3e  unsorted %g  D[0] %.5f D[n-1] %.5f  |ZtZ-I| before %.2e after one round %.2e  resid %.2e\n",
                   g, rec, orth, sortbad, Dd[(size_t)g * n], Dd[(size_t)g * n + n - 1], f1, f3, f2);
        }
    }
    for (int g = 0; g < (argc > 4 ? G : 0); ++g) {
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
