// Direct symmetric eigensolver for SMALL CMA-ES instances (n <= 32: config 2's n = H*U = 30) inside ONE 1024-thread
// workgroup -- the part of `u, B, _ = tf.linalg.svd(C); D = diag(sqrt(u))` (cma_es.py:195-198) that the one-sided Jacobi of
// kernels_cma.hpp (cma_svd_small_body) needs 6-8 sweeps x 29 barrier-separated rounds = 65-80 us for.  The same four
// stages as the n = 300 solver of kernels_eigh.hpp, sized for a matrix that fits one wave's registers:
//   1. tridiagonalisation   wave 0 alone, lane j holds column j of E = C - (tr C / n) I in 32 registers; a Householder
//                           step is a matrix-vector product and a rank-2 update on those registers with the reflector's
//                           elements broadcast by v_readlane (static lane numbers: the step loop is unrolled), two wave
//                           reductions, no barrier, no LDS traffic except the stored reflector
//   2. eigen-decomposition of T   one half-wave (32 lanes) per eigenvalue slot, 16 waves = 32 slots: six passes of 33-fold
//                           multisection on Sturm counts, then the twisted factorisation's eigenvector (kernels_eigh.hpp
//                           stage 2, the same splitting rule |e_k| <= 4 eps max(|alpha|, |T|) and block bookkeeping)
//   3. Newton-Schulz polish  Z <- Z (1.5 I - 0.5 Z^T Z), two rounds, one thread per matrix element
//   4. back-transformation  B = H_0 .. H_{n-3} Z, one half-wave per column (lane = row), the reflectors from LDS
// then ranks (descending, ties: lower slot first), D = sqrt|lambda + alpha|, B's columns in rank order.
// The acceptance tests of kernels_eigh.hpp (twisted residual, orthogonality before / after the first polish round) decide
// workgroup-uniformly whether B and D are written; a refused instance is left to the caller's Jacobi.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_cma.hpp"
#include "kernels_refit.hpp"
#include "kernels_rollout.hpp"

namespace bbmpc {

constexpr int ES_N = 32, ES_P = 33, ES_PV = 36;      // largest dimension, LDS row pitch, pitch of the reflector rows (16-byte aligned)

#ifdef BBMPC_TU_CMA

struct alignas(16) EighSmallLds {
    float dd[ES_N + 8];          // d   (the Sturm chain reads one float4 ahead)
    float e2p[ES_N + 8];         // e2p[i] = ee[i-1]^2 floored at 1e-36 (kernels_eigh.hpp)
    float ee[ES_N + 8];          // thresholded e (ee[k] couples k, k + 1)
    float tau[ES_N], lam[ES_N];
    float V[ES_N][ES_PV];        // reflector k in row k (read back four elements at a time, every lane the same address)
    float xb[ES_N + 4], wb[ES_N + 4];   // the step's x and w for the same kind of read (all three: element i at (i & 1) * 16 + (i >> 1))
    float Z[ES_N][ES_P];         // Z[i][j]: component i of the eigenvector of slot j
    float Z2[ES_N][ES_P];
    float Pm[ES_N][ES_P];        // 1.5 I - 0.5 Z^T Z
    float fw[ES_N][ES_P];        // fw[i][slot]: D+ pivots of T - lam_slot
    float bw[ES_N][ES_P];        // bw[i][slot]: D- pivots
    int bs[ES_N], bt[ES_N];      // unreduced block [bs[i], bt[i]) around index i
    float alpha, tn, gl, gu;
    unsigned g1, g2, resid;      // max |Z^T Z - I| before the polish / after one round, max twisted residual (float bits)
};

__device__ __forceinline__ float es_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float es_half_sum(float v) { v = row16_sum(v); return v + __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float es_half_min(float v) { v = row16_min(v); return fminf(v, __shfl_xor(v, 16, 64)); }
__device__ __forceinline__ float es_half_max(float v) { return -es_half_min(-v); }
// sums / minima over the wave when lanes 32..63 hold the neutral element
__device__ __forceinline__ float es_sum32(float v) { v = row16_sum(v); return es_readlane(v, 0) + es_readlane(v, 16); }
__device__ __forceinline__ float es_min32(float v) { v = row16_min(v); return fminf(es_readlane(v, 0), es_readlane(v, 16)); }
// the value the lower half (lanes 0..31) / the upper half holds in the lane's column, in both halves
__device__ __forceinline__ void es_halves(float v, float& lo, float& hi) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    lo = __int_as_float(r[0]); hi = __int_as_float(r[1]);
}
// 1/x to fp32 accuracy: v_rcp_f32 (1 ulp) and one Newton step
__device__ __forceinline__ float es_rcp(float x) { const float r = __builtin_amdgcn_rcpf(x); return r * fmaf(-x, r, 2.0f); }

// blockDim.x == 1024.  Returns (uniformly) whether p.B / p.Dd of instance g were written.
// Not inlined: inside the one-launch control step (kernels_fused_cma.hpp, 128 registers per lane) the unrolled stages would
// share the caller's register allocation and spill.
__device__ __attribute__((noinline)) bool cma_eigh_small_body(const CmaArgs& p, int g, bool force_fail) {
    __shared__ EighSmallLds L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n;
    const float* __restrict__ C = p.C + (size_t)g * n * n;
#ifdef BBMPC_KERNEL_DBG
    long long es_t0 = (long long)wall_clock64(), es_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long es_w0 = es_t0, es_c0 = (long long)clock64();
#define ES_MARK(i) do { const long long now_ = (long long)wall_clock64(); es_acc[i] += now_ - es_t0; es_t0 = now_; } while (0)
#else
#define ES_MARK(i) do {} while (0)
#endif
    // ---- 1. tridiagonalisation (wave 0): lane (h, j) = (lane >> 5, lane & 31) holds the rows of parity h of column j of
    // E = C - alpha I, row i = 2 r + h in register r.  Vector elements needed "by row" (x_i, v_i, w_i) are read from LDS,
    // de-interleaved by parity so that a half's sixteen are four 16-byte reads at an address the half shares; values needed
    // "by column" are the same in both halves.  The halves meet through v_permlane32_swap (no LDS round trip).
    if (wv == 0) {
        const int h = lane >> 5, j = lane & 31;
        const bool cj = j < n;
        float a[ES_N / 2];
#pragma unroll
        for (int r = 0; r < ES_N / 2; ++r) {
            const int i = 2 * r + h;
            const float v = C[(size_t)min(i, n - 1) * n + min(j, n - 1)];
            a[r] = (cj && i < n) ? v : 0.0f;
        }
        const float dj = C[(size_t)min(lane, n - 1) * (n + 1)];
        const float alpha = es_sum32(lane < n ? dj : 0.0f) / (float)n;
#pragma unroll
        for (int r = 0; r < ES_N / 2; ++r) a[r] = a[r] - ((2 * r + h == j && cj) ? alpha : 0.0f);
        float* xb = L.xb;                                          // [2][16]: x_i at (i & 1) * 16 + (i >> 1), as V[k] and wb
        const int dei = (j & 1) * 16 + (j >> 1);                   // where element j of a column-wise vector goes
#pragma unroll
        for (int k = 0; k < ES_N; ++k) {
            const int hk = k & 1, rk = k >> 1;                     // row k: half hk, register rk
            if (lane == k + 32 * hk) L.dd[k] = a[rk];
            if (k + 2 < n) {
                // x = E[k+1.., k] (= row k, by symmetry); E x does not wait for the reflector's scalars:
                // v = scale x + (1 - scale x_{k+1}) e_{k+1}, so E v = scale (E x) + (1 - scale x_{k+1}) E[:, k+1]
                float xlo, xhi;
                es_halves(a[rk], xlo, xhi);
                const float xr = hk ? xhi : xlo;                   // E[k][j] in both halves
                const float x = (j > k) ? xr : 0.0f;               // (zero for columns >= n)
                if (h == 0) xb[dei] = x;
                const float ain = es_readlane(x, k + 1);
                const float xs = (j > k + 1) ? x : 0.0f;
                const float sig = es_sum32(xs * xs);
                const int R0 = (((k + 1) >> 1) & ~3);              // first register group with a live row (rows <= k: x_i = 0)
                float pa[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int r4 = R0; r4 < ES_N / 2; r4 += 4) {
                    const float4 xq = *reinterpret_cast<const float4*>(&xb[h * 16 + r4]);
                    pa[0] = fmaf(a[r4], xq.x, pa[0]); pa[1] = fmaf(a[r4 + 1], xq.y, pa[1]);
                    pa[2] = fmaf(a[r4 + 2], xq.z, pa[2]); pa[3] = fmaf(a[r4 + 3], xq.w, pa[3]);
                }
                float plo, phi;
                es_halves((pa[0] + pa[1]) + (pa[2] + pa[3]), plo, phi);
                const float ex = plo + phi;                        // (E x)_j, the same in both halves
                float beta = ain, tauk = 0.0f, scale = 0.0f;
                if (sig != 0.0f) {                                         // slarfg with v_sqrt / v_rcp + Newton
                    beta = -copysignf(__builtin_amdgcn_sqrtf(fmaf(ain, ain, sig)), ain);
                    tauk = (beta - ain) * es_rcp(beta);
                    scale = es_rcp(ain - beta);
                }
                const float v = (j == k + 1) ? 1.0f : xs * scale;
                if (lane == 0) { L.ee[k] = beta; L.tau[k] = tauk; }
                if (h == 0) L.V[k][dei] = v;
                if (tauk != 0.0f) {
                    // p = tau E v, w = p - (tau/2 p.v) v, E -= v w^T + w v^T
                    float clo, chi;
                    es_halves(a[(k + 1) >> 1], clo, chi);
                    const float ek1 = ((k + 1) & 1) ? chi : clo;           // E[k+1][j]
                    const float pj = tauk * fmaf(scale, ex, fmaf(-scale, ain, 1.0f) * ek1);
                    const float pv = es_sum32(pj * v);
                    const float cc = 0.5f * tauk * pv;
                    const float w = (j > k) ? fmaf(-cc, v, pj) : 0.0f;
                    if (h == 0) L.wb[dei] = w;
#pragma unroll
                    for (int r4 = R0; r4 < ES_N / 2; r4 += 4) {
                        const float4 vq = *reinterpret_cast<const float4*>(&L.V[k][h * 16 + r4]);
                        const float4 wq = *reinterpret_cast<const float4*>(&L.wb[h * 16 + r4]);
                        // (v_i = w_i = 0 for i <= k)
                        a[r4] = fmaf(-vq.x, w, fmaf(-wq.x, v, a[r4]));
                        a[r4 + 1] = fmaf(-vq.y, w, fmaf(-wq.y, v, a[r4 + 1]));
                        a[r4 + 2] = fmaf(-vq.z, w, fmaf(-wq.z, v, a[r4 + 2]));
                        a[r4 + 3] = fmaf(-vq.w, w, fmaf(-wq.w, v, a[r4 + 3]));
                    }
                }
            } else {
                // no reflector: e_k = E[k+1][k] sits in lane (half of row k + 1, column k)
                if (k + 1 < ES_N) { if (lane == k + 32 * ((k + 1) & 1)) L.ee[k] = a[(k + 1 < ES_N ? k + 1 : k) >> 1]; }
                else if (lane == 0) L.ee[k] = 0.0f;
                if (lane == 0) L.tau[k] = 0.0f;
                if (h == 0) L.V[k][dei] = 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float dv = L.dd[j], ev = L.ee[j], tv = L.tau[j];
        ES_MARK(0);
        // ---- T (lane i: d_i, e_i): norm, split threshold, Gershgorin interval, e^2, the unreduced blocks -- from registers
        const float di = lane < n ? dv : 0.0f, ei = (lane + 1 < n) ? ev : 0.0f;
        const float tn = -es_min32(-fmaxf(fabsf(di), fabsf(ei)));
        const float thr = 4.0f * 1.1920929e-07f * fmaxf(fabsf(alpha), tn);
        const float eth = fabsf(ei) <= thr ? 0.0f : ei;
        // (the shuffles outside the selects: a ds_bpermute under a lane mask reads 0 from the masked-off source lane)
        const float ei_up = __shfl_up(ei, 1, 64), eth_up = __shfl_up(eth, 1, 64);
        const float em = lane > 0 ? ei_up : 0.0f, ethm = lane > 0 ? eth_up : 0.0f;
        const float rad = fabsf(eth) + fabsf(em);                       // (unthresholded neighbour: only widens the interval)
        float gl = es_min32(lane < n ? di - rad : 3.0e38f), gu = -es_min32(lane < n ? -(di + rad) : 3.0e38f);
        const float span = gu - gl;
        gl -= span * (2.0f * 1.1920929e-07f * (float)n) + thr;
        gu += span * (2.0f * 1.1920929e-07f * (float)n) + thr;
        // split after index k where the coupling is zero (and after n - 1): the block of i starts behind the last split
        // below i and ends at the first split at or above i
        const unsigned long long splits = __ballot(lane < ES_N && (lane + 1 >= n || eth == 0.0f));
        const int ii = min(lane, n - 1);
        const unsigned long long below = splits & ((1ull << ii) - 1ull), above = splits >> ii;
        const int bs_i = below ? 64 - __builtin_clzll(below) : 0;
        const int bt_i = ii + __builtin_ctzll(above) + 1;
        if (lane < ES_N) {
            L.dd[lane] = di; L.ee[lane] = eth; L.tau[lane] = tv;
            L.bs[lane] = bs_i; L.bt[lane] = bt_i;
        }
        if (lane < ES_N + 8) L.e2p[lane] = (lane > 0 && lane < n) ? fmaxf(ethm * ethm, 1.0e-36f) : 1.0e-36f;
        if (lane < 8) { L.dd[ES_N + lane] = 0.0f; L.ee[ES_N + lane] = 0.0f; }
        if (lane == 0) { L.alpha = alpha; L.tn = tn; L.gl = gl; L.gu = gu; L.g1 = 0u; L.g2 = 0u; L.resid = 0u; }
    }
    __syncthreads();
    ES_MARK(1);
    // ---- 2a. eigenvalues: one half-wave per slot, five passes of 33-fold multisection on Sturm counts
    const int half = lane >> 5, sub = lane & 31;
    const float alpha = L.alpha, tn = L.tn;
    {
        const int j = 2 * wv + half;
        const bool live = j < n;
        const int s = live ? L.bs[j] : 0, t = live ? L.bt[j] : 1;
        const int m = (live ? j : 0) - s;
        float lo = L.gl, hi = L.gu;
        const bool whole_w = __all(!live || (s == 0 && t == n)) != 0;
        if (t - s > 1) {
            const float4* dd4 = reinterpret_cast<const float4*>(L.dd);
            const float4* e24 = reinterpret_cast<const float4*>(L.e2p);
            const int n4 = (n + 3) >> 2;
            for (int pass = 0; pass < 5; ++pass) {
                const float h = (hi - lo) * (1.0f / 33.0f);
                const float x = fmaf((float)(sub + 1), h, lo);
                int cnt = 0;
                float qv = 1.0f;
                float4 dA = dd4[0], eA = e24[0];
                if (whole_w) {
                    // every slot of the wave belongs to the one unreduced block [0, n) (the usual case once C has left the
                    // identity): no range test on the chain -- with four waves per SIMD the phase is issue bound
                    for (int i4 = 0; i4 < (n >> 2); ++i4) {
                        const float4 dB = dd4[i4 + 1], eB = e24[i4 + 1];
                        const float dx4[4] = {dA.x - x, dA.y - x, dA.z - x, dA.w - x}, ev4[4] = {eA.x, eA.y, eA.z, eA.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            qv = fmaf(-ev4[c], __builtin_amdgcn_rcpf(qv), dx4[c]);
                            cnt += qv < 0.0f ? 1 : 0;
                        }
                        dA = dB; eA = eB;
                    }
                    {
                        const float dx4[4] = {dA.x - x, dA.y - x, dA.z - x, dA.w - x}, ev4[4] = {eA.x, eA.y, eA.z, eA.w};
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            if (c < (n & 3)) {                                         // (uniform)
                                qv = fmaf(-ev4[c], __builtin_amdgcn_rcpf(qv), dx4[c]);
                                cnt += qv < 0.0f ? 1 : 0;
                            }
                    }
                } else
                for (int i4 = 0; i4 < n4; ++i4) {
                    const float4 dB = dd4[i4 + 1], eB = e24[i4 + 1];
                    const float dx4[4] = {dA.x - x, dA.y - x, dA.z - x, dA.w - x}, ev4[4] = {eA.x, eA.y, eA.z, eA.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i = 4 * i4 + c;
                        qv = fmaf(-ev4[c], __builtin_amdgcn_rcpf(qv), dx4[c]);        // (no pivmin test: see e2p, kernels_eigh.hpp)
                        cnt += ((unsigned)(i - s) < (unsigned)(t - s) && qv < 0.0f) ? 1 : 0;
                    }
                    dA = dB; eA = eB;
                }
                // eigenvalue m of the block lies above every shift with count <= m and not above any shift with count > m
                lo = es_half_max(cnt <= m ? x : lo);
                hi = es_half_min(cnt > m ? x : hi);
                if (!(hi > lo)) hi = lo;
            }
        } else {
            lo = hi = L.dd[s];
        }
        if (sub == 0) L.lam[j & (ES_N - 1)] = live ? 0.5f * (lo + hi) : -3.0e38f;
    }
    __syncthreads();
    ES_MARK(2);
    // ---- 2b. eigenvectors by twisted factorisation (wave 0): lane = slot; lanes 0..31 run the forward (D+) recurrence and
    // the part of the vector above the twist, lanes 32..63 the backward (D-) recurrence and the part below -- the same
    // code in the lane's OWN coordinates q (q = i for the forward lanes, q = 31 - i for the backward ones; T is padded with
    // zeros to 32, and a zero coupling restarts a recurrence by itself, so neither the padding nor the blocks need care on
    // the chains).  All loops are unrolled: d, e in registers, the LDS traffic (pivots out, pivots in) off the chains.
    if (wv == 0) {
        const int slot = sub;
        const bool live = slot < n;
        const float lam = live ? L.lam[slot] : 0.0f;
        const int s = live ? L.bs[slot] : 0, t = live ? L.bt[slot] : 1;
        const float pivmin = fmaxf(1.0e-30f, tn * tn * 1.0e-30f);
        float (*own)[ES_P] = half ? L.bw : L.fw;
        float (*oth)[ES_P] = half ? L.fw : L.bw;
        float Dl[ES_N], El[ES_N];                                      // own order: El[q] couples positions q - 1 and q
#pragma unroll
        for (int q = 0; q < ES_N; ++q) {
            Dl[q] = L.dd[half ? ES_N - 1 - q : q];
            const float ec = L.ee[half ? ES_N - 1 - q : (q > 0 ? q - 1 : 0)];
            El[q] = (q == 0 && !half) ? 0.0f : ec;                    // (backward: ee[31] = 0)
        }
        {
            float rp = 0.0f;                                           // 1 / previous pivot (first position: no coupling)
#pragma unroll
            for (int q = 0; q < ES_N; ++q) {
                float piv = (Dl[q] - lam) - (El[q] * rp) * El[q];
                piv = fabsf(piv) < pivmin ? -pivmin : piv;
                own[half ? ES_N - 1 - q : q][slot] = piv;
                rp = es_rcp(piv);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // gamma_i = D+_i + D-_i - (d_i - lam); r = argmin |gamma_i| inside the slot's block, the lowest i among equals (both
        // halves see the same values and find the same r)
        float gmin = 3.0e38f;
        int r = s;
#pragma unroll
        for (int q = 0; q < ES_N; ++q) {
            const int i = half ? ES_N - 1 - q : q;
            const float gam = fabsf((own[i][slot] + oth[i][slot]) - (Dl[q] - lam));
            if (i >= s && i < t && (gam < gmin || (gam == gmin && i < r))) { gmin = gam; r = i; }
        }
        // z_r = 1 and, from the twist outwards on the lane's own side (q < q_r), z_q = -(E_{q+1} / P_q) z_{q+1}
        const int q_r = half ? ES_N - 1 - r : r, q_lo = half ? ES_N - t : s;
        float zz = half ? 0.0f : 1.0f;                                 // sum of squares of this half's part (z_r counted by the upper half)
        // the pivots come in as one batch of reads in front of the chain and the vector stays in the same registers until it
        // is normalised: one store per row, no read-modify-write pass
        float P[ES_N];
#pragma unroll
        for (int q = 0; q < ES_N - 1; ++q) P[q] = own[half ? ES_N - 1 - q : q][slot];
        {
            float z = 1.0f;
#pragma unroll
            for (int q = ES_N - 2; q >= 0; --q) {
                const bool act = live && q < q_r && q >= q_lo;
                if (act) {
                    z = -(El[q + 1] * es_rcp(P[q])) * z;
                    zz = fmaf(z, z, zz);
                }
                P[q] = act ? z : 0.0f;                                // (outside the block the vector is zero)
            }
        }
        zz = zz + __shfl_xor(zz, 32, 64);
        const float rn = 1.0f / sqrtf(zz);
        if (!half) L.Z[r][slot] = live ? rn : 0.0f;
#pragma unroll
        for (int q = 0; q < ES_N - 1; ++q) {
            const int i = half ? ES_N - 1 - q : q;
            if (q < q_r) L.Z[i][slot] = P[q] * rn;
        }
        float res = (live && !half) ? gmin * rn : 0.0f;             // |(T - lam) z| for the normalised z
        res = -es_min32(-res);
        if (lane == 0) L.resid = __float_as_uint(res);
    }
    __syncthreads();
    ES_MARK(3);
    // ---- 3. Newton-Schulz polish: thread (a, b) of the 32 x 32 matrices; the second round only when the twisted vectors
    // were less orthogonal than EIGH_ONE_ROUND (kernels_eigh.hpp)
    float (*Zf)[ES_P] = L.Z2;
    {
        const int a = tid >> 5, b = tid & 31;
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            float (*Zi)[ES_P] = round == 0 ? L.Z : L.Z2;
            float (*Zo)[ES_P] = round == 0 ? L.Z2 : L.Z;
            float g0 = 0.0f, g1 = 0.0f;
#pragma unroll 8
            for (int i = 0; i < ES_N; i += 2) {
                g0 = fmaf(Zi[i][a], Zi[i][b], g0);
                g1 = fmaf(Zi[i + 1][a], Zi[i + 1][b], g1);
            }
            const float gab = g0 + g1, id = a == b ? 1.0f : 0.0f;
            const bool inn = a < n && b < n;
            float dev = inn ? fabsf(gab - id) : 0.0f;
            dev = -wave_min(-dev);
            if (lane == 0) atomicMax(round == 0 ? &L.g1 : &L.g2, __float_as_uint(dev));
            L.Pm[a][b] = inn ? fmaf(-0.5f, gab, 1.5f * id) : 0.0f;
            __syncthreads();
            float z0 = 0.0f, z1 = 0.0f;                               // Zo[a][b] = sum_c Zi[a][c] Pm[c][b]
#pragma unroll 8
            for (int c = 0; c < ES_N; c += 2) {
                z0 = fmaf(Zi[a][c], L.Pm[c][b], z0);
                z1 = fmaf(Zi[a][c + 1], L.Pm[c + 1][b], z1);
            }
            Zo[a][b] = z0 + z1;
            __syncthreads();
            Zf = Zo;
            if (round == 0 && __uint_as_float(L.g1) <= 1.0e-3f) break;
        }
    }
    ES_MARK(4);
    // ---- acceptance (kernels_eigh.hpp: eigh_instance_ok): nothing of B / D has been touched yet
    {
        const float q0 = __uint_as_float(L.g1), q1 = __uint_as_float(L.g2), res = __uint_as_float(L.resid);
        const float scale = fmaxf(fabsf(alpha), tn);
        const bool ok = !force_fail && q0 <= 0.25f && q1 <= 2.0e-3f && res <= 1.0e-5f * scale && scale < 3.0e38f;
#ifdef BBMPC_KERNEL_DBG
        if (!ok && g == 0 && tid == 0) printf("[es] refused: g1 %.3e g2 %.3e res %.3e scale %.3e tn %.3e\n", q0, q1, res, scale, tn);
#endif
        if (!ok) return false;
    }
    // ---- 4. back-transformation (half-wave per column, lane = row), rank, store
    {
        const int j = 2 * wv + half;
        const bool live = j < n;
        float z = Zf[sub][j & (ES_N - 1)];
        for (int k = n - 3; k >= 0; --k) {
            const float v = L.V[k][(sub & 1) * 16 + (sub >> 1)];          // (stored de-interleaved by row parity)
            const float tk = L.tau[k];
            const float dot = es_half_sum(v * z);
            z = fmaf(-(tk * dot), v, z);
        }
        const float lj = L.lam[j & (ES_N - 1)], lo_ = L.lam[sub];
        // ranked by |eigenvalue of C| (the SVD's order, cma_es.py:195-197), as in kernels_eigh.hpp
        const float sj = fabsf(lj + alpha), so = fabsf(lo_ + alpha);
        const float before = (sub < n && (so > sj || (so == sj && sub < j))) ? 1.0f : 0.0f;
        const int rank = (int)es_half_sum(before);
        if (live && sub < n) p.B[(size_t)g * n * n + (size_t)sub * n + rank] = z;
        if (live && sub == 0) p.Dd[(size_t)g * n + rank] = sqrtf(fabsf(lj + alpha));
    }
    ES_MARK(5);
#ifdef BBMPC_KERNEL_DBG
    if (g == 0 && tid == 0)
        printf("[es] n=%d | tridiag %lld  setup+barrier %lld  multisection %lld  twisted %lld  polish %lld  back+store %lld (10 ns units) g1 %.2e g2 %.2e res %.2e | %lld shader cycles in %lld x 10 ns\n", n,
               es_acc[0], es_acc[1], es_acc[2], es_acc[3], es_acc[4], es_acc[5], __uint_as_float(L.g1), __uint_as_float(L.g2), __uint_as_float(L.resid), (long long)clock64() - es_c0, (long long)wall_clock64() - es_w0);
#endif
    return true;
}

// warm start of the Jacobi: At[j][:] = C B0[:, j], one fmaf chain over k per element (k_cma_warm's sums)
__device__ __forceinline__ void cma_warm_small_body(const CmaArgs& p, int g, float* At_all) {
    const int n = p.n, nn = n * n;
    const float* C = p.C + (size_t)g * nn;
    const float* B = p.B + (size_t)g * nn;
    float* At = At_all + (size_t)g * nn;
    for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
        const int j = idx / n, e = idx - j * n;
        float acc = 0.0f;
        for (int kk = 0; kk < n; ++kk) acc = fmaf(C[(size_t)kk * n + e], B[(size_t)kk * n + j], acc);
        At[(size_t)j * n + e] = acc;
    }
}

// B, D of instance g from C: the direct solver, or -- when it refuses the instance -- the warm-started Jacobi as before
// (k_cma_warm, k_cma_svd_small, k_cma_svd_finish: the same device functions).  blockDim.x == 1024, at_s: n * n floats.
__device__ __forceinline__ void cma_factor_small_body(const CmaArgs& p, int g, float* evec, float* eval, int* info, bool force_fail, float* at_s) {
    if (cma_eigh_small_body(p, g, force_fail)) return;
    cma_warm_small_body(p, g, evec);
    __syncthreads();
    cma_svd_small_body<4>(p, g, evec, 15, at_s);
    __syncthreads();
    cma_svd_finish_body(p, g, evec, eval, info);
}

// grid G, block 1024, dynamic LDS n * n floats
static __global__ __launch_bounds__(1024) void k_cma_factor_small(CmaArgs p, float* evec, float* eval, int* info, int force_fail) {
    extern __shared__ __attribute__((aligned(16))) float at_s[];
    cma_factor_small_body(p, blockIdx.x, evec, eval, info, force_fail != 0, at_s);
}

// ---- Three launches per iteration for small instances (n <= 32) instead of eleven: sample | roll out | update.
// The per-iteration kernels of kernels_cma.hpp are a few microseconds of work each behind ~4.5 us of launch (and the
// host cannot enqueue 57 launches per control step as fast as the device retires them); the one-launch control step
// (kernels_fused_cma.hpp) puts phases that want many workgroups on one.  Here the phases that are one workgroup per
// instance anyway (selection, evolution paths, covariance, factorisation) share a launch, and noise + B D + the sampling
// product share another, spread over the population.  Same device functions / same operation order as the separate
// kernels: bit-identical control steps (tests/test_gpu_cmaes.py).

// z ~ N(0, I), y = z (B D), samples = m + sigma y  (cma_es.py:139-141; k_cma_noise + k_cma_bd + k_cma_gemm_y)
// grid (ceil(N / 64), G), block 256: 64 particles per workgroup, the four waves share the Philox blocks of the draws and
// take eight rows of the product each.  LDS: B D (pitch 32) | z [n][64]
// cs (optional): [n][64] floats of LDS that receive the samples too (for the rollouts of k_cma_sample_roll_small)
__device__ __forceinline__ void cma_sample_small_body(const CmaArgs& p, int g, float* bd, float (*zs)[64], float (*cs)[64]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n, Nst = p.Nst;
    const int q = blockIdx.x * 64 + lane;
    const size_t off = (size_t)g * n, nn = (size_t)n * n;
    for (int i = tid; i < ES_N * ES_N; i += 256) {
        const int l = i >> 5, c = i & 31;
        bd[i] = (l < n && c < n) ? p.B[(size_t)g * nn + (size_t)l * n + c] * p.Dd[off + c] : 0.0f;      // k_cma_bd
    }
    const bool live = q < p.N;
    if (p.inj) {
        for (int l = wv; l < n; l += 4) zs[l][lane] = live ? p.inj[(off + l) * Nst + q] : 0.0f;
    } else {
        // elem_normal's draws (k_cma_noise): one Philox block gives the normals of four consecutive elements of an agent
        const int apg = p.agents_per_group, hu = n / apg, nb = (hu + 3) >> 2;
        for (int t = wv; t < apg * nb; t += 4) {
            const int a = t / nb, j4 = (t - a * nb) * 4;
            const U4 b = rng_block(p.key, 4u, p.iter, (uint32_t)(q + p.pop_offset), (uint32_t)(p.agent_offset + g * apg + a), (uint32_t)j4);
            float z4[4];
            words_to_normal2(b.x, b.y, z4[0], z4[1]);
            words_to_normal2(b.z, b.w, z4[2], z4[3]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j4 + u < hu) zs[a * hu + j4 + u][lane] = z4[u];
        }
    }
    __syncthreads();
    const int i0 = 8 * wv;
    if (live && i0 < n) {
        // one fmaf chain over l per element (k_cma_gemm_y's order), eight rows share each z load
        float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int l = 0; l < n; ++l) {
            const float zv = zs[l][lane];
            const float4 b0 = *reinterpret_cast<const float4*>(&bd[l * ES_N + i0]);
            const float4 b1 = *reinterpret_cast<const float4*>(&bd[l * ES_N + i0 + 4]);
            acc[0] = fmaf(b0.x, zv, acc[0]); acc[1] = fmaf(b0.y, zv, acc[1]); acc[2] = fmaf(b0.z, zv, acc[2]); acc[3] = fmaf(b0.w, zv, acc[3]);
            acc[4] = fmaf(b1.x, zv, acc[4]); acc[5] = fmaf(b1.y, zv, acc[5]); acc[6] = fmaf(b1.z, zv, acc[6]); acc[7] = fmaf(b1.w, zv, acc[7]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u < n) {
                const float x = p.m[off + i0 + u] + p.sigma[off + i0 + u] * acc[u];
                p.cand[(off + i0 + u) * Nst + q] = x;
                if (cs) cs[i0 + u][lane] = x;
            }
    }
}

static __global__ __launch_bounds__(256) void k_cma_sample_small(CmaArgs p) {
    __shared__ __attribute__((aligned(16))) float bd[ES_N * ES_N];
    __shared__ float zs[ES_N][64];
    cma_sample_small_body(p, blockIdx.y, bd, zs, nullptr);
}

// ... and, for the analytic pendulum with one agent per instance, the rollouts of the workgroup's 64 particles behind it
// (k_rollout_pendulum<SRC_BUF, penalty>: clip, squared clip distance, H steps, reward minus penalty -- cma_es.py:144-157),
// one launch fewer per iteration.  The clipped candidates are not written back (kernels_cma.hpp: the path update clips
// the elites it reads).
template <bool FASTM>
__global__ __launch_bounds__(256) void k_cma_sample_roll_small(CmaArgs p, const float* state, int fix_q1) {
    __shared__ __attribute__((aligned(16))) float bd[ES_N * ES_N];
    __shared__ float zs[ES_N][64];
    __shared__ float cs[ES_N][64];
    const int g = blockIdx.y;
    cma_sample_small_body(p, g, bd, zs, cs);
    __syncthreads();
    const int tid = threadIdx.x, q = blockIdx.x * 64 + tid;
    if (tid >= 64 || q >= p.N) return;
    const float lo0 = p.lo[0], hi0 = p.hi[0];
    Roller<FASTM> roll(fix_q1 != 0, state[g * 3 + 0], state[g * 3 + 1], state[g * 3 + 2]);
    float total = 0.0f, pen = 0.0f;
    for (int t = 0; t < p.n; ++t) {
        float x = cs[t][tid];
        const float xf = clipf(x, lo0, hi0);
        const float d = x - xf;
        pen = pen + d * d;
        x = xf;
        roll.step_acc(x);
    }
    total = roll.total();
    if (total != total) total = -1.0e6f;
    const float nr = sqrtf(pen);
    pen = nr * nr;
    total = total - pen;
    const_cast<float*>(p.rewards)[(size_t)g * p.Nst + q] = total;
}

// covariance on the upper triangle, mirrored (k_cma_cov's sums), by one workgroup.  ye: k * n floats of LDS when the elite
// deviations fit there (one round trip for all of them instead of two loads per term of every sum), or null
__device__ __forceinline__ void cma_cov_small_body(const CmaArgs& p, int g, float* ye = nullptr) {
    const int n = p.n, nn = n * n;
    const size_t off = (size_t)g * n;
    const float* Ye = p.Ye + (size_t)g * p.k * n;
    float* C = p.C + off * n;
    if (ye) {
        for (int i = threadIdx.x; i < p.k * n; i += blockDim.x) ye[i] = Ye[i];
        __syncthreads();
        Ye = ye;
    }
    for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
        const int r = idx / n, c = idx - r * n;
        if (r > c) continue;
        float ys = 0.0f;
        for (int i = 0; i < p.k; ++i) ys = fmaf(Ye[(size_t)i * n + r] * Ye[(size_t)i * n + c], p.weights[i], ys);
        const float v = ((1.0f - p.c.c1) - p.c.c_mu) * C[(size_t)r * n + c] + (p.c.c1 * p.p_C[off + r]) * p.p_C[off + c] +      /* (c1 * p_C) * p_C^T as cma_es.py:183 evaluates it */
                        p.c.c_mu * ys;
        C[(size_t)r * n + c] = v;
        C[(size_t)c * n + r] = v;
    }
}

// selection, evolution paths / step size / mean, covariance, factorisation  (cma_es.py:158-206)
// grid G, block 1024; dynamic LDS: max(selection's words, n * n floats, ye_floats [>= n*n + k*n + 2n: the path update's staging], or ye_floats = 0)
static __global__ __launch_bounds__(1024) void k_cma_update_small(CmaArgs p, float* evec, float* eval, int* info, int force_fail, int ye_floats) {
    extern __shared__ __attribute__((aligned(16))) float usm[];
    const int g = blockIdx.x;
#ifdef BBMPC_KERNEL_DBG
    long long um[5]; um[0] = (long long)wall_clock64();
#define UPD_MARK(i) do { um[i] = (long long)wall_clock64(); } while (0)
#else
#define UPD_MARK(i) do {} while (0)
#endif
    cma_select_body(p, g, usm);
    __syncthreads();
    UPD_MARK(1);
    if (ye_floats) cma_paths_body_t<true>(p, g, usm); else cma_paths_body(p, g);
    __syncthreads();
    UPD_MARK(2);
    cma_cov_small_body(p, g, (size_t)p.k * p.n <= (size_t)ye_floats ? usm : nullptr);
    __syncthreads();
    UPD_MARK(3);
    cma_factor_small_body(p, g, evec, eval, info, force_fail != 0, usm);
#ifdef BBMPC_KERNEL_DBG
    UPD_MARK(4);
    if (g == 0 && threadIdx.x == 0 && p.iter == 2) printf("[upd] select %lld  paths %lld  cov %lld  factor %lld (10 ns)\n", um[1] - um[0], um[2] - um[1], um[3] - um[2], um[4] - um[3]);
#endif
}

// the last iteration's update on the analytic pendulum (one agent per instance, dim_U = 1): action = m[0] (cma_es.py:211-212),
// exploration noise, predicted next state + reward, packed record, completion word (optimizer_base.py:82-94) in the same
// launch -- k_take_first and k_finalize_pendulum otherwise
static __global__ __launch_bounds__(1024) void k_cma_update_final_small(CmaArgs p, float* evec, float* eval, int* info, int force_fail, int ye_floats,
                                                                        FinalArgs fin, unsigned* done_flag, unsigned* done_count, unsigned done_value) {
    extern __shared__ __attribute__((aligned(16))) float usm[];
    const int g = blockIdx.x;
    cma_select_body(p, g, usm);
    __syncthreads();
    if (ye_floats) cma_paths_body_t<true>(p, g, usm); else cma_paths_body(p, g);
    __syncthreads();
    // the mean is final here: the record does not wait for the covariance and its factorisation
    if (threadIdx.x == 0) {
        finalize_pendulum_agent(fin, g, p.m[(size_t)g * p.n]);
        publish_records_done(done_flag, done_count, done_value, gridDim.x);
    }
    cma_cov_small_body(p, g, (size_t)p.k * p.n <= (size_t)ye_floats ? usm : nullptr);
    __syncthreads();
    cma_factor_small_body(p, g, evec, eval, info, force_fail != 0, usm);
}

#endif  // BBMPC_TU_CMA
}  // namespace bbmpc
