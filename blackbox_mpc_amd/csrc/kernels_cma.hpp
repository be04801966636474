// CMA-ES kernels -- CMAESOptimizer  optimizers/cma_es.py:43-227.
//
// One "group" is one CMA-ES instance of dimension n:
//   coupled mode (reference behaviour): a single group, n = A*H*U, rewards summed over agents (quirk Q6)
//   per-agent mode (BBMPC_CMAES_PER_AGENT): A groups of n = H*U, shards over GPUs like the other optimizers
// Vectors/matrices of group g live at offset g*n (resp. g*n*n); matrices are row-major.
// Candidates use the engine's internal layout [A][H*U][Nst]; with the joint index i = a*HU + j a group's
// sample matrix is simply X^T [n][Nst] starting at row g*n.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "kernels_refit.hpp"
#include "rng.hpp"
#include "topk.hpp"

namespace bbmpc {

struct CmaConst {            // cma_es.py:62-92,118-126 (computed on the host in fp32, same op order)
    float mu_eff, c_sigma, d_sigma, cc, c1, c_mu, e_norm, h_sigma;
};

struct CmaArgs {
    int N, A, HU, Nst, k;
    int G, n;                // groups, dimension per group
    int agents_per_group;    // A (coupled) or 1
    int agent_offset;
    CmaConst c;
    const float* weights;    // [k] recombination weights (the rest of the N weights are zero)
    float* m;                // [G*n]
    float* sigma;            // [G*n]
    float* C;                // [G][n][n]
    float* B;                // [G][n][n]
    float* Dd;               // [G*n]   diagonal of D
    float* p_sigma;          // [G*n]
    float* p_C;              // [G*n]
    float* BD;               // [G][n][n] scratch
    float* z;                // [G*n][Nst] standard normals (internal layout)
    float* cand;             // [G*n][Nst] samples (clipped in place by the rollout)
    const float* rewards;    // [A][Nst] (penalty already subtracted)
    int* eidx;               // [G][k]
    float* Ye;               // [G][k][n]  (x_sorted - m)/sigma of the elites
    float* xmean;            // [G*n]
    float* ymean;            // [G*n]
    RngKey key;
    uint32_t iter;
    const float* inj;        // injected z (internal layout) or null
    int pop_offset;          // global index of local particle 0 (population sharding): draws keyed by the GLOBAL particle
    const float* lo;         // [U] action bounds: the elites are clipped where they are read (samples_feasible, cma_es.py:144-145) -- the
    const float* hi;         // rollout does not have to write the clipped candidates back (clipping twice changes nothing)
    int U;
};

// z ~ N(0,1): element j of a 4-block uses Box-Muller on word pairs (w0,w1)->(z0,z1), (w2,w3)->(z2,z3)
__device__ __forceinline__ float elem_normal(const RngKey& key, uint32_t iter, int n, int ga, int j) {
    const U4 b = rng_block(key, 4u, iter, (uint32_t)n, (uint32_t)ga, (uint32_t)j);
    float z0, z1;
    if ((j & 2) == 0) words_to_normal2(b.x, b.y, z0, z1);
    else words_to_normal2(b.z, b.w, z0, z1);
    return (j & 1) ? z1 : z0;
}


// Everything below -- device code and the host-side size helpers -- is compiled in the CMA-ES translation unit only
// (csrc/bbmpc_cma.hip defines BBMPC_TU_CMA); the other units see the argument structures above.
#ifdef BBMPC_TU_CMA

// grid (ceil(N/256), HU, A)
static __global__ void k_cma_noise(CmaArgs p) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, a = blockIdx.z;
    if (n >= p.N) return;
    const size_t i = ((size_t)a * p.HU + j) * p.Nst + n;
    p.z[i] = p.inj ? p.inj[i] : elem_normal(p.key, p.iter, n + p.pop_offset, p.agent_offset + a, j);
}

// BD = B @ D (D diagonal)   cma_es.py:140
static __global__ void k_cma_bd(CmaArgs p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nn = (size_t)p.n * p.n;
    if (i >= nn * p.G) return;
    const int g = (int)(i / nn), c = (int)(i % p.n);
    p.BD[i] = p.B[i] * p.Dd[(size_t)g * p.n + c];
}

// Y^T[i][q] = sum_l BD[l][i] * Z^T[l][q]      (y = z @ BD, cma_es.py:140)  64x64 tiles, 4x4 per thread
// grid (ceil(N/64), ceil(n/64), G), block 256
static __global__ __launch_bounds__(256) void k_cma_gemm_y(CmaArgs p) {
    __shared__ float As[16][64 + 1];
    __shared__ float Bs[16][64 + 1];
    const int g = blockIdx.z;
    const int i0 = blockIdx.y * 64, q0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const float* A = p.BD + (size_t)g * p.n * p.n;            // [l][i]
    const float* Z = p.z + (size_t)g * p.n * p.Nst;           // [l][q]
    float acc[4][4] = {};
    for (int l0 = 0; l0 < p.n; l0 += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int l = e >> 6, c = e & 63;
            As[l][c] = (l0 + l < p.n && i0 + c < p.n) ? A[(size_t)(l0 + l) * p.n + i0 + c] : 0.0f;
            Bs[l][c] = (l0 + l < p.n && q0 + c < p.N) ? Z[(size_t)(l0 + l) * p.Nst + q0 + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            float av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = As[l][ty * 4 + r]; bv[r] = Bs[l][tx * 4 + r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(av[r], bv[c], acc[r][c]);
        }
        __syncthreads();
    }
    // samples = m + sigma * y   (cma_es.py:141), written straight into the candidate buffer
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty * 4 + r;
        if (i >= p.n) continue;
        const float mi = p.m[(size_t)g * p.n + i], si = p.sigma[(size_t)g * p.n + i];
        for (int c = 0; c < 4; ++c) {
            const int q = q0 + tx * 4 + c;
            if (q < p.N) p.cand[((size_t)g * p.n + i) * p.Nst + q] = mi + si * acc[r][c];
        }
    }
}

// The same product on the matrix cores for large n (n % 4 == 0): v_mfma_f32_16x16x4_f32, operands straight from L2 --
// for a fixed l both BD[l][i0 .. i0+15] and Z^T[l][q0 .. q0+15] are contiguous, which is exactly the A / B fragment of
// lane (l & 3, i or q) -- no LDS staging.  Workgroup tile 64 (i) x 64 (q): wave w owns rows 16w .. 16w+15 and four
// q fragments; the B fragments are shared by the four waves through L1.  76 -> ~20 us at n = 300, N = 2000, G = 4.
// (The small-n path keeps k_cma_gemm_y: its accumulation order is what the fused control-step kernel reproduces.)
static __global__ __launch_bounds__(256) void k_cma_gemm_y_mfma(CmaArgs p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int g = blockIdx.z, n = p.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 64 + wave * 16, q0 = blockIdx.x * 64;
    if (i0 >= n) return;
    const int lm = lane & 15, lk = lane >> 4;
    // B D is formed where it is loaded (the product k_cma_bd would have stored: same bits, one launch fewer)
    const float* A = p.B + (size_t)g * n * n + min(i0 + lm, n - 1);                  // + l * n
    const float dsc = p.Dd[(size_t)g * n + min(i0 + lm, n - 1)];
    const float* Z = p.z + (size_t)g * n * p.Nst + q0 + lm;                           // + l * Nst (+ 16 * f); Nst is a multiple of 64
    f4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 5
    for (int l0 = 0; l0 < n; l0 += 4) {
        const int l = l0 + lk;
        const float a = A[(size_t)l * n] * dsc;
        const float* zr = Z + (size_t)l * p.Nst;
        const float b0 = zr[0], b1 = zr[16], b2 = zr[32], b3 = zr[48];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b3, acc[3], 0, 0, 0);
    }
    // samples = m + sigma * y   (cma_es.py:141): lane holds rows i0 + 4*lk + r, column q0 + 16*f + lm
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * lk + r;
        if (i >= n) continue;
        const float mi = p.m[(size_t)g * n + i], si = p.sigma[(size_t)g * n + i];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int q = q0 + 16 * f + lm;
            if (q < p.N) p.cand[((size_t)g * n + i) * p.Nst + q] = mi + si * acc[f][r];
        }
    }
}

// per group: (sum of) rewards -> sorted top-k.   LDS: rsum[Nst] | hist | ekeys[kp]
// `part` != null (population sharded over ranks): this rank's particles only -- the local elites go to
// part[g][e] = (summed reward, GLOBAL particle index as bits, candidate x[n]) in sorted order; k_cma_merge merges the ranks'
// lists and puts the swarm's k elites into columns 0..k-1 of the candidate matrix, where the path update finds them.
__device__ __forceinline__ void cma_select_body(const CmaArgs& p, int g, float* smem, float* part = nullptr) {
    const int tid = threadIdx.x;
    float* rs = smem;
    uint32_t* hist = (uint32_t*)(rs + p.Nst);
    unsigned long long* ekeys = (unsigned long long*)(hist + TOPK_HIST_WORDS);
    __shared__ int eidx_s[1024];
    for (int q = tid; q < p.N; q += REFIT_THREADS) {
        float s = 0.0f;
        for (int a = 0; a < p.agents_per_group; ++a)              // tf.reduce_sum over agents, cma_es.py:158
            s = s + p.rewards[(size_t)(g * p.agents_per_group + a) * p.Nst + q];
        rs[q] = s;
    }
    __syncthreads();
    block_topk_sorted(rs, p.N, p.k, eidx_s, hist, ekeys, tid, REFIT_THREADS);   // argsort DESCENDING, first k
    if (part) {
        const int n = p.n, rowlen = n + 2;
        float* out = part + (size_t)g * p.k * rowlen;
        for (int e = tid; e < p.k; e += REFIT_THREADS) {
            out[(size_t)e * rowlen] = rs[eidx_s[e]];
            out[(size_t)e * rowlen + 1] = __int_as_float(eidx_s[e] + p.pop_offset);
        }
        const float* X = p.cand + (size_t)g * n * p.Nst;
        for (int t = tid; t < p.k * n; t += REFIT_THREADS) {
            const int e = t / n, c = t - e * n;
            out[(size_t)e * rowlen + 2 + c] = X[(size_t)c * p.Nst + eidx_s[e]];
        }
        return;
    }
    for (int e = tid; e < p.k; e += REFIT_THREADS) p.eidx[g * p.k + e] = eidx_s[e];
}
static __global__ __launch_bounds__(REFIT_THREADS) void k_cma_select(CmaArgs p, float* part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cma_select_body(p, blockIdx.x, smem, part);
}

// CMA-ES with the population sharded over ranks: all[r][g][e] = (reward, global index, x[n]), every rank's list sorted
// (larger reward first, ties -> lower index).  The rank of an entry in the merged order is its place in its own list plus,
// for every other list, the number of entries that precede it there (binary search); the first k take columns 0..k-1 of
// the candidate matrix (the local candidates are not needed any more) and eidx = 0..k-1, so the path / covariance update
// runs unchanged; gidx (optional) receives the global particle indices for the parity trace.  grid G, block 1024, k <= 1024
static __global__ __launch_bounds__(1024) void k_cma_merge(CmaArgs p, const float* all, int R, int* gidx) {
    __shared__ int s_src[1024];
    const int g = blockIdx.x, tid = threadIdx.x, n = p.n, k = p.k, rowlen = n + 2;
    const size_t pw = (size_t)p.G * k * rowlen;
    auto row = [&](int r, int e) { return all + pw * r + ((size_t)g * k + e) * rowlen; };
    for (int t = tid; t < R * k; t += blockDim.x) {
        const int r = t / k, e = t - r * k;
        const float v = row(r, e)[0];
        const int idx = __float_as_int(row(r, e)[1]);
        int rank = e;
        for (int o = 0; o < R; ++o) {
            if (o == r) continue;
            int lo = 0, hi = k;                                   // entries of list o that precede (v, idx)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float vo = row(o, mid)[0];
                const int io = __float_as_int(row(o, mid)[1]);
                if (vo > v || (vo == v && io < idx)) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            s_src[rank] = t;
            p.eidx[g * k + rank] = rank;
            if (gidx) gidx[g * k + rank] = idx;
        }
    }
    __syncthreads();
    float* X = p.cand + (size_t)g * n * p.Nst;
    for (int t = tid; t < k * n; t += blockDim.x) {
        const int w = t / n, c = t - w * n, src = s_src[w];
        X[(size_t)c * p.Nst + w] = row(src / k, src - (src / k) * k)[2 + c];
    }
}

// per group: elite deviations, weighted mean step, evolution paths, step size, new mean
// (cma_es.py:161-177).  One workgroup per group; threads stride the n coordinates.
// STAGE (small instances, one launch for the whole update): B, the k elite columns and the two intermediate vectors live in
// `stage` (n*n + k*n + 2n floats of LDS) -- every dependent access is then an LDS access instead of an L2 round trip (the
// chain of ~15 round trips is what the stand-alone kernel's 11-12 us at n = 30 consist of); same sums in the same order.
#ifdef BBMPC_KERNEL_DBG
__device__ long long g_paths_dbg[8];
#define PATHS_MARK(i) do { if (threadIdx.x == 0 && g == 0) g_paths_dbg[i] = (long long)wall_clock64(); } while (0)
#else
#define PATHS_MARK(i) do {} while (0)
#endif
template <bool STAGE>
__device__ __forceinline__ void cma_paths_body_t(const CmaArgs& p, int g, float* stage) {
    PATHS_MARK(7);
    // blockDim: any multiple of 64 up to 1024.  The loops keep several independent loads in flight and the row-wise
    // product runs one wave per row (coalesced), instead of one L2 latency per term of an n-term sum.
    __shared__ float red[16];
    __shared__ float s_norm;
    __shared__ int s_el[REFIT_THREADS];
    __shared__ float s_w[REFIT_THREADS];
    const int tid = threadIdx.x, n = p.n, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, NW = nthr >> 6;
    const size_t off = (size_t)g * n;
    const float* __restrict__ X = p.cand + off * p.Nst;
    float* __restrict__ Ye = p.Ye + (size_t)g * p.k * n;
    for (int i = tid; i < p.k; i += nthr) { s_el[i] = p.eidx[g * p.k + i]; s_w[i] = p.weights[i]; }
    float* lB = stage;
    float* xe = stage + (STAGE ? n * n : 0);
    float* lym = xe + (STAGE ? p.k * n : 0);
    float* lt2 = lym + (STAGE ? n : 0);
    if (STAGE)
        for (int i = tid; i < n * n; i += nthr) lB[i] = p.B[(size_t)g * n * n + i];
    __syncthreads();
    if (STAGE) {
        for (int i = tid; i < p.k * n; i += nthr) { const int e = i / n, c = i - e * n; xe[i] = X[(size_t)c * p.Nst + s_el[e]]; }
        __syncthreads();
    }
    PATHS_MARK(0);
    // x_diff, x_mean, y_mean, Ye
    if (STAGE) {
        // every thread one (elite, coordinate) element: clip, x_diff (left in xe), Ye; the weighted sums over the elites then
        // run from LDS in elite order -- n threads walking k elites each, a division and a global store per term, was 6 us of
        // the 12 us this body took at n = 30, k = 50
        for (int i = tid; i < p.k * n; i += nthr) {
            const int e = i / n, c = i - e * n;
            const float xd = clipf(xe[i], p.lo[c % p.U], p.hi[c % p.U]) - p.m[off + c];      // :161
            xe[i] = xd;
            Ye[(size_t)e * n + c] = xd / p.sigma[off + c];                                    // :180
        }
        __syncthreads();
        for (int c = tid; c < n; c += nthr) {
            const float sc = p.sigma[off + c];
            float xm = 0.0f;
            for (int e = 0; e < p.k; ++e) xm = xm + xe[e * n + c] * s_w[e];                      // :162
            p.xmean[off + c] = xm;
            p.ymean[off + c] = xm / sc;                                                       // :167
            lym[c] = xm / sc;
        }
    } else
    for (int c = tid; c < n; c += nthr) {
        const float mc = p.m[off + c], sc = p.sigma[off + c];
        const float lo_c = p.lo[c % p.U], hi_c = p.hi[c % p.U];     // solution layout [agents][H][U]
        float xm = 0.0f;
        for (int i0 = 0; i0 < p.k; i0 += 8) {
            float xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = (i0 + q < p.k) ? (STAGE ? xe[(i0 + q) * n + c] : X[(size_t)c * p.Nst + s_el[i0 + q]]) : 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = clipf(xv[q], lo_c, hi_c);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (i0 + q < p.k) {
                    const float xd = xv[q] - mc;                                 // :161
                    xm = xm + xd * s_w[i0 + q];                                  // :162
                    Ye[(size_t)(i0 + q) * n + c] = xd / sc;                      // :180
                }
            }
        }
        p.xmean[off + c] = xm;
        p.ymean[off + c] = xm / sc;                                        // :167
        if (STAGE) lym[c] = xm / sc;
    }
    __syncthreads();
    PATHS_MARK(1);
    // t1 = B^T y_mean ; t2 = t1 / diag(D)   (C^{-1/2} y = B D^{-1} B^T y, :168-169)
    float* t2 = STAGE ? lt2 : p.BD + (size_t)g * n * n;      // BD scratch is free after the sampling GEMM
    const float* __restrict__ B = STAGE ? lB : p.B + (size_t)g * n * n;
    const float* __restrict__ ym = STAGE ? lym : p.ymean + off;
    {
        // the sum over i is split into K-slices over thread groups of ceil(n / 64) waves (as many as the workgroup holds,
        // at most 4) and combined in slice order: one thread per column walked all n terms, eight L2 trips at a time
        __shared__ float s_part[4][512];
        const int npad = (n + 63) & ~63;
        const int ng = (n > 128 && n <= 512) ? max(1, min(4, nthr / npad)) : 1;     // (small n: the order the fused kernel shares)
        if (ng > 1) {
            const int gi = tid / npad, j = tid - gi * npad;
            const int Kc = (n + ng - 1) / ng, i_lo = gi * Kc, i_hi = min(n, i_lo + Kc);
            if (gi < ng && j < n) {
                float s = 0.0f;
                for (int i0 = i_lo; i0 < i_hi; i0 += 8) {
                    float bv[8], yv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = min(i0 + q, i_hi - 1);
                        bv[q] = B[(size_t)i * n + j];
                        yv[q] = ym[i];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (i0 + q < i_hi) s = fmaf(bv[q], yv[q], s);
                }
                s_part[gi][j] = s;
            }
            __syncthreads();
            for (int j2 = tid; j2 < n; j2 += nthr) {
                float s = s_part[0][j2];
                for (int q = 1; q < ng; ++q) s = s + s_part[q][j2];
                t2[j2] = s * (1.0f / p.Dd[off + j2]);
            }
        } else {
            for (int j = tid; j < n; j += nthr) {
                float s = 0.0f;
                for (int i0 = 0; i0 < n; i0 += 8) {
                    float bv[8], yv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = min(i0 + q, n - 1);
                        bv[q] = B[(size_t)i * n + j];
                        yv[q] = ym[i];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (i0 + q < n) s = fmaf(bv[q], yv[q], s);
                }
                t2[j] = s * (1.0f / p.Dd[off + j]);
            }
        }
    }
    __syncthreads();
    PATHS_MARK(2);
    const float cs = p.c.c_sigma, cc = p.c.cc;
    const float coef_s = sqrtf((cs * (2.0f - cs)) * p.c.mu_eff);
    const float coef_c = p.c.h_sigma * sqrtf((cc * (2.0f - cc)) * p.c.mu_eff);
    // s_i = sum_j B[i][j] t2[j]: one wave per row i, lanes along j (64-lane partial sums in j order, then the wave reduction)
    float part = 0.0f;
    for (int ib = wv; ib < n; ib += 4 * NW) {              // four rows per trip: their loads are in flight together
        float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = lane; j < n; j += 64) {
            const float tj = t2[j];
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] = fmaf(B[(size_t)min(ib + q * NW, n - 1) * n + j], tj, s4[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ib + q * NW;
            const float s = wave_sum(s4[q]);
            if (lane == 0 && i < n) {
                const float ps = (1.0f - cs) * p.p_sigma[off + i] + coef_s * s;    // :170-171
                p.p_sigma[off + i] = ps;
                part += ps * ps;
                p.p_C[off + i] = (1.0f - cc) * p.p_C[off + i] + coef_c * ym[i];    // :177
            }
        }
    }
    PATHS_MARK(3);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (tid == 0) {
        float s = 0.0f;
        for (int w = 0; w < NW; ++w) s += red[w];
        s_norm = sqrtf(s);
    }
    __syncthreads();
    const float fac = expf((cs / p.c.d_sigma) * (s_norm / p.c.e_norm - 1.0f));   // :172-173
    for (int c = tid; c < n; c += nthr) {
        p.sigma[off + c] = p.sigma[off + c] * fac;
        p.m[off + c] = p.m[off + c] + p.xmean[off + c];                    // :163
    }
#ifdef BBMPC_KERNEL_DBG
    if (threadIdx.x == 0 && g == 0 && p.iter == 2) {
        const long long t1 = (long long)wall_clock64();
        printf("[paths] stage %lld  Ye/xmean %lld  t2 %lld  rows %lld  norm+update %lld (10 ns)\n", g_paths_dbg[0] - g_paths_dbg[7], g_paths_dbg[1] - g_paths_dbg[0], g_paths_dbg[2] - g_paths_dbg[1], g_paths_dbg[3] - g_paths_dbg[2], t1 - g_paths_dbg[3]);
    }
#endif
}
__device__ __forceinline__ void cma_paths_body(const CmaArgs& p, int g) { cma_paths_body_t<false>(p, g, nullptr); }
static __global__ __launch_bounds__(1024) void k_cma_paths(CmaArgs p) { cma_paths_body(p, blockIdx.x); }
// selection and path update of an instance in one launch (both are one workgroup per instance); dynamic LDS: the selection's
static __global__ __launch_bounds__(1024) void k_cma_select_paths(CmaArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cma_select_body(p, blockIdx.x, smem, nullptr);
    __syncthreads();
    cma_paths_body(p, blockIdx.x);
}

// C = (1-c1-cmu) C + c1 pC pC^T + cmu sum_i w_i y_i y_i^T on the upper triangle, mirrored (cma_es.py:179-190)
// grid (ceil(n/16), ceil(n/16), G), block (16,16)
static __global__ void k_cma_cov(CmaArgs p) {
    const int g = blockIdx.z, n = p.n;
    const int c = blockIdx.x * 16 + threadIdx.x, r = blockIdx.y * 16 + threadIdx.y;
    if (r >= n || c >= n || r > c) return;
    const size_t off = (size_t)g * n;
    const float* Ye = p.Ye + (size_t)g * p.k * n;
    float ys = 0.0f;
    for (int i = 0; i < p.k; ++i) ys = fmaf(Ye[(size_t)i * n + r] * Ye[(size_t)i * n + c], p.weights[i], ys);
    float* C = p.C + off * n;
    const float v = ((1.0f - p.c.c1) - p.c.c_mu) * C[(size_t)r * n + c] + (p.c.c1 * p.p_C[off + r]) * p.p_C[off + c] +      /* (c1 * p_C) * p_C^T as cma_es.py:183 evaluates it */
                    p.c.c_mu * ys;
    C[(size_t)r * n + c] = v;
    C[(size_t)c * n + r] = v;
}

// SVD of the symmetric covariance, s,U,_ = tf.linalg.svd(C)  (cma_es.py:195): B = U, D = diag(sqrt(s)).
//
// One-sided (Hestenes) Jacobi: rotate pairs of columns of A (= C) until they are mutually orthogonal;
// then A V = U diag(s), so s_j = |a_j| and u_j = a_j / s_j.  Only column-pair operations are needed, and
// the n/2 pairs of a round-robin round touch disjoint columns, so a round is embarrassingly parallel:
// one workgroup per CMA-ES instance, one wave per pair, lanes along the column.  Columns are kept as
// ROWS of At (C is symmetric, so At = C to start with) to make every access contiguous.  Singular values
// are sorted descending at the end, as TF returns them.
// scratch: At [G][n][n], norms [G*n], perm (int) [G*n]
static __global__ __launch_bounds__(REFIT_THREADS) void k_cma_svd(CmaArgs p, float* At_all, float* norms_all, int* perm_all,
                                                           int max_sweeps) {
    __shared__ int s_rotated;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n;
    constexpr int NW = REFIT_THREADS / 64;
    const size_t nn = (size_t)n * n;
    float* At = At_all + (size_t)g * nn;
    const float* C = p.C + (size_t)g * nn;
    for (size_t i = tid; i < nn; i += REFIT_THREADS) At[i] = C[i];
    __syncthreads();
    const int m = (n + 1) & ~1;                 // players of the round-robin (one dummy when n is odd)
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) s_rotated = 0;
        __syncthreads();
        for (int r = 0; r < m - 1; ++r) {
            for (int i = wv; i < m / 2; i += NW) {
                int pa, pb;
                if (i == 0) { pa = m - 1; pb = r; }
                else { pa = (r + i) % (m - 1); pb = (r - i + (m - 1)) % (m - 1); }
                if (pa >= n || pb >= n) continue;                        // dummy player sits out
                float* x = At + (size_t)pa * n;
                float* y = At + (size_t)pb * n;
                float al = 0.0f, be = 0.0f, ga = 0.0f;
                for (int e = lane; e < n; e += 64) {
                    const float xv = x[e], yv = y[e];
                    al = fmaf(xv, xv, al); be = fmaf(yv, yv, be); ga = fmaf(xv, yv, ga);
                }
                al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
                if (fabsf(ga) <= 2e-6f * sqrtf(al * be) || ga == 0.0f) continue;   // fp32 dot-product noise floor
                const float zeta = (be - al) / (2.0f * ga);
                const float t = copysignf(1.0f, zeta) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                const float cs = 1.0f / sqrtf(1.0f + t * t), sn = cs * t;
                for (int e = lane; e < n; e += 64) {
                    const float xv = x[e], yv = y[e];
                    x[e] = fmaf(cs, xv, -(sn * yv));
                    y[e] = fmaf(sn, xv, cs * yv);
                }
                if (lane == 0) s_rotated = 1;
            }
            __syncthreads();
        }
        if (!s_rotated) break;
    }
    // singular values = column norms; order them descending (ties -> lower index)
    float* norms = norms_all + (size_t)g * n;
    int* perm = perm_all + (size_t)g * n;
    for (int j = wv; j < n; j += NW) {
        float al = 0.0f;
        for (int e = lane; e < n; e += 64) { const float v = At[(size_t)j * n + e]; al = fmaf(v, v, al); }
        al = wave_sum(al);
        if (lane == 0) norms[j] = sqrtf(al);
    }
    __syncthreads();
    for (int j = tid; j < n; j += REFIT_THREADS) {
        const float nj = norms[j];
        int rank = 0;
        for (int o = 0; o < n; ++o) rank += (norms[o] > nj || (norms[o] == nj && o < j)) ? 1 : 0;
        perm[rank] = j;
    }
    __syncthreads();
    float* B = p.B + (size_t)g * nn;
    for (size_t i = tid; i < nn; i += REFIT_THREADS) {
        const int r = (int)(i / n), c = (int)(i % n);
        const int src = perm[c];
        const float sv = norms[src];
        B[i] = (sv > 0.0f) ? At[(size_t)src * n + r] / sv : ((r == c) ? 1.0f : 0.0f);
    }
    for (int c = tid; c < n; c += REFIT_THREADS) p.Dd[(size_t)g * n + c] = sqrtf(norms[perm[c]]);   // D = diag(sqrt(s))
}

// ---- Multi-workgroup, warm-started version of the Jacobi SVD above (n = H*U = 300 at BASELINE config 5: one
// workgroup per instance needed 50 ms).
//  * Warm start: C changes by a small rank-(k+1) update per iteration, so the previous eigenvectors B0 nearly
//    diagonalise it.  Run the one-sided Jacobi on A = C*B0 (columns ~ lambda_j b_j, already almost orthogonal):
//    A V = U S with C = U S (B0 V)^T, i.e. the same left singular vectors / values, reached in 2-3 sweeps instead of 8+.
//  * 16 waves per instance, a pair's two columns held in registers between the dot products and the rotation (one
//    global round trip per pair instead of two).
//  * The kernel can also spread an instance over WPG workgroups with a global-atomic barrier per round.  Measured:
//    42 us per round -- an agent-scope release/acquire on this multi-XCD part writes back / invalidates L2 -- against
//    ~10 us for the whole round in one workgroup, so WPG = 1 is what the engine launches.
// k_cma_warm: At[j][:] = C * B0[:, j] (C symmetric => coalesced along the column);  k_cma_svd_rounds: the sweeps;
// k_cma_svd_finish: norms, descending order, B and D.
static __global__ __launch_bounds__(256) void k_cma_warm(CmaArgs p, float* At_all) {
    // At[j][e] = sum_k C[k][e] * B[k][j]: 32x32 output tile per workgroup, K staged through LDS in slabs of 32,
    // each thread owns a 2x2 micro tile.  grid (ceil(n/32), ceil(n/32), G), block 256 = 16 x 16
    __shared__ float sC[32][33], sB[32][33];
    const int g = blockIdx.z, n = p.n;
    const int e0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const float* C = p.C + (size_t)g * n * n;
    const float* B = p.B + (size_t)g * n * n;
    float acc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    for (int k0 = 0; k0 < n; k0 += 32) {
        for (int i = threadIdx.x; i < 32 * 32; i += 256) {
            const int kk = i >> 5, c = i & 31;
            sC[kk][c] = (k0 + kk < n && e0 + c < n) ? C[(size_t)(k0 + kk) * n + e0 + c] : 0.0f;
            sB[kk][c] = (k0 + kk < n && j0 + c < n) ? B[(size_t)(k0 + kk) * n + j0 + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const float c0 = sC[kk][tx], c1 = sC[kk][tx + 16];
            const float b0 = sB[kk][ty], b1 = sB[kk][ty + 16];
            acc[0][0] = fmaf(c0, b0, acc[0][0]); acc[0][1] = fmaf(c1, b0, acc[0][1]);
            acc[1][0] = fmaf(c0, b1, acc[1][0]); acc[1][1] = fmaf(c1, b1, acc[1][1]);
        }
        __syncthreads();
    }
    float* At = At_all + (size_t)g * n * n;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ee = 0; ee < 2; ++ee) {
            const int j = j0 + ty + 16 * jj, e = e0 + tx + 16 * ee;
            if (j < n && e < n) At[(size_t)j * n + e] = acc[jj][ee];
        }
}

// The same product on the matrix cores for large n (n % 4 == 0), as k_cma_gemm_y_mfma: for a fixed k both B[k][j0 ..] and
// C[k][e0 ..] are contiguous, i.e. the A / B fragments of v_mfma_f32_16x16x4_f32 load straight from L2.  Workgroup tile
// 64 (j) x 32 (e), wave w owns rows 16w .. 16w+15.  grid (ceil(n/32), ceil(n/64), G), block 256.  32 -> ~8 us at n = 300.
// `need` (optional, [G][8] words): the instance runs only when need[8 g] != 0 -- the direct eigensolver (kernels_eigh.hpp) leaves
// that flag set for the instances whose result it did not accept; the same parameter on the kernels below.
static __global__ __launch_bounds__(256) void k_cma_warm_mfma(CmaArgs p, float* At_all, const unsigned* need = nullptr) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int g = blockIdx.z, n = p.n;
    if (need && need[(size_t)g * 8] == 0u) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = blockIdx.y * 64 + wave * 16, e0 = blockIdx.x * 32;
    if (j0 >= n) return;
    const int lm = lane & 15, lk = lane >> 4;
    const float* Bm = p.B + (size_t)g * n * n + min(j0 + lm, n - 1);                 // + k * n
    const float* Cm = p.C + (size_t)g * n * n;
    const int ec0 = min(e0 + lm, n - 1), ec1 = min(e0 + 16 + lm, n - 1);
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
    for (int k0 = 0; k0 < n; k0 += 4) {
        const size_t kr = (size_t)(k0 + lk) * n;
        const float a = Bm[kr];
        const float b0 = Cm[kr + ec0], b1 = Cm[kr + ec1];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
    }
    float* At = At_all + (size_t)g * n * n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + 4 * lk + r;
        if (j >= n) continue;
        if (e0 + lm < n) At[(size_t)j * n + e0 + lm] = acc0[r];
        if (e0 + 16 + lm < n) At[(size_t)j * n + e0 + 16 + lm] = acc1[r];
    }
}

__device__ __forceinline__ void cma_instance_barrier(unsigned* ctr, unsigned target) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ctr, 1u);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

// Cache-bypassing accessors for data that workgroups on different XCDs exchange inside one kernel: relaxed atomics at
// agent scope compile to plain loads/stores with the sc1 bit (no fence).  With them the instance barrier does not
// need __threadfence(), whose agent-scope release/acquire walks the XCD's L2 (measured ~60 us per barrier).
__device__ __forceinline__ float coh_load(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coh_store(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all data exchanged through coh_load / coh_store: the barrier only has to wait for this workgroup's stores
__device__ __forceinline__ void cma_instance_barrier_light(unsigned* ctr, unsigned target) {
    __builtin_amdgcn_s_waitcnt(0);                       // vmcnt/lgkmcnt = 0: my stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

constexpr int CMA_SYNC_WORDS = 512;
constexpr int CMA_SYNC_XCC_MASK = 24;
constexpr int CMA_SYNC_ROTATIONS = 480;               // k_cma_svd_block: [480 + sweep] column pairs rotated in that sweep (statistics)                  // k_cma_svd_block: bit x = some workgroup of the instance runs on XCD x
// sync: [G][CMA_SYNC_WORDS] unsigned: [0] barrier counter, [1 + sweep] "some pair rotated in this sweep"; the block kernel keeps
// its block-pair bookkeeping in words 32.. (2 * NB + NB * NB of them)
static __global__ __launch_bounds__(1024) void k_cma_svd_rounds(CmaArgs p, float* At_all, unsigned* sync_all, int max_sweeps, int lds_floats) {
    const int g = blockIdx.y, WPG = gridDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n;
    const int NW = blockDim.x >> 6;
    // rotation threshold on the cosine between two columns: the fp32 dot-product noise floor grows with n
    const float tol = fminf(fmaxf(3.0e-8f * (float)n, 2.0e-6f), 1.0e-5f);
    float* At_g = At_all + (size_t)g * n * n;
    // small instances (n*n floats fit the dynamic LDS the launch provides, WPG == 1): the whole matrix lives in LDS
    // for the sweeps -- a round then costs LDS latency instead of L2 latency (n = 30: 437 -> ~100 us per iteration)
    extern __shared__ __attribute__((aligned(16))) float at_lds[];
    const bool in_lds = lds_floats >= n * n && WPG == 1;
    float* At = in_lds ? at_lds : At_g;
    if (in_lds) {
        for (int i = tid; i < n * n; i += blockDim.x) at_lds[i] = At_g[i];
        __syncthreads();
    }
    unsigned* sync = sync_all + (size_t)g * CMA_SYNC_WORDS;
    const int gw = blockIdx.x * NW + wv, nwaves = WPG * NW;
    const int m = (n + 1) & ~1;                 // players of the round-robin (one dummy when n is odd)
    unsigned bar = 0;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        for (int r = 0; r < m - 1; ++r) {
            // two pairs per wave in flight: the loads of the second pair overlap the reductions of the first
            for (int i0 = gw; i0 < m / 2; i0 += 2 * nwaves) {
                float xv[2][8], yv[2][8];
                float* xp[2];
                float* yp[2];
                bool act[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = i0 + q * nwaves;
                    int pa = 0, pb = 0;
                    if (i == 0) { pa = m - 1; pb = r; }
                    else if (i < m / 2) { pa = (r + i) % (m - 1); pb = (r - i + (m - 1)) % (m - 1); }
                    act[q] = i < m / 2 && pa < n && pb < n;              // dummy player sits out
                    xp[q] = At + (size_t)(act[q] ? pa : 0) * n;
                    yp[q] = At + (size_t)(act[q] ? pb : 0) * n;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int e = lane + 64 * c;
                        xv[q][c] = (act[q] && e < n) ? xp[q][e] : 0.0f;
                        yv[q][c] = (act[q] && e < n) ? yp[q][e] : 0.0f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float al = 0.0f, be = 0.0f, ga = 0.0f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        al = fmaf(xv[q][c], xv[q][c], al); be = fmaf(yv[q][c], yv[q][c], be); ga = fmaf(xv[q][c], yv[q][c], ga);
                    }
                    al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
                    if (!act[q] || fabsf(ga) <= tol * sqrtf(al * be) || ga == 0.0f) continue;
                    const float zeta = (be - al) / (2.0f * ga);
                    const float t = copysignf(1.0f, zeta) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                    const float cs = 1.0f / sqrtf(1.0f + t * t), sn = cs * t;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int e = lane + 64 * c;
                        if (e < n) {
                            xp[q][e] = fmaf(cs, xv[q][c], -(sn * yv[q][c]));
                            yp[q][e] = fmaf(sn, xv[q][c], cs * yv[q][c]);
                        }
                    }
#ifdef BBMPC_KERNEL_DBG
                    if (lane == 0) atomicAdd(&sync[1 + sweep], 1u);
#else
                    if (lane == 0) sync[1 + sweep] = 1u;
#endif
                }
            }
            ++bar;
            if (WPG > 1) cma_instance_barrier(sync, bar * (unsigned)WPG);
            else __syncthreads();
        }
        if (WPG == 1) __syncthreads();
        if (__hip_atomic_load(sync + 1 + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
#ifdef BBMPC_KERNEL_DBG
        if (g == 0 && blockIdx.x == 0 && tid == 0) printf("[svd] sweep %d done at %lld rotations %u\n", sweep, (long long)wall_clock64(), sync[1 + sweep]);
#endif
    }
    if (in_lds) {
        __syncthreads();
        for (int i = tid; i < n * n; i += blockDim.x) At_g[i] = at_lds[i];
    }
}

__device__ __forceinline__ void cma_svd_finish_body(const CmaArgs& p, int g, const float* At_all, float* norms_all, int* perm_all) {
    // column norms = singular values; ranks by counting over an LDS copy of the norms (n <= 2048); B, D.  blockDim any multiple of 64
    __shared__ float s_norm[2048];
    __shared__ int s_perm[2048];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n, nthr = blockDim.x, NW = nthr >> 6;
    const size_t nn = (size_t)n * n;
    const float* At = At_all + (size_t)g * nn;
    float* norms = norms_all + (size_t)g * n;
    int* perm = perm_all + (size_t)g * n;
    for (int j = wv; j < n; j += NW) {
        float al = 0.0f;
        for (int e = lane; e < n; e += 64) { const float v = At[(size_t)j * n + e]; al = fmaf(v, v, al); }
        al = wave_sum(al);
        if (lane == 0) { const float nj = sqrtf(al); s_norm[j] = nj; norms[j] = nj; }
    }
    __syncthreads();
    for (int j = tid; j < n; j += nthr) {
        const float nj = s_norm[j];
        int rank = 0;
        for (int o = 0; o < n; ++o) { const float no = s_norm[o]; rank += (no > nj || (no == nj && o < j)) ? 1 : 0; }
        s_perm[rank] = j;
        perm[rank] = j;
    }
    __syncthreads();
    float* B = p.B + (size_t)g * nn;
    for (size_t i = tid; i < nn; i += nthr) {
        const int r = (int)(i / n), c = (int)(i % n);
        const int src = s_perm[c];
        const float sv = s_norm[src];
        B[i] = (sv > 0.0f) ? At[(size_t)src * n + r] / sv : ((r == c) ? 1.0f : 0.0f);
    }
    for (int c = tid; c < n; c += nthr) p.Dd[(size_t)g * n + c] = sqrtf(s_norm[s_perm[c]]);   // D = diag(sqrt(s))
}
static __global__ __launch_bounds__(1024) void k_cma_svd_finish(CmaArgs p, const float* At_all, float* norms_all, int* perm_all) {
    cma_svd_finish_body(p, blockIdx.x, At_all, norms_all, perm_all);
}

// The same finish for large n as two launches: norms / ranks / D per instance, then B in 32 x 32 tiles over many
// workgroups.  (One workgroup per instance walked its n^2 elements with an integer division each and read At along the
// strided direction: 79 us at n = 300, a tenth of which is left.)  Same arithmetic, same bits.
static __global__ __launch_bounds__(1024) void k_cma_svd_norms(CmaArgs p, const float* At_all, float* norms_all, int* perm_all, const unsigned* need = nullptr) {
    __shared__ float s_norm[2048];
    if (need && need[(size_t)blockIdx.x * 8] == 0u) return;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n, nthr = blockDim.x, NW = nthr >> 6;
    const float* At = At_all + (size_t)g * n * n;
    float* norms = norms_all + (size_t)g * n;
    int* perm = perm_all + (size_t)g * n;
    for (int jb = wv; jb < n; jb += 4 * NW) {              // four columns per trip (same sums, their loads in flight together)
        float a4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int e = lane; e < n; e += 64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float v = At[(size_t)min(jb + q * NW, n - 1) * n + e]; a4[q] = fmaf(v, v, a4[q]); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = jb + q * NW;
            const float al = wave_sum(a4[q]);
            if (lane == 0 && j < n) { const float nj = sqrtf(al); s_norm[j] = nj; norms[j] = nj; }
        }
    }
    __syncthreads();
    for (int j = tid; j < n; j += nthr) {
        const float nj = s_norm[j];
        int rank = 0;
        for (int o = 0; o < n; ++o) { const float no = s_norm[o]; rank += (no > nj || (no == nj && o < j)) ? 1 : 0; }
        perm[rank] = j;
        p.Dd[(size_t)g * n + rank] = sqrtf(nj);                      // D = diag(sqrt(s)), descending
    }
}
static __global__ __launch_bounds__(256) void k_cma_svd_build_b(CmaArgs p, const float* At_all, const float* norms_all, const int* perm_all, const unsigned* need = nullptr) {
    __shared__ float tile[32][33];
    if (need && need[(size_t)blockIdx.z * 8] == 0u) return;
    __shared__ int s_src[32];
    __shared__ float s_sv[32];
    const int g = blockIdx.z, n = p.n, tx = threadIdx.x, ty = threadIdx.y;       // block (32, 8)
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const float* At = At_all + (size_t)g * n * n;
    if (ty == 0) {
        const int c = c0 + tx;
        const int src = c < n ? perm_all[(size_t)g * n + c] : 0;
        s_src[tx] = src;
        s_sv[tx] = c < n ? norms_all[(size_t)g * n + src] : 0.0f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {                                // column c0 + k of B = row src of At, along r
        const int r = r0 + tx;
        tile[k][tx] = (c0 + k < n && r < n) ? At[(size_t)s_src[k] * n + r] : 0.0f;
    }
    __syncthreads();
    float* B = p.B + (size_t)g * n * n;
    for (int k = ty; k < 32; k += 8) {                                // row r0 + k of B, along c
        const int r = r0 + k, c = c0 + tx;
        if (r < n && c < n) {
            const float sv = s_sv[tx];
            B[(size_t)r * n + c] = (sv > 0.0f) ? tile[tx][k] / sv : ((r == c) ? 1.0f : 0.0f);
        }
    }
}

// 16-byte write-through (sc0 sc1) accesses for blocks that travel between workgroups: the 4-byte scalar form costs ~6x
// per byte on the fabric (MI355X_MICROARCH.md, inter-workgroup visibility) -- the load phase was a quarter of the block
// kernel's run time with it.
typedef float cma_f32x4 __attribute__((ext_vector_type(4)));
// four 16-byte loads and the wait for them in ONE asm statement: the compiler cannot see that an asm load's result
// register is still in flight, so nothing (a copy, a select) may sit between the loads and the s_waitcnt
__device__ __forceinline__ void coh_load16x4(const float* p0, const float* p1, const float* p2, const float* p3, cma_f32x4& v0,
                                             cma_f32x4& v1, cma_f32x4& v2, cma_f32x4& v3) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}
__device__ __forceinline__ void coh_store16(float* p, cma_f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
// the same store WITHOUT the write-through bits: the line stays (dirty) in this XCD's L2, where the `sc0 sc1` loads of
// workgroups on the SAME XCD find it -- a `sc0 sc1` store drops the line, and the next reader pays the fabric round
// trip (measured in this kernel: 29 us instead of 3.5 us per block-pair load).  Only valid when every reader is on
// this XCD (k_cma_svd_block checks that at run time).
__device__ __forceinline__ void l2_store16(float* p, cma_f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// ---- Block Jacobi: the columns are cut into 8 blocks; 4 workgroups per instance each hold one PAIR of blocks in
// LDS and orthogonalise every column pair inside it without touching global memory; the block pairs follow a
// round-robin tournament (7 block rounds per sweep), so an instance needs 7 instance-wide barriers per sweep instead
// of n-1, and the one-CU memory path that bounds k_cma_svd_rounds (67 GB/s, 3.2 ms per sweep at n = 300) is out of
// the picture.  Same rotation, threshold and convergence rule as above.
// LDS: 2*bs columns at a pitch of 64 * ceil(n / 64) floats.  sync: [G][CMA_SYNC_WORDS] as above.
// Placement: a 1-D grid of 8 * WPG * ceil(G / 8) workgroups.  Workgroup `id` is dispatched to XCD id % 8 (observed, not
// promised: MI355X_MICROARCH.md, workgroup dispatch), so instance g = (id % 8) + 8 * (slot / WPG), member slot % WPG with
// slot = id / 8 puts an instance's workgroups behind ONE L2.  Every workgroup reads its XCC id and the instance agrees
// (first barrier) whether that held; only then do the blocks travel as L2-resident lines, otherwise as write-through
// stores as before -- placement buys speed, never correctness.
// NC = ceil(n / 64): the resident columns sit in LDS at a pitch of ld = 64 * NC floats, zero beyond n, so that every
// per-element loop has a compile-time trip count and no bounds test (at n = 300 the tests, masks and branches were two
// thirds of the instructions of a cross round).
template <int NC, int NB>
__global__ __launch_bounds__(1024) void k_cma_svd_block(CmaArgs p, float* At_all, unsigned* sync_all, int max_sweeps, const unsigned* need) {
    extern __shared__ __attribute__((aligned(16))) float cols[];
    // NB blocks of columns, NB / 2 workgroups per instance (8 / 4 or 16 / 8)
    constexpr int WPG = NB / 2;
    const int slot = blockIdx.x >> 3;
    const int g = (blockIdx.x & 7) + 8 * (slot / WPG), wg = slot % WPG;
    if (g >= p.G) return;
    if (need && need[(size_t)g * 8] == 0u) return;          // (every workgroup of the instance leaves: no barrier is left waiting)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n;
    constexpr int ld = 64 * NC;
    const int NW = blockDim.x >> 6;
    const int bs = (n + NB - 1) / NB;                       // columns per block (the last block may be short)
    float* At = At_all + (size_t)g * n * n;
    unsigned* sync = sync_all + (size_t)g * CMA_SYNC_WORDS;
    // Block-pair bookkeeping (all zero at launch): a pair of blocks that was checked without a single rotation stays
    // clean until one of its blocks changes, so late sweeps -- and the final verification sweep entirely -- skip it.
    //   ver[b] + 1 = version of block b; iseen[b] = version at which block b's inner pairs were last found clean;
    //   pseen[min*NB+max] = (version of the lower block << 16 | version of the higher block) at the last clean check
    unsigned* ver = sync + 32;
    unsigned* iseen = ver + NB;
    unsigned* pseen = iseen + NB;                   // [NB][NB]
    static_assert(32 + 2 * NB + NB * NB <= CMA_SYNC_ROTATIONS && CMA_SYNC_ROTATIONS + 16 <= CMA_SYNC_WORDS, "sync words");
    __shared__ int s_skip[3], s_rotf[3];          // [0] cross pairs, [1] inner pairs of block x, [2] of block y
    __shared__ int s_same_xcd;
    __shared__ float s_ynrm[64], s_ysc[64];     // tracked |y_j|^2 and scale of the resident y-columns (bs <= 64)
    const float tol = fminf(fmaxf(3.0e-8f * (float)n, 2.0e-6f), 1.0e-5f);
    unsigned bar = 0;
    for (int i = tid; i < 2 * bs * ld; i += blockDim.x) cols[i] = 0.0f;     // the padding beyond n stays zero for good
    // which XCD am I on?  (s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), bits 3:0)
    if (tid == 0) __hip_atomic_fetch_or(sync + CMA_SYNC_XCC_MASK, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
    ++bar;
    cma_instance_barrier_light(sync, bar * (unsigned)WPG);
    if (tid == 0) {
        const unsigned mask = __hip_atomic_load(sync + CMA_SYNC_XCC_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_same_xcd = (mask & (mask - 1u)) == 0u;
    }
    __syncthreads();
    const bool same_xcd = s_same_xcd != 0;
#ifdef BBMPC_KERNEL_DBG
    if (tid == 0 && wg == 0) printf("[svdb] instance %d block %d: XCC mask 0x%x same_xcd %d\n", g, (int)blockIdx.x, sync[CMA_SYNC_XCC_MASK], (int)same_xcd);
#endif

    // one column pair per 16-lane DPP row (LDS columns ia, ib), four pairs per wave: 16-byte LDS accesses, 4-step row
    // reductions, hardware rcp / rsq (+ one Newton step on the cosine) for the rotation scalars.  (Until round 3 a
    // whole wave took one pair with scalar LDS accesses and IEEE divisions; with 19 pairs on 16 waves a round of the
    // inner tournament then cost two pair times.)
    const int sub = lane & 15, prow = wv * 4 + (lane >> 4);
    auto rotate16 = [&](int ia, int ib, bool act) -> bool {
        float* x = cols + (size_t)ia * ld;
        float* y = cols + (size_t)ib * ld;
        float4 xv[NC], yv[NC];
        float al = 0.0f, be = 0.0f, ga = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e = 4 * (sub + 16 * c);
            xv[c] = *reinterpret_cast<const float4*>(x + e);
            yv[c] = *reinterpret_cast<const float4*>(y + e);
            al = fmaf(xv[c].x, xv[c].x, al); al = fmaf(xv[c].y, xv[c].y, al); al = fmaf(xv[c].z, xv[c].z, al); al = fmaf(xv[c].w, xv[c].w, al);
            be = fmaf(yv[c].x, yv[c].x, be); be = fmaf(yv[c].y, yv[c].y, be); be = fmaf(yv[c].z, yv[c].z, be); be = fmaf(yv[c].w, yv[c].w, be);
            ga = fmaf(xv[c].x, yv[c].x, ga); ga = fmaf(xv[c].y, yv[c].y, ga); ga = fmaf(xv[c].z, yv[c].z, ga); ga = fmaf(xv[c].w, yv[c].w, ga);
        }
        al = row16_sum(al); be = row16_sum(be); ga = row16_sum(ga);
        const bool rot = act && ga != 0.0f && (ga * ga > (tol * tol) * (al * be));
        if (rot) {
            const float zeta = (be - al) * __builtin_amdgcn_rcpf(2.0f * ga);
            const float t = copysignf(__builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.0f))), zeta);
            const float w = fmaf(t, t, 1.0f);
            float cs = __builtin_amdgcn_rsqf(w);
            cs = cs * fmaf(-0.5f * w * cs, cs, 1.5f);
            const float sn = cs * t;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int e = 4 * (sub + 16 * c);
                float4 xn, yn;
                xn.x = fmaf(cs, xv[c].x, -(sn * yv[c].x)); yn.x = fmaf(sn, xv[c].x, cs * yv[c].x);
                xn.y = fmaf(cs, xv[c].y, -(sn * yv[c].y)); yn.y = fmaf(sn, xv[c].y, cs * yv[c].y);
                xn.z = fmaf(cs, xv[c].z, -(sn * yv[c].z)); yn.z = fmaf(sn, xv[c].z, cs * yv[c].z);
                xn.w = fmaf(cs, xv[c].w, -(sn * yv[c].w)); yn.w = fmaf(sn, xv[c].w, cs * yv[c].w);
                *reinterpret_cast<float4*>(x + e) = xn;          // (the zero padding stays zero)
                *reinterpret_cast<float4*>(y + e) = yn;
            }
        }
        return rot;
    };

#ifdef BBMPC_KERNEL_DBG
    long long tacc[5] = {0, 0, 0, 0, 0}, tl = (long long)wall_clock64();
#define SVDB_MARK(i) do { const long long now_ = (long long)wall_clock64(); tacc[i] += now_ - tl; tl = now_; } while (0)
#else
#define SVDB_MARK(i) do {} while (0)
#endif
    unsigned nrot = 0;                                  // rotations this 16-lane row performed in the current sweep (lane 0 of the row counts)
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        bool rotated = false;
        for (int R = 0; R < NB - 1; ++R) {
            // tournament: workgroup 0 plays (NB-1, R), workgroup i plays ((R+i) % (NB-1), (R-i) mod (NB-1))
            int bx, by;
            if (wg == 0) { bx = NB - 1; by = R; }
            else { bx = (R + wg) % (NB - 1); by = (R - wg + (NB - 1)) % (NB - 1); }
            const int x0 = bx * bs, y0 = by * bs;
            const int nx = max(0, min(bs, n - x0)), ny = max(0, min(bs, n - y0));
            const int blo = min(bx, by), bhi = max(bx, by);
            unsigned vx = 0, vy = 0;
            if (tid == 0) {
                vx = __hip_atomic_load(ver + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                vy = __hip_atomic_load(ver + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                const unsigned packed = ((bx < by ? vx : vy) << 16) | (bx < by ? vy : vx);
                s_skip[0] = __hip_atomic_load(pseen + blo * NB + bhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == packed;
                s_skip[1] = R != 0 || __hip_atomic_load(iseen + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == vx;
                s_skip[2] = R != 0 || __hip_atomic_load(iseen + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == vy;
                s_rotf[0] = s_rotf[1] = s_rotf[2] = 0;
            }
            __syncthreads();
            const bool skip_cross = s_skip[0] != 0, skip_ix = s_skip[1] != 0, skip_iy = s_skip[2] != 0;
            if (!(skip_cross && skip_ix && skip_iy)) {
            // ---- load the two blocks (columns are rows of At: contiguous)
            // whole columns per wave (no index divisions), all of a wave's loads issued before the first LDS write
            {
                const int n4 = n >> 2;                                   // n % 4 == 0 on this path
                for (int c0 = wv; c0 < nx + ny; c0 += 4 * NW) {
                    const float* src[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = min(c0 + q * NW, nx + ny - 1);         // clamped: a duplicate load instead of a branch
                        src[q] = At + (size_t)(c < nx ? x0 + c : y0 + (c - nx)) * n;
                    }
                    for (int h0 = 0; h0 < n4; h0 += 64) {
                        const int e4 = min(h0 + lane, n4 - 1);
                        cma_f32x4 v0, v1, v2, v3;
                        coh_load16x4(src[0] + 4 * e4, src[1] + 4 * e4, src[2] + 4 * e4, src[3] + 4 * e4, v0, v1, v2, v3);
                        if (h0 + lane < n4) {
                            if (c0 < nx + ny) *reinterpret_cast<cma_f32x4*>(cols + (size_t)c0 * ld + 4 * e4) = v0;
                            if (c0 + NW < nx + ny) *reinterpret_cast<cma_f32x4*>(cols + (size_t)(c0 + NW) * ld + 4 * e4) = v1;
                            if (c0 + 2 * NW < nx + ny) *reinterpret_cast<cma_f32x4*>(cols + (size_t)(c0 + 2 * NW) * ld + 4 * e4) = v2;
                            if (c0 + 3 * NW < nx + ny) *reinterpret_cast<cma_f32x4*>(cols + (size_t)(c0 + 3 * NW) * ld + 4 * e4) = v3;
                        }
                    }
                }
            }
            __syncthreads();
            SVDB_MARK(0);
            // ---- pairs inside each block, once per sweep (block round 0 has every block in some workgroup)
            if (R == 0) {
                for (int blk = 0; blk < 2; ++blk) {
                    if (blk ? skip_iy : skip_ix) continue;
                    const int base = blk ? nx : 0, cnt = blk ? ny : nx;
                    const int m = (cnt + 1) & ~1;
                    bool rot_here = false;
                    for (int r = 0; r < m - 1; ++r) {
                        for (int i0 = 0; i0 < m / 2; i0 += 4 * NW) {       // one trip for bs <= 8 * NW columns
                            const int i = i0 + prow;
                            if (i0 + wv * 4 < m / 2) {
                                int pa, pb;
                                if (i == 0) { pa = m - 1; pb = r; }
                                else { pa = (r + i) % (m - 1); pb = (r - i + (m - 1)) % (m - 1); }
                                const bool act = i < m / 2 && pa < cnt && pb < cnt;
                                const bool rt = rotate16(base + (act ? pa : 0), base + (act ? pb : 0), act);
                                rot_here |= rt;
                                nrot += (rt && sub == 0) ? 1u : 0u;
                            }
                        }
                        __syncthreads();
                    }
                    if (rot_here && sub == 0) s_rotf[1 + blk] = 1;
                    rotated |= rot_here;
                }
            }
            SVDB_MARK(1);
            // ---- pairs across the two blocks: round r pairs x-column i with y-column (i + r) % mm: disjoint.
            // One 16-lane DPP row per pair (four pairs per wave instruction stream, reductions = 4 DPP steps); a row
            // keeps its x-column in registers for all mm rounds, only the y-columns travel through LDS, as 16-byte
            // accesses (n % 4 == 0).  Waves whose rows hold no x-column only take part in the barriers, and the
            // rotation scalars use the hardware rcp / rsq (+ one Newton step where orthogonality depends on it):
            // the round is instruction-issue bound, IEEE divide / sqrt expansions were half of it.
            const int mm = max(nx, ny);
            if (!skip_cross) {
                bool rot_cross = false;
                constexpr int EC = NC;                               // float4 chunks per lane
                const int row = prow;
                const bool has_x = row < nx, wave_on = wv * 4 < nx;
                // Round 3: the rounds are instruction-issue bound (38 pairs x ~200 instructions on one CU), so the pair's
                // arithmetic was cut roughly in half:
                //  * |x|^2 and |y|^2 are TRACKED, not recomputed: a Jacobi rotation maps them to al - t*ga and be + t*ga
                //    exactly, so a round computes ONE dot product (x.y) instead of three; they are recomputed from the
                //    data at every block-pair visit (here), i.e. after at most bs rotations of a column;
                //  * the rotation is applied in its scaled ("fast Givens") form: with x = sx*x~, y = sy*y~ the update
                //    x' = cs*(x - t*y), y' = cs*(y + t*x) is x~' = x~ - (t*sy/sx)*y~, y~' = y~ + (t*sx/sy)*x~ and
                //    sx' = cs*sx, sy' = cs*sy: one FMA per element instead of a multiply and an FMA; the scales are
                //    folded back into the columns at the end of the visit (cs >= 0.707: at most 2^-32 after bs = 64).
                float4 xr[EC];
                float alx = 0.0f, sx = 1.0f;
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const int e = 4 * (sub + 16 * c);
                    xr[c] = has_x ? *reinterpret_cast<const float4*>(cols + (size_t)row * ld + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    alx = fmaf(xr[c].x, xr[c].x, alx); alx = fmaf(xr[c].y, xr[c].y, alx);
                    alx = fmaf(xr[c].z, xr[c].z, alx); alx = fmaf(xr[c].w, xr[c].w, alx);
                }
                alx = row16_sum(alx);
                {   // |y_j|^2 of the y-columns: row `row` takes column `row`
                    float by = 0.0f;
                    const bool has_y = row < ny;
#pragma unroll
                    for (int c = 0; c < EC; ++c) {
                        const int e = 4 * (sub + 16 * c);
                        if (has_y) {
                            const float4 v = *reinterpret_cast<const float4*>(cols + (size_t)(nx + row) * ld + e);
                            by = fmaf(v.x, v.x, by); by = fmaf(v.y, v.y, by); by = fmaf(v.z, v.z, by); by = fmaf(v.w, v.w, by);
                        }
                    }
                    by = row16_sum(by);
                    if (has_y && sub == 0) { s_ynrm[row] = by; s_ysc[row] = 1.0f; }
                }
                __syncthreads();
#ifdef BBMPC_KERNEL_DBG
                long long cr[6] = {0, 0, 0, 0, 0, 0}, ct = 0;
#define SVDB_CLK(i) do { const long long now_ = (long long)__builtin_readcyclecounter(); cr[i] += now_ - ct; ct = now_; } while (0)
#else
#define SVDB_CLK(i) do {} while (0)
#endif
                int jrot = mm > 0 ? row % mm : 0;
                // The x-update of a rotation (x' = x - tau1 y_old) is not needed before the NEXT round's dot product: it is
                // deferred into the shadow of that round's column loads (tau1p, and the old y kept in the other of two
                // register buffers -- rounds alternate between them, so nothing is copied); tau1p = 0 leaves x as it is.
                float4 ya[EC], yb[EC];
#pragma unroll
                for (int c = 0; c < EC; ++c) yb[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                float tau1p = 0.0f;
                auto cross_round = [&](float4 (&yv)[EC], const float4 (&yp)[EC]) {
#ifdef BBMPC_KERNEL_DBG
                    ct = (long long)__builtin_readcyclecounter();
#endif
                    if (wave_on) {
                        const int j = jrot;                               // (row + r) % mm without the division on the chain
                        jrot = (jrot + 1 == mm) ? 0 : jrot + 1;
                        const bool act = has_x && j < ny;
                        const int jc = act ? j : 0;
                        float* y = cols + (size_t)(nx + jc) * ld;
                        // tracked scale and norm of the y-column: read FIRST, through asm (the compiler sinks plain loads behind the
                        // dot product and the threshold test -- two more LDS round trips on the round's dependent chain, a quarter
                        // of a round by the segment clocks).  LDS returns in order, so once the column loads issued after them
                        // have been waited for these two are there as well; the empty asm below ties their first use to the dot
                        // product, i.e. behind that wait.
                        float sy, be;
                        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3"
                                     : "=&v"(sy), "=&v"(be)
                                     : "v"((unsigned)(uintptr_t)(s_ysc + jc)), "v"((unsigned)(uintptr_t)(s_ynrm + jc))
                                     : "memory");
                        // the round is one dependent chain (LDS read -> dot product -> row reduction -> rotation scalars ->
                        // update -> LDS write -> barrier): four partial sums instead of a 4 EC-deep FMA chain
#pragma unroll
                        for (int c = 0; c < EC; ++c)
                            yv[c] = *reinterpret_cast<const float4*>(y + 4 * (sub + 16 * c));   // (an idle row reads column 0 and does not rotate)
                        // the two scalars are back once at most EC LDS operations are outstanding: everything that depends on them
                        // and not on the dot product -- the reciprocals of the scales (+ Newton step), be - al -- is computed here,
                        // in the shadow of the column loads, instead of behind the threshold test
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(sy), "+v"(be) : "n"(EC) : "memory");
                        const float al = alx;
#pragma unroll
                        for (int c = 0; c < EC; ++c) {                    // the pending x-update of the previous round
                            xr[c].x = fmaf(-tau1p, yp[c].x, xr[c].x); xr[c].y = fmaf(-tau1p, yp[c].y, xr[c].y);
                            xr[c].z = fmaf(-tau1p, yp[c].z, xr[c].z); xr[c].w = fmaf(-tau1p, yp[c].w, xr[c].w);
                        }
                        tau1p = 0.0f;
                        // sy / sx and sx / sy to fp32 (reciprocal + one Newton step): the pair of updates is an exact
                        // rotation only if the two factors are reciprocal to each other
                        float rx = __builtin_amdgcn_rcpf(sx), ry = __builtin_amdgcn_rcpf(sy);
                        rx = rx * fmaf(-sx, rx, 2.0f); ry = ry * fmaf(-sy, ry, 2.0f);
                        const float sy_rx = sy * rx, sx_ry = sx * ry, dd = be - al, sxy = sx * sy, thr = (tol * tol) * (al * be);
                        float gs0 = 0.0f, gs1 = 0.0f, gs2 = 0.0f, gs3 = 0.0f;
#pragma unroll
                        for (int c = 0; c < EC; ++c) {
                            gs0 = fmaf(xr[c].x, yv[c].x, gs0); gs1 = fmaf(xr[c].y, yv[c].y, gs1);
                            gs2 = fmaf(xr[c].z, yv[c].z, gs2); gs3 = fmaf(xr[c].w, yv[c].w, gs3);
                        }
                        const float gs = (gs0 + gs1) + (gs2 + gs3);
#ifdef BBMPC_KERNEL_DBG
                        asm volatile("" :: "v"(gs));
#endif
                        SVDB_CLK(0);
                        const float ga = sxy * row16_sum(gs);
#ifdef BBMPC_KERNEL_DBG
                        asm volatile("" :: "v"(ga));
#endif
                        SVDB_CLK(1);
                        // |ga| <= tol * sqrt(al*be)  <=>  ga^2 <= tol^2 * al * be  (no sqrt)
                        const bool rot = act && ga != 0.0f && (ga * ga > thr);
                        if (rot) {
                            // t = sign(zeta) / (|zeta| + sqrt(zeta^2 + 1)), zeta = (be - al) / (2 ga), without forming zeta:
                            // |2 ga| / (|d| + sqrt(d^2 + (2 ga)^2)), d = be - al -- one reciprocal less on the chain
                            const float g2 = ga + ga;
                            const float t = copysignf(fabsf(g2) * __builtin_amdgcn_rcpf(fabsf(dd) + __builtin_amdgcn_sqrtf(fmaf(dd, dd, g2 * g2))),
                                                      (dd < 0.0f) != (g2 < 0.0f) ? -1.0f : 1.0f);
                            const float w = fmaf(t, t, 1.0f);
                            float cs = __builtin_amdgcn_rsqf(w);
                            cs = cs * fmaf(-0.5f * w * cs, cs, 1.5f);           // Newton step: cs^2 * (1 + t^2) = 1 to fp32
                            const float tau1 = t * sy_rx, tau2 = t * sx_ry;
#pragma unroll
                            for (int c = 0; c < EC; ++c) {
                                const int e = 4 * (sub + 16 * c);
                                float4 yn;
                                yn.x = fmaf(tau2, xr[c].x, yv[c].x); yn.y = fmaf(tau2, xr[c].y, yv[c].y);
                                yn.z = fmaf(tau2, xr[c].z, yv[c].z); yn.w = fmaf(tau2, xr[c].w, yv[c].w);
                                *reinterpret_cast<float4*>(y + e) = yn;
                            }
                            tau1p = tau1;                                     // x' = x - tau1 y_old: next round (or the flush below)
                            sx = sx * cs;
                            alx = fmaf(-t, ga, al);
                            if (sub == 0) { s_ysc[jc] = sy * cs; s_ynrm[jc] = fmaf(t, ga, be); }
                            rot_cross = true;
                            nrot += sub == 0 ? 1u : 0u;
                        }
                        SVDB_CLK(2);
                    }
#ifdef BBMPC_KERNEL_DBG
                    asm volatile("s_waitcnt lgkmcnt(0)");
                    SVDB_CLK(3);
#endif
                    __syncthreads();
                    SVDB_CLK(4);
                };
                {
                    int r = 0;
                    for (; r + 1 < mm; r += 2) { cross_round(ya, yb); cross_round(yb, ya); }
                    if (r < mm) cross_round(ya, yb);
                    const bool last_a = (mm & 1) != 0;                    // the buffer the last round loaded holds the pending y
#pragma unroll
                    for (int c = 0; c < EC; ++c) {
                        const float4 yl = last_a ? ya[c] : yb[c];
                        xr[c].x = fmaf(-tau1p, yl.x, xr[c].x); xr[c].y = fmaf(-tau1p, yl.y, xr[c].y);
                        xr[c].z = fmaf(-tau1p, yl.z, xr[c].z); xr[c].w = fmaf(-tau1p, yl.w, xr[c].w);
                    }
                }
#ifdef BBMPC_KERNEL_DBG
                if (g == 0 && wg == 1 && tid == 0 && sweep == 0 && R == 3)
                    printf("[svdb] cross rounds (block round 3, %d rounds, cycles): load+dot %lld reduce %lld rotate %lld lds-drain %lld barrier %lld\n", mm, cr[0], cr[1], cr[2], cr[3], cr[4]);
#endif
                // fold the scales back: x from the registers, y in place
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const int e = 4 * (sub + 16 * c);
                    if (has_x)
                        *reinterpret_cast<float4*>(cols + (size_t)row * ld + e) = make_float4(sx * xr[c].x, sx * xr[c].y, sx * xr[c].z, sx * xr[c].w);
                }
                for (int c = wv; c < ny; c += NW) {
                    const float sc = s_ysc[c];
                    if (sc != 1.0f)
                        for (int q4 = lane; q4 < (n >> 2); q4 += 64) {
                            float4 v = *reinterpret_cast<const float4*>(cols + (size_t)(nx + c) * ld + 4 * q4);
                            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                            *reinterpret_cast<float4*>(cols + (size_t)(nx + c) * ld + 4 * q4) = v;
                        }
                }
                if (rot_cross && sub == 0) s_rotf[0] = 1;
                rotated |= rot_cross;
                __syncthreads();
            }
            SVDB_MARK(2);
            __syncthreads();
            const bool rc = s_rotf[0] != 0, rix = s_rotf[1] != 0, riy = s_rotf[2] != 0;
            rotated |= rc || rix || riy;            // (workgroup-wide: a row's own flag is not the wave's)
            // ---- write back what changed
            if (rc || rix || riy) {
                for (int c = wv; c < nx + ny; c += NW) {
                    if (c < nx ? !(rc || rix) : !(rc || riy)) continue;
                    float* dst = At + (size_t)(c < nx ? x0 + c : y0 + (c - nx)) * n;
                    if (same_xcd)
                        for (int q4 = lane; q4 < (n >> 2); q4 += 64) l2_store16(dst + 4 * q4, *reinterpret_cast<const cma_f32x4*>(cols + (size_t)c * ld + 4 * q4));
                    else
                        for (int q4 = lane; q4 < (n >> 2); q4 += 64) coh_store16(dst + 4 * q4, *reinterpret_cast<const cma_f32x4*>(cols + (size_t)c * ld + 4 * q4));
                }
            }
            if (tid == 0) {
                const unsigned nvx = vx + ((rc || rix) ? 1u : 0u), nvy = vy + ((rc || riy) ? 1u : 0u);
                if (rc || rix) __hip_atomic_store(ver + bx, nvx - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rc || riy) __hip_atomic_store(ver + by, nvy - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!skip_ix && !rix && !rc) __hip_atomic_store(iseen + bx, nvx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!skip_iy && !riy && !rc) __hip_atomic_store(iseen + by, nvy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!skip_cross && !rc)
                    __hip_atomic_store(pseen + blo * NB + bhi, ((bx < by ? nvx : nvy) << 16) | (bx < by ? nvy : nvx), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            }   // not everything skipped
            if (R == NB - 2 && rotated && lane == 0) __hip_atomic_store(sync + 1 + sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (R == NB - 2) {
                if (nrot) __hip_atomic_fetch_add(sync + CMA_SYNC_ROTATIONS + sweep, nrot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nrot = 0;
            }
            SVDB_MARK(3);
            ++bar;
            cma_instance_barrier_light(sync, bar * (unsigned)WPG);
            SVDB_MARK(4);
        }
        if (__hip_atomic_load(sync + 1 + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
#ifdef BBMPC_KERNEL_DBG
        if (g == 0 && blockIdx.x == 0 && tid == 0) printf("[svdb] sweep %d: load %lld intra %lld cross %lld store %lld barrier %lld (10ns, cumulative)\n", sweep, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]);
#endif
    }
}

// ---- Block Jacobi in the Gram domain (replaces the loop above as the default for n = H*U = 300) ---------------------
// Same decomposition as k_cma_svd_block -- 8 column blocks, 4 workgroups per instance each holding a PAIR of blocks in
// LDS, round-robin tournament of block pairs, the same rotation / threshold / clean-pair bookkeeping -- but the work
// on a block pair no longer walks 300-long columns once per rotation:
//   1. G = [X Y]^T [X Y]  (2bs x 2bs Gram matrix of the 2bs resident columns) on the matrix cores, once per visit
//   2. the cyclic sweep (intra-block pairs in block round 0, then the bs cross rounds) runs on G alone: a pair's
//      (|x|^2, |y|^2, x.y) are three entries of G, a rotation is a two-sided 2x2 update of G's rows and columns, and the
//      rotations are accumulated in V (2bs x 2bs).  A round is ~12 k element updates spread over 1024 threads and two
//      barriers, with no cross-lane reductions -- against 38 column pairs x 300 elements x (3 dot products + rotation)
//   3. [X Y] <- [X Y] V  on the matrix cores, once per visit
// The rotations are the same function of the same three numbers as in the one-sided form, so convergence, threshold
// and the final invariants are unchanged; G is rebuilt from the columns at every visit, so its rounding drift never
// outlives 75 rounds.  Blocks travel between workgroups as 16-byte write-through (sc0 sc1) stores and loads: the
// 4-byte scalar form the kernel above uses costs ~6x per byte (a quarter of its run time was the load phase).
// LDS: cols [Wp][n] | G [Wp][Wp+1] | V [Wp][Wp+1] | pair tables, Wp = 2*bs + 2 rounded up to 16 (two zero columns for
// the sit-out players of odd blocks).
inline int cma_gram_wp(int n) { const int bs = (n + 7) / 8; return ((2 * bs + 2 + 15) / 16) * 16; }
inline size_t cma_gram_lds_bytes(int n) {
    const int Wp = cma_gram_wp(n);
    return ((size_t)Wp * n + 2 * (size_t)Wp * (Wp + 1) + 4 * 64 + 16) * sizeof(float);
}

static __global__ __launch_bounds__(1024) void k_cma_svd_gram(CmaArgs p, float* At_all, unsigned* sync_all, int max_sweeps) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    constexpr int NB = 8;
    const int g = blockIdx.y, wg = blockIdx.x, WPG = gridDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n;
    const int NW = blockDim.x >> 6, nthr = blockDim.x;
    const int bs = (n + NB - 1) / NB, W = 2 * bs;
    const int Wp = ((W + 2 + 15) / 16) * 16, WT = Wp / 16, Gs = Wp + 1, n4 = n >> 2;
    float* cols = gsm;                                   // [Wp][n]  column c of the pair = cols + c*n
    float* Gm = cols + (size_t)Wp * n;                   // [Wp][Gs]
    float* Vm = Gm + (size_t)Wp * Gs;                    // [Wp][Gs]
    int* tp = (int*)(Vm + (size_t)Wp * Gs);              // [64] first column of pair slot i
    int* tq = tp + 64;                                   // [64] second column
    float* tc = (float*)(tq + 64);                       // [64] cosine   (1 when the pair does not rotate)
    float* tsn = tc + 64;                                // [64] sine     (0 when the pair does not rotate)
    float* At = At_all + (size_t)g * n * n;
    unsigned* sync = sync_all + (size_t)g * CMA_SYNC_WORDS;
    unsigned* ver = sync + 32;
    unsigned* iseen = sync + 40;
    unsigned* pseen = sync + 48;
    __shared__ int s_skip[3], s_rotf[3];
    const float tol = fminf(fmaxf(3.0e-8f * (float)n, 2.0e-6f), 1.0e-5f);
    const float tol2 = tol * tol;
    unsigned bar = 0;
    const int mi = (bs + 1) & ~1;                        // players of the intra-block tournament (a dummy when bs is odd)
    const int P = max(bs, mi);                           // pair slots per round: bs cross pairs | 2 * mi/2 intra pairs
    // the padding columns never change
    for (int i = tid; i < (Wp - W) * n; i += nthr) cols[(size_t)W * n + i] = 0.0f;

    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        bool rotated = false;
        for (int R = 0; R < NB - 1; ++R) {
            int bx, by;
            if (wg == 0) { bx = NB - 1; by = R; }
            else { bx = (R + wg) % (NB - 1); by = (R - wg + (NB - 1)) % (NB - 1); }
            const int x0 = bx * bs, y0 = by * bs;
            const int nx = max(0, min(bs, n - x0)), ny = max(0, min(bs, n - y0));
            const int blo = min(bx, by), bhi = max(bx, by);
            unsigned vx = 0, vy = 0;
            if (tid == 0) {
                vx = __hip_atomic_load(ver + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                vy = __hip_atomic_load(ver + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                const unsigned packed = ((bx < by ? vx : vy) << 16) | (bx < by ? vy : vx);
                s_skip[0] = __hip_atomic_load(pseen + blo * 8 + bhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == packed;
                s_skip[1] = R != 0 || __hip_atomic_load(iseen + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == vx;
                s_skip[2] = R != 0 || __hip_atomic_load(iseen + by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == vy;
                s_rotf[0] = s_rotf[1] = s_rotf[2] = 0;
            }
            __syncthreads();
            const bool skip_cross = s_skip[0] != 0, skip_ix = s_skip[1] != 0, skip_iy = s_skip[2] != 0;
            if (!(skip_cross && skip_ix && skip_iy)) {
                // ---- load the two blocks: one column per wave at a time, a lane's 16-byte loads all in flight
                for (int c = wv; c < W; c += NW) {
                    const bool isx = c < bs;
                    const int ci = isx ? c : c - bs;
                    const bool present = isx ? ci < nx : ci < ny;
                    float* dst = cols + (size_t)c * n;
                    if (present) {
                        const float* src = At + (size_t)((isx ? x0 : y0) + ci) * n;
                        cma_f32x4 v0, v1, v2, v3;                               // addresses clamped: every lane loads something valid
                        coh_load16x4(src + 4 * min(lane, n4 - 1), src + 4 * min(lane + 64, n4 - 1), src + 4 * min(lane + 128, n4 - 1),
                                     src + 4 * min(lane + 192, n4 - 1), v0, v1, v2, v3);                       // n <= 1024
                        if (lane < n4) *reinterpret_cast<cma_f32x4*>(dst + 4 * lane) = v0;
                        if (lane + 64 < n4) *reinterpret_cast<cma_f32x4*>(dst + 4 * (lane + 64)) = v1;
                        if (lane + 128 < n4) *reinterpret_cast<cma_f32x4*>(dst + 4 * (lane + 128)) = v2;
                        if (lane + 192 < n4) *reinterpret_cast<cma_f32x4*>(dst + 4 * (lane + 192)) = v3;
                    } else {
                        for (int e = lane; e < n; e += 64) dst[e] = 0.0f;               // short last block: zero columns never rotate
                    }
                }
                // V = I
                for (int i = tid; i < Wp * Gs; i += nthr) Vm[i] = ((i / Gs) == (i % Gs)) ? 1.0f : 0.0f;
                __syncthreads();
                // ---- 1. Gram matrix on the matrix cores: upper tiles, mirrored.  A = 16 columns x 4 elements, B likewise
                {
                    const int ntile = WT * (WT + 1) / 2;
                    for (int t = wv; t < ntile; t += NW) {
                        int ti = 0, rem = t;
                        while (rem >= WT - ti) { rem -= WT - ti; ++ti; }
                        const int tj = ti + rem;
                        const float* ap = cols + (size_t)(ti * 16 + (lane & 15)) * n + (lane >> 4);
                        const float* bp = cols + (size_t)(tj * 16 + (lane & 15)) * n + (lane >> 4);
                        cma_f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
                        int k0 = 0;
                        for (; k0 + 8 <= n; k0 += 8) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k0], bp[k0], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k0 + 4], bp[k0 + 4], acc1, 0, 0, 0);
                        }
                        for (; k0 < n; k0 += 4) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k0], bp[k0], acc0, 0, 0, 0);
                        const int cj = tj * 16 + (lane & 15);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ci = ti * 16 + 4 * (lane >> 4) + r;
                            const float v = acc0[r] + acc1[r];
                            Gm[ci * Gs + cj] = v;
                            Gm[cj * Gs + ci] = v;
                        }
                    }
                }
                __syncthreads();
                // ---- 2. the sweep over this block pair, on G.  Rounds: [intra-block tournament (block round 0)] + cross
                const int n_intra = (R == 0 && !(skip_ix && skip_iy)) ? mi - 1 : 0;
                const int n_cross = skip_cross ? 0 : bs;
                bool rot_any = false;
                for (int rr = 0; rr < n_intra + n_cross; ++rr) {
                    const bool intra = rr < n_intra;
                    const int np = intra ? mi : bs;                                    // pair slots in use this round
                    if (tid < np) {
                        int pa, pb, kind;
                        if (intra) {
                            const int half = mi / 2, blk = tid >= half ? 1 : 0, i = tid - blk * half, r = rr;
                            int a, b;
                            if (i == 0) { a = mi - 1; b = r; }
                            else { a = (r + i) % (mi - 1); b = (r - i + (mi - 1)) % (mi - 1); }
                            // player -> column: a real column of the block, or (odd bs) the block's zero column
                            pa = a < bs ? blk * bs + a : W + blk;
                            pb = b < bs ? blk * bs + b : W + blk;
                            kind = 1 + blk;
                            if (blk ? skip_iy : skip_ix) kind = -1;
                        } else {
                            const int r = rr - n_intra;
                            pa = tid;
                            pb = bs + (tid + r) % bs;
                            kind = 0;
                        }
                        const float al = Gm[pa * Gs + pa], be = Gm[pb * Gs + pb], ga = Gm[pa * Gs + pb];
                        float cs = 1.0f, sn = 0.0f;
                        if (kind >= 0 && ga != 0.0f && ga * ga > tol2 * (al * be)) {
                            const float zeta = (be - al) * __builtin_amdgcn_rcpf(2.0f * ga);
                            const float t = copysignf(__builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.0f))), zeta);
                            const float w = fmaf(t, t, 1.0f);
                            cs = __builtin_amdgcn_rsqf(w);
                            cs = cs * fmaf(-0.5f * w * cs, cs, 1.5f);                   // Newton step: cs^2 (1 + t^2) = 1 to fp32
                            sn = cs * t;
                            s_rotf[kind] = 1;
                        }
                        tp[tid] = pa; tq[tid] = pb; tc[tid] = cs; tsn[tid] = sn;
                    }
                    __syncthreads();
                    // two-sided update of G: the np x np blocks of pair slots (a, b) partition the touched rows / columns
                    for (int t = tid; t < np * np; t += nthr) {
                        const int a = t / np, b = t - a * np;
                        const float sa = tsn[a], sb = tsn[b];
                        if (sa == 0.0f && sb == 0.0f) continue;
                        const float ca = tc[a], cb = tc[b];
                        const int pa = tp[a], qa = tq[a], pb = tp[b], qb = tq[b];
                        const float m00 = Gm[pa * Gs + pb], m01 = Gm[pa * Gs + qb], m10 = Gm[qa * Gs + pb], m11 = Gm[qa * Gs + qb];
                        const float r00 = fmaf(ca, m00, -(sa * m10)), r01 = fmaf(ca, m01, -(sa * m11));
                        const float r10 = fmaf(sa, m00, ca * m10), r11 = fmaf(sa, m01, ca * m11);
                        Gm[pa * Gs + pb] = fmaf(cb, r00, -(sb * r01));
                        Gm[pa * Gs + qb] = fmaf(sb, r00, cb * r01);
                        Gm[qa * Gs + pb] = fmaf(cb, r10, -(sb * r11));
                        Gm[qa * Gs + qb] = fmaf(sb, r10, cb * r11);
                    }
                    // V <- V J
                    for (int t = tid; t < Wp * np; t += nthr) {
                        const int k = t / np, b = t - k * np;
                        const float sb = tsn[b];
                        if (sb == 0.0f) continue;
                        const float cb = tc[b];
                        const int pb = tp[b], qb = tq[b];
                        const float v0 = Vm[k * Gs + pb], v1 = Vm[k * Gs + qb];
                        Vm[k * Gs + pb] = fmaf(cb, v0, -(sb * v1));
                        Vm[k * Gs + qb] = fmaf(sb, v0, cb * v1);
                    }
                    __syncthreads();
                }
                const bool rc = s_rotf[0] != 0, rix = s_rotf[1] != 0, riy = s_rotf[2] != 0;
                rot_any = rc || rix || riy;
                rotated |= rot_any;
                // ---- 3. [X Y] <- [X Y] V on the matrix cores: a wave owns 16 rows (elements) of every column
                if (rot_any) {
                    const int ET = (n + 15) / 16;
                    for (int et = wv; et < ET; et += NW) {
                        cma_f32x4 acc[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[c] = cma_f32x4{0, 0, 0, 0};
                        const float* ap = cols + (size_t)(lane >> 4) * n + et * 16 + (lane & 15);
                        const float* bp = Vm + (size_t)(lane >> 4) * Gs + (lane & 15);
                        for (int k0 = 0; k0 < W; k0 += 4) {
                            const float a = ap[(size_t)k0 * n];
#pragma unroll
                            for (int c = 0; c < 8; ++c)
                                if (c < WT) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[(size_t)k0 * Gs + c * 16], acc[c], 0, 0, 0);
                        }
                        const int e = et * 16 + 4 * (lane >> 4);
                        if (e < n) {
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const int col = c * 16 + (lane & 15);
                                if (c < WT && col < W) *reinterpret_cast<cma_f32x4*>(cols + (size_t)col * n + e) = acc[c];
                            }
                        }
                    }
                    __syncthreads();
                    // ---- write back what changed, 16-byte write-through stores
                    for (int c = wv; c < W; c += NW) {
                        const bool isx = c < bs;
                        const int ci = isx ? c : c - bs;
                        if (!(isx ? ci < nx : ci < ny)) continue;
                        if (isx ? !(rc || rix) : !(rc || riy)) continue;
                        float* dst = At + (size_t)((isx ? x0 : y0) + ci) * n;
                        const float* src = cols + (size_t)c * n;
                        for (int q = lane; q < n4; q += 64) coh_store16(dst + 4 * q, *reinterpret_cast<const cma_f32x4*>(src + 4 * q));
                    }
                }
                if (tid == 0) {
                    const unsigned nvx = vx + ((rc || rix) ? 1u : 0u), nvy = vy + ((rc || riy) ? 1u : 0u);
                    if (rc || rix) __hip_atomic_store(ver + bx, nvx - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (rc || riy) __hip_atomic_store(ver + by, nvy - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!skip_ix && !rix && !rc) __hip_atomic_store(iseen + bx, nvx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!skip_iy && !riy && !rc) __hip_atomic_store(iseen + by, nvy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!skip_cross && !rc)
                        __hip_atomic_store(pseen + blo * 8 + bhi, ((bx < by ? nvx : nvy) << 16) | (bx < by ? nvy : nvx), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (R == NB - 2 && rotated && lane == 0) __hip_atomic_store(sync + 1 + sweep, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

            ++bar;
            cma_instance_barrier_light(sync, bar * (unsigned)WPG);
        }
        if (__hip_atomic_load(sync + 1 + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
    }
}

// ---- Small instances (n <= 128, e.g. the pendulum's n = H*U = 30): the whole matrix in LDS, one 16-lane DPP row per
// column pair (up to eight elements per lane), rcp / rsq rotation scalars as in the block kernel.  A round is ~40
// instructions per wave + one barrier; the general kernels above spend ~1 us per round on predicated 512-wide code.
// LDS: n*n floats.  blockDim = 64 * ceil(pairs / 4).  sync: [G][32] as above (sweep flags only).
template <int EC>   // elements per lane: 4 (n <= 64) or 8 (n <= 128)
__device__ __forceinline__ void cma_svd_small_body(const CmaArgs& p, int g, float* At_all, int max_sweeps, float* at_s) {
    // any blockDim >= 64 * ceil(pairs / 4): 16-lane rows beyond the pair slots only take part in the barriers
    __shared__ int s_rot;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = p.n;
    float* At_g = At_all + (size_t)g * n * n;
    const float tol = fminf(fmaxf(3.0e-8f * (float)n, 2.0e-6f), 1.0e-5f);
    const int nc = (n + 15) >> 4;                              // elements per lane (<= 8: n <= 128)
    for (int i = tid; i < n * n; i += blockDim.x) at_s[i] = At_g[i];
    const int m = (n + 1) & ~1;
    const int sub = lane & 15, slot = wv * 4 + (lane >> 4);     // pair slot of this 16-lane row
    __syncthreads();
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int r = 0; r < m - 1; ++r) {
            if (wv * 4 >= m / 2) { __syncthreads(); continue; }          // a wave without pair slots only keeps the barrier count
            int pa = 0, pb = 0;
            bool act = slot < m / 2;
            if (act) {
                if (slot == 0) { pa = m - 1; pb = r; }
                else { pa = (r + slot) % (m - 1); pb = (r - slot + (m - 1)) % (m - 1); }
                act = pa < n && pb < n;
            }
            float* x = at_s + (size_t)(act ? pa : 0) * n;
            float* y = at_s + (size_t)(act ? pb : 0) * n;
            float xv[EC], yv[EC], al = 0.0f, be = 0.0f, ga = 0.0f;
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                if (c < nc) {
                    const int e = sub + 16 * c;
                    xv[c] = (act && e < n) ? x[e] : 0.0f;
                    yv[c] = (act && e < n) ? y[e] : 0.0f;
                    al = fmaf(xv[c], xv[c], al); be = fmaf(yv[c], yv[c], be); ga = fmaf(xv[c], yv[c], ga);
                }
            }
            al = row16_sum(al); be = row16_sum(be); ga = row16_sum(ga);
            if (act && ga != 0.0f && ga * ga > (tol * tol) * (al * be)) {
                const float zeta = (be - al) * __builtin_amdgcn_rcpf(2.0f * ga);
                const float t = copysignf(__builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.0f))), zeta);
                const float w = fmaf(t, t, 1.0f);
                float cs = __builtin_amdgcn_rsqf(w);
                cs = cs * fmaf(-0.5f * w * cs, cs, 1.5f);
                const float sn = cs * t;
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const int e = sub + 16 * c;
                    if (c < nc && e < n) {
                        x[e] = fmaf(cs, xv[c], -(sn * yv[c]));
                        y[e] = fmaf(sn, xv[c], cs * yv[c]);
                    }
                }
                if (sub == 0) s_rot = 1;
            }
            __syncthreads();
        }
        if (!s_rot) break;
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < n * n; i += blockDim.x) At_g[i] = at_s[i];
}
template <int EC>
__global__ __launch_bounds__(1024) void k_cma_svd_small(CmaArgs p, float* At_all, unsigned* sync_all, int max_sweeps) {
    extern __shared__ __attribute__((aligned(16))) float at_s[];
    (void)sync_all;
    cma_svd_small_body<EC>(p, blockIdx.x, At_all, max_sweeps, at_s);
}

#endif  // BBMPC_TU_CMA

}  // namespace bbmpc
