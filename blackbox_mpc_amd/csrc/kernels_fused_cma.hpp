// CMA-ES on the analytic pendulum, ONE launch per control step for small search dimensions (n = H*U <= 64, e.g. the
// n = 30 of config 2's size): the per-iteration path issues eleven launch-bound kernels per iteration (noise, B*D,
// sampling GEMM, rollout, top-k, evolution paths, covariance, warm start, Jacobi, finish + a memset), each a few
// microseconds of work behind ~5 us of dispatch -- 760 us per control step of which 63 us per iteration is the Jacobi.
// Here one 1024-thread workgroup per CMA-ES instance walks the same phases with barriers in between.  The phases are
// the SAME device functions the stand-alone kernels run (kernels_cma.hpp: cma_select_body, cma_paths_body,
// cma_svd_small_body, cma_svd_finish_body) or element-wise restatements with the same operation order (sampling GEMM,
// covariance, warm start: sequential fmaf chains), and they work on the same global scratch buffers (L2 / L1 resident:
// all of an instance's data is < 200 KB), so the fused control step is bit-identical to the per-iteration one
// (tests/test_gpu_cmaes.py).  Instances with agents_per_group == 1 only (per-agent mode, or one agent).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_cma.hpp"
#include "kernels_eigh_small.hpp"
#include "kernels_opt.hpp"
#include "kernels_refit.hpp"
#include "kernels_rollout.hpp"

namespace bbmpc {
#ifdef BBMPC_TU_CMA      // compiled in the CMA-ES translation unit only (csrc/bbmpc_cma.hip)

struct FusedCmaArgs {
    CmaArgs q;                 // q.iter / q.inj are set per iteration inside the kernel
    int iters, H;
    const float* inj;          // injected standard normals [iters][A][HU][Nst] (internal layout) or null
    size_t inj_stride;
    float* evec;               // [G][n][n] Jacobi work matrix (columns as rows)
    float* eval;               // [G*n] singular values
    int* info;                 // [G*n] column permutation
    FinalArgs fin;             // tail of OptimizerBase.__call__ (state, bounds, exploration noise, record, next_state)
    unsigned* done_flag;       // publish_records_done or null
    unsigned* done_count;
    unsigned done_value;
    int eigh_small, eigh_fail; // Engine::cma_use_eigh_small(), BBMPC_CMA_EIGH_FAIL
};

template <bool FASTM>
__global__ __launch_bounds__(1024) void k_fused_cma_pendulum(FusedCmaArgs f) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    CmaArgs p = f.q;
    const int g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int n = p.n, N = p.N, Nst = p.Nst;
    const size_t off = (size_t)g * n, nn = (size_t)n * n;
    const int ga = p.agent_offset + g;                       // global agent id of this instance (agents_per_group == 1)
    const float lo0 = f.fin.lo[0], hi0 = f.fin.hi[0];
#ifdef BBMPC_KERNEL_DBG
    long long fc_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fc_t0 = (long long)wall_clock64();
#define FCMA_MARK(i) do { __syncthreads(); const long long now_ = (long long)wall_clock64(); fc_acc[i] += now_ - fc_t0; fc_t0 = now_; } while (0)
#else
#define FCMA_MARK(i) do {} while (0)
#endif
    for (int it = 0; it < f.iters; ++it) {
        p.iter = (uint32_t)it;
        p.inj = f.inj ? f.inj + f.inj_stride * it : nullptr;
        // ---- z ~ N(0, I)  (k_cma_noise: one Philox block gives the normals of four consecutive elements)  and
        //      BD = B D  (k_cma_bd), kept in LDS                                                         cma_es.py:139-140
        if (p.inj) {
            for (int idx = tid; idx < n * N; idx += nthr) {
                const int j = idx / N, q = idx - j * N;
                const size_t i = (off + j) * Nst + q;
                p.z[i] = p.inj[i];
            }
        } else {
            const int nq4 = (n + 3) >> 2;
            for (int idx = tid; idx < nq4 * N; idx += nthr) {
                const int jq = idx / N, q = idx - jq * N;
                const U4 b = rng_block(p.key, 4u, p.iter, (uint32_t)q, (uint32_t)ga, (uint32_t)(jq * 4));
                float zz[4];
                words_to_normal2(b.x, b.y, zz[0], zz[1]);
                words_to_normal2(b.z, b.w, zz[2], zz[3]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (jq * 4 + u < n) p.z[(off + jq * 4 + u) * Nst + q] = zz[u];
            }
        }
        for (int i = tid; i < (int)nn; i += nthr) fsm[i] = p.B[(size_t)g * nn + i] * p.Dd[off + i % n];
        __syncthreads();
        FCMA_MARK(0);
        // ---- y = z (B D), samples = m + sigma * y  (k_cma_gemm_y: one fmaf chain over l per element)    :140-141
        // thread = (block of rows i, particle q); five chains share each z load, B D is read from LDS (broadcast)
        {
            const float* __restrict__ Z = p.z + off * Nst;
            constexpr int IB = 5;
            const int P = min(n, max(1, nthr / N));
            const int per = (n + P - 1) / P;
            const int part = tid / N, q = tid - part * N;
            if (part < P) {
                const int i1 = min(n, (part + 1) * per);
                for (int ib = part * per; ib < i1; ib += IB) {
                    float acc[IB];
                    int iu[IB];
#pragma unroll
                    for (int u = 0; u < IB; ++u) { acc[u] = 0.0f; iu[u] = min(ib + u, n - 1); }
                    for (int l = 0; l < n; ++l) {
                        const float zv = Z[(size_t)l * Nst + q];
#pragma unroll
                        for (int u = 0; u < IB; ++u) acc[u] = fmaf(fsm[l * n + iu[u]], zv, acc[u]);
                    }
#pragma unroll
                    for (int u = 0; u < IB; ++u)
                        if (ib + u < i1) p.cand[(off + ib + u) * Nst + q] = p.m[off + ib + u] + p.sigma[off + ib + u] * acc[u];
                }
            }
        }
        __syncthreads();
        FCMA_MARK(1);
        // ---- rollouts: clip + penalty, H pendulum steps  (k_rollout_pendulum<SRC_BUF, PEN>)              :147-157
        for (int q = tid; q < N; q += nthr) {
            Roller<FASTM> roll(f.fin.fix_q1 != 0, f.fin.state[g * 3 + 0], f.fin.state[g * 3 + 1], f.fin.state[g * 3 + 2]);
            float total = 0.0f, pen = 0.0f;
            for (int t0 = 0; t0 < f.H; t0 += 8) {
                float xs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xs[u] = p.cand[(off + min(t0 + u, f.H - 1)) * Nst + q];    // eight loads in flight
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (t0 + u < f.H) {
                        float x = xs[u];
                        const float xf = clipf(x, lo0, hi0);
                        const float d = x - xf;
                        pen = pen + d * d;
                        x = xf;
                        p.cand[(off + t0 + u) * Nst + q] = x;
                        roll.step_acc(x);
                    }
                }
            }
            total = roll.total();
            if (total != total) total = -1.0e6f;
            const float nr = sqrtf(pen);
            pen = nr * nr;
            total = total - pen;
            const_cast<float*>(p.rewards)[(size_t)g * Nst + q] = total;
        }
        __syncthreads();
        FCMA_MARK(2);
        cma_select_body(p, g, fsm);                                                                      // :158-159
        __syncthreads();
        FCMA_MARK(3);
        cma_paths_body(p, g);                                                                            // :161-177
        __syncthreads();
        FCMA_MARK(4);
        cma_cov_small_body(p, g);                                                                        // :179-190 (k_cma_cov)
        __syncthreads();
        FCMA_MARK(5);
        // ---- B, D from C: the direct solver (kernels_eigh_small.hpp) or warm start + Jacobi + finish            :195-206
        if (f.eigh_small) {
            cma_factor_small_body(p, g, f.evec, f.eval, f.info, f.eigh_fail != 0, fsm);
        } else {
            cma_warm_small_body(p, g, f.evec);
            __syncthreads();
            cma_svd_small_body<4>(p, g, f.evec, 15, fsm);
            __syncthreads();
            cma_svd_finish_body(p, g, f.evec, f.eval, f.info);
        }
        __syncthreads();
        FCMA_MARK(8);
    }
    // ---- action = m[:, 0] (:211-212); exploration noise, predicted next state + reward (optimizer_base.py:82-94)
#ifdef BBMPC_KERNEL_DBG
    if (g == 0 && tid == 0)
        printf("[fcma] n=%d N=%d k=%d iters=%d | noise+BD %lld  gemm_y %lld  rollout %lld  select %lld  paths %lld  cov %lld  factor %lld (10 ns units, all iterations)\n",
               n, N, p.k, f.iters, fc_acc[0], fc_acc[1], fc_acc[2], fc_acc[3], fc_acc[4], fc_acc[5], fc_acc[8]);
#endif
    if (tid == 0) {
        finalize_pendulum_agent(f.fin, g, p.m[off]);
        publish_records_done(f.done_flag, f.done_count, f.done_value, gridDim.x);
    }
}

#endif  // BBMPC_TU_CMA
}  // namespace bbmpc
