// CMA-ES half of the engine (CMAESOptimizer, optimizers/cma_es.py:43-227): state, the per-iteration launch sequence, the
// eigen-decomposition (direct solver kernels_eigh.hpp, block Jacobi kernels_cma.hpp) and the one-launch control step of small
// search dimensions.  A translation unit of its own: its kernels are compiled here only (BBMPC_TU_CMA).
#define BBMPC_TU_CMA
#include "engine.hpp"
#include "engine_util.hpp"

namespace bbmpc {

void bbmpc_tu_cma_upload_tnq(const float2* table) { tnq_upload(table); }

// ---- CMA-ES host side -----------------------------------------------------------------------------
void Engine::cma_init() {
    const bool per_agent = fix(BBMPC_CMAES_PER_AGENT);
    cma_G = per_agent ? A : 1;
    cma_n = per_agent ? HU : A * HU;
    const int n = cma_n, G = cma_G;
    // recombination weights + constants, fp32 with the reference's op order (cma_es.py:62-92, 118-126)
    std::vector<float> w((size_t)k);
    const float lk = (float)log((double)((float)k + 0.5f));
    float wsum = 0.0f;
    for (int i = 0; i < k; ++i) { w[i] = lk - (float)log((double)(float)(i + 1)); wsum += w[i]; }
    float s1 = 0.0f, s2 = 0.0f;
    for (int i = 0; i < k; ++i) { w[i] = w[i] / wsum; s1 += w[i]; s2 += w[i] * w[i]; }
    CmaConst& c = cma_c;
    const float nf = (float)n, ac = cfg.cma_alpha_cov;
    c.mu_eff = (s1 * s1) / s2;
    c.c_sigma = (c.mu_eff + 2.0f) / ((nf + c.mu_eff) + 5.0f);
    c.d_sigma = (1.0f + 2.0f * std::max(0.0f, sqrtf((c.mu_eff - 1.0f) / (nf + 1.0f)) - 1.0f)) + c.c_sigma;
    c.cc = (4.0f + c.mu_eff / nf) / ((nf + 4.0f) + (2.0f * c.mu_eff) / nf);
    c.c1 = ac / ((nf + 1.3f) * (nf + 1.3f) + c.mu_eff);
    const float cmu2 = ac * ((c.mu_eff - 2.0f) + 1.0f / c.mu_eff) / ((nf + 2.0f) * (nf + 2.0f) + (ac * c.mu_eff) / 2.0f);
    c.c_mu = std::min(1.0f - c.c1, cmu2);
    c.e_norm = sqrtf(nf * ((1.0f - 1.0f / (4.0f * nf)) + 1.0f / (21.0f * (nf * nf))));
    c.h_sigma = cfg.cma_h_sigma;
    upload(c_w, w);
    const size_t gn = (size_t)G * n, gnn = gn * n;
    c_m.alloc(gn); c_sigma.alloc(gn); c_Dd.alloc(gn); c_ps.alloc(gn); c_pc.alloc(gn); c_xm.alloc(gn); c_ym.alloc(gn);
    c_eval.alloc(gn); c_E.alloc(gn);
    c_C.alloc(gnn); c_B.alloc(gnn); c_BD.alloc(gnn); c_evec.alloc(gnn);
    c_z.alloc((size_t)A * HU * Nst);
    c_Ye.alloc((size_t)G * k * n);
    c_eidx.alloc((size_t)G * k);
    c_info.alloc(gn);           // SVD: column permutation
    c_sync.alloc((size_t)G * CMA_SYNC_WORDS);
    if (cma_use_eigh()) {
        const size_t ld = EIGH_LD, mat = ld * ld;
        e_d.alloc(G * ld); e_e.alloc(G * ld); e_tau.alloc(G * ld); e_alpha.alloc(G); e_lam.alloc(G * ld);
        e_Vt.alloc(G * mat); e_Z.alloc(G * mat); e_Z2.alloc(G * mat); e_P.alloc(G * mat);
        e_Tf.alloc((size_t)G * EIGH_TF_WGS * 1024);
        e_flags.alloc((size_t)G * 8);
        e_flags.zero(stream);
    }
    // C = B = D = I, paths = 0 (cma_es.py:98-117)
    std::vector<float> eye(gnn, 0.0f), ones(gn, 1.0f);
    for (int g = 0; g < G; ++g)
        for (int i = 0; i < n; ++i) eye[(size_t)g * n * n + (size_t)i * n + i] = 1.0f;
    HIP_CHECK(hipMemcpy(c_C.p, eye.data(), gnn * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c_B.p, eye.data(), gnn * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c_Dd.p, ones.data(), gn * 4, hipMemcpyHostToDevice));
    c_ps.zero(stream);
    c_pc.zero(stream);
    cma_reset_mean_sigma();
}

void Engine::cma_reset_mean_sigma() {
    // m = bounds midpoint, sigma = sqrt((lo-hi)^2/16) per coordinate (cma_es.py:48-59,95-97; reset :215-227)
    const size_t gn = (size_t)cma_G * cma_n;          // == A*HU in both modes, same (a,h,u) order
    std::vector<float> m(gn), sg(gn);
    for (size_t i = 0; i < gn; ++i) {
        const int u = (int)(i % U);
        m[i] = (lo[u] + hi[u]) / 2.0f;
        const float d = lo[u] - hi[u];
        sg[i] = sqrtf((d * d) / 16.0f);
    }
    HIP_CHECK(hipMemcpy(c_m.p, m.data(), gn * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(c_sigma.p, sg.data(), gn * 4, hipMemcpyHostToDevice));
}

CmaArgs Engine::cma_args(uint32_t step, uint32_t iter) {
    CmaArgs q;
    memset(&q, 0, sizeof(q));
    q.N = N; q.A = A; q.HU = HU; q.Nst = Nst; q.k = k;
    q.G = cma_G; q.n = cma_n;
    q.agents_per_group = fix(BBMPC_CMAES_PER_AGENT) ? 1 : A;
    q.agent_offset = cfg.agent_offset;
    q.c = cma_c;
    q.weights = c_w.p;
    q.m = c_m.p; q.sigma = c_sigma.p; q.C = c_C.p; q.B = c_B.p; q.Dd = c_Dd.p; q.p_sigma = c_ps.p; q.p_C = c_pc.p;
    q.lo = d_lo.p; q.hi = d_hi.p; q.U = U;
    q.BD = c_BD.p; q.z = c_z.p; q.cand = d_cand_a.p; q.rewards = d_rewards.p; q.eidx = c_eidx.p; q.Ye = c_Ye.p;
    q.xmean = c_xm.p; q.ymean = c_ym.p;
    q.key = key(step);
    q.iter = iter;
    q.pop_offset = cfg.population_offset;
    return q;
}

// s, U, _ = tf.linalg.svd(C); B = U, D = diag(sqrt(s))  (cma_es.py:195-198) by the direct eigensolver of kernels_eigh.hpp:
// eight launches on the handle's stream; instances that fail its checks keep e_flags[8 g] = 1 and B, D untouched
void Engine::cma_eigh_launch(const CmaArgs& cq) {
    EighArgs q;
    memset(&q, 0, sizeof(q));
    q.n = cma_n; q.G = cma_G;
    q.force_fail = sw.cma_eigh_fail ? 1 : 0;
    q.jacobi_sync = c_sync.p; q.jacobi_sync_words = CMA_SYNC_WORDS;
    q.C = cq.C; q.B = cq.B; q.Dd = cq.Dd;
    q.d = e_d.p; q.e = e_e.p; q.tau = e_tau.p; q.Vt = e_Vt.p; q.alpha = e_alpha.p; q.lam = e_lam.p;
    q.Z = e_Z.p; q.Z2 = e_Z2.p; q.P = e_P.p; q.Tf = e_Tf.p; q.flags = e_flags.p;
    const int G = cma_G;
    const size_t lds1 = sizeof(EighTriLds);
    const size_t lds2 = std::max(sizeof(EighSolveLds), (size_t)(32 * (EIGH_LD + 1) + 32 * 33) * sizeof(float));
    ensure_max_lds((const void*)k_eigh_tridiag, (int)lds1);
    ensure_max_lds((const void*)k_eigh_tri_solve, (int)lds2);
    if (eigh_side_state < 0) {
        eigh_side = masked_stream_from_env("BBMPC_EIGH_SIDE_CUS", cu_count);
        eigh_side_state = eigh_side ? 1 : 0;
        if (eigh_side)
            for (int i = 0; i < 2; ++i) HIP_CHECK(hipEventCreateWithFlags(&eigh_ev[i], hipEventDisableTiming));
    }
    if (eigh_side_state == 1) {
        // experiment (profiles/r5_cfg5cma_overlap.md): the one-workgroup-per-instance phase leaves the launch stream to the
        // handles that share the GPU; everything after it waits for it
        HIP_CHECK(hipEventRecord(eigh_ev[0], stream));
        HIP_CHECK(hipStreamWaitEvent(eigh_side, eigh_ev[0], 0));
        hipLaunchKernelGGL(k_eigh_tridiag, dim3(G), dim3(EIGH_TRI_THREADS), lds1, eigh_side, q);
        HIP_CHECK(hipEventRecord(eigh_ev[1], eigh_side));
        HIP_CHECK(hipStreamWaitEvent(stream, eigh_ev[1], 0));
    } else
    hipLaunchKernelGGL(k_eigh_tridiag, dim3(G), dim3(EIGH_TRI_THREADS), lds1, stream, q);
    hipLaunchKernelGGL(k_eigh_tri_solve, dim3(EIGH_SLOT_WGS + EIGH_TF_WGS, G), dim3(EIGH_SOLVE_THREADS), lds2, stream, q);
    const dim3 gg(EIGH_LD / 64, EIGH_LD / 16, G);
    hipLaunchKernelGGL(k_eigh_gemm<0>, gg, dim3(256), 0, stream, q, (const float*)q.Z, (const float*)nullptr, q.P, 1, 0);
    hipLaunchKernelGGL(k_eigh_gemm<1>, gg, dim3(256), 0, stream, q, (const float*)q.Z, (const float*)q.P, q.Z2, 0, 0);
    hipLaunchKernelGGL(k_eigh_gemm<0>, gg, dim3(256), 0, stream, q, (const float*)q.Z2, (const float*)nullptr, q.P, 3, 1);
    hipLaunchKernelGGL(k_eigh_gemm<1>, gg, dim3(256), 0, stream, q, (const float*)q.Z2, (const float*)q.P, q.Z, 0, 1);
    hipLaunchKernelGGL(k_eigh_backtransform, dim3(EIGH_LD / 16, G), dim3(EIGH_BT_THREADS), 0, stream, q, (const float*)q.Z, (const float*)q.Z2);
    HIP_CHECK(hipGetLastError());
}

// CMAESOptimizer._optimize  cma_es.py:129-213
void Engine::optimize_cma(RolloutArgs& ra, uint32_t step) {
    const int n = cma_n, G = cma_G;
    const float* inj_n = injected(BBMPC_NOISE_NORMAL);
    const size_t inj_stride = (size_t)A * HU * Nst;
    const size_t gnn = (size_t)G * n * n;
    for (int it = 0; it < iters; ++it) {
        CmaArgs q = cma_args(step, (uint32_t)it);
        q.inj = inj_n ? inj_n + inj_stride * it : nullptr;
        const int kp = (k + 3) & ~3;
        const size_t lds = (size_t)(Nst + TOPK_HIST_WORDS + 2 * kp) * 4;
        const bool small3 = sw.cma_small3 && cma_use_eigh_small() && !pop_sharded();
        if (small3) {
            // n <= 32: sample | roll out | update, three launches per iteration (kernels_eigh_small.hpp)
            const bool write_back = trace_on || user_path();
            // the analytic pendulum, one agent per instance: the rollouts ride on the sampling launch
            const bool roll_in = !write_back && cfg.dynamics == BBMPC_DYN_PENDULUM && cfg.reward == BBMPC_REW_PENDULUM && cma_G == A && U == 1 && S == 3;
            if (roll_in) {
                dominant_kernel = "k_cma_sample_roll_small";          // (the launch that carries the rollouts)
                prof_begin();
                if (!fix(BBMPC_STRICT_MATH)) hipLaunchKernelGGL(k_cma_sample_roll_small<true>, dim3((N + 63) / 64, G), dim3(256), 0, stream, q, ra.state, ra.fix_q1 ? 1 : 0);
                else hipLaunchKernelGGL(k_cma_sample_roll_small<false>, dim3((N + 63) / 64, G), dim3(256), 0, stream, q, ra.state, ra.fix_q1 ? 1 : 0);
                prof_end();
            } else {
                hipLaunchKernelGGL(k_cma_sample_small, dim3((N + 63) / 64, G), dim3(256), 0, stream, q);
                ra.cand = d_cand_a.p; ra.samples = write_back ? d_cand_a.p : nullptr; ra.rewards = d_rewards.p; ra.penalty_out = nullptr;
                launch_rollout(SRC_BUF, true, ra);                          // clip + penalty (cma_es.py:147-157)
            }
            // (the elite deviations, k x n floats, go through LDS for the covariance sums when they fit beside the static words)
            const size_t yef = (size_t)(n * n + k * n + 2 * n) * sizeof(float) <= 48 * 1024 ? (size_t)(n * n + k * n + 2 * n) : 0;
            const size_t ulds = std::max(std::max(lds, (size_t)n * n * sizeof(float)), yef * sizeof(float));
            want_lds((const void*)k_cma_update_small, ulds, 48 * 1024);
            if (roll_in && it == iters - 1) {
                // held back: finalize() launches it together with the tail of the control step (k_cma_update_final_small)
                want_lds((const void*)k_cma_update_final_small, ulds, 48 * 1024);
                pending_cma_update.set = true; pending_cma_update.q = q; pending_cma_update.lds = ulds; pending_cma_update.yef = (int)yef;
            } else
            hipLaunchKernelGGL(k_cma_update_small, dim3(G), dim3(1024), ulds, stream, q, c_evec.p, c_eval.p, c_info.p, sw.cma_eigh_fail ? 1 : 0, (int)yef);
            HIP_CHECK(hipGetLastError());
        } else {
        const bool mfma_y = n > 128 && (n & 3) == 0;              // k_cma_gemm_y_mfma forms B D itself
        if (!mfma_y) hipLaunchKernelGGL(k_cma_bd, dim3((unsigned)((gnn + 255) / 256)), dim3(256), 0, stream, q);
        want_lds((const void*)k_cma_select, lds, 4096 + 512);       // eidx_s[1024] + the selection's small static words
        // selection + path update of an unsharded population share a launch (both one workgroup per instance)
        const bool merge_sp = !pop_sharded() && n > 128;
        if (merge_sp) want_lds((const void*)k_cma_select_paths, lds, 4096 + 512 + 4 * 512 * 4 + 2 * 4096 + 256);
        // sample -> roll out -> sorted top-k of this handle's particles (part != null: sharded population)
        auto shard_pass = [&](float* part) {
            hipLaunchKernelGGL(k_cma_noise, dim3((N + 255) / 256, HU, A), dim3(256), 0, stream, q);
            if (mfma_y) hipLaunchKernelGGL(k_cma_gemm_y_mfma, dim3((N + 63) / 64, (n + 63) / 64, G), dim3(256), 0, stream, q);
            else hipLaunchKernelGGL(k_cma_gemm_y, dim3((N + 63) / 64, (n + 63) / 64, G), dim3(256), 0, stream, q);
            HIP_CHECK(hipGetLastError());
            // the clipped candidates go back to the buffer only where somebody reads them whole (the parity trace, the sharded
            // selection, user functions): the path update clips the k elites it reads itself, and the write-back is 9.6 MB of
            // strided stores per launch at config 5's shape
            const bool write_back = trace_on || pop_sharded() || user_path();
            ra.cand = d_cand_a.p; ra.samples = write_back ? d_cand_a.p : nullptr; ra.rewards = d_rewards.p; ra.penalty_out = nullptr;
            launch_rollout(SRC_BUF, true, ra);                          // clip + penalty (cma_es.py:147-157)
            if (part || !merge_sp) hipLaunchKernelGGL(k_cma_select, dim3(G), dim3(REFIT_THREADS), lds, stream, q, part);
            HIP_CHECK(hipGetLastError());
        };
        if (pop_sharded()) {
            // population sharded over ranks: every rank samples and rolls out ITS particles, the sorted local
            // elites (reward, global index, candidate) are exchanged and merged (kernels_cma.hpp); the path / covariance
            // update and the eigen-decomposition run replicated on every rank
            const int R = ps_loopback > 1 ? ps_loopback : std::max(1, rc.comm ? rc.nranks : 1);
            const size_t pw = (size_t)G * k * (n + 2);
            if (!ps_part.p || ps_part.n < pw) ps_part.alloc(pw);
            if (ps_all.n < pw * R) ps_all.alloc(pw * R);
            if (ps_loopback > 1) {
                for (int r = 0; r < R; ++r) {
                    q.pop_offset = r * N;
                    shard_pass(ps_all.p + pw * r);
                }
                q.pop_offset = cfg.population_offset;
            } else {
                shard_pass(ps_part.p);
                if (rc.comm) {
                    const Rccl& r = *rc.api;
                    r.check(r.AllGather(ps_part.p, ps_all.p, pw, Rccl::kFloat32, rc.comm, stream), "ncclAllGather (CMA-ES local elites)");
                } else {
                    REQUIRE(cfg.population_global <= N, BBMPC_E_STATE, "population sharding needs a communicator: call bbmpc_comm_init first");
                    HIP_CHECK(hipMemcpyAsync(ps_all.p, ps_part.p, pw * 4, hipMemcpyDeviceToDevice, stream));
                }
            }
            if (trace_on && !c_eidx_glob.p) c_eidx_glob.alloc((size_t)G * k);
            hipLaunchKernelGGL(k_cma_merge, dim3(G), dim3(1024), 0, stream, q, ps_all.p, R, trace_on ? c_eidx_glob.p : (int*)nullptr);
            HIP_CHECK(hipGetLastError());
        } else {
            shard_pass(nullptr);
        }
        if (merge_sp) hipLaunchKernelGGL(k_cma_select_paths, dim3(G), dim3(1024), lds, stream, q);
        else hipLaunchKernelGGL(k_cma_paths, dim3(G), dim3(n > 128 ? 1024 : REFIT_THREADS), 0, stream, q);
        hipLaunchKernelGGL(k_cma_cov, dim3((n + 15) / 16, (n + 15) / 16, G), dim3(16, 16), 0, stream, q);
        HIP_CHECK(hipGetLastError());
        const bool eigh = cma_use_eigh();
        const unsigned* need = eigh ? e_flags.p : nullptr;      // the Jacobi below then runs only for instances the direct solver gave up
        if (eigh) cma_eigh_launch(q);
        if (cma_use_eigh_small()) {
            // one workgroup per instance: direct solver, the Jacobi only for an instance it refuses (kernels_eigh_small.hpp)
            hipLaunchKernelGGL(k_cma_factor_small, dim3(G), dim3(1024), (size_t)n * n * sizeof(float), stream, q, c_evec.p, c_eval.p, c_info.p,
                               sw.cma_eigh_fail ? 1 : 0);
        } else if (n <= 512 && !sw.cma_svd_v1) {
            // warm-started Jacobi, one 1024-thread workgroup per instance (kernels_cma.hpp)
            if (!eigh) HIP_CHECK(hipMemsetAsync(c_sync.p, 0, (size_t)G * CMA_SYNC_WORDS * sizeof(unsigned), stream));      // (with the direct solver: k_eigh_tri_solve clears it)
            if (n > 128 && (n & 3) == 0) hipLaunchKernelGGL(k_cma_warm_mfma, dim3((n + 31) / 32, (n + 63) / 64, G), dim3(256), 0, stream, q, c_evec.p, need);
            else hipLaunchKernelGGL(k_cma_warm, dim3((n + 31) / 32, (n + 31) / 32, G), dim3(256), 0, stream, q, c_evec.p);
            const int bsz = (n + 7) / 8;
            const int ncb = (n + 63) / 64;                               // k_cma_svd_block<ncb, NB>: LDS column pitch 64 * ncb
            // NB = 16 column blocks on 8 workgroups per instance when there are CUs for them (BBMPC_CMA_NB overrides)
            // 20 blocks for 256 < n <= 320: 15-16 columns per block = 16-lane rows of FOUR waves, one per SIMD, in the cross rounds
            // (19 columns are five waves, two of them on one SIMD: a round is instruction-issue bound and takes twice as long)
            const int nb_auto = (ncb == 5 && 80 * ((G + 7) / 8) <= 256) ? 20 : ((64 * ((G + 7) / 8) <= 256 && n >= 256) ? 16 : 8);
            const int nbk = (sw.cma_nb == 8 || sw.cma_nb == 16 || (sw.cma_nb == 20 && ncb == 5)) ? sw.cma_nb : nb_auto;
            const int bsk = (n + nbk - 1) / nbk;
            const size_t blds = (size_t)2 * bsk * 64 * ncb * sizeof(float);
            if (n >= 128 && (n & 3) == 0 && bsz <= 64 && cma_gram_lds_bytes(n) <= 159 * 1024 && cma_gram_wp(n) <= 128 && G * 4 <= 256 &&
                !sw.cma_svd_rounds && sw.cma_svd_gram) {
                // block Jacobi in the Gram domain: Gram matrix / column update on the matrix cores, rotations on 2bs x 2bs data
                ensure_max_lds((const void*)k_cma_svd_gram, 159 * 1024);     // + a few static words
                float* evp = c_evec.p;
                unsigned* syp = c_sync.p;
                int sweeps = 15;
                void* kargs[] = {(void*)&q, (void*)&evp, (void*)&syp, (void*)&sweeps};
                HIP_CHECK(hipLaunchCooperativeKernel((const void*)k_cma_svd_gram, dim3(4, G), dim3(1024), kargs, cma_gram_lds_bytes(n), stream));
            } else if (n >= 128 && (n & 3) == 0 && bsz <= 64 && blds <= 159 * 1024 && 8 * (nbk / 2) * ((G + 7) / 8) <= 256 && !sw.cma_svd_rounds) {
                // block Jacobi: 4 workgroups per instance, block pairs resident in LDS, 7 instance barriers per sweep
                const void* kfn = nullptr;
#define BBMPC_SVD_CASE(NC_) case NC_: kfn = nbk == 16 ? (const void*)k_cma_svd_block<NC_, 16> : (const void*)k_cma_svd_block<NC_, 8>; break;
                if (nbk == 20) kfn = (const void*)k_cma_svd_block<5, 20>;
                else
                switch (ncb) {
                    BBMPC_SVD_CASE(2) BBMPC_SVD_CASE(3) BBMPC_SVD_CASE(4) BBMPC_SVD_CASE(5) BBMPC_SVD_CASE(6) BBMPC_SVD_CASE(7)
                    default: kfn = nbk == 16 ? (const void*)k_cma_svd_block<8, 16> : (const void*)k_cma_svd_block<8, 8>; break;
                }
#undef BBMPC_SVD_CASE
                ensure_max_lds(kfn, 159 * 1024);     // + a few static words
                // The instance barrier spins, so an instance's workgroups must be resident together.  The grid is at most
                // 256 workgroups of one per CU (97 KB of LDS each), i.e. it always fits the idle part of a 256-CU device, and
                // a plain launch has the residency of a cooperative one (MI355X_MICROARCH.md); the cooperative form only
                // adds the launch-time size check -- and 15-19 us of host time per launch during which this thread cannot
                // run ahead of the GPU (five of them per control step: act() 10.5 ms against 9.3 ms device-resident).
                // BBMPC_CMA_COOP=1 brings it back.
                {
                    float* evp = c_evec.p;
                    unsigned* syp = c_sync.p;
                    int sweeps = 15;
                    const unsigned* needp = need;
                    void* kargs[] = {(void*)&q, (void*)&evp, (void*)&syp, (void*)&sweeps, (void*)&needp};
                    // 1-D grid, an instance's four workgroups on one XCD (kernels_cma.hpp); surplus workgroups return at once
                    const dim3 sgrid(8 * (nbk / 2) * ((G + 7) / 8)), sblock(nbk >= 16 ? 512 : 1024);
                    // the plain launch is only as good as a cooperative one while every workgroup finds a CU of its own at once:
                    // on a partition with fewer CUs (CPX / DPX modes, CU masks) the spinning barrier would wait for workgroups
                    // that were never dispatched -- there the cooperative launch, which refuses what does not fit
                    if (sw.cma_coop || (int)sgrid.x > cu_count) HIP_CHECK(hipLaunchCooperativeKernel(kfn, sgrid, sblock, kargs, blds, stream));
                    else HIP_CHECK(hipLaunchKernel(kfn, sgrid, sblock, kargs, blds, stream));
                }
            } else {
                if (n <= 128 && !sw.cma_svd_general) {
                    const int pairs = (n + 1) / 2;
                    if (n <= 64) {
                        hipLaunchKernelGGL(k_cma_svd_small<4>, dim3(G), dim3(64 * ((pairs + 3) / 4)), (size_t)n * n * sizeof(float), stream,
                                           q, c_evec.p, c_sync.p, 15);
                    } else {
                        if ((size_t)n * n * sizeof(float) > 48 * 1024) ensure_max_lds((const void*)k_cma_svd_small<8>, 96 * 1024);
                        hipLaunchKernelGGL(k_cma_svd_small<8>, dim3(G), dim3(64 * ((pairs + 3) / 4)), (size_t)n * n * sizeof(float), stream,
                                           q, c_evec.p, c_sync.p, 15);
                    }
                } else {
                // small n: the matrix fits LDS; a workgroup sized to the number of pairs
                const size_t rl = (size_t)n * n * sizeof(float) <= 64 * 1024 ? (size_t)n * n * sizeof(float) : 0;
                const int rthreads = std::min(1024, std::max(64, 64 * ((n + 1) / 2)));
                hipLaunchKernelGGL(k_cma_svd_rounds, dim3(1, G), dim3(rthreads), rl, stream, q, c_evec.p, c_sync.p, 15,
                                   (int)(rl / sizeof(float)));
                }
            }
            if (n > 128 && n <= 2048) {
                hipLaunchKernelGGL(k_cma_svd_norms, dim3(G), dim3(1024), 0, stream, q, c_evec.p, c_eval.p, c_info.p, need);
                hipLaunchKernelGGL(k_cma_svd_build_b, dim3((n + 31) / 32, (n + 31) / 32, G), dim3(32, 8), 0, stream, q, c_evec.p, c_eval.p, c_info.p, need);
            } else {
                hipLaunchKernelGGL(k_cma_svd_finish, dim3(G), dim3(n > 256 ? 1024 : 256), 0, stream, q, c_evec.p, c_eval.p, c_info.p);
            }
        } else {
            hipLaunchKernelGGL(k_cma_svd, dim3(G), dim3(REFIT_THREADS), 0, stream, q, c_evec.p, c_eval.p, c_info.p, 15);
        }
        HIP_CHECK(hipGetLastError());
        }   // !small3
        if (trace_on) {
            ensure_trace();
            const size_t nr = (size_t)A * Nst, nm = (size_t)A * HU, ns = (size_t)A * HU * Nst;
            HIP_CHECK(hipMemcpyAsync(t_rewards.p + nr * it, d_rewards.p, nr * 4, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(t_mean.p + nm * it, c_m.p, nm * 4, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(t_samples.p + ns * it, d_cand_a.p, ns * 4, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(t_elites.p + (size_t)A * std::max(k, 1) * it, pop_sharded() ? c_eidx_glob.p : c_eidx.p, (size_t)G * k * 4,
                                     hipMemcpyDeviceToDevice, stream));      // (sharded: the GLOBAL particle indices of the elites)
            // the eigen-system this iteration produced (B, D) and the covariance it factorises: parity tests feed the
            // oracle the engine's own (D^2, B) every iteration and check the factorisation's invariants
            if (!t_cma_B.p) {
                t_cma_B.alloc(gnn * std::max(iters, 1));
                t_cma_C.alloc(gnn * std::max(iters, 1));
                t_cma_D.alloc((size_t)G * n * std::max(iters, 1));
            }
            HIP_CHECK(hipMemcpyAsync(t_cma_B.p + gnn * it, c_B.p, gnn * 4, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(t_cma_C.p + gnn * it, c_C.p, gnn * 4, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(t_cma_D.p + (size_t)G * n * it, c_Dd.p, (size_t)G * n * 4, hipMemcpyDeviceToDevice, stream));
            if (!t_cma_stats.p) { t_cma_stats.alloc((size_t)G * 16 * std::max(iters, 1)); t_cma_stats.zero(stream); }
            if (small3 || cma_use_eigh_small()) {
                // (n <= 32: the factorisation kernel keeps no statistics -- include/bbmpc.h)
                HIP_CHECK(hipMemsetAsync(t_cma_stats.p + (size_t)G * 16 * it, 0, (size_t)G * 16 * sizeof(int), stream));
            } else if (n <= 512 && !sw.cma_svd_v1) {
                // rotation counts of the sweeps ([G][CMA_SYNC_WORDS] words -> [G][16]); word 15: did the Jacobi run for the instance
                HIP_CHECK(hipMemcpy2DAsync(t_cma_stats.p + (size_t)G * 16 * it, 16 * sizeof(int), c_sync.p + CMA_SYNC_ROTATIONS,
                                           CMA_SYNC_WORDS * sizeof(unsigned), 15 * sizeof(int), G, hipMemcpyDeviceToDevice, stream));
                if (cma_use_eigh()) HIP_CHECK(hipMemcpy2DAsync(t_cma_stats.p + (size_t)G * 16 * it + 15, 16 * sizeof(int), e_flags.p, 8 * sizeof(unsigned),
                                                     sizeof(int), G, hipMemcpyDeviceToDevice, stream));
                else HIP_CHECK(hipMemcpy2DAsync(t_cma_stats.p + (size_t)G * 16 * it + 15, 16 * sizeof(int), c_sync.p + 1, CMA_SYNC_WORDS * sizeof(unsigned),
                                                sizeof(int), G, hipMemcpyDeviceToDevice, stream));
            }
        }
    }
    if (!pending_cma_update.set)
    hipLaunchKernelGGL(k_take_first, dim3((A * U + 63) / 64), dim3(64), 0, stream, A, HU, U, c_m.p, d_action.p);   // :211-212
    HIP_CHECK(hipGetLastError());
}

void Engine::launch_pending_cma_update(const FinalArgs& fa) {
    PendingCmaUpdate pu = pending_cma_update;
    pending_cma_update.set = false;
    if (tail_flag) tail_attached = true;
    launch_with_tail(*this, k_cma_update_final_small, dim3(cma_G), dim3(1024), pu.lds, pu.q, c_evec.p, c_eval.p, c_info.p, sw.cma_eigh_fail ? 1 : 0, pu.yef,
                     fa, tail_flag, tail_count, tail_value);
    HIP_CHECK(hipGetLastError());
}

// CMA-ES on the analytic pendulum in one launch per control step when the search dimension is small (kernels_fused_cma.hpp)
bool Engine::use_fused_cma() const {
    if (cfg.optimizer != BBMPC_OPT_CMAES || cfg.dynamics != BBMPC_DYN_PENDULUM || cfg.reward != BBMPC_REW_PENDULUM) return false;
    // opt-in: measured no faster than the per-iteration kernels (both are bound by the Jacobi sweeps)
    if (!sw.cma_fused || fused_mode == 0 || trace_on || pop_sharded()) return false;      // the parity trace is captured between the per-iteration kernels
    if (sw.cma_svd_v1 || sw.cma_svd_rounds || sw.cma_svd_general) return false;
    return cma_G == A && cma_n <= 64 && N <= 1024 && k <= 1024;
}

void Engine::optimize_fused_cma(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step) {
    FusedCmaArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.q = cma_args(step, 0u);
    fa.iters = iters; fa.H = H;
    fa.inj = injected(BBMPC_NOISE_NORMAL);
    fa.inj_stride = (size_t)A * HU * Nst;
    fa.evec = c_evec.p; fa.eval = c_eval.p; fa.info = c_info.p;
    fa.eigh_small = cma_use_eigh_small() ? 1 : 0; fa.eigh_fail = sw.cma_eigh_fail ? 1 : 0;
    FinalArgs& fin = fa.fin;
    fin.A = A; fin.U = U; fin.S = S;
    fin.agent_offset = cfg.agent_offset;
    fin.fix_q1 = fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
    fin.fix_q7 = fix(BBMPC_FIX_Q7_EXPL_NOISE_ZERO_MEAN);
    fin.add_noise = add_noise;
    fin.state = d_state_in;
    fin.action = d_action.p;
    fin.lo = d_lo.p; fin.hi = d_hi.p;
    fin.inj = injected(BBMPC_NOISE_EXPLORATION);
    fin.record = d_record_out;
    fin.next_state = d_next_out;
    fin.key = key(step);
    fin.key.q_per_agent = (uint32_t)((U + 3) / 4);
    if (tail_flag) {
        fa.done_flag = tail_flag; fa.done_count = tail_count; fa.done_value = tail_value;
        tail_attached = true;
    }
    const int kp = (k + 3) & ~3;
    const size_t lds = std::max((size_t)(Nst + TOPK_HIST_WORDS + 2 * kp) * 4, (size_t)cma_n * cma_n * 4);
    // (the factorisation's workspace is static LDS: with the dynamic part the kernel is past the 64 KB default)
    ensure_max_lds((const void*)k_fused_cma_pendulum<true>, 32 * 1024);
    ensure_max_lds((const void*)k_fused_cma_pendulum<false>, 32 * 1024);
    REQUIRE(lds <= 32 * 1024, BBMPC_E_UNSUPPORTED, "fused CMA-ES control step: population too large for the workgroup's LDS");
    prof_begin();
    if (!fix(BBMPC_STRICT_MATH)) launch_with_tail(*this, k_fused_cma_pendulum<true>, dim3(cma_G), dim3(1024), lds, fa);
    else launch_with_tail(*this, k_fused_cma_pendulum<false>, dim3(cma_G), dim3(1024), lds, fa);
    HIP_CHECK(hipGetLastError());
    prof_end();
}

}  // namespace bbmpc
