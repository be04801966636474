// Persistent control-step kernels on the analytic pendulum (kernels_fused.hpp, kernels_fused_pso.hpp): the whole
// OptimizerBase.__call__ graph of RandomSearch / CEM / PI2 / SPSA / PSO in one launch -- choice of the instantiation, the
// noise prefetch, the resident (LINGER) hand-off.  A translation unit of its own: the template zoo is instantiated here only.
#define BBMPC_TU_FUSED
#include "engine.hpp"
#include "engine_util.hpp"

#include <atomic>
#include <chrono>

namespace bbmpc {

void bbmpc_tu_fused_upload_tnq(const float2* table) { tnq_upload(table); }

bool Engine::use_fused() const {
    if (cfg.dynamics != BBMPC_DYN_PENDULUM || cfg.reward != BBMPC_REW_PENDULUM) return false;
    if (pop_sharded()) return false;            // the refit is split around a collective: per-iteration kernels
    if (cfg.optimizer == BBMPC_OPT_SPSA) {
        if (iters > FUSED_MAX_SPSA_ITERS) return false;
    } else if (cfg.optimizer != BBMPC_OPT_RANDOM_SEARCH && cfg.optimizer != BBMPC_OPT_CEM && cfg.optimizer != BBMPC_OPT_PI2) {
        return false;
    }
    if (fused_mode == 0) return false;
    if (fused_mode == 1) return true;
    // auto: one workgroup per agent keeps a whole control step in one launch.  When a handful of agents
    // own very large populations the per-iteration kernels spread the rollouts over more CUs instead.
    return (long)N <= 2048 || A >= 64;
}

static void note_inst(Engine& e, int opt, bool samples_lds, bool fastm, int inj, int ilp, bool linger) {
    snprintf(e.dominant_inst, sizeof(e.dominant_inst), "k_fused_pendulum<%d, %s, %s, %d, %d, %s>", opt, samples_lds ? "true" : "false",
             fastm ? "true" : "false", inj, ilp, linger ? "true" : "false");
}

template <int OPT, bool FASTM, int INJ, int ILP>
static void launch_fused4(Engine& e, FusedArgs& fa, int threads, size_t lds_base, size_t lds_samples) {
#ifdef BBMPC_KERNEL_DBG
    const size_t limit = 158 * 1024;   // the debug clocks live in static LDS
#else
    const size_t limit = 160 * 1024;   // all of a CU's LDS
#endif
    if (lds_base + lds_samples <= limit) {
        fa.test_quit_agent = -1;
        if constexpr (INJ == 2 && FASTM && ILP == 1) {
            if (e.linger_launch && e.subset_n == 0 && fa.done_flag && e.tail_event == nullptr) {
                // the resident form: this launch serves the current call and then every workgroup waits for its agent's next
                // request on its own (kernels_fused.hpp)
                auto fl = k_fused_pendulum<OPT, true, FASTM, INJ, ILP, true>;
                note_inst(e, OPT, true, FASTM, INJ, ILP, true);
                ensure_max_lds((const void*)fl, (int)limit);
                for (int a = 0; a < e.A; ++a) {
                    volatile uint32_t* m = e.mbox_host(a);
                    for (int i = 0; i < 13; ++i) m[i] = fa.done_value & 0xffffu;   // no payload word may carry the next request's tag by accident
                    m[15] = fa.done_value;                            // nothing pending (a stale stop word must not end it)
                    *(volatile uint32_t*)e.gone_host(a) = 0u;
                }
                std::atomic_thread_fence(std::memory_order_release);
                fa.done_flag = e.sync_dev(e.ack_host(0));
                fa.mbox = e.sync_dev(e.mbox_host(0));
                fa.gone = e.sync_dev(e.gone_host(0));
                fa.linger_ticks = (unsigned)e.sw.linger_us * 100u;
                fa.test_quit_agent = e.linger_test_quit;
                hipLaunchKernelGGL(fl, dim3(e.A), dim3(threads), lds_base + lds_samples, e.stream, fa);
                HIP_CHECK(hipGetLastError());
                e.resident_alive = true;
                if (!e.mbox_pub.load(std::memory_order_relaxed)) {
                    e.mbox_pub_agents = e.A;
                    e.mbox_pub.store(e.mbox_host(0), std::memory_order_release);
                    note_resident_handle();
                }
                return;
            }
        }
        if (e.subset_n > 0) {
            // the agents whose resident workgroups had left when this control step was posted (Engine::resident_step)
            auto fs = k_fused_pendulum<OPT, true, FASTM, INJ, ILP>;
            note_inst(e, OPT, true, FASTM, INJ, ILP, false);
            ensure_max_lds((const void*)fs, (int)limit);
            fa.amap = reinterpret_cast<const int*>(e.sync_dev(e.amap_host()));
            hipLaunchKernelGGL(fs, dim3(e.subset_n), dim3(threads), lds_base + lds_samples, e.stream, fa);
            HIP_CHECK(hipGetLastError());
            e.subset_n = 0;
            return;
        }
        auto fn = k_fused_pendulum<OPT, true, FASTM, INJ, ILP>;
        note_inst(e, OPT, true, FASTM, INJ, ILP, false);
        ensure_max_lds((const void*)fn, (int)limit);
        launch_with_tail(e, fn, dim3(e.A), dim3(threads), lds_base + lds_samples, fa);
    } else {
        auto fn = k_fused_pendulum<OPT, false, FASTM, INJ, ILP>;
        note_inst(e, OPT, false, FASTM, INJ, ILP, false);
        launch_with_tail(e, fn, dim3(e.A), dim3(threads), lds_base, fa);
    }
    HIP_CHECK(hipGetLastError());
}

template <int OPT>
static void launch_fused(Engine& e, FusedArgs& fa, int ilp, int threads, size_t lds_base, size_t lds_samples, int inj_layout) {
    const bool fastm = !e.fix(BBMPC_STRICT_MATH);
    const int inj = fa.inj == nullptr ? 0 : inj_layout;     // 0 in-kernel Philox, 1 caller-injected, 2 prefetched float4
#define LF(F, I)                                                                              \
    do {                                                                                      \
        if (ilp == 2) launch_fused4<OPT, F, I, 2>(e, fa, threads, lds_base, lds_samples);    \
        else launch_fused4<OPT, F, I, 1>(e, fa, threads, lds_base, lds_samples);             \
    } while (0)
    if (inj == 2) {                                         // ILP = 1 only
        if (fastm) launch_fused4<OPT, true, 2, 1>(e, fa, threads, lds_base, lds_samples);
        else launch_fused4<OPT, false, 2, 1>(e, fa, threads, lds_base, lds_samples);
    } else if (fastm && !inj) LF(true, 0);
    else if (fastm && inj) LF(true, 1);
    else if (!fastm && !inj) LF(false, 0);
    else LF(false, 1);
#undef LF
}

// Standard draws for a chunk of control steps in the layout the persistent kernel's INJ=2 path reads:
// [step][iter][A][Nst][Q] float4, one float4 = the 4 words of Philox block q of particle n.  Same counters and
// transforms as the in-kernel generator (rng.hpp), so the values are bit-identical.  thread = (n, q), coalesced.
__global__ void k_noise_fill(RngKey key, uint32_t rstream, int kind /* 0 trunc normal, 1 uniform, 2 rademacher */, int n_it, int N, int Nst, int A, int Q,
                             int agent_offset, float4* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * Q) return;
    const int n = idx / Q, q = idx % Q;
    const int z = blockIdx.y;
    const int a = z % A, it = (z / A) % n_it, s = z / (A * n_it);
    key.step += (uint32_t)s;
    const U4 w = rng_block(key, rstream, (uint32_t)it, (uint32_t)n, (uint32_t)(agent_offset + a), (uint32_t)(4 * q));
    float4 v;
    if (kind == 1) v = make_float4(word_to_uniform(w.x), word_to_uniform(w.y), word_to_uniform(w.z), word_to_uniform(w.w));
    else if (kind == 2) v = make_float4(word_to_rademacher(w.x), word_to_rademacher(w.y), word_to_rademacher(w.z), word_to_rademacher(w.w));
    else v = make_float4(word_to_trunc_normal(w.x), word_to_trunc_normal(w.y), word_to_trunc_normal(w.z), word_to_trunc_normal(w.w));
    out[(((size_t)s * n_it + it) * A + a) * Nst * Q + (size_t)n * Q + q] = v;
}

void Engine::launch_noise_fill(int64_t chunk, int buf, hipStream_t on) {
    const bool rs = cfg.optimizer == BBMPC_OPT_RANDOM_SEARCH, sp = cfg.optimizer == BBMPC_OPT_SPSA;
    const int n_it = rs ? 1 : iters, Q = (HU + 3) / 4;
    dim3 grid((N * Q + 255) / 256, A * n_it * pf_steps), block(256);
    hipLaunchKernelGGL(k_noise_fill, grid, block, 0, on, key((uint32_t)(chunk * pf_steps)), rs ? 2u : (sp ? 3u : 1u), rs ? 1 : (sp ? 2 : 0), n_it,
                       N, Nst, A, Q, cfg.agent_offset, reinterpret_cast<float4*>(d_noise_pf[buf].p));
    HIP_CHECK(hipGetLastError());
}

void Engine::optimize_fused(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step) {
    FusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.N = N; fa.A = A; fa.H = H; fa.U = U; fa.HU = HU; fa.Nst = Nst; fa.k = k; fa.iters = iters;
    fa.agent_offset = cfg.agent_offset;
    fa.fix_q1 = fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
    fa.fix_q7 = fix(BBMPC_FIX_Q7_EXPL_NOISE_ZERO_MEAN);
    fa.add_noise = add_noise;
    fa.warm_start = fix(BBMPC_FIX_Q2_CEM_WARM_START);
    fa.balance = sw.balance;
    if (tail_flag) {
        fa.done_flag = tail_flag; fa.done_count = tail_count; fa.done_value = tail_value;
        tail_attached = true;
    }
    fa.alpha = cfg.alpha;
    fa.inv_lamda = 1.0f / cfg.lamda;
    fa.state = d_state_in;
    fa.lo = d_lo.p; fa.hi = d_hi.p;
    fa.prev_mean = d_prev_mean.p; fa.var0 = d_var0.p;
    fa.mean_out = d_mean.p; fa.var_out = d_var.p;
    fa.samples_g = d_samples.p;
    fa.inj = injected(cfg.optimizer == BBMPC_OPT_RANDOM_SEARCH ? BBMPC_NOISE_UNIFORM
                      : cfg.optimizer == BBMPC_OPT_SPSA ? BBMPC_NOISE_RADEMACHER : BBMPC_NOISE_TRUNC_NORMAL);
    fa.inj_expl = injected(BBMPC_NOISE_EXPLORATION);
    if (cfg.optimizer == BBMPC_OPT_SPSA) {                                   // gain sequences spsa.py:56,69-70
        const float big_a = (float)iters / 10.0f;
        for (int it = 0; it < iters && it < FUSED_MAX_SPSA_ITERS; ++it) {
            const float tf = (float)it;
            fa.spsa_ak[it] = cfg.spsa_a / (float)pow((double)((tf + 1.0f) + big_a), (double)cfg.spsa_alpha);
            fa.spsa_ck[it] = cfg.spsa_c / (float)pow((double)(tf + 1.0f), (double)cfg.spsa_gamma);
        }
    }
    fa.record = d_record_out;
    fa.next_state = d_next_out;
    if (trace_on) {
        ensure_trace();
        fa.t_rewards = t_rewards.p; fa.t_mean = t_mean.p; fa.t_var = t_var.p; fa.t_elites = t_elites.p; fa.t_samples = t_samples.p;
        if (cfg.optimizer == BBMPC_OPT_SPSA) {
            if (!t_rewards2.p) t_rewards2.alloc((size_t)A * Nst * std::max(iters, 1));
            fa.t_rewards2 = t_rewards2.p;
        }
    }
#ifdef BBMPC_KERNEL_DBG
    static long long* dbg_buf = nullptr;
    if (sw.dbg) {
        if (!dbg_buf) HIP_CHECK(hipHostMalloc((void**)&dbg_buf, 64 * 8, hipHostMallocDefault));
        fa.dbg = dbg_buf;
    }
#endif
    fa.key = key(step);
    // ---- noise prefetch (see engine.hpp): unless the caller injected its own draws.  One fill launch covers a
    // chunk of pf_steps control steps (the host adds one launch + three event calls per chunk, not per step).
    const int pf_nit = cfg.optimizer == BBMPC_OPT_RANDOM_SEARCH ? 1 : iters;
    if (pf_mode < 0) {
        const char* ev = getenv("BBMPC_NOISE_PREFETCH");
        pf_step_floats = (size_t)pf_nit * A * Nst * ((HU + 3) / 4) * 4;
        const size_t budget = (size_t)128 << 20;     // bytes per chunk buffer
        pf_steps = pf_step_floats ? (int)std::min<size_t>(8, budget / (pf_step_floats * 4)) : 0;
        pf_mode = ((ev ? atoi(ev) != 0 : true) && pf_steps >= 1 && U == 1) ? 1 : 0;
        if (pf_mode) {
            {   // its own priority class, hence its own pool of hardware queues: a fill must never sit in the queue
                // behind a resident control-step kernel of the launch stream (normal priority)
                int least = 0, greatest = 0;
                HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIP_CHECK(hipStreamCreateWithPriority(&pf_stream, hipStreamNonBlocking, greatest));
            }
            HIP_CHECK(hipEventCreateWithFlags(&pf_free, hipEventDisableTiming));
            for (int b = 0; b < 2; ++b) {
                d_noise_pf[b].alloc(pf_step_floats * pf_steps);
                HIP_CHECK(hipEventCreateWithFlags(&pf_done[b], hipEventDisableTiming));
            }
        }
    }
    const bool use_pf = pf_mode == 1 && fa.inj == nullptr && pf_nit > 0;
    if (use_pf) {
        const int64_t c = (int64_t)step / pf_steps;
        const int pb = (int)(c & 1), nb = pb ^ 1;
        if (pf_chunk[pb] != c) {                   // not prefetched (first step, or steps did not advance by one): in line
            if (pf_inflight[pb]) HIP_CHECK(hipStreamWaitEvent(stream, pf_done[pb], 0));   // stale fill still running
            launch_noise_fill(c, pb, stream);
            pf_chunk[pb] = c; pf_waited[pb] = true; pf_inflight[pb] = false;
        } else if (!pf_waited[pb]) {
            HIP_CHECK(hipStreamWaitEvent(stream, pf_done[pb], 0));
            pf_waited[pb] = true; pf_inflight[pb] = false;
        }
        fa.inj = d_noise_pf[pb].p + (size_t)((int64_t)step - c * pf_steps) * pf_step_floats;
        if (pf_chunk[nb] != c + 1) {
            // the other buffer was last read by kernels already enqueued on `stream`: the side stream fills it for
            // the next chunk while this chunk's control steps run
            HIP_CHECK(hipEventRecord(pf_free, stream));
            HIP_CHECK(hipStreamWaitEvent(pf_stream, pf_free, 0));
            launch_noise_fill(c + 1, nb, pf_stream);
            HIP_CHECK(hipEventRecord(pf_done[nb], pf_stream));
            pf_chunk[nb] = c + 1; pf_waited[nb] = false; pf_inflight[nb] = true;
        }
    }
    // two trajectories per lane (one wave per SIMD for N <= 512) unless overridden
    int ilp = 1;                 // measured: 2 waves/SIMD x 1 trajectory beats 1 wave/SIMD x 2 trajectories
    ilp = sw.ilp;
    if (use_pf) ilp = 1;
    const int per = (N + ilp - 1) / ilp;
    const int threads = std::min(1024, std::max(((per + 63) / 64) * 64, ((std::max(k, 1) + 63) / 64) * 64));   // top-k needs k <= threads
    const int HUp = (HU + 3) & ~3, kp = (std::max(k, 1) + 3) & ~3;
    const size_t lds_base = (size_t)(Nst + 3 * HUp + kp + 64 + TOPK_HIST_WORDS + 2 * kp) * 4;
    const size_t lds_samples = (size_t)HU * Nst * 4;
    prof_begin();
    switch (cfg.optimizer) {
        case BBMPC_OPT_RANDOM_SEARCH: launch_fused<FOPT_RS>(*this, fa, ilp, threads, lds_base, lds_samples, use_pf ? 2 : 1); break;
        case BBMPC_OPT_CEM: launch_fused<FOPT_CEM>(*this, fa, ilp, threads, lds_base, lds_samples, use_pf ? 2 : 1); break;
        case BBMPC_OPT_SPSA: {
            const bool fastm = !fix(BBMPC_STRICT_MATH);
            if (use_pf) {
                if (fastm) launch_fused4<FOPT_SPSA, true, 2, 1>(*this, fa, threads, lds_base, lds_samples);
                else launch_fused4<FOPT_SPSA, false, 2, 1>(*this, fa, threads, lds_base, lds_samples);
            } else if (fa.inj) {
                if (fastm) launch_fused4<FOPT_SPSA, true, 1, 1>(*this, fa, threads, lds_base, lds_samples);
                else launch_fused4<FOPT_SPSA, false, 1, 1>(*this, fa, threads, lds_base, lds_samples);
            } else {
                if (fastm) launch_fused4<FOPT_SPSA, true, 0, 1>(*this, fa, threads, lds_base, lds_samples);
                else launch_fused4<FOPT_SPSA, false, 0, 1>(*this, fa, threads, lds_base, lds_samples);
            }
            break;
        }
        default: launch_fused<FOPT_PI2>(*this, fa, ilp, threads, lds_base, lds_samples, use_pf ? 2 : 1); break;
    }
    prof_end();
    if (fa.dbg) {
        HIP_CHECK(hipStreamSynchronize(stream));
        if (step == 5) {
            fprintf(stderr, "[dbg] phase clocks (10ns units) rel. to start:");
            for (int i = 0; i <= 1 + iters * 4; ++i) fprintf(stderr, " %lld", fa.dbg[i] - fa.dbg[0]);
            fprintf(stderr, "\n[dbg] iter0 per-wave rollout end:");
            for (int i = 24; i < 32; ++i) fprintf(stderr, " %lld", fa.dbg[i] - fa.dbg[0]);
            fprintf(stderr, "  gather-done %lld stats-done %lld", fa.dbg[32] - fa.dbg[0], fa.dbg[33] - fa.dbg[0]);
            fprintf(stderr, "\n[dbg] iter1 top-k select marks rel. to rollout end:");
            for (int i = 0; i < 10; ++i) fprintf(stderr, " %lld", fa.dbg[48 + i] - fa.dbg[3 + 4]);
            fprintf(stderr, "  (iter1 marks: start %lld rollout-end %lld barrier %lld topk-end %lld)", fa.dbg[5] - fa.dbg[0], fa.dbg[6] - fa.dbg[0], fa.dbg[7] - fa.dbg[0], fa.dbg[8] - fa.dbg[0]);
            fprintf(stderr, "\n[dbg] shader clocks: %lld over %lld wall ticks => %.1f MHz\n", fa.dbg[41] - fa.dbg[40], fa.dbg[1 + iters * 4] - fa.dbg[0], (double)(fa.dbg[41] - fa.dbg[40]) / ((double)(fa.dbg[1 + iters * 4] - fa.dbg[0]) * 0.01));
        }
    }
}

// PSO on the true pendulum model in one launch per control step (kernels_fused_pso.hpp) when the swarm's positions
// and velocities fit one CU's LDS
static size_t fused_pso_lds(int H, int Nst) { return ((size_t)2 * H * Nst + ((H + 3) & ~3) + 16 + 16 + 8) * sizeof(float); }

bool Engine::use_fused_pso() const {
    if (cfg.dynamics != BBMPC_DYN_PENDULUM || cfg.reward != BBMPC_REW_PENDULUM || cfg.optimizer != BBMPC_OPT_PSO || U != 1) return false;
    if (fused_mode == 0 || pop_sharded()) return false;           // a sharded swarm exchanges its bests every iteration
    return N <= 1024 && fused_pso_lds(H, Nst) <= 160 * 1024;
}

void Engine::optimize_fused_pso(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step) {
    FusedPsoArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.N = N; fa.A = A; fa.H = H; fa.Nst = Nst; fa.iters = iters;
    fa.agent_offset = cfg.agent_offset;
    fa.fix_q1 = fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
    fa.fix_q7 = fix(BBMPC_FIX_Q7_EXPL_NOISE_ZERO_MEAN);
    fa.add_noise = add_noise;
    fa.w = cfg.pso_w; fa.c1 = cfg.pso_c1; fa.c2 = cfg.pso_c2; fa.v0frac = cfg.pso_v0_fraction;
    if (tail_flag) {             // the records are complete when this kernel ends (k_pso_seed only re-seeds the swarm)
        fa.done_flag = tail_flag; fa.done_count = tail_count; fa.done_value = tail_value;
        tail_attached = true;
    }
    fa.state = d_state_in;
    fa.lo = d_lo.p; fa.hi = d_hi.p; fa.var0 = d_var0.p;
    fa.s = pso_state();
    fa.inj2 = injected(BBMPC_NOISE_PSO_SCALARS);
    fa.inj_pos = injected(BBMPC_NOISE_PSO_RESEED_TRUNC);
    fa.inj_vel = injected(BBMPC_NOISE_PSO_RESEED_UNIFORM);
    fa.inj_expl = injected(BBMPC_NOISE_EXPLORATION);
    fa.record = d_record_out;
    fa.next_state = d_next_out;
    if (trace_on) {
        ensure_trace();
        fa.t_rewards = t_rewards.p; fa.t_mean = t_mean.p; fa.t_elites = t_elites.p;
        fa.t_elite_stride = std::max(k, 1);
    }
    fa.key = key(step);
    const size_t lds = fused_pso_lds(H, Nst);
    const int threads = std::max(64, ((N + 63) / 64) * 64);
    prof_begin();
    if (!fix(BBMPC_STRICT_MATH)) {
        ensure_max_lds((const void*)k_fused_pso_pendulum<true>, 160 * 1024);
        hipLaunchKernelGGL(k_fused_pso_pendulum<true>, dim3(A), dim3(threads), lds, stream, fa);
    } else {
        ensure_max_lds((const void*)k_fused_pso_pendulum<false>, 160 * 1024);
        hipLaunchKernelGGL(k_fused_pso_pendulum<false>, dim3(A), dim3(threads), lds, stream, fa);
    }
    HIP_CHECK(hipGetLastError());
    prof_end();
    const OptArgs oa = opt_args(step, 0u);
    hipLaunchKernelGGL(k_pso_seed, dim3((N + 255) / 256, HU, A), dim3(256), 0, stream, oa, fa.s, d_var0.p, cfg.pso_v0_fraction, 0,
                       fa.inj_pos, fa.inj_vel);                                                                         // :116-138
    HIP_CHECK(hipGetLastError());
}

}  // namespace bbmpc
