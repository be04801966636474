// C ABI (include/bbmpc.h) + engine implementation.  gfx950 only, no CPU fallback.
#include "engine.hpp"
#include "engine_util.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <functional>
#include <chrono>
#include <mutex>
#include <set>
#include <thread>
#include <utility>

namespace bbmpc {

// every live handle of the process (see Engine::mbox_pub)
static std::mutex g_engines_mu;
static std::set<Engine*> g_engines;
static std::atomic<int> g_resident_handles{0};
void note_resident_handle() { g_resident_handles.fetch_add(1, std::memory_order_relaxed); }   // (bbmpc_fused.hip publishes a mailbox)

// Called at every entry point: resident workgroups of OTHER handles on this device are asked to leave (a few stores into
// pinned memory, no synchronisation); costs one relaxed load when nothing is resident.
void stop_foreign_residents(Engine* self) {
    const int mine = (self && self->mbox_pub.load(std::memory_order_relaxed)) ? 1 : 0;
    if (g_resident_handles.load(std::memory_order_relaxed) - mine <= 0) return;
    std::lock_guard<std::mutex> lock(g_engines_mu);
    for (Engine* o : g_engines) {
        if (o == self || (self && o->device != self->device)) continue;
        uint32_t* m = o->mbox_pub.load(std::memory_order_acquire);
        if (!m) continue;
        for (int a = 0; a < o->mbox_pub_agents; ++a) ((volatile uint32_t*)m)[a * 16 + 15] = 0xffffffffu;   // the stop word
        std::atomic_thread_fence(std::memory_order_release);
    }
}

static thread_local std::string g_last_error;

// ------------------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------------------
// truncated-normal quantile table (rng.hpp): q(i/2048) in float64 -> fp32, uploaded once per device
static double erfinv_d(double y) {
    // Newton on erf from a Giles-type start; converges to double precision in a few steps for |y| < 0.96
    double w = -log((1.0 - y) * (1.0 + y)), x;
    if (w < 5.0) {
        w -= 2.5;
        x = 2.81022636e-08; x = 3.43273939e-07 + x * w; x = -3.5233877e-06 + x * w; x = -4.39150654e-06 + x * w;
        x = 0.00021858087 + x * w; x = -0.00125372503 + x * w; x = -0.00417768164 + x * w; x = 0.246640727 + x * w;
        x = 1.50140941 + x * w;
    } else {
        w = sqrt(w) - 3.0;
        x = -0.000200214257; x = 0.000100950558 + x * w; x = 0.00134934322 + x * w; x = -0.00367342844 + x * w;
        x = 0.00573950773 + x * w; x = -0.0076224613 + x * w; x = 0.00943887047 + x * w; x = 1.00167406 + x * w;
        x = 2.83297682 + x * w;
    }
    x *= y;
    for (int it = 0; it < 4; ++it) x -= (erf(x) - y) / (1.1283791670955126 * exp(-x * x));
    return x;
}

static void init_tnq_table() {
    // per-device symbol, uploaded once; handles may be created from several threads (a thread-per-GPU driver)
    static std::mutex mu;
    static std::set<int> done;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count(dev)) return;
    std::vector<float> q(TNQ_SIZE + 1);
    const double p2 = erf(sqrt(2.0));                       // P(|z| < 2)
    for (int i = 0; i <= TNQ_SIZE; ++i) q[i] = (float)(sqrt(2.0) * erfinv_d((2.0 * i / TNQ_SIZE - 1.0) * p2));
    q[0] = -2.0f;
    q[TNQ_SIZE] = 2.0f;
    std::vector<float2> tab(TNQ_SIZE);
    for (int i = 0; i < TNQ_SIZE; ++i) tab[i] = make_float2(q[i], q[i + 1] - q[i]);
    tnq_upload(tab.data());                                  // this translation unit's copy ...
    bbmpc_tu_cma_upload_tnq(tab.data());                     // ... and the other two
    bbmpc_tu_mlp_upload_tnq(tab.data());
    bbmpc_tu_fused_upload_tnq(tab.data());
    done.insert(dev);                                       // only after the upload succeeded
}

// "lo-hi" in the environment variable `name` -> a stream whose workgroups only go to CUs lo .. hi (hipExtStreamCreateWithCUMask,
// bit i of the mask = CU i), or null when the variable is not set
hipStream_t masked_stream_from_env(const char* name, int cu_count) {
    const char* v = getenv(name);
    int lo = 0, hi = 0;
    if (!v || sscanf(v, "%d-%d", &lo, &hi) != 2 || lo < 0 || hi < lo || hi >= cu_count) return nullptr;
    std::vector<uint32_t> mask((size_t)(cu_count + 31) / 32, 0u);
    for (int c = lo; c <= hi; ++c) mask[(size_t)c >> 5] |= 1u << (c & 31);
    hipStream_t st = nullptr;
    HIP_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    return st;
}

Engine::Engine(const bbmpc_config& c) : cfg(c) {
    REQUIRE(c.abi_version == BBMPC_ABI_VERSION, BBMPC_E_INVALID, "bbmpc_config.abi_version mismatch");
    stop_foreign_residents(nullptr);     // creation allocates and copies: nothing should wait behind a lingering kernel
    N = c.population_size; A = c.num_agents; H = c.planning_horizon; U = c.dim_u; S = c.dim_s;
    iters = c.optimizer == BBMPC_OPT_RANDOM_SEARCH ? 1 : c.max_iterations;
    if (c.optimizer == BBMPC_OPT_NONE) iters = 0;
    k = c.num_elite;
    REQUIRE(A >= 1 && H >= 1 && U >= 1 && S >= 1, BBMPC_E_INVALID, "num_agents, planning_horizon, dim_u, dim_s must be >= 1");
    REQUIRE(c.action_low && c.action_high, BBMPC_E_INVALID, "action_low/action_high are required");
    REQUIRE(c.optimizer >= BBMPC_OPT_NONE && c.optimizer <= BBMPC_OPT_SPSA, BBMPC_E_INVALID, "unknown optimizer");
    REQUIRE(c.dynamics == BBMPC_DYN_PENDULUM || c.dynamics == BBMPC_DYN_MLP || c.dynamics == BBMPC_DYN_USER, BBMPC_E_INVALID, "unknown dynamics kind");
    REQUIRE(c.reward == BBMPC_REW_PENDULUM || c.reward == BBMPC_REW_CHEETAH || c.reward == BBMPC_REW_USER, BBMPC_E_INVALID, "unknown reward kind");
    if (c.dynamics == BBMPC_DYN_USER || c.reward == BBMPC_REW_USER)
        REQUIRE(S <= 256 && U <= 256, BBMPC_E_UNSUPPORTED, "user device functions: dim_s, dim_u <= 256 (per-row arrays live in registers)");
    if (c.dynamics == BBMPC_DYN_PENDULUM)
        REQUIRE(S == 3 && U == 1, BBMPC_E_INVALID, "PendulumTrueModel needs dim_s == 3 and dim_u == 1");
    if (c.reward == BBMPC_REW_PENDULUM) REQUIRE(S >= 3, BBMPC_E_INVALID, "pendulum reward needs dim_s >= 3");
    if (c.reward == BBMPC_REW_CHEETAH) REQUIRE(S >= 18, BBMPC_E_INVALID, "cheetah reward indexes state[17]: dim_s >= 18");
    if (c.optimizer != BBMPC_OPT_NONE) {
        REQUIRE(N >= 1, BBMPC_E_INVALID, "population_size must be >= 1");
        REQUIRE(iters >= 0, BBMPC_E_INVALID, "max_iterations must be >= 0");
        if (N > 32768) {
            // The refit kernels keep an agent's rewards in one CU's LDS (32768 floats).  A larger population is played as G
            // equal shards of the population-sharding machinery on this one GPU: shard r = particles
            // [r N/G, (r+1) N/G), draws keyed by the global particle, one merge per iteration -- the loopback hook, switched on
            // by the size.  The parity hooks (trace, injected noise, per-particle state) stay per shard and are refused.
            REQUIRE(c.population_global == 0 && c.population_offset == 0, BBMPC_E_UNSUPPORTED,
                    "population_size > 32768 per rank: give every rank at most 32768 particles of the sharded population");
            int g = (N + 32767) / 32768;
            while (g <= 64 && N % g != 0) ++g;
            REQUIRE(g <= 64, BBMPC_E_UNSUPPORTED, "population_size > 32768 must divide into at most 64 equal shards of at most 32768 particles");
            if (c.optimizer == BBMPC_OPT_CMAES) REQUIRE(k <= N / g, BBMPC_E_UNSUPPORTED, "num_elite must not exceed a shard of the population");
            auto_split = g;
            cfg.population_global = N;
            N /= g;
        }
    } else {
        N = 0;
    }
    if (c.optimizer == BBMPC_OPT_CEM || c.optimizer == BBMPC_OPT_CMAES)
        REQUIRE(k >= 1 && k <= std::max(N, (int)c.population_global), BBMPC_E_INVALID, "num_elite must be in [1, population_size]");
    if (c.optimizer == BBMPC_OPT_CEM) REQUIRE(k <= 1024, BBMPC_E_UNSUPPORTED, "num_elite > 1024 not supported");
    if (c.optimizer == BBMPC_OPT_CMAES) {
        const bool per_agent = (c.quirks & BBMPC_CMAES_PER_AGENT) != 0;
        const long nn = per_agent ? (long)H * U : (long)A * H * U;
        REQUIRE(nn <= 4096, BBMPC_E_UNSUPPORTED,
                "CMA-ES joint dimension num_agents*H*U > 4096: use BBMPC_CMAES_PER_AGENT (the reference's own "
                "[N,n,n] map_fn stack cannot run at this size either, cma_es.py:13)");
        REQUIRE(per_agent || c.num_agents_global <= c.num_agents, BBMPC_E_UNSUPPORTED,
                "coupled CMA-ES sums rewards over ALL agents and does not shard; use BBMPC_CMAES_PER_AGENT");
        REQUIRE(k <= 1024, BBMPC_E_UNSUPPORTED, "num_elite > 1024 not supported");
    }

    if (c.population_global != 0 || c.population_offset != 0) {
        REQUIRE(c.population_global >= N && c.population_offset >= 0 && c.population_offset + N <= c.population_global, BBMPC_E_INVALID,
                "population_offset / population_global: this handle's particles must lie inside the global population");
        if (c.population_global > N) {
            REQUIRE(c.optimizer != BBMPC_OPT_NONE, BBMPC_E_UNSUPPORTED, "population sharding needs an optimizer (evaluate-only handles roll out what they are given)");
            if (c.optimizer == BBMPC_OPT_CMAES)
                REQUIRE(k <= N && k <= 1024, BBMPC_E_UNSUPPORTED, "sharded CMA-ES: num_elite must not exceed this rank's share of the population (nor 1024)");
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw HipError(BBMPC_E_NO_DEVICE, "no HIP device available: this library has no CPU fallback");
    if (c.device >= 0) {
        REQUIRE(c.device < ndev, BBMPC_E_NO_DEVICE, "bbmpc_config.device out of range");
        HIP_CHECK(hipSetDevice(c.device));        // bbmpc_create restores the caller's device (DeviceGuard)
    }
    HIP_CHECK(hipGetDevice(&device));
    HIP_CHECK(hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, device));
    {   // control steps are latency-critical, and a resident control-step kernel must not hold back unrelated work of the
        // process: hardware queues are pooled per priority, so the handle's streams live in the high-priority pool, away
        // from PyTorch's and the caller's normal-priority streams (the handles among themselves: stop_foreign_residents)
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // experiment switch (profiles/r5_cfg5cma_overlap.md): BBMPC_STREAM_CUS=lo-hi confines the handle's stream to those CUs
        if (!(own_stream = masked_stream_from_env("BBMPC_STREAM_CUS", cu_count)))
            HIP_CHECK(hipStreamCreateWithPriority(&own_stream, hipStreamNonBlocking, greatest));
    }
    stream = own_stream;
    init_tnq_table();

    if (const char* fm = getenv("BBMPC_FUSED")) fused_mode = atoi(fm);
    {
        auto flag = [](const char* n) { return getenv(n) != nullptr; };
        auto ival = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
        sw.cma_svd_v1 = flag("BBMPC_CMA_SVD_V1"); sw.cma_svd_rounds = flag("BBMPC_CMA_SVD_ROUNDS");
        sw.cma_svd_general = flag("BBMPC_CMA_SVD_GENERAL");
        sw.cma_svd_gram = flag("BBMPC_CMA_SVD_GRAM");
        sw.cma_coop = flag("BBMPC_CMA_COOP"); sw.cma_nb = ival("BBMPC_CMA_NB", 0);
        sw.cma_eigh = ival("BBMPC_CMA_EIGH", 1); sw.cma_eigh_fail = flag("BBMPC_CMA_EIGH_FAIL");
        sw.cma_fused = flag("BBMPC_CMA_FUSED");
        sw.mlp_generic = flag("BBMPC_MLP_GENERIC");
        { const int b = ival("BBMPC_MLP_BF16", 0); sw.mlp_bf16 = (b == 1 || b == 3) ? b : 0; }
        sw.mlp_pair = ival("BBMPC_MLP_PAIR", -1); sw.mlp_q4 = ival("BBMPC_MLP_Q4", -1); sw.mlp_q4s = ival("BBMPC_MLP_Q4S", ival("BBMPC_MLP_Q4R", 1)); sw.mlp_w4 = ival("BBMPC_MLP_W4", 1); sw.pi2_skip_init = ival("BBMPC_PI2_SKIP_INIT", 1); sw.step_graph = ival("BBMPC_STEP_GRAPH", 1); sw.cma_small3 = ival("BBMPC_CMA_SMALL3", 1);
        sw.refit_wgs = ival("BBMPC_REFIT_WGS", 0);
        sw.mlp_wave = ival("BBMPC_MLP_WAVE", 1);
        sw.linger_us = std::max(0, ival("BBMPC_LINGER_US", 200));
        linger_test_quit = ival("BBMPC_LINGER_TEST_QUIT", 0) - 1;
        sw.balance = ival("BBMPC_BALANCE", 0);      // (1: SIMD mates pace each other in the fused pendulum rollout -- paid while a model step took 160 ns, costs 1-2 % at 60)
        sw.ilp = ival("BBMPC_ILP", 1) == 2 ? 2 : 1;
        sw.refit_v1 = flag("BBMPC_REFIT_V1");
        sw.zero_copy = !flag("BBMPC_NO_ZERO_COPY");
        sw.host_poll = !flag("BBMPC_NO_HOST_POLL");
        sw.mlp_no_half_tail = flag("BBMPC_MLP_NO_HALF_TAIL");
        sw.dbg = flag("BBMPC_DBG");
        user_stepwise_only = flag("BBMPC_USER_STEPWISE");
        if (c.optimizer != BBMPC_OPT_NONE) {
            ps_loopback = auto_split > 1 ? auto_split : ival("BBMPC_POPSHARD_LOOPBACK", 0);
            ps_force = flag("BBMPC_POPSHARD_FORCE");
        }
    }
    HU = H * U;
    Nst = ((std::max(N, 1) + 63) / 64) * 64;
    rec = U + S + 1;
    lo.assign(c.action_low, c.action_low + U);
    hi.assign(c.action_high, c.action_high + U);
    upload(d_lo, lo);
    upload(d_hi, hi);
    d_state.alloc((size_t)A * S);
    d_record.alloc((size_t)A * rec);
    d_action.alloc((size_t)A * U);
    d_action.zero(stream);
    if (c.optimizer != BBMPC_OPT_NONE) {
        // mean = (lo+hi)/2, var = (lo-hi)^2/16 tiled to [A,H,U]   cem.py:55-72, pi2.py:44-55
        std::vector<float> m((size_t)A * HU), v((size_t)A * HU);
        for (int i = 0; i < A * HU; ++i) {
            const int u = i % U;
            m[i] = (lo[u] + hi[u]) / 2.0f;
            const float d = lo[u] - hi[u];
            v[i] = (d * d) / 16.0f;
        }
        upload(d_prev_mean, m);
        upload(d_var0, v);
        d_mean.alloc(m.size());
        d_var.alloc(m.size());
        d_sigma.alloc(m.size());
        d_samples.alloc((size_t)A * HU * Nst);
        d_rewards.alloc((size_t)A * Nst);
        d_penalty.alloc((size_t)A * Nst);
        d_elites.alloc((size_t)A * std::max(k, 1));
        const size_t big = (size_t)A * HU * Nst;
        if (c.optimizer == BBMPC_OPT_SPSA) {
            d_cand_a.alloc(big);
            d_cand_b.alloc(big);
            d_rewards2.alloc((size_t)A * Nst);
        }
        if (c.optimizer == BBMPC_OPT_CMAES) {
            d_cand_a.alloc(big);
            cma_init();
        }
        if (c.optimizer == BBMPC_OPT_PSO) {
            // constructor state of the reference: every Variable zero (pso.py:50-59), quirk Q4
            const size_t gl = ps_loopback > 1 ? (size_t)ps_loopback : 1;      // the loopback hook keeps every shard's swarm
            d_cand_a.alloc(big * gl); d_vel.alloc(big * gl); d_pbest.alloc(big * gl);
            d_pbest_r.alloc((size_t)A * Nst * gl); d_cond.alloc((size_t)A * Nst * gl);
            d_gbest.alloc((size_t)A * HU); d_gbest_r.alloc((size_t)A); d_gidx.alloc((size_t)A);
            d_cand_a.zero(stream); d_vel.zero(stream); d_pbest.zero(stream); d_pbest_r.zero(stream);
            d_cond.zero(stream); d_gbest.zero(stream); d_gbest_r.zero(stream); d_gidx.zero(stream);
        }
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    // registered last: a constructor that throws never runs the destructor
    { std::lock_guard<std::mutex> lock(g_engines_mu); g_engines.insert(this); }
}

Engine::~Engine() {
    try { resident_stop(); } catch (...) {}
    if (step_graph) { (void)hipGraphExecDestroy(step_graph); step_graph = nullptr; }
    if (step_words) { (void)hipHostFree(step_words); step_words = nullptr; }
    if (step_words_dev) { (void)hipFree(step_words_dev); step_words_dev = nullptr; }
    { std::lock_guard<std::mutex> lock(g_engines_mu); g_engines.erase(this); }
    if (lazy_sync && stream) (void)hipStreamSynchronize(stream);
    if (own_stream) (void)hipStreamSynchronize(own_stream);
    rc.destroy();
    if (eigh_side) {
        (void)hipStreamSynchronize(eigh_side);
        (void)hipStreamDestroy(eigh_side);
        for (int i = 0; i < 2; ++i) if (eigh_ev[i]) (void)hipEventDestroy(eigh_ev[i]);
    }
    if (pf_stream) {
        (void)hipStreamSynchronize(pf_stream);
        (void)hipStreamDestroy(pf_stream);
        for (int b = 0; b < 2; ++b)
            if (pf_done[b]) (void)hipEventDestroy(pf_done[b]);
        if (pf_free) (void)hipEventDestroy(pf_free);
    }
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    user_reward.release();
    user_dynamics.release();
    user_rollout.release();
    if (h_pin) (void)hipHostFree(h_pin);
    for (auto*& hs : h_record_stage) { if (hs) (void)hipHostFree(hs); hs = nullptr; }
    if (host_done) (void)hipHostFree(host_done);
    if (host_count) (void)hipFree(host_count);
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

void Engine::settle() {
    if (settle_hook) settle_hook(this);          // staged records of bbmpc_optimize_gather go to the collective first
    resident_stop();
    if (lazy_sync) {
        lazy_sync = false;
        HIP_CHECK(hipStreamSynchronize(stream));
    }
}

static void resident_unpublish(Engine* e);

// The resident workgroups' side of this is at the end of k_fused_pendulum.  Returns false when the call has to go through
// a launch after all: the workgroups left (the stream is idle then), this step's noise chunk is not there yet, or -- a
// request that crossed some workgroups' exit -- only part of the agents were served: subset_n / amap_host() then name the
// rest and the launch that follows covers exactly those.
bool Engine::resident_step(const float* state, int add_noise, uint32_t seq) {
    {   // all workgroups gone already (linger time over, or another handle asked them to leave): nothing to stop or wait for
        bool all_gone = true;
        for (int a = 0; a < A && all_gone; ++a) all_gone = *(volatile const uint32_t*)gone_host(a) != 0u;
        if (all_gone) {
            resident_unpublish(this);
            resident_alive = false;
            return false;
        }
    }
    const uint32_t step = step_counter;
    const int64_t c = (int64_t)step / std::max(pf_steps, 1);
    const int pb = (int)(c & 1), nb = pb ^ 1;
    if (pf_mode != 1 || pf_chunk[pb] != c) { resident_stop(); return false; }
    if (!pf_waited[pb]) {
        if (hipEventQuery(pf_done[pb]) != hipSuccess) { resident_stop(); return false; }
        pf_waited[pb] = true; pf_inflight[pb] = false;
    }
    const float* inj = d_noise_pf[pb].p + (size_t)((int64_t)step - c * pf_steps) * pf_step_floats;
    ++step_counter;
    const uint64_t ip = (uint64_t)(uintptr_t)inj;
    // One request line per agent.  Every payload word carries 16 bits of payload and the low 16 bits of the request's
    // sequence number, word 15 the whole sequence number: the kernel accepts a line only when all fourteen words name
    // the request it waits for, so nothing depends on the 64-byte line being read (or written) in one piece or in order.
    for (int a = 0; a < A; ++a) {
        volatile uint32_t* m = mbox_host(a);
        uint32_t sw3[3];
        memcpy(sw3, state + (size_t)a * 3, 12);
        const uint32_t tag = seq & 0xffffu;
        const uint16_t half[13] = {(uint16_t)step, (uint16_t)(step >> 16), (uint16_t)(add_noise != 0),
                                   (uint16_t)ip, (uint16_t)(ip >> 16), (uint16_t)(ip >> 32), (uint16_t)(ip >> 48),
                                   (uint16_t)sw3[0], (uint16_t)(sw3[0] >> 16), (uint16_t)sw3[1], (uint16_t)(sw3[1] >> 16),
                                   (uint16_t)sw3[2], (uint16_t)(sw3[2] >> 16)};
        for (int i = 0; i < 13; ++i) m[i] = ((uint32_t)half[i] << 16) | tag;
        m[15] = seq;
    }
    std::atomic_thread_fence(std::memory_order_release);
    // host work that hides under the kernel: the next chunk's noise, the previous call's collective
    if (pf_chunk[nb] != c + 1) {
        // the other buffer held chunk c-1: every control step that read it has handed its record to this thread already
        launch_noise_fill(c + 1, nb, pf_stream);
        HIP_CHECK(hipEventRecord(pf_done[nb], pf_stream));
        pf_chunk[nb] = c + 1; pf_waited[nb] = false; pf_inflight[nb] = true;
    }
    if (in_flight_hook && !in_flight_called) { in_flight_called = true; in_flight_hook(this); }
    const auto t0 = std::chrono::steady_clock::now();
    bool all = true;
    for (int a = 0; a < A; ++a) {
        volatile const uint32_t* ack = ack_host(a);
        volatile const uint32_t* gone = gone_host(a);
        uint32_t spins = 0;
        for (;;) {
            if (*ack == seq) break;
            if (*gone != 0u) { all = false; break; }               // it left (before or after this request?)
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { all = false; break; }
        }
    }
    if (all) return true;
    // some workgroups have left: end the others too (they finish the step they are in), then see who is missing
    resident_stop();
    int missing = 0;
    int32_t* amap = amap_host();
    for (int a = 0; a < A; ++a)
        if (*(volatile const uint32_t*)ack_host(a) != seq) amap[missing++] = a;
    if (missing == 0) return true;                                // everybody took the request on the way out
    --step_counter;                                               // the launch that follows does this step ...
    subset_n = missing < A ? missing : 0;                         // ... for the agents that were not served (all of them: an ordinary launch)
    return false;
}

static void resident_unpublish(Engine* e) {
    if (e->mbox_pub.load(std::memory_order_relaxed)) {
        e->mbox_pub.store(nullptr, std::memory_order_release);
        g_resident_handles.fetch_sub(1, std::memory_order_relaxed);
    }
}

void Engine::resident_stop() {
    if (!resident_alive) return;
    resident_unpublish(this);
    for (int a = 0; a < A; ++a) mbox_host(a)[15] = 0xffffffffu;   // the stop word (word 15 alone decides)
    std::atomic_thread_fence(std::memory_order_release);
    resident_alive = false;
    HIP_CHECK(hipStreamSynchronize(stream));
}

float* Engine::pinned(size_t count) {
    if (count > h_pin_n) {
        invalidate_step_graph();             // a captured control step reads and writes the old buffer through h_pin_dev
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        h_pin_n = 0;
        HIP_CHECK(hipHostMalloc((void**)&h_pin, count * sizeof(float), hipHostMallocCoherent | hipHostMallocMapped));
        h_pin_n = count;
        HIP_CHECK(hipHostGetDevicePointer((void**)&h_pin_dev, h_pin, 0));
    }
    return h_pin;
}

OptArgs Engine::opt_args(uint32_t step, uint32_t iter) const {
    OptArgs o;
    o.N = N; o.A = A; o.H = H; o.U = U; o.HU = HU; o.Nst = Nst;
    o.agent_offset = cfg.agent_offset;
    o.lo = d_lo.p; o.hi = d_hi.p;
    o.key = key(step);
    o.iter = iter;
    o.pop_offset = cfg.population_offset;
    return o;
}

PsoState Engine::pso_state(int shard) {
    PsoState s;
    const size_t big = (size_t)A * HU * Nst * shard, small = (size_t)A * Nst * shard;
    s.pos = d_cand_a.p + big; s.vel = d_vel.p + big; s.pbest = d_pbest.p + big; s.pbest_r = d_pbest_r.p + small;
    s.gbest = d_gbest.p; s.gbest_r = d_gbest_r.p; s.cond = d_cond.p + small; s.gidx = d_gidx.p;
    return s;
}


void Engine::reset() {
    // CEM/PI2/SPSA reset(): previous solution <- bounds midpoint (cem.py:138-149, pi2.py:98-105)
    if (cfg.optimizer == BBMPC_OPT_NONE || cfg.optimizer == BBMPC_OPT_RANDOM_SEARCH) return;
    cem_sigma0_ready = false;                      // (prev_mean is rewritten below)
    if (cfg.optimizer == BBMPC_OPT_CMAES) {          // restores m and sigma only (cma_es.py:215-227)
        HIP_CHECK(hipStreamSynchronize(stream));
        cma_reset_mean_sigma();
        return;
    }
    if (cfg.optimizer == BBMPC_OPT_PSO) {
        // PSOOptimizer.reset(): uniform positions / velocities, pbest = pos, rewards -inf  (pso.py:143-160)
        OptArgs oa = opt_args(step_counter, 0xFFFFu);
        for (int r = 0; r < (ps_loopback > 1 ? ps_loopback : 1); ++r) {
            if (ps_loopback > 1) oa.pop_offset = r * N;
            hipLaunchKernelGGL(k_pso_seed, dim3((N + 255) / 256, HU, A), dim3(256), 0, stream, oa, pso_state(r), d_var0.p,
                               cfg.pso_v0_fraction, 1, injected(BBMPC_NOISE_PSO_RESET_POS), injected(BBMPC_NOISE_PSO_RESET_VEL));
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(stream));
        return;
    }
    std::vector<float> m((size_t)A * HU);
    for (int i = 0; i < A * HU; ++i) m[i] = (lo[i % U] + hi[i % U]) / 2.0f;
    HIP_CHECK(hipMemcpyAsync(d_prev_mean.p, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

// ------------------------------------------------------------------------------------------------
// layout helpers (host)
// ------------------------------------------------------------------------------------------------
void Engine::to_internal(const float* ref, int n_pop, float* out) const {
    // [n,A,H,U] -> [A][HU][Nst]
    for (int n = 0; n < n_pop; ++n)
        for (int a = 0; a < A; ++a)
            for (int j = 0; j < HU; ++j) out[((size_t)a * HU + j) * Nst + n] = ref[((size_t)n * A + a) * HU + j];
}
void Engine::from_internal(const float* in, int n_pop, float* ref) const {
    for (int n = 0; n < n_pop; ++n)
        for (int a = 0; a < A; ++a)
            for (int j = 0; j < HU; ++j) ref[((size_t)n * A + a) * HU + j] = in[((size_t)a * HU + j) * Nst + n];
}

// ------------------------------------------------------------------------------------------------
// side streams
// ------------------------------------------------------------------------------------------------
// HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and a process that also runs
// PyTorch / c10d easily holds more streams than that.  If the communication stream lands on the launch stream's queue
// its cross-stream wait serialises the two (record all-gather, config 2: 53.7 -> 70 us per control step, depending only
// on how many streams the process happened to create before).  Queues are pooled per priority, so the communication
// stream is created with the highest priority -- a latency-critical, tiny collective: 56-57 us whatever else the process
// creates.  (Do not combine with GPU_MAX_HW_QUEUES=8: measured 112 us.)  Round 2: ALL of a handle's streams (launch,
// noise prefetch, communication) are in that pool, three streams for its four queues, so that the resident control-step
// kernel never holds back PyTorch's or the caller's normal-priority work; between handles see stop_foreign_residents.
hipStream_t create_comm_stream() {
    int least = 0, greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
    return s;
}

// ------------------------------------------------------------------------------------------------
// profiling (HIP events on the launch stream around the dominant kernel)
// ------------------------------------------------------------------------------------------------
void Engine::prof_begin() {
    prof_this = profiling && (prof_seq++ % (uint64_t)prof_every) == 0;
    if (!prof_this) return;
    if (ev_used + 2 > ev_pool.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
    }
    HIP_CHECK(hipEventRecord(ev_pool[ev_used], stream));
}
void Engine::prof_end() {
    if (!prof_this) return;
    HIP_CHECK(hipEventRecord(ev_pool[ev_used + 1], stream));
    ev_used += 2;
}
void Engine::get_profile(double* ms, int64_t* launches) {
    HIP_CHECK(hipStreamSynchronize(stream));
    double tot = 0.0;
    for (size_t i = 0; i + 1 < ev_used; i += 2) {
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, ev_pool[i], ev_pool[i + 1]));
        tot += t;
    }
    *ms = tot;
    *launches = (int64_t)(ev_used / 2);
    ev_used = 0;
}


// ------------------------------------------------------------------------------------------------
// user device functions (rtc.hpp) + the step-wise evaluator (kernels_user.hpp)
// ------------------------------------------------------------------------------------------------
void Engine::set_user_source(int kind, const char* src) {
    REQUIRE(src && *src, BBMPC_E_INVALID, "empty HIP source");
    if (kind == USER_KIND_REWARD) REQUIRE(cfg.reward == BBMPC_REW_USER, BBMPC_E_STATE, "handle was not created with BBMPC_REW_USER");
    else REQUIRE(cfg.dynamics == BBMPC_DYN_USER, BBMPC_E_STATE, "handle was not created with BBMPC_DYN_USER");
    std::vector<char> code;
    try {
        code = compile_user_program(src, kind, S, U);
    } catch (const std::exception& ex) {
        throw HipError(BBMPC_E_INVALID, ex.what());
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    UserFunction& f = kind == USER_KIND_REWARD ? user_reward : user_dynamics;
    f.release();
    HIP_CHECK(hipModuleLoadData(&f.module, code.data()));
    HIP_CHECK(hipModuleGetFunction(&f.fn, f.module, kind == USER_KIND_REWARD ? "bbmpc_user_reward_rows" : "bbmpc_user_dynamics_rows"));
    if (kind == USER_KIND_REWARD) HIP_CHECK(hipModuleGetFunction(&f.fn_traj, f.module, "bbmpc_user_reward_traj"));
    f.source = src;
    f.cb = nullptr; f.cb_user = nullptr;
    user_rollout_stale = true;
}

void Engine::set_user_callback(int kind, bbmpc_rows_callback fn, void* user) {
    if (kind == USER_KIND_REWARD) REQUIRE(cfg.reward == BBMPC_REW_USER, BBMPC_E_STATE, "handle was not created with BBMPC_REW_USER");
    else REQUIRE(cfg.dynamics == BBMPC_DYN_USER, BBMPC_E_STATE, "handle was not created with BBMPC_DYN_USER");
    HIP_CHECK(hipStreamSynchronize(stream));
    UserFunction& f = kind == USER_KIND_REWARD ? user_reward : user_dynamics;
    if (fn) { f.release(); f.source.clear(); }
    f.cb = fn;
    f.cb_user = fn ? user : nullptr;
    user_rollout_stale = true;
}

// total (+)= the rewards a callback wrote for one batch of rows
__global__ void k_rows_accumulate(const float* __restrict__ r, int batch, float* __restrict__ total, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch) total[i] = accumulate ? total[i] + r[i] : r[i];
}

// next = process_output(state, dynamics(process_input(state, action)))  on [batch] rows   deterministic.py:79-103
void Engine::dynamics_rows(const float* d_states, const float* d_actions, int astride, int batch, float* d_next) {
    if (cfg.dynamics == BBMPC_DYN_USER && user_dynamics.cb) {
        const float* acts_c = d_actions;
        if (astride != U) {                               // the callback sees a dense [batch, U] block
            if (d_step_act.n < (size_t)batch * U) d_step_act.alloc((size_t)batch * U);
            HIP_CHECK(hipMemcpy2DAsync(d_step_act.p, (size_t)U * 4, d_actions, (size_t)astride * 4, (size_t)U * 4, batch,
                                       hipMemcpyDeviceToDevice, stream));
            acts_c = d_step_act.p;
        }
        if (user_dynamics.cb(user_dynamics.cb_user, d_states, acts_c, nullptr, batch, d_next, (void*)stream) != 0)
            throw HipError(BBMPC_E_INVALID, "the dynamics callback reported an error");
        return;
    }
    if (cfg.dynamics == BBMPC_DYN_USER) {
        REQUIRE(user_dynamics.fn, BBMPC_E_STATE, "user dynamics: call bbmpc_set_dynamics_source (or bbmpc_set_dynamics_callback) before computing");
        void* args[] = {(void*)&d_states, (void*)&d_actions, (void*)&astride, (void*)&batch, (void*)&d_next};
        HIP_CHECK(hipModuleLaunchKernel(user_dynamics.fn, (unsigned)((batch + 255) / 256), 1, 1, 256, 1, 1, 0, stream, args, nullptr));
        return;
    }
    if (cfg.dynamics == BBMPC_DYN_PENDULUM) {
        hipLaunchKernelGGL(k_step_pendulum, dim3((batch + 63) / 64), dim3(64), 0, stream, d_states, d_actions, astride, batch,
                           (int)fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER), d_next, (float*)nullptr);
        HIP_CHECK(hipGetLastError());
        return;
    }
    step_dev(d_states, d_actions, astride, batch, d_next, nullptr);        // learned model (its built-in reward kind is REW_NONE here)
}

// total (+)= reward_function(cur, actions, next) on [batch] rows   deterministic.py:65-66, 105-127
void Engine::reward_rows(const float* d_cur, const float* d_next, const float* d_actions, int astride, int batch, float* d_total,
                         int accumulate) {
    if (cfg.reward == BBMPC_REW_USER && user_reward.cb) {
        const float* acts_c = d_actions;
        if (astride != U) {
            if (d_step_act.n < (size_t)batch * U) d_step_act.alloc((size_t)batch * U);
            HIP_CHECK(hipMemcpy2DAsync(d_step_act.p, (size_t)U * 4, d_actions, (size_t)astride * 4, (size_t)U * 4, batch,
                                       hipMemcpyDeviceToDevice, stream));
            acts_c = d_step_act.p;
        }
        if (u_cb_rew.n < (size_t)batch) u_cb_rew.alloc((size_t)batch);
        if (user_reward.cb(user_reward.cb_user, d_cur, acts_c, d_next, batch, u_cb_rew.p, (void*)stream) != 0)
            throw HipError(BBMPC_E_INVALID, "the reward callback reported an error");
        hipLaunchKernelGGL(k_rows_accumulate, dim3((batch + 255) / 256), dim3(256), 0, stream, u_cb_rew.p, batch, d_total, accumulate);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (cfg.reward == BBMPC_REW_USER) {
        REQUIRE(user_reward.fn, BBMPC_E_STATE, "user reward: call bbmpc_set_reward_source (or bbmpc_set_reward_callback) before computing");
        void* args[] = {(void*)&d_cur, (void*)&d_next, (void*)&d_actions, (void*)&astride, (void*)&batch, (void*)&d_total, (void*)&accumulate};
        HIP_CHECK(hipModuleLaunchKernel(user_reward.fn, (unsigned)((batch + 255) / 256), 1, 1, 256, 1, 1, 0, stream, args, nullptr));
        return;
    }
    hipLaunchKernelGGL(k_reward_rows_acc, dim3((batch + 255) / 256), dim3(256), 0, stream, d_cur, d_next, d_actions, astride, batch, S, U,
                       (int)cfg.reward, (int)fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER), d_total, accumulate);
    HIP_CHECK(hipGetLastError());
}

// DeterministicTrajectoryEvaluator.__call__ one planning step at a time (kernels_user.hpp)
void Engine::rollout_stepwise(int mode, bool pen, RolloutArgs& ra) {
    const int n_pop = ra.n_pop, Hh = ra.H, HUh = ra.HU;
    const size_t B = (size_t)A * n_pop;
    if (u_rows.n < (size_t)Hh * B * U) u_rows.alloc((size_t)Hh * B * U);
    if (u_x0.n < B * S) { u_x0.alloc(B * S); u_x1.alloc(B * S); }
    if (u_total.n < B) { u_total.alloc(B); u_pen.alloc(B); }
    dim3 grid((n_pop + 255) / 256, A), block(256);
    if (mode == SRC_UNIFORM || mode == SRC_TRUNC) {
        // the draws the fused kernels make on the fly: candidates of this iteration into the sample buffer
        REQUIRE(ra.samples, BBMPC_E_STATE, "step-wise rollout: no sample buffer");
        if (mode == SRC_UNIFORM) hipLaunchKernelGGL(k_gen_candidates<SRC_UNIFORM>, grid, block, 0, stream, ra);
        else hipLaunchKernelGGL(k_gen_candidates<SRC_TRUNC>, grid, block, (size_t)2 * HUh * sizeof(float), stream, ra);
        HIP_CHECK(hipGetLastError());
    }
    RowsArgs rw;
    memset(&rw, 0, sizeof(rw));
    rw.n_pop = n_pop; rw.A = A; rw.H = Hh; rw.U = U; rw.S = S; rw.HU = HUh; rw.Nst = ra.Nst;
    rw.from_ref = mode == SRC_REF ? 1 : 0;
    rw.pen = pen ? 1 : 0;
    rw.seq = ra.seq;
    rw.cand = mode == SRC_BUF ? ra.cand : ra.samples;
    rw.samples = (mode == SRC_BUF && pen) ? ra.samples : nullptr;       // the feasible candidates go back (PSO / SPSA / CMA-ES / PI2)
    if (mode == SRC_TRUNC && pen) rw.samples = ra.samples;
    rw.lo = ra.lo; rw.hi = ra.hi;
    rw.state = ra.state;
    rw.rows = u_rows.p; rw.x0 = u_x0.p; rw.penalty = u_pen.p;
    prof_begin();
    hipLaunchKernelGGL(k_rows_prepare, grid, block, 0, stream, rw);
    HIP_CHECK(hipGetLastError());
    float* cur = u_x0.p;
    float* nxt = u_x1.p;
    for (int t = 0; t < Hh; ++t) {
        const float* acts = u_rows.p + (size_t)t * B * U;
        dynamics_rows(cur, acts, U, (int)B, nxt);
        reward_rows(cur, nxt, acts, U, (int)B, u_total.p, t > 0 ? 1 : 0);
        std::swap(cur, nxt);
    }
    hipLaunchKernelGGL(k_rows_finish, grid, block, 0, stream, n_pop, A, ra.Nst, pen ? 1 : 0, u_total.p, u_pen.p, ra.rewards, ra.penalty_out);
    HIP_CHECK(hipGetLastError());
    prof_end();
}

// The fused form for analytic models: one lane per trajectory, user function(s) inlined next to the engine's own
// model / rewards (rtc.hpp user_rollout_source).  Compiled on first use, after both sources are known.
void Engine::rollout_user_fused(int mode, bool pen, RolloutArgs& ra) {
    if (user_rollout_stale || !user_rollout.fn) {
        if (cfg.reward == BBMPC_REW_USER) REQUIRE(user_reward.fn, BBMPC_E_STATE, "user reward: call bbmpc_set_reward_source before computing");
        if (cfg.dynamics == BBMPC_DYN_USER) REQUIRE(user_dynamics.fn, BBMPC_E_STATE, "user dynamics: call bbmpc_set_dynamics_source before computing");
        std::vector<char> code;
        try {
            code = compile_user_rollout(cfg.reward == BBMPC_REW_USER ? user_reward.source : std::string(),
                                        cfg.dynamics == BBMPC_DYN_USER ? user_dynamics.source : std::string(), cfg.dynamics, cfg.reward, S, U);
        } catch (const std::exception& ex) {
            throw HipError(BBMPC_E_INVALID, ex.what());
        }
        user_rollout.release();
        HIP_CHECK(hipModuleLoadData(&user_rollout.module, code.data()));
        HIP_CHECK(hipModuleGetFunction(&user_rollout.fn, user_rollout.module, "bbmpc_user_rollout"));
        user_rollout_stale = false;
    }
    int n_pop = ra.n_pop, Aa = A, Hh = ra.H, Nst_ = ra.Nst;
    dim3 ggrid((n_pop + 255) / 256, A), gblock(256);
    if (mode == SRC_UNIFORM || mode == SRC_TRUNC) {
        REQUIRE(ra.samples, BBMPC_E_STATE, "user rollout: no sample buffer");
        if (mode == SRC_UNIFORM) hipLaunchKernelGGL(k_gen_candidates<SRC_UNIFORM>, ggrid, gblock, 0, stream, ra);
        else hipLaunchKernelGGL(k_gen_candidates<SRC_TRUNC>, ggrid, gblock, (size_t)2 * ra.HU * sizeof(float), stream, ra);
        HIP_CHECK(hipGetLastError());
    }
    int from_ref = mode == SRC_REF ? 1 : 0, ipen = pen ? 1 : 0, fq1 = (int)fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
    const float* state = ra.state;
    const float* seq = ra.seq;
    const float* cand = mode == SRC_BUF ? ra.cand : ra.samples;
    float* samples = (pen && mode != SRC_REF) ? ra.samples : nullptr;          // the feasible candidates go back
    const float* lo_ = ra.lo;
    const float* hi_ = ra.hi;
    float* rewards = ra.rewards;
    float* penalty_out = ra.penalty_out;
    void* args[] = {&n_pop, &Aa, &Hh, &Nst_, &from_ref, &ipen, &fq1, &state, &seq, &cand, &samples, &lo_, &hi_, &rewards, &penalty_out};
    // few trajectories -> one wave per workgroup (latency); many -> 256-thread workgroups
    const unsigned bs = ((long)n_pop * A <= 16384) ? 64 : 256;
    prof_begin();
    HIP_CHECK(hipModuleLaunchKernel(user_rollout.fn, (unsigned)((n_pop + bs - 1) / bs), (unsigned)A, 1, bs, 1, 1, 0, stream, args, nullptr));
    prof_end();
}

// Learned MLP + user reward: the whole-horizon MFMA rollout (16-particle tiles) records the state after every step,
// then ONE launch of the user's function scores every trajectory -- 2 launches instead of 2*H + 2.
void Engine::rollout_mlp_user_reward(int mode, bool pen, RolloutArgs& ra) {
    REQUIRE(user_reward.fn_traj, BBMPC_E_STATE, "user reward: call bbmpc_set_reward_source before computing");
    const size_t need = (size_t)ra.H * A * ra.Nst * S;
    if (u_traj.n < need) u_traj.alloc(need);
    mlp_traj_out = u_traj.p;
    try {
        launch_rollout_mlp(mode, pen, ra, false, nullptr);          // reward kind REW_NONE: leaves -(penalty) in ra.rewards
    } catch (...) {
        mlp_traj_out = nullptr;
        throw;
    }
    mlp_traj_out = nullptr;
    int n_pop = ra.n_pop, Aa = A, Hh = ra.H, Nst_ = ra.Nst, from_ref = mode == SRC_REF ? 1 : 0;
    const float* state = ra.state;
    const float* traj = u_traj.p;
    const float* seq = ra.seq;
    const float* cand = mode == SRC_BUF ? ra.cand : ra.samples;
    float* rewards = ra.rewards;
    if (mode != SRC_REF) REQUIRE(cand, BBMPC_E_STATE, "user reward over the learned model: no candidate buffer");
    void* args[] = {&n_pop, &Aa, &Hh, &Nst_, &from_ref, &state, &traj, &seq, &cand, &rewards};
    HIP_CHECK(hipModuleLaunchKernel(user_reward.fn_traj, (unsigned)((n_pop + 255) / 256), (unsigned)A, 1, 256, 1, 1, 0, stream, args, nullptr));
}

// DeterministicMLP.__call__ on already-processed rows (deterministic_mlp.py:27-51)
__global__ __launch_bounds__(TAIL_THREADS) void k_rows_mlp_raw(RowMlp net, const float* x_in, float* out) {
    __shared__ float x[192];
    __shared__ float bufA[TAIL_MAXW], bufB[TAIL_MAXW], part[(TAIL_THREADS / 64) * TAIL_MAXW];
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int K = net.m.dims[0], M = net.m.dims[net.m.n_layers];
    for (int i = tid; i < K; i += nthr) x[i] = x_in[(size_t)b * K + i];
    __syncthreads();
    const float* raw = row_mlp_forward(net, x, bufA, bufB, part, tid, nthr);
    for (int i = tid; i < M; i += nthr) out[(size_t)b * M + i] = raw[i];
}

void Engine::mlp_forward_rows(const float* d_x, int batch, float* d_out) {
    REQUIRE(cfg.dynamics == BBMPC_DYN_MLP && mlp_ready, BBMPC_E_STATE, "bbmpc_mlp_forward: needs a learned-dynamics handle with weights set");
    hipLaunchKernelGGL(k_rows_mlp_raw, dim3(batch), dim3(TAIL_THREADS), 0, stream, row_mlp(), d_x, d_out);
    HIP_CHECK(hipGetLastError());
}

void Engine::launch_rollout(int mode, bool pen, RolloutArgs& ra) {
    if (user_path()) {
        if (user_callbacks()) {                           // a host callback per planning step: step-wise only
            dominant_kernel = "stepwise(user callback)";
            rollout_stepwise(mode, pen, ra);
        } else if (cfg.dynamics != BBMPC_DYN_MLP && !user_stepwise_only) {
            dominant_kernel = "bbmpc_user_rollout(hiprtc)";
            rollout_user_fused(mode, pen, ra);
        } else if (cfg.dynamics == BBMPC_DYN_MLP && !user_stepwise_only) {
            dominant_kernel = "k_rollout_mlp";
            rollout_mlp_user_reward(mode, pen, ra);
        } else {
            dominant_kernel = "stepwise(user device function)";
            rollout_stepwise(mode, pen, ra);
        }
        return;
    }
    if (cfg.dynamics == BBMPC_DYN_MLP) {
        dominant_kernel = "k_rollout_mlp";
        launch_rollout_mlp(mode, pen, ra, false, nullptr);
        return;
    }
    // few trajectories -> one wave per workgroup so every wave gets its own SIMD (latency);
    // many -> 256-thread workgroups.
    const int bs = ((long)ra.n_pop * A <= 16384) ? 64 : 256;
    dim3 grid((ra.n_pop + bs - 1) / bs, A), block(bs);
    prof_begin();
    const bool fm = !fix(BBMPC_STRICT_MATH);
    const size_t lds_ms = (size_t)2 * HU * sizeof(float);
#define LAUNCH_ROLL(M, P, L)                                                                               \
    do {                                                                                                   \
        if (fm) hipLaunchKernelGGL((k_rollout_pendulum<M, P, true>), grid, block, (L), stream, ra);       \
        else hipLaunchKernelGGL((k_rollout_pendulum<M, P, false>), grid, block, (L), stream, ra);         \
    } while (0)
    if (mode == SRC_REF) LAUNCH_ROLL(SRC_REF, false, (size_t)bs * 33 * sizeof(float));
    else if (mode == SRC_UNIFORM) LAUNCH_ROLL(SRC_UNIFORM, false, 0);
    else if (mode == SRC_TRUNC && !pen) LAUNCH_ROLL(SRC_TRUNC, false, lds_ms);
    else if (mode == SRC_TRUNC && pen) LAUNCH_ROLL(SRC_TRUNC, true, lds_ms);
    else if (mode == SRC_BUF) LAUNCH_ROLL(SRC_BUF, true, 0);
    else throw HipError(BBMPC_E_INVALID, "bad rollout mode");
#undef LAUNCH_ROLL
    HIP_CHECK(hipGetLastError());
    prof_end();
}

void Engine::ensure_trace() {
    const size_t nr = (size_t)A * Nst, nm = (size_t)A * HU, ns = (size_t)A * HU * Nst, ne = (size_t)A * std::max(k, 1);
    const int nit = std::max(iters, 1);
    if (!t_rewards.p) {
        t_rewards.alloc(nr * nit);
        t_mean.alloc(nm * nit);
        t_var.alloc(nm * nit);
        t_samples.alloc(ns * nit);
        t_elites.alloc(ne * nit);
    }
}

void Engine::capture_trace(int it) {
    if (!trace_on) return;
    const size_t nr = (size_t)A * Nst, nm = (size_t)A * HU, ns = (size_t)A * HU * Nst, ne = (size_t)A * std::max(k, 1);
    ensure_trace();
    HIP_CHECK(hipMemcpyAsync(t_rewards.p + nr * it, d_rewards.p, nr * 4, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(t_mean.p + nm * it, d_mean.p, nm * 4, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(t_var.p + nm * it, d_var.p, nm * 4, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(t_samples.p + ns * it, d_samples.p, ns * 4, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(t_elites.p + ne * it, d_elites.p, ne * 4, hipMemcpyDeviceToDevice, stream));
}


void Engine::finalize(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step) {
    FinalArgs fa;
    fa.A = A; fa.U = U; fa.S = S;
    fa.agent_offset = cfg.agent_offset;
    fa.fix_q1 = fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
    fa.fix_q7 = fix(BBMPC_FIX_Q7_EXPL_NOISE_ZERO_MEAN);
    fa.add_noise = add_noise;
    fa.state = d_state_in;
    fa.action = d_action.p;
    fa.lo = d_lo.p; fa.hi = d_hi.p;
    fa.inj = injected(BBMPC_NOISE_EXPLORATION);
    fa.record = d_record_out;
    fa.next_state = d_next_out;
    fa.key = key(step);
    fa.key.q_per_agent = (uint32_t)((U + 3) / 4);
    if (pending_cma_update.set) {                     // (analytic pendulum, CMA-ES at n <= 32: bbmpc_cma.hip)
        launch_pending_cma_update(fa);
        return;
    }
    if (cfg.dynamics == BBMPC_DYN_PENDULUM && !user_path()) {
        if (tail_flag) tail_attached = true;         // the record is complete when this kernel ends: it publishes the sequence number
        hipLaunchKernelGGL(k_finalize_pendulum, dim3((A + 63) / 64), dim3(64), 0, stream, fa, tail_flag, tail_count, tail_value);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (user_path()) {
        // a user device function is involved: exploration noise, one batched model step + reward on the [A] rows, pack
        if (add_noise) {
            hipLaunchKernelGGL(k_explore, dim3((A * U + 63) / 64), dim3(64), 0, stream, fa, d_action.p);
            HIP_CHECK(hipGetLastError());
        }
        if (!d_fin_next.p) {
            d_fin_next.alloc((size_t)A * S);
            d_fin_rew.alloc((size_t)((A + 63) / 64) * 64);
        }
        step_dev(d_state_in, d_action.p, U, A, d_fin_next.p, d_fin_rew.p);
        hipLaunchKernelGGL(k_pack_record, dim3((A * rec + 63) / 64), dim3(64), 0, stream, A, U, S, d_action.p, d_fin_next.p,
                           d_fin_rew.p, d_record_out, d_next_out);
        HIP_CHECK(hipGetLastError());
        if (pending_warm == 1) hipLaunchKernelGGL(k_shift_left, dim3((A * HU + 255) / 256), dim3(256), 0, stream, A, H, U, d_mean.p, d_prev_mean.p);
        else if (pending_warm == 2) HIP_CHECK(hipMemcpyAsync(d_prev_mean.p, d_mean.p, (size_t)A * HU * 4, hipMemcpyDeviceToDevice, stream));
        HIP_CHECK(hipGetLastError());
        pending_warm = 0;
        return;
    }
    // learned dynamics: exploration noise, one model step on the [A] rows, reward, packed record and the warm start of
    // the next control step in ONE launch (kernels_tail.hpp; was k_explore + k_step_mlp + k_pack_record + k_shift_left)
    REQUIRE(mlp_ready, BBMPC_E_STATE, "learned dynamics: call bbmpc_set_mlp before computing");
    TailArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.f = fa;
    ta.net = row_mlp();
    ta.reward_kind = builtin_reward_kind();
    ta.warm_mode = pending_warm;
    ta.H = H; ta.HU = HU;
    ta.mean = d_mean.p;
    ta.prev_mean = d_prev_mean.p;
    pending_warm = 0;
    if (tail_flag) {             // the record is complete when this kernel ends: it publishes the sequence number itself
        ta.done_flag = tail_flag; ta.done_count = tail_count; ta.done_value = tail_value;
        ta.step_words = step_capturing ? step_words_dev : nullptr;
        tail_attached = true;
    }
    launch_with_tail(*this, k_tail_mlp, dim3(A), dim3(TAIL_THREADS), 0, ta);
    HIP_CHECK(hipGetLastError());
}

RowMlp Engine::row_mlp() const {
    RowMlp r;
    memset(&r, 0, sizeof(r));
    r.m = mlp;
    for (int l = 0; l < mlp.n_layers; ++l) { r.wraw[l] = d_wraw[l].p; r.braw[l] = d_braw[l].p; }
    return r;
}


void Engine::optimize_dev(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out) {
    REQUIRE(cfg.optimizer != BBMPC_OPT_NONE, BBMPC_E_STATE, "handle was created without an optimizer");
    const uint32_t step = step_counter++;
    pending_cma_update.set = false;
    if (use_fused()) {
        dominant_kernel = "k_fused_pendulum";
        optimize_fused(d_state_in, add_noise, d_record_out, d_next_out, step);
        return;
    }
    if (use_fused_pso()) {
        dominant_kernel = "k_fused_pso_pendulum";
        optimize_fused_pso(d_state_in, add_noise, d_record_out, d_next_out, step);
        return;
    }
    if (use_fused_cma()) {
        dominant_kernel = "k_fused_cma_pendulum";
        optimize_fused_cma(d_state_in, add_noise, d_record_out, d_next_out, step);
        return;
    }
    dominant_kernel = "k_rollout_pendulum";
    RolloutArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.n_pop = N; ra.A = A; ra.H = H; ra.U = U; ra.S = S; ra.HU = HU; ra.Nst = Nst;
    ra.agent_offset = cfg.agent_offset;
    ra.fix_q1 = fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
    ra.reward_kind = builtin_reward_kind();
    ra.state = d_state_in;
    ra.mean = d_mean.p; ra.sigma = d_sigma.p;
    ra.lo = d_lo.p; ra.hi = d_hi.p;
    ra.samples = d_samples.p;
    ra.rewards = d_rewards.p;
    ra.penalty_out = d_penalty.p;
    ra.key = key(step);
    ra.pop_offset = cfg.population_offset;

    RefitArgs rf;
    memset(&rf, 0, sizeof(rf));
    rf.N = N; rf.A = A; rf.H = H; rf.U = U; rf.HU = HU; rf.Nst = Nst; rf.k = k;
    rf.alpha = cfg.alpha;
    rf.inv_lamda = 1.0f / cfg.lamda;
    rf.rewards = d_rewards.p; rf.samples = d_samples.p;
    rf.lo = d_lo.p; rf.hi = d_hi.p;
    rf.mean = d_mean.p; rf.var = d_var.p; rf.sigma = d_sigma.p;
    rf.elites = d_elites.p; rf.action = d_action.p;

    const int nelem = A * HU;
    const size_t inj_stride = (size_t)A * HU * Nst;
    switch (cfg.optimizer) {
        case BBMPC_OPT_RANDOM_SEARCH: {
            ra.stream = BBMPC_NOISE_UNIFORM; ra.iter = 0;
            ra.inj = injected(BBMPC_NOISE_UNIFORM);
            if (pop_sharded()) {
                // population sharded over ranks: local first maximum, one exchange, first maximum by global index
                const int G = ps_loopback > 1 ? ps_loopback : std::max(1, rc.comm ? rc.nranks : 1);
                const size_t pw = (size_t)A * (U + 2);
                if (!ps_part.p || ps_part.n < pw) ps_part.alloc(pw);
                if (ps_all.n < pw * G) ps_all.alloc(pw * G);
                if (ps_loopback > 1) {
                    for (int r = 0; r < G; ++r) {
                        ra.pop_offset = r * N;
                        launch_rollout(SRC_UNIFORM, false, ra);
                        hipLaunchKernelGGL(k_refit_argmax, dim3(A), dim3(REFIT_THREADS), 0, stream, rf, ps_all.p + pw * r, r * N);
                    }
                    ra.pop_offset = cfg.population_offset;
                } else {
                    launch_rollout(SRC_UNIFORM, false, ra);
                    hipLaunchKernelGGL(k_refit_argmax, dim3(A), dim3(REFIT_THREADS), 0, stream, rf, ps_part.p, (int)cfg.population_offset);
                    if (rc.comm) {
                        const Rccl& r = *rc.api;
                        r.check(r.AllGather(ps_part.p, ps_all.p, pw, Rccl::kFloat32, rc.comm, stream), "ncclAllGather (RandomSearch local bests)");
                    } else {
                        REQUIRE(cfg.population_global <= N, BBMPC_E_STATE, "population sharding needs a communicator: call bbmpc_comm_init first");
                        HIP_CHECK(hipMemcpyAsync(ps_all.p, ps_part.p, pw * 4, hipMemcpyDeviceToDevice, stream));
                    }
                }
                HIP_CHECK(hipGetLastError());
                hipLaunchKernelGGL(k_argmax_merge, dim3(A), dim3(64), 0, stream, rf, ps_all.p, G);
                HIP_CHECK(hipGetLastError());
                capture_trace(0);
                break;
            }
            launch_rollout(SRC_UNIFORM, false, ra);
            hipLaunchKernelGGL(k_refit_argmax, dim3(A), dim3(REFIT_THREADS), 0, stream, rf, (float*)nullptr, 0);
            HIP_CHECK(hipGetLastError());
            capture_trace(0);
            break;
        }
        case BBMPC_OPT_CEM: {
            // Learned model on the quad kernel, the reference's restart from the constructor distribution every control step
            // (no BBMPC_FIX_Q2_CEM_WARM_START): prev_mean, var0 and the sigma they give are constants, so from the second
            // control step on k_dist_init is skipped as for PI2 below -- the first rollout samples from (prev_mean, sigma0)
            // and reads the state from the pinned buffer, the first refit smooths against (prev_mean, var0)
            const bool cem_skip = sw.pi2_skip_init && cfg.dynamics == BBMPC_DYN_MLP && !user_path() && !pop_sharded() && !trace_on &&
                                  iters >= 1 && k <= 64 && !sw.refit_v1 && !fix(BBMPC_FIX_Q2_CEM_WARM_START) && cem_sigma0_ready &&
                                  pi2_copy_seen && stage_state_src != nullptr;
            const float* cem_pinned = nullptr;
            last_step_steady = cem_skip;
            if (cem_skip) {
                cem_pinned = stage_state_src;
            } else {
                hipLaunchKernelGGL(k_dist_init, dim3((nelem + 255) / 256), dim3(256), 0, stream, A, HU, U, d_lo.p, d_hi.p,
                                   d_prev_mean.p, d_var0.p, d_mean.p, d_var.p, d_sigma.p, 1, stage_state_src, d_state.p, A * S);
                if (!cem_sigma0_ready && !fix(BBMPC_FIX_Q2_CEM_WARM_START)) {
                    if (!step_capturing) invalidate_step_graph();   // (a replayed control step samples from d_sigma0: never capture across its allocation)
                    d_sigma0.alloc(nelem);
                    HIP_CHECK(hipMemcpyAsync(d_sigma0.p, d_sigma.p, (size_t)nelem * 4, hipMemcpyDeviceToDevice, stream));
                    cem_sigma0_ready = true;
                }
            }
            stage_state_src = nullptr;
            // iters == 0: action = mean[:,0] of the untouched distribution (otherwise the last refit writes it)
            if (iters == 0)
                HIP_CHECK(hipMemcpy2DAsync(d_action.p, U * 4, d_prev_mean.p, HU * 4, U * 4, A, hipMemcpyDeviceToDevice, stream));
            const float* inj_t = injected(BBMPC_NOISE_TRUNC_NORMAL);
            // LDS budget for the refit: rewards + elite idx + elite tile
            const int kpad = (k + 3) & ~3;
            const int fixed = Nst + kpad + TOPK_HIST_WORDS + 2 * kpad;
            const int budget = fixed * 4 > 48 * 1024 ? 158 * 1024 / 4 : 62 * 1024 / 4;     // floats; big populations take the whole LDS
            int JC = (int)((size_t)std::max(budget - fixed, k) / (size_t)k);
            JC = std::max(1, std::min(JC, HU));
            const size_t lds = (size_t)(fixed + (size_t)k * JC) * 4;
            want_lds((const void*)k_refit_cem_v2, (size_t)fixed * 4);
            want_lds((const void*)k_refit_cem, lds);
            for (int it = 0; it < iters; ++it) {
                ra.stream = BBMPC_NOISE_TRUNC_NORMAL; ra.iter = (uint32_t)it;
                ra.inj = inj_t ? inj_t + inj_stride * it : nullptr;
                if (pop_sharded()) {
                    // population sharded over ranks: local top-k + rows, one exchange, global top-k + refit
                    const int G = ps_loopback > 1 ? ps_loopback : std::max(1, rc.comm ? rc.nranks : 1);
                    const size_t pw = (size_t)A * k * (HU + 2);
                    if (!ps_part.p) ps_part.alloc(pw);
                    if (ps_all.n < pw * G) ps_all.alloc(pw * G);
                    const size_t tl = (size_t)fixed * 4;
                    want_lds((const void*)k_cem_local_topk, tl);
                    const int psg = sw.refit_wgs > 0 ? sw.refit_wgs : std::max(1, std::min(8, HU / 16));   // workgroups per agent (kernels_tail.hpp)
                    RefitArgs rfs = rf;
                    if (!trace_on) rfs.elites = nullptr;
                    if (ps_loopback > 1) {
                        for (int r = 0; r < G; ++r) {
                            ra.pop_offset = r * N;
                            launch_rollout(SRC_TRUNC, false, ra);
                            hipLaunchKernelGGL(k_cem_local_topk, dim3(psg, A), dim3(1024), tl, stream, rf, r * N, ps_all.p + pw * r);
                        }
                        ra.pop_offset = cfg.population_offset;
                    } else {
                        launch_rollout(SRC_TRUNC, false, ra);
                        hipLaunchKernelGGL(k_cem_local_topk, dim3(psg, A), dim3(1024), tl, stream, rf, (int)cfg.population_offset, ps_part.p);
                        if (rc.comm) {
                            const Rccl& r = *rc.api;
                            r.check(r.AllGather(ps_part.p, ps_all.p, pw, Rccl::kFloat32, rc.comm, stream), "ncclAllGather (CEM local elites)");
                        } else {
                            REQUIRE(cfg.population_global <= N, BBMPC_E_STATE, "population sharding needs a communicator: call bbmpc_comm_init first");
                            HIP_CHECK(hipMemcpyAsync(ps_all.p, ps_part.p, pw * 4, hipMemcpyDeviceToDevice, stream));
                        }
                    }
                    HIP_CHECK(hipGetLastError());
                    const size_t ml = ((size_t)2 * G * k + k) * 4;
                    want_lds((const void*)k_cem_merge, ml);
                    hipLaunchKernelGGL(k_cem_merge, dim3(psg, A), dim3(256), ml, stream, rfs, ps_all.p, G);
                    HIP_CHECK(hipGetLastError());
                    capture_trace(it);
                    continue;
                }
                bool first_from_constants = false;
                if (it == 0 && cfg.dynamics == BBMPC_DYN_MLP && !user_path()) {
                    if (cem_skip) { ra.mean = d_prev_mean.p; ra.sigma = d_sigma0.p; ra.state = cem_pinned; }
                    mlp_state_copy = d_state.p;                        // (without the skip: only asks whether the kernel would take it)
                    launch_rollout(SRC_TRUNC, false, ra);
                    const bool took = mlp_state_copy == nullptr;
                    mlp_state_copy = nullptr;
                    pi2_copy_seen = took;
                    if (cem_skip) {
                        if (!took)                                       // a shape the quad kernel refused after all: nobody stored the state
                            HIP_CHECK(hipMemcpyAsync(d_state.p, cem_pinned, (size_t)A * S * 4, hipMemcpyDefault, stream));
                        ra.mean = d_mean.p; ra.sigma = d_sigma.p; ra.state = d_state_in;
                        first_from_constants = true;
                    }
                } else {
                    launch_rollout(SRC_TRUNC, false, ra);
                }
                if (k <= 64 && !sw.refit_v1) {
                    const int rthreads = N > 512 ? 1024 : (N > 256 ? 512 : 256);
                    RefitArgs rf2 = rf;
                    if (!trace_on) rf2.elites = nullptr;          // the sorted elite list is only needed by the parity trace
                    if (first_from_constants) { rf2.mean_in = d_prev_mean.p; rf2.var_in = d_var0.p; }
                    // G workgroups per agent share the elite gather (kernels_refit.hpp); at least 16 rows each
                    const int rg = sw.refit_wgs > 0 ? sw.refit_wgs : std::max(1, std::min(8, HU / 16));
                    hipLaunchKernelGGL(k_refit_cem_v2, dim3(rg, A), dim3(rthreads), (size_t)fixed * 4, stream, rf2);
                } else {
                    hipLaunchKernelGGL(k_refit_cem, dim3(A), dim3(REFIT_THREADS), lds, stream, rf, JC);
                }
                HIP_CHECK(hipGetLastError());
                capture_trace(it);
            }
            if (fix(BBMPC_FIX_Q2_CEM_WARM_START)) {
                if (cfg.dynamics == BBMPC_DYN_MLP || user_path()) pending_warm = 2;      // prev = mean, in k_tail_mlp
                else HIP_CHECK(hipMemcpyAsync(d_prev_mean.p, d_mean.p, (size_t)nelem * 4, hipMemcpyDeviceToDevice, stream));
            }
            break;
        }
        case BBMPC_OPT_PI2: {
            // Learned model on the quad kernel (k_rollout_mlp_q4s): from the second control step on k_dist_init (4.3 us, a
            // launch of its own in front of a 360 us control step) has nothing left to do -- PI2 never changes sigma, the
            // first rollout samples around prev_mean directly, the refit writes every element of the mean, and the first
            // rollout reads the state from the pinned buffer itself (its workgroup 0 stores it for the later launches).
            const bool skip_init = sw.pi2_skip_init && cfg.dynamics == BBMPC_DYN_MLP && !user_path() && !pop_sharded() && !trace_on &&
                                   iters >= 1 && pi2_dist_ready && pi2_copy_seen && stage_state_src != nullptr;
            const float* pinned_state = nullptr;
            last_step_steady = skip_init;
            if (skip_init) {
                pinned_state = stage_state_src;
            } else {
                hipLaunchKernelGGL(k_dist_init, dim3((nelem + 255) / 256), dim3(256), 0, stream, A, HU, U, d_lo.p, d_hi.p,
                                   d_prev_mean.p, d_var0.p, d_mean.p, d_var.p, d_sigma.p, 0, stage_state_src, d_state.p, A * S);
                pi2_dist_ready = true;
            }
            stage_state_src = nullptr;
            if (iters == 0)              // otherwise the last refit writes the action
                HIP_CHECK(hipMemcpy2DAsync(d_action.p, U * 4, d_prev_mean.p, HU * 4, U * 4, A, hipMemcpyDeviceToDevice, stream));
            const float* inj_t = injected(BBMPC_NOISE_TRUNC_NORMAL);
            const size_t lds = (size_t)(Nst + 64) * 4;
            want_lds((const void*)k_refit_pi2_mw, lds);
            want_lds((const void*)k_refit_pi2, lds);
            want_lds((const void*)k_refit_pi2_partial, lds);
            for (int it = 0; it < iters; ++it) {
                ra.stream = BBMPC_NOISE_TRUNC_NORMAL; ra.iter = (uint32_t)it;
                ra.inj = inj_t ? inj_t + inj_stride * it : nullptr;
                if (pop_sharded()) {
                    // population sharded over ranks: partial sums here, one exchange, merge in rank order
                    const int G = ps_loopback > 1 ? ps_loopback : std::max(1, rc.comm ? rc.nranks : 1);
                    const size_t pw = (size_t)A * (HU + 2);
                    if (!ps_part.p) { ps_part.alloc(pw); }
                    if (ps_all.n < pw * G) ps_all.alloc(pw * G);
                    dim3 pgrid((HU + PI2_ROWS - 1) / PI2_ROWS, A), pblock(64 * PI2_ROWS);
                    if (ps_loopback > 1) {
                        // one handle plays every shard in turn (test / measurement hook): shard r = particles [r*N, (r+1)*N)
                        for (int r = 0; r < G; ++r) {
                            ra.pop_offset = r * N;
                            launch_rollout(SRC_TRUNC, true, ra);
                            hipLaunchKernelGGL(k_refit_pi2_partial, pgrid, pblock, lds, stream, rf, ps_all.p + pw * r);
                        }
                        ra.pop_offset = cfg.population_offset;
                    } else {
                        launch_rollout(SRC_TRUNC, true, ra);
                        hipLaunchKernelGGL(k_refit_pi2_partial, pgrid, pblock, lds, stream, rf, ps_part.p);
                        if (rc.comm) {
                            // the exchange sits ON the launch stream: the merge needs it, nothing can overlap it
                            const Rccl& r = *rc.api;
                            r.check(r.AllGather(ps_part.p, ps_all.p, pw, Rccl::kFloat32, rc.comm, stream), "ncclAllGather (PI2 partials)");
                        } else {
                            REQUIRE(cfg.population_global <= N, BBMPC_E_STATE, "population sharding needs a communicator: call bbmpc_comm_init first");
                            HIP_CHECK(hipMemcpyAsync(ps_all.p, ps_part.p, pw * 4, hipMemcpyDeviceToDevice, stream));
                        }
                    }
                    HIP_CHECK(hipGetLastError());
                    hipLaunchKernelGGL(k_refit_pi2_merge, dim3((HU + 255) / 256, A), dim3(256), 0, stream, rf, ps_all.p, G);
                    HIP_CHECK(hipGetLastError());
                    capture_trace(it);
                    continue;
                }
                if (it == 0 && cfg.dynamics == BBMPC_DYN_MLP && !user_path()) {
                    if (skip_init) { ra.mean = d_prev_mean.p; ra.state = pinned_state; mlp_state_copy = d_state.p; }
                    else mlp_state_copy = d_state.p;                   // (only asks: would this launch take the request?)
                    launch_rollout(SRC_TRUNC, true, ra);
                    const bool took = mlp_state_copy == nullptr;
                    mlp_state_copy = nullptr;
                    pi2_copy_seen = took;
                    if (skip_init) {
                        if (!took)                                       // a shape the quad kernel refused after all: nobody stored the state
                            HIP_CHECK(hipMemcpyAsync(d_state.p, pinned_state, (size_t)A * S * 4, hipMemcpyDefault, stream));
                        ra.mean = d_mean.p; ra.state = d_state_in;
                    }
                } else {
                    launch_rollout(SRC_TRUNC, true, ra);
                }
                if (sw.refit_v1) hipLaunchKernelGGL(k_refit_pi2, dim3(A), dim3(REFIT_THREADS), lds, stream, rf);
                else hipLaunchKernelGGL(k_refit_pi2_mw, dim3((HU + PI2_ROWS - 1) / PI2_ROWS, A), dim3(64 * PI2_ROWS), lds, stream, rf);
                HIP_CHECK(hipGetLastError());
                capture_trace(it);
            }
            if (cfg.dynamics == BBMPC_DYN_MLP || user_path()) {
                pending_warm = 1;                      // prev = shift_left(mean) (pi2.py:92-93) happens in k_tail_mlp
            } else {
                hipLaunchKernelGGL(k_shift_left, dim3((nelem + 255) / 256), dim3(256), 0, stream, A, H, U, d_mean.p, d_prev_mean.p);
                HIP_CHECK(hipGetLastError());
            }
            break;
        }
        case BBMPC_OPT_SPSA:
            optimize_spsa(ra, step);
            break;
        case BBMPC_OPT_PSO:
            optimize_pso(ra, step);
            break;
        case BBMPC_OPT_CMAES:
            optimize_cma(ra, step);
            break;
        default:
            throw HipError(BBMPC_E_UNSUPPORTED, "optimizer not built yet");
    }
    finalize(d_state_in, add_noise, d_record_out, d_next_out, step);
}


// SPSAOptimizer._optimize  spsa.py:61-117
void Engine::optimize_spsa(RolloutArgs& ra, uint32_t step) {
    const int nelem = A * HU;
    // solution starts from _current_parameters (d_prev_mean); keep it in d_mean while iterating
    HIP_CHECK(hipMemcpyAsync(d_mean.p, d_prev_mean.p, (size_t)nelem * 4, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpy2DAsync(d_action.p, U * 4, d_prev_mean.p, HU * 4, U * 4, A, hipMemcpyDeviceToDevice, stream));
    const float* inj_r = injected(BBMPC_NOISE_RADEMACHER);
    const size_t inj_stride = (size_t)A * HU * Nst;
    const float big_a = (float)iters / 10.0f;                                   // spsa.py:56
    for (int it = 0; it < iters; ++it) {
        const float tf = (float)it;
        const float ak = cfg.spsa_a / (float)pow((double)((tf + 1.0f) + big_a), (double)cfg.spsa_alpha);   // :69
        const float ck = cfg.spsa_c / (float)pow((double)(tf + 1.0f), (double)cfg.spsa_gamma);             // :70
        OptArgs oa = opt_args(step, (uint32_t)it);
        want_lds((const void*)k_refit_spsa, (size_t)Nst * 4);
        // candidates -> the two rollouts -> row sums of this handle's perturbation pairs (part != null: sharded population)
        auto shard_pass = [&](float* part) {
            hipLaunchKernelGGL(k_spsa_candidates, dim3((N + 255) / 256, HU, A), dim3(256), 0, stream, oa, d_mean.p, ck,
                               inj_r ? inj_r + inj_stride * it : nullptr, d_samples.p, d_cand_a.p, d_cand_b.p);
            HIP_CHECK(hipGetLastError());
            ra.samples = nullptr;              // candidates are clipped in place; delta lives in d_samples
            ra.penalty_out = nullptr;
            ra.cand = d_cand_a.p; ra.samples = d_cand_a.p; ra.rewards = d_rewards.p;
            launch_rollout(SRC_BUF, true, ra);
            ra.cand = d_cand_b.p; ra.samples = d_cand_b.p; ra.rewards = d_rewards2.p;
            launch_rollout(SRC_BUF, true, ra);
            const int rgr = std::max(1, std::min(16, HU / (REFIT_THREADS / 64)));        // workgroups per agent: >= one row per wave
            hipLaunchKernelGGL(k_refit_spsa, dim3(rgr, A), dim3(REFIT_THREADS), (size_t)Nst * 4, stream, oa, d_rewards.p, d_rewards2.p,
                               d_samples.p, ak, ck, d_mean.p, d_action.p, part);
            HIP_CHECK(hipGetLastError());
        };
        if (pop_sharded()) {
            // population sharded over ranks: row sums here, one exchange, the step in rank order (kernels_opt.hpp)
            const int G = ps_loopback > 1 ? ps_loopback : std::max(1, rc.comm ? rc.nranks : 1);
            const size_t pw = (size_t)A * HU;
            if (!ps_part.p || ps_part.n < pw) ps_part.alloc(pw);
            if (ps_all.n < pw * G) ps_all.alloc(pw * G);
            if (ps_loopback > 1) {
                // one handle plays every shard in turn (test / measurement hook): shard r = particles [r*N, (r+1)*N)
                for (int r = 0; r < G; ++r) {
                    oa.pop_offset = r * N;
                    shard_pass(ps_all.p + pw * r);
                }
                oa.pop_offset = cfg.population_offset;
            } else {
                shard_pass(ps_part.p);
                if (rc.comm) {
                    const Rccl& r = *rc.api;
                    r.check(r.AllGather(ps_part.p, ps_all.p, pw, Rccl::kFloat32, rc.comm, stream), "ncclAllGather (SPSA row sums)");
                } else {
                    REQUIRE(cfg.population_global <= N, BBMPC_E_STATE, "population sharding needs a communicator: call bbmpc_comm_init first");
                    HIP_CHECK(hipMemcpyAsync(ps_all.p, ps_part.p, pw * 4, hipMemcpyDeviceToDevice, stream));
                }
            }
            const int n_global = ps_loopback > 1 ? G * N : std::max(N, (int)cfg.population_global);
            hipLaunchKernelGGL(k_spsa_merge, dim3((HU + 255) / 256, A), dim3(256), 0, stream, oa, ps_all.p, G, n_global, ak, d_mean.p, d_action.p);
            HIP_CHECK(hipGetLastError());
        } else {
            shard_pass(nullptr);
        }
        if (trace_on) {
            capture_trace(it);
            if (!t_rewards2.p) t_rewards2.alloc((size_t)A * Nst * std::max(iters, 1));
            HIP_CHECK(hipMemcpyAsync(t_rewards2.p + (size_t)A * Nst * it, d_rewards2.p, (size_t)A * Nst * 4,
                                     hipMemcpyDeviceToDevice, stream));
        }
    }
    if (cfg.dynamics == BBMPC_DYN_MLP || user_path()) {
        pending_warm = 1;                              // :114-115, in k_tail_mlp
    } else {
        hipLaunchKernelGGL(k_shift_left, dim3((nelem + 255) / 256), dim3(256), 0, stream, A, H, U, d_mean.p, d_prev_mean.p);  // :114-115
        HIP_CHECK(hipGetLastError());
    }
}


// PSOOptimizer._optimize  pso.py:70-141
void Engine::optimize_pso(RolloutArgs& ra, uint32_t step) {
    PsoState ps = pso_state();
    const float* inj_s = injected(BBMPC_NOISE_PSO_SCALARS);
    // population sharded over ranks: every rank moves ITS particles; the one cross-particle operation, the
    // argmax of the personal bests (pso.py:94), becomes local best -> all-gather -> first maximum by global index
    const bool sharded = pop_sharded();
    const int G = !sharded ? 1 : (ps_loopback > 1 ? ps_loopback : std::max(1, rc.comm ? rc.nranks : 1));
    const int shards_here = ps_loopback > 1 ? ps_loopback : 1;         // the loopback hook plays every shard in turn
    const size_t pw = (size_t)A * (HU + 2);
    if (sharded) {
        if (!ps_part.p || ps_part.n < pw) ps_part.alloc(pw);
        if (ps_all.n < pw * G) ps_all.alloc(pw * G);
    }
    for (int it = 0; it < iters; ++it) {
        OptArgs oa = opt_args(step, (uint32_t)it);
        for (int r = 0; r < shards_here; ++r) {
            const PsoState pr = pso_state(r);
            if (ps_loopback > 1) oa.pop_offset = r * N;
            ra.cand = pr.pos; ra.samples = pr.pos; ra.rewards = d_rewards.p; ra.penalty_out = nullptr;
            launch_rollout(SRC_BUF, true, ra);                 // clip + penalty + write the feasible positions back
            hipLaunchKernelGGL(k_pso_best, dim3(A), dim3(REFIT_THREADS), 0, stream, oa, pr, d_rewards.p,
                               !sharded ? nullptr : (ps_loopback > 1 ? ps_all.p + pw * r : ps_part.p));
            HIP_CHECK(hipGetLastError());
        }
        if (sharded) {
            if (ps_loopback <= 1) {
                if (rc.comm) {
                    const Rccl& r = *rc.api;
                    r.check(r.AllGather(ps_part.p, ps_all.p, pw, Rccl::kFloat32, rc.comm, stream), "ncclAllGather (PSO local bests)");
                } else {
                    REQUIRE(cfg.population_global <= N, BBMPC_E_STATE, "population sharding needs a communicator: call bbmpc_comm_init first");
                    HIP_CHECK(hipMemcpyAsync(ps_all.p, ps_part.p, pw * 4, hipMemcpyDeviceToDevice, stream));
                }
            }
            hipLaunchKernelGGL(k_pso_merge, dim3(A), dim3(256), 0, stream, oa, ps, ps_all.p, G);
            HIP_CHECK(hipGetLastError());
        }
        for (int r = 0; r < shards_here; ++r) {
            hipLaunchKernelGGL(k_pso_move, dim3((N + 255) / 256, HU, A), dim3(256), 0, stream, oa, pso_state(r), cfg.pso_w, cfg.pso_c1,
                               cfg.pso_c2, inj_s ? inj_s + 2 * it : nullptr);
            HIP_CHECK(hipGetLastError());
        }
        if (trace_on) {
            ensure_trace();
            const size_t nr = (size_t)A * Nst, nm = (size_t)A * HU;
            HIP_CHECK(hipMemcpyAsync(t_rewards.p + nr * it, d_rewards.p, nr * 4, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(t_mean.p + nm * it, ps.gbest, nm * 4, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(t_elites.p + (size_t)A * std::max(k, 1) * it, ps.gidx, (size_t)A * 4, hipMemcpyDeviceToDevice, stream));
        }
    }
    hipLaunchKernelGGL(k_take_first, dim3((A * U + 63) / 64), dim3(64), 0, stream, A, HU, U, ps.gbest, d_action.p);      // :114
    OptArgs oa = opt_args(step, 0u);
    for (int r = 0; r < shards_here; ++r) {
        if (ps_loopback > 1) oa.pop_offset = r * N;
        hipLaunchKernelGGL(k_pso_seed, dim3((N + 255) / 256, HU, A), dim3(256), 0, stream, oa, pso_state(r), d_var0.p, cfg.pso_v0_fraction, 0,
                           injected(BBMPC_NOISE_PSO_RESEED_TRUNC), injected(BBMPC_NOISE_PSO_RESEED_UNIFORM));          // :116-138
        HIP_CHECK(hipGetLastError());
    }
}

void Engine::evaluate_dev(const float* d_state_in, const float* d_seq, int n_pop, float* d_rew_out) {
    // rewards come back in the reference layout [n_pop, A]; the kernel writes [A][stride] so use a scratch
    // and a strided 2D copy (A is small).
    REQUIRE(n_pop >= 1, BBMPC_E_INVALID, "n_pop must be >= 1");
    const int st = ((n_pop + 63) / 64) * 64;
    if (d_eval_rew.n < (size_t)A * st) d_eval_rew.alloc((size_t)A * st);
    RolloutArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.n_pop = n_pop; ra.A = A; ra.H = H; ra.U = U; ra.S = S; ra.HU = HU; ra.Nst = st;
    ra.agent_offset = cfg.agent_offset;
    ra.fix_q1 = fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
    ra.reward_kind = builtin_reward_kind();
    ra.state = d_state_in;
    ra.seq = d_seq;
    ra.lo = d_lo.p; ra.hi = d_hi.p;
    ra.rewards = d_eval_rew.p;
    launch_rollout(SRC_REF, false, ra);
    // [A][st] -> [n_pop][A]: per agent a strided copy (dst pitch A floats, width 1 float)
    for (int a = 0; a < A; ++a)
        HIP_CHECK(hipMemcpy2DAsync(d_rew_out + a, (size_t)A * 4, d_eval_rew.p + (size_t)a * st, 4, 4, n_pop,
                                   hipMemcpyDeviceToDevice, stream));
}

void Engine::step_dev(const float* d_states, const float* d_actions, int astride, int batch, float* d_next, float* d_rew) {
    REQUIRE(batch >= 1, BBMPC_E_INVALID, "batch must be >= 1");
    if (user_path() && (cfg.dynamics == BBMPC_DYN_USER || d_rew)) {
        // user dynamics, or a reward the built-in step kernels cannot evaluate: dynamics rows, then reward rows
        float* nx = d_next;
        if (!nx) {
            if (u_next.n < (size_t)batch * S) u_next.alloc((size_t)batch * S);
            nx = u_next.p;
        }
        if (cfg.dynamics == BBMPC_DYN_USER) dynamics_rows(d_states, d_actions, astride, batch, nx);
        else step_dev(d_states, d_actions, astride, batch, nx, nullptr);
        if (d_rew) reward_rows(d_states, nx, d_actions, astride, batch, d_rew, 0);
        return;
    }
    if (cfg.dynamics == BBMPC_DYN_MLP && batch <= 32 && !sw.mlp_generic) {
        // a handful of rows: one workgroup per row on plain FMAs -- the same per-row code as the control step's tail,
        // so act()'s predicted next state and evaluator.predict_next_state(obs, action) agree bit for bit
        REQUIRE(mlp_ready, BBMPC_E_STATE, "learned dynamics: call bbmpc_set_mlp before computing");
        hipLaunchKernelGGL(k_rows_mlp, dim3(batch), dim3(TAIL_THREADS), 0, stream, row_mlp(), S, U, builtin_reward_kind(),
                           (int)fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER), d_states, d_actions, astride, d_next, d_rew);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (cfg.dynamics == BBMPC_DYN_MLP) {
        // one-step rollout of `batch` independent rows: per-particle start states, H = 1, actions as a
        // [batch,1,1,U] sequence (gathered to a contiguous block first when they come strided)
        const float* acts_c = d_actions;
        if (astride != U) {
            if (d_step_act.n < (size_t)batch * U) d_step_act.alloc((size_t)batch * U);
            HIP_CHECK(hipMemcpy2DAsync(d_step_act.p, (size_t)U * 4, d_actions, (size_t)astride * 4, (size_t)U * 4, batch,
                                       hipMemcpyDeviceToDevice, stream));
            acts_c = d_step_act.p;
        }
        const int st_ = ((batch + 63) / 64) * 64;
        float* rew = d_rew;
        if (!rew) {
            if (d_step_c.n < (size_t)st_) d_step_c.alloc((size_t)st_);
            rew = d_step_c.p;
        }
        RolloutArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.n_pop = batch; ra.A = 1; ra.H = 1; ra.U = U; ra.S = S; ra.HU = U; ra.Nst = st_;
        ra.fix_q1 = fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER);
        ra.reward_kind = builtin_reward_kind();
        ra.state = d_states;
        ra.seq = acts_c;
        ra.lo = d_lo.p; ra.hi = d_hi.p;
        ra.rewards = rew;
        const bool prof = profiling;
        profiling = false;
        launch_rollout_mlp(SRC_REF, false, ra, true, d_next);
        profiling = prof;
        return;
    }
    hipLaunchKernelGGL(k_step_pendulum, dim3((batch + 63) / 64), dim3(64), 0, stream, d_states, d_actions, astride, batch,
                       (int)fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER), d_next, d_rew);
    HIP_CHECK(hipGetLastError());
}

void Engine::reward_dev(const float* d_cur, const float* d_next, const float* d_act, int batch, float* d_rew) {
    if (cfg.reward == BBMPC_REW_USER) {
        reward_rows(d_cur, d_next, d_act, U, batch, d_rew, 0);
        return;
    }
    hipLaunchKernelGGL(k_reward_only, dim3((batch + 63) / 64), dim3(64), 0, stream, d_cur, d_next, d_act, batch, S, U,
                       cfg.reward, (int)fix(BBMPC_FIX_Q1_REWARD_ARG_ORDER), d_rew);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// parity hooks
// ------------------------------------------------------------------------------------------------
void Engine::inject(int kind, const float* data, int64_t count) {
    REQUIRE(auto_split <= 1 || !data, BBMPC_E_UNSUPPORTED, "injected noise is per shard: not available for a population > 32768 (played as shards)");
    if (!data) {
        inj.erase(kind);
        return;
    }
    switch (kind) {
        case BBMPC_NOISE_PSO_SCALARS: {
            REQUIRE(count == (int64_t)2 * std::max(iters, 1), BBMPC_E_INVALID, "PSO scalar noise must be [iters][2]");
            auto& b = inj[kind];
            b.alloc((size_t)count);
            HIP_CHECK(hipMemcpy(b.p, data, (size_t)count * 4, hipMemcpyHostToDevice));
            break;
        }
        case BBMPC_NOISE_TRUNC_NORMAL:
        case BBMPC_NOISE_UNIFORM:
        case BBMPC_NOISE_RADEMACHER:
        case BBMPC_NOISE_PSO_RESEED_TRUNC:
        case BBMPC_NOISE_PSO_RESEED_UNIFORM:
        case BBMPC_NOISE_PSO_RESET_POS:
        case BBMPC_NOISE_NORMAL:
        case BBMPC_NOISE_PSO_RESET_VEL: {
            const int nit = (kind == BBMPC_NOISE_TRUNC_NORMAL || kind == BBMPC_NOISE_RADEMACHER || kind == BBMPC_NOISE_NORMAL)
                                ? std::max(iters, 1) : 1;
            const int64_t per = (int64_t)N * A * HU;
            REQUIRE(count == per * nit, BBMPC_E_INVALID, "injected noise has the wrong element count");
            std::vector<float> tmp((size_t)A * HU * Nst * nit, 0.0f);
            for (int it = 0; it < nit; ++it) to_internal(data + per * it, N, tmp.data() + (size_t)A * HU * Nst * it);
            auto& b = inj[kind];
            b.alloc(tmp.size());
            HIP_CHECK(hipMemcpy(b.p, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
            break;
        }
        case BBMPC_NOISE_EXPLORATION: {
            REQUIRE(count == (int64_t)A * U, BBMPC_E_INVALID, "exploration noise must be [A,U]");
            auto& b = inj[kind];
            b.alloc((size_t)count);
            HIP_CHECK(hipMemcpy(b.p, data, (size_t)count * 4, hipMemcpyHostToDevice));
            break;
        }
        default:
            throw HipError(BBMPC_E_UNSUPPORTED, "noise kind not supported yet");
    }
}

__global__ void k_dump_noise(RngKey key, uint32_t stream, uint32_t iter, int N, int A, int HU, int agent_offset,
                             float* out /* reference layout [N,A,HU] */) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * A * HU) return;
    const int j = idx % HU, a = (idx / HU) % A, n = idx / (HU * A);
    const U4 b = rng_block(key, stream, iter, (uint32_t)n, (uint32_t)(agent_offset + a), (uint32_t)j);
    const uint32_t w = pick_word(b, (uint32_t)j);
    float v;
    if (stream == BBMPC_NOISE_UNIFORM || stream == BBMPC_NOISE_PSO_RESEED_UNIFORM || stream == BBMPC_NOISE_PSO_RESET_POS ||
        stream == BBMPC_NOISE_PSO_RESET_VEL)
        v = word_to_uniform(w);
    else if (stream == BBMPC_NOISE_RADEMACHER)
        v = word_to_rademacher(w);
    else if (stream == BBMPC_NOISE_NORMAL)
        v = elem_normal(key, iter, n, agent_offset + a, j);
    else
        v = word_to_trunc_normal(w);
    out[idx] = v;
}

// the two scalar N(0,1) draws of PSO iteration `iter` (quirk Q3), same counter as pso_scalars (kernels_opt.hpp)
__global__ void k_dump_pso_scalars(OptArgs p, float* out) {
    float r1, r2;
    pso_scalars(p, nullptr, r1, r2);
    out[0] = r1;
    out[1] = r2;
}

void Engine::dump_noise(int kind, int control_step, int iteration, float* out, int64_t count) {
    int n = auto_split > 1 ? N * auto_split : N, a = A, hu = HU;       // (draws are keyed by the global particle)
    RngKey kk = key((uint32_t)control_step);
    if (kind == BBMPC_NOISE_PSO_SCALARS) {
        REQUIRE(count == 2, BBMPC_E_INVALID, "dump_noise: PSO scalars are [2] per (control step, iteration)");
        DevBuf<float> tmp;
        tmp.alloc(2);
        hipLaunchKernelGGL(k_dump_pso_scalars, dim3(1), dim3(1), 0, stream, opt_args((uint32_t)control_step, (uint32_t)iteration), tmp.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, tmp.p, 8, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return;
    }
    if (kind == BBMPC_NOISE_EXPLORATION) {
        n = 1; hu = U;
        kk.q_per_agent = (uint32_t)((U + 3) / 4);
    }
    const int64_t total = (int64_t)n * a * hu;
    REQUIRE(count == total, BBMPC_E_INVALID, "dump_noise: wrong element count");
    DevBuf<float> tmp;
    tmp.alloc((size_t)total);
    hipLaunchKernelGGL(k_dump_noise, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, kk, (uint32_t)kind,
                       (uint32_t)iteration, n, a, hu, cfg.agent_offset, tmp.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out, tmp.p, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

void Engine::get_trace(int it, int item, void* out, int64_t bytes) {
    REQUIRE(trace_on && t_rewards.p, BBMPC_E_STATE, "trace capture is not enabled or no optimize() call yet");
    REQUIRE(it >= 0 && it < std::max(iters, 1), BBMPC_E_INVALID, "trace iteration out of range");
    HIP_CHECK(hipStreamSynchronize(stream));
    const size_t nr = (size_t)A * Nst, nm = (size_t)A * HU, ns = (size_t)A * HU * Nst, ne = (size_t)A * std::max(k, 1);
    switch (item) {
        case BBMPC_TRACE_REWARDS: {
            const bool spsa = cfg.optimizer == BBMPC_OPT_SPSA;
            REQUIRE(bytes == (int64_t)N * A * 4 * (spsa ? 2 : 1), BBMPC_E_INVALID, "trace rewards: wrong size");
            std::vector<float> tmp(nr);
            float* o = (float*)out;
            for (int half = 0; half < (spsa ? 2 : 1); ++half) {
                HIP_CHECK(hipMemcpy(tmp.data(), (half ? t_rewards2.p : t_rewards.p) + nr * it, nr * 4, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; ++n)
                    for (int a = 0; a < A; ++a) o[((size_t)half * N + n) * A + a] = tmp[(size_t)a * Nst + n];
            }
            break;
        }
        case BBMPC_TRACE_MEAN:
        case BBMPC_TRACE_VAR: {
            REQUIRE(bytes == (int64_t)nm * 4, BBMPC_E_INVALID, "trace mean/var: wrong size");
            HIP_CHECK(hipMemcpy(out, (item == BBMPC_TRACE_MEAN ? t_mean.p : t_var.p) + nm * it, nm * 4, hipMemcpyDeviceToHost));
            break;
        }
        case BBMPC_TRACE_ELITES: {
            const size_t cnt = (cfg.optimizer == BBMPC_OPT_CEM) ? (size_t)A * k
                               : (cfg.optimizer == BBMPC_OPT_CMAES ? (size_t)cma_G * k : (size_t)A);
            REQUIRE(bytes == (int64_t)cnt * 4, BBMPC_E_INVALID, "trace elites: wrong size");
            HIP_CHECK(hipMemcpy(out, t_elites.p + ne * it, cnt * 4, hipMemcpyDeviceToHost));
            break;
        }
        case BBMPC_TRACE_SAMPLES: {
            REQUIRE(bytes == (int64_t)N * A * HU * 4, BBMPC_E_INVALID, "trace samples: wrong size");
            std::vector<float> tmp(ns);
            HIP_CHECK(hipMemcpy(tmp.data(), t_samples.p + ns * it, ns * 4, hipMemcpyDeviceToHost));
            from_internal(tmp.data(), N, (float*)out);
            break;
        }
        case BBMPC_TRACE_CMA_SVD_STATS: {
            REQUIRE(cfg.optimizer == BBMPC_OPT_CMAES && t_cma_stats.p, BBMPC_E_STATE, "CMA-ES trace items need a traced CMA-ES control step");
            REQUIRE(bytes == (int64_t)cma_G * 16 * 4, BBMPC_E_INVALID, "CMA-ES SVD statistics: [G,16] int32");
            HIP_CHECK(hipMemcpy(out, t_cma_stats.p + (size_t)cma_G * 16 * it, (size_t)cma_G * 16 * 4, hipMemcpyDeviceToHost));
            break;
        }
        case BBMPC_TRACE_CMA_B:
        case BBMPC_TRACE_CMA_C:
        case BBMPC_TRACE_CMA_D: {
            REQUIRE(cfg.optimizer == BBMPC_OPT_CMAES && t_cma_B.p, BBMPC_E_STATE, "CMA-ES trace items need a traced CMA-ES control step");
            const size_t gn = (size_t)cma_G * cma_n, gnn = gn * cma_n;
            const size_t cnt = item == BBMPC_TRACE_CMA_D ? gn : gnn;
            REQUIRE(bytes == (int64_t)cnt * 4, BBMPC_E_INVALID, "CMA-ES trace item: wrong size");
            const float* src = item == BBMPC_TRACE_CMA_B ? t_cma_B.p : (item == BBMPC_TRACE_CMA_C ? t_cma_C.p : t_cma_D.p);
            HIP_CHECK(hipMemcpy(out, src + cnt * it, cnt * 4, hipMemcpyDeviceToHost));
            break;
        }
        default:
            throw HipError(BBMPC_E_INVALID, "unknown trace item");
    }
}

void Engine::get_state(const std::string& name, float* out, int64_t count) {
    HIP_CHECK(hipStreamSynchronize(stream));
    const size_t nm = (size_t)A * HU;
    const float* src = nullptr;
    if (cfg.optimizer == BBMPC_OPT_CMAES) goto cma_names;     // "sigma"/"m" etc. mean the CMA-ES state there
    if (name == "prev_mean") src = d_prev_mean.p;
    else if (name == "mean") src = d_mean.p;
    else if (name == "var") src = d_var.p;
    else if (name == "sigma") src = d_sigma.p;
cma_names:
    if (!src && cfg.optimizer == BBMPC_OPT_CMAES) {
        const size_t gn = (size_t)cma_G * cma_n, gnn = gn * cma_n;
        const float* v = nullptr;
        size_t cnt = gn;
        if (name == "m") v = c_m.p;
        else if (name == "sigma") v = c_sigma.p;
        else if (name == "p_sigma") v = c_ps.p;
        else if (name == "p_C") v = c_pc.p;
        else if (name == "D") v = c_Dd.p;
        else if (name == "C") { v = c_C.p; cnt = gnn; }
        else if (name == "B") { v = c_B.p; cnt = gnn; }
        REQUIRE(v, BBMPC_E_INVALID, "unknown state tensor '" + name + "'");
        REQUIRE(count == (int64_t)cnt, BBMPC_E_INVALID, "CMA-ES state tensor: wrong element count");
        HIP_CHECK(hipMemcpy(out, v, cnt * 4, hipMemcpyDeviceToHost));
        return;
    }
    if (!src && cfg.optimizer == BBMPC_OPT_PSO) {
        if (name == "pos" || name == "vel" || name == "pbest" || name == "pbest_r")
            REQUIRE(auto_split <= 1, BBMPC_E_UNSUPPORTED,      // (the loopback TEST hook keeps its documented meaning: shard 0's particles)
                    "per-particle state of a population_size > 32768 is kept per shard of the split: not available");
        const float* big = nullptr;
        if (name == "pos") big = d_cand_a.p;
        else if (name == "vel") big = d_vel.p;
        else if (name == "pbest") big = d_pbest.p;
        if (big) {            // [N,A,H,U] reference layout
            REQUIRE(count == (int64_t)N * A * HU, BBMPC_E_INVALID, "state tensor has N*A*H*U elements");
            std::vector<float> tmp((size_t)A * HU * Nst);
            HIP_CHECK(hipMemcpy(tmp.data(), big, tmp.size() * 4, hipMemcpyDeviceToHost));
            from_internal(tmp.data(), N, out);
            return;
        }
        if (name == "pbest_r") {
            REQUIRE(count == (int64_t)N * A, BBMPC_E_INVALID, "pbest_r has N*A elements");
            std::vector<float> tmp((size_t)A * Nst);
            HIP_CHECK(hipMemcpy(tmp.data(), d_pbest_r.p, tmp.size() * 4, hipMemcpyDeviceToHost));
            for (int n = 0; n < N; ++n)
                for (int a = 0; a < A; ++a) out[(size_t)n * A + a] = tmp[(size_t)a * Nst + n];
            return;
        }
        if (name == "gbest") src = d_gbest.p;
        if (name == "gbest_r") {
            REQUIRE(count == (int64_t)A, BBMPC_E_INVALID, "gbest_r has A elements");
            HIP_CHECK(hipMemcpy(out, d_gbest_r.p, (size_t)A * 4, hipMemcpyDeviceToHost));
            return;
        }
    }
    REQUIRE(src, BBMPC_E_INVALID, "unknown state tensor '" + name + "'");
    REQUIRE(count == (int64_t)nm, BBMPC_E_INVALID, "state tensor has A*H*U elements (C: G*n*n)");
    HIP_CHECK(hipMemcpy(out, src, nm * 4, hipMemcpyDeviceToHost));
}

void Engine::set_state(const std::string& name, const float* data, int64_t count) {
    HIP_CHECK(hipStreamSynchronize(stream));
    size_t nm = (size_t)A * HU;
    float* dst = nullptr;
    if (name == "prev_mean") dst = d_prev_mean.p;
    else if (name == "C" && cfg.optimizer == BBMPC_OPT_CMAES) { dst = c_C.p; nm = (size_t)cma_G * cma_n * cma_n; }   // (tests: a covariance at another scale)
    else if (name == "var0") { dst = d_var0.p; pi2_dist_ready = false; }
    cem_sigma0_ready = false;                  // (sigma0 follows prev_mean and var0)
    REQUIRE(dst, BBMPC_E_INVALID, "unknown/unsettable state tensor '" + name + "'");
    REQUIRE(count == (int64_t)nm, BBMPC_E_INVALID, "state tensor has A*H*U elements");
    HIP_CHECK(hipMemcpy(dst, data, nm * 4, hipMemcpyHostToDevice));
}

}  // namespace bbmpc

// ================================================================================================
// C ABI
// ================================================================================================
using bbmpc::Engine;
using bbmpc::HipError;
using bbmpc::Rccl;
using bbmpc::RecordComm;
using bbmpc::create_comm_stream;

struct bbmpc_handle_s {
    Engine* e;
};

#define API_BEGIN try {
#define API_END                                   \
    }                                             \
    catch (const HipError& ex) {                  \
        bbmpc::g_last_error = ex.what();          \
        return ex.code;                           \
    }                                             \
    catch (const std::exception& ex) {            \
        bbmpc::g_last_error = ex.what();          \
        return BBMPC_E_INVALID;                   \
    }                                             \
    return BBMPC_OK;

// Every entry point runs with the handle's device current and leaves the caller's current device as it found it: a
// process may hold handles on several GPUs (bbmpc_config.device) next to a PyTorch caller with its own idea of the
// current device; lazy allocations, stream / event creation, hipFuncSetAttribute and launches all bind to "current".
struct DeviceGuard {
    int prev = -1;
    bool restore = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev && dev >= 0) {
            HIP_CHECK(hipSetDevice(dev));
            restore = true;
        }
    }
    ~DeviceGuard() {
        if (restore) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define CHECK_HANDLE_NOSETTLE(h)                                            \
    if (!(h) || !(h)->e) throw HipError(BBMPC_E_INVALID, "null handle");    \
    DeviceGuard _device_guard((h)->e->device);                              \
    bbmpc::stop_foreign_residents((h)->e)
#define CHECK_HANDLE(h)        \
    CHECK_HANDLE_NOSETTLE(h);  \
    (h)->e->settle()
#define CHECK_PTR(p) \
    if (!(p)) throw HipError(BBMPC_E_INVALID, "null pointer argument: " #p)

extern "C" {

int bbmpc_abi_version(void) { return BBMPC_ABI_VERSION; }
const char* bbmpc_last_error(void) { return bbmpc::g_last_error.c_str(); }

int bbmpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int bbmpc_create(const bbmpc_config* cfg, bbmpc_handle* out) {
    API_BEGIN
    CHECK_PTR(cfg);
    CHECK_PTR(out);
    *out = nullptr;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    struct Restore {
        int d;
        ~Restore() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{caller_dev};                       // the constructor selects cfg->device; the caller's device comes back
    Engine* e = new Engine(*cfg);
    *out = new bbmpc_handle_s{e};
    API_END
}

int bbmpc_destroy(bbmpc_handle h) {
    API_BEGIN
    if (h) {
        if (h->e) {
            DeviceGuard g(h->e->device);
            delete h->e;
        }
        delete h;
    }
    API_END
}

int bbmpc_set_stream(bbmpc_handle h, void* s) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    HIP_CHECK(hipStreamSynchronize(h->e->stream));
    h->e->stream = s ? (hipStream_t)s : h->e->own_stream;
    API_END
}

int bbmpc_set_stream_default(bbmpc_handle h) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    HIP_CHECK(hipStreamSynchronize(h->e->stream));
    h->e->stream = nullptr;
    API_END
}

int bbmpc_set_mlp(bbmpc_handle h, int32_t n_layers, const int32_t* dims, const int32_t* acts, const float* const* w,
                  const float* const* b, int32_t is_normalized, const float* const* stats) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    h->e->set_mlp(n_layers, dims, acts, w, b, is_normalized, stats);
    API_END
}

int bbmpc_set_reward_source(bbmpc_handle h, const char* src) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    CHECK_PTR(src);
    h->e->set_user_source(bbmpc::USER_KIND_REWARD, src);
    API_END
}

int bbmpc_set_reward_callback(bbmpc_handle h, bbmpc_rows_callback fn, void* user) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    h->e->set_user_callback(bbmpc::USER_KIND_REWARD, fn, user);
    API_END
}

int bbmpc_set_dynamics_callback(bbmpc_handle h, bbmpc_rows_callback fn, void* user) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    h->e->set_user_callback(bbmpc::USER_KIND_DYNAMICS, fn, user);
    API_END
}

int bbmpc_set_dynamics_source(bbmpc_handle h, const char* src) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    CHECK_PTR(src);
    h->e->set_user_source(bbmpc::USER_KIND_DYNAMICS, src);
    API_END
}

int bbmpc_check_user_source(int32_t kind, const char* src, int32_t dim_s, int32_t dim_u) {
    API_BEGIN
    CHECK_PTR(src);
    if (kind != bbmpc::USER_KIND_REWARD && kind != bbmpc::USER_KIND_DYNAMICS) throw HipError(BBMPC_E_INVALID, "kind must be 1 (reward) or 2 (dynamics)");
    if (dim_s < 1 || dim_u < 1 || dim_s > 256 || dim_u > 256) throw HipError(BBMPC_E_INVALID, "dim_s / dim_u must be in [1, 256]");
    try {
        (void)bbmpc::compile_user_program(src, kind, dim_s, dim_u);
    } catch (const std::exception& ex) {
        throw HipError(BBMPC_E_INVALID, ex.what());
    }
    API_END
}

int bbmpc_check_user_rollout(int32_t dynamics, int32_t reward, const char* dyn_src, const char* rew_src, int32_t dim_s, int32_t dim_u) {
    API_BEGIN
    if (dynamics != BBMPC_DYN_PENDULUM && dynamics != BBMPC_DYN_USER) throw HipError(BBMPC_E_INVALID, "fused user rollouts exist for analytic dynamics (pendulum / user)");
    if (reward < BBMPC_REW_PENDULUM || reward > BBMPC_REW_USER) throw HipError(BBMPC_E_INVALID, "unknown reward kind");
    if ((dynamics == BBMPC_DYN_USER && !dyn_src) || (reward == BBMPC_REW_USER && !rew_src)) throw HipError(BBMPC_E_INVALID, "missing source");
    if (dim_s < 1 || dim_u < 1 || dim_s > 256 || dim_u > 256) throw HipError(BBMPC_E_INVALID, "dim_s / dim_u must be in [1, 256]");
    try {
        (void)bbmpc::compile_user_rollout(reward == BBMPC_REW_USER ? rew_src : "", dynamics == BBMPC_DYN_USER ? dyn_src : "", dynamics, reward, dim_s, dim_u);
    } catch (const std::exception& ex) {
        throw HipError(BBMPC_E_INVALID, ex.what());
    }
    API_END
}

int bbmpc_mlp_forward(bbmpc_handle h, const float* x, int32_t batch, float* out) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(x);
    CHECK_PTR(out);
    Engine& e = *h->e;
    if (batch < 1) throw HipError(BBMPC_E_INVALID, "batch must be >= 1");
    if (e.cfg.dynamics != BBMPC_DYN_MLP || !e.mlp_ready) throw HipError(BBMPC_E_STATE, "bbmpc_mlp_forward: needs a learned-dynamics handle with weights set");
    const size_t nin = (size_t)batch * e.mlp.dims[0], nout = (size_t)batch * e.mlp.dims[e.mlp.n_layers];
    if (e.d_step_a.n < nin + nout) e.d_step_a.alloc(nin + nout);
    HIP_CHECK(hipMemcpyAsync(e.d_step_a.p, x, nin * 4, hipMemcpyHostToDevice, e.stream));
    e.mlp_forward_rows(e.d_step_a.p, batch, e.d_step_a.p + nin);
    HIP_CHECK(hipMemcpyAsync(out, e.d_step_a.p + nin, nout * 4, hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    API_END
}

// which = 0: process_input(states[B,S], actions[B,U]) -> [B,S+U];  1: process_output(states[B,S], raw[B,S]) -> [B,S]
static void process_io(Engine& e, int which, const float* a, const float* b, int batch, const float* const* stats, float* out) {
    if (batch < 1) throw HipError(BBMPC_E_INVALID, "batch must be >= 1");
    const int S = e.S, U = e.U;
    const size_t na = (size_t)batch * S, nb = (size_t)batch * (which == 0 ? U : S), no = (size_t)batch * (which == 0 ? S + U : S);
    const size_t nst = stats ? (size_t)4 * S + 2 * U : 0;
    if (e.d_step_b.n < na + nb + no + nst) e.d_step_b.alloc(na + nb + no + nst);
    float* da = e.d_step_b.p; float* db = da + na; float* dout = db + nb; float* dst = dout + no;
    HIP_CHECK(hipMemcpyAsync(da, a, na * 4, hipMemcpyHostToDevice, e.stream));
    HIP_CHECK(hipMemcpyAsync(db, b, nb * 4, hipMemcpyHostToDevice, e.stream));
    if (stats) {
        std::vector<float> st;
        const int lens[6] = {S, S, U, U, S, S};
        for (int i = 0; i < 6; ++i) {
            if (!stats[i]) throw HipError(BBMPC_E_INVALID, "null statistics vector");
            st.insert(st.end(), stats[i], stats[i] + lens[i]);
        }
        HIP_CHECK(hipMemcpy(dst, st.data(), nst * 4, hipMemcpyHostToDevice));
    }
    if (which == 0) hipLaunchKernelGGL(bbmpc::k_process_input, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, e.stream, da, db, batch, S, U, stats ? dst : nullptr, dout);
    else hipLaunchKernelGGL(bbmpc::k_process_output, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, e.stream, da, db, batch, S, U, stats ? dst : nullptr, dout);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out, dout, no * 4, hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
}

int bbmpc_process_input(bbmpc_handle h, const float* states, const float* actions, int32_t batch, const float* const* stats, float* out) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(states);
    CHECK_PTR(actions);
    CHECK_PTR(out);
    process_io(*h->e, 0, states, actions, batch, stats, out);
    API_END
}

int bbmpc_process_output(bbmpc_handle h, const float* states, const float* raw_output, int32_t batch, const float* const* stats, float* out) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(states);
    CHECK_PTR(raw_output);
    CHECK_PTR(out);
    process_io(*h->e, 1, states, raw_output, batch, stats, out);
    API_END
}

int bbmpc_reset(bbmpc_handle h) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    h->e->reset();
    API_END
}

int bbmpc_optimize_dev(bbmpc_handle h, const float* d_state, int32_t, int32_t noise, float* d_record,
                       float* d_next_state) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    CHECK_PTR(d_state);
    CHECK_PTR(d_record);
    if (d_next_state == d_state) throw HipError(BBMPC_E_INVALID, "d_next_state must not alias d_state");
    h->e->optimize_dev(d_state, noise, d_record, d_next_state);
    API_END
}

}  // extern "C"

// One host-in / host-out control step (the body of bbmpc_optimize, shared with bbmpc_optimize_gather): returns the packed
// records [A][U+S+1] in the handle's pinned buffer.  The caller has NOT settled the handle: consecutive calls are ordered
// by the stream (or served by the resident kernel), everything else settles first.
static const float* optimize_host(bbmpc::Engine& e, const float* state, int32_t noise, float* action, float* next_state, float* reward) {
    using namespace bbmpc;
    const size_t ns = (size_t)e.A * e.S, nr = (size_t)e.A * e.rec;
    float* pin = e.pinned(ns + nr);
    memcpy(pin, state, ns * 4);
    bool published = false;
    bool handled_resident = false;       // served by the resident workgroups: the records are already in the pinned buffer
    // single-kernel control steps read the state straight from the pinned, device-mapped host buffer; those and the
    // learned-dynamics path (whose last kernel, k_tail_mlp, owns the record) write the packed record straight into it
    const bool fused_step = e.sw.zero_copy && (e.use_fused() || e.use_fused_pso());   // PSO: the swarm re-seed that follows touches neither
    // every other optimizer path: the state goes to HBM once (many workgroups read it), the record comes back on its own --
    // whichever kernel packs it writes straight into the pinned buffer; k_tail_mlp, k_finalize_pendulum and the fused
    // CMA-ES kernel also publish the completion word, the rest is waited for on the stream
    const bool mlp_tail = e.sw.zero_copy && !fused_step && e.cfg.optimizer != BBMPC_OPT_NONE;
    if (fused_step || mlp_tail) {
        // the persistent kernel reads the [A,S] state and writes the packed record straight from / to the pinned,
        // device-mapped host buffer (a few PCIe transactions) -- no copy-engine round trips around a ~50 us kernel
        float* dpin = e.h_pin_dev;
        if (e.sw.host_poll && !e.trace_on) {
            // ... and its last workgroup publishes a sequence number into a pinned host word right after the record
            // stores (publish_records_done): the call returns when the host sees it, ~10 us earlier than
            // hipStreamSynchronize notices the kernel's completion; the stream itself is joined lazily (settle)
            if (!e.host_done) {
                HIP_CHECK(hipHostMalloc((void**)&e.host_done, (size_t)e.sync_lines() * 64, hipHostMallocCoherent | hipHostMallocMapped));   // engine.hpp: completion words | request lines | exit words
                memset(e.host_done, 0, (size_t)e.sync_lines() * 64);
                HIP_CHECK(hipHostGetDevicePointer((void**)&e.host_done_dev, e.host_done, 0));
                HIP_CHECK(hipMalloc((void**)&e.host_count, 8));
                HIP_CHECK(hipMemset(e.host_count, 0, 8));
            }
            if (++e.host_seq == 0) e.host_seq = 1;
            e.tail_flag = e.host_done_dev;
            e.tail_count = e.host_count;
            e.tail_value = e.host_seq;
            e.tail_attached = false;
        }
        // one agent on the persistent pendulum kernel: the previous call's kernel may still be there, waiting for this one
        const bool linger_ok = fused_step && e.tail_flag != nullptr && e.sw.linger_us > 0 && e.A <= Engine::kLingerMaxAgents && e.use_fused() &&
                               !e.profiling && e.tail_event == nullptr &&
                               e.stream == e.own_stream;      // on a caller's stream it would hold back the caller's next work
        bool handled = false;
        if (e.resident_alive) {
            if (linger_ok) handled = e.resident_step(pin, noise, e.host_seq);
            else e.resident_stop();
        }
        if (handled) ++e.calls_resident; else ++e.calls_launched;
        if (handled) {
            published = true;
            handled_resident = true;
            e.tail_flag = nullptr;
        } else
        try {
            e.linger_launch = linger_ok && e.subset_n == 0;      // a subset launch (after a crossed exit) is an ordinary one
            if (fused_step) {
                e.optimize_dev(dpin, noise, dpin + ns, nullptr);
            } else {
                // hundreds of rollout workgroups read the state: it goes to HBM once, the record comes back on its own.
                // CEM / PI2 begin with k_dist_init, which fetches it from the pinned buffer itself; the others get a copy
                const bool stage = (e.cfg.optimizer == BBMPC_OPT_CEM || e.cfg.optimizer == BBMPC_OPT_PI2) && !e.pop_sharded() &&
                                   !e.user_path();
                // steady state of the learned-model PI2 / CEM path: the same launches with the same arguments every call --
                // replayed as a graph (engine.hpp: step_graph)
                const bool graph_ok = e.sw.step_graph && stage && e.cfg.dynamics == BBMPC_DYN_MLP && e.tail_flag != nullptr &&
                                      e.tail_event == nullptr && !e.profiling && !e.trace_on && e.stream == e.own_stream && !e.any_injected();
                const uint64_t sig = ((uint64_t)e.mutations << 8) | (uint64_t)(noise ? 1 : 0) | 2u;
                bool replayed = false;
                if (graph_ok && e.step_graph && e.step_graph_sig == sig) {
                    replayed = true;
                } else if (graph_ok && e.step_graph_warm >= 3 && e.step_warm_sig == sig) {
                    if (e.step_graph) { (void)hipGraphExecDestroy(e.step_graph); e.step_graph = nullptr; }      // (captured for other arguments)
                    // capture this call's launches; nothing runs until the graph is launched below
                    if (!e.step_words) {
                        HIP_CHECK(hipHostMalloc((void**)&e.step_words, 64, 0));        // staging for the (rare) re-synchronisation of the device words
                        memset(e.step_words, 0, 64);
                        HIP_CHECK(hipMalloc((void**)&e.step_words_dev, 64));
                        HIP_CHECK(hipMemset(e.step_words_dev, 0, 64));
                        e.step_mirror[0] = e.step_mirror[1] = 0xFFFFFFFFu;
                    }
                    const uint32_t saved_step = e.step_counter;
                    hipGraph_t g = nullptr;
                    bool ok = hipStreamBeginCapture(e.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        e.step_capturing = true;
                        try {
                            e.stage_state_src = dpin;
                            e.optimize_dev(e.d_state.p, noise, dpin + ns, nullptr);
                        } catch (...) { ok = false; }
                        e.step_capturing = false;
                        ok = (hipStreamEndCapture(e.stream, &g) == hipSuccess) && ok && g != nullptr && e.last_step_steady && e.stage_state_src == nullptr;
                        e.stage_state_src = nullptr;
                        if (ok) ok = hipGraphInstantiate(&e.step_graph, g, nullptr, nullptr, 0) == hipSuccess;
                        if (g) (void)hipGraphDestroy(g);
                    }
                    e.step_counter = saved_step;                 // the captured call has not run
                    (void)hipGetLastError();
                    if (ok) { replayed = true; e.step_graph_sig = sig; }
                    else {
                        // this handle keeps enqueueing its launches one by one; why is kept (bbmpc_graph_stats' callers see 0 replays,
                        // BBMPC_TRACE_GRAPH=1 prints it): a capture-illegal call added to the steady path must not go unnoticed
                        e.step_graph = nullptr; e.sw.step_graph = 0;
                        ++e.graph_capture_failures;
                        if (getenv("BBMPC_TRACE_GRAPH")) fprintf(stderr, "[bbmpc] control-step graph capture / instantiate failed (steady=%d): replay disabled for this handle\n", (int)e.last_step_steady);
                    }
                }
                if (replayed) {
                    // the device's (control step, completion value) advance by themselves at the end of every replay; calls that
                    // did not go through the graph in between put them out of step with the host's: one 8-byte copy then
                    if (e.step_mirror[0] != e.step_counter || e.step_mirror[1] != e.host_seq) {
                        HIP_CHECK(hipStreamSynchronize(e.stream));           // (the staging words may still be in use by an earlier copy)
                        e.step_words[0] = e.step_counter; e.step_words[1] = e.host_seq;
                        HIP_CHECK(hipMemcpyAsync(e.step_words_dev, e.step_words, 8, hipMemcpyHostToDevice, e.stream));
                    }
                    HIP_CHECK(hipGraphLaunch(e.step_graph, e.stream));
                    ++e.calls_graph;
                    ++e.step_counter;
                    e.step_mirror[0] = e.step_counter;
                    e.step_mirror[1] = e.host_seq + 1u == 0u ? 1u : e.host_seq + 1u;
                    e.tail_attached = true;
                } else {
                if (stage) e.stage_state_src = dpin;
                else HIP_CHECK(hipMemcpyAsync(e.d_state.p, pin, ns * 4, hipMemcpyHostToDevice, e.stream));
                e.optimize_dev(e.d_state.p, noise, dpin + ns, nullptr);
                if (e.stage_state_src) {
                    e.stage_state_src = nullptr;
                    throw HipError(BBMPC_E_HIP, "internal: the control step did not stage its state");
                }
                if (graph_ok && e.last_step_steady && e.step_warm_sig == sig) ++e.step_graph_warm;
                else { e.step_graph_warm = (graph_ok && e.last_step_steady) ? 1 : 0; e.step_warm_sig = sig; }
                }
            }
            e.linger_launch = false;
            published = e.tail_flag != nullptr && e.tail_attached;
            e.tail_flag = nullptr;
        } catch (...) {
            e.linger_launch = false;
            e.tail_flag = nullptr;
            e.subset_n = 0;                                  // a subset named by resident_step must not outlive the call it was for
            throw;
        }
    } else {
        ++e.calls_launched;
        HIP_CHECK(hipMemcpyAsync(e.d_state.p, pin, ns * 4, hipMemcpyHostToDevice, e.stream));
        e.optimize_dev(e.d_state.p, noise, e.d_record.p, nullptr);
        HIP_CHECK(hipMemcpyAsync(pin + ns, e.d_record.p, nr * 4, hipMemcpyDeviceToHost, e.stream));
    }
    if (e.in_flight_hook && !e.in_flight_called) { e.in_flight_called = true; e.in_flight_hook(&e); }   // launch path: the kernels are enqueued
    if (handled_resident) {
        std::atomic_thread_fence(std::memory_order_acquire);
        e.lazy_sync = true;
    } else if (published && e.resident_alive) {
        // a LINGER launch: every agent's workgroup publishes its own completion word
        const uint32_t want = e.host_seq;
        const auto t0 = std::chrono::steady_clock::now();
        for (int a = 0; a < e.A; ++a) {
            volatile const uint32_t* f = e.ack_host(a);
            uint32_t spins = 0;
            while (*f != want) {
                if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                    e.resident_stop();                               // a fault surfaces in its synchronize
                    if (*f != want) throw HipError(BBMPC_E_HIP, "bbmpc_optimize: the control step finished without publishing its records");
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        e.lazy_sync = true;
    } else if (published) {
        volatile const uint32_t* f = e.host_done;
        const uint32_t want = e.host_seq;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (*f != want) {
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                HIP_CHECK(hipStreamSynchronize(e.stream));       // a fault surfaces here; otherwise the word must be there
                if (*f != want) throw HipError(BBMPC_E_HIP, "bbmpc_optimize: the control step finished without publishing its records");
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        e.lazy_sync = true;
    } else {
        HIP_CHECK(hipStreamSynchronize(e.stream));
        e.lazy_sync = false;
    }
    const float* r = pin + ns;
    for (int a = 0; a < e.A; ++a) {
        if (action) memcpy(action + (size_t)a * e.U, r + (size_t)a * e.rec, e.U * 4);
        if (next_state) memcpy(next_state + (size_t)a * e.S, r + (size_t)a * e.rec + e.U, e.S * 4);
        if (reward) reward[a] = r[(size_t)a * e.rec + e.U + e.S];
    }
    return r;
}

extern "C" {

int bbmpc_optimize(bbmpc_handle h, const float* state, int32_t t, int32_t noise, float* action, float* next_state,
                   float* reward) {
    API_BEGIN
    CHECK_HANDLE_NOSETTLE(h);            // consecutive calls are ordered by the stream; everything else settles first
    CHECK_PTR(state);
    (void)t;  // the reference evaluator accepts and ignores time_step (deterministic.py:26)
    (void)optimize_host(*h->e, state, noise, action, next_state, reward);
    API_END
}

int bbmpc_rollout_episode(bbmpc_handle h, const float* start_state, int32_t num_steps, int32_t noise, float* records_out) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    CHECK_PTR(start_state);
    CHECK_PTR(records_out);
    Engine& e = *h->e;
    if (num_steps < 1) throw HipError(BBMPC_E_INVALID, "num_steps must be >= 1");
    const size_t ns = (size_t)e.A * e.S, nr = (size_t)e.A * e.rec;
    if (e.d_step_d.n < 2 * ns + nr * (size_t)num_steps) e.d_step_d.alloc(2 * ns + nr * (size_t)num_steps);
    float* st[2] = {e.d_step_d.p, e.d_step_d.p + ns};
    float* recs = e.d_step_d.p + 2 * ns;
    HIP_CHECK(hipMemcpyAsync(st[0], start_state, ns * 4, hipMemcpyHostToDevice, e.stream));
    for (int t = 0; t < num_steps; ++t) e.optimize_dev(st[t & 1], noise, recs + nr * t, st[(t + 1) & 1]);
    HIP_CHECK(hipMemcpyAsync(records_out, recs, nr * (size_t)num_steps * 4, hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    API_END
}

int bbmpc_evaluate_dev(bbmpc_handle h, const float* d_state, const float* d_seq, int32_t n_pop, float* d_rewards) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(d_state);
    CHECK_PTR(d_seq);
    CHECK_PTR(d_rewards);
    h->e->evaluate_dev(d_state, d_seq, n_pop, d_rewards);
    API_END
}

int bbmpc_evaluate(bbmpc_handle h, const float* state, const float* seq, int32_t n_pop, float* rewards) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(state);
    CHECK_PTR(seq);
    CHECK_PTR(rewards);
    Engine& e = *h->e;
    if (n_pop < 1) throw HipError(BBMPC_E_INVALID, "n_pop must be >= 1");
    const size_t nseq = (size_t)n_pop * e.A * e.HU, nrew = (size_t)n_pop * e.A;
    if (e.d_eval_seq.n < nseq + nrew) e.d_eval_seq.alloc(nseq + nrew);
    HIP_CHECK(hipMemcpyAsync(e.d_state.p, state, (size_t)e.A * e.S * 4, hipMemcpyHostToDevice, e.stream));
    HIP_CHECK(hipMemcpyAsync(e.d_eval_seq.p, seq, nseq * 4, hipMemcpyHostToDevice, e.stream));
    e.evaluate_dev(e.d_state.p, e.d_eval_seq.p, n_pop, e.d_eval_seq.p + nseq);
    HIP_CHECK(hipMemcpyAsync(rewards, e.d_eval_seq.p + nseq, nrew * 4, hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    API_END
}

int bbmpc_step_dev(bbmpc_handle h, const float* d_states, const float* d_actions, int32_t astride, int32_t batch,
                   float* d_next, float* d_rew) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(d_states);
    CHECK_PTR(d_actions);
    h->e->step_dev(d_states, d_actions, astride, batch, d_next, d_rew);
    API_END
}

int bbmpc_predict_next_state(bbmpc_handle h, const float* states, const float* actions, int32_t batch, float* next) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(states);
    CHECK_PTR(actions);
    CHECK_PTR(next);
    Engine& e = *h->e;
    if (batch < 1) throw HipError(BBMPC_E_INVALID, "batch must be >= 1");
    const size_t ns = (size_t)batch * e.S, na = (size_t)batch * e.U;
    if (e.d_step_a.n < 2 * ns + na) e.d_step_a.alloc(2 * ns + na);
    float* ds = e.d_step_a.p; float* da = ds + ns; float* dn = da + na;
    HIP_CHECK(hipMemcpyAsync(ds, states, ns * 4, hipMemcpyHostToDevice, e.stream));
    HIP_CHECK(hipMemcpyAsync(da, actions, na * 4, hipMemcpyHostToDevice, e.stream));
    e.step_dev(ds, da, e.U, batch, dn, nullptr);
    HIP_CHECK(hipMemcpyAsync(next, dn, ns * 4, hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    API_END
}

int bbmpc_evaluate_next_reward(bbmpc_handle h, const float* states, const float* next_states, const float* actions,
                               int32_t batch, float* rewards) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(states);
    CHECK_PTR(next_states);
    CHECK_PTR(actions);
    CHECK_PTR(rewards);
    Engine& e = *h->e;
    if (batch < 1) throw HipError(BBMPC_E_INVALID, "batch must be >= 1");
    const size_t ns = (size_t)batch * e.S, na = (size_t)batch * e.U;
    if (e.d_step_b.n < 2 * ns + na + batch) e.d_step_b.alloc(2 * ns + na + batch);
    float* dc = e.d_step_b.p; float* dn = dc + ns; float* da = dn + ns; float* dr = da + na;
    HIP_CHECK(hipMemcpyAsync(dc, states, ns * 4, hipMemcpyHostToDevice, e.stream));
    HIP_CHECK(hipMemcpyAsync(dn, next_states, ns * 4, hipMemcpyHostToDevice, e.stream));
    HIP_CHECK(hipMemcpyAsync(da, actions, na * 4, hipMemcpyHostToDevice, e.stream));
    e.reward_dev(dc, dn, da, batch, dr);
    HIP_CHECK(hipMemcpyAsync(rewards, dr, (size_t)batch * 4, hipMemcpyDeviceToHost, e.stream));
    HIP_CHECK(hipStreamSynchronize(e.stream));
    API_END
}

int bbmpc_inject_noise(bbmpc_handle h, int32_t kind, const float* data, int64_t count) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    HIP_CHECK(hipStreamSynchronize(h->e->stream));
    h->e->inject(kind, data, count);
    API_END
}

int bbmpc_dump_noise(bbmpc_handle h, int32_t kind, int32_t control_step, int32_t iteration, float* out, int64_t count) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(out);
    h->e->dump_noise(kind, control_step, iteration, out, count);
    API_END
}

int bbmpc_set_trace(bbmpc_handle h, int32_t enabled) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    if (enabled && h->e->auto_split > 1)
        throw HipError(BBMPC_E_UNSUPPORTED, "the parity trace is per shard: not available for a population > 32768 (played as shards)");
    h->e->trace_on = enabled != 0;
    API_END
}

int bbmpc_get_trace(bbmpc_handle h, int32_t iteration, int32_t item, void* out, int64_t bytes) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(out);
    h->e->get_trace(iteration, item, out, bytes);
    API_END
}

int bbmpc_get_state(bbmpc_handle h, const char* name, float* out, int64_t count) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(name);
    CHECK_PTR(out);
    h->e->get_state(name, out, count);
    API_END
}

int bbmpc_set_state(bbmpc_handle h, const char* name, const float* data, int64_t count) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    CHECK_PTR(name);
    CHECK_PTR(data);
    h->e->set_state(name, data, count);
    API_END
}

int bbmpc_set_profiling(bbmpc_handle h, int32_t enabled) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    h->e->profiling = enabled != 0;
    h->e->prof_every = enabled > 1 ? enabled : 1;
    h->e->prof_seq = 0;
    API_END
}

int bbmpc_get_profile(bbmpc_handle h, double* ms, int64_t* launches, const char** name) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(ms);
    CHECK_PTR(launches);
    h->e->get_profile(ms, launches);
    if (name) *name = h->e->dominant_kernel;
    API_END
}

int bbmpc_profile_instantiation(bbmpc_handle h, const char** name) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(name);
    const Engine& e = *h->e;
    // the recorded instantiation belongs to the dominant kernel only while that kernel is the one being launched
    const size_t nb = strlen(e.dominant_kernel);
    *name = (e.dominant_inst[0] && strncmp(e.dominant_inst, e.dominant_kernel, nb) == 0 && e.dominant_inst[nb] == '<') ? e.dominant_inst
                                                                                                                    : e.dominant_kernel;
    API_END
}

int bbmpc_synchronize(bbmpc_handle h) {
    API_BEGIN
    CHECK_HANDLE(h);
    HIP_CHECK(hipStreamSynchronize(h->e->stream));
    if (h->e->rc.stream) HIP_CHECK(hipStreamSynchronize(h->e->rc.stream));
    API_END
}

int bbmpc_comm_unique_id(void* out, int64_t bytes) {
    API_BEGIN
    CHECK_PTR(out);
    if (bytes < (int64_t)sizeof(Rccl::UniqueId)) throw HipError(BBMPC_E_INVALID, "bbmpc_comm_unique_id: buffer smaller than BBMPC_COMM_ID_BYTES");
    const Rccl& r = Rccl::get();
    Rccl::UniqueId id;
    r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out, &id, sizeof(id));
    API_END
}

// stream, events and hand-off flags of a handle's communicator, whoever serves it (`join` creates c.comm)
static void comm_setup(Engine* e, int32_t nranks, int32_t rank, const Rccl& api, const char* what, const std::function<int(Rccl::Comm*)>& join) {
    if (nranks < 1 || rank < 0 || rank >= nranks) throw HipError(BBMPC_E_INVALID, "bbmpc_comm_init: rank / nranks out of range");
    if (e->rc.comm) throw HipError(BBMPC_E_STATE, "bbmpc_comm_init: the handle already has a communicator");
    RecordComm& c = e->rc;
    c.stream = create_comm_stream();
    for (int s = 0; s < RecordComm::kSlots; ++s) {
        HIP_CHECK(hipEventCreateWithFlags(&c.ready[s], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c.done[s], hipEventDisableTiming));
    }
    const int irc = join(&c.comm);
    if (irc != 0) {
        c.comm = nullptr;
        c.destroy();                     // stream + events created above
        api.check(irc, what);
    }
    c.api = &api;
    c.nranks = nranks;
    c.rank = rank;
    // BBMPC_COMM_SYNC=event: events only; default: flags in signal memory where the device supports stream wait-value
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, e->device);
    const char* sm = getenv("BBMPC_COMM_SYNC");
    c.sync_mode = can_wait ? 1 : 0;
    if (sm && !strcmp(sm, "event")) c.sync_mode = 0;
    if (c.sync_mode != 0) {
        // signal memory comes in 8-byte units, lives in host memory and can be polled by the host
        bool ok = hipExtMallocWithFlags((void**)&c.flag, 8, hipMallocSignalMemory) == hipSuccess;
        for (int s = 0; ok && s < RecordComm::kSlots; ++s)
            ok = hipExtMallocWithFlags((void**)&c.done_flag[s], 8, hipMallocSignalMemory) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            c.sync_mode = 0;
        } else {
            HIP_CHECK(hipMalloc((void**)&c.count, 8));
            HIP_CHECK(hipMemset(c.count, 0, 8));
            HIP_CHECK(hipMemset(c.flag, 0, 8));
            for (int s = 0; s < RecordComm::kSlots; ++s) HIP_CHECK(hipMemset(c.done_flag[s], 0, 8));
            // test hook: start the sequence numbers just below the wrap (tests/test_gpu_comm.py)
            if (const char* s0 = getenv("BBMPC_COMM_SEQ_START")) {
                c.seq = (uint32_t)strtoul(s0, nullptr, 0);
                HIP_CHECK(hipMemcpy(c.flag, &c.seq, 4, hipMemcpyHostToDevice));
                for (int s = 0; s < RecordComm::kSlots; ++s) {
                    HIP_CHECK(hipMemcpy(c.done_flag[s], &c.seq, 4, hipMemcpyHostToDevice));
                    c.done_seq[s] = c.seq;
                }
            }
        }
    }
}

int bbmpc_comm_init(bbmpc_handle h, const void* unique_id, int32_t nranks, int32_t rank) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    CHECK_PTR(unique_id);
    const Rccl& r = Rccl::get();            // (the handle's device is current: CHECK_HANDLE)
    Rccl::UniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    comm_setup(h->e, nranks, rank, r, "ncclCommInitRank", [&](Rccl::Comm* out) { return r.CommInitRank(out, nranks, id, rank); });
    API_END
}

int bbmpc_comm_init_local(bbmpc_handle h, uint64_t group_key, int32_t nranks, int32_t rank) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    Engine* e = h->e;
    comm_setup(e, nranks, rank, Rccl::local(), "bbmpc_comm_init_local",
               [&](Rccl::Comm* out) { return bbmpc::local_comm_join(group_key, nranks, rank, e->device, out); });
    API_END
}

// sequence number of the next hand-off (flag modes)
static uint32_t next_comm_seq(Engine* e) {
    RecordComm& c = e->rc;
    if (c.seq >= 0x7fffffffu) {                                        // wrap: once per 2^31 control steps
        HIP_CHECK(hipStreamSynchronize(e->stream));
        HIP_CHECK(hipStreamSynchronize(c.stream));
        HIP_CHECK(hipMemset(c.flag, 0, 8));
        for (int s = 0; s < RecordComm::kSlots; ++s) {
            HIP_CHECK(hipMemset(c.done_flag[s], 0, 8));
            c.done_seq[s] = 0;
            c.pending[s] = false;
        }
        c.seq = 0;
    }
    return ++c.seq;
}

// Enqueue the all-gather of `count` floats per rank behind everything the launch stream holds so far.
//
// Hand-off launch stream -> communication stream, two forms:
//  * `published`: the control step was one persistent kernel and its last workgroup publishes sequence number `v` in
//    signal memory (kernels_fused.hpp); the communication stream waits for the value.  Nothing is added to the
//    launch stream: +2 us per control step against no gather at all (tools/gather_overhead.py).
//  * otherwise an event (`event_attached`: it already rides on the last kernel's dispatch packet).  A cross-stream
//    event costs the launch stream 5-9 us per control step -- the runtime adds a barrier packet and a completion
//    signal between two back-to-back kernels -- which only matters for ~50 us control steps, and those are the
//    single-kernel ones.  (A wait-value that sits in the queue across many launches is worse: the command processor
//    polls it between the other queue's dispatches; measured +77 us on a 25-launch control step.)
// Completion (sync_mode 1) goes to a second flag that the host polls in bbmpc_gather_wait: an event recorded on the
// communication stream and queried from the host was measured to cost the launch stream another 4 us per step.
// the collective itself + "slot done" on the communication stream (whatever made d_records ready is already ordered before it there)
static void gather_enqueue(Engine* e, const float* d_records, float* d_gathered, size_t count, int slot, uint32_t v) {
    RecordComm& c = e->rc;
    const Rccl& r = *c.api;
    r.check(r.AllGather(d_records, d_gathered, count, Rccl::kFloat32, c.comm, c.stream), "ncclAllGather");
    if (c.sync_mode == 1) {
        HIP_CHECK(hipStreamWriteValue32(c.stream, c.done_flag[slot], v, 0));
        c.done_seq[slot] = v;
    } else {
        HIP_CHECK(hipEventRecord(c.done[slot], c.stream));
    }
    c.pending[slot] = true;
}

// staged records of earlier bbmpc_optimize_gather calls -> HBM -> collective, on the communication stream
static void flush_deferred_gathers(Engine* e) {
    RecordComm& c = e->rc;
    if (!c.comm) return;
    const size_t nr = (size_t)e->A * e->rec;
    for (int s = 0; s < RecordComm::kSlots; ++s) {
        if (!e->gather_deferred[s]) continue;
        e->gather_deferred[s] = false;
        HIP_CHECK(hipMemcpyAsync(e->d_record_slot[s].p, e->h_record_stage[s], nr * sizeof(float), hipMemcpyHostToDevice, c.stream));
        gather_enqueue(e, e->d_record_slot[s].p, e->gather_deferred_dst[s], nr, s, c.sync_mode == 1 ? next_comm_seq(e) : 0u);
    }
}

static void gather_records(Engine* e, const float* d_records, float* d_gathered, size_t count, int slot, bool event_attached,
                           bool published, uint32_t v) {
    RecordComm& c = e->rc;
    if (published) {
        HIP_CHECK(hipStreamWaitValue32(c.stream, c.flag, v, hipStreamWaitValueGte, 0xffffffffu));
    } else {
        if (!event_attached) HIP_CHECK(hipEventRecord(c.ready[slot], e->stream));
        HIP_CHECK(hipStreamWaitEvent(c.stream, c.ready[slot], 0));
    }
    gather_enqueue(e, d_records, d_gathered, count, slot, v);
}

int bbmpc_gather_records_dev(bbmpc_handle h, const float* d_records, float* d_gathered, int64_t count, int32_t slot) {
    API_BEGIN
    CHECK_HANDLE(h);
    // the communicator of a population-sharded handle carries the per-iteration partials on the LAUNCH stream; one RCCL
    // communicator must not be driven from two unsynchronised streams, so such a handle has no record gather (its ranks
    // all hold the same agents: there is nothing to gather).  A population that is only split into shards on THIS GPU
    // (population_size > 32768, or the loopback hook) never drives the communicator and gathers like any other handle
    if (h->e->pop_sharded_across_ranks()) throw HipError(BBMPC_E_UNSUPPORTED, "record gather on a handle whose population is sharded over ranks (every rank already holds every agent's record)");
    CHECK_PTR(d_records);
    CHECK_PTR(d_gathered);
    Engine* e = h->e;
    RecordComm& c = e->rc;
    if (!c.comm) throw HipError(BBMPC_E_STATE, "bbmpc_gather_records_dev: call bbmpc_comm_init first");
    if (slot < 0 || slot >= RecordComm::kSlots) throw HipError(BBMPC_E_INVALID, "bbmpc_gather_records_dev: slot must be 0 or 1");
    if (count <= 0) throw HipError(BBMPC_E_INVALID, "bbmpc_gather_records_dev: count must be positive");
    if (c.pending[slot]) throw HipError(BBMPC_E_STATE, "bbmpc_gather_records_dev: slot still pending (call bbmpc_gather_wait)");
    gather_records(e, d_records, d_gathered, (size_t)count, slot, false, false, c.sync_mode ? next_comm_seq(e) : 0u);
    API_END
}

int bbmpc_optimize_gather_dev(bbmpc_handle h, const float* d_state, int32_t, int32_t noise, float* d_records,
                              float* d_next_state, float* d_gathered, int32_t slot) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    // the communicator of a population-sharded handle carries the per-iteration partials on the LAUNCH stream; one RCCL
    // communicator must not be driven from two unsynchronised streams, so such a handle has no record gather (its ranks
    // all hold the same agents: there is nothing to gather).  A population that is only split into shards on THIS GPU
    // (population_size > 32768, or the loopback hook) never drives the communicator and gathers like any other handle
    if (h->e->pop_sharded_across_ranks()) throw HipError(BBMPC_E_UNSUPPORTED, "record gather on a handle whose population is sharded over ranks (every rank already holds every agent's record)");
    CHECK_PTR(d_state);
    CHECK_PTR(d_records);
    CHECK_PTR(d_gathered);
    if (d_next_state == d_state) throw HipError(BBMPC_E_INVALID, "d_next_state must not alias d_state");
    Engine* e = h->e;
    RecordComm& c = e->rc;
    if (!c.comm) throw HipError(BBMPC_E_STATE, "bbmpc_optimize_gather_dev: call bbmpc_comm_init first");
    if (slot < 0 || slot >= RecordComm::kSlots) throw HipError(BBMPC_E_INVALID, "bbmpc_optimize_gather_dev: slot must be 0 or 1");
    if (c.pending[slot]) throw HipError(BBMPC_E_STATE, "bbmpc_optimize_gather_dev: slot still pending (call bbmpc_gather_wait)");
    // single-launch control steps carry the hand-off themselves: the sequence number published by the kernel's last
    // workgroup, or (BBMPC_COMM_SYNC=event) the ready event on the kernel's dispatch packet
    uint32_t v = 0;
    if (c.sync_mode == 1) {
        v = next_comm_seq(e);
        e->tail_flag = c.flag;
        e->tail_count = c.count;
        e->tail_value = v;
    } else {
        e->tail_event = c.ready[slot];
    }
    e->tail_attached = false;
    try {
        e->optimize_dev(d_state, noise, d_records, d_next_state);
    } catch (...) {
        e->tail_event = nullptr;
        e->tail_flag = nullptr;
        throw;
    }
    e->tail_event = nullptr;
    e->tail_flag = nullptr;
    const bool published = c.sync_mode == 1 && e->tail_attached;
    gather_records(e, d_records, d_gathered, (size_t)e->A * e->rec, slot, c.sync_mode == 0 && e->tail_attached, published, v);
    API_END
}

// MPCPolicy.act of one rank of an agent-sharded run: host state in, host action / next state / reward out for the
// LOCAL agents (synchronous), plus the all-gather of their records enqueued on the communication stream, where it
// overlaps the caller's next control step (bbmpc_gather_wait(slot) before the slot is reused).
int bbmpc_optimize_gather(bbmpc_handle h, const float* state, int32_t, int32_t noise, float* action, float* next_state,
                          float* reward, float* d_gathered, int32_t slot) {
    API_BEGIN
    CHECK_HANDLE_NOSETTLE(h);            // as bbmpc_optimize: consecutive calls are ordered by the stream / the resident kernel
    h->e->invalidate_step_graph();
    // the communicator of a population-sharded handle carries the per-iteration partials on the LAUNCH stream; one RCCL
    // communicator must not be driven from two unsynchronised streams, so such a handle has no record gather (its ranks
    // all hold the same agents: there is nothing to gather).  A population that is only split into shards on THIS GPU
    // (population_size > 32768, or the loopback hook) never drives the communicator and gathers like any other handle
    if (h->e->pop_sharded_across_ranks()) throw HipError(BBMPC_E_UNSUPPORTED, "record gather on a handle whose population is sharded over ranks (every rank already holds every agent's record)");
    CHECK_PTR(state);
    CHECK_PTR(d_gathered);
    Engine* e = h->e;
    RecordComm& c = e->rc;
    if (!c.comm) throw HipError(BBMPC_E_STATE, "bbmpc_optimize_gather: call bbmpc_comm_init first");
    if (slot < 0 || slot >= RecordComm::kSlots) throw HipError(BBMPC_E_INVALID, "bbmpc_optimize_gather: slot must be 0 or 1");
    if (c.pending[slot]) throw HipError(BBMPC_E_STATE, "bbmpc_optimize_gather: slot still pending (call bbmpc_gather_wait)");
    // this rank's agents: exactly bbmpc_optimize (zero-copy state, records polled from pinned memory, resident kernel).
    // While the control step runs on the GPU the host enqueues the collective of the PREVIOUS call's records: the
    // ~20 us of host API time (H2D copy, ncclAllGather, completion word) hide under the kernel instead of sitting
    // between two control steps.  This call's own records are staged at the end and travel during the next call, or
    // when anything settles the handle (bbmpc_gather_wait on the slot, bbmpc_synchronize, any other entry point).
    const size_t nr = (size_t)e->A * e->rec;
    if (!e->d_record_slot[slot].p) e->d_record_slot[slot].alloc(nr);
    if (!e->h_record_stage[slot])
        HIP_CHECK(hipHostMalloc((void**)&e->h_record_stage[slot], nr * sizeof(float), hipHostMallocDefault));
    e->in_flight_hook = flush_deferred_gathers;
    e->in_flight_called = false;
    const float* rec = nullptr;
    try {
        rec = optimize_host(*e, state, noise, action, next_state, reward);
    } catch (...) {
        e->in_flight_hook = nullptr;
        throw;
    }
    e->in_flight_hook = nullptr;
    if (!e->in_flight_called) flush_deferred_gathers(e);
    memcpy(e->h_record_stage[slot], rec, nr * sizeof(float));      // the slot is free: its last gather was waited for (pending == false)
    e->gather_deferred[slot] = true;
    e->gather_deferred_dst[slot] = d_gathered;
    e->settle_hook = flush_deferred_gathers;
    c.pending[slot] = true;
    API_END
}

// What the communicator itself reports (ncclCommCount / ncclCommUserRank) + the hand-off mode in use:
// sync_mode 1 = sequence numbers in signal memory, 0 = events.
int bbmpc_call_stats(bbmpc_handle h, int64_t* served_resident, int64_t* launched) {
    API_BEGIN
    CHECK_HANDLE(h);
    if (served_resident) *served_resident = h->e->calls_resident;
    if (launched) *launched = h->e->calls_launched;
    API_END
}

int bbmpc_graph_stats(bbmpc_handle h, int64_t* replayed) {
    API_BEGIN
    CHECK_HANDLE(h);
    if (replayed) *replayed = h->e->calls_graph;
    API_END
}

int bbmpc_handle_device(bbmpc_handle h, int32_t* device) {
    API_BEGIN
    CHECK_HANDLE(h);
    CHECK_PTR(device);
    *device = h->e->device;
    API_END
}

int bbmpc_comm_info(bbmpc_handle h, int32_t* nranks, int32_t* rank, int32_t* sync_mode) {
    API_BEGIN
    CHECK_HANDLE(h);
    RecordComm& c = h->e->rc;
    if (!c.comm) throw HipError(BBMPC_E_STATE, "bbmpc_comm_info: no communicator (bbmpc_comm_init)");
    const Rccl& r = *c.api;
    int n = 0, me = 0;
    r.check(r.CommCount(c.comm, &n), "ncclCommCount");
    r.check(r.CommUserRank(c.comm, &me), "ncclCommUserRank");
    if (nranks) *nranks = n;
    if (rank) *rank = me;
    if (sync_mode) *sync_mode = c.sync_mode;
    API_END
}

int bbmpc_gather_wait(bbmpc_handle h, int32_t slot, int32_t host_block) {
    API_BEGIN
    CHECK_HANDLE_NOSETTLE(h);            // part of the control loop: must not stop a resident kernel
    Engine* e = h->e;
    RecordComm& c = e->rc;
    if (slot < 0 || slot >= RecordComm::kSlots) throw HipError(BBMPC_E_INVALID, "bbmpc_gather_wait: slot must be 0 or 1");
    if (c.pending[slot]) {
        if (e->gather_deferred[slot]) {
            flush_deferred_gathers(e);                           // staged by bbmpc_optimize_gather, not enqueued yet
            host_block = 1;                                      // its staging buffer is host memory: the host has to see it consumed
        } else if (e->h_record_stage[slot] && !host_block && c.sync_mode == 1 && *(volatile const uint32_t*)c.done_flag[slot] < c.done_seq[slot]) {
            host_block = 1;                                      // ditto for an enqueued one that has not finished (rare: it is a control step old)
        }
        // about to block on the collective: should it sit in a hardware queue behind this handle's resident kernel, that kernel
        // has to go first (it would otherwise leave only when its linger time is over)
        if (host_block && e->resident_alive &&
            (c.sync_mode == 0 ? hipEventQuery(c.done[slot]) != hipSuccess : *(volatile const uint32_t*)c.done_flag[slot] < c.done_seq[slot]))
            e->resident_stop();
        if (c.sync_mode == 0) {
            if (host_block) {
                HIP_CHECK(hipEventSynchronize(c.done[slot]));
            } else {
                const hipError_t q = hipEventQuery(c.done[slot]);
                if (q == hipErrorNotReady) HIP_CHECK(hipStreamWaitEvent(e->stream, c.done[slot], 0));
                else HIP_CHECK(q);
            }
        } else {
            // normally long finished (it is a control step old): then the launch stream needs no dependency at all
            volatile const uint32_t* f = c.done_flag[slot];
            const uint32_t want = c.done_seq[slot];
            if (host_block) {
                // the collective needs every rank: a dead peer shows up as a timeout, not as a hang
                const auto t0 = std::chrono::steady_clock::now();
                while (*f < want) {
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                        throw HipError(BBMPC_E_HIP, "bbmpc_gather_wait: record all-gather did not finish within 120 s");
                    std::this_thread::yield();
                }
            } else if (*f < want) {
                HIP_CHECK(hipStreamWaitValue32(e->stream, c.done_flag[slot], want, hipStreamWaitValueGte, 0xffffffffu));
            }
        }
        c.pending[slot] = false;
    }
    API_END
}

int bbmpc_comm_destroy(bbmpc_handle h) {
    API_BEGIN
    CHECK_HANDLE(h);
    h->e->invalidate_step_graph();
    h->e->rc.destroy();
    API_END
}

}  // extern "C"

