// Host-side engine behind the C ABI: owns device memory, the stream, RNG step
// counter, injected-noise buffers and traces, and enqueues the per-control-step
// kernel sequence of each optimizer.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bbmpc.h"
#include "comm.hpp"
#include "kernels_cma.hpp"
#include "kernels_eigh.hpp"
#include "kernels_eigh_small.hpp"
#include "kernels_fused.hpp"
#include "kernels_fused_cma.hpp"
#include "kernels_fused_pso.hpp"
#include "kernels_mlp.hpp"
#include "kernels_mlp_q4s.hpp"
#include "kernels_mlp_w4.hpp"
#include "kernels_mlp_wave.hpp"
#include "kernels_opt.hpp"
#include "kernels_refit.hpp"
#include "kernels_rollout.hpp"
#include "kernels_tail.hpp"
#include "kernels_user.hpp"
#include "rtc.hpp"

namespace bbmpc {

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            throw HipError(BBMPC_E_HIP, std::string(#expr) + " failed: " + hipGetErrorString(_e) + \
                                            " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define REQUIRE(cond, code, msg) \
    do {                         \
        if (!(cond)) throw HipError((code), (msg)); \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        release();
        n = count;
        if (count) HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
    }
    void zero(hipStream_t s) {
        if (p) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

hipStream_t masked_stream_from_env(const char* name, int cu_count);

struct Engine {
    bbmpc_config cfg;
    int N, A, H, U, S, HU, Nst, iters, k;
    int rec;                 // record width U+S+1
    std::vector<float> lo, hi;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int cu_count = 0;        // its compute units (kernels with spinning barriers are sized against it)
    int device = 0;          // the device the handle lives on (cfg.device, or the caller's current device when that is < 0)
    uint32_t step_counter = 0;
    // A steady-state control step of the learned-model PI2 / CEM path replayed as a hipGraph (bbmpc_optimize): its eleven
    // dependent launches follow each other ~1 us closer than enqueued one by one (tools/microbench/launch_chain.hip:
    // 2.9 us per tiny dependent kernel on a stream, 1.95 us as a graph).  The launch arguments are frozen at capture; the two
    // words that change per call -- the control step the draws are keyed by and the completion value the tail publishes --
    // are read by the kernels from device memory that the step's last kernel advances (and the host re-synchronises
    // after calls that did not go through the graph).
    hipGraphExec_t step_graph = nullptr;
    uint64_t step_graph_sig = 0;           // what the graph was captured for (mutation count, exploration-noise flag)
    uint64_t step_warm_sig = 0;            // ... and what the current run of identical calls looks like
    int step_graph_warm = 0;               // consecutive eligible calls that ran the steady-state launch sequence
    uint32_t mutations = 0;                // bumped by everything that can change what a control step launches
    uint32_t* step_words_dev = nullptr;    // device memory: [0] control step, [1] completion value, [2] arrival counter of the tail's workgroups
    uint32_t* step_words = nullptr;        // pinned staging for re-synchronising them
    uint32_t step_mirror[2] = {0, 0};      // what the device words hold now
    bool step_capturing = false;
    int64_t calls_graph = 0;               // replays so far (bbmpc_graph_stats)
    int64_t graph_capture_failures = 0;    // captures / instantiations that failed (the handle then stops trying)
    bool last_step_steady = false;         // the last optimize_dev ran without k_dist_init (kernels as they will be replayed)
    void invalidate_step_graph() {
        ++mutations;
        step_graph_warm = 0;
        if (step_graph) { (void)hipGraphExecDestroy(step_graph); step_graph = nullptr; }
    }
    bool trace_on = false, profiling = false;
    int prof_every = 1;      // events around every prof_every-th launch of the dominant kernel
    uint64_t prof_seq = 0;
    bool prof_this = false;
    int fused_mode = -1;     // -1 auto, 0 never, 1 always (env BBMPC_FUSED)
    // development / parity switches, read from the environment ONCE when the handle is created
    struct Switches {
        bool cma_svd_v1 = false, cma_svd_rounds = false, cma_svd_general = false, cma_svd_gram = false, cma_fused = false, cma_coop = false;
        bool cma_eigh_fail = false;    // BBMPC_CMA_EIGH_FAIL: test hook, the direct solver reports every instance as failed
        int cma_eigh = 1;              // BBMPC_CMA_EIGH=0: block Jacobi instead of the direct eigensolver (kernels_eigh.hpp) at 128 < n <= 320
        int cma_nb = 0;                // BBMPC_CMA_NB: 8 / 16 column blocks in the block Jacobi (0 = automatic)   // BBMPC_CMA_SVD_V1 / _ROUNDS / _GENERAL
        bool mlp_generic = false;      // BBMPC_MLP_GENERIC
        int mlp_bf16 = 0;              // BBMPC_MLP_BF16: 0 off (default, fp32), 1 plain bf16 inputs, 3 split bf16 (hi+lo, three products)
        int mlp_pair = -1, mlp_q4 = -1;   // BBMPC_MLP_PAIR / BBMPC_MLP_Q4: -1 automatic, 0 / 1 forced
        int mlp_w4 = 1;                   // BBMPC_MLP_W4=0: keep k_rollout_mlp_wave where k_rollout_mlp_w4 (hidden <= 32) would run
        int mlp_q4s = 1;                  // BBMPC_MLP_Q4S=0 (and the older spelling BBMPC_MLP_Q4R=0): keep k_rollout_mlp_q4 where k_rollout_mlp_q4s would run
        int cma_small3 = 1;               // BBMPC_CMA_SMALL3=0: n <= 32 keeps one launch per phase (eleven per iteration) instead of sample | roll out | update
        int step_graph = 1;               // BBMPC_STEP_GRAPH=0: never replay a control step as a hipGraph
        int pi2_skip_init = 1;            // BBMPC_PI2_SKIP_INIT=0: k_dist_init opens every PI2 / CEM control step on the learned-model path too
        int refit_wgs = 0;                // BBMPC_REFIT_WGS=n: workgroups per agent in k_refit_cem_v2 (0 = by problem size)
        int mlp_wave = 1;                 // BBMPC_MLP_WAVE=0: never the one-wave-per-tile kernel for small networks
        int linger_us = 200;              // BBMPC_LINGER_US: how long a one-agent control-step kernel waits for the next call (0 = never)
        int balance = 0;               // BBMPC_BALANCE=1: SIMD mates pace each other in the fused pendulum rollout
        int ilp = 1;                   // BBMPC_ILP
        bool refit_v1 = false;         // BBMPC_REFIT_V1
        bool zero_copy = true;         // !BBMPC_NO_ZERO_COPY
        bool mlp_no_half_tail = false; // BBMPC_MLP_NO_HALF_TAIL: keep hidden features in index order (no skipped MFMAs)
        bool host_poll = true;         // !BBMPC_NO_HOST_POLL: host calls return on the kernel's own completion word
        bool dbg = false;              // BBMPC_DBG
    } sw;

    // device state
    DevBuf<float> d_lo, d_hi, d_state, d_record, d_action;
    DevBuf<float> d_prev_mean, d_var0, d_mean, d_var, d_sigma;
    DevBuf<float> d_samples, d_rewards, d_penalty;
    DevBuf<int> d_elites;
    // learned dynamics (bbmpc_set_mlp)
    bool mlp_ready = false;
    MlpDesc mlp;
    int mlp_nw = 1;
    DevBuf<float> d_wpack[MLP_MAX_LAYERS], d_wpack4[MLP_MAX_LAYERS], d_bpack[MLP_MAX_LAYERS], d_wraw[MLP_MAX_LAYERS], d_braw[MLP_MAX_LAYERS], d_wq4[MLP_MAX_LAYERS], d_wq4s0, d_w4pack, d_wbf[MLP_MAX_LAYERS], d_stats;
    DevBuf<float> d_fin_next, d_fin_rew, d_step_act;
    // SPSA / PSO state (internal layout)
    DevBuf<float> d_cand_a, d_cand_b, d_rewards2, d_vel, d_pbest, d_pbest_r, d_gbest, d_gbest_r, d_cond;
    DevBuf<int> d_gidx;
    DevBuf<float> t_rewards2;
    bool pso_seeded = false;
    // CMA-ES state
    int cma_G = 0, cma_n = 0;
    CmaConst cma_c;
    DevBuf<float> c_w, c_m, c_sigma, c_C, c_B, c_Dd, c_ps, c_pc, c_BD, c_z, c_Ye, c_xm, c_ym, c_evec, c_eval, c_E;
    DevBuf<int> c_eidx, c_info;
    DevBuf<unsigned> c_sync;       // SVD instance barriers / sweep flags [G][32]
    // direct eigensolver scratch (kernels_eigh.hpp): tridiagonal, reflectors, eigenvectors in the tridiagonal basis, flags
    DevBuf<float> e_d, e_e, e_tau, e_Vt, e_alpha, e_lam, e_Z, e_Z2, e_P, e_Tf;
    DevBuf<unsigned> e_flags;
    // the one-workgroup direct solver of kernels_eigh_small.hpp
    bool cma_use_eigh_small() const { return sw.cma_eigh && cma_n >= 2 && cma_n <= ES_N && !sw.cma_svd_v1 && !sw.cma_svd_rounds && !sw.cma_svd_general && !sw.cma_svd_gram; }
    bool cma_use_eigh() const { return sw.cma_eigh && cma_n > 128 && cma_n <= EIGH_MAX_N && (cma_n & 3) == 0 && !sw.cma_svd_v1 && !sw.cma_svd_rounds && !sw.cma_svd_general && !sw.cma_svd_gram; }
    void cma_eigh_launch(const CmaArgs& q);
    // experiment switch BBMPC_EIGH_SIDE_CUS=lo-hi: the tridiagonalisation on a side stream confined to those CUs (event hand-offs)
    hipStream_t eigh_side = nullptr;
    hipEvent_t eigh_ev[2] = {nullptr, nullptr};
    int eigh_side_state = -1;         // -1 undecided, 0 off, 1 on
    // evaluate() scratch (grown on demand)
    DevBuf<float> d_eval_seq, d_eval_rew, d_step_a, d_step_b, d_step_c, d_step_d;
    // injected noise (internal layout), keyed by BBMPC_NOISE_*
    std::map<int, DevBuf<float>> inj;
    // traces
    DevBuf<float> t_rewards, t_mean, t_var, t_samples;
    DevBuf<int> t_elites;
    DevBuf<int> t_cma_stats;                     // [iters][G][16] BBMPC_TRACE_CMA_SVD_STATS
    DevBuf<float> t_cma_B, t_cma_C, t_cma_D;   // CMA-ES: eigenvectors / covariance / sqrt eigenvalues after each iteration
    // pinned staging
    float* h_pin = nullptr;
    float* h_pin_dev = nullptr;   // device address of h_pin (looked up once per allocation)
    size_t h_pin_n = 0;
    // noise prefetch for the persistent kernel: the standard draws of control step t+1 are generated by otherwise
    // idle CUs on a side stream while step t's kernel runs (same Philox counters => bit-identical to in-kernel draws)
    DevBuf<float> d_noise_pf[2];      // two chunks of pf_steps control steps each
    RecordComm rc;           // multi-GPU record all-gather (comm.hpp); unused until bbmpc_comm_init
    DevBuf<float> d_record_slot[RecordComm::kSlots];   // bbmpc_optimize_gather: the all-gather of step t reads its records while step t+1 runs
    float* h_record_stage[RecordComm::kSlots] = {nullptr, nullptr};   // pinned: the rank's records on their way host -> HBM (communication stream)
    // bbmpc_optimize_gather enqueues the collective of a control step while the NEXT one runs on the GPU (its ~20 us of
    // host API time would otherwise sit between two control steps): staged records whose gather is not enqueued yet
    bool gather_deferred[RecordComm::kSlots] = {false, false};
    float* gather_deferred_dst[RecordComm::kSlots] = {nullptr, nullptr};
    void (*settle_hook)(Engine*) = nullptr;      // flushes them when anything settles the handle
    void (*in_flight_hook)(Engine*) = nullptr;   // called once per host-in / host-out control step, right after it was handed to the GPU
    bool in_flight_called = false;
    hipEvent_t tail_event = nullptr;   // completion event wanted on the control step's last kernel (launch_with_tail)
    bool tail_attached = false;
    uint32_t* tail_flag = nullptr;     // or: sequence number the last kernel should publish itself (RecordComm::flag / host_done)
    uint32_t tail_value = 0;
    uint32_t* tail_count = nullptr;    // arrival counter that goes with tail_flag (device memory)
    // host-in/host-out calls of a single-kernel control step return as soon as the kernel has published its records
    // into host memory; the stream is joined lazily by the next call that needs it (settle)
    uint32_t* host_done = nullptr;     // pinned host word the kernel publishes to
    uint32_t* host_done_dev = nullptr; // its device address
    uint32_t* host_count = nullptr;    // device memory arrival counter
    uint32_t host_seq = 0;
    bool lazy_sync = false;
    void settle();
    // resident control-step kernel (one agent, persistent pendulum kernel, host-in / host-out calls): the kernel of one
    // call stays on the GPU for linger_us and takes the next call's request from a pinned mailbox line (kernels_fused.hpp)
    const float* stage_state_src = nullptr;   // pinned [A,S] state that the control step's first kernel (k_dist_init) copies to d_state
    bool linger_launch = false;        // bbmpc_optimize asks optimize_fused for the LINGER variant
    bool resident_alive = false;       // a LINGER kernel may still be polling the mailbox
    // pinned block host_done, 16-word lines: 0 completion word of the launch-per-call paths | 1..A per-agent completion
    // words | A+1..2A request lines | 2A+1..3A exit words | 3A+1.. agent map of a subset launch (one int32 per agent:
    // ceil(A/16) lines -- every agent but one may be listed)
    static constexpr int kLingerMaxAgents = 64;
    int sync_lines() const {
        const int al = std::max(1, std::min(A, kLingerMaxAgents));
        return 1 + 3 * al + (al + 15) / 16;
    }
    uint32_t* ack_host(int a) { return host_done + 16 * (1 + a); }
    uint32_t* mbox_host(int a) { return host_done + 16 * (1 + std::min(A, kLingerMaxAgents) + a); }
    uint32_t* gone_host(int a) { return host_done + 16 * (1 + 2 * std::min(A, kLingerMaxAgents) + a); }
    int32_t* amap_host() { return reinterpret_cast<int32_t*>(host_done + 16 * (1 + 3 * std::min(A, kLingerMaxAgents))); }
    uint32_t* sync_dev(const void* host_ptr) { return host_done_dev + (reinterpret_cast<const uint32_t*>(host_ptr) - host_done); }
    // Another handle of the process starting work on the same device asks the resident workgroups of this one to leave
    // (streams share a few hardware queues: a lingering kernel would hold back whatever lands behind it): it writes the stop
    // word into the request lines published here, nothing else -- this handle finds the exit words at its next call.
    std::atomic<uint32_t*> mbox_pub{nullptr};      // request lines of a (possibly) live resident kernel, null otherwise
    int mbox_pub_agents = 0;
    int subset_n = 0;                  // > 0: the next persistent-kernel launch covers only the agents listed in amap_host()
    int linger_test_quit = -1;         // BBMPC_LINGER_TEST_QUIT (test hook, kernels_fused.hpp)
    int64_t calls_resident = 0, calls_launched = 0;   // bbmpc_call_stats
    bool resident_step(const float* state, int add_noise, uint32_t seq);
    void resident_stop();
    hipStream_t pf_stream = nullptr;
    hipEvent_t pf_done[2] = {nullptr, nullptr}, pf_free = nullptr;
    int64_t pf_chunk[2] = {-1, -1};   // chunk id (control step / pf_steps) a buffer holds
    bool pf_waited[2] = {false, false}, pf_inflight[2] = {false, false};
    int pf_steps = 0;                 // control steps per chunk
    size_t pf_step_floats = 0;        // floats per control step
    int pf_mode = -1;                 // -1 undecided, 0 off, 1 on (env BBMPC_NOISE_PREFETCH)
    void launch_noise_fill(int64_t chunk, int buf, hipStream_t on);
    // profiling
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    const char* dominant_kernel = "k_rollout_pendulum";
    // the template instantiation of the dominant kernel that the last control step launched, spelt the way rocprofv3 prints
    // kernel names (minus "void "): the key bench.py matches the committed PMC profiles on.  Empty = the plain name is unique.
    char dominant_inst[128] = "";

    explicit Engine(const bbmpc_config& c);
    ~Engine();

    RngKey key(uint32_t step) const {
        RngKey kk;
        kk.k0 = (uint32_t)(cfg.seed & 0xffffffffu);
        kk.k1 = (uint32_t)(cfg.seed >> 32);
        kk.step = step;
        kk.q_per_agent = (uint32_t)((HU + 3) / 4);
        kk.step_src = step_capturing ? step_words_dev : nullptr;
        return kk;
    }
    bool fix(uint32_t bit) const { return (cfg.quirks & bit) != 0; }
    // user-supplied reward / dynamics device functions (rtc.hpp) and the step-wise evaluator that calls them
    UserFunction user_reward, user_dynamics, user_rollout;   // user_rollout: the fused lane-per-trajectory kernel (rtc.hpp), built lazily
    bool user_rollout_stale = true;
    bool user_stepwise_only = false;   // BBMPC_USER_STEPWISE: never fuse (test / comparison hook)
    void rollout_user_fused(int mode, bool pen, RolloutArgs& ra);
    void rollout_mlp_user_reward(int mode, bool pen, RolloutArgs& ra);
    DevBuf<float> u_traj;              // [H][A][Nst][S] states after every step of the MFMA rollout
    float* mlp_traj_out = nullptr;     // set around a launch_rollout_mlp call that should record the trajectory
    // set around a launch_rollout_mlp call whose state argument is the pinned host buffer: a kernel that can store the
    // state for the later launches itself (k_rollout_mlp_q4s) takes the request and clears it
    float* mlp_state_copy = nullptr;
    bool pi2_dist_ready = false;       // d_sigma holds the constructor variance's root (PI2 never changes it)
    bool pi2_copy_seen = false;        // the previous control step's first rollout took such a request
    bool cem_sigma0_ready = false;     // d_sigma0 holds cem_sigma(prev_mean, var0): constant while CEM restarts from the constructor distribution
    DevBuf<float> d_sigma0;
    DevBuf<float> u_rows, u_x0, u_x1, u_total, u_pen, u_next;
    bool user_path() const { return cfg.reward == BBMPC_REW_USER || cfg.dynamics == BBMPC_DYN_USER; }
    int builtin_reward_kind() const { return cfg.reward == BBMPC_REW_USER ? REW_NONE : cfg.reward; }
    void set_user_source(int kind, const char* src);
    void set_user_callback(int kind, bbmpc_rows_callback fn, void* user);
    bool user_callbacks() const { return user_reward.cb != nullptr || user_dynamics.cb != nullptr; }
    DevBuf<float> u_cb_rew;            // rewards of one planning step as a callback wrote them
    void rollout_stepwise(int mode, bool pen, RolloutArgs& ra);
    void dynamics_rows(const float* d_states, const float* d_actions, int astride, int batch, float* d_next);
    void reward_rows(const float* d_cur, const float* d_next, const float* d_actions, int astride, int batch, float* d_total, int accumulate);
    void mlp_forward_rows(const float* d_x, int batch, float* d_out);
    // population sharding (PI2): per-iteration partials of this rank and the gathered partials of all ranks
    DevBuf<float> ps_part, ps_all;
    DevBuf<int> c_eidx_glob;                  // sharded CMA-ES: global particle indices of the merged elites (parity trace)
    int auto_split = 0;      // > 1: population_size > 32768 is played as this many loopback shards (bbmpc.hip, constructor)
    int ps_loopback = 0;     // BBMPC_POPSHARD_LOOPBACK=G: one handle plays all G shards in turn (single-GPU test / measurement hook)
    bool ps_force = false;   // BBMPC_POPSHARD_FORCE: take the sharded code path (incl. the collective) even with one shard
    bool pop_sharded() const { return cfg.population_global > N || ps_loopback > 1 || ps_force; }
    // ... and over RANKS (the per-iteration exchange goes through the communicator): not the one-GPU splits
    bool pop_sharded_across_ranks() const { return ps_force || (ps_loopback <= 1 && cfg.population_global > N); }
    // CMA-ES at n <= 32 on the analytic pendulum: the last iteration's update launch is held back until finalize() knows the
    // record's arguments and then carries the tail of the control step as well (kernels_eigh_small.hpp)
    struct PendingCmaUpdate { bool set = false; CmaArgs q; size_t lds = 0; int yef = 0; } pending_cma_update;
    int pending_warm = 0;    // learned-dynamics path: warm start the tail kernel performs (kernels_tail.hpp TailArgs::warm_mode)
    RowMlp row_mlp() const;
    bool any_injected() const {
        for (const auto& kv : inj) if (kv.second.p) return true;
        return false;
    }
    const float* injected(int kind) const {
        auto it = inj.find(kind);
        return (it == inj.end() || !it->second.p) ? nullptr : it->second.p;
    }
    float* pinned(size_t count);

    void reset();
    void optimize_dev(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out);
    void launch_pending_cma_update(const FinalArgs& fa);
    bool use_fused_pso() const;
    void optimize_fused_pso(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step);
    bool use_fused_cma() const;
    void optimize_fused_cma(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step);
    bool use_fused() const;
    void optimize_fused(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step);
    void ensure_trace();
    void optimize_spsa(RolloutArgs& ra, uint32_t step);
    void optimize_pso(RolloutArgs& ra, uint32_t step);
    void optimize_cma(RolloutArgs& ra, uint32_t step);
    void cma_init();
    void cma_reset_mean_sigma();
    CmaArgs cma_args(uint32_t step, uint32_t iter);
    OptArgs opt_args(uint32_t step, uint32_t iter) const;
    PsoState pso_state(int shard = 0);       // shard > 0: the loopback hook's further copies of the per-particle state
    void evaluate_dev(const float* d_state_in, const float* d_seq, int n_pop, float* d_rew_out);
    void step_dev(const float* d_states, const float* d_actions, int astride, int batch, float* d_next, float* d_rew);
    void reward_dev(const float* d_cur, const float* d_next, const float* d_act, int batch, float* d_rew);

    void inject(int kind, const float* data, int64_t count);
    void dump_noise(int kind, int control_step, int iteration, float* out, int64_t count);
    void get_trace(int iteration, int item, void* out, int64_t bytes);
    void get_state(const std::string& name, float* out, int64_t count);
    void set_state(const std::string& name, const float* data, int64_t count);
    void get_profile(double* ms, int64_t* launches);

    // helpers
    void launch_rollout(int mode, bool pen, RolloutArgs& ra);
    void launch_rollout_mlp(int mode, bool pen, RolloutArgs& ra, bool per_particle_state, float* final_state);
    void set_mlp(int n_layers, const int32_t* dims, const int32_t* acts, const float* const* w, const float* const* b,
                 int is_normalized, const float* const* stats);
    void prof_begin();
    void prof_end();
    void capture_trace(int it);
    void finalize(const float* d_state_in, int add_noise, float* d_record_out, float* d_next_out, uint32_t step);
    void to_internal(const float* ref, int n_pop, float* internal_host) const;     // [n,A,H,U] -> [A][HU][Nst]
    void from_internal(const float* internal_host, int n_pop, float* ref) const;
};

}  // namespace bbmpc
