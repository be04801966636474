/*
 * bbmpc.h -- C ABI of the MI355X-native sampling-MPC rollout engine.
 *
 * The reference (ossamaAhmed/blackbox_mpc v0.3) has no FFI: its hot path is a
 * Python class API that hands one tf.function graph call per control step to
 * TensorFlow.  This header declares what a maintainer of the reference would
 * bind (ctypes stub in INTEGRATION.md) to replace that graph call.  Each entry
 * point cites the reference interface it stands in for; paths are relative to
 * /root/reference/blackbox_mpc/.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative
 *     BBMPC_E_* code; bbmpc_last_error() gives the message for the calling thread.
 *   - host-pointer entry points are synchronous: outputs are valid on return.
 *     *_dev entry points take device (HBM) pointers, enqueue on the handle's
 *     stream and return immediately.
 *   - the caller owns every buffer it passes; the handle owns device memory,
 *     stream, events.  One handle = one caller thread at a time (the reference's
 *     optimizers hold mutable tf.Variable state and are not re-entrant either).
 *   - tensors are float32, C-contiguous, in the reference's layouts:
 *       states [A,S], action sequences [N,A,H,U], rewards [N,A].
 *   - there is NO CPU fallback: every compute entry point fails with
 *     BBMPC_E_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef BBMPC_H
#define BBMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBMPC_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------- */
#define BBMPC_OK              0
#define BBMPC_E_INVALID      -1   /* bad argument / unsupported configuration */
#define BBMPC_E_NO_DEVICE    -2   /* no usable HIP device                      */
#define BBMPC_E_HIP          -3   /* a HIP runtime call failed                 */
#define BBMPC_E_STATE        -4   /* call sequence error (e.g. MLP weights not set) */
#define BBMPC_E_UNSUPPORTED  -5   /* valid in the reference, not built yet     */

/* ---- enums -------------------------------------------------------------- */
/* optimizers/: random_search.py, cem.py, pi2.py, pso.py, cma_es.py, spsa.py */
#define BBMPC_OPT_NONE          0   /* evaluator-only handle */
#define BBMPC_OPT_RANDOM_SEARCH 1
#define BBMPC_OPT_CEM           2
#define BBMPC_OPT_PI2           3
#define BBMPC_OPT_PSO           4
#define BBMPC_OPT_CMAES         5
#define BBMPC_OPT_SPSA          6

/* dynamics plugin: utils/pendulum.py:38-92 | dynamics_functions/deterministic_mlp.py:5-51 */
#define BBMPC_DYN_PENDULUM 1
#define BBMPC_DYN_MLP      2
#define BBMPC_DYN_USER     3   /* caller-supplied: HIP source (bbmpc_set_dynamics_source) or a device-memory callback */

/* reward plugin: utils/pendulum.py:10-35 | tutorials/mujoco/cost_func.py:5-22 */
#define BBMPC_REW_PENDULUM 1
#define BBMPC_REW_CHEETAH  2
#define BBMPC_REW_USER     3   /* caller-supplied: HIP source (bbmpc_set_reward_source) or a device-memory callback */

/* Dense activations (tutorials use tf.math.tanh and None) */
#define BBMPC_ACT_NONE    0
#define BBMPC_ACT_TANH    1
#define BBMPC_ACT_RELU    2
#define BBMPC_ACT_SIGMOID 3

/* quirk switches; a set bit OPTS OUT of the reference's as-executed behaviour
 * (SURVEY.md section 8 quirk register).  Default 0 = bug-compatible. */
#define BBMPC_FIX_Q1_REWARD_ARG_ORDER   (1u << 0)  /* pendulum reward uses actions, not next_state */
#define BBMPC_FIX_Q2_CEM_WARM_START     (1u << 1)  /* CEM keeps shifted mean across control steps  */
#define BBMPC_FIX_Q7_EXPL_NOISE_ZERO_MEAN (1u << 2)
#define BBMPC_CMAES_PER_AGENT           (1u << 8)  /* independent CMA-ES per agent (deviation, shards) */
#define BBMPC_STRICT_MATH               (1u << 9)  /* pendulum rollouts op-for-op as the reference (atan2/sincos every
                                                      step) instead of carrying the angle; same tolerances, slower */

/* noise kinds for bbmpc_inject_noise (standard draws, reference layout) */
#define BBMPC_NOISE_TRUNC_NORMAL 1  /* [iters][N,A,H,U] unit normal, |z|<2: cem.py:90 pi2.py:65 */
#define BBMPC_NOISE_UNIFORM      2  /* [N,A,H,U] U[0,1): random_search.py:40                     */
#define BBMPC_NOISE_RADEMACHER   3  /* [iters][N,A,H,U] +-1: spsa.py:73-75                        */
#define BBMPC_NOISE_NORMAL       4  /* CMA-ES [iters][N,n] N(0,1): cma_es.py:139                  */
#define BBMPC_NOISE_PSO_SCALARS  5  /* [iters][2] N(0,1): pso.py:107-109                          */
#define BBMPC_NOISE_PSO_RESEED_TRUNC   6  /* [N,A,H,U]: pso.py:121-127 */
#define BBMPC_NOISE_PSO_RESEED_UNIFORM 7  /* [N,A,H,U]: pso.py:130-131 */
#define BBMPC_NOISE_PSO_RESET_POS      8  /* [N,A,H,U]: pso.py:147-149 */
#define BBMPC_NOISE_PSO_RESET_VEL      9  /* [N,A,H,U]: pso.py:151-152 */
#define BBMPC_NOISE_EXPLORATION       10  /* [A,U] unit trunc normal: optimizer_base.py:83-86 */

/* trace items for bbmpc_get_trace (per-iteration parity data) */
#define BBMPC_TRACE_REWARDS 1   /* [N,A]  (SPSA: [2N,A], plus then minus)   */
#define BBMPC_TRACE_MEAN    2   /* [A,H,U] distribution mean / solution after the iteration */
#define BBMPC_TRACE_VAR     3   /* [A,H,U] (CEM)                             */
#define BBMPC_TRACE_ELITES  4   /* int32 [A,k] (CEM), [A] best index (RandomSearch/PSO) */
#define BBMPC_TRACE_SAMPLES 5   /* [N,A,H,U] action sequences that were rolled out */
#define BBMPC_TRACE_CMA_B   6   /* CMA-ES [G,n,n] eigenvectors B after the iteration (cma_es.py:195-198,204)   */
#define BBMPC_TRACE_CMA_C   7   /* CMA-ES [G,n,n] covariance C after the iteration (cma_es.py:183-190,202)     */
#define BBMPC_TRACE_CMA_SVD_STATS 9 /* CMA-ES int32 [G,16]: the iteration's eigen-decomposition, for search dimensions 32 < n <= 512.
                                     * [0..14] column pairs the Jacobi rotated in sweep s (all zero when the direct solver's result
                                     * was taken).  [15]: with the direct solver (128 < n <= 320) 1 = it refused the instance and the
                                     * Jacobi ran for it; on the Jacobi-only paths 1 = sweep 0 rotated something.  n <= 32 (one
                                     * workgroup factorises, falls back and finishes in one kernel) records nothing: all zero. */
#define BBMPC_TRACE_CMA_D   8   /* CMA-ES [G,n]   diag(D) = sqrt(eigenvalues) after the iteration (:197,205)   */

typedef struct bbmpc_handle_s* bbmpc_handle;

/* Mirrors the constructor kwargs of OptimizerBase + the six optimizers
 * (optimizer_base.py:6-50; cem.py:7-10; pi2.py:9-11; random_search.py:7-8;
 *  pso.py:7-11; cma_es.py:7-10; spsa.py:7-12). */
typedef struct bbmpc_config {
    int32_t abi_version;        /* BBMPC_ABI_VERSION */
    int32_t optimizer;          /* BBMPC_OPT_* */
    int32_t dynamics;           /* BBMPC_DYN_* */
    int32_t reward;             /* BBMPC_REW_* */
    int32_t population_size;    /* N; above 32768 it must divide into <= 64 equal shards of <= 32768 (played in turn on the one GPU) */
    int32_t num_agents;         /* A  (agents owned by THIS handle / GPU) */
    int32_t planning_horizon;   /* H */
    int32_t dim_u;              /* U = env_action_space.shape[0] */
    int32_t dim_s;              /* S = env_observation_space.shape[0] */
    int32_t max_iterations;     /* ignored by RandomSearch */
    int32_t num_elite;          /* CEM, CMA-ES */
    int32_t agent_offset;       /* global id of local agent 0 (agent sharding; RNG is keyed by global id) */
    int32_t num_agents_global;  /* total agents across all shards (>= num_agents) */
    int32_t device;             /* HIP device ordinal, -1 = current */
    uint32_t quirks;            /* BBMPC_FIX_* / mode bits */
    uint32_t reserved0;
    uint64_t seed;              /* engine RNG seed (Philox key) */
    float alpha;                /* CEM smoothing (cem.py:10) */
    float lamda;                /* PI2 temperature (pi2.py:11) */
    float pso_c1, pso_c2, pso_w, pso_v0_fraction;           /* pso.py:9-11 */
    float spsa_alpha, spsa_gamma, spsa_a, spsa_c;           /* spsa.py:9-12 */
    float cma_alpha_cov, cma_h_sigma;                       /* cma_es.py:10 */
    const float* action_low;    /* [U] env_action_space.low  */
    const float* action_high;   /* [U] env_action_space.high */
    /* population sharding for num_agents < n_gpus (all six optimizers): this handle rolls out particles
     * [population_offset, population_offset + population_size) of a population of population_global particles that
     * every rank shares for the SAME agents; per iteration the ranks exchange (min cost, sum of weights, weighted
     * sums [H*U]) per agent (PI2, pi2.py:80-87) or their local top-k with the sample rows (CEM, cem.py:97-112) over the
     * handle's communicator (bbmpc_comm_init) and merge them in rank order, so every rank refits the same distribution.
     * 0 / 0 = not sharded. */
    int32_t population_offset;
    int32_t population_global;
} bbmpc_config;

/* ---- library ------------------------------------------------------------ */
int         bbmpc_abi_version(void);
const char* bbmpc_last_error(void);
/* number of usable gfx950 devices (0 when none; never fails) */
int         bbmpc_device_count(void);

/* ---- lifecycle ---------------------------------------------------------- */
/* Stands in for MPCPolicy.__init__ / Optimizer.__init__ + set_trajectory_evaluator
 * (policies/mpc_policy.py:58-122; optimizer_base.py:105-115). */
int bbmpc_create(const bbmpc_config* cfg, bbmpc_handle* out);
int bbmpc_destroy(bbmpc_handle h);
/* Launch on a caller-provided hipStream_t; NULL = the handle's own (non-blocking) stream.  Note that the legacy
 * default stream's handle IS NULL (PyTorch's current stream, unless the caller entered a side stream): use
 * bbmpc_set_stream_default to launch there -- or, better, run the control loop on a side stream. */
int bbmpc_set_stream(bbmpc_handle h, void* hip_stream);
int bbmpc_set_stream_default(bbmpc_handle h);   /* the legacy default (NULL) stream */

/* DeterministicMLP weights + SystemDynamicsHandler normalisation stats
 * (deterministic_mlp.py:5-25 Dense kernels [in,out] row-major, biases [out];
 *  system_dynamics_handler.py:84-95 six stats vectors).
 * dims has n_layers+1 entries, dims[0] = S+U, dims[n_layers] = S.
 * stats = {mean_states[S], std_states[S], mean_actions[U], std_actions[U],
 *          mean_targets[S], std_targets[S]} or NULL when is_normalized == 0. */
int bbmpc_set_mlp(bbmpc_handle h, int32_t n_layers, const int32_t* dims, const int32_t* activations,
                  const float* const* weights, const float* const* biases,
                  int32_t is_normalized, const float* const* stats);

/* User-supplied plug-ins.  The reference takes ANY callable as reward_function / dynamics_function
 * (trajectory_evaluators/deterministic.py:13-18; called at :65-66 as reward(current_state, actions, next_state) and at
 * :99-100 as dynamics(x[B,S+U], train=False) -> delta[B,S]).  A Python callable cannot run inside a kernel; the
 * counterpart is HIP source, compiled at run time (hiprtc, gfx950), that defines per row
 *     __device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U);
 *     __device__ void  bbmpc_user_dynamics(const float* x, float* delta, int S, int U);     (x = [state | action])
 * for a handle created with BBMPC_REW_USER / BBMPC_DYN_USER (user dynamics are a true model: next = state + delta,
 * utils/transforms.py:34).  Any combination with the built-in plug-ins works; evaluation then runs step by step
 * (one batched dynamics and one batched reward launch per planning step, as the reference's own graph does) instead
 * of through the built-in pairs' fused kernels -- unless the dynamics is analytic (user or PendulumTrueModel): then the
 * engine compiles ONE fused lane-per-trajectory rollout kernel with the user function(s) inlined.  Compile errors come back as BBMPC_E_INVALID with the compiler log in
 * bbmpc_last_error().  bbmpc_check_user_source only compiles (needs no GPU): kind 1 = reward, 2 = dynamics. */
int bbmpc_set_reward_source(bbmpc_handle h, const char* hip_source);
int bbmpc_set_dynamics_source(bbmpc_handle h, const char* hip_source);
int bbmpc_check_user_source(int32_t kind, const char* hip_source, int32_t dim_s, int32_t dim_u);
/* The same two plug-ins as HOST callbacks that work on DEVICE memory -- for callers whose functions are written in a
 * GPU array framework (the reference's users pass TensorFlow callables, deterministic.py:13-18; the Python layer wraps
 * PyTorch callables this way).  The engine calls back once per planning step with row batches that
 * live in its own HBM buffers; the callback enqueues its work on `hip_stream` (the handle's launch stream) and
 * returns without synchronising -- nothing is copied to the host, nothing runs on the CPU but the enqueueing.
 *   reward:   d_cur [batch,S], d_actions [batch,U], d_next [batch,S]  ->  d_out [batch]     (the reference's CALL order)
 *   dynamics: d_cur [batch,S], d_actions [batch,U], d_next = NULL     ->  d_out [batch,S] = the ABSOLUTE next states, i.e.
 *             process_output(state, f(process_input(state, action))) of dynamics_handlers/system_dynamics_handler.py:97-161
 * A non-zero return aborts the control step with BBMPC_E_INVALID.  A handle with a callback evaluates step by step
 * (2 * H + 2 launches plus the callback's own); setting a callback replaces HIP source of the same kind and vice versa.
 * fn == NULL clears it. */
typedef int32_t (*bbmpc_rows_callback)(void* user, const float* d_cur, const float* d_actions, const float* d_next,
                                       int32_t batch, float* d_out, void* hip_stream);
int bbmpc_set_reward_callback(bbmpc_handle h, bbmpc_rows_callback fn, void* user);
int bbmpc_set_dynamics_callback(bbmpc_handle h, bbmpc_rows_callback fn, void* user);
/* Compile-only check of the FUSED rollout kernel the engine builds for analytic models (one lane per trajectory, the
 * user function(s) inlined next to the built-in PendulumTrueModel / rewards): dynamics / reward = BBMPC_DYN_* /
 * BBMPC_REW_* kinds, sources NULL for built-ins.  Needs no GPU. */
int bbmpc_check_user_rollout(int32_t dynamics, int32_t reward, const char* dynamics_source, const char* reward_source,
                             int32_t dim_s, int32_t dim_u);
/* DeterministicMLP.__call__(x[B, S+U], train) -> [B, S]: the raw Dense stack on already-processed inputs
 * (dynamics_functions/deterministic_mlp.py:27-51), no normalisation, no residual. */
int bbmpc_mlp_forward(bbmpc_handle h, const float* x, int32_t batch, float* out);

/* SystemDynamicsHandler.process_input(states[B,S], actions[B,U]) -> [B,S+U] and .process_output(states[B,S],
 * raw_output[B,S]) -> next_states[B,S] as stand-alone calls (dynamics_handlers/system_dynamics_handler.py:97-161 +
 * utils/transforms.py:20-34).  stats = the six statistics vectors in bbmpc_set_mlp's order for a normalised learned
 * model, NULL for a true model / un-normalised handler (plain concat; next = raw + state).  Inside rollouts the same
 * arithmetic is fused into the kernels (prologue / epilogue). */
int bbmpc_process_input(bbmpc_handle h, const float* states, const float* actions, int32_t batch,
                        const float* const* stats, float* out);
int bbmpc_process_output(bbmpc_handle h, const float* states, const float* raw_output, int32_t batch,
                         const float* const* stats, float* out);

/* Optimizer.reset()  (cem.py:138-149, pi2.py:98-105, pso.py:143-160, cma_es.py:215-227, spsa.py:119-127) */
int bbmpc_reset(bbmpc_handle h);

/* ---- hot path ----------------------------------------------------------- */
/* OptimizerBase.__call__(current_state, time_step, add_exploration_noise)
 * -> (action[A,U], next_state[A,S], reward[A])      optimizer_base.py:55-95
 * Host buffers in, host buffers out, synchronous.  On the analytic path with up to 64 agents the control step's kernel
 * stays resident on the GPU for BBMPC_LINGER_US (default 200) after the call returns and serves the next bbmpc_optimize
 * without a launch; any other entry point of the handle stops it first, and a device-wide synchronisation by the
 * caller waits at most that long.  Results do not depend on it (BBMPC_LINGER_US=0: one launch per call). */
int bbmpc_optimize(bbmpc_handle h, const float* state, int32_t time_step, int32_t add_exploration_noise,
                   float* action, float* next_state, float* reward);
/* How the host-in / host-out calls (bbmpc_optimize, bbmpc_optimize_gather) of this handle were served so far: by the
 * resident kernel of the previous call (request mailbox, no launch) or by launching.  Either pointer may be NULL.
 * No counterpart in the reference (its act() is one TF graph call per control step, policies/mpc_policy.py:160-164);
 * exists so that a benchmark can say which of the two paths its number was measured on. */
int bbmpc_call_stats(bbmpc_handle h, int64_t* served_resident, int64_t* launched);
/* ... and how many of the launched ones were replays of a captured hipGraph (the steady-state control step of the
 * learned-model PI2 / CEM path: same launches, same arguments every call).  No counterpart in the reference. */
int bbmpc_graph_stats(bbmpc_handle h, int64_t* replayed);
/* (A capture or instantiation that fails switches the replay off for the handle and the calls go on as plain launches;
 * BBMPC_TRACE_GRAPH=1 in the environment reports it on stderr.) */
/* The GPU the handle lives on: bbmpc_config.device, or the caller's current HIP device at bbmpc_create when that was < 0.
 * Callers that alias the handle's HBM buffers or stream in another framework must do so on THIS device. */
int bbmpc_handle_device(bbmpc_handle h, int32_t* device);

/* Same with device pointers; record is [A, U+S+1] = (action | next_state | reward) per agent.
 * d_next_state (optional, may be NULL) additionally receives the predicted next state as a contiguous
 * [A,S] tensor, so a closed-loop caller can feed it straight back as the next d_state. */
int bbmpc_optimize_dev(bbmpc_handle h, const float* d_state, int32_t time_step, int32_t add_exploration_noise,
                       float* d_record, float* d_next_state);

/* DeterministicTrajectoryEvaluator.__call__(current_states[A,S], action_sequences[n_pop,A,H,U], t)
 * -> rewards[n_pop,A]                 trajectory_evaluators/deterministic.py:26-77 */
int bbmpc_evaluate(bbmpc_handle h, const float* state, const float* action_sequences, int32_t n_pop,
                   float* rewards);
int bbmpc_evaluate_dev(bbmpc_handle h, const float* d_state, const float* d_action_sequences, int32_t n_pop,
                       float* d_rewards);

/* .predict_next_state(states[B,S], actions[B,U]) -> [B,S]      deterministic.py:79-103 */
int bbmpc_predict_next_state(bbmpc_handle h, const float* states, const float* actions, int32_t batch,
                             float* next_states);
/* .evaluate_next_reward(cur[B,S], next[B,S], actions[B,U]) -> [B]   deterministic.py:105-127 */
int bbmpc_evaluate_next_reward(bbmpc_handle h, const float* states, const float* next_states,
                               const float* actions, int32_t batch, float* rewards);
/* device-pointer env step used by the closed-loop harness: action rows are read with
 * `action_stride` floats between consecutive rows (so a record buffer can be passed). */
int bbmpc_step_dev(bbmpc_handle h, const float* d_states, const float* d_actions, int32_t action_stride,
                   int32_t batch, float* d_next_states, float* d_rewards);

/* Closed-loop episode on the device -- counterpart of utils/rollouts.py:60-139 (_sample) with the engine's own
 * model as the environment: T control steps, each feeding its predicted next state back as the next observation;
 * nothing leaves HBM until the end.  records_out is [T][A][U+S+1] (action | next_state | reward) on the host.
 * Equivalent to T calls of bbmpc_optimize with state_{t+1} = next_state_t (no exploration noise). */
int bbmpc_rollout_episode(bbmpc_handle h, const float* start_state, int32_t num_steps, int32_t add_exploration_noise,
                          float* records_out);

/* ---- parity / test hooks ------------------------------------------------ */
/* Replace the engine's Philox draws of `kind` by caller-supplied standard noise
 * (reference layout, see BBMPC_NOISE_*).  count = number of floats.  data == NULL
 * clears the injection for that kind.  Injected tensors are consumed by every
 * subsequent control step until cleared. */
int bbmpc_inject_noise(bbmpc_handle h, int32_t kind, const float* data, int64_t count);
/* Write the standard noise the engine's own Philox scheme produces for (kind, control_step, iteration),
 * in the reference layout -- lets tests check the generator against its documented definition. */
int bbmpc_dump_noise(bbmpc_handle h, int32_t kind, int32_t control_step, int32_t iteration, float* out,
                     int64_t count);
/* Enable per-iteration trace capture (costs extra device copies; off by default). */
int bbmpc_set_trace(bbmpc_handle h, int32_t enabled);
int bbmpc_get_trace(bbmpc_handle h, int32_t iteration, int32_t item, void* out, int64_t bytes);
/* Read / write optimizer state tensors by name ("mean","var","pos","vel","pbest","pbest_r",
 * "gbest","gbest_r","m","sigma","C","B","D","p_sigma","p_C"), reference layout.  Settable: "prev_mean", "var0",
 * and with CMA-ES "C" (G*n*n). */
int bbmpc_get_state(bbmpc_handle h, const char* name, float* out, int64_t count);
int bbmpc_set_state(bbmpc_handle h, const char* name, const float* data, int64_t count);

/* ---- measurement -------------------------------------------------------- */
/* When enabled the handle brackets launches of its dominant (rollout) kernel with HIP events on the launch
 * stream: enabled = 1 every launch, enabled = n > 1 every n-th launch (an event pair costs a few microseconds
 * of stream time, which matters when the whole control step is ~50 us); bbmpc_get_profile returns the
 * accumulated device time and the number of bracketed launches since the last call and resets them. */
int bbmpc_set_profiling(bbmpc_handle h, int32_t enabled);
int bbmpc_get_profile(bbmpc_handle h, double* rollout_ms_total, int64_t* rollout_launches,
                      const char** kernel_name);
/* The dominant kernel of the last control step with its template arguments, spelt as rocprofv3 prints kernel names minus the
 * leading "void " (e.g. "k_fused_pendulum<2, true, true, 2, 1, false>": optimizer, samples in LDS, carried-angle model,
 * noise source, trajectories per lane, resident) when the engine holds several instantiations of it, the plain name
 * otherwise.  bench.py attaches the committed PMC counters (profiles/) to a measurement by THIS name, so that a counter of
 * the BBMPC_STRICT_MATH or the resident instantiation can never be divided by the default instantiation's time.
 * The string lives in the handle and changes with the next control step. */
int bbmpc_profile_instantiation(bbmpc_handle h, const char** kernel_instantiation);
/* Block until everything enqueued on the handle's stream (and its communication stream) has finished. */
int bbmpc_synchronize(bbmpc_handle h);

/* ---- multi-GPU: the agent-sharded path's one exchange ------------------------
 * One process per GPU, each handle owning agents [agent_offset, agent_offset + num_agents) of num_agents_global
 * (bbmpc_config).  Nothing on the optimisation path crosses GPUs; per control step the packed records
 * [num_agents, dim_u + dim_s + 1] are all-gathered (RCCL over xGMI) so every rank holds every agent's action /
 * predicted next state / predicted reward.  The reference has no counterpart: its agents are a batch dimension of a
 * single process (optimizers/optimizer_base.py:59-94 returns all agents' results at once); the gathered array is
 * exactly that return value, reassembled.
 *
 *   rank 0: bbmpc_comm_unique_id(id)  -> ship the BBMPC_COMM_ID_BYTES to every rank (any side channel)
 *   all   : bbmpc_comm_init(h, id, nranks, rank)              (collective; librccl is bound at run time)
 *   step t: bbmpc_gather_wait(h, t & 1, 0);                    the gather that last used this slot's buffers is done
 *           bbmpc_optimize_dev(h, ..., d_records[t & 1], ...);
 *           bbmpc_gather_records_dev(h, d_records[t & 1], d_gathered[t & 1], num_agents * record_width, t & 1);
 * The collective runs on the handle's own stream and overlaps the next control step; rows of d_gathered are in
 * global agent order when every rank has the same num_agents.  bbmpc_gather_wait with host_block != 0 blocks the
 * calling thread until the slot's gathered array can be read. */
#define BBMPC_COMM_ID_BYTES 128
int bbmpc_comm_unique_id(void* out, int64_t bytes);
int bbmpc_comm_init(bbmpc_handle h, const void* unique_id, int32_t nranks, int32_t rank);
/* The same communicator for `nranks` handles of ONE process on ONE device, without RCCL (which refuses two ranks on a GPU):
 * handles that pass the same `group_key` form the group, rank = 0 .. nranks-1 (at most 16).  Its all-gather is copies and events on the callers'
 * streams around a HOST rendezvous of the ranks (csrc/comm.hpp): every rank must be driven by its own host thread, as ranks
 * are processes elsewhere, and a rank that does not show up within 60 s fails the others' call.  Everything else -- record gathers, the
 * per-iteration exchanges of a sharded population, bbmpc_comm_info / _destroy -- works as with bbmpc_comm_init.  It exists so
 * that the rank > 0 / nranks > 1 paths run on a one-GPU box (tests/test_gpu_local_ranks.py), and for several handles sharing a
 * device.  At most 1 MiB
 * per rank and operation.  No counterpart in the reference. */
int bbmpc_comm_init_local(bbmpc_handle h, uint64_t group_key, int32_t nranks, int32_t rank);
int bbmpc_gather_records_dev(bbmpc_handle h, const float* d_records, float* d_gathered, int64_t count_per_rank,
                             int32_t slot);
/* bbmpc_optimize_dev + bbmpc_gather_records_dev in one call (count = num_agents * record width): lets a control
 * step that is a single kernel launch carry the "records ready" event on its own dispatch packet instead of a
 * separate event record on the launch stream. */
int bbmpc_optimize_gather_dev(bbmpc_handle h, const float* d_state, int32_t time_step, int32_t add_exploration_noise,
                              float* d_records, float* d_next_state, float* d_gathered, int32_t slot);
/* MPCPolicy.act (policies/mpc_policy.py:124-172) of ONE RANK of an agent-sharded run: bbmpc_optimize for the local
 * agents (host buffers, synchronous) + the all-gather of their records into d_gathered [num_agents_global, U+S+1]
 * (device memory), enqueued on the communication stream and overlapped with the caller's next control step. */
int bbmpc_optimize_gather(bbmpc_handle h, const float* state, int32_t time_step, int32_t add_exploration_noise,
                          float* action, float* next_state, float* reward, float* d_gathered, int32_t slot);
/* ncclCommCount / ncclCommUserRank of the handle's communicator and the hand-off mode in use
 * (1 = sequence numbers in signal memory, 0 = events); any out pointer may be NULL. */
int bbmpc_comm_info(bbmpc_handle h, int32_t* nranks, int32_t* rank, int32_t* sync_mode);
int bbmpc_gather_wait(bbmpc_handle h, int32_t slot, int32_t host_block);
int bbmpc_comm_destroy(bbmpc_handle h);

#ifdef __cplusplus
}
#endif
#endif /* BBMPC_H */
