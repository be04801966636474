// Small-batch kernels around the per-iteration rollouts of the learned-dynamics path:
//
//   k_refit_pi2_mw   PI2 refit (pi2.py:78-87) spread over many workgroups: the soft-min weighted mean is a
//                    [HU x N] matrix-vector product, one wave per row of the particle-minor sample matrix.
//   k_tail_mlp       the tail of OptimizerBase.__call__ (optimizer_base.py:82-94) for a learned model in ONE launch:
//                    exploration noise, one model step on the agent's row, reward, packed record, and the warm start
//                    of the next control step (pi2.py:92-93 / spsa.py:114-115 shift-left).
//   k_rows_mlp       predict_next_state / evaluate_next_reward for a handful of rows (deterministic.py:79-127) with the
//                    same per-row code as the tail, so that `policy.act`'s predicted next state and a later
//                    `evaluator.predict_next_state(obs, action)` agree bit for bit.
//
// A single row through a 26-200-200-20 network is 49 k multiply-adds behind 198 KB of weights: nothing for the matrix
// cores to chew on (an MFMA tile would be 1/16 full) and latency bound either way, so these use plain fp32 FMAs:
// one workgroup per row, every Dense layer split over (K-slices x outputs) threads with coalesced reads of the
// reference-layout kernel W[in][out], partial sums combined in a fixed order through LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_mlp.hpp"
#include "kernels_opt.hpp"
#include "kernels_refit.hpp"

namespace bbmpc {

constexpr int TAIL_THREADS = 1024;
constexpr int TAIL_MAXW = 512;                 // widest layer (set_mlp enforces hidden <= 512, S+U <= 128, S <= 64)

#ifdef BBMPC_KERNEL_DBG
static __device__ long long g_tail_dbg[16];     // (static: the header is seen by four translation units)
#endif
struct RowMlp {
    MlpDesc m;
    const float* wraw[MLP_MAX_LAYERS];         // Dense kernels [in][out] (reference layout)
    const float* braw[MLP_MAX_LAYERS];         // biases [out]
};

// One Dense stack on the row held in `x` (LDS, dims[0] values, already normalised).  Returns a pointer (LDS) to the
// raw network output dims[L].  bufA/bufB: TAIL_MAXW floats each; part: [splits][TAIL_MAXW].
__device__ __forceinline__ const float* row_mlp_forward(const RowMlp& q, const float* x, float* bufA, float* bufB,
                                                        float* part, int tid, int nthr) {
    const float* in = x;
    for (int l = 0; l < q.m.n_layers; ++l) {
        const int K = q.m.dims[l], M = q.m.dims[l + 1];
        const float* __restrict__ W = q.wraw[l];
        const int Mr = (M + 63) & ~63;                           // outputs per K-slice, wave aligned
        const int G = max(1, min(nthr / Mr, (K + 7) / 8));       // K-slices (>= 8 terms each)
        const int Kc = (K + G - 1) / G;
        for (int t = tid; t < G * Mr; t += nthr) {
            const int g = t / Mr, o = t - g * Mr;
            if (o < M) {
                const int k0 = g * Kc, k1 = min(K, k0 + Kc);
                // a K-slice is one L2 round trip per batch of loads: 16 weights in flight per lane (a 50-term slice is four
                // trips instead of thirteen), four accumulators
                float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
                int k = k0;
                for (; k + 15 < k1; k += 16) {
                    float w[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) w[i] = W[(size_t)(k + i) * M + o];
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        acc0 = fmaf(in[k + i], w[i], acc0);
                        acc1 = fmaf(in[k + i + 1], w[i + 1], acc1);
                        acc2 = fmaf(in[k + i + 2], w[i + 2], acc2);
                        acc3 = fmaf(in[k + i + 3], w[i + 3], acc3);
                    }
                }
                {
                    float w[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) w[i] = (k + i < k1) ? W[(size_t)(k + i) * M + o] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        acc0 = fmaf((k + i < k1) ? in[k + i] : 0.0f, w[i], acc0);
                        acc1 = fmaf((k + i + 1 < k1) ? in[k + i + 1] : 0.0f, w[i + 1], acc1);
                        acc2 = fmaf((k + i + 2 < k1) ? in[k + i + 2] : 0.0f, w[i + 2], acc2);
                        acc3 = fmaf((k + i + 3 < k1) ? in[k + i + 3] : 0.0f, w[i + 3], acc3);
                    }
                }
                part[g * TAIL_MAXW + o] = (acc0 + acc1) + (acc2 + acc3);
            }
        }
        __syncthreads();
#ifdef BBMPC_KERNEL_DBG
        if (tid == 0 && blockIdx.x == 0) g_tail_dbg[1 + 2 * l] = (long long)wall_clock64();
#endif
        float* out = (l & 1) ? bufB : bufA;
        for (int o = tid; o < M; o += nthr) {
            float acc = q.braw[l][o];
            for (int g = 0; g < G; ++g) acc = acc + part[g * TAIL_MAXW + o];
            out[o] = apply_act(acc, q.m.act[l]);
        }
        __syncthreads();
#ifdef BBMPC_KERNEL_DBG
        if (tid == 0 && blockIdx.x == 0) g_tail_dbg[2 + 2 * l] = (long long)wall_clock64();
#endif
        in = out;
    }
    return in;
}

// process_input -> Dense stack -> process_output -> reward for the row (cur[S] | act[U]) staged in LDS.
// nxt (LDS, S floats) receives the next state; returns the reward (valid in every thread).
__device__ __forceinline__ float row_model_step(const RowMlp& q, int S, int U, int reward_kind, bool fix_q1, const float* cur,
                                                const float* act, float* nxt, float* x, float* bufA, float* bufB, float* part,
                                                int tid, int nthr) {
    const bool normd = q.m.normalized != 0;
    for (int f = tid; f < S + U; f += nthr) {                    // system_dynamics_handler.py:97-126
        const float v = f < S ? cur[f] : act[f - S];
        const float mu = normd ? (f < S ? q.m.mean_s[f] : q.m.mean_a[f - S]) : 0.0f;
        const float sd = normd ? (f < S ? q.m.std_s[f] : q.m.std_a[f - S]) : 1.0f;
        const float inv = normd ? 1.0f / (sd + 1e-7f) : 1.0f;    // same form as the rollout kernels' prologue
        x[f] = (v - mu) * inv;
    }
    __syncthreads();
    const float* raw = row_mlp_forward(q, x, bufA, bufB, part, tid, nthr);
    for (int f = tid; f < S; f += nthr) {                        // :128-161 + transforms.py:20-34
        const float dev = normd ? q.m.mean_t[f] + raw[f] * (q.m.std_t[f] + 1e-7f) : raw[f];
        nxt[f] = dev + cur[f];
    }
    __syncthreads();
    return reward_generic(reward_kind, fix_q1, cur, act, nxt, S, U);
}

struct TailArgs {
    FinalArgs f;            // A, U, S, exploration noise, state [A,S], action [A,U] (in: the optimizer's solution), record, next_state
    RowMlp net;
    int reward_kind;
    // warm start of the next control step: 0 none, 1 prev = shift_left(mean) (pi2.py:92-93, spsa.py:114-115),
    // 2 prev = mean (CEM with BBMPC_FIX_Q2_CEM_WARM_START)
    int warm_mode, H, HU;
    const float* mean;      // [A][HU]
    float* prev_mean;       // [A][HU]
    unsigned* done_flag;    // publish_records_done (kernels_refit.hpp) or null
    unsigned* done_count;
    unsigned done_value;
    // a control step replayed as a graph (engine.hpp: step_graph): [0] the control step its draws are keyed by (rng.hpp:
    // RngKey::step_src), [1] the completion value -- device memory, read at the start of this kernel and advanced by its
    // last workgroup for the next replay
    unsigned* step_words;
};

// grid A, block TAIL_THREADS
static __global__ __launch_bounds__(TAIL_THREADS) void k_tail_mlp(TailArgs p) {
    __shared__ float cur[64], act[128], nxt[64], x[192];
    __shared__ float bufA[TAIL_MAXW], bufB[TAIL_MAXW], part[(TAIL_THREADS / 64) * TAIL_MAXW];
    const int a = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int S = p.f.S, U = p.f.U;
    const unsigned done_value = p.step_words ? p.step_words[1] : p.done_value;
#ifdef BBMPC_KERNEL_DBG
    if (tid == 0 && a == 0) g_tail_dbg[0] = (long long)wall_clock64();
#endif
    if (tid < S) cur[tid] = p.f.state[a * S + tid];
    if (tid >= 64 && tid < 64 + U) {
        const int u = tid - 64;
        act[u] = exploration_action(p.f, a, u, p.f.action[a * U + u]);       // optimizer_base.py:82-90
    }
    // warm start (independent of the model step; before it so the loads overlap)
    if (p.warm_mode) {
        for (int j = tid; j < p.HU; j += nthr) {
            int src = j;
            if (p.warm_mode == 1) {
                const int u = j % U, h = j / U;
                src = ((h + 1 < p.H) ? h + 1 : p.H - 1) * U + u;
            }
            p.prev_mean[a * p.HU + j] = p.mean[a * p.HU + src];
        }
    }
    __syncthreads();
    const float r = row_model_step(p.net, S, U, p.reward_kind, p.f.fix_q1 != 0, cur, act, nxt, x, bufA, bufB, part, tid, nthr);
    const int rec = U + S + 1;
    float* out = p.f.record + (size_t)a * rec;
    if (tid < U) out[tid] = act[tid];
    if (tid >= 64 && tid < 64 + S) {
        const float v = nxt[tid - 64];
        out[U + tid - 64] = v;
        if (p.f.next_state) p.f.next_state[a * S + tid - 64] = v;
    }
    if (tid == 128) out[U + S] = r;
#ifdef BBMPC_KERNEL_DBG
    if (tid == 0 && a == 0) {
        const long long t1 = (long long)wall_clock64();
        printf("[tail] stage + L0 mac %lld (= %lld) red %lld | L1 mac %lld red %lld | L2 mac %lld red %lld | out %lld (10 ns)\n", g_tail_dbg[1] - g_tail_dbg[0],
               g_tail_dbg[1] - g_tail_dbg[0], g_tail_dbg[2] - g_tail_dbg[1], g_tail_dbg[3] - g_tail_dbg[2], g_tail_dbg[4] - g_tail_dbg[3], g_tail_dbg[5] - g_tail_dbg[4], g_tail_dbg[6] - g_tail_dbg[5], t1 - g_tail_dbg[6]);
    }
#endif
    if (p.done_flag) {
        __syncthreads();
        if (tid == 0) {
            publish_records_done(p.done_flag, p.done_count, done_value, gridDim.x);
            if (p.step_words) {
                // every workgroup has read the words by the time the last one gets here (they are read first thing)
                const unsigned old = gridDim.x == 1u ? 0u : __hip_atomic_fetch_add(p.step_words + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old == gridDim.x - 1u) {
                    if (gridDim.x > 1u) __hip_atomic_store(p.step_words + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    p.step_words[0] = p.step_words[0] + 1u;
                    const unsigned nv = done_value + 1u;
                    p.step_words[1] = nv == 0u ? 1u : nv;            // (as bbmpc_optimize counts: 0 is never a completion value)
                }
            }
        }
    }
}

// grid = rows, block TAIL_THREADS.  next_states and/or rewards may be null.
static __global__ __launch_bounds__(TAIL_THREADS) void k_rows_mlp(RowMlp net, int S, int U, int reward_kind, int fix_q1,
                                                           const float* states, const float* actions, int action_stride,
                                                           float* next_states, float* rewards) {
    __shared__ float cur[64], act[128], nxt[64], x[192];
    __shared__ float bufA[TAIL_MAXW], bufB[TAIL_MAXW], part[(TAIL_THREADS / 64) * TAIL_MAXW];
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    if (tid < S) cur[tid] = states[(size_t)b * S + tid];
    if (tid >= 64 && tid < 64 + U) act[tid - 64] = actions[(size_t)b * action_stride + tid - 64];
    __syncthreads();
    const float r = row_model_step(net, S, U, reward_kind, fix_q1 != 0, cur, act, nxt, x, bufA, bufB, part, tid, nthr);
    if (next_states && tid < S) next_states[(size_t)b * S + tid] = nxt[tid];
    if (rewards && tid == 64) rewards[b] = r;
}

// ---- PI2 refit over many workgroups ---------------------------------------------------------------------------
// new_mean[j] = sum_n omega[n] * samples[j][n], omega = softmin(cost / lamda)   (pi2.py:78-87).
// Every workgroup recomputes beta = min cost, eta = sum exp(-(cost - beta)/lamda) and omega for the agent's whole
// population (N <= 32768 values: a few loads per thread and two block reductions -- cheaper than a launch that would
// produce them once), then each of its waves takes one row j of the particle-minor sample matrix with all of a
// lane's loads in flight.  One 1024-thread workgroup walking all HU rows (the previous kernel) needed 58 us at
// N = 1000, HU = 180: 11 rows per wave, each a 16-deep chain of dependent L2 round trips.
// grid (ceil(HU / PI2_ROWS), A), block 64 * PI2_ROWS.   LDS: omega[Nst] | red[PI2_ROWS]
constexpr int PI2_ROWS = 4;
static __global__ __launch_bounds__(64 * PI2_ROWS) void k_refit_pi2_mw(RefitArgs p) {
    extern __shared__ float smem[];
    float* om = smem;
    float* red = smem + p.Nst;
    const int a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = PI2_ROWS, NT = 64 * PI2_ROWS;
    // this wave's row of the sample matrix does not depend on the weights: for populations up to 64 PF its loads go out
    // with the reward loads (one memory round trip for the kernel instead of two)
    constexpr int PF = 16;
    const int j = blockIdx.x * PI2_ROWS + wv;
    const float* __restrict__ row = p.samples + (size_t)(a * p.HU + (j < p.HU ? j : 0)) * p.Nst;
    const bool pf = p.N <= 64 * PF;
    float xr[PF];
    if (pf) {
#pragma unroll
        for (int m = 0; m < PF; ++m) xr[m] = (lane + 64 * m < p.N) ? row[lane + 64 * m] : 0.0f;
    }
    float lmin = INFINITY;
    for (int n = tid; n < p.N; n += NT) {
        const float c = -p.rewards[(size_t)a * p.Nst + n];               // costs = -rewards   pi2.py:78-79
        om[n] = c;
        lmin = fminf(lmin, c);
    }
    lmin = wave_min(lmin);
    if (lane == 0) red[wv] = lmin;
    __syncthreads();
    float beta = red[lane < NW ? lane : 0];
    beta = wave_min(beta);                                               // pi2.py:81
    __syncthreads();
    float lsum = 0.0f;
    for (int n = tid; n < p.N; n += NT) {
        const float pr = expf((-p.inv_lamda) * (om[n] - beta));          // pi2.py:82
        om[n] = pr;
        lsum += pr;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[wv] = lsum;
    __syncthreads();
    float eta = (lane < NW) ? red[lane] : 0.0f;
    eta = wave_sum(eta);                                                 // pi2.py:83
    const float inv_eta = 1.0f / eta;
    __syncthreads();
    for (int n = tid; n < p.N; n += NT) om[n] = inv_eta * om[n];         // pi2.py:85
    __syncthreads();
    if (j >= p.HU) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (pf) {
        // the same sums in the same order as the loop below, on the prefetched registers
        bool tail = false;
#pragma unroll
        for (int i = 0; i < PF / 4; ++i) {
            const int n = lane + 256 * i;
            if (!tail && n + 192 < p.N) {
                acc[0] = fmaf(xr[4 * i + 0], om[n], acc[0]);
                acc[1] = fmaf(xr[4 * i + 1], om[n + 64], acc[1]);
                acc[2] = fmaf(xr[4 * i + 2], om[n + 128], acc[2]);
                acc[3] = fmaf(xr[4 * i + 3], om[n + 192], acc[3]);
            } else {
                tail = true;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (n + 64 * c < p.N) acc[0] = fmaf(xr[4 * i + c], om[n + 64 * c], acc[0]);
            }
        }
    } else {
        int n = lane;
        for (; n + 192 < p.N; n += 256) {                                    // four independent loads per trip
            const float x0 = row[n], x1 = row[n + 64], x2 = row[n + 128], x3 = row[n + 192];
            acc[0] = fmaf(x0, om[n], acc[0]);
            acc[1] = fmaf(x1, om[n + 64], acc[1]);
            acc[2] = fmaf(x2, om[n + 128], acc[2]);
            acc[3] = fmaf(x3, om[n + 192], acc[3]);
        }
        for (; n < p.N; n += 64) acc[0] = fmaf(row[n], om[n], acc[0]);
    }
    const float s = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) {
        p.mean[a * p.HU + j] = s;                                        // pi2.py:86-87
        if (j < p.U) p.action[a * p.U + j] = s;                          // new_mean[:, 0]
    }
}

// ---- PI2 refit with the population sharded over ranks --------------------------------------------
// Each rank holds N_local particles of the SAME agents.  pi2.py:80-87 split:
//   partial (this rank) : beta_r = min cost, p_n = exp(-(cost_n - beta_r)/lamda), eta_r = sum p_n, S_r[j] = sum p_n x[j][n]
//   exchange            : all ranks see every rank's (beta_r, eta_r, S_r[HU]) per agent  (one collective per iteration)
//   merge (every rank)  : beta = min_r beta_r, s_r = exp(-(beta_r - beta)/lamda), eta = sum_r s_r eta_r,
//                         mean[j] = (sum_r s_r S_r[j]) / eta        -- in rank order, so all ranks get the same bits
// The only difference to the un-sharded refit is the order of the fp32 sums (tolerance stated in the tests: 2e-5).
// part layout per agent: [0] beta_r, [1] eta_r, [2 .. 2+HU) S_r.
// k_refit_pi2_partial: grid (ceil(HU / PI2_ROWS), A), block 64 * PI2_ROWS, LDS pr[Nst] | red[PI2_ROWS]
static __global__ __launch_bounds__(64 * PI2_ROWS) void k_refit_pi2_partial(RefitArgs p, float* part) {
    extern __shared__ float smem[];
    float* pr = smem;
    float* red = smem + p.Nst;
    const int a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = PI2_ROWS, NT = 64 * PI2_ROWS;
    float* out = part + (size_t)a * (p.HU + 2);
    float lmin = INFINITY;
    for (int n = tid; n < p.N; n += NT) {
        const float c = -p.rewards[(size_t)a * p.Nst + n];
        pr[n] = c;
        lmin = fminf(lmin, c);
    }
    lmin = wave_min(lmin);
    if (lane == 0) red[wv] = lmin;
    __syncthreads();
    float beta = red[lane < NW ? lane : 0];
    beta = wave_min(beta);
    __syncthreads();
    float lsum = 0.0f;
    for (int n = tid; n < p.N; n += NT) {
        const float w = expf((-p.inv_lamda) * (pr[n] - beta));
        pr[n] = w;
        lsum += w;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[wv] = lsum;
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        float eta = 0.0f;
        for (int w = 0; w < NW; ++w) eta += red[w];
        out[0] = beta;
        out[1] = eta;
    }
    const int j = blockIdx.x * PI2_ROWS + wv;
    if (j >= p.HU) return;
    const float* __restrict__ row = p.samples + (size_t)(a * p.HU + j) * p.Nst;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int n = lane;
    for (; n + 192 < p.N; n += 256) {
        const float x0 = row[n], x1 = row[n + 64], x2 = row[n + 128], x3 = row[n + 192];
        acc[0] = fmaf(x0, pr[n], acc[0]);
        acc[1] = fmaf(x1, pr[n + 64], acc[1]);
        acc[2] = fmaf(x2, pr[n + 128], acc[2]);
        acc[3] = fmaf(x3, pr[n + 192], acc[3]);
    }
    for (; n < p.N; n += 64) acc[0] = fmaf(row[n], pr[n], acc[0]);
    const float s = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if (lane == 0) out[2 + j] = s;
}

// gathered: [G][A][HU+2].  grid (ceil(HU/256), A), block 256
static __global__ void k_refit_pi2_merge(RefitArgs p, const float* gathered, int G) {
    const int a = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p.HU) return;
    const size_t stride = (size_t)p.A * (p.HU + 2);
    float beta = INFINITY;
    for (int r = 0; r < G; ++r) beta = fminf(beta, gathered[r * stride + (size_t)a * (p.HU + 2)]);
    float eta = 0.0f, acc = 0.0f;
    for (int r = 0; r < G; ++r) {
        const float* g = gathered + r * stride + (size_t)a * (p.HU + 2);
        const float sc = expf((-p.inv_lamda) * (g[0] - beta));
        eta = fmaf(sc, g[1], eta);
        acc = fmaf(sc, g[2 + j], acc);
    }
    const float m = acc * (1.0f / eta);
    p.mean[a * p.HU + j] = m;
    if (j < p.U) p.action[a * p.U + j] = m;
}

// ---- CEM refit with the population sharded over ranks --------------------------------------------
// cem.py:97-125 split: every rank ships its LOCAL top-k per agent -- (reward, GLOBAL particle index, the H*U sample row)
// -- in one all-gather; every rank then takes the global top-k of the G*k candidates (larger reward first, ties ->
// lower global index: tf.nn.top_k's rule on the unsharded population) and refits mean / variance from the gathered
// rows.  The global elite SET is exactly the unsharded one (a global winner is always a local winner of its rank);
// only the order of the fp32 sums differs.  cand layout per agent: [k][HU + 2] = (reward | global index as float bits |
// row); slots a short shard cannot fill carry reward -inf.
// Both kernels run on (Gw, A) workgroups: every workgroup of an agent repeats the (cheap) selection / ranking and takes
// its share of the H*U columns -- the scattered gather of k x H*U sample values, and the strided reads of the merge's
// statistics, are what these kernels spend their time on (one workgroup per agent: +34 us per iteration at N = 1000).
// k_cem_local_topk: grid (Gw, A), block 1024.  LDS: rewards[Nst] | eidx[kpad] | hist | ekeys[2*kpad]
static __global__ __launch_bounds__(1024) void k_cem_local_topk(RefitArgs p, int pop_offset, float* cand) {
    extern __shared__ float smem[];
    const int a = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    float* r = smem;
    int* eidx = (int*)(smem + p.Nst);
    const int kpad = (p.k + 3) & ~3;
    uint32_t* hist = (uint32_t*)(eidx + kpad);
    unsigned long long* ekeys = (unsigned long long*)(hist + TOPK_HIST_WORDS);
    for (int n = tid; n < p.N; n += nthr) r[n] = p.rewards[(size_t)a * p.Nst + n];
    __syncthreads();
    const int kk = min(p.k, p.N);
    block_topk_sorted(r, p.N, kk, eidx, hist, ekeys, tid, nthr);
    const int W = p.HU + 2;
    const int cw = (W + (int)gridDim.x - 1) / (int)gridDim.x, c_lo = blockIdx.x * cw, c_n = max(0, min(W, c_lo + cw) - c_lo);
    float* out = cand + (size_t)a * p.k * W;
    for (int i = tid; i < p.k * c_n; i += nthr) {
        const int e = i / c_n, c = c_lo + (i - e * c_n);
        float v;
        if (e >= kk) v = c == 0 ? -INFINITY : 0.0f;
        else if (c == 0) v = r[eidx[e]];
        else if (c == 1) v = __int_as_float(pop_offset + eidx[e]);
        else v = p.samples[(size_t)(a * p.HU + (c - 2)) * p.Nst + eidx[e]];
        out[(size_t)e * W + c] = v;
    }
}

// gathered: [G][A][k][HU+2].  grid (Gw, A), block 256.  LDS: key rewards[G*k] | global idx[G*k] | chosen slot[k]
// Statistics of a row on a 16-lane group (elites over the lanes, DPP reduction), as in k_refit_cem_v2.
static __global__ __launch_bounds__(256) void k_cem_merge(RefitArgs p, const float* gathered, int G) {
    extern __shared__ float smem[];
    const int a = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    const int W = p.HU + 2, M = G * p.k;
    float* cr = smem;
    int* ci = (int*)(smem + M);
    int* chosen = ci + M;                                    // chosen[rank] = candidate slot, rank < k
    const size_t rstride = (size_t)p.A * p.k * W;
    auto slot_ptr = [&](int m) { return gathered + (size_t)(m / p.k) * rstride + ((size_t)a * p.k + (m % p.k)) * W; };
    for (int m = tid; m < M; m += nthr) {
        const float* s = slot_ptr(m);
        cr[m] = s[0];
        ci[m] = __float_as_int(s[1]);
    }
    __syncthreads();
    for (int m = tid; m < M; m += nthr) {                    // rank by counting: better = larger reward, then lower global index
        const float rm = cr[m];
        const int im = ci[m];
        int rank = 0;
        for (int o = 0; o < M; ++o) {
            const float ro = cr[o];
            rank += (ro > rm || (ro == rm && (ci[o] < im || (ci[o] == im && o < m)))) ? 1 : 0;
        }
        if (rank < p.k) chosen[rank] = m;
    }
    __syncthreads();
    const float kf = (float)p.k, one_m = 1.0f - p.alpha;
    if (p.elites && blockIdx.x == 0)
        for (int e = tid; e < p.k; e += nthr) p.elites[a * p.k + e] = ci[chosen[e]];        // GLOBAL indices, best first
    const int rows_wg = (p.HU + (int)gridDim.x - 1) / (int)gridDim.x;
    const int jlo = blockIdx.x * rows_wg, jhi = min(p.HU, jlo + rows_wg);
    const int sub = tid & 15, grp = tid >> 4, ngrp = nthr >> 4;
    for (int j0 = jlo; j0 < jhi; j0 += ngrp) {
        const int j = j0 + grp;
        const bool live = j < jhi;
        float sum = 0.0f;
        for (int e = sub; e < p.k; e += 16) sum = sum + (live ? slot_ptr(chosen[e])[2 + j] : 0.0f);
        sum = row16_sum(sum);
        const float em = sum / kf;                                           // cem.py:112
        float vs = 0.0f;
        for (int e = sub; e < p.k; e += 16) {
            const float d = (live ? slot_ptr(chosen[e])[2 + j] : em) - em;
            vs = vs + d * d;
        }
        vs = row16_sum(vs);
        if (live && sub == 0) {
            const float ev = vs / kf;                                        // :113-119 (biased)
            const int aj = a * p.HU + j;
            const float m = p.alpha * p.mean[aj] + one_m * em;               // :121-122
            const float v = p.alpha * p.var[aj] + one_m * ev;                // :123-125
            p.mean[aj] = m;
            p.var[aj] = v;
            const int u = j % p.U;
            p.sigma[aj] = cem_sigma(m, v, p.lo[u], p.hi[u]);
            if (j < p.U) p.action[a * p.U + j] = m;                          // mean[:, 0]  cem.py:135
        }
    }
}

}  // namespace bbmpc
