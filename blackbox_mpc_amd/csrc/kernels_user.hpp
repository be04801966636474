// Step-wise trajectory evaluator: the un-fused form of DeterministicTrajectoryEvaluator.__call__
// (trajectory_evaluators/deterministic.py:26-77) used whenever the reward or the dynamics is a user-supplied device
// function (rtc.hpp).  It is the reference's own structure -- one batched dynamics call and one batched reward call
// per planning step over B = n_pop * A rows -- instead of the fused whole-horizon kernels the built-in pairs get:
//
//   k_rows_prepare      candidates (internal layout [A][H*U][Nst], or the caller's [n,A,H,U]) -> action rows [H][B][U],
//                       optional clip + squared bound violation (the PI2 / PSO / SPSA / CMA-ES penalty), start state
//                       tiled to [B,S] (deterministic.py:52-57)
//   per step t          dynamics rows  (user function | k_step_pendulum | k_step_mlp)     :64, :79-103
//                       reward rows    (user function | k_reward_rows_acc), accumulated   :65-67
//   k_rows_finish       NaN -> -1e6 (:75-77), minus the penalty, into rewards [A][Nst]
//
// Row b = a * n_pop + n.  Sampling for RandomSearch / CEM / PI2 is a separate kernel here (k_gen_candidates), the
// refits are the ordinary ones.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_rollout.hpp"
#include "models.hpp"

namespace bbmpc {

// candidates of one optimizer iteration into the sample buffer (internal layout): what the fused rollout kernels draw
// on the fly.  grid (ceil(N/256), A), block 256; LDS 2*HU floats for SRC_TRUNC.
template <int MODE>
__global__ void k_gen_candidates(RolloutArgs p) {
    extern __shared__ float ms[];
    const int a = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (MODE == SRC_TRUNC) {
        for (int j = threadIdx.x; j < p.HU; j += blockDim.x) {
            ms[j] = p.mean[a * p.HU + j];
            ms[p.HU + j] = p.sigma[a * p.HU + j];
        }
        __syncthreads();
    }
    if (n >= p.n_pop) return;
    U4 blk = {0, 0, 0, 0};
    for (int j = 0; j < p.HU; ++j) p.samples[(size_t)(a * p.HU + j) * p.Nst + n] = candidate<MODE>(p, n, a, j, j % p.U, blk, ms);
}

struct RowsArgs {
    int n_pop, A, H, U, S, HU, Nst;
    int from_ref;            // 1: p_seq is the caller's [n_pop, A, H, U]; 0: p_cand is the internal layout
    int pen;                 // clip + penalty
    const float* seq;
    const float* cand;
    float* samples;          // where the feasible sequence is stored (internal layout) or null
    const float* lo;
    const float* hi;
    const float* state;      // [A,S]
    float* rows;             // [H][B][U]
    float* x0;               // [B,S]
    float* penalty;          // [B]
};

// grid (ceil(n_pop/256), A), block 256
static __global__ void k_rows_prepare(RowsArgs p) {
    const int a = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= p.n_pop) return;
    const size_t B = (size_t)p.A * p.n_pop, b = (size_t)a * p.n_pop + n;
    float pen = 0.0f;
    for (int j = 0; j < p.HU; ++j) {
        const int t = j / p.U, u = j - t * p.U;
        float x = p.from_ref ? p.seq[((size_t)n * p.A + a) * p.HU + j] : p.cand[(size_t)(a * p.HU + j) * p.Nst + n];
        if (p.pen) {
            const float xf = clipf(x, p.lo[u], p.hi[u]);
            const float d = x - xf;
            pen = pen + d * d;
            x = xf;
        }
        if (p.samples) p.samples[(size_t)(a * p.HU + j) * p.Nst + n] = x;
        p.rows[((size_t)t * B + b) * p.U + u] = x;
    }
    for (int s = 0; s < p.S; ++s) p.x0[b * p.S + s] = p.state[a * p.S + s];        // tf.tile(current_states, [nopt, 1])
    if (p.penalty) {
        const float nr = sqrtf(pen);                                               // tf.norm(...)**2  pi2.py:72-75
        p.penalty[b] = nr * nr;
    }
}

// built-in reward on rows, accumulated over the planning steps
static __global__ void k_reward_rows_acc(const float* cur, const float* nxt, const float* act, int astride, int batch, int S, int U,
                                  int reward_kind, int fix_q1, float* total, int accumulate) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const float r = reward_generic(reward_kind, fix_q1 != 0, cur + (size_t)b * S, act + (size_t)b * astride, nxt + (size_t)b * S, S, U);
    total[b] = accumulate ? total[b] + r : r;
}

// grid (ceil(n_pop/256), A)
static __global__ void k_rows_finish(int n_pop, int A, int Nst, int pen, const float* total, const float* penalty, float* rewards,
                              float* penalty_out) {
    const int a = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_pop) return;
    const size_t b = (size_t)a * n_pop + n;
    float t = total[b];
    if (t != t) t = -1.0e6f;                                                       // deterministic.py:75-77
    if (pen) {
        t = t - penalty[b];
        if (penalty_out) penalty_out[(size_t)a * Nst + n] = penalty[b];
    }
    rewards[(size_t)a * Nst + n] = t;
}

// SystemDynamicsHandler.process_input / process_output as stand-alone calls
// (dynamics_handlers/system_dynamics_handler.py:97-126, 128-161 + utils/transforms.py:20-34).
// stats = mean_states | std_states | mean_actions | std_actions | mean_targets | std_targets (contiguous) or null.
static __global__ void k_process_input(const float* states, const float* actions, int batch, int S, int U, const float* stats, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * (S + U)) return;
    const int b = i / (S + U), f = i % (S + U);
    float v = f < S ? states[(size_t)b * S + f] : actions[(size_t)b * U + (f - S)];
    if (stats) {
        const float mu = f < S ? stats[f] : stats[2 * S + (f - S)];
        const float sd = f < S ? stats[S + f] : stats[2 * S + U + (f - S)];
        v = (v - mu) / (sd + 1e-7f);                                        // :119-122
    }
    out[i] = v;
}
static __global__ void k_process_output(const float* states, const float* raw, int batch, int S, int U, const float* stats, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * S) return;
    const int f = i % S;
    float dev = raw[i];
    if (stats) dev = stats[2 * S + 2 * U + f] + dev * (stats[3 * S + 2 * U + f] + 1e-7f);   // :152-155
    out[i] = dev + states[i];                                                // transforms.py:34
}

}  // namespace bbmpc
