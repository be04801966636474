// Lane-per-trajectory rollout kernels for the analytic (true-model) dynamics:
// the fused counterpart of
//   optimizer sample block            (cem.py:90-94, pi2.py:65-75, random_search.py:40-41, ...)
//   DeterministicTrajectoryEvaluator.__call__   (trajectory_evaluators/deterministic.py:26-77)
//   SystemDynamicsHandler.process_input/_output (dynamics_handlers/system_dynamics_handler.py:97-161)
//   PendulumTrueModel + pendulum_reward_function (utils/pendulum.py)
// One lane owns one candidate trajectory: state lives in VGPRs for the whole
// H-step recurrence, the action element is drawn (Philox) or fetched at the
// point of use, the reward is accumulated in-register, and only the reward and
// (when the refit needs it) the action sequence touch memory.
//
// Internal layouts (particle-minor so that wave accesses are coalesced):
//   samples  [A][H*U][Nst]     rewards [A][Nst]      mean/sigma [A][H*U]
#pragma once
#include "models.hpp"
#include "rng.hpp"

namespace bbmpc {

// where a candidate action element comes from
constexpr int SRC_REF = 0;      // caller's action_sequences, reference layout [n_pop, A, H, U]
constexpr int SRC_UNIFORM = 1;  // lo + u*(hi-lo)                 random_search.py:40-41
constexpr int SRC_TRUNC = 2;    // mean + sigma*xi, xi trunc-normal  cem.py:90-94 / pi2.py:65-69
constexpr int SRC_BUF = 3;      // candidate already in an internal-layout buffer (PSO positions, CMA-ES samples)

struct RolloutArgs {
    int n_pop;            // particles per agent in this launch
    int A, H, U, S, HU;
    int Nst;              // particle stride of internal buffers
    int agent_offset;     // global id of local agent 0
    int fix_q1;
    int reward_kind;
    const float* state;   // [A,S]
    const float* seq;     // SRC_REF
    const float* inj;     // injected standard noise for this iteration, internal layout, or null
    const float* mean;    // [A][HU]
    const float* sigma;   // [A][HU]
    const float* lo;      // [U]
    const float* hi;      // [U]
    const float* cand;    // SRC_BUF: candidate buffer (internal layout)
    float* samples;       // where the rolled-out (feasible) sequence is stored, or null
    float* rewards;       // [A][Nst]
    float* penalty_out;   // optional [A][Nst]
    RngKey key;
    uint32_t stream;
    uint32_t iter;
    int pop_offset;       // global index of local particle 0 (population sharding): the RNG is keyed by the GLOBAL particle
};

// Candidate element j of particle n, agent a.  `blk` caches the Philox block across calls.
// `ms` (LDS) holds this agent's mean[HU] | sigma[HU] for SRC_TRUNC so that the sequential recurrence
// never waits on a global load (the sample stores would otherwise order behind/ahead of them).
template <int MODE>
__device__ __forceinline__ float candidate(const RolloutArgs& p, int n, int a, int j, int u, U4& blk, const float* ms) {
    const int aj = a * p.HU + j;
    if constexpr (MODE == SRC_BUF) {
        return p.cand[(size_t)aj * p.Nst + n];
    } else {
        float xi;
        if (p.inj) {
            xi = p.inj[(size_t)aj * p.Nst + n];
        } else {
            if ((j & 3) == 0 || j == 0) blk = rng_block(p.key, p.stream, p.iter, (uint32_t)(n + p.pop_offset), (uint32_t)(p.agent_offset + a), (uint32_t)j);
            const uint32_t w = pick_word(blk, (uint32_t)j);
            xi = (MODE == SRC_UNIFORM) ? word_to_uniform(w) : word_to_trunc_normal(w);
        }
        if constexpr (MODE == SRC_UNIFORM) {
            const float l = p.lo[u];
            return xi * (p.hi[u] - l) + l;                      // tf.random.uniform: rnd*(max-min)+min
        } else {
            return xi * ms[p.HU + j] + ms[j];                   // tf.random.truncated_normal: rnd*stddev+mean
        }
    }
}

// blockDim.x threads = consecutive particles of agent blockIdx.y.
template <int MODE, bool PEN, bool FASTM>
__global__ void k_rollout_pendulum(RolloutArgs p) {
    constexpr int U = PendulumModel::U;
    const int a = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = n < p.n_pop;
    Roller<FASTM> roll(p.fix_q1 != 0, p.state[a * 3 + 0], p.state[a * 3 + 1], p.state[a * 3 + 2]);
    float total = 0.0f, pen = 0.0f;
    U4 blk = {0, 0, 0, 0};
    extern __shared__ float tile[];

    if constexpr (MODE == SRC_REF) {
        // Stage the block's action rows through LDS: global reads run along each particle's
        // contiguous [H*U] row (coalesced), the recurrence then reads its own row from LDS
        // (row pitch TJ+1 words -> conflict-free).
        constexpr int TJ = 32;
        const int rows = blockDim.x;
        const int n0 = blockIdx.x * blockDim.x;
        for (int j0 = 0; j0 < p.HU; j0 += TJ) {
            const int tj = min(TJ, p.HU - j0);
            __syncthreads();
            for (int idx = threadIdx.x; idx < rows * TJ; idx += blockDim.x) {
                const int r = idx / TJ, c = idx % TJ;
                float v = 0.0f;
                if (c < tj && n0 + r < p.n_pop)
                    v = p.seq[((size_t)(n0 + r) * p.A + a) * p.HU + j0 + c];
                tile[r * (TJ + 1) + c] = v;
            }
            __syncthreads();
            if (active) {
                for (int c = 0; c < tj; c += U) {
                    roll.step_acc(tile[threadIdx.x * (TJ + 1) + c]);
                }
            }
        }
    } else {
        if constexpr (MODE == SRC_TRUNC) {
            for (int j = threadIdx.x; j < p.HU; j += blockDim.x) {
                tile[j] = p.mean[a * p.HU + j];
                tile[p.HU + j] = p.sigma[a * p.HU + j];
            }
            __syncthreads();
        }
        const float lo0 = p.lo[0], hi0 = p.hi[0];
        if (active) {
            if constexpr (MODE == SRC_BUF && U == 1) {
                // candidates come from a buffer (PSO / SPSA / CMA-ES; may alias p.samples, which is written below):
                // eight loads in flight per lane instead of one L2 latency per step inside the recurrence
                for (int t0 = 0; t0 < p.H; t0 += 8) {
                    float xb[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        xb[i] = (t0 + i < p.H) ? p.cand[((size_t)a * p.HU + t0 + i) * p.Nst + n] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (t0 + i < p.H) {
                            float x = xb[i];
                            if constexpr (PEN) {
                                const float xf = clipf(x, lo0, hi0);
                                const float d = x - xf;
                                pen = pen + d * d;
                                x = xf;
                            }
                            if (p.samples) p.samples[(size_t)(a * p.HU + t0 + i) * p.Nst + n] = x;
                            roll.step_acc(x);
                        }
                    }
                }
            } else
            for (int t = 0; t < p.H; ++t) {
                float act[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = t * U + u;
                    float x = candidate<MODE>(p, n, a, j, u, blk, tile);
                    if constexpr (PEN) {
                        const float xf = clipf(x, lo0, hi0);
                        const float d = x - xf;
                        pen = pen + d * d;
                        x = xf;
                    }
                    if (p.samples) p.samples[(size_t)(a * p.HU + j) * p.Nst + n] = x;
                    act[u] = x;
                }
                roll.step_acc(act[0]);
            }
        }
    }
    if (!active) return;
    total = roll.total();                                       // (the H-step sum is the roller's: models.hpp)
    if (total != total) total = -1.0e6f;                        // deterministic.py:75-77
    if constexpr (PEN) {
        const float nr = sqrtf(pen);                            // tf.norm(...)**2  pi2.py:72-75
        pen = nr * nr;
        total = total - pen;
        if (p.penalty_out) p.penalty_out[(size_t)a * p.Nst + n] = pen;
    }
    p.rewards[(size_t)a * p.Nst + n] = total;
}

// Single environment/model step on [B] rows: predict_next_state + evaluate_next_reward
// (deterministic.py:79-127) for the analytic pendulum.  actions rows are `astride` floats apart.
static __global__ void k_step_pendulum(const float* states, const float* actions, int astride, int batch, int fix_q1,
                                float* next_states, float* rewards) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const PendulumModel model{fix_q1 != 0};
    float s[3] = {states[b * 3 + 0], states[b * 3 + 1], states[b * 3 + 2]};
    float act[1] = {actions[(size_t)b * astride]};
    const float r = model.step(s, act);
    if (next_states) {
        next_states[b * 3 + 0] = s[0];
        next_states[b * 3 + 1] = s[1];
        next_states[b * 3 + 2] = s[2];
    }
    if (rewards) rewards[b] = r;
}

// evaluate_next_reward on caller-provided (cur, next, actions) rows, any reward kind.
static __global__ void k_reward_only(const float* cur, const float* nxt, const float* act, int batch, int S, int U,
                              int reward_kind, int fix_q1, float* rewards) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    rewards[b] = reward_generic(reward_kind, fix_q1 != 0, cur + (size_t)b * S, act + (size_t)b * U,
                                nxt + (size_t)b * S, S, U);
}

}  // namespace bbmpc
