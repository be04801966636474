// Sample / refit kernels of the SPSA and PSO optimizers (dynamics-agnostic: the rollouts in between
// go through k_rollout_pendulum / k_rollout_mlp with SRC_BUF candidates).
//   SPSAOptimizer  optimizers/spsa.py:61-117        PSOOptimizer  optimizers/pso.py:70-160
// Buffers use the engine's internal layout [A][H*U][Nst] (particle-minor, coalesced).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "kernels_refit.hpp"
#include "models.hpp"
#include "rng.hpp"

namespace bbmpc {

struct OptArgs {
    int N, A, H, U, HU, Nst;
    int agent_offset;
    const float* lo;
    const float* hi;
    RngKey key;
    uint32_t iter;
    int pop_offset;       // global index of local particle 0 (population sharding): draws are keyed by the GLOBAL particle
};

__device__ __forceinline__ uint32_t elem_word(const RngKey& key, uint32_t stream, uint32_t iter, int n, int ga, int j) {
    const U4 b = rng_block(key, stream, iter, (uint32_t)n, (uint32_t)ga, (uint32_t)j);
    return pick_word(b, (uint32_t)j);
}

// ------------------------------------------------------------------------------------------------
// SPSA
// ------------------------------------------------------------------------------------------------
// delta in {-1,+1}; theta+- = solution +- c_k*delta   (spsa.py:73-77).  Clipping + penalties happen in the
// rollout kernel (SRC_BUF, pen).  grid (ceil(N/256), HU, A)
static __global__ void k_spsa_candidates(OptArgs p, const float* solution /*[A][HU]*/, float ck, const float* inj /*[A][HU][Nst]|null*/,
                                  float* delta, float* cand_plus, float* cand_minus) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, a = blockIdx.z;
    if (n >= p.N) return;
    const size_t idx = ((size_t)a * p.HU + j) * p.Nst + n;
    const float d = inj ? inj[idx] : word_to_rademacher(elem_word(p.key, 3u, p.iter, n + p.pop_offset, p.agent_offset + a, j));
    const float sol = solution[a * p.HU + j];
    const float step = ck * d;
    delta[idx] = d;
    cand_plus[idx] = sol + step;
    cand_minus[idx] = sol - step;
}

// ghat[j] = mean_n (r+ - r-)[n] / (2 c_k delta[j][n]);  solution = clip(solution + a_k ghat)   (spsa.py:101-107)
// grid (Gr, A): workgroup x takes the rows j = x * waves + wave, x * waves + wave + Gr * waves, ... (one wave per row; a single
// workgroup per agent walked all H*U rows: 180 k divisions and 720 KB of delta through one CU at the north-star shape).
// `part` != null (population sharded over ranks): this rank's particles only -- the row sums go to part[a][j] and the
// update is k_spsa_merge's.
static __global__ __launch_bounds__(REFIT_THREADS) void k_refit_spsa(OptArgs p, const float* rew_plus, const float* rew_minus,
                                                              const float* delta, float ak, float ck, float* solution,
                                                              float* action, float* part) {
    extern __shared__ float diff[];
    const int a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = REFIT_THREADS / 64;
    for (int n = tid; n < p.N; n += REFIT_THREADS) diff[n] = rew_plus[(size_t)a * p.Nst + n] - rew_minus[(size_t)a * p.Nst + n];
    __syncthreads();
    const float two_ck = 2.0f * ck;
    for (int j = blockIdx.x * NW + wv; j < p.HU; j += NW * gridDim.x) {
        const float* drow = delta + ((size_t)a * p.HU + j) * p.Nst;
        float acc = 0.0f;
        for (int n = lane; n < p.N; n += 64) acc += diff[n] / (two_ck * drow[n]);
        acc = wave_sum(acc);
        if (part) {
            if (lane == 0) part[a * p.HU + j] = acc;
            continue;
        }
        if (lane == 0) {
            const float ghat = acc / (float)p.N;
            const int u = j % p.U;
            const float s = clipf(solution[a * p.HU + j] + ak * ghat, p.lo[u], p.hi[u]);
            solution[a * p.HU + j] = s;
            if (j < p.U) action[a * p.U + j] = s;
        }
    }
}

// SPSA with the population sharded over ranks: spsa.py:101-107 is a mean over the perturbation pairs, so
// every rank sums its own pairs (k_refit_spsa with `part`), one all-gather hands every rank all G row-sum vectors
// all[r][a][j], and each rank adds them in rank order, divides by the GLOBAL population and takes the step: identical bits
// on every rank; against the unsharded refit only the order of the fp32 sums differs.  grid (ceil(HU/256), A)
static __global__ void k_spsa_merge(OptArgs p, const float* all, int G, int n_global, float ak, float* solution, float* action) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
    if (j >= p.HU) return;
    float acc = 0.0f;
    for (int r = 0; r < G; ++r) acc = acc + all[((size_t)r * p.A + a) * p.HU + j];
    const float ghat = acc / (float)n_global;
    const int u = j % p.U;
    const float s = clipf(solution[a * p.HU + j] + ak * ghat, p.lo[u], p.hi[u]);
    solution[a * p.HU + j] = s;
    if (j < p.U) action[a * p.U + j] = s;
}

// ------------------------------------------------------------------------------------------------
// PSO
// ------------------------------------------------------------------------------------------------
struct PsoState {
    float* pos;       // [A][HU][Nst]
    float* vel;
    float* pbest;
    float* pbest_r;   // [A][Nst]
    float* gbest;     // [A][HU]
    float* gbest_r;   // [A]
    float* cond;      // [A][Nst] 1.0 where pbest_r < reward this iteration
    int* gidx;        // [A]
};

// per agent: personal-best rewards, global best index (first maximum), global best position (pso.py:84-100)
// `part` != null (population sharded over ranks): this rank's particles only -- the local best goes to
// part[a] = (value, GLOBAL particle index as bits, position[HU]) and k_pso_merge picks the swarm's.
static __global__ __launch_bounds__(REFIT_THREADS) void k_pso_best(OptArgs p, PsoState s, const float* rewards, float* part) {
    __shared__ float sv[REFIT_THREADS / 64];
    __shared__ int si[REFIT_THREADS / 64];
    __shared__ int s_gi;
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = REFIT_THREADS / 64;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int n = tid; n < p.N; n += REFIT_THREADS) {
        const size_t i = (size_t)a * p.Nst + n;
        const float r = rewards[i], pr = s.pbest_r[i];
        const bool c = pr < r;                                   // tf.less(pbest_r, rewards)
        const float npr = c ? r : pr;
        s.cond[i] = c ? 1.0f : 0.0f;
        s.pbest_r[i] = npr;
        if (npr > bv || (npr == bv && n < bi)) { bv = npr; bi = n; }
    }
    if (bi == 0x7fffffff && tid < p.N) bi = tid;                 // all -inf / NaN: argmax is still an index
    wave_argmax(bv, bi);
    if (lane == 0) { sv[wv] = bv; si[wv] = bi; }
    __syncthreads();
    if (wv == 0) {
        bv = (lane < NW) ? sv[lane] : -INFINITY;
        bi = (lane < NW) ? si[lane] : 0x7fffffff;
        wave_argmax(bv, bi);
        if (bi == 0x7fffffff) bi = 0;
        if (lane == 0) {
            s_gi = bi;
            const float br = s.pbest_r[(size_t)a * p.Nst + bi];
            if (part) {
                part[(size_t)a * (p.HU + 2)] = br;
                part[(size_t)a * (p.HU + 2) + 1] = __int_as_float(bi + p.pop_offset);
            } else {
                s.gidx[a] = bi;
                s.gbest_r[a] = br;
            }
        }
    }
    __syncthreads();
    const int gi = s_gi;
    const bool cg = s.cond[(size_t)a * p.Nst + gi] != 0.0f;
    float* dst = part ? part + (size_t)a * (p.HU + 2) + 2 : s.gbest + a * p.HU;
    for (int j = tid; j < p.HU; j += REFIT_THREADS) {
        const size_t i = ((size_t)a * p.HU + j) * p.Nst + gi;
        dst[j] = cg ? s.pos[i] : s.pbest[i];                     // pbest after this iteration's update
    }
}

// PSO with the swarm sharded over ranks: all[r][a] = (best value, global index, position[HU]) of rank r's particles; the
// swarm's best is the first maximum by GLOBAL index (tf.argmax over the whole population: pso.py:94), so the sharded run
// is the unsharded one bit for bit (the only cross-particle operation of PSO is this argmax).  grid A
static __global__ void k_pso_merge(OptArgs p, PsoState s, const float* all, int G) {
    const int a = blockIdx.x, tid = threadIdx.x;
    const size_t pw = (size_t)p.A * (p.HU + 2);
    int br = 0, bi = 0x7fffffff;
    float bv = -INFINITY;
    for (int r = 0; r < G; ++r) {                                  // every thread the same scan: G is a handful
        const float* e = all + pw * r + (size_t)a * (p.HU + 2);
        const float v = e[0];
        const int idx = __float_as_int(e[1]);
        if (v > bv || (v == bv && idx < bi) || bi == 0x7fffffff) { bv = v; bi = idx; br = r; }
    }
    const float* src = all + pw * br + (size_t)a * (p.HU + 2) + 2;
    for (int j = tid; j < p.HU; j += blockDim.x) s.gbest[a * p.HU + j] = src[j];
    if (tid == 0) { s.gidx[a] = bi; s.gbest_r[a] = bv; }
}

// r1, r2: the two SCALAR N(0,1) draws shared by every particle / dim / agent (quirk Q3, pso.py:107-109)
__device__ __forceinline__ void pso_scalars(const OptArgs& p, const float* inj2, float& r1, float& r2) {
    if (inj2) { r1 = inj2[0]; r2 = inj2[1]; return; }
    const U4 b = rng_block(p.key, 5u, p.iter, 0u, 0u, 0u);     // agent-independent counter: identical on every shard
    words_to_normal2(b.x, b.y, r1, r2);
}

// velocity / position update (pso.py:86-88, 104-108).  grid (ceil(N/256), HU, A)
static __global__ void k_pso_move(OptArgs p, PsoState s, float w, float c1, float c2, const float* inj2) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, a = blockIdx.z;
    if (n >= p.N) return;
    float r1, r2;
    pso_scalars(p, inj2, r1, r2);
    const size_t i = ((size_t)a * p.HU + j) * p.Nst + n;
    const float pos = s.pos[i];
    const float pb = (s.cond[(size_t)a * p.Nst + n] != 0.0f) ? pos : s.pbest[i];
    s.pbest[i] = pb;
    const float gb = s.gbest[a * p.HU + j];
    const float t1 = s.vel[i] * w;
    const float t2 = ((pb - pos) * c1) * r1;
    const float t3 = ((gb - pos) * c2) * r2;
    const float v = (t1 + t2) + t3;
    s.vel[i] = v;
    s.pos[i] = pos + v;
}

// swarm re-seed after the loop (pso.py:116-138) or reset() (:143-160).
//   reseed: pos = shift_left(gbest) + sqrt(constrained var) * xi_trunc ; vel = U(-v0, v0)
//   reset : pos = U(lo, hi)                                           ; vel = U(-v0, v0)
// grid (ceil(N/256), HU, A)
static __global__ void k_pso_seed(OptArgs p, PsoState s, const float* var0 /*[A][HU]*/, float v0frac, int is_reset,
                           const float* inj_pos, const float* inj_vel) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, a = blockIdx.z;
    if (n >= p.N) return;
    const int u = j % p.U, h = j / p.U;
    const float lo = p.lo[u], hi = p.hi[u];
    const size_t i = ((size_t)a * p.HU + j) * p.Nst + n;
    const int ga = p.agent_offset + a;
    float pos;
    if (is_reset) {
        const float x = inj_pos ? inj_pos[i] : word_to_uniform(elem_word(p.key, 8u, p.iter, n + p.pop_offset, ga, j));
        pos = x * (hi - lo) + lo;
    } else {
        const float g = s.gbest[a * p.HU + j];
        const float lb = (g - lo) / 2.0f, ub = (hi - g) / 2.0f;
        const float cv = fminf(fminf(lb * lb, ub * ub), var0[a * p.HU + j]);
        const int hs = (h + 1 < p.H) ? h + 1 : p.H - 1;
        const float mean = s.gbest[a * p.HU + hs * p.U + u];
        const float xi = inj_pos ? inj_pos[i] : word_to_trunc_normal(elem_word(p.key, 6u, p.iter, n + p.pop_offset, ga, j));
        pos = xi * sqrtf(cv) + mean;
    }
    const float v0 = v0frac * (hi - lo);
    const float uv = inj_vel ? inj_vel[i] : word_to_uniform(elem_word(p.key, is_reset ? 9u : 7u, p.iter, n + p.pop_offset, ga, j));
    s.pos[i] = pos;
    s.vel[i] = uv * (v0 - (-v0)) + (-v0);
    s.pbest[i] = pos;
    if (j == 0) s.pbest_r[(size_t)a * p.Nst + n] = -INFINITY;
    if (j == 0 && n == 0) s.gbest_r[a] = -INFINITY;
}

// action = gbest[:, 0, :]
static __global__ void k_take_first(int A, int HU, int U, const float* src, float* action) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * U) return;
    action[i] = src[(i / U) * HU + (i % U)];
}

}  // namespace bbmpc
