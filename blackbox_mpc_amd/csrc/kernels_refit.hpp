// Distribution-refit kernels: the "refit block" of each optimizer iteration
// (SURVEY.md 3.2).  One workgroup per agent; rewards of the agent's population
// sit in LDS, cross-lane reductions use wave64 shuffles.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "models.hpp"
#include "rng.hpp"
#include "topk.hpp"

namespace bbmpc {

constexpr int REFIT_THREADS = 1024;

// ---- cross-lane primitives on DPP row operations (no LDS round trip; a ds_bpermute-based __shfl costs ~100+
// cycles per step, a DPP-modified v_add ~8) -------------------------------------------------------------------
#define BB_DPP_QUAD_XOR1 0xB1      /* quad_perm [1,0,3,2] */
#define BB_DPP_QUAD_XOR2 0x4E      /* quad_perm [2,3,0,1] */
#define BB_DPP_ROW_HALF_MIRROR 0x141
#define BB_DPP_ROW_MIRROR 0x140
#define BB_DPP_ROW_SHR(n) (0x110 + (n))

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// every lane of a 16-lane row ends up with the row's sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<BB_DPP_QUAD_XOR1>(v);
    v += dpp_f<BB_DPP_QUAD_XOR2>(v);
    v += dpp_f<BB_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f<BB_DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, dpp_f<BB_DPP_QUAD_XOR1>(v));
    v = fminf(v, dpp_f<BB_DPP_QUAD_XOR2>(v));
    v = fminf(v, dpp_f<BB_DPP_ROW_HALF_MIRROR>(v));
    v = fminf(v, dpp_f<BB_DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}
__device__ __forceinline__ float wave_min(float v) {
    v = row16_min(v);
    const int i = __float_as_int(v);
    return fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                 fminf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}
// (value,index) arg-max with tf.math.argmax tie rule: first (lowest index) maximum wins.
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

struct RefitArgs {
    int N, A, H, U, HU, Nst, k;
    float alpha;           // CEM
    float inv_lamda;       // PI2: 1/lamda
    const float* rewards;  // [A][Nst]
    const float* samples;  // [A][HU][Nst]
    const float* lo;       // [U]
    const float* hi;       // [U]
    float* mean;           // [A][HU]  in/out
    float* var;            // [A][HU]  in/out (CEM)
    float* sigma;          // [A][HU]  out: sqrt of the constrained variance the NEXT iteration samples with
    int* elites;           // [A][k] out (sorted, best first)   | RandomSearch/PSO: [A] best index
    float* action;         // [A][U] out
    const float* mean_in;  // CEM (k_refit_cem_v2): the distribution the smoothing starts from when it is not mean / var themselves
    const float* var_in;   // (first iteration of a control step that skipped k_dist_init: prev_mean / var0), or null
};

// sigma = sqrt(min(((mean-lo)/2)^2, ((hi-mean)/2)^2, var))        cem.py:79-88
__device__ __forceinline__ float cem_sigma(float mean, float var, float lo, float hi) {
    const float lb = (mean - lo) / 2.0f;
    const float ub = (hi - mean) / 2.0f;
    return sqrtf(fminf(fminf(lb * lb, ub * ub), var));
}

// (Re)initialise the distribution at the start of a control step.
// CEM quirk Q2: every control step restarts from the constructor mean/variance (cem.py:129-134).
// state_src (optional): the [A,S] state of a host-in / host-out call, in pinned host memory -- the first kernel of the
// control step brings it to HBM (state_dst) itself instead of a copy-engine transfer in front of it
static __global__ void k_dist_init(int A, int HU, int U, const float* lo, const float* hi, const float* prev_mean,
                            const float* var0, float* mean, float* var, float* sigma, int constrain,
                            const float* state_src, float* state_dst, int nstate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (state_src)
        for (int k = i; k < nstate; k += gridDim.x * blockDim.x) state_dst[k] = state_src[k];
    if (i >= A * HU) return;
    const int u = i % U;
    const float m = prev_mean[i];
    const float v = var0[i];
    mean[i] = m;
    if (var) var[i] = v;
    sigma[i] = constrain ? cem_sigma(m, v, lo[u], hi[u]) : sqrtf(v);
}

// CEM refit  cem.py:97-125: top-k (sorted, ties -> lower index), elite mean / biased variance
// (accumulated sequentially in elite order), alpha smoothing.
// LDS: rewards[Nst] | elite idx[kpad] | hist[272] | ekeys[2*kpad] | elite tile [k][JC]
static __global__ __launch_bounds__(REFIT_THREADS) void k_refit_cem(RefitArgs p, int JC) {
    extern __shared__ float smem[];
    const int a = blockIdx.x;
    const int tid = threadIdx.x;
    float* r = smem;
    int* eidx = (int*)(smem + p.Nst);
    const int kpad = (p.k + 3) & ~3;
    uint32_t* hist = (uint32_t*)(eidx + kpad);
    unsigned long long* ekeys = (unsigned long long*)(hist + TOPK_HIST_WORDS);
    float* tile = (float*)(ekeys + kpad);

    for (int n = tid; n < p.N; n += REFIT_THREADS) r[n] = p.rewards[(size_t)a * p.Nst + n];
    __syncthreads();
    block_topk_sorted(r, p.N, p.k, eidx, hist, ekeys, tid, REFIT_THREADS);
    if (p.elites)
        for (int e = tid; e < p.k; e += REFIT_THREADS) p.elites[a * p.k + e] = eidx[e];

    const float kf = (float)p.k;
    for (int j0 = 0; j0 < p.HU; j0 += JC) {
        const int jc = min(JC, p.HU - j0);
        __syncthreads();
        for (int idx = tid; idx < p.k * jc; idx += REFIT_THREADS) {
            const int jj = idx / p.k, e = idx % p.k;
            tile[e * JC + jj] = p.samples[(size_t)(a * p.HU + j0 + jj) * p.Nst + eidx[e]];
        }
        __syncthreads();
        if (tid < jc) {
            const int j = j0 + tid;
            float sum = 0.0f;
            for (int e = 0; e < p.k; ++e) sum = sum + tile[e * JC + tid];
            const float em = sum / kf;                                  // reduce_mean = sum / k
            float vs = 0.0f;
            for (int e = 0; e < p.k; ++e) {
                const float d = tile[e * JC + tid] - em;
                vs = vs + d * d;
            }
            const float ev = vs / kf;
            const int aj = a * p.HU + j;
            const float one_m = 1.0f - p.alpha;
            const float m = p.alpha * p.mean[aj] + one_m * em;             // cem.py:121-122
            const float v = p.alpha * p.var[aj] + one_m * ev;              // cem.py:123-125
            p.mean[aj] = m;
            p.var[aj] = v;
            const int u = j % p.U;
            p.sigma[aj] = cem_sigma(m, v, p.lo[u], p.hi[u]);
            if (j < p.U) p.action[a * p.U + j] = m;                        // mean[:, 0]  cem.py:135
        }
    }
}

// CEM refit for k <= 64 elites, latency-lean version of the kernel above (same results as the persistent kernel's
// refit: elite SET from the radix select, listed in ascending index order; statistics over 16-lane rows).
// blockDim up to 1024: the gathers are 4 independent global loads per lane with up to four rows in flight, the
// elite statistics a 16-lane DPP reduction -- no LDS tile, no serial 50-element loops.
// grid (G, A): the elite gather reads 4 bytes out of every cache line it touches (the elites are scattered along the
// particle-minor axis), k x HU lines through one CU's L1 -- 1.2 MB at config 4, 1.9 MB at config 5 and most of the
// kernel's time.  So G workgroups per agent each repeat the (cheap, reward-only) selection and take HU / G rows.
// LDS: rewards[Nst] | elite idx[kpad] | hist[TOPK_HIST_WORDS] | ekeys[2*kpad]
static __global__ __launch_bounds__(1024) void k_refit_cem_v2(RefitArgs p) {
    extern __shared__ float smem[];
    const int a = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    const int rows_wg = (p.HU + (int)gridDim.x - 1) / (int)gridDim.x;
    const int jlo = blockIdx.x * rows_wg, jhi = min(p.HU, jlo + rows_wg);
    float* r = smem;
    int* eidx = (int*)(smem + p.Nst);
    const int kpad = (p.k + 3) & ~3;
    uint32_t* hist = (uint32_t*)(eidx + kpad);
    unsigned long long* ekeys = (unsigned long long*)(hist + TOPK_HIST_WORDS);

    for (int n = tid; n < p.N; n += nthr) r[n] = p.rewards[(size_t)a * p.Nst + n];
    __syncthreads();
    const TopkSel sel = block_topk_select(r, p.N, p.k, hist, tid, nthr);
    if (p.elites && blockIdx.x == 0) {                // parity trace wants tf.nn.top_k's sorted order
        block_topk_finish_sorted(r, p.N, p.k, eidx, hist, ekeys, sel, tid, nthr);
        for (int e = tid; e < p.k; e += nthr) p.elites[a * p.k + e] = eidx[e];
        __syncthreads();
    }
    block_topk_finish_indexed(r, p.N, p.k, eidx, hist, sel, tid, nthr);

    constexpr int EC = 4, RC = 4;                     // elites per lane, rows in flight per 16-lane group
    const int sub = tid & 15, grp = tid >> 4, ngrp = nthr >> 4;
    int ei[EC];
#pragma unroll
    for (int i = 0; i < EC; ++i) ei[i] = (sub + 16 * i < p.k) ? eidx[sub + 16 * i] : -1;
    const float kf = (float)p.k, one_m = 1.0f - p.alpha;
    const float* __restrict__ samples = p.samples + (size_t)a * p.HU * p.Nst;
    for (int j0 = jlo; j0 < jhi; j0 += RC * ngrp) {
        float x[RC][EC];
#pragma unroll
        for (int rr = 0; rr < RC; ++rr) {
            const int j = j0 + rr * ngrp + grp;
#pragma unroll
            for (int i = 0; i < EC; ++i) x[rr][i] = 0.0f;
            if (j < jhi) {                                                 // 16-lane groups without a row issue no loads
                const float* row = samples + (size_t)j * p.Nst;
#pragma unroll
                for (int i = 0; i < EC; ++i) x[rr][i] = row[ei[i] >= 0 ? ei[i] : 0];
            }
        }
#pragma unroll
        for (int rr = 0; rr < RC; ++rr) {
            const int j = j0 + rr * ngrp + grp;
            float sum = 0.0f, vs = 0.0f;
#pragma unroll
            for (int i = 0; i < EC; ++i) if (ei[i] >= 0) sum += x[rr][i];
            sum = row16_sum(sum);
            const float em = sum / kf;                                     // cem.py:112
#pragma unroll
            for (int i = 0; i < EC; ++i)
                if (ei[i] >= 0) {
                    const float d = x[rr][i] - em;
                    vs += d * d;
                }
            vs = row16_sum(vs);
            if (j < jhi && sub == 0) {
                const float ev = vs / kf;                                  // cem.py:113-119
                const int aj = a * p.HU + j;
                const float m = p.alpha * (p.mean_in ? p.mean_in : p.mean)[aj] + one_m * em;         // cem.py:121-122
                const float v = p.alpha * (p.var_in ? p.var_in : p.var)[aj] + one_m * ev;            // cem.py:123-125
                p.mean[aj] = m;
                p.var[aj] = v;
                const int u = j % p.U;
                p.sigma[aj] = cem_sigma(m, v, p.lo[u], p.hi[u]);
                if (j < p.U) p.action[a * p.U + j] = m;                    // mean[:, 0]  cem.py:135
            }
        }
    }
}

// PI2 refit  pi2.py:78-87: softmin weights over the population, weighted mean of the (feasible) samples.
// LDS: omega[Nst] | scratch[32]
static __global__ __launch_bounds__(REFIT_THREADS) void k_refit_pi2(RefitArgs p) {
    extern __shared__ float smem[];
    float* om = smem;
    float* red = smem + p.Nst;
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = REFIT_THREADS / 64;

    float lmin = INFINITY;
    for (int n = tid; n < p.N; n += REFIT_THREADS) {
        const float c = -p.rewards[(size_t)a * p.Nst + n];               // costs = -rewards
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
    for (int n = tid; n < p.N; n += REFIT_THREADS) {
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
    for (int n = tid; n < p.N; n += REFIT_THREADS) om[n] = inv_eta * om[n];   // pi2.py:85
    __syncthreads();
    // new_mean[j] = sum_n samples[j][n] * omega[n]: one wave per j, lanes stride the population (coalesced)
    for (int j = wv; j < p.HU; j += NW) {
        const float* row = p.samples + (size_t)(a * p.HU + j) * p.Nst;
        float acc = 0.0f;
        for (int n = lane; n < p.N; n += 64) acc += row[n] * om[n];
        acc = wave_sum(acc);
        if (lane == 0) {
            p.mean[a * p.HU + j] = acc;                                  // pi2.py:86-87
            if (j < p.U) p.action[a * p.U + j] = acc;                    // new_mean[:, 0]
        }
    }
}

// warm start: prev = [mean[:,1:], mean[:,-1:]]   pi2.py:92-93 / spsa.py:114-115
static __global__ void k_shift_left(int A, int H, int U, const float* mean, float* prev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * H * U) return;
    const int u = i % U, h = (i / U) % H, a = i / (U * H);
    const int hs = (h + 1 < H) ? h + 1 : H - 1;
    prev[i] = mean[(a * H + hs) * U + u];
}

// RandomSearch refit  random_search.py:43-47: per-agent argmax (first maximum), take its first action.
// `part` != null (population sharded over ranks): part[a] = (best value, GLOBAL particle index as bits,
// its first action[U]) of this rank's particles; k_argmax_merge takes the first maximum by global index.
static __global__ __launch_bounds__(REFIT_THREADS) void k_refit_argmax(RefitArgs p, float* part, int pop_offset) {
    __shared__ float sv[REFIT_THREADS / 64];
    __shared__ int si[REFIT_THREADS / 64];
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = REFIT_THREADS / 64;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int n = tid; n < p.N; n += REFIT_THREADS) {
        const float v = p.rewards[(size_t)a * p.Nst + n];
        if (v > bv || (v == bv && n < bi)) { bv = v; bi = n; }
    }
    // a population of all -inf / never-greater values still needs a valid index: argmax returns 0 then
    if (bi == 0x7fffffff && tid < p.N) bi = tid;
    wave_argmax(bv, bi);
    if (lane == 0) { sv[wv] = bv; si[wv] = bi; }
    __syncthreads();
    if (wv == 0) {
        bv = (lane < NW) ? sv[lane] : -INFINITY;
        bi = (lane < NW) ? si[lane] : 0x7fffffff;
        wave_argmax(bv, bi);
        if (bi == 0x7fffffff) bi = 0;
        if (part) {
            float* e = part + (size_t)a * (p.U + 2);
            if (lane == 0) { e[0] = p.rewards[(size_t)a * p.Nst + bi]; e[1] = __int_as_float(bi + pop_offset); }
            if (lane < p.U) e[2 + lane] = p.samples[(size_t)(a * p.HU + lane) * p.Nst + bi];
            return;
        }
        if (lane == 0 && p.elites) p.elites[a] = bi;
        if (lane < p.U) p.action[a * p.U + lane] = p.samples[(size_t)(a * p.HU + lane) * p.Nst + bi];
    }
}

// RandomSearch with the population sharded over ranks: all[r][a] = (value, global index, action[U]).  grid A, block 64
static __global__ void k_argmax_merge(RefitArgs p, const float* all, int G) {
    const int a = blockIdx.x, lane = threadIdx.x;
    const size_t pw = (size_t)p.A * (p.U + 2);
    int br = 0, bi = 0x7fffffff;
    float bv = -INFINITY;
    for (int r = 0; r < G; ++r) {
        const float* e = all + pw * r + (size_t)a * (p.U + 2);
        const float v = e[0];
        const int idx = __float_as_int(e[1]);
        if (v > bv || (v == bv && idx < bi) || bi == 0x7fffffff) { bv = v; bi = idx; br = r; }
    }
    const float* src = all + pw * br + (size_t)a * (p.U + 2) + 2;
    if (lane == 0 && p.elites) p.elites[a] = bi;
    if (lane < p.U) p.action[a * p.U + lane] = src[lane];
}

// Tail of OptimizerBase.__call__  optimizer_base.py:82-94 for the analytic pendulum:
// optional exploration noise (+clip), one model step on the [A] rows, pack (action|next_state|reward).
struct FinalArgs {
    int A, U, S;
    int agent_offset;
    int fix_q1, fix_q7;
    int add_noise;
    const float* state;     // [A,S]
    const float* action;    // [A,U]
    const float* lo;
    const float* hi;
    const float* inj;       // injected exploration noise [A,U] or null
    float* record;          // [A][U+S+1]
    float* next_state;      // optional contiguous [A,S]
    RngKey key;
};

__device__ __forceinline__ float exploration_action(const FinalArgs& p, int a, int u, float act) {
    if (!p.add_noise) return act;
    float xi;
    if (p.inj) xi = p.inj[a * p.U + u];
    else {
        U4 b = rng_block(rng_key_now(p.key), 10u /*BBMPC_NOISE_EXPLORATION*/, 0u, 0u, (uint32_t)(p.agent_offset + a), (uint32_t)u);
        xi = word_to_trunc_normal(pick_word(b, (uint32_t)u));
    }
    const float lo = p.lo[u], hi = p.hi[u];
    const float d = lo - hi;
    const float var = (d * d) / 16.0f * 0.05f;                    // optimizer_base.py:46-48
    const float mean = p.fix_q7 ? 0.0f : (hi + lo) / 2.0f;        // :49-50 (quirk Q7: bounds midpoint)
    const float noise = xi * sqrtf(var) + mean;                   // :83-86
    return clipf(act + noise, lo, hi);                            // :87-90
}

__device__ __forceinline__ void finalize_pendulum_agent(const FinalArgs& p, int a, float action_in) {
    const PendulumModel model{p.fix_q1 != 0};
    float s[3] = {p.state[a * 3 + 0], p.state[a * 3 + 1], p.state[a * 3 + 2]};
    float act[1];
    act[0] = exploration_action(p, a, 0, action_in);
    const float r = model.step(s, act);
    float* rec = p.record + (size_t)a * (1 + 3 + 1);
    rec[0] = act[0];
    rec[1] = s[0];
    rec[2] = s[1];
    rec[3] = s[2];
    rec[4] = r;
    if (p.next_state) {
        p.next_state[a * 3 + 0] = s[0];
        p.next_state[a * 3 + 1] = s[1];
        p.next_state[a * 3 + 2] = s[2];
    }
}
// "records ready" for an all-gather that waits on another stream (comm.hpp, gather_records in bbmpc.hip): called by
// one thread per workgroup after its record stores; the last of `nwg` workgroups publishes the sequence number in
// signal memory.  No event and no extra packet on the launch stream.
__device__ __forceinline__ void publish_records_done(unsigned* flag, unsigned* count, unsigned value, unsigned nwg) {
    if (!flag) return;
    // Each workgroup's arrival is a release (its record stores are written back before it is counted); the last arriver
    // alone takes the matching acquire -- as a fence after its relaxed observation of the full count, so that the chain
    // "record stores of every workgroup -> their release arrivals -> acquire fence -> system-scope release store of the
    // flag" holds formally for a consumer that reads the records as soon as it sees the flag (the host polling a pinned
    // word in bbmpc_optimize, or the communication stream's wait-value).  An ACQ_REL arrival in EVERY workgroup would
    // invalidate the XCD's L2 under the workgroups that are still running; one fence in the last one costs nothing.
    if (nwg == 1u) {       // a single workgroup (one agent): its own release store is the whole chain, no counter round trip
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const unsigned old = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg - 1u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// done_flag: optional publish_records_done -- the host-polled bbmpc_optimize returns when it sees it
static __global__ void k_finalize_pendulum(FinalArgs p, unsigned* done_flag, unsigned* done_count, unsigned done_value) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < p.A) finalize_pendulum_agent(p, a, p.action[a]);
    if (done_flag) {
        __syncthreads();
        if (threadIdx.x == 0) publish_records_done(done_flag, done_count, done_value, gridDim.x);
    }
}

}  // namespace bbmpc
