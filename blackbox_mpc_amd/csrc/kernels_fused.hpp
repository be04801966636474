// Persistent per-control-step kernel for the analytic (true-model) path.
//
// One workgroup owns one agent for the WHOLE control step: distribution init, every optimizer
// iteration (sample -> H-step rollout -> reward -> top-k / softmin / argmax -> refit) and the
// OptimizerBase.__call__ tail (exploration noise, predicted next state + reward) run inside a single
// launch, synchronised only by workgroup barriers.  Nothing but the [A,S] state comes in and the
// packed [A,U+S+1] record goes out; rewards, the sampling distribution and (when they fit) the
// candidate action sequences live in LDS, so an optimizer iteration moves no HBM traffic at all.
//
// Replaces, per control step, what the reference executes as one tf.function graph:
//   OptimizerBase.__call__ (optimizer_base.py:55-95) -> {CEM,PI2,RandomSearch}._optimize
//   (cem.py:74-136, pi2.py:58-96, random_search.py:38-48) -> DeterministicTrajectoryEvaluator.__call__
//   (deterministic.py:26-77) -> PendulumTrueModel / pendulum_reward_function (utils/pendulum.py).
#pragma once
#include "kernels_opt.hpp"
#include "kernels_refit.hpp"
#include "kernels_rollout.hpp"
#include "topk.hpp"

namespace bbmpc {

constexpr int FOPT_RS = 1;
constexpr int FOPT_CEM = 2;
constexpr int FOPT_PI2 = 3;
constexpr int FOPT_SPSA = 4;
constexpr int FUSED_MAX_SPSA_ITERS = 16;

struct FusedArgs {
    int N, A, H, U, HU, Nst, k, iters;
    int agent_offset;
    int fix_q1, fix_q7, add_noise;
    int warm_start;          // CEM: BBMPC_FIX_Q2 (keep the mean across control steps)
    int balance;             // progress-balanced wave priorities in the rollout (BBMPC_BALANCE, default on)
    unsigned* done_flag;     // optional completion counter in signal memory (see the kernel's tail), else null
    unsigned* done_count;
    unsigned done_value;
    float alpha, inv_lamda;
    const float* state;      // [A,3]
    const float* lo;
    const float* hi;
    float* prev_mean;        // [A][HU] in/out (warm start)
    const float* var0;       // [A][HU]
    float* mean_out;         // [A][HU] last mean (get_state)
    float* var_out;          // [A][HU]
    float* samples_g;        // [A][HU][Nst] global scratch when the samples do not fit in LDS
    const float* inj;        // standard noise or null: INJ=1 caller-injected [iters][A][HU][Nst];
                             // INJ=2 prefetched by k_noise_fill, [iters][A][Nst][Q] float4 (particle-major Philox blocks)
    const float* inj_expl;   // injected exploration noise [A,U] or null
    float* record;           // [A][U+S+1]
    float* next_state;       // optional contiguous [A,S]
    // traces (null when disabled): [iters][A][Nst], [iters][A][HU], [iters][A][HU], [iters][A][k], [iters][A][HU][Nst]
    float spsa_ak[FUSED_MAX_SPSA_ITERS], spsa_ck[FUSED_MAX_SPSA_ITERS];   // SPSA gain sequences (spsa.py:69-70), per iteration
    float* t_rewards2;       // SPSA trace: rewards of the minus candidates [iters][A][Nst]
    float* t_rewards;
    float* t_mean;
    float* t_var;
    int* t_elites;
    float* t_samples;
    long long* dbg;          // optional phase clocks (debug)
    RngKey key;
    // LINGER variant (one agent, host-in / host-out calls): after publishing its record the workgroup stays on the GPU and
    // polls a 64-byte request line in pinned host memory for the next control step (see the end of the kernel)
    // Every agent's workgroup is on its own: a request line, a completion word and an exit word per agent (one cache line
    // each, arrays indexed by the agent), so no two workgroups ever have to agree on anything.
    const unsigned* mbox;    // [A][16] words: [0] = [15] = sequence number, [1] step, [2] add_noise, [3..4] noise pointer, [5..7] state
    unsigned* gone;          // [A][16] pinned words: the last sequence number handled, written when the workgroup leaves
    unsigned linger_ticks;   // how long to wait for a request, in wall_clock64 ticks (100 MHz)
    int test_quit_agent;     // test hook (BBMPC_LINGER_TEST_QUIT = a + 1): agent a's workgroup leaves after every control step, -1 = none;
                             // 1000 + a: every agent from a on leaves (more than one 16-entry line of the agent map)
    const int* amap;         // optional: blockIdx.x -> agent (a launch for a subset of the agents), null = identity
};

// phase clocks for kernel development: build with -DBBMPC_KERNEL_DBG and run with BBMPC_DBG=1
#ifdef BBMPC_KERNEL_DBG
#define BB_DBG(slot) do { if (p.dbg && tid == 0 && a == 0) dbg_lds[(slot)] = (long long)wall_clock64(); } while (0)
#else
#define BB_DBG(slot) do {} while (0)
#endif

__device__ __forceinline__ float block_min(float v, float* red, int tid, int nw) {
    v = wave_min(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[(tid & 63) < nw ? (tid & 63) : 0];
    r = wave_min(r);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* red, int tid, int nw) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = ((tid & 63) < nw) ? red[tid & 63] : 0.0f;
    r = wave_sum(r);
    __syncthreads();
    return r;
}

// LDS carve (4-byte words): rewards[Nst] | mean[HUp] | var[HUp] | sigma[HUp] | eidx[kp] | red[64] | hist[272] |
//                            ekeys[2*kp] | samples[HU][Nst]      (every piece a multiple of 16 B)
#ifndef BBMPC_FUSED_ACTIONS_AHEAD
#define BBMPC_FUSED_ACTIONS_AHEAD 1
#endif
template <int OPT, bool SAMPLES_LDS, bool FASTM, int INJ, int ILP, bool LINGER = false>
__global__ void k_fused_pendulum(FusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int a = p.amap ? p.amap[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int nw = nthr >> 6;
    const int HUp = (p.HU + 3) & ~3;
    const int kp = (p.k + 3) & ~3;
    float* rew = smem;
    float* mean = rew + p.Nst;
    float* var = mean + HUp;
    float* sigma = var + HUp;
    int* eidx = (int*)(sigma + HUp);
    float* red = (float*)(eidx + kp);
    uint32_t* hist = (uint32_t*)(red + 64);
    unsigned long long* ekeys = (unsigned long long*)(hist + TOPK_HIST_WORDS);
    float* samp = SAMPLES_LDS ? (float*)(ekeys + kp) : (p.samples_g + (size_t)a * p.HU * p.Nst);
#ifdef BBMPC_KERNEL_DBG
    __shared__ long long dbg_lds[48];
#endif
    const PendulumModel model{p.fix_q1 != 0};
    const float lo = p.lo[0], hi = p.hi[0];
    // what changes from one control step to the next (the LINGER variant serves several per launch)
    RngKey key_s = p.key;
    const float* inj_s = p.inj;
    int add_noise_s = p.add_noise;
    unsigned done_value_s = p.done_value;
    // the [A,3] state may live in pinned host memory (bbmpc_optimize's zero-copy path): three lanes fetch it -- one PCIe
    // read per workgroup instead of three per wave (3072 at 64 agents x 16 waves) -- and LDS hands it to everybody
    if (tid < 3) red[60 + tid] = p.state[a * 3 + tid];
    // a resident kernel keeps the next control step's starting distribution in registers (CEM: the same restart mean /
    // variance every time, quirk Q2; PI2 / SPSA / warm-started CEM: the mean it has just written to prev_mean)
    const bool keep_init = LINGER && OPT != FOPT_RS && p.HU <= nthr;
    [[maybe_unused]] float m_keep = 0.0f, v_keep = 0.0f, s_keep = 0.0f;
    [[maybe_unused]] bool have_init = false;
    for (;;) {      // one pass per control step; a single pass unless LINGER

        // ---- distribution init (cem.py:129-132 starts every control step from the ctor mean/var, quirk Q2)
        if (keep_init && have_init) {        // resident pass: no global round trip
            if (tid < p.HU) {
                mean[tid] = m_keep;
                var[tid] = v_keep;
                sigma[tid] = (OPT == FOPT_CEM && p.warm_start) ? cem_sigma(m_keep, v_keep, lo, hi) : s_keep;
            }
        } else {
            for (int j = tid; j < p.HU; j += nthr) {
                const float m = p.prev_mean[a * p.HU + j];
                const float v = p.var0[a * p.HU + j];
                const float sg = (OPT == FOPT_CEM) ? cem_sigma(m, v, lo, hi) : ((OPT == FOPT_SPSA) ? 0.0f : sqrtf(v));
                mean[j] = m;
                var[j] = v;
                sigma[j] = sg;
                if (keep_init) { m_keep = m; v_keep = v; s_keep = sg; }
            }
            have_init = true;
        }
        __syncthreads();
        const float s0 = red[60], s1 = red[61], s2 = red[62];

        float action0 = (OPT == FOPT_RS) ? 0.0f : mean[0];          // iters == 0 -> untouched mean[:,0]

        BB_DBG(0);
#ifdef BBMPC_KERNEL_DBG
        if (p.dbg && tid == 0 && a == 0) dbg_lds[40] = (long long)clock64();
#endif
        // (INJ == 2) the first two blocks of a thread's first trajectory are fetched while the PREVIOUS iteration selects and
        // refits: at the top of the rollout their loads would be one exposed memory round trip per iteration
        typedef float pfvec4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) pfvec4* pfvec4p;
        [[maybe_unused]] pfvec4 pf0 = {0.0f, 0.0f, 0.0f, 0.0f}, pf1 = {0.0f, 0.0f, 0.0f, 0.0f};
        [[maybe_unused]] auto prefetch_first = [&](int it_n) {
            if constexpr (INJ == 2) {
                const int Q = (p.HU + 3) >> 2;
                const pfvec4p q = (pfvec4p)(reinterpret_cast<const float4*>(inj_s) + (((size_t)it_n * p.A + a) * p.Nst + min(tid, p.N - 1)) * Q);
                pf0 = q[0];
                if (OPT != FOPT_SPSA) pf1 = q[min(1, Q - 1)];
            }
        };
        if (p.iters > 0) prefetch_first(0);
        for (int it = 0; it < p.iters; ++it) {
            BB_DBG(1 + it * 4);
            // ---- sample + rollout: one lane per trajectory, state in VGPRs
            // The 4 candidate actions of Philox block b+1 are generated while the recurrence steps through
            // block b: their instructions carry no dependence on the state, so they fill the latency
            // shadows of the sequential theta/thdot chain (all straight-line code inside a block).
            const float* inj = (INJ == 1) ? inj_s + ((size_t)it * p.A + a) * p.HU * p.Nst : nullptr;
            const int nblk = p.H >> 2, rem = p.H & 3;
            const uint32_t rstream = (OPT == FOPT_RS) ? 2u : 1u;
            // ILP independent trajectories per lane: with half as many waves each SIMD runs a single wave whose two
            // recurrences interleave in program order, instead of two waves fighting over issue slots.
            if constexpr (OPT == FOPT_SPSA) {
                // SPSA (spsa.py:61-107): theta +- c_k * delta with delta in {-1,+1}; both candidates of a particle are
                // rolled out by the same lane (two independent recurrences: four chains per SIMD at N = 500).  The
                // Rademacher draws stay in LDS (samp) for the gradient estimate.
                const float ck = p.spsa_ck[it];
                const int nb4 = (p.H + 3) >> 2;
                for (int n = tid; n < p.N; n += nthr) {
                    Roller<FASTM> rp, rm;
                    rp.init(p.fix_q1 != 0, s0, s1, s2);
                    rm.init(p.fix_q1 != 0, s0, s1, s2);
                    float tp = 0.0f, tm = 0.0f, pp = 0.0f, pm = 0.0f;
                    // (a GLOBAL pointer, said so: as a generic pointer -- inj_s may have come out of the mailbox -- its loads are
                    // flat loads, which count on the LDS counter as well: the mean[t] reads below then wait for the prefetch)
                    typedef float gvec4s __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(1))) gvec4s* gvec4sp;
                    [[maybe_unused]] gvec4sp mine4 = nullptr;
                    [[maybe_unused]] gvec4s cur4 = {1.0f, 1.0f, 1.0f, 1.0f};
                    if constexpr (INJ == 2) {            // draws prefetched by k_noise_fill, one float4 per 4 steps
                        mine4 = (gvec4sp)(reinterpret_cast<const float4*>(inj_s) + (((size_t)it * p.A + a) * p.Nst + n) * nb4);
                        if (n == tid) cur4 = pf0;                    // fetched while the previous iteration refitted
                        else cur4 = mine4[0];
                    }
                    for (int b = 0; b < nb4; ++b) {
                        float d[4];
                        if constexpr (INJ == 2) {
                            const gvec4s nxt4 = mine4[min(b + 1, nb4 - 1)];
                            d[0] = cur4.x; d[1] = cur4.y; d[2] = cur4.z; d[3] = cur4.w;
                            cur4 = nxt4;
                        } else if (INJ == 1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) d[i] = (4 * b + i < p.H) ? inj[(size_t)(4 * b + i) * p.Nst + n] : 1.0f;
                        } else {
                            const U4 w = rng_block(key_s, 3u, (uint32_t)it, (uint32_t)n, (uint32_t)(p.agent_offset + a), (uint32_t)(4 * b));
                            d[0] = word_to_rademacher(w.x); d[1] = word_to_rademacher(w.y);
                            d[2] = word_to_rademacher(w.z); d[3] = word_to_rademacher(w.w);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int t = 4 * b + i;
                            if (t < p.H) {
                                const float th = mean[t], stp = ck * d[i];
                                const float xp = th + stp, xm = th - stp;                      // spsa.py:76-77
                                const float xpf = clipf(xp, lo, hi), xmf = clipf(xm, lo, hi);  // :78-81
                                const float dp = xp - xpf, dm = xm - xmf;
                                pp = pp + dp * dp;
                                pm = pm + dm * dm;
                                samp[(size_t)t * p.Nst + n] = d[i];
                                rp.step_acc(xpf);
                                rm.step_acc(xmf);
                            }
                        }
                    }
                    tp = rp.total(); tm = rm.total();
                    if (tp != tp) tp = -1.0e6f;                                                // deterministic.py:75-77
                    if (tm != tm) tm = -1.0e6f;
                    const float np_ = sqrtf(pp), nm_ = sqrtf(pm);                              // tf.norm(...)**2  spsa.py:82-89
                    const float r_p = tp - np_ * np_, r_m = tm - nm_ * nm_;                    // :98-99
                    if (p.t_rewards) {
                        p.t_rewards[((size_t)it * p.A + a) * p.Nst + n] = r_p;
                        p.t_rewards2[((size_t)it * p.A + a) * p.Nst + n] = r_m;
                    }
                    rew[n] = r_p - r_m;
                }
            } else if constexpr (INJ == 2) {
                // Draws prefetched by idle CUs (k_noise_fill): one float4 = one Philox block = 4 steps of this trajectory.
                // Loads run TWO blocks (8 steps, ~2 us) ahead of their use so that L2/HBM latency never reaches the
                // recurrence; the Philox rounds (a quarter of this kernel's VALU work) are gone from the critical path.
                const int Q = (p.HU + 3) >> 2;
                const float4* inj4 = reinterpret_cast<const float4*>(inj_s) + ((size_t)it * p.A + a) * p.Nst * Q;
                int* prog = (int*)red;                             // per-wave progress (red[] is idle during the rollout)
                const int wave = tid >> 6, lane = tid & 63;
                const bool balance = p.balance != 0 && nw > 4 && nw <= 64 && p.N <= nthr;
                for (int n = tid; n < p.N; n += nthr) {
                    Roller<FASTM> roll;
                    roll.init(p.fix_q1 != 0, s0, s1, s2);
                    float pen = 0.0f;
                    // (a GLOBAL pointer, said so: inj_s may have come out of the mailbox, and as a generic pointer its loads are
                    // flat loads -- which count on the LDS counter too, so that every LDS wait of the loop below waited for the
                    // block it had just prefetched: one memory round trip per eight model steps)
                    typedef float gvec4 __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(1))) gvec4* gvec4p;
                    const gvec4p mine = (gvec4p)(inj4 + (size_t)n * Q);
                    auto ld = [&](int b) { const gvec4 v = mine[min(b, Q - 1)]; return make_float4(v.x, v.y, v.z, v.w); };
                    auto step1 = [&](int t, float xi) {
                        float x = (OPT == FOPT_RS) ? xi * (hi - lo) + lo : xi * sigma[t] + mean[t];
                        if (OPT == FOPT_PI2) {
                            const float xf = clipf(x, lo, hi);
                            const float d = x - xf;
                            pen = pen + d * d;
                            x = xf;
                        }
                        samp[(size_t)t * p.Nst + n] = x;
                        roll.step_acc(x);
                    };
#if BBMPC_FUSED_ACTIONS_AHEAD
                    // a block's four actions are formed (sigma[t], mean[t] read from LDS) BEFORE its model steps: inside the
                    // steps the reads would sit between the sine and the angle update of the pendulum's recurrence
                    auto block4 = [&](const float4& z, int b) {
                        const float zz[4] = {z.x, z.y, z.z, z.w};
                        float x[4];
                        if (OPT == FOPT_RS) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) x[i] = zz[i] * (hi - lo) + lo;
                        } else {
                            // (a full block: t = 4b .. 4b + 3 < H; mean / sigma are 16-byte aligned and padded to a multiple of four)
                            const float4 sg = *reinterpret_cast<const float4*>(sigma + 4 * b), mn = *reinterpret_cast<const float4*>(mean + 4 * b);
                            x[0] = zz[0] * sg.x + mn.x; x[1] = zz[1] * sg.y + mn.y; x[2] = zz[2] * sg.z + mn.z; x[3] = zz[3] * sg.w + mn.w;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(x[i]));
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = x[i];
                            if (OPT == FOPT_PI2) {
                                const float xf = clipf(v, lo, hi);
                                const float d = v - xf;
                                pen = pen + d * d;
                                v = xf;
                            }
                            samp[(size_t)(4 * b + i) * p.Nst + n] = v;
                            roll.step_acc(v);
                        }
                    };
#else
                    auto block4 = [&](const float4& z, int b) {
                        step1(4 * b + 0, z.x); step1(4 * b + 1, z.y); step1(4 * b + 2, z.z); step1(4 * b + 3, z.w);
                    };
#endif
                    float4 c0, c1;
                    if (n == tid) { c0 = make_float4(pf0.x, pf0.y, pf0.z, pf0.w); c1 = make_float4(pf1.x, pf1.y, pf1.z, pf1.w); }   // fetched an iteration ago
                    else { c0 = ld(0); c1 = ld(1); }
                    int b = 0;
                    // Waves that share a SIMD (wave ids equal mod 4) run the same instruction stream, and the arbiter
                    // favours the older one: it finishes early and the younger one then runs alone at a lone wave's
                    // issue rate (measured: 4.5 us vs 6.0 us for the two halves of a 500-particle population).  Each
                    // wave publishes its block counter in LDS and lowers its priority while it is ahead of a SIMD mate.
                    if (balance && lane == 0) prog[wave] = 0;
                    for (; b + 1 < nblk; b += 2) {
                        const float4 n0 = ld(b + 2), n1 = ld(b + 3);
                        if (balance) {
                            if (lane == 0) prog[wave] = b;
                            int behind = b;
                            for (int w2 = wave & 3; w2 < nw; w2 += 4) behind = min(behind, prog[w2]);
                            if (behind < b) __builtin_amdgcn_s_setprio(0);
                            else __builtin_amdgcn_s_setprio(2);
                        }
                        block4(c0, b);
                        block4(c1, b + 1);
                        c0 = n0; c1 = n1;
                    }
                    if (balance) {
                        if (lane == 0) prog[wave] = 1 << 30;
                        __builtin_amdgcn_s_setprio(0);
                    }
                    if (b < nblk) { block4(c0, b); c0 = c1; ++b; }
                    if (rem > 0) step1(4 * b + 0, c0.x);
                    if (rem > 1) step1(4 * b + 1, c0.y);
                    if (rem > 2) step1(4 * b + 2, c0.z);
                    float tot = roll.total();
                    if (tot != tot) tot = -1.0e6f;                                   // deterministic.py:75-77
                    if (OPT == FOPT_PI2) {
                        const float nr = sqrtf(pen);
                        tot = tot - nr * nr;
                    }
                    rew[n] = tot;
                }
            } else
            for (int n0 = tid; n0 < p.N; n0 += ILP * nthr) {
                int nn[ILP];
                bool live[ILP];
                Roller<FASTM> roll[ILP];
                float pen[ILP], xn[ILP][4];
#pragma unroll
                for (int q = 0; q < ILP; ++q) {
                    live[q] = n0 + q * nthr < p.N;
                    nn[q] = live[q] ? n0 + q * nthr : n0;          // idle slots shadow particle n0 (never stored)
                    roll[q].init(p.fix_q1 != 0, s0, s1, s2);
                    pen[q] = 0.0f;
                }
                auto gen = [&](int q, int b) {
                    const int n = nn[q];
                    float xi[4];
                    if (INJ) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) xi[i] = (4 * b + i < p.H) ? inj[(size_t)(4 * b + i) * p.Nst + n] : 0.0f;
                    } else {
                        const U4 w = rng_block(key_s, rstream, (uint32_t)it, (uint32_t)n, (uint32_t)(p.agent_offset + a),
                                               (uint32_t)(4 * b));
                        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            xi[i] = (OPT == FOPT_RS) ? word_to_uniform(ww[i]) : word_to_trunc_normal(ww[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = min(4 * b + i, p.H - 1);
                        if (OPT == FOPT_RS) xn[q][i] = xi[i] * (hi - lo) + lo;      // random_search.py:40-41
                        else xn[q][i] = xi[i] * sigma[t] + mean[t];                 // cem.py:90-94 / pi2.py:65-69
                    }
                };
                auto consume = [&](int q, int t, float x) {
                    if (OPT == FOPT_PI2) {                                           // pi2.py:70-75
                        const float xf = clipf(x, lo, hi);
                        const float d = x - xf;
                        pen[q] = pen[q] + d * d;
                        x = xf;
                    }
                    if (live[q]) samp[(size_t)t * p.Nst + nn[q]] = x;
                    roll[q].step_acc(x);     // (the H-step sum is the roller's: the same form as the prefetched-draws path above, bit for bit)
                };
#pragma unroll
                for (int q = 0; q < ILP; ++q) gen(q, 0);
                for (int b = 0; b < nblk; ++b) {
                    float xc[ILP][4];
#pragma unroll
                    for (int q = 0; q < ILP; ++q) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) xc[q][i] = xn[q][i];
                        gen(q, b + 1);       // one block ahead (the block past the end is generated and dropped)
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < ILP; ++q) consume(q, 4 * b + i, xc[q][i]);
                }
                for (int i = 0; i < rem; ++i)
#pragma unroll
                    for (int q = 0; q < ILP; ++q) consume(q, 4 * nblk + i, xn[q][i]);
#pragma unroll
                for (int q = 0; q < ILP; ++q) {
                    float tot = roll[q].total();
                    if (tot != tot) tot = -1.0e6f;                                   // deterministic.py:75-77
                    if (OPT == FOPT_PI2) {
                        const float nr = sqrtf(pen[q]);
                        tot = tot - nr * nr;
                    }
                    if (live[q]) rew[nn[q]] = tot;
                }
            }
#ifdef BBMPC_KERNEL_DBG
            if (p.dbg && a == 0 && it == 0 && (tid & 63) == 0) dbg_lds[24 + (tid >> 6) % 8] = (long long)wall_clock64();
#endif
            // the selection's first part reads only the rewards this thread has just written: in front of the barrier the
            // rollout needs anyway instead of behind it with a barrier of its own (topk.hpp)
            if (it + 1 < p.iters) prefetch_first(it + 1);
            if (OPT == FOPT_CEM) block_topk_prepass(rew, p.N, p.k, hist, tid, nthr);
            BB_DBG(2 + it * 4);
            __syncthreads();
            BB_DBG(3 + it * 4);
            if (p.t_rewards) {
                if (OPT != FOPT_SPSA)
                    for (int n = tid; n < p.N; n += nthr) p.t_rewards[((size_t)it * p.A + a) * p.Nst + n] = rew[n];
                for (int i = tid; i < p.HU * p.Nst; i += nthr) {
                    const int j = i / p.Nst, n = i % p.Nst;
                    if (n < p.N) p.t_samples[(((size_t)it * p.A + a) * p.HU + j) * p.Nst + n] = samp[(size_t)j * p.Nst + n];
                }
            }

            // ---- refit
            if (OPT == FOPT_CEM) {
                // exact sorted top-k (tf.nn.top_k, cem.py:97-99): LDS radix select, see topk.hpp
                // The refit needs the elite SET only, so the winners are listed in ascending index order (no ranking);
                // the sorted list tf.nn.top_k returns is produced only for the parity trace.  The statistics below always
                // run over the index-ordered list, so results do not depend on tracing.
                const TopkSel sel = block_topk_select(rew, p.N, p.k, hist, tid, nthr, true);
                if (p.t_elites) {
                    block_topk_finish_sorted(rew, p.N, p.k, eidx, hist, ekeys, sel, tid, nthr);
                    for (int e = tid; e < p.k; e += nthr) p.t_elites[((size_t)it * p.A + a) * p.k + e] = eidx[e];
                    __syncthreads();
                }
                block_topk_finish_indexed(rew, p.N, p.k, eidx, hist, sel, tid, nthr);
                BB_DBG(4 + it * 4);
#ifdef BBMPC_KERNEL_DBG
                if (p.dbg && tid == 0 && a == 0 && it == 1) for (int i = 0; i < 12; ++i) p.dbg[48 + i] = g_topk_dbg[i];
#endif
                // elite statistics (cem.py:112-125): one 16-lane DPP row per (h,u) element -- 4 elements per wave,
                // every element of the horizon at once for H*U <= 4*waves; mean and biased variance two-pass,
                // as the reference computes them.
                const float kf = (float)p.k;
                const int sub = tid & 15;
                // up to 64 elites: a lane's (<= 4) elite indices and sample values stay in registers -- every LDS read of
                // a phase is issued before the first use (a dependent eidx -> row read per element costs two LDS
                // latencies each, 16 in a row for k = 50)
                constexpr int EC = 4;
                const bool small_k = p.k <= 16 * EC;
                int ei[EC];
#pragma unroll
                for (int i = 0; i < EC; ++i) ei[i] = (small_k && sub + 16 * i < p.k) ? eidx[sub + 16 * i] : -1;
                for (int j = tid >> 4; j < ((p.HU + 3) & ~3); j += nthr >> 4) {       // all 16 lanes of a row share j
                    const bool live = j < p.HU;
                    const float* row = samp + (size_t)(live ? j : 0) * p.Nst;
                    float sum = 0.0f, vs = 0.0f, em;
                    if (small_k) {
                        float x[EC];
#pragma unroll
                        for (int i = 0; i < EC; ++i) x[i] = row[ei[i] >= 0 ? ei[i] : 0];
#pragma unroll
                        for (int i = 0; i < EC; ++i) if (ei[i] >= 0) sum += x[i];
                        sum = row16_sum(sum);
                        em = sum / kf;                                               // cem.py:112
#pragma unroll
                        for (int i = 0; i < EC; ++i)
                            if (ei[i] >= 0) {
                                const float d = x[i] - em;
                                vs += d * d;
                            }
                    } else {
                        for (int e = sub; e < p.k; e += 16) sum += row[eidx[e]];
                        sum = row16_sum(sum);
                        em = sum / kf;
                        for (int e = sub; e < p.k; e += 16) {
                            const float d = row[eidx[e]] - em;
                            vs += d * d;
                        }
                    }
                    vs = row16_sum(vs);
                    if (live && sub == 0) {
                        const float ev = vs / kf;                                    // cem.py:113-119
                        const float one_m = 1.0f - p.alpha;
                        const float m = p.alpha * mean[j] + one_m * em;              // cem.py:121-122
                        const float v = p.alpha * var[j] + one_m * ev;               // cem.py:123-125
                        mean[j] = m;
                        var[j] = v;
                        sigma[j] = cem_sigma(m, v, lo, hi);
                    }
                }
                if (it == 0) BB_DBG(33);
                __syncthreads();
                action0 = mean[0];                                                   // cem.py:135
            } else if (OPT == FOPT_PI2) {
                // pi2.py:78-87.  Cross-wave reductions use their own LDS slots (no second barrier to recycle them) and the
                // per-wave slots are combined by every wave with one LDS round trip + a DPP reduction.
                float lmin = INFINITY;
                for (int n = tid; n < p.N; n += nthr) {
                    const float c = -rew[n];                                         // pi2.py:78
                    rew[n] = c;
                    lmin = fminf(lmin, c);
                }
                lmin = wave_min(lmin);
                if ((tid & 63) == 0) red[tid >> 6] = lmin;
                __syncthreads();
                float beta = ((tid & 63) < nw) ? red[tid & 63] : INFINITY;
                beta = wave_min(beta);                                               // pi2.py:81
                float lsum = 0.0f;
                for (int n = tid; n < p.N; n += nthr) {
                    const float pr = expf((-p.inv_lamda) * (rew[n] - beta));         // pi2.py:82
                    rew[n] = pr;
                    lsum += pr;
                }
                lsum = wave_sum(lsum);
                if ((tid & 63) == 0) red[32 + (tid >> 6)] = lsum;
                __syncthreads();
                float eta = ((tid & 63) < nw) ? red[32 + (tid & 63)] : 0.0f;
                eta = wave_sum(eta);                                                 // pi2.py:83
                const float inv_eta = 1.0f / eta;
                // new_mean[j] = sum_n x[j][n] * omega[n],  omega = inv_eta * prob   (pi2.py:85-87): one wave per j
                for (int j = tid >> 6; j < p.HU; j += nw) {
                    const float* row = samp + (size_t)j * p.Nst;
                    float acc = 0.0f;
                    for (int n = tid & 63; n < p.N; n += 64) acc += row[n] * (inv_eta * rew[n]);
                    acc = wave_sum(acc);
                    if ((tid & 63) == 0) mean[j] = acc;
                }
                __syncthreads();
                action0 = mean[0];                                                   // pi2.py:94
            } else if (OPT == FOPT_SPSA) {
                // ghat[j] = mean_n (r+ - r-)[n] / (2 c_k delta[j][n]);  theta = clip(theta + a_k ghat)   (spsa.py:101-107)
                const float two_ck = 2.0f * p.spsa_ck[it], ak = p.spsa_ak[it];
                for (int j = tid >> 6; j < p.HU; j += nw) {
                    const float* row = samp + (size_t)j * p.Nst;
                    float acc = 0.0f;
                    for (int n = tid & 63; n < p.N; n += 64) acc += rew[n] / (two_ck * row[n]);
                    acc = wave_sum(acc);
                    if ((tid & 63) == 0) mean[j] = clipf(mean[j] + ak * (acc / (float)p.N), lo, hi);
                }
                __syncthreads();
                action0 = mean[0];                                                   // spsa.py:117
            } else {  // RandomSearch: argmax, first maximum (random_search.py:43-47)
                float bv = -INFINITY;
                int bi = 0x7fffffff;
                for (int n = tid; n < p.N; n += nthr) {
                    const float v = rew[n];
                    if (v > bv || (v == bv && n < bi)) { bv = v; bi = n; }
                }
                wave_argmax(bv, bi);
                int* redi = (int*)(red + 32);
                if ((tid & 63) == 0) { red[tid >> 6] = bv; redi[tid >> 6] = bi; }
                __syncthreads();
                bv = ((tid & 63) < nw) ? red[tid & 63] : -INFINITY;
                bi = ((tid & 63) < nw) ? redi[tid & 63] : 0x7fffffff;
                wave_argmax(bv, bi);
                if (bi == 0x7fffffff) bi = 0;
                action0 = samp[bi];                                                  // sample[best][t=0]
                if (p.t_elites && tid == 0) p.t_elites[a] = bi;
                __syncthreads();
            }
            if (p.t_mean && OPT != FOPT_RS)
                for (int j = tid; j < p.HU; j += nthr) {
                    p.t_mean[((size_t)it * p.A + a) * p.HU + j] = mean[j];
                    p.t_var[((size_t)it * p.A + a) * p.HU + j] = var[j];
                }
        }

        BB_DBG(1 + p.iters * 4);
#ifdef BBMPC_KERNEL_DBG
        if (p.dbg && tid == 0 && a == 0) {
            dbg_lds[41] = (long long)clock64();
            for (int i = 0; i < 48; ++i) p.dbg[i] = dbg_lds[i];
        }
#endif
        // ---- state carried to the next control step
        if (OPT != FOPT_RS) {
            for (int j = tid; j < p.HU; j += nthr) {
                p.mean_out[a * p.HU + j] = mean[j];
                p.var_out[a * p.HU + j] = var[j];
                if (OPT == FOPT_PI2 || OPT == FOPT_SPSA) {                           // shift-left warm start pi2.py:92-93 / spsa.py:114-115
                    const int js = (j + 1 < p.HU) ? j + 1 : p.HU - 1;
                    const float pm = mean[js];
                    p.prev_mean[a * p.HU + j] = pm;
                    if (keep_init) m_keep = pm;
                } else if (p.warm_start) {
                    const float pm = mean[j];
                    p.prev_mean[a * p.HU + j] = pm;
                    if (keep_init) m_keep = pm;
                }
            }
        }

        // ---- OptimizerBase.__call__ tail (optimizer_base.py:82-94)
        if (tid == 0) {
            FinalArgs fa;
            fa.A = p.A; fa.U = 1; fa.S = 3;
            fa.agent_offset = p.agent_offset;
            fa.fix_q1 = p.fix_q1; fa.fix_q7 = p.fix_q7;
            fa.add_noise = add_noise_s;
            fa.lo = p.lo; fa.hi = p.hi;
            fa.inj = p.inj_expl;
            fa.key = key_s;
            fa.key.q_per_agent = 1;
            float s[3] = {s0, s1, s2};
            float act[1];
            act[0] = exploration_action(fa, a, 0, action0);
            const float r = model.step(s, act);
            float* rec = p.record + (size_t)a * 5;
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
            // "records ready" for the all-gather that waits on another stream (comm.hpp): the last agent's workgroup
            // publishes the sequence number once every agent's record is in HBM.  No event, no extra packet on the
            // launch stream.
            if constexpr (LINGER) publish_records_done(p.done_flag + a * 16, nullptr, done_value_s, 1u);    // this agent's own completion word
            else publish_records_done(p.done_flag, p.done_count, done_value_s, gridDim.x);
        }
        if constexpr (!LINGER) {
            break;
        } else {
            // ---- stay resident: wave 0 polls the request line over PCIe (one 64-byte read per poll).  Words 0..12 of a
            // request carry 16 bits of payload each plus the low 16 bits of the request's sequence number, word 15 the
            // whole number (Engine::resident_step): a line is accepted when ALL fourteen words name the request this
            // workgroup waits for, so neither the order of the host's stores nor how the fabric splits the read matters.
            // A request carrying the NEXT sequence number starts another pass without a launch (measured: 2.1 us
            // host-to-host for a resident kernel's mailbox round trip against 6.8 us for launch + completion,
            // tools/microbench/launch_latency.hip); the stop word in word 15, or linger_ticks without a request, ends the
            // kernel, which says so in `gone` first and never looks at the line again, so the host knows whether to launch.
            unsigned* mb = (unsigned*)ekeys;                       // 16 words of LDS that are idle outside the top-k
            __syncthreads();
            if (tid < 64) {
                const long long t0 = (long long)wall_clock64();
                const unsigned want = done_value_s + 1u;
                unsigned w = 0u;
                bool quit = false;
                for (;;) {
                    w = (tid < 16) ? __hip_atomic_load(p.mbox + a * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                    const bool mine = tid < 13 ? (w & 0xffffu) == (want & 0xffffu) : (tid == 15 ? w == want : true);
                    const unsigned w15 = __builtin_amdgcn_readlane(w, 15);
                    if (__builtin_amdgcn_ballot_w64(mine) == ~0ull) break;
                    if (w15 == 0xffffffffu || (long long)wall_clock64() - t0 > (long long)p.linger_ticks || a == p.test_quit_agent || (p.test_quit_agent >= 999 && a >= p.test_quit_agent - 999)) { quit = true; break; }
                }
                if (tid < 16) mb[tid] = (quit && tid == 15) ? 0xffffffffu : w;
            }
            __syncthreads();
            if (mb[15] == 0xffffffffu) {
                if (tid == 0) __hip_atomic_store(p.gone + a * 16, done_value_s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            done_value_s = mb[15];
            key_s.step = (mb[0] >> 16) | (mb[1] & 0xffff0000u);
            add_noise_s = (int)(mb[2] >> 16);
            inj_s = reinterpret_cast<const float*>(((unsigned long long)(mb[3] >> 16)) | ((unsigned long long)(mb[4] >> 16) << 16) |
                                                   ((unsigned long long)(mb[5] >> 16) << 32) | ((unsigned long long)(mb[6] >> 16) << 48));
            const int sw_ = 7 + 2 * min(tid, 2);
            const float st_next = __uint_as_float((mb[sw_] >> 16) | (mb[sw_ + 1] & 0xffff0000u));
            __syncthreads();                                        // every thread has read the request before `red` / `ekeys` are reused
            if (tid < 3) red[60 + tid] = st_next;
        }
    }
}

}  // namespace bbmpc
