// Persistent per-control-step kernel for PSO on the analytic (true-model) path.
//
// PSOOptimizer._optimize (optimizers/pso.py:70-141) + OptimizerBase.__call__ tail (optimizer_base.py:82-94) in one
// launch, one workgroup per agent, one lane per particle.  The swarm's positions and velocities live in LDS for the
// whole control step (2 x H x Nst floats: 123 KB at N=500, H=30); personal bests stay in HBM/L2 -- they are read
// once per iteration, eight loads in flight, and written only where a particle improved.  Same arithmetic, in the same
// order, as the per-iteration kernels (k_rollout_pendulum<SRC_BUF,PEN> -> k_pso_best -> k_pso_move), which remain the
// path for swarms that do not fit; tests hold the two bit-identical.  The post-loop swarm re-seed stays a separate,
// many-workgroup launch (k_pso_seed).
#pragma once
#include "kernels_opt.hpp"
#include "kernels_refit.hpp"
#include "kernels_rollout.hpp"

namespace bbmpc {

struct FusedPsoArgs {
    int N, A, H, Nst, iters;
    int agent_offset;
    int fix_q1, fix_q7, add_noise;
    float w, c1, c2, v0frac;
    const float* state;      // [A,3]
    const float* lo;
    const float* hi;
    const float* var0;       // [A][H]
    PsoState s;              // global swarm state (pos / vel / pbest [A][H][Nst], pbest_r [A][Nst], gbest [A][H], ...)
    const float* inj2;       // injected r1,r2 [iters][2] or null
    const float* inj_pos;    // injected re-seed truncated normals [A][H][Nst] or null
    const float* inj_vel;    // injected re-seed uniforms [A][H][Nst] or null
    const float* inj_expl;   // injected exploration noise [A,1] or null
    float* record;           // [A][5]
    float* next_state;       // optional [A,3]
    float* t_rewards;        // traces (null when disabled): [iters][A][Nst], [iters][A][H], [iters][A][kstride]
    float* t_mean;
    int* t_elites;
    int t_elite_stride;
    unsigned* done_flag;     // optional: see publish_records_done (kernels_refit.hpp)
    unsigned* done_count;
    unsigned done_value;
    RngKey key;
};

// LDS (floats): pos[H][Nst] | vel[H][Nst] | gb[Hp] | sv[16] | si[16] | misc[8]
template <bool FASTM>
__global__ __launch_bounds__(1024) void k_fused_pso_pendulum(FusedPsoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int a = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
    const int H = p.H, Nst = p.Nst, Hp = (H + 3) & ~3;
    float* posL = smem;
    float* velL = posL + (size_t)H * Nst;
    float* gb = velL + (size_t)H * Nst;
    float* sv = gb + Hp;
    int* si = (int*)(sv + 16);
    int* misc = si + 16;                                      // [0] global best index, [1] it improved this iteration, [4..6] start state
    const int n = tid;
    const bool live = n < p.N;
    const float lo = p.lo[0], hi = p.hi[0];
    float* sst = (float*)(misc + 4);
    if (tid < 3) sst[tid] = p.state[a * 3 + tid];             // one fetch per workgroup (pinned host memory on the zero-copy path)
    float* pos_g = p.s.pos + (size_t)a * H * Nst;
    float* vel_g = p.s.vel + (size_t)a * H * Nst;
    float* pb_g = p.s.pbest + (size_t)a * H * Nst;

    {   // swarm -> LDS: 16-byte loads, four per array in flight per lane (Nst is a multiple of 64)
        const int total4 = (H * Nst) >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(pos_g);
        const float4* v4 = reinterpret_cast<const float4*>(vel_g);
        float4* pl4 = reinterpret_cast<float4*>(posL);
        float4* vl4 = reinterpret_cast<float4*>(velL);
        for (int i0 = tid; i0 < total4; i0 += 4 * nthr) {
            float4 a4[4], b4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * nthr;
                a4[q] = (i < total4) ? p4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                b4[q] = (i < total4) ? v4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * nthr;
                if (i < total4) { pl4[i] = a4[q]; vl4[i] = b4[q]; }
            }
        }
    }
    for (int t = tid; t < H; t += nthr) gb[t] = p.s.gbest[a * H + t];     // iters == 0: action = stored gbest[:,0]
    float pr = live ? p.s.pbest_r[(size_t)a * Nst + n] : -INFINITY;
#ifdef BBMPC_KERNEL_DBG
    long long tk[6] = {0, 0, 0, 0, 0, 0}, tl = (long long)wall_clock64();
    const long long tstart = tl;
#define PSO_MARK(i) do { const long long now_ = (long long)wall_clock64(); tk[i] += now_ - tl; tl = now_; } while (0)
#else
#define PSO_MARK(i) do {} while (0)
#endif
    __syncthreads();
    const float s0 = sst[0], s1 = sst[1], s2 = sst[2];
    PSO_MARK(0);

    for (int it = 0; it < p.iters; ++it) {
        // ---- clip + penalty + rollout (pso.py:76-82; deterministic.py:26-77)
        float R = -INFINITY;
        if (live) {
            Roller<FASTM> roll(p.fix_q1 != 0, s0, s1, s2);
            float total = 0.0f, pen = 0.0f;
            for (int t = 0; t < H; ++t) {
                float x = posL[(size_t)t * Nst + n];
                const float xf = clipf(x, lo, hi);
                const float d = x - xf;
                pen = pen + d * d;
                x = xf;
                posL[(size_t)t * Nst + n] = x;                            // self.pos = feasible positions (:80)
                roll.step_acc(x);
            }
            total = roll.total();
            if (total != total) total = -1.0e6f;                          // deterministic.py:75-77
            const float nr = sqrtf(pen);                                  // tf.norm(...)**2
            R = total - nr * nr;
            if (p.t_rewards) p.t_rewards[((size_t)it * p.A + a) * Nst + n] = R;
        }
        PSO_MARK(1);
        // ---- personal / global best (pso.py:84-100): first maximum wins
        const bool c = live && (pr < R);                                  // tf.less(pbest_r, rewards)
        if (c) pr = R;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        if (live) {
            if (pr > bv || (pr == bv && n < bi)) { bv = pr; bi = n; }
            if (bi == 0x7fffffff) bi = n;                                 // NaN: argmax still returns an index
        }
        wave_argmax(bv, bi);
        if (lane == 0) { sv[wv] = bv; si[wv] = bi; }
        __syncthreads();
        if (wv == 0) {
            bv = (lane < nw) ? sv[lane] : -INFINITY;
            bi = (lane < nw) ? si[lane] : 0x7fffffff;
            wave_argmax(bv, bi);
            if (bi == 0x7fffffff) bi = 0;
            if (lane == 0) misc[0] = bi;
        }
        __syncthreads();
        const int gi = misc[0];
        if (n == gi) {
            p.s.gidx[a] = gi;
            p.s.gbest_r[a] = pr;
            if (p.t_elites) p.t_elites[(size_t)it * p.A * p.t_elite_stride + a] = gi;
            misc[1] = c ? 1 : 0;
        }
        __syncthreads();
        // the best particle's position: one lane per time step (a single lane walking t would pay one memory latency
        // per step whenever the best particle did not move this iteration)
        {
            const bool cg = misc[1] != 0;
            for (int t = tid; t < H; t += nthr) gb[t] = cg ? posL[(size_t)t * Nst + gi] : pb_g[(size_t)t * Nst + gi];
        }
        __syncthreads();
        PSO_MARK(2);
        if (p.t_mean) for (int t = tid; t < H; t += nthr) p.t_mean[((size_t)it * p.A + a) * H + t] = gb[t];
        // ---- velocity / position update (pso.py:86-88, 104-108); r1, r2 are two scalar normals (quirk Q3)
        float r1, r2;
        if (p.inj2) { r1 = p.inj2[2 * it]; r2 = p.inj2[2 * it + 1]; }
        else {
            const U4 b = rng_block(p.key, 5u, (uint32_t)it, 0u, 0u, 0u);
            words_to_normal2(b.x, b.y, r1, r2);
        }
        if (live) {
            // personal bests of particles that did not improve: up to 32 loads in flight per lane
            for (int t0 = 0; t0 < H; t0 += 32) {
                float pbv[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) pbv[i] = (!c && t0 + i < H) ? pb_g[(size_t)(t0 + i) * Nst + n] : 0.0f;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int t = t0 + i;
                    if (t < H) {
                        const float pos = posL[(size_t)t * Nst + n];
                        const float pb = c ? pos : pbv[i];
                        if (c) pb_g[(size_t)t * Nst + n] = pos;
                        const float t1 = velL[(size_t)t * Nst + n] * p.w;
                        const float t2 = ((pb - pos) * p.c1) * r1;
                        const float t3 = ((gb[t] - pos) * p.c2) * r2;
                        const float v = (t1 + t2) + t3;
                        velL[(size_t)t * Nst + n] = v;
                        posL[(size_t)t * Nst + n] = pos + v;
                    }
                }
            }
        }
        __syncthreads();
        PSO_MARK(3);
    }

    // ---- action = gbest[:, 0] (:114).  The swarm re-seed (:116-138: two Philox draws per particle and time step) is
    // left to k_pso_seed, which spreads it over many CUs -- inside this one workgroup it cost 20-25 us.
    const float action0 = gb[0];
    for (int t = tid; t < H; t += nthr) p.s.gbest[a * H + t] = gb[t];
    PSO_MARK(4);
#ifdef BBMPC_KERNEL_DBG
    if (a == 0 && tid == 0 && p.key.step == 5)
        printf("[psodbg] init %lld rollout %lld best %lld move %lld reseed %lld total %lld (10ns)\n", tk[0], tk[1], tk[2], tk[3], tk[4],
               (long long)wall_clock64() - tstart);
#endif
    if (tid == 0) {
        // ---- OptimizerBase.__call__ tail (optimizer_base.py:82-94)
        FinalArgs fa;
        fa.A = p.A; fa.U = 1; fa.S = 3;
        fa.agent_offset = p.agent_offset;
        fa.fix_q1 = p.fix_q1; fa.fix_q7 = p.fix_q7;
        fa.add_noise = p.add_noise;
        fa.lo = p.lo; fa.hi = p.hi;
        fa.inj = p.inj_expl;
        fa.key = p.key;
        fa.key.q_per_agent = 1;
        const PendulumModel model{p.fix_q1 != 0};
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
        publish_records_done(p.done_flag, p.done_count, p.done_value, (unsigned)p.A);
    }
}

}  // namespace bbmpc
