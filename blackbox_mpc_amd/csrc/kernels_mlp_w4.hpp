// Learned-dynamics rollout for SMALL networks, one wave per four particles ("w4"): the 4-32-32-32-3 Pendulum models of the
// reference's learn_dynamics / model_based_RL / low_level_api tutorials and the 26-32-32-32-20 HalfCheetah model of
// tutorials/mujoco/tutorial_one.py:22-31.  Same fused path as the other rollout kernels (reference files under blackbox_mpc/):
//   SystemDynamicsHandler.process_input / process_output   dynamics_handlers/system_dynamics_handler.py:97-161
//   DeterministicMLP.__call__                              dynamics_functions/deterministic_mlp.py:27-51
//   DeterministicTrajectoryEvaluator.__call__              trajectory_evaluators/deterministic.py:26-77
//
// k_rollout_mlp_wave gives a 16-particle tile to HT waves and pays an LDS all-gather + barrier per layer: a model step of
// the four-layer Pendulum network is 1.75 us of which the matrix work is a tenth.  Here a wave owns four particles for the
// whole recurrence and NOTHING crosses LDS between layers: every Dense layer is padded to 32 x 32 and runs as 16
// v_mfma_f32_4x4x1_16b_f32 whose 16 blocks are (output quad 4*(row & 1) + g) x (K half row >> 1); the two K halves are
// folded by one v_permlane32_swap + add per register pair (a reduce-scatter: rows 0, 1 end up with features 4q + {0, 2},
// rows 2, 3 with 4q + {1, 3} of their quad -- two activations per lane instead of four), and the next layer's B operands
// are made from those two registers by one v_permlane16_swap + v_permlane32_swap each (every block then holds the four
// features of quad 4*(K half) + g of its own K half) and three DPP row rotations (the other three quads; the stationary A
// operands are loaded in the order the rotations deliver the k).  Output feature f of a layer lands in the lane / register
// input feature f of the next one is read from, and so does the state across planning steps: the epilogue (last
// activation, de-normalisation, residual, normalisation) is a handful of operations on two registers.  The input is laid
// out as state features 0..19, action features 20..27 (dim_S <= 20, dim_U <= 8); the raw state of every step goes to an
// LDS ring, from which all threads score the H x 16 (state, action, next state) triples after the recurrence -- the
// Pendulum reward's atan2 + floor-mod never sits in the dependent chain.
// The four waves of a workgroup share the action block's prologue and nothing else: no barrier inside the recurrence.
#pragma once
#include "kernels_mlp_q4s.hpp"

namespace bbmpc {

constexpr int W4_TP = 16;                                 // particles per workgroup (four per wave)
// LDS, in floats: acts [H][16][U] | pens | xa [H][16][8] normalised actions | ring [H+1][16][Sp] raw states | rstep [H][16] |
//                 d2s [H][16][U] squared clip distances (summed per (particle, u) in step order behind the recurrence)
struct MlpW4Lds {
    int acts, pens, xa, ring, rstep, d2s, total;
};
constexpr int W4_OPS = 18;                                // operands per lane and layer: 16 A operands + 2 biases
// lane -> (row, quad g, i): what bbmpc_set_mlp packs for it (and the kernel below reads back as one coalesced load each)
__host__ __device__ inline void mlp_w4_lane(int lane, int& kh, int& qo, int& i0, int& g) {
    const int row = lane >> 4;
    g = (lane >> 2) & 3;
    kh = row >> 1;                                        // the K half the lane's block multiplies
    qo = 4 * (row & 1) + g;                               // the output quad the block produces
    i0 = row >> 1;                                        // the lane keeps features 4*qo + i0 and 4*qo + i0 + 2 after the fold
}
// Operand `op` (< 16: rotation j = op >> 2, register c = op & 3; 16, 17: the two biases) of layer l for `lane`: index into the
// Dense kernel [in][out] / the bias, or -1 for padding.  The first layer's k runs over (state 0..19 | action 20..27).
__host__ __device__ inline int mlp_w4_operand_index(int l, int op, int lane, int K, int M, int S, int U) {
    int kh, qo, i0, g;
    mlp_w4_lane(lane, kh, qo, i0, g);
    if (op >= 16) {
        const int f = 4 * qo + i0 + 2 * (op - 16);
        return f < M ? f : -1;
    }
    const int j = op >> 2, c = op & 3;
    const int k = 4 * (4 * kh + ((g - j) & 3)) + c, o = 4 * qo + (lane & 3);
    int kk = k;
    if (l == 0) kk = k < 20 ? (k < S ? k : -1) : (k - 20 < U ? S + k - 20 : -1);
    return (kk >= 0 && kk < K && o < M) ? kk * M + o : -1;
}
__host__ __device__ inline MlpW4Lds mlp_w4_lds_layout(int H, int U, int S) {
    MlpW4Lds l;
    const int Sp = (S + 3) & ~3;
    int o = 0;
    l.acts = o; o += ((H * W4_TP * U + 3) & ~3);
    l.pens = o; o += ((W4_TP * U + 63) & ~63);
    l.xa = o;   o += H * W4_TP * 8;
    l.ring = o; o += (H + 1) * W4_TP * Sp;
    l.rstep = o; o += H * W4_TP;
    l.d2s = o; o += ((H * W4_TP * U + 3) & ~3);
    l.total = o;
    return l;
}

// NL Dense layers (2..4) of at most 32 inputs / outputs each (the first: dim_S <= 20 state + dim_U <= 8 action inputs).
// TANH: every hidden activation is tanh and the output layer is linear (the tutorials' networks): no run-time dispatch.
// grid (ceil(n_pop / 16), A or 1), block 256.
template <int NL, bool TANH>
__global__ __launch_bounds__(256) void k_rollout_mlp_w4(MlpRolloutArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    const int a = blockIdx.y, n0 = blockIdx.x * W4_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = p.S, U = p.U, H = p.H, Sp = (S + 3) & ~3;
    const MlpW4Lds lay = mlp_w4_lds_layout(H, U, S);
    float* acts = smem + lay.acts;
    float* pens = smem + lay.pens;
    float* xa = smem + lay.xa;
    float* ring = smem + lay.ring;
    float* rstep = smem + lay.rstep;
    float* d2s = smem + lay.d2s;
    const bool normd = m.normalized != 0;
#ifdef BBMPC_KERNEL_DBG
    long long dbg_t[8]; int dbg_i = 0;
#define W4_MARK() do { __builtin_amdgcn_sched_barrier(0); dbg_t[dbg_i++] = (long long)wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define W4_MARK() do {} while (0)
#endif
    W4_MARK();
    const int row = lane >> 4, g = (lane >> 2) & 3, pl = lane & 3;
    const int kh = row >> 1;                              // the K half my block multiplies
    const int qo = 4 * (row & 1) + g;                     // the output quad my block produces (and the input quad my two registers hold)
    const int i0 = row >> 1, i1 = i0 + 2;                 // ... of which I keep features 4*qo + i0 and 4*qo + i1 after the fold
    const int fa = 4 * qo + i0, fb = 4 * qo + i1;
    const int pp = 4 * wave + pl, n = n0 + pp;            // my particle

    // ---- stationary A operands: layer l, rotation j, register c multiplies k = 4*(4*kh + ((g - j) & 3)) + c into output
    // feature 4*qo + (lane & 3) (mlp_w4_operand_index); packed in lane order by bbmpc_set_mlp: one coalesced load each
    // (gathered here from the Dense kernels with their index arithmetic they were 5.5 us of a 40 us launch)
    float wS[NL][16], bS[NL][2];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
#pragma unroll
        for (int op = 0; op < 16; ++op) wS[l][op] = q.w4pack[(l * W4_OPS + op) * 64 + lane];
        bS[l][0] = q.w4pack[(l * W4_OPS + 16) * 64 + lane];
        bS[l][1] = q.w4pack[(l * W4_OPS + 17) * 64 + lane];
    }
    // ---- my two slots of the (state | action) vector: constants of process_input / process_output
    //      (system_dynamics_handler.py:119-122, 152-155; un-normalised: (x - 0) * 1, 0 + z * 1)
    const bool sa = fa < S, sb = fb < S;
    const int ua = fa - 20, ub = fb - 20;                 // action index of the slot (valid when 0 <= u < U)
    const bool aa = ua >= 0 && ua < U, ab = ub >= 0 && ub < U;
    const float nmA = (normd && sa) ? m.mean_s[fa] : 0.0f, nmB = (normd && sb) ? m.mean_s[fb] : 0.0f;
    const float niA = (normd && sa) ? 1.0f / (m.std_s[fa] + 1e-7f) : 1.0f, niB = (normd && sb) ? 1.0f / (m.std_s[fb] + 1e-7f) : 1.0f;
    const float tmA = (normd && sa) ? m.mean_t[fa] : 0.0f, tmB = (normd && sb) ? m.mean_t[fb] : 0.0f;
    const float tsA = (normd && sa) ? (m.std_t[fa] + 1e-7f) : 1.0f, tsB = (normd && sb) ? (m.std_t[fb] + 1e-7f) : 1.0f;
    float curA = 0.0f, curB = 0.0f;
    if (sa) curA = q.per_particle_state ? ((n < p.n_pop) ? p.state[(size_t)n * S + fa] : 0.0f) : p.state[a * S + fa];
    if (sb) curB = q.per_particle_state ? ((n < p.n_pop) ? p.state[(size_t)n * S + fb] : 0.0f) : p.state[a * S + fb];

    // ---- prologue: the 16 particles' action block [H][16][U] (candidate -> clip / penalty -> store), as the other kernels;
    //      then the normalised copies in the slot order of the input vector, [t][particle][8]
    W4_MARK();
    // (mlp_fill_actions' arithmetic with every element clipped by the thread that made it; the squared clip distances are
    // parked and summed per (particle, u) in step order behind the recurrence instead of in front of it)
    for (int e2 = tid; e2 < H * W4_TP * U; e2 += 256) {   // particle fastest: the stores to the particle-minor sample matrix are 64-byte segments
        const int tpp = e2 % W4_TP, j = e2 / W4_TP;
        const int t = j / U, u = j - t * U;
        const int e = (t * W4_TP + tpp) * U + u;
        const int nn = n0 + tpp;
        float x = 0.0f, d2 = 0.0f;
        if (nn < p.n_pop) {
            x = mlp_candidate_value(q, a, nn, j, u);
            if (q.pen) {
                const float xf = clipf(x, p.lo[u], p.hi[u]);
                const float d = x - xf;
                d2 = d * d;
                x = xf;
            }
            if (p.samples) p.samples[((size_t)a * p.HU + j) * p.Nst + nn] = x;
        }
        acts[e] = x;
        d2s[e] = d2;
    }
    __syncthreads();
    W4_MARK();
    {
        const int u = tid & 7;                            // (256 is a multiple of 8: a thread's action slot is the same in every pass)
        const bool on = u < U;
        const float mu = (normd && on) ? m.mean_a[u] : 0.0f, iv = (normd && on) ? 1.0f / (m.std_a[u] + 1e-7f) : 1.0f;
        for (int e = tid; e < H * W4_TP * 8; e += 256)    // e >> 3 = t * 16 + particle
            xa[e] = on ? (acts[(e >> 3) * U + u] - mu) * iv : 0.0f;
    }
    if (sa) ring[pp * Sp + fa] = curA;
    if (sb) ring[pp * Sp + fb] = curB;
    for (int e = tid; e < (H + 1) * W4_TP; e += 256)      // the padding slots f in [S, Sp) of every ring row
        for (int f = S; f < Sp; ++f) ring[e * Sp + f] = 0.0f;
    __syncthreads();

    const float* xaA = xa + pp * 8 + (aa ? ua : 0);       // + t * 128
    const float* xaB = xa + pp * 8 + (ab ? ub : 0);
    float* ringA = ring + pp * Sp + (sa ? fa : 0);        // + (t + 1) * 16 * Sp
    float* ringB = ring + pp * Sp + (sb ? fb : 0);
    // the input vector of step 0 in my two slots
    float v0 = sa ? (curA - nmA) * niA : (aa ? xaA[0] : 0.0f);
    float v1 = sb ? (curB - nmB) * niB : (ab ? xaB[0] : 0.0f);
    W4_MARK();
    for (int t = 0; t < H; ++t) {
        const int tn = (t + 1 < H) ? t + 1 : t;
        const float an0 = xaA[tn * 128], an1 = xaB[tn * 128];      // next step's action slots: static data, a step ahead
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            // ---- B operands: the four features of quad 4*kh + g from my K half's two source rows, then the row's other quads
            float z0 = v0, z1 = v0, z2 = v1, z3 = v1;
            swap16(z0, z1); swap32(z0, z1);               // z0: feature 4q' + 0 (from row kh), z1: 4q' + 1 (from row kh + 2)
            swap16(z2, z3); swap32(z2, z3);               // z2: 4q' + 2, z3: 4q' + 3
            const float r10 = dpp_mov<DPP_ROW_ROR4>(z0), r11 = dpp_mov<DPP_ROW_ROR4>(z1), r12 = dpp_mov<DPP_ROW_ROR4>(z2), r13 = dpp_mov<DPP_ROW_ROR4>(z3);
            const float r20 = dpp_mov<DPP_ROW_ROR8>(z0), r21 = dpp_mov<DPP_ROW_ROR8>(z1), r22 = dpp_mov<DPP_ROW_ROR8>(z2), r23 = dpp_mov<DPP_ROW_ROR8>(z3);
            const float r30 = dpp_mov<DPP_ROW_ROR12>(z0), r31 = dpp_mov<DPP_ROW_ROR12>(z1), r32 = dpp_mov<DPP_ROW_ROR12>(z2), r33 = dpp_mov<DPP_ROW_ROR12>(z3);
            __builtin_amdgcn_sched_barrier(0);
            mfma4_operands_settled();
            // two accumulator chains, alternating (a dependent 4x4x1 needs its producer 12 cycles back: two issues)
            f32x4 ce, co;
            mfma4_v_0(ce, wS[l][0], z0);   mfma4_v_0(co, wS[l][1], z1);   mfma4_v(ce, wS[l][2], z2);    mfma4_v(co, wS[l][3], z3);
            mfma4_v(ce, wS[l][4], r10);    mfma4_v(co, wS[l][5], r11);    mfma4_v(ce, wS[l][6], r12);   mfma4_v(co, wS[l][7], r13);
            mfma4_v(ce, wS[l][8], r20);    mfma4_v(co, wS[l][9], r21);    mfma4_v(ce, wS[l][10], r22);  mfma4_v(co, wS[l][11], r23);
            mfma4_v(ce, wS[l][12], r30);   mfma4_v(co, wS[l][13], r31);   mfma4_v(ce, wS[l][14], r32);  mfma4_v(co, wS[l][15], r33);
            asm volatile("s_nop 4" : "+v"(ce), "+v"(co));
            __builtin_amdgcn_sched_barrier(0);
            // ---- fold the two K halves (rows r and r ^ 2): a reduce-scatter, two registers per swap
            float d0 = ce.x + co.x, d1 = ce.y + co.y, d2 = ce.z + co.z, d3 = ce.w + co.w;
            swap32(d0, d1);
            swap32(d2, d3);
            float s0 = (d0 + d1) + bS[l][0], s1 = (d2 + d3) + bS[l][1];
            if (l + 1 < NL) {
                const int act = TANH ? ACT_TANH : m.act[l];
                if (TANH) { v0 = bb_tanhf(s0); v1 = bb_tanhf(s1); }
                else { v0 = apply_act(s0, act); v1 = apply_act(s1, act); }
            } else {
                // ---- epilogue on my two slots: last activation, de-normalise, residual (system_dynamics_handler.py:152-155,
                //      transforms.py:34), the ring, the next step's input (process_input)
                if (!TANH) { s0 = apply_act(s0, m.act[NL - 1]); s1 = apply_act(s1, m.act[NL - 1]); }
                curA = sa ? (tmA + s0 * tsA) + curA : 0.0f;
                curB = sb ? (tmB + s1 * tsB) + curB : 0.0f;
                if (sa) ringA[(t + 1) * W4_TP * Sp] = curA;
                if (sb) ringB[(t + 1) * W4_TP * Sp] = curB;
                v0 = sa ? (curA - nmA) * niA : (aa ? an0 : 0.0f);
                v1 = sb ? (curB - nmB) * niB : (ab ? an1 : 0.0f);
            }
        }
    }
    W4_MARK();
    __syncthreads();
    W4_MARK();
    // ---- step rewards by all threads from the ring (deterministic.py:62-73), summed in step order below
    for (int e = tid; e < H * W4_TP; e += 256)
        rstep[e] = reward_generic(p.reward_kind, p.fix_q1 != 0, ring + (size_t)e * Sp, acts + e * U, ring + (size_t)(e + W4_TP) * Sp, S, U);
    if (tid >= 128 && tid < 128 + W4_TP * U) {            // clip penalties (wave 2 and up, next to the step rewards)
        const int tpp = (tid - 128) / U, u = (tid - 128) % U;
        float pen_part = 0.0f;
        if (q.pen && n0 + tpp < p.n_pop)
            for (int t = 0; t < H; ++t) pen_part = pen_part + d2s[(t * W4_TP + tpp) * U + u];
        pens[tpp * U + u] = pen_part;
    }
    if (q.traj) {                                         // a user reward function scores the recorded trajectory afterwards
        for (int e = tid; e < H * W4_TP * S; e += 256) {
            const int f = e % S, tp = e / S, tpp = tp % W4_TP, tt = tp / W4_TP;
            if (n0 + tpp < p.n_pop) q.traj[((((size_t)tt * p.A + a) * p.Nst) + n0 + tpp) * S + f] = ring[(size_t)(tp + W4_TP) * Sp + f];
        }
    }
    if (q.final_state) {
        for (int e = tid; e < W4_TP * S; e += 256) {
            const int f = e % S, tpp = e / S;
            if (n0 + tpp < p.n_pop) q.final_state[((size_t)a * p.n_pop + n0 + tpp) * S + f] = ring[((size_t)H * W4_TP + tpp) * Sp + f];
        }
    }
    __syncthreads();
    if (tid < W4_TP && n0 + tid < p.n_pop) {
        float tot = 0.0f;
        for (int t = 0; t < H; ++t) tot = tot + rstep[t * W4_TP + tid];
        if (tot != tot) tot = -1.0e6f;                                  // deterministic.py:75-77
        if (q.pen) {
            float pen = 0.0f;
            for (int u = 0; u < U; ++u) pen = pen + pens[tid * U + u];
            const float nr = sqrtf(pen);
            pen = nr * nr;
            tot = tot - pen;
            if (p.penalty_out) p.penalty_out[(size_t)a * p.Nst + n0 + tid] = pen;
        }
        p.rewards[(size_t)a * p.Nst + n0 + tid] = tot;
    }
    W4_MARK();
#ifdef BBMPC_KERNEL_DBG
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        printf("[w4dbg] H=%d (10 ns units) constants+weights issued %lld | action block %lld | xa+ring %lld | loop %lld | wait for the other waves %lld | rewards+sums %lld\n",
               H, dbg_t[1] - dbg_t[0], dbg_t[2] - dbg_t[1], dbg_t[3] - dbg_t[2], dbg_t[4] - dbg_t[3], dbg_t[5] - dbg_t[4], dbg_t[6] - dbg_t[5]);
#endif
#undef W4_MARK
}

}  // namespace bbmpc
