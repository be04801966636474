// Learned-dynamics rollout for SMALL networks (hidden width <= 64, e.g. the 4-32-32-32-3 Pendulum models of the
// reference's learn_dynamics / model_based_RL tutorials): HT waves (one per 16-feature hidden tile) own a 16-particle
// tile for the whole H-step recurrence, with the recurrence itself -- raw state, normalisation, residual, next input --
// replicated in every wave's registers.
//
// With the transposed evaluation of kernels_mlp.hpp (out^T = W^T x^T on v_mfma_f32_16x16x4_f32) the D fragment of a
// 16-feature output tile IS the B operand of the next layer's K tile (lane l holds features 4*(l>>4)+{0..3} of particle
// l&15, and MFMA number s of a K tile takes "feature 4*(l>>4)+s" as its k index).  The same holds across planning
// steps: output feature f of the last layer lands in the lane/register slot that input feature f of the next step is
// read from.  So the general kernel's per-step epilogue over LDS (reduce the K-split partial sums, bias, de-normalise,
// add the state, normalise, re-tile: two barriers and ~16*(S+U) scattered LDS accesses) collapses to a handful of
// register operations once every wave has the summed last-layer output: each wave keeps the raw state of "its" slots
// and stages the next input itself, redundantly.  Per step and wave: one ds_write_b128 + barrier + HT ds_read_b128 per
// hidden-layer all-gather, one such exchange for the last layer's K-split partial sums (summed in wave order by every
// wave: identical values everywhere), nothing else through LDS.  The reward needs several features of one particle,
// which are spread over four lanes: the wave whose turn it is (t mod HT) copies (state, next state) to a private LDS
// scratch and lets 16 lanes evaluate it -- or evaluates it straight from registers when S+U <= 4 (Pendulum: one lane
// holds cos, sin, thdot and the torque).
//
// A single wave per tile with the whole Dense stack in its registers (no exchange at all) was measured first and is
// SLOWER than the general kernel (2.9 vs 2.5 us per step of a 4-32-32-32-3 network): a lone wave issues one VALU
// instruction per ~5 cycles and its fp32 MFMAs do not overlap them, so a step is the SUM of 64 x 32 MFMA cycles and
// ~700 x 5 VALU cycles.  Splitting the tiles over HT waves on HT SIMDs halves both.
//
// Same semantics as rollout_mlp_body (process_input -> Dense stack -> process_output -> reward, deterministic.py:26-77);
// the K-split partial sums are added in wave order starting from the bias, as there.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_mlp.hpp"

namespace bbmpc {

// LDS, in floats: acts [H][16][U] | pens | xch [NH-1][HT][64][4] hidden all-gathers | part [HT][OTLM][64][4] |
//                 (double buffered) | ring [H+1][16][Sp] the raw state before every step and after the last | rstep [H][16]
struct MlpWaveLds {
    int acts, pens, xch, part, ring, rstep, total;
};
__host__ __device__ inline MlpWaveLds mlp_wave_lds_layout(int H, int U, int S, int NH, int HT) {
    MlpWaveLds l;
    const int Sp = (S + 3) & ~3;
    int o = 0;
    l.acts = o; o += ((H * MLP_TP * U + 3) & ~3);
    l.pens = o; o += ((MLP_TP * U + 63) & ~63);
    l.xch = o;  o += (NH > 1 ? NH - 1 : 1) * HT * 256;
    l.part = o; o += 2 * HT * 2 * 256;
    l.ring = o; o += (H + 1) * MLP_TP * Sp;
    l.rstep = o; o += H * MLP_TP;
    l.total = o;
    return l;
}

__device__ __forceinline__ f32x4 act4(f32x4 v, int a) {
    if (a == ACT_TANH) { v.x = bb_tanhf(v.x); v.y = bb_tanhf(v.y); v.z = bb_tanhf(v.z); v.w = bb_tanhf(v.w); }
    else if (a != ACT_NONE) { v.x = apply_act(v.x, a); v.y = apply_act(v.y, a); v.z = apply_act(v.z, a); v.w = apply_act(v.w, a); }
    return v;
}

// waves per workgroup: HT compute waves + one wave that only scores the steps
__host__ __device__ constexpr int mlp_wave_waves(int HT) { return HT + 1; }

// NH hidden layers of exactly HT 16-feature tiles each; S+U <= 32, S <= 32.  grid (ceil(n_pop/16), A), block 64*waves.
// TANH: every hidden activation is tanh and the output layer is linear (the tutorials' networks): no run-time dispatch
template <int NH, int HT, bool TANH>
__global__ __launch_bounds__(64 * mlp_wave_waves(HT)) void k_rollout_mlp_wave(MlpRolloutArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    const int a = blockIdx.y;
    const int n0 = blockIdx.x * MLP_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pp = lane & 15;
    const int S = p.S, U = p.U, H = p.H;
    const int Sp = (S + 3) & ~3;
    const MlpWaveLds lay = mlp_wave_lds_layout(H, U, S, NH, HT);
    float* acts = smem + lay.acts;
    float* pens = smem + lay.pens;
    float* xch = smem + lay.xch;
    float* part = smem + lay.part;
    float* ring = smem + lay.ring;
    float* rstep = smem + lay.rstep;
    const bool normd = m.normalized != 0;
    constexpr int IT0M = 2, OTLM = 2, L = NH + 1;
    const int IT0 = m.tiles[0], OTL = m.tiles[L];
    const int n = n0 + pp;

    // ---- my A slabs: output tile `wave` of every hidden layer, K slab `wave` of the last layer (the reward wave, which
    //      never uses them, reads wave HT-1's instead of running past the arrays)
    const int cw = wave < HT ? wave : HT - 1;
    float w_in[IT0M][4];
    float w_hid[NH > 1 ? NH - 1 : 1][HT][4];
    float w_out[OTLM][4];
    f32x4 b_hid[NH], b_out[OTLM];
#pragma unroll
    for (int it = 0; it < IT0M; ++it)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            w_in[it][s] = (it < IT0) ? m.wpack[0][(((size_t)cw * IT0 + it) * 4 + s) * 64 + lane] : 0.0f;
#pragma unroll
    for (int h = 1; h < NH; ++h)
#pragma unroll
        for (int it = 0; it < HT; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                w_hid[h - 1][it][s] = m.wpack[h][(((size_t)cw * HT + it) * 4 + s) * 64 + lane];
#pragma unroll
    for (int ot = 0; ot < OTLM; ++ot)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            w_out[ot][s] = (ot < OTL) ? m.wpack[NH][(((size_t)ot * HT + cw) * 4 + s) * 64 + lane] : 0.0f;
#pragma unroll
    for (int h = 0; h < NH; ++h) b_hid[h] = *reinterpret_cast<const f32x4*>(m.bpack[h] + ((size_t)cw * 64 + lane) * 4);
#pragma unroll
    for (int ot = 0; ot < OTLM; ++ot)
        b_out[ot] = (ot < OTL) ? *reinterpret_cast<const f32x4*>(m.bpack[NH] + ((size_t)ot * 64 + lane) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- my input slots: tile it, register r <-> feature f = 16*it + 4*g + r; state (f < S), action (S <= f < S+U), padding
    float nmean[IT0M][4], ninv[IT0M][4], tmean[IT0M][4], tstd[IT0M][4], sraw[IT0M][4], xin[IT0M][4];
    int aidx[IT0M][4];                       // action index u of the slot or -1
#pragma unroll
    for (int it = 0; it < IT0M; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * it + 4 * g + r;
            const bool is_s = it < IT0 && f < S, is_a = it < IT0 && f >= S && f < S + U;
            const float mu = !normd ? 0.0f : (is_s ? m.mean_s[f] : (is_a ? m.mean_a[f - S] : 0.0f));
            const float sd = !normd ? 1.0f : (is_s ? m.std_s[f] : (is_a ? m.std_a[f - S] : 1.0f));
            nmean[it][r] = mu;
            ninv[it][r] = (normd && (is_s || is_a)) ? 1.0f / (sd + 1e-7f) : ((is_s || is_a) ? 1.0f : 0.0f);   // system_dynamics_handler.py:119-122
            tmean[it][r] = (normd && is_s) ? m.mean_t[f] : 0.0f;
            tstd[it][r] = (normd && is_s) ? (m.std_t[f] + 1e-7f) : 1.0f;
            aidx[it][r] = is_a ? f - S : -1;
            float v = 0.0f;
            if (is_s) v = q.per_particle_state ? ((n < p.n_pop) ? p.state[(size_t)n * S + f] : 0.0f) : p.state[a * S + f];
            sraw[it][r] = v;
        }

    // ---- prologue: the tile's action block [H][16][U] (candidate -> clip/penalty -> store), as the general kernels
    mlp_fill_actions<MLP_TP>(q, a, n0, tid, 64 * mlp_wave_waves(HT), acts, pens);
    // the raw state of every step goes to the ring (wave 0 writes it); the reward reads (state t, action t, state t+1) there
    auto ring_store = [&](int t) {            // one 16-byte store per tile: the padding slots f in [S, Sp) carry zeros
#pragma unroll
        for (int it = 0; it < OTLM; ++it)
            if (it < OTL && 16 * it + 4 * g < Sp)
                *reinterpret_cast<f32x4*>(ring + ((size_t)t * MLP_TP + pp) * Sp + 16 * it + 4 * g) =
                    f32x4{sraw[it][0], sraw[it][1], sraw[it][2], sraw[it][3]};
    };
    auto score = [&](int t) {                 // lanes 0..15: reward of step t for particle `lane`
        if (lane < MLP_TP)
            rstep[t * MLP_TP + lane] = reward_generic(p.reward_kind, p.fix_q1 != 0, ring + ((size_t)t * MLP_TP + lane) * Sp,
                                                      acts + (t * MLP_TP + lane) * U, ring + ((size_t)(t + 1) * MLP_TP + lane) * Sp, S, U);
    };
    if (wave == 0) ring_store(0);
    __syncthreads();
    // ---- the reward wave: the pendulum reward (atan2 + floor-mod, ~0.7 us of a lone wave's time per step) leaves the
    // recurrence's critical path.  It keeps the compute waves' barrier count (NBAR per step) and scores step t-1 after the
    // first barrier of step t, which orders wave 0's ring_store(t) before it.
    constexpr int NBAR = HT > 1 ? NH : 1;     // HT > 1: NH - 1 all-gathers + the K-split exchange; HT == 1: one, for this wave
    if (wave == HT) {
        for (int t = 0; t < H; ++t) {
#pragma unroll
            for (int b = 0; b < NBAR; ++b) {
                __syncthreads();
                if (b == 0 && t > 0) score(t - 1);
            }
        }
        __syncthreads();
        score(H - 1);
        __syncthreads();
        return;
    }
    float anext[IT0M][4];                    // my action slots' values for the coming step, fetched a step ahead
    auto fetch_actions = [&](int t) {
#pragma unroll
        for (int it = 0; it < IT0M; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {      // branch-free: slots that are no action read action 0 and drop it
                const float v = acts[(t * MLP_TP + pp) * U + max(aidx[it][r], 0)];
                anext[it][r] = (aidx[it][r] >= 0) ? v : 0.0f;
            }
    };
    auto stage_input = [&]() {
#pragma unroll
        for (int it = 0; it < IT0M; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (aidx[it][r] >= 0) ? anext[it][r] : sraw[it][r];
                xin[it][r] = (v - nmean[it][r]) * ninv[it][r];
            }
    };
    fetch_actions(0);
    stage_input();

    for (int t = 0; t < H; ++t) {
        fetch_actions(t + 1 < H ? t + 1 : t);     // LDS latency hidden behind this step's layers
        // ---- layer 0: my output tile
        f32x4 acc = b_hid[0];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[0][s], xin[0][s], acc, 0, 0, 0);
        if (IT0 > 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w_in[1][s], xin[1][s], acc, 0, 0, 0);
        }
        acc = act4(acc, TANH ? ACT_TANH : m.act[0]);
        // ---- hidden -> hidden: all-gather the HT tiles through LDS (a buffer per layer: reused a whole step later)
#pragma unroll
        for (int h = 1; h < NH; ++h) {
            float* buf = xch + (size_t)(h - 1) * HT * 256;
            if constexpr (HT > 1) {
                *reinterpret_cast<f32x4*>(buf + ((size_t)wave * 64 + lane) * 4) = acc;
                __syncthreads();
            }
            f32x4 nx = b_hid[h];
#pragma unroll
            for (int it = 0; it < HT; ++it) {
                f32x4 b;
                if constexpr (HT > 1) b = *reinterpret_cast<const f32x4*>(buf + ((size_t)it * 64 + lane) * 4);
                else b = acc;
#pragma unroll
                for (int s = 0; s < 4; ++s) nx = __builtin_amdgcn_mfma_f32_16x16x4f32(w_hid[h - 1][it][s], b[s], nx, 0, 0, 0);
            }
            acc = act4(nx, TANH ? ACT_TANH : m.act[h]);
        }
        // ---- last layer, K split: my last-hidden tile (still in `acc`) times my slab of W_last
        f32x4 o[OTLM];
        {
            f32x4 po[OTLM];
#pragma unroll
            for (int ot = 0; ot < OTLM; ++ot) po[ot] = (HT > 1) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : b_out[ot];
#pragma unroll
            for (int s = 0; s < 4; ++s) po[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_out[0][s], acc[s], po[0], 0, 0, 0);
            if (OTL > 1) {
#pragma unroll
                for (int s = 0; s < 4; ++s) po[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_out[1][s], acc[s], po[1], 0, 0, 0);
            }
            if constexpr (HT > 1) {
                float* pb = part + (size_t)(t & 1) * HT * OTLM * 256;   // double buffered: with one hidden layer this barrier is the only one of a step
                *reinterpret_cast<f32x4*>(pb + (((size_t)wave * OTLM + 0) * 64 + lane) * 4) = po[0];
                if (OTL > 1) *reinterpret_cast<f32x4*>(pb + (((size_t)wave * OTLM + 1) * 64 + lane) * 4) = po[1];
                __syncthreads();
#pragma unroll
                for (int ot = 0; ot < OTLM; ++ot) {
                    o[ot] = b_out[ot];
                    if (ot < OTL) {
#pragma unroll
                        for (int w = 0; w < HT; ++w) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(pb + (((size_t)w * OTLM + ot) * 64 + lane) * 4);
                            o[ot].x = o[ot].x + v.x; o[ot].y = o[ot].y + v.y; o[ot].z = o[ot].z + v.z; o[ot].w = o[ot].w + v.w;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int ot = 0; ot < OTLM; ++ot) o[ot] = po[ot];
            }
        }
        // ---- epilogue in registers, every wave: last activation, de-normalise, residual
        //      (system_dynamics_handler.py:152-155, transforms.py:34)
#pragma unroll
        for (int it = 0; it < OTLM; ++it) {
            if (it < OTL) {
                const f32x4 ov = act4(o[it], TANH ? ACT_NONE : m.act[L - 1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * it + 4 * g + r;
                    const float dev = tmean[it][r] + ov[r] * tstd[it][r];     // (0, 1) when not normalised
                    const float ns = dev + sraw[it][r];
                    sraw[it][r] = (f < S) ? ns : 0.0f;                       // a select, no branch
                }
            }
        }
        if (q.traj && wave == 0 && n < p.n_pop) {     // a user reward function scores the recorded trajectory afterwards
#pragma unroll
            for (int it = 0; it < OTLM; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * it + 4 * g + r;
                    if (it < OTL && f < S) q.traj[((((size_t)t * p.A + a) * p.Nst) + n) * S + f] = sraw[it][r];
                }
        }
        if (wave == 0) ring_store(t + 1);
        stage_input();
        if constexpr (HT == 1) __syncthreads();   // hands ring_store(t + 1) to the reward wave
    }
    __syncthreads();                          // the reward wave scores the last step after this one
    // ---- total reward: the step rewards summed in step order, as the evaluator's loop does (deterministic.py:62-73)
    __syncthreads();
    if (tid < MLP_TP && n < p.n_pop) {
        float tot = 0.0f;
        for (int t = 0; t < H; ++t) tot = tot + rstep[t * MLP_TP + tid];
        if (tot != tot) tot = -1.0e6f;                                  // deterministic.py:75-77
        if (q.pen) {
            float pen = 0.0f;
            for (int u = 0; u < U; ++u) pen = pen + pens[tid * U + u];
            const float nr = sqrtf(pen);
            pen = nr * nr;
            tot = tot - pen;
            if (p.penalty_out) p.penalty_out[(size_t)a * p.Nst + n] = pen;
        }
        p.rewards[(size_t)a * p.Nst + n] = tot;
    }
    if (q.final_state && wave == 0 && n < p.n_pop) {
#pragma unroll
        for (int it = 0; it < IT0M; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * it + 4 * g + r;
                if (it < IT0 && f < S) q.final_state[((size_t)a * p.n_pop + n) * S + f] = sraw[it][r];
            }
    }
}

}  // namespace bbmpc
