// Learned-dynamics rollout on the matrix cores (fp32-in / fp32-accumulate MFMA, exact fp32 products).
//
// Fuses, per candidate trajectory and planning step (reference files under blackbox_mpc/):
//   SystemDynamicsHandler.process_input   dynamics_handlers/system_dynamics_handler.py:97-126  (z-score + concat)
//   DeterministicMLP.__call__             dynamics_functions/deterministic_mlp.py:27-51        (Dense stack)
//   SystemDynamicsHandler.process_output  :128-161 + utils/transforms.py:20-34                 (de-normalise + s + delta)
//   reward_function                       tutorials/mujoco/cost_func.py:5-22 | utils/pendulum.py:10-35
//   DeterministicTrajectoryEvaluator.__call__  trajectory_evaluators/deterministic.py:26-77    (H-step loop, NaN guard)
//
// Mapping.  A workgroup owns a tile of 16 particles (MFMA N dimension) of one agent for the whole
// H-step recurrence.  The layer is evaluated transposed, out^T[M x 16] = W^T[M x K] . x^T[K x 16], with
// v_mfma_f32_16x16x4_f32: A = a 16x4 slab of W^T, B = 4 input features x 16 particles, D = 16 output
// features x 16 particles.  The D fragment of lane l holds features 4*(l>>4)+{0..3} of particle l&15
// -- exactly the four B operands that lane needs for the next layer when the k index inside an MFMA is
// taken as "feature 4*(l>>4)+s".  So a layer's output is written to LDS as one float4 per lane per
// 16-feature tile and every wave of the next layer reads its B operands back with one ds_read_b128
// per tile: no transposes, no shuffles.  Output tiles of a layer are dealt round-robin to the waves;
// the last (narrow) layer is split along K instead (each wave multiplies the tiles it owns into all
// output tiles) and the partial sums meet in LDS, where the epilogue (bias, de-normalise, residual,
// next-step normalisation, reward) runs.  Weights are pre-packed in operand order
// (wpack[layer][ot][it][s][lane]) so every A operand is one coalesced 256-byte load.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_refit.hpp"
#include "kernels_rollout.hpp"
#include "models.hpp"
#include "rng.hpp"

namespace bbmpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_MAX_LAYERS = 8;      // Dense layers (the generic kernel walks any number; the specialisations cover 2-4)
constexpr int MLP_TP = 16;          // particles per workgroup tile
constexpr int MLP_TMAX = 2;         // output tiles per wave per layer

constexpr int ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2, ACT_SIGMOID = 3;

struct MlpDesc {
    int n_layers;
    int dims[MLP_MAX_LAYERS + 1];        // dims[0] = S+U ... dims[L] = S
    int tiles[MLP_MAX_LAYERS + 1];       // ceil(dims/16)
    int act[MLP_MAX_LAYERS];
    const float* wpack[MLP_MAX_LAYERS];  // [OT][IT][4][64]
    const float* bpack[MLP_MAX_LAYERS];  // [OT][64][4]
    int half_tail[MLP_MAX_LAYERS + 1];   // hidden layer whose last 16-feature tile holds <= 8 features: they sit in the
                                         // slots 4g+{0,1} (set_mlp), so MFMAs 2 and 3 of that K tile multiply zeros
    int normalized;
    const float* mean_s;                 // [S]
    const float* std_s;                  // [S]
    const float* mean_a;                 // [U]
    const float* std_a;                  // [U]
    const float* mean_t;                 // [S]
    const float* std_t;                  // [S]
};

struct MlpRolloutArgs {
    RolloutArgs r;            // shared fields (n_pop, dims, sources, outputs, rng)
    MlpDesc m;
    int mode;                 // SRC_*
    int pen;                  // clip + penalty (PI2 / PSO / CMA-ES / SPSA candidates)
    int per_particle_state;   // state is [n_pop, S] (single-step API) instead of [A, S]
    float* final_state;       // optional [A? n_pop][S] written after the last step (per_particle_state layout)
    int nw;                   // waves per workgroup
    const float* wraw[MLP_MAX_LAYERS];   // unpacked Dense kernels [in][out] (quad-mode kernel)
    const float* braw[MLP_MAX_LAYERS];   // unpacked biases [out]
    const float* wq4[MLP_MAX_LAYERS];    // quad-mode operands [ceil(in/4)][Mp][4], Mp = out rounded up to 64, zero padded
    const float* w4pack;                 // k_rollout_mlp_w4's operands in lane order: [layer][16 operands + 2 biases][64 lanes] (bbmpc_set_mlp)
    const float* wq4s0;                  // layer 0 in quad-mode order with the STATE rows padded to 20: input k < dim_S at row k, action u at row 20 + u (k_rollout_mlp_q4s)
    const float* wp4[MLP_MAX_LAYERS];    // the operands of MlpDesc::wpack as [OT][IT][64 lanes][4]: a lane's four A operands of a k tile in ONE 16-byte load (generic kernel)
    const uint4* wbf[MLP_MAX_LAYERS];    // bf16 mode operands [OT][IT][64] x (4 bf16 hi | 4 bf16 lo), k = 16*it + 4*(lane>>4) + r
    float* traj;              // optional [H][A][Nst][S]: the state after every step (a user reward function scores them afterwards)
    float* state_copy;        // optional [A][S] (k_rollout_mlp_q4s): r.state is the pinned host buffer of this control step, workgroup 0 of
                              // an agent's row stores the state here for the later launches and the tail
};

// tanh on the hardware exp/rcp units: sign(x) * (1 - 2 / (2^{c|x|} + 1)), c = 2 log2(e): six instructions
// (v_mul with |x|, v_exp_f32, v_add, v_rcp_f32, v_fma, v_bfi).  Absolute error <= ~2.5e-7 over the whole range (v_exp_f32 /
// v_rcp_f32 are ~1 ulp); near zero the RELATIVE error grows (cancellation) but an activation feeds a dot product, where
// only absolute error matters -- it is the size of one fp32 rounding of an O(1) pre-activation.  fp32-input MFMA
// executes at the vector rate on the same datapath as VALU work (measured: step time = MFMA time + VALU time, not the
// max), so every VALU instruction shaved off the activations is matrix time gained: this form replaced
// exp(2|x|) -> 1 - 2r spelled as (|x|+|x|) * log2e, exp2, +1, rcp, r+r, 1-  (eight instructions).
__device__ __forceinline__ float bb_tanhf(float x) {
    // tanh x = 1 - 2 / (1 + e^(2x)) holds for either sign: +inf for large x -> 1 - 0, 0 for large -x -> 1 - 2; round 5 dropped the
    // |x| / copysign pair around it (one v_bfi per value: five instructions instead of six, same absolute error bound -- the
    // reciprocal's argument lies in [1, 2) for x < 0 and the cancellation near zero is the positive side's mirrored).
    const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);
    // v_rcp_f32 (1 ulp).  __frcp_rn is the correctly rounded reciprocal, i.e. a full IEEE division: ten instructions
    // (v_div_scale x2, v_rcp, four fmas, v_div_fmas, v_div_fixup) per activation value, on the MFMAs' issue port.
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);   // NaN stays NaN (exp2(NaN) = NaN)
}

__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == ACT_TANH) return bb_tanhf(x);
    if (act == ACT_RELU) return fmaxf(x, 0.0f);
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-x));
    return x;
}

// LDS carve, in floats (all pieces multiples of 4 floats = 16 B):
//   xs   [IT0][64][4]        normalised layer-0 input tiles
//   actA [ITmax][64][4]      ping
//   actB [ITmax][64][4]      pong
//   part [NW][OTlast][64][4] K-split partial sums of the last layer
//   st   [2][16][Sp]         raw state, double buffered (Sp = S rounded up to 4)
//   acts [H][16][Up]         raw (feasible) actions of the tile (Up = U rounded up to 4... kept U)
//   misc [16*U]             per-(particle,u) penalty shares
//   norm [..]               per-feature (mean, 1/(std+1e-7)) of the inputs, (mean_t, std_t+1e-7, last bias) of the outputs
struct MlpLds {
    int xs, actA, actB, part, st, acts, misc, norm, total;
};
__host__ __device__ inline MlpLds mlp_lds_layout(const MlpDesc& m, int H, int U, int S, int nw) {
    MlpLds l;
    int itmax = 1;
    for (int i = 1; i < m.n_layers; ++i) itmax = itmax > m.tiles[i] ? itmax : m.tiles[i];
    const int Sp = (S + 3) & ~3;
    int o = 0;
    l.xs = o;   o += m.tiles[0] * 256;
    l.actA = o; o += itmax * 256;
    l.actB = o; o += itmax * 256;
    l.part = o; o += nw * m.tiles[m.n_layers] * 256;
    l.st = o;   o += 2 * MLP_TP * Sp;
    l.acts = o; o += ((H * MLP_TP * U + 3) & ~3);
    l.misc = o; o += ((MLP_TP * U + 63) & ~63);
    l.norm = o; o += (((S + U) * 2 + S * 3 + 63) & ~63);
    l.total = o;
    return l;
}

// One dense layer, output-tile split.  in: LDS tiles [IT][64] float4; out: LDS tiles [OT][64] float4.
// Operands are streamed from L2 every model step (the generic kernel keeps nothing stationary), so the layer runs at the
// rate the loads are in flight: round 5 reads them as ONE 16-byte load per lane and k tile from the [OT][IT][lane][4] copy
// (four 4-byte loads 256 bytes apart before: 32 KB in flight per CU, 31 GB/s per CU at the reference's 26-500-500-500-20
// network, a third of what the matrix pipe can take), takes a wave's output tiles two at a time -- one LDS read of the input
// tile feeds two independent accumulator chains -- and unrolls four k tiles, i.e. eight 1-KB loads in flight per wave.
#ifndef MLP_GEN_PF
#define MLP_GEN_PF 2
#endif
__device__ __forceinline__ void mlp_layer_out_split(const MlpDesc& m, const float* wp4, int l, int in_off, int out_off, int wave,
                                                    int lane, int nw) {
    // The activation tiles are addressed as OFFSETS into the dynamic LDS array (round 6): through the `float*` the layer loop
    // swaps between two buffers the compiler loses the address space and reads them with flat_load -- which counts on the
    // vector memory counter as well, so every k tile waited (s_waitcnt vmcnt(0)) for the whole operand ring it had just
    // refilled: 778-788 -> 705-707 us per 4048 x 15-step launch of the 26-500-500-500-20 network.  With the ring working,
    // depth 2 / 3 / 4: 706 / 734 / 747 us; six k tiles of a hidden layer stationary in the spare registers on top: 719 (it
    // had bought 1-2 % while the flat loads were there; removed).
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const float* in = smem + in_off;
    float* out = smem + out_off;
    const int IT = m.tiles[l], OT = m.tiles[l + 1];
    const f32x4* __restrict__ W = reinterpret_cast<const f32x4*>(wp4);
    const float* __restrict__ bp = m.bpack[l];
    const int a = m.act[l];
    for (int ot0 = wave; ot0 < OT; ot0 += 2 * nw) {
        const int ot1 = ot0 + nw;
        if (ot1 < OT) {
            f32x4 acc0 = *reinterpret_cast<const f32x4*>(bp + ((size_t)ot0 * 64 + lane) * 4);   // bias enters as C
            f32x4 acc1 = *reinterpret_cast<const f32x4*>(bp + ((size_t)ot1 * 64 + lane) * 4);
            const f32x4* w0 = W + (size_t)ot0 * IT * 64 + lane;
            const f32x4* w1 = W + (size_t)ot1 * IT * 64 + lane;
            // a ring of MLP_GEN_PF k tiles of operands in registers: 2 x MLP_GEN_PF 1-KB loads in flight per wave at all times
            // (left to `#pragma unroll` the compiler kept the loads next to their use: 887 us per 4048 x 15-step launch of the
            // 26-500-500-500-20 network against 1007 before; ring depth 1 / 2 / 3 / 4 / 6: 839 / 779 / 810 / 830 / 818 us.  Deeper
            // rings did not pay -- the flat loads above, not the L2s: see the top of the function.)
            f32x4 r0[MLP_GEN_PF], r1[MLP_GEN_PF];
#pragma unroll
            for (int j = 0; j < MLP_GEN_PF; ++j) {
                const int kk = j < IT ? j : IT - 1;
                r0[j] = w0[(size_t)kk * 64];
                r1[j] = w1[(size_t)kk * 64];
            }
            for (int it = 0; it < IT; it += MLP_GEN_PF) {
#pragma unroll
                for (int j = 0; j < MLP_GEN_PF; ++j) {
                    const int k = it + j;
                    if (k < IT) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(in + ((size_t)k * 64 + lane) * 4);
                        const f32x4 a0 = r0[j], a1 = r1[j];
                        const int kn = k + MLP_GEN_PF < IT ? k + MLP_GEN_PF : IT - 1;      // (the last ring refills re-read the last tile: no bounds branch in the stream)
                        r0[j] = w0[(size_t)kn * 64];
                        r1[j] = w1[(size_t)kn * 64];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0);
                    }
                }
            }
            acc0.x = apply_act(acc0.x, a); acc0.y = apply_act(acc0.y, a); acc0.z = apply_act(acc0.z, a); acc0.w = apply_act(acc0.w, a);
            acc1.x = apply_act(acc1.x, a); acc1.y = apply_act(acc1.y, a); acc1.z = apply_act(acc1.z, a); acc1.w = apply_act(acc1.w, a);
            *reinterpret_cast<f32x4*>(out + ((size_t)ot0 * 64 + lane) * 4) = acc0;
            *reinterpret_cast<f32x4*>(out + ((size_t)ot1 * 64 + lane) * 4) = acc1;
        } else {
            f32x4 acc = *reinterpret_cast<const f32x4*>(bp + ((size_t)ot0 * 64 + lane) * 4);
            const f32x4* w0 = W + (size_t)ot0 * IT * 64 + lane;
#pragma unroll 4
            for (int it = 0; it < IT; ++it) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(in + ((size_t)it * 64 + lane) * 4);
                const f32x4 a0 = w0[(size_t)it * 64];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc, 0, 0, 0);
            }
            acc.x = apply_act(acc.x, a); acc.y = apply_act(acc.y, a); acc.z = apply_act(acc.z, a); acc.w = apply_act(acc.w, a);
            *reinterpret_cast<f32x4*>(out + ((size_t)ot0 * 64 + lane) * 4) = acc;
        }
    }
}

// Last layer, K split: wave multiplies the input tiles it owns (it = wave, wave+nw, ...) into every
// output tile and leaves partial sums in part[wave][ot][lane].
__device__ __forceinline__ void mlp_layer_k_split(const MlpDesc& m, const float* wp4, int l, int in_off, int part_off, int wave,
                                                  int lane, int nw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];        // (offsets, not pointers: see mlp_layer_out_split)
    const float* in = smem + in_off;
    float* part = smem + part_off;
    const int IT = m.tiles[l], OT = m.tiles[l + 1];
    const f32x4* __restrict__ W = reinterpret_cast<const f32x4*>(wp4);
    for (int ot = 0; ot < OT; ++ot) {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = wave; it < IT; it += nw) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(in + ((size_t)it * 64 + lane) * 4);
            const f32x4 w = W[((size_t)ot * IT + it) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b.w, acc, 0, 0, 0);
        }
        *reinterpret_cast<f32x4*>(part + (((size_t)wave * OT + ot) * 64 + lane) * 4) = acc;
    }
}

// address (in floats) of feature f of particle p inside a tile array [T][64][4]
__device__ __forceinline__ int tile_addr(int f, int p) {
    return (((f >> 4) * 64) + (((f & 15) >> 2) * 16 + p)) * 4 + (f & 3);
}

// SPEC selects a weights-stationary specialisation (all A operands of a wave live in VGPRs for the whole
// H-step recurrence, the last hidden tile feeds the K-split layer straight from its accumulator):
//   SPEC 0: generic -- any 1..4 layers, operands streamed from L2 every step
//   SPEC 1: 2 hidden layers of equal width <= 256 (one output tile per wave, nw == hidden tiles),
//           S+U <= 32, S <= 32            e.g. 26-200-200-20 (BASELINE configs 4-5): 68 weight VGPRs
//   SPEC 2: 3 hidden layers of equal width <= 64, S+U <= 32, S <= 32   e.g. 4-32-32-32-3 (low-level tutorial)
// The workgroup's action block acts[t][particle][u] (candidate -> clip/penalty -> store), shared by the rollout
// kernels.  Every element is independent (its own Philox block / table lookup / mean-sigma read), so the work is
// spread over ALL threads: one thread per (particle,u) walking t serially pays one global-memory latency per step
// (H x ~1 us -- a fifth of the quad kernel's run time at config 4).  The squared bound violation is summed per
// (particle,u) in t order afterwards, from LDS.  Contains a barrier when q.pen is set (uniform).
// XCD-aware workgroup -> tile map for grids of (tiles, agents).  Workgroup `lin` = bx + gx * by is dispatched to XCD
// lin % 8 (observed, for speed only: MI355X_MICROARCH.md, workgroup dispatch), and each XCD has its own L2.  The
// workgroups of one XCD get a CONTIGUOUS range of an agent's tiles, so that the 16-byte pieces a 4-particle workgroup
// writes into a row of the particle-minor sample matrix [A][H*U][Nst] fill whole 128-byte lines inside ONE L2 instead
// of leaving eight L2s with a partial line each (config 4: 1.4 MB written back for 0.72 MB of samples).  A bijection of
// [0, gx) for every by, whatever the placement really is.
__device__ __forceinline__ int xcd_tile(int bx, int gx, int by) {
    const int off = (gx * by) & 7;
    const int xr = (bx + off) & 7;                        // my XCD
    int base = 0;
    for (int x = 0; x < xr; ++x) {
        const int first = (x - off) & 7;                    // smallest bx of this row on XCD x
        base += first < gx ? (gx - first + 7) >> 3 : 0;
    }
    return base + ((bx - ((xr - off) & 7)) >> 3);
}

// Candidate value of action-sequence element j = t*U + u of particle n (< n_pop), agent a, before the clip: the given
// sequence (SRC_REF / SRC_BUF), or the sampling distribution around a standard draw (injected, or the engine's own counter
// generator) -- optimizer_base.py:55-95's candidates as the rollout kernels see them.
__device__ __forceinline__ float mlp_candidate_value(const MlpRolloutArgs& q, int a, int n, int j, int u) {
    const RolloutArgs& p = q.r;
    if (q.mode == SRC_REF) return p.seq[((size_t)n * p.A + a) * p.HU + j];
    if (q.mode == SRC_BUF) return p.cand[((size_t)a * p.HU + j) * p.Nst + n];
    float xi;
    if (p.inj) xi = p.inj[((size_t)a * p.HU + j) * p.Nst + n];
    else {
        const U4 blk = rng_block(p.key, p.stream, p.iter, (uint32_t)(n + p.pop_offset), (uint32_t)(p.agent_offset + a), (uint32_t)j);
        const uint32_t w = pick_word(blk, (uint32_t)j);
        xi = (q.mode == SRC_UNIFORM) ? word_to_uniform(w) : word_to_trunc_normal(w);
    }
    if (q.mode == SRC_UNIFORM) return xi * (p.hi[u] - p.lo[u]) + p.lo[u];
    return xi * p.sigma[a * p.HU + j] + p.mean[a * p.HU + j];
}

template <int TP>
__device__ __forceinline__ void mlp_fill_actions(const MlpRolloutArgs& q, int a, int n0, int tid, int nthr,
                                                 float* acts, float* pens) {
    const RolloutArgs& p = q.r;
    const int U = p.U, H = p.H;
    const int total = H * TP * U;                       // element e = (t*TP + pp)*U + u is also its index in acts
    // thread <-> element: the particle index runs fastest, so that a wave's stores to the particle-minor sample matrix
    // [A][H*U][Nst] are TP-particle (64-byte at TP = 16) segments.  With the action index fastest -- the order of `acts` --
    // every lane wrote 4 bytes into a different row, 8 KB apart: rocprofv3 counted 27 MB written per launch for 9.6 MB of
    // samples at the config-5 shape (profiles/r2_cfg5cem.md).
    for (int e2 = tid; e2 < total; e2 += nthr) {
        const int pp = e2 % TP, j = e2 / TP;
        const int t = j / U, u = j - t * U;
        const int e = (t * TP + pp) * U + u;
        const int n = n0 + pp;
        float x = 0.0f;
        if (n < p.n_pop) {
            x = mlp_candidate_value(q, a, n, j, u);
            if (!q.pen && p.samples) p.samples[((size_t)a * p.HU + j) * p.Nst + n] = x;
        }
        acts[e] = x;                                    // unclipped when q.pen: clipped in place below
    }
    if (q.pen) {
        __syncthreads();
        for (int i2 = tid; i2 < TP * U; i2 += nthr) {
            const int pp = i2 % TP, u = i2 / TP;          // particle fastest here too (the clipped samples go back)
            const int n = n0 + pp;
            const float lo = p.lo[u], hi = p.hi[u];
            float pen_part = 0.0f;
            if (n < p.n_pop)
                for (int t = 0; t < H; ++t) {
                    const float x = acts[(t * TP + pp) * U + u];
                    const float xf = clipf(x, lo, hi);
                    const float d = x - xf;
                    pen_part = pen_part + d * d;
                    acts[(t * TP + pp) * U + u] = xf;
                    if (p.samples) p.samples[((size_t)a * p.HU + t * U + u) * p.Nst + n] = xf;
                }
            pens[pp * U + u] = pen_part;
        }
    } else {
        for (int i = tid; i < TP * U; i += nthr) pens[i] = 0.0f;
    }
}

template <int SPEC>
__device__ __forceinline__ void rollout_mlp_body(const MlpRolloutArgs& q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    const int a = blockIdx.y;
    const int n0 = blockIdx.x * MLP_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = q.nw, nthr = nw * 64;
    const int S = p.S, U = p.U, H = p.H, L = m.n_layers;
    const int Sp = (S + 3) & ~3;
    const MlpLds lay = mlp_lds_layout(m, H, U, S, nw);
    float* xs = smem + lay.xs;
    float* actbuf[2] = {smem + lay.actA, smem + lay.actB};
    float* part = smem + lay.part;
    float* st = smem + lay.st;
    float* acts = smem + lay.acts;
    const bool normd = m.normalized != 0;
    float* misc = smem + lay.misc;
    float* nmean = smem + lay.norm;             // [S+U] input means (0 when not normalised)
    float* ninv = nmean + (S + U);              // [S+U] 1/(std + 1e-7)   (1 when not normalised)
    float* tmean = ninv + (S + U);              // [S] target mean
    float* tstd = tmean + S;                    // [S] target std + 1e-7
    float* lbias = tstd + S;                    // [S] bias of the last layer

    constexpr int NH = SPEC == 1 ? 2 : (SPEC == 2 ? 3 : 1);        // hidden layers of the specialisation
    constexpr int HTM = SPEC == 1 ? 16 : (SPEC == 2 ? 4 : 1);      // max hidden tiles
    constexpr int IT0M = 2, OTLM = 2;
    float wr_in[IT0M * 4];                     // layer 0: my output tile x input tiles
    float wr_hid[(NH > 1 ? (NH - 1) : 1) * HTM * 4];               // hidden->hidden: my output tile x input tiles
    float wr_out[OTLM * 4];                    // last layer: output tiles x MY input tile (K split)
    f32x4 bias_r[NH];
    if constexpr (SPEC != 0) {
        const int HT = m.tiles[1];
#pragma unroll
        for (int it = 0; it < IT0M; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                wr_in[it * 4 + s] = (it < m.tiles[0]) ? m.wpack[0][(((size_t)wave * m.tiles[0] + it) * 4 + s) * 64 + lane] : 0.0f;
#pragma unroll
        for (int h = 1; h < NH; ++h)
#pragma unroll
            for (int it = 0; it < HTM; ++it)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    wr_hid[((h - 1) * HTM + it) * 4 + s] =
                        (it < HT) ? m.wpack[h][(((size_t)wave * HT + it) * 4 + s) * 64 + lane] : 0.0f;
#pragma unroll
        for (int ot = 0; ot < OTLM; ++ot)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                wr_out[ot * 4 + s] = (ot < m.tiles[NH + 1]) ? m.wpack[NH][(((size_t)ot * HT + wave) * 4 + s) * 64 + lane] : 0.0f;
#pragma unroll
        for (int h = 0; h < NH; ++h) bias_r[h] = *reinterpret_cast<const f32x4*>(m.bpack[h] + ((size_t)wave * 64 + lane) * 4);
    }

    // ---- prologue 1: the tile's whole action block [H][16][U] (candidate -> clip/penalty -> store)
    mlp_fill_actions<MLP_TP>(q, a, n0, tid, nthr, acts, misc);
    for (int f = tid; f < S + U; f += nthr) {
        const float mu = normd ? (f < S ? m.mean_s[f] : m.mean_a[f - S]) : 0.0f;
        const float sd = normd ? (f < S ? m.std_s[f] : m.std_a[f - S]) : 1.0f;
        nmean[f] = mu;
        ninv[f] = normd ? 1.0f / (sd + 1e-7f) : 1.0f;          // system_dynamics_handler.py:119-122 (x - mu)/(sd + 1e-7)
        if (f < S) {
            tmean[f] = normd ? m.mean_t[f] : 0.0f;
            tstd[f] = normd ? (m.std_t[f] + 1e-7f) : 1.0f;
            lbias[f] = m.bpack[L - 1][((size_t)(f >> 4) * 64 + ((f & 15) >> 2) * 16) * 4 + (f & 3)];
        }
    }
    // ---- prologue 2: initial raw state + zero the padded input tiles
    for (int i = tid; i < m.tiles[0] * 256; i += nthr) xs[i] = 0.0f;
    for (int i = tid; i < MLP_TP * S; i += nthr) {
        const int pp = i / S, s = i % S;
        const int n = n0 + pp;
        float v = 0.0f;
        if (q.per_particle_state) v = (n < p.n_pop) ? p.state[(size_t)n * S + s] : 0.0f;
        else v = p.state[a * S + s];
        st[pp * Sp + s] = v;
    }
    __syncthreads();
    for (int i = tid; i < MLP_TP * (S + U); i += nthr) {          // normalised layer-0 input for t = 0
        const int f = i / MLP_TP, pp = i % MLP_TP;
        const float v = (f < S) ? st[pp * Sp + f] : acts[(0 * MLP_TP + pp) * U + (f - S)];
        xs[tile_addr(f, pp)] = (v - nmean[f]) * ninv[f];
    }
    __syncthreads();

    float total = 0.0f;                       // lanes 0..15 of wave 0: reward accumulator of particle `lane`
    const int OTl = m.tiles[L];
    for (int t = 0; t < H; ++t) {
        float* cur = st + (t & 1) * MLP_TP * Sp;
        float* nxt = st + ((t + 1) & 1) * MLP_TP * Sp;
        // ---- dense layers
        if constexpr (SPEC == 0) {
            int in_off = lay.xs;
            for (int l = 0; l < L - 1; ++l) {
                const int out_off = (l & 1) ? lay.actB : lay.actA;
                mlp_layer_out_split(m, q.wp4[l], l, in_off, out_off, wave, lane, nw);
                __syncthreads();
                in_off = out_off;
            }
            mlp_layer_k_split(m, q.wp4[L - 1], L - 1, in_off, lay.part, wave, lane, nw);
        } else {
            const int HT = m.tiles[1];
            f32x4 acc = bias_r[0];
#pragma unroll
            for (int it = 0; it < IT0M; ++it) {
                if (it < m.tiles[0]) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(xs + ((size_t)it * 64 + lane) * 4);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_in[it * 4 + 0], b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_in[it * 4 + 1], b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_in[it * 4 + 2], b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_in[it * 4 + 3], b.w, acc, 0, 0, 0);
                }
            }
            acc.x = apply_act(acc.x, m.act[0]); acc.y = apply_act(acc.y, m.act[0]);
            acc.z = apply_act(acc.z, m.act[0]); acc.w = apply_act(acc.w, m.act[0]);
#pragma unroll
            for (int h = 1; h < NH; ++h) {
                float* out = actbuf[(h - 1) & 1];
                *reinterpret_cast<f32x4*>(out + ((size_t)wave * 64 + lane) * 4) = acc;     // all-gather through LDS
                __syncthreads();
                acc = bias_r[h];
#pragma unroll
                for (int it = 0; it < HTM; ++it) {
                    if (it < HT) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(out + ((size_t)it * 64 + lane) * 4);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[((h - 1) * HTM + it) * 4 + 0], b.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[((h - 1) * HTM + it) * 4 + 1], b.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[((h - 1) * HTM + it) * 4 + 2], b.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[((h - 1) * HTM + it) * 4 + 3], b.w, acc, 0, 0, 0);
                    }
                }
                acc.x = apply_act(acc.x, m.act[h]); acc.y = apply_act(acc.y, m.act[h]);
                acc.z = apply_act(acc.z, m.act[h]); acc.w = apply_act(acc.w, m.act[h]);
            }
            // last layer, K split: my own last-hidden tile (still in `acc`) times my slab of W_last
#pragma unroll
            for (int ot = 0; ot < OTLM; ++ot) {
                if (ot < OTl) {
                    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_out[ot * 4 + 0], acc.x, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_out[ot * 4 + 1], acc.y, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_out[ot * 4 + 2], acc.z, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_out[ot * 4 + 3], acc.w, o, 0, 0, 0);
                    *reinterpret_cast<f32x4*>(part + (((size_t)wave * OTl + ot) * 64 + lane) * 4) = o;
                }
            }
        }
        __syncthreads();
        // ---- epilogue: reduce partials, bias, last activation, de-normalise, residual; stage step t+1's input
        const int nwp = min(nw, m.tiles[L - 1]);          // waves that actually produced partials
        for (int i = tid; i < MLP_TP * (S + U); i += nthr) {
            const int f = i / MLP_TP, pp = i % MLP_TP;
            float v;
            if (f < S) {
                const int ot = f >> 4, ln = ((f & 15) >> 2) * 16 + pp, rg = f & 3;
                const float* pp0 = part + (((size_t)ot) * 64 + ln) * 4 + rg;
                float acc = lbias[f];
#pragma unroll 4
                for (int w = 0; w < nwp; ++w) acc = acc + pp0[(size_t)w * OTl * 256];
                acc = apply_act(acc, m.act[L - 1]);
                const float dev = normd ? tmean[f] + acc * tstd[f] : acc;       // system_dynamics_handler.py:152-155
                const float ns = dev + cur[pp * Sp + f];                        // transforms.py:34
                nxt[pp * Sp + f] = ns;
                if (q.traj && n0 + pp < p.n_pop) q.traj[((((size_t)t * p.A + a) * p.Nst) + n0 + pp) * S + f] = ns;
                v = ns;
            } else {
                const int tn = (t + 1 < H) ? t + 1 : t;
                v = acts[(tn * MLP_TP + pp) * U + (f - S)];
            }
            xs[tile_addr(f, pp)] = (v - nmean[f]) * ninv[f];
        }
        __syncthreads();
        // ---- reward of step t (wave 0, one lane per particle) overlaps the next step's first layer
        if (tid < MLP_TP) {
            const float r = reward_generic(p.reward_kind, p.fix_q1 != 0, cur + tid * Sp, acts + (t * MLP_TP + tid) * U,
                                           nxt + tid * Sp, S, U);
            total = total + r;
        }
    }
    // ---- penalties: sum the (p,u) shares in u order
    __syncthreads();
    const float* pens = misc;
    if (tid < MLP_TP) {
        const int n = n0 + tid;
        if (n < p.n_pop) {
            if (total != total) total = -1.0e6f;                        // deterministic.py:75-77
            if (q.pen) {
                float pen = 0.0f;
                for (int u = 0; u < U; ++u) pen = pen + pens[tid * U + u];
                const float nr = sqrtf(pen);
                pen = nr * nr;
                total = total - pen;
                if (p.penalty_out) p.penalty_out[(size_t)a * p.Nst + n] = pen;
            }
            p.rewards[(size_t)a * p.Nst + n] = total;
        }
    }
    if (q.final_state) {
        const float* fin = st + (H & 1) * MLP_TP * Sp;
        for (int i = tid; i < MLP_TP * S; i += nthr) {
            const int pp = i / S, s = i % S;
            const int n = n0 + pp;
            if (n < p.n_pop) q.final_state[((size_t)a * p.n_pop + n) * S + s] = fin[pp * Sp + s];
        }
    }
}

template <int SPEC>
__global__ void k_rollout_mlp(MlpRolloutArgs q) {
    rollout_mlp_body<SPEC>(q);
}
// same body under its own name for the H = 1 model step (predict_next_state / the __call__ tail), so that
// profiler averages of the rollout kernel are not diluted by the tiny launches
static __global__ void k_step_mlp(MlpRolloutArgs q) {
    rollout_mlp_body<0>(q);
}

// ---- small helpers for the OptimizerBase.__call__ tail on the learned-dynamics path ----------------

// action[a][u] += exploration noise, clip (optimizer_base.py:82-90)
static __global__ void k_explore(FinalArgs p, float* action) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.A * p.U) return;
    const int a = i / p.U, u = i % p.U;
    action[i] = exploration_action(p, a, u, action[i]);
}

// record[a] = (action | next_state | reward)
static __global__ void k_pack_record(int A, int U, int S, const float* action, const float* next_state, const float* reward,
                              float* record, float* next_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int rec = U + S + 1;
    if (i >= A * rec) return;
    const int a = i / rec, c = i % rec;
    float v;
    if (c < U) v = action[a * U + c];
    else if (c < U + S) {
        v = next_state[a * S + (c - U)];
        if (next_out) next_out[a * S + (c - U)] = v;
    } else v = reward[a];
    record[i] = v;
}


// =================================================================================================
// Pair mode: one workgroup advances TWO 16-particle tiles, software-pipelined.
//
// A tile-step is three barrier-separated stages
//   A: layer 0 (IT0*4 MFMAs per wave) + activation, all-gather of h0 through LDS
//   B: layer 1 (HT*4 dependent MFMAs per wave) + activation + K-split of the last layer (OTL*4 MFMAs)
//   C: epilogue (reduce the partial sums, bias, de-normalise, residual, next input) -- VALU/LDS only
// and the matrix pipe idles during C and most of A.  Tile Y runs one stage behind tile X, so every interval
// pairs a matrix-heavy stage of one tile with the VALU/LDS stage of the other
//      interval 3t: A_X(t) | C_Y(t-1)      3t+1: B_X(t) | A_Y(t)      3t+2: C_X(t) | B_Y(t)
// (same three barriers per step, but for two tiles), weights are shared in VGPRs.  Tile counts are
// compile-time so that each interval is straight-line code the scheduler can interleave.
// Restricted to 2 hidden layers of HT tiles each, S+U <= 32, S <= 32 (BASELINE configs 4-5: HT = 13).
template <int ACT>
__device__ __forceinline__ float apply_act_ct(float x) {
    if constexpr (ACT == ACT_TANH) return bb_tanhf(x);
    else if constexpr (ACT == ACT_RELU) return fmaxf(x, 0.0f);
    else if constexpr (ACT == ACT_SIGMOID) return 1.0f / (1.0f + expf(-x));
    else return x;
}

// Two-tile mode runs HT - 1 waves (12 for 200 hidden units: three per SIMD, 168 registers each -- fourteen waves had
// 128 and spilled).  Waves land on SIMD (wave & 3).  Feature tile HT-1 (the 13th, half-empty one) has no wave of its
// own: its layer-1 job is one more dependent chain of 50 MFMAs, and whichever SIMD carries it whole runs 4 jobs against
// 3 while the barrier at the end of the interval waits -- a quarter of the matrix time of every layer-1 interval.  So
// the K loop of that job is split in four quarters taken by the last four waves (wid HT-5 .. HT-2, one per SIMD) as a
// second, independent accumulator chain inside the first groups of their own layer-1 loop: quarters 0..2 leave their
// partial pre-activations in LDS and raise a flag; quarter 3's wave (the owner: also layer 0 of that tile, the shortest
// quarter) collects them two thirds of the way down its own chain, applies the activation and does the K slab of the
// last layer.  The flags are LDS words polled inside the interval (long up when the owner looks) -- no extra barrier.
// The one-tile mode's last wave sums its K loop in the same quarter order, so the two modes agree bit for bit.
// Static __shared__ objects must not appear in this kernel: they move the base of the dynamic carve (measured: wrong
// results and half the speed).
constexpr int mlp_pair_waves(int HT, int NTILES) { return NTILES == 2 ? HT - 1 : HT; }
// development hook (tools/microbench/pair_probe.hip): per-wave clocks at the interval boundaries of one step
#ifndef BBMPC_PAIR_CLK
#define BBMPC_PAIR_CLK(slot)
#endif
#ifndef BBMPC_PAIR_CLK2
#define BBMPC_PAIR_CLK2(slot, seq)      // inside stage_B: slot 8 + 8 * tile + {0: loop entry, 1..3: behind k tile 3 / 7 / 11, 4: loop exit, 5: slab stored}
#endif
constexpr int mlp_pair_kq(int HT) { return (HT + 3) / 4; }          // k tiles per helper quarter
template <int V> struct IC { static constexpr int value = V; };

// CS / CU / CH: dim_S / dim_U / the planning horizon at compile time (0 = read them from the arguments).  With run-time dimensions the compiler keeps
// some forty loop-invariant LDS addresses per thread, does not fit them into the 128 registers a 14-wave workgroup
// leaves per lane and spills 84 bytes per thread: 84 B x 896 threads x 250 workgroups = 18.8 MB of scratch written per
// launch -- the "27 MB written for 9.6 MB of samples" of profiles/r2_cfg5cem.md (WRITE_SIZE itself is exact:
// tools/microbench/write_size_calib.hip) -- and ~20 scratch reloads per pipeline step.  (With 12 waves and 168 registers
// the config-5 instance spills nothing; the run-time-dimension instances still spill a few words.)
// CR: the built-in reward kind at compile time (-1 = read it from the arguments).
// CHALF: whether the last hidden 16-feature tile is half empty (MlpDesc::half_tail of both hidden layers), -1 = run time.
template <int HT, int A0, int A1, int A2, int NTILES, int CS = 0, int CU = 0, int CH = 0, int CR = -1, int CHALF = -1>
__global__ __launch_bounds__(mlp_pair_waves(HT, NTILES) * 64) void k_rollout_mlp_pair(MlpRolloutArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    constexpr int NW = HT, NT = mlp_pair_waves(HT, NTILES) * 64, IT0 = 2, OTL = 2;   // NW: feature tiles = partial-sum slots
    const int a = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: roles are scalar branches
    const int wave = wid;                                     // the feature tile this wave serves
    // two-tile mode: waves HT-5 .. HT-2 also take quarter hq of feature tile HT-1's K loop, hq = 3 owns that tile
    constexpr int KQ = mlp_pair_kq(HT), XT = HT - 1;
    const int hq = (NTILES == 2 && wid >= HT - 5) ? wid - (HT - 5) : -1;
    const bool helper = hq >= 0, owner = hq == 3;
    const int kbase = helper ? hq * KQ : 0;                   // first k tile of the quarter
    const int S = CS ? CS : p.S, U = CU ? CU : p.U, H = CH ? CH : p.H;
    const int Sp = (S + 3) & ~3;
    const bool normd = m.normalized != 0;
    const bool half1 = CHALF >= 0 ? CHALF != 0 : m.half_tail[1] != 0;      // inputs of layer 1 / of the last layer
    const bool half2 = CHALF >= 0 ? CHALF != 0 : m.half_tail[2] != 0;
    // Round 5, padding MFMAs.  (i) The last K tile of layer 0 holds rem0 = (S + U) - 16 (it0n - 1) input features; placed into the
    // slots 4g + {0 .. tk0-1} (tk0 = ceil(rem0 / 4): the half_tail rule of the hidden tiles, generalised) they leave the tile's
    // MFMAs tk0 .. 3 with nothing but zeros, and those are not issued (26 inputs: seven MFMAs per feature tile instead of
    // eight).  The operands come from the one packed array every kernel shares (set_mlp keeps the inputs in order), gathered in
    // this kernel's slot order by the prologue.  (ii) With 16 < dim_S <= 20 the second output tile of the last layer has at
    // most four live rows: its K slab runs as four v_mfma_f32_4x4x1_16b_f32 (block = (k row g, particle quad): 4 features x 16
    // particles x 4 k per instruction, 8 cycles) instead of four 16x16x4 (32 cycles each, twelve of sixteen rows zero); the four
    // k-row partials of a (feature, particle) are summed in registers by lane swaps before the partial sum goes to LDS (left to
    // the epilogue thread -- four words side by side, a 16-byte read and three adds behind a test of the feature index inside
    // the layer-1 loop -- the change cost 10 us of the 422 instead of saving any).
    constexpr bool CT0 = CS != 0 && CU != 0;
    constexpr int C_IT0N = CT0 ? (CS + CU + 15) / 16 : 0;
    const int it0n = CT0 ? C_IT0N : m.tiles[0];
    const int rem0 = (CT0 ? CS + CU : m.dims[0]) - 16 * (it0n - 1);
    const int tk0 = (rem0 + 3) >> 2;
    const bool out4 = S > 16 && S <= 20;
    auto l0_live = [&](int it, int sidx) { return it + 1 < it0n || (it + 1 == it0n && sidx < tk0); };
    // LDS address of input feature f of particle pp in the xs tiles (tile_addr with the last tile's slots permuted)
    auto xs_addr = [&](int f, int pp) {
        const int t = f >> 4, j = f & 15;
        const int slot = (t + 1 == it0n) ? 4 * (j / tk0) + (j % tk0) : j;
        return ((t * 64) + ((slot >> 2) * 16 + pp)) * 4 + (slot & 3);
    };
    // ---- LDS carve
    const int sz_xs = IT0 * 256, sz_h0 = HT * 256, sz_part = NW * OTL * 256, sz_st = 2 * MLP_TP * Sp,
              sz_acts = (H * MLP_TP * U + 3) & ~3, sz_pen = (MLP_TP * U + 63) & ~63;
    const int sz_qp = NTILES == 2 ? 3 * 256 + 64 : 0;   // helper partials [3][64 lanes][4] + their flags
    const int tile_sz = sz_xs + sz_h0 + sz_part + sz_st + sz_acts + sz_pen + sz_qp;
    float* nmean = smem + NTILES * tile_sz;
    float* ninv = nmean + (S + U);
    float* tmean = ninv + (S + U);
    float* tstd = tmean + S;
    float* lbias = tstd + S;
    auto T_xs = [&](int ti) { return smem + ti * tile_sz; };
    auto T_h0 = [&](int ti) { return smem + ti * tile_sz + sz_xs; };
    auto T_part = [&](int ti) { return smem + ti * tile_sz + sz_xs + sz_h0; };
    auto T_st = [&](int ti) { return smem + ti * tile_sz + sz_xs + sz_h0 + sz_part; };
    auto T_acts = [&](int ti) { return smem + ti * tile_sz + sz_xs + sz_h0 + sz_part + sz_st; };
    auto T_pen = [&](int ti) { return smem + ti * tile_sz + sz_xs + sz_h0 + sz_part + sz_st + sz_acts; };
    auto T_qp = [&](int ti) { return smem + ti * tile_sz + sz_xs + sz_h0 + sz_part + sz_st + sz_acts + sz_pen; };

    // ---- stationary weights (operand order, see set_mlp)
    // layer-1 and last-layer slabs stay in VGPRs for the whole recurrence; with two tiles in flight the 8
    // layer-0 operands do not fit the 128-register budget any more and are re-read per stage (L1/L2-resident)
    float wr_in[IT0 * 4], wr_hid[HT * 4], wr_out[OTL * 4];
    // per-lane 32-bit offsets from the (uniform) packed-array bases: the loads take an SGPR base + VGPR offset and no
    // 64-bit per-lane pointer has to stay live across the recurrence
    const float* __restrict__ w_in_b = m.wpack[0];
    auto load_w_in = [&](float* w, int ft) {
        const unsigned base = (unsigned)(ft * it0n * 256);
#pragma unroll
        for (int it = 0; it < IT0; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // packed order: [it][s'][lane'] holds k slot 4 (lane' >> 4) + s' of tile it; this kernel's slot 4g + s of the last
                // tile is the packed slot j = tk0 g + s
                const int g = lane >> 4, j = (it + 1 == it0n) ? tk0 * g + s : 4 * g + s;
                const bool live = l0_live(it, s) && (it + 1 < it0n || j < rem0);
                w[it * 4 + s] = live ? w_in_b[base + (unsigned)((it * 4 + (j & 3)) * 64 + 16 * (j >> 2) + (lane & 15))] : 0.0f;
            }
    };
    load_w_in(wr_in, wave);
#pragma unroll
    for (int it = 0; it < HT; ++it)
#pragma unroll
        for (int s = 0; s < 4; ++s) wr_hid[it * 4 + s] = m.wpack[1][(((size_t)wave * HT + it) * 4 + s) * 64 + lane];
    // feature tile XT: a helper's quarter of the layer-1 operands stays in registers (the owner's is one k tile); what only
    // the owner needs -- layer-0 operands, last-layer slab, the two biases of that tile -- waits in LDS, [slot][lane][4]:
    // slots 0 .. IT0-1 layer 0, IT0 .. IT0+OTL-1 last layer, then bias 0, bias 1
    constexpr int XO_B0 = IT0 + OTL, XO_B1 = XO_B0 + 1, XO_N = XO_B1 + 1;
    static_assert(HT - 3 * KQ <= 1, "owner quarter = one k tile");
    float wx[NTILES == 2 ? KQ * 4 : 1];
    float* xo = nmean + (((S + U) * 2 + S * 3 + 3) & ~3);          // 16-byte aligned behind the statistics
    if constexpr (NTILES == 2) {
#pragma unroll
        for (int i = 0; i < KQ * 4; ++i) wx[i] = 0.0f;
        if (helper) {
#pragma unroll
            for (int j = 0; j < KQ; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (kbase + j < HT) wx[j * 4 + s] = m.wpack[1][(((size_t)XT * HT + kbase + j) * 4 + s) * 64 + lane];
        }
        if (owner) {
            float t8[IT0 * 4];
            load_w_in(t8, XT);
#pragma unroll
            for (int it = 0; it < IT0; ++it)
#pragma unroll
                for (int s = 0; s < 4; ++s) xo[(it * 64 + lane) * 4 + s] = t8[it * 4 + s];
#pragma unroll
            for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int ln = (ot == 1 && out4) ? 16 * (lane >> 4) + (lane & 3) : lane;      // 4x4x1: A lane 4b + i = row i of block b (its k row: lane >> 4)
                    xo[((IT0 + ot) * 64 + lane) * 4 + s] = (ot < m.tiles[3]) ? m.wpack[2][(((size_t)ot * HT + XT) * 4 + s) * 64 + ln] : 0.0f;
                }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                xo[(XO_B0 * 64 + lane) * 4 + s] = m.bpack[0][(unsigned)((XT * 64 + lane) * 4 + s)];
                xo[(XO_B1 * 64 + lane) * 4 + s] = m.bpack[1][(unsigned)((XT * 64 + lane) * 4 + s)];
            }
        }
    }
    auto xo_get = [&](int slot) { return *reinterpret_cast<const f32x4*>(xo + ((size_t)slot * 64 + lane) * 4); };
    // Interval 1 of the two-tile pipeline holds layer 0 of tile X and the epilogue of tile Y, both latency bound; a wave
    // that runs them one after the other makes the interval twice as long as it has to be.  The EWS waves whose threads
    // reduce state features (tid < 16 S) therefore do no layer 0 there: wave EWS + i (no reduction work) takes feature
    // tile i next to its own as a second, independent chain, with tile i's operands in LDS ([i][it | bias][lane][4]).
    const int EWS = (MLP_TP * S + 63) >> 6;
    const bool split_i1 = NTILES == 2 && 2 * EWS <= HT - 2;
    // (Round 5 tried wave 8 with tile 4 as a THIRD chain instead of wave 9's second -- three layer-0 jobs on SIMDs 0 - 2 and four
    // on SIMD 3 instead of 2 / 4 / 3 / 4: 396.3 us against 396.1, nothing; not kept.)
    const int t2 = (split_i1 && wid >= EWS && wid < 2 * EWS) ? wid - EWS : -1;         // second feature tile of this wave
    float* xo2 = xo + XO_N * 256;
    if constexpr (NTILES == 2) {
        if (t2 >= 0) {
            float t8[IT0 * 4];
            load_w_in(t8, t2);
#pragma unroll
            for (int it = 0; it < IT0; ++it)
#pragma unroll
                for (int s = 0; s < 4; ++s) xo2[((t2 * (IT0 + 1) + it) * 64 + lane) * 4 + s] = t8[it * 4 + s];
#pragma unroll
            for (int s = 0; s < 4; ++s) xo2[((t2 * (IT0 + 1) + IT0) * 64 + lane) * 4 + s] = m.bpack[0][(unsigned)((t2 * 64 + lane) * 4 + s)];
        }
    }
#pragma unroll
    for (int ot = 0; ot < OTL; ++ot)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ln = (ot == 1 && out4) ? 16 * (lane >> 4) + (lane & 3) : lane;
            wr_out[ot * 4 + s] = (ot < m.tiles[3]) ? m.wpack[2][(((size_t)ot * HT + wave) * 4 + s) * 64 + ln] : 0.0f;
        }
    const unsigned bias_off = (unsigned)((wave * 64 + lane) * 4);
    const f32x4 bias0_r = *reinterpret_cast<const f32x4*>(m.bpack[0] + bias_off);
    const f32x4 bias1_r = *reinterpret_cast<const f32x4*>(m.bpack[1] + bias_off);

    for (int f = tid; f < S + U; f += NT) {
        const float mu = normd ? (f < S ? m.mean_s[f] : m.mean_a[f - S]) : 0.0f;
        const float sd = normd ? (f < S ? m.std_s[f] : m.std_a[f - S]) : 1.0f;
        nmean[f] = mu;
        ninv[f] = normd ? 1.0f / (sd + 1e-7f) : 1.0f;
        if (f < S) {
            tmean[f] = normd ? m.mean_t[f] : 0.0f;
            tstd[f] = normd ? (m.std_t[f] + 1e-7f) : 1.0f;
            lbias[f] = m.bpack[2][((size_t)(f >> 4) * 64 + ((f & 15) >> 2) * 16) * 4 + (f & 3)];
        }
    }
    // ---- prologue per tile: action block, start state
    for (int ti = 0; ti < NTILES; ++ti) {
        const int n0 = (blockIdx.x * NTILES + ti) * MLP_TP;
        float* acts = T_acts(ti);
        float* pens = T_pen(ti);
        float* st = T_st(ti);
        float* xs = T_xs(ti);
        mlp_fill_actions<MLP_TP>(q, a, n0, tid, NT, acts, pens);
        for (int i = tid; i < sz_xs; i += NT) xs[i] = 0.0f;
        for (int i = tid; i < MLP_TP * S; i += NT) st[(i / S) * Sp + (i % S)] = p.state[a * S + (i % S)];
        if (NTILES == 2 && tid < 64) reinterpret_cast<int*>(T_qp(ti) + 3 * 256)[tid] = 0;       // helper flags
    }
    __syncthreads();
    for (int ti = 0; ti < NTILES; ++ti) {
        float* xs = T_xs(ti);
        const float* st = T_st(ti);
        const float* acts = T_acts(ti);
        for (int i = tid; i < MLP_TP * (S + U); i += NT) {
            const int f = i / MLP_TP, pp = i % MLP_TP;
            const float v = (f < S) ? st[pp * Sp + f] : acts[pp * U + (f - S)];
            xs[xs_addr(f, pp)] = (v - nmean[f]) * ninv[f];
        }
    }
    __syncthreads();

    // epilogue of step t: one thread per (feature, particle); index math is done unconditionally on a clamped
    // index so that it can be scheduled under the other tile's MFMAs, only the stores are predicated
    const int ef = min(tid / MLP_TP, S + U - 1), epp = tid % MLP_TP;
    const bool e_live = tid < MLP_TP * (S + U);
    const int e_ot = ef >> 4, e_ln = ((ef & 15) >> 2) * 16 + epp, e_rg = ef & 3;
    const int e_xaddr = xs_addr(ef, epp);
    // (4-row form of output tile 1: the producing wave has summed the four k rows; feature 16 + i of a particle is the word of
    // lane 16 {0, 2, 1, 3}[i] + particle)
    const bool e4 = out4 && ef >= 16;
    auto epi_part = [&](int ti) {
        const int i4 = (ef - 16) & 3, r4 = ((i4 & 1) << 1) | (i4 >> 1);
        return T_part(ti) + (e4 ? 256 + 16 * r4 + epp : (((size_t)e_ot) * 64 + e_ln) * 4 + e_rg);
    };
    auto epi_term = [&](const float* pw) { return pw[0]; };
    auto epi_reduce = [&](int ti) {
        if (NTILES == 2 && wid * 64 >= MLP_TP * S) return 0.0f;      // a wave of action features only: nothing to reduce
        const float* part = epi_part(ti);
        float acc = lbias[min(ef, S - 1)];
#pragma unroll
        for (int w = 0; w < NW; ++w) acc = acc + epi_term(part + (size_t)w * OTL * 256);
        return acc;
    };
    // ---- stages
    // the live MFMAs of K tile `it` of layer 0 (w0 .. w3: the tile's four operands)
    auto l0_tile = [&](int it, float w0, float w1, float w2, float w3, const f32x4& b, f32x4 acc) {
        if (l0_live(it, 0)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, b.x, acc, 0, 0, 0);
        if (l0_live(it, 1)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, b.y, acc, 0, 0, 0);
        if (l0_live(it, 2)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2, b.z, acc, 0, 0, 0);
        if (l0_live(it, 3)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w3, b.w, acc, 0, 0, 0);
        return acc;
    };
    // the K slab of the last layer from one feature tile's activations `h` (wo0: operands of output tile 0, wo1: of tile 1);
    // `tail_half`: the slab is the half-empty hidden tile (its k slots 4g + {2, 3} are padding)
    auto out_slab = [&](int ft, const f32x4& h, const f32x4& wo0, const f32x4& wo1, bool tail_half, float* part) {
        f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(wo0.x, h.x, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(wo0.y, h.y, o, 0, 0, 0);
        if (!tail_half) {
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(wo0.z, h.z, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(wo0.w, h.w, o, 0, 0, 0);
        }
        *reinterpret_cast<f32x4*>(part + (((size_t)ft * OTL + 0) * 64 + lane) * 4) = o;
        f32x4 o1 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (out4) {
            // block b = lane >> 2 = (k row g = lane >> 4, particle quad): D register i, lane 4b + j = feature 16 + i of particle
            // lane & 15 over the k slots 4g + {0..3} of this tile.  (Alternating this chain with the 16x16x4 one above, MFMA by
            // MFMA, measured 3 us slower: 399 against 396.)
            o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wo1.x, h.x, o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wo1.y, h.y, o1, 0, 0, 0);
            if (!tail_half) {
                o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wo1.z, h.z, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wo1.w, h.w, o1, 0, 0, 0);
            }
            // the four k rows are summed in registers (two lane swaps + adds fold four registers into one: row r of the
            // wave ends up with feature 16 + {0, 2, 1, 3}[r]) and ONE word per lane goes to LDS
            float a0 = o1.x, a1 = o1.y, a2 = o1.z, a3 = o1.w;
            auto r01 = __builtin_amdgcn_permlane32_swap(__float_as_int(a0), __float_as_int(a1), false, false);
            auto r23 = __builtin_amdgcn_permlane32_swap(__float_as_int(a2), __float_as_int(a3), false, false);
            float t01 = __int_as_float(r01[0]) + __int_as_float(r01[1]), t23 = __int_as_float(r23[0]) + __int_as_float(r23[1]);
            auto rr = __builtin_amdgcn_permlane16_swap(__float_as_int(t01), __float_as_int(t23), false, false);
            part[((size_t)ft * OTL + 1) * 256 + lane] = __int_as_float(rr[0]) + __int_as_float(rr[1]);
        } else {
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wo1.x, h.x, o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wo1.y, h.y, o1, 0, 0, 0);
            if (!tail_half) {
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wo1.z, h.z, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wo1.w, h.w, o1, 0, 0, 0);
            }
            *reinterpret_cast<f32x4*>(part + (((size_t)ft * OTL + 1) * 64 + lane) * 4) = o1;
        }
    };
    auto layer0 = [&](int ti, int ft, const float* w, f32x4 acc) {
        const float* xs = T_xs(ti);
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(xs + ((size_t)it * 64 + lane) * 4);
            acc = l0_tile(it, w[it * 4 + 0], w[it * 4 + 1], w[it * 4 + 2], w[it * 4 + 3], b, acc);
        }
        acc.x = apply_act_ct<A0>(acc.x); acc.y = apply_act_ct<A0>(acc.y);
        acc.z = apply_act_ct<A0>(acc.z); acc.w = apply_act_ct<A0>(acc.w);
        *reinterpret_cast<f32x4*>(T_h0(ti) + ((size_t)ft * 64 + lane) * 4) = acc;
    };
    auto stage_A = [&](int ti) {
        layer0(ti, wave, wr_in, bias0_r);
        if constexpr (NTILES == 2)
            if (owner) {                                                    // feature tile XT as well
                float w[IT0 * 4];
#pragma unroll
                for (int it = 0; it < IT0; ++it) {
                    const f32x4 v = xo_get(it);
                    w[it * 4 + 0] = v.x; w[it * 4 + 1] = v.y; w[it * 4 + 2] = v.z; w[it * 4 + 3] = v.w;
                }
                layer0(ti, XT, w, xo_get(XO_B0));
            }
    };
    // layer 0 of this wave's tile and, as a second chain, of tile t2 (or XT for the owner): interval 1 with split_i1
    auto stage_A1 = [&](int ti) {
        const float* xs = T_xs(ti);
        const bool has2 = t2 >= 0 || owner;
        const float* w2 = owner ? xo : xo2 + (size_t)(t2 < 0 ? 0 : t2) * (IT0 + 1) * 256;
        const float* b2 = owner ? xo + XO_B0 * 256 : w2 + IT0 * 256;
        f32x4 acc = bias0_r, acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (has2) acc2 = *reinterpret_cast<const f32x4*>(b2 + (size_t)lane * 4);
#pragma unroll
        for (int it = 0; it < IT0; ++it) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(xs + ((size_t)it * 64 + lane) * 4);
            acc = l0_tile(it, wr_in[it * 4 + 0], wr_in[it * 4 + 1], wr_in[it * 4 + 2], wr_in[it * 4 + 3], b, acc);
            if (has2) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(w2 + ((size_t)it * 64 + lane) * 4);
                acc2 = l0_tile(it, w.x, w.y, w.z, w.w, b, acc2);
            }
        }
        acc.x = apply_act_ct<A0>(acc.x); acc.y = apply_act_ct<A0>(acc.y);
        acc.z = apply_act_ct<A0>(acc.z); acc.w = apply_act_ct<A0>(acc.w);
        *reinterpret_cast<f32x4*>(T_h0(ti) + ((size_t)wave * 64 + lane) * 4) = acc;
        if (has2) {
            acc2.x = apply_act_ct<A0>(acc2.x); acc2.y = apply_act_ct<A0>(acc2.y);
            acc2.z = apply_act_ct<A0>(acc2.z); acc2.w = apply_act_ct<A0>(acc2.w);
            *reinterpret_cast<f32x4*>(T_h0(ti) + ((size_t)(owner ? XT : t2) * 64 + lane) * 4) = acc2;
        }
    };
    // the owner's end of feature tile XT: collect quarters 0..2, activation, K slab of the last layer
    auto finish_x = [&](int ti, int seq, f32x4 acc) {
        float* qp = T_qp(ti);
        const int* qflag = reinterpret_cast<const int*>(qp + 3 * 256);
        while (__hip_atomic_load(qflag + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq ||
               __hip_atomic_load(qflag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq ||
               __hip_atomic_load(qflag + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != seq)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(qp + ((size_t)h * 64 + lane) * 4);
            acc.x = acc.x + o.x; acc.y = acc.y + o.y; acc.z = acc.z + o.z; acc.w = acc.w + o.w;
        }
        acc.x = apply_act_ct<A1>(acc.x); acc.y = apply_act_ct<A1>(acc.y);
        acc.z = apply_act_ct<A1>(acc.z); acc.w = apply_act_ct<A1>(acc.w);
        out_slab(XT, acc, xo_get(IT0 + 0), xo_get(IT0 + 1), half2, T_part(ti));
    };
    // Layer 1 + K split of tile `ti`.  When `co` >= 0 the partial-sum reduction of tile `co`'s epilogue
    // (one LDS read + one add per producing wave) is issued between this tile's dependent MFMA groups, so
    // it costs no time of its own; `cacc` returns the reduced value.
    // Two-tile mode: a helper wave's quarter of feature tile XT rides in the first KQ groups as a second, independent
    // accumulator chain (published right after); the owner collects the quarters two thirds of the way down its chain.
    // `ta` >= 0 (two-tile mode): layer 0 of tile `ta` rides in groups apos, apos+1 (+2, +3 for the owner's tile XT) as one
    // more independent chain instead of running, latency bound, before or after this one.
    // The K loop runs on two accumulators (even / odd k tiles) so that a wave alone on its SIMD still issues back to back.
    auto stage_B = [&](int ti, int co, const float* cpart, float& cacc, int seq, auto ta_c, auto apos_c) {
        constexpr int ta = decltype(ta_c)::value, apos = decltype(apos_c)::value;
        const float* h0 = T_h0(ti);
        f32x4 acc = bias1_r;
        f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 accx = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 acca = bias0_r, accb = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (NTILES == 2)
            if (owner) {
                accx = xo_get(XO_B1);
                if (ta >= 0) accb = xo_get(XO_B0);
            }
        const float* xsa = T_xs(ta >= 0 ? ta : 0);
        const bool xt_wave = NTILES == 1 && wave == HT - 1;
        if (xt_wave) {
            // one-tile mode: the wave of feature tile XT sums its K loop in the two-tile mode's order (quarters 0..2 from
            // zero, the owner's from the bias, then owner + q0 + q1 + q2), so that the two modes agree bit for bit
            f32x4 aq[4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, bias1_r};
#pragma unroll
            for (int it = 0; it < HT; ++it) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(h0 + ((size_t)it * 64 + lane) * 4);
                f32x4& a = aq[it / KQ];
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 0], b.x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 1], b.y, a, 0, 0, 0);
                if (it + 1 < HT || !half1) {
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 2], b.z, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 3], b.w, a, 0, 0, 0);
                }
            }
            acc = aq[3];
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                acc.x = acc.x + aq[h].x; acc.y = acc.y + aq[h].y; acc.z = acc.z + aq[h].z; acc.w = acc.w + aq[h].w;
            }
        } else {
            // (Taking the k tiles in pairs and alternating the two accumulators MFMA by MFMA -- so that a wave alone on its SIMD
            // still issues back to back -- was built and measured in round 5: 407 us against 407, no change; not kept.)
            f32x4 bn = *reinterpret_cast<const f32x4*>(h0 + (size_t)lane * 4);
            BBMPC_PAIR_CLK2(8 + 8 * ti + 0, seq);
#pragma unroll
            for (int it = 0; it < HT; ++it) {
                const f32x4 b = bn;                       // operand of this group was loaded during the previous one
                if (it + 1 < HT) bn = *reinterpret_cast<const f32x4*>(h0 + ((size_t)(it + 1) * 64 + lane) * 4);
                float pv = 0.0f;
                if (co >= 0) pv = epi_term(cpart + (size_t)it * OTL * 256);
                if constexpr (NTILES == 2) {
                    if (it < KQ && helper && kbase + it < HT) {               // wave-uniform
                        const f32x4 bx = *reinterpret_cast<const f32x4*>(h0 + ((size_t)(kbase + it) * 64 + lane) * 4);
                        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[it * 4 + 0], bx.x, accx, 0, 0, 0);
                        accx = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[it * 4 + 1], bx.y, accx, 0, 0, 0);
                        if (kbase + it + 1 < HT || !half1) {
                            accx = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[it * 4 + 2], bx.z, accx, 0, 0, 0);
                            accx = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[it * 4 + 3], bx.w, accx, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < IT0; ++j) {
                        if (ta >= 0 && it == apos + j) {
                            const f32x4 ba = *reinterpret_cast<const f32x4*>(xsa + ((size_t)j * 64 + lane) * 4);
                            acca = l0_tile(j, wr_in[j * 4 + 0], wr_in[j * 4 + 1], wr_in[j * 4 + 2], wr_in[j * 4 + 3], ba, acca);
                        }
                        if (ta >= 0 && owner && it == apos + IT0 + j) {
                            const f32x4 ba = *reinterpret_cast<const f32x4*>(xsa + ((size_t)j * 64 + lane) * 4);
                            const f32x4 wa = xo_get(j);
                            accb = l0_tile(j, wa.x, wa.y, wa.z, wa.w, ba, accb);
                        }
                    }
                }
                if ((it & 1) == 0) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 0], b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 1], b.y, acc, 0, 0, 0);
                    if (it + 1 < HT || !half1) {               // the padded half of the last K tile multiplies zeros
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 2], b.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 3], b.w, acc, 0, 0, 0);
                    }
                } else {
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 0], b.x, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 1], b.y, acc2, 0, 0, 0);
                    if (it + 1 < HT || !half1) {
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 2], b.z, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr_hid[it * 4 + 3], b.w, acc2, 0, 0, 0);
                    }
                }
                if (co >= 0) cacc = cacc + pv;
                if constexpr (NTILES == 2) {
                    if (it == KQ - 1 && helper && !owner) {
                        float* qp = T_qp(ti);
                        *reinterpret_cast<f32x4*>(qp + ((size_t)hq * 64 + lane) * 4) = accx;
                        // LDS operations of a wave complete in order: whoever sees the flag sees the partials
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0)
                            __hip_atomic_store(reinterpret_cast<int*>(qp + 3 * 256) + hq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (ta >= 0 && it == apos + IT0 - 1) {
                        acca.x = apply_act_ct<A0>(acca.x); acca.y = apply_act_ct<A0>(acca.y);
                        acca.z = apply_act_ct<A0>(acca.z); acca.w = apply_act_ct<A0>(acca.w);
                        *reinterpret_cast<f32x4*>(T_h0(ta) + ((size_t)wave * 64 + lane) * 4) = acca;
                    }
                    if (ta >= 0 && owner && it == apos + 2 * IT0 - 1) {
                        accb.x = apply_act_ct<A0>(accb.x); accb.y = apply_act_ct<A0>(accb.y);
                        accb.z = apply_act_ct<A0>(accb.z); accb.w = apply_act_ct<A0>(accb.w);
                        *reinterpret_cast<f32x4*>(T_h0(ta) + ((size_t)XT * 64 + lane) * 4) = accb;
                    }
                    if (it == (2 * HT) / 3 && owner) finish_x(ti, seq, accx);
                }
                if (it == 3) { BBMPC_PAIR_CLK2(8 + 8 * ti + 1, seq); }
                if (it == 7) { BBMPC_PAIR_CLK2(8 + 8 * ti + 2, seq); }
                if (it == 11) { BBMPC_PAIR_CLK2(8 + 8 * ti + 3, seq); }
            }
            BBMPC_PAIR_CLK2(8 + 8 * ti + 4, seq);
            acc.x = acc.x + acc2.x; acc.y = acc.y + acc2.y; acc.z = acc.z + acc2.z; acc.w = acc.w + acc2.w;
        }
        acc.x = apply_act_ct<A1>(acc.x); acc.y = apply_act_ct<A1>(acc.y);
        acc.z = apply_act_ct<A1>(acc.z); acc.w = apply_act_ct<A1>(acc.w);
        {
            const f32x4 wo0 = {wr_out[0], wr_out[1], wr_out[2], wr_out[3]}, wo1 = {wr_out[4], wr_out[5], wr_out[6], wr_out[7]};
            // one-tile mode: the last wave's K slice is the half-empty tile
            out_slab(wave, acc, wo0, wo1, !(NTILES == 2 || wave + 1 < HT || !half2), T_part(ti));
        }
        BBMPC_PAIR_CLK2(8 + 8 * ti + 5, seq);
    };
    // finish the epilogue of step t given the reduced pre-activation `acc`
    // cheetah reward (cost_func.py:5-22) needs cur[5..7], cur[17], next[17] and the actions only: the epilogue
    // thread that produces feature 17 of particle epp has next[17] in hand and evaluates it right there
    // (its reward accumulator is collected at the end); other rewards go through reward() below.
    const int rkind = CR >= 0 ? CR : p.reward_kind;
    const bool rew_inline = rkind == REW_CHEETAH && S > 17;
    float rew_acc[2] = {0.0f, 0.0f};
    auto epi_finish = [&](int ti, int t, float acc) {
        float* st = T_st(ti);
        float* cur = st + (t & 1) * MLP_TP * Sp;
        float* nxt = st + ((t + 1) & 1) * MLP_TP * Sp;
        float v;
        if (ef < S) {
            acc = apply_act_ct<A2>(acc);
            const float dev = normd ? tmean[ef] + acc * tstd[ef] : acc;
            const float c17 = cur[epp * Sp + ef];
            v = dev + c17;
            if (e_live) nxt[epp * Sp + ef] = v;
            if (CR < 0 && q.traj && e_live) {                        // state after step t (a user reward scores it afterwards: never with a built-in reward at compile time)
                // uniform 64-bit base + a 32-bit per-lane offset: no 64-bit per-lane index lives across the recurrence
                const int n = (blockIdx.x * NTILES + ti) * MLP_TP + epp;
                float* trow = q.traj + (((size_t)t * p.A + a) * p.Nst) * S;
                if (n < p.n_pop) trow[(unsigned)(n * S + ef)] = v;
            }
            if (rew_inline && ef == 17) {
                const float c5 = cur[epp * Sp + 5], c6 = cur[epp * Sp + 6], c7 = cur[epp * Sp + 7];
                const float* ac = T_acts(ti) + (t * MLP_TP + epp) * U;
                float ss = 0.0f;
                for (int u = 0; u < U; ++u) ss = ss + ac[u] * ac[u];
                float r = 0.0f;
                if (c5 >= 0.2f) r = r + (-10.0f);
                if (c6 >= 0.0f) r = r + (-10.0f);
                if (c7 >= 0.0f) r = r + (-10.0f);
                r = r + (v - c17) / 0.01f;
                r = r - 0.0f * ss;
                rew_acc[ti] = rew_acc[ti] + r;
            }
        } else {
            const int tn = (t + 1 < H) ? t + 1 : t;
            v = T_acts(ti)[(tn * MLP_TP + epp) * U + (ef - S)];
        }
        if (e_live) T_xs(ti)[e_xaddr] = (v - nmean[ef]) * ninv[ef];
    };
    float total[2] = {0.0f, 0.0f};                 // lanes 0..15 of wave 0: particle `lane` of tile 0 / 1
    auto reward = [&](int ti, int t) {
        if (!rew_inline && tid < MLP_TP) {
            const float* st = T_st(ti);
            const float* cur = st + (t & 1) * MLP_TP * Sp;
            const float* nxt = st + ((t + 1) & 1) * MLP_TP * Sp;
            total[ti] = total[ti] + reward_generic(rkind, p.fix_q1 != 0, cur + tid * Sp,
                                                   T_acts(ti) + (t * MLP_TP + tid) * U, nxt + tid * Sp, S, U);
        }
    };

    // ---- pipelined recurrence: tile 1 runs one stage behind tile 0
    float dummy = 0.0f;
    if constexpr (NTILES == 1) {
        // one tile: the three stages back to back (same stage bodies, B-operand prefetch, compile-time activations)
        for (int t = 0; t < H; ++t) {
            stage_A(0);
            if (t > 0) reward(0, t - 1);
            __syncthreads();
            stage_B(0, -1, nullptr, dummy, 0, IC<-1>{}, IC<0>{});
            __syncthreads();
            epi_finish(0, t, epi_reduce(0));
            __syncthreads();
        }
        reward(0, H - 1);
    } else {
        // Waves of one SIMD are split into two groups that run the two (independent) stages of an interval in
        // opposite order: right after a barrier every wave would otherwise reach its activation (VALU) section at
        // the same time and leave the matrix pipe idle.  Waves w, w+4, w+8, ... share a SIMD, so (w >> 2) & 1
        // alternates within each SIMD.
        const bool grp = ((wid >> 2) & 1) != 0;
        const bool epi_wave = wid * 64 < MLP_TP * (S + U);
        // the helpers (the longest instruction streams of their SIMDs) and the second-dispatched third of the waves, which lose
        // every arbitration by age, at priority 1; waves 0 - 3 take what is left.  Measured in round 5 (us per launch): helpers
        // only 403, helpers + waves 4 - 7 397, helpers at 2 and waves 4 - 7 at 1: 401, waves 4 - 7 only 405.
        if (helper || grp) __builtin_amdgcn_s_setprio(1);
        for (int t = 0; t < H; ++t) {
            BBMPC_PAIR_CLK(0);
            // epilogue threads are tid < 16 (S + U): waves 0 .. EW-1 (the others have nothing to reduce)
            if (split_i1) {
                if (wid >= EWS) stage_A1(0);                          // A_X(t), two feature tiles per wave
                // (the reducing waves at priority 2 / 3 for this interval -- their epilogue is one dependent chain next to the other
                // waves' layer-0 MFMAs --: 397.7 us against 395.8, not kept)
                if (t > 0 && epi_wave) epi_finish(1, t - 1, epi_reduce(1));       // C_Y(t-1)
            } else if (!grp) {
                stage_A(0);                                           // A_X(t)
                if (t > 0 && epi_wave) epi_finish(1, t - 1, epi_reduce(1));       // C_Y(t-1)
            } else {
                if (t > 0 && epi_wave) epi_finish(1, t - 1, epi_reduce(1));
                stage_A(0);
            }
            if (t > 0) reward(0, t - 1);                  // state pair (t-1, t) of tile 0 is complete since the last barrier
            BBMPC_PAIR_CLK(1);
            __syncthreads();
            BBMPC_PAIR_CLK(2);
            // B_X(t) with A_Y(t) inside, early or late
            if (!grp) stage_B(0, -1, nullptr, dummy, t + 1, IC<1>{}, IC<0>{});
            else stage_B(0, -1, nullptr, dummy, t + 1, IC<1>{}, IC<HT - 5>{});
            if (t > 0) reward(1, t - 1);
            BBMPC_PAIR_CLK(3);
            __syncthreads();
            BBMPC_PAIR_CLK(4);
            if (!epi_wave) {
                stage_B(1, -1, nullptr, dummy, t + 1, IC<-1>{}, IC<0>{});
            } else if (!grp || wid * 64 < MLP_TP * S) {          // every reducing wave rides (round 5: wave 4 did its epilogue first and was the interval's pole, -4 us)
                float cacc = lbias[min(ef, S - 1)];                   // C_X(t): reduction rides under B_Y(t)'s MFMA chain
                stage_B(1, 0, epi_part(0), cacc, t + 1, IC<-1>{}, IC<0>{});
                epi_finish(0, t, cacc);
            } else {
                epi_finish(0, t, epi_reduce(0));
                stage_B(1, -1, nullptr, dummy, t + 1, IC<-1>{}, IC<0>{});
            }
            BBMPC_PAIR_CLK(5);
            __syncthreads();
            BBMPC_PAIR_CLK(6);
        }
        if (epi_wave) epi_finish(1, H - 1, epi_reduce(1));
        reward(0, H - 1);
        __syncthreads();
        reward(1, H - 1);
    }

    // ---- results
    if (rew_inline) {
        __syncthreads();
        if (ef == 17 && e_live)
            for (int ti = 0; ti < NTILES; ++ti) T_part(ti)[epp] = rew_acc[ti];
        __syncthreads();
        if (tid < MLP_TP)
            for (int ti = 0; ti < NTILES; ++ti) total[ti] = T_part(ti)[tid];
    }
    for (int ti = 0; ti < NTILES; ++ti) {
        if (tid < MLP_TP) {
            const int n = (blockIdx.x * NTILES + ti) * MLP_TP + tid;
            if (n < p.n_pop) {
                float tot = total[ti];
                if (tot != tot) tot = -1.0e6f;
                if (q.pen) {
                    const float* pens = T_pen(ti);
                    float pen = 0.0f;
                    for (int u = 0; u < U; ++u) pen = pen + pens[tid * U + u];
                    const float nr = sqrtf(pen);
                    pen = nr * nr;
                    tot = tot - pen;
                    if (p.penalty_out) (p.penalty_out + (size_t)a * p.Nst)[(unsigned)n] = pen;
                }
                (p.rewards + (size_t)a * p.Nst)[(unsigned)n] = tot;
            }
        }
    }
}

// LDS floats the pair kernel needs
inline int mlp_pair_lds_floats(int HT, int H, int U, int S, int ntiles) {
    const int Sp = (S + 3) & ~3;
    const int tile = 2 * 256 + HT * 256 + HT * 2 * 256 + 2 * MLP_TP * Sp + ((H * MLP_TP * U + 3) & ~3) + ((MLP_TP * U + 63) & ~63) +
                     (ntiles == 2 ? 3 * 256 + 64 : 0);
    return ntiles * tile + (((S + U) * 2 + S * 3 + 3 + 63) & ~63) + (ntiles == 2 ? (6 + 5 * 3) * 256 : 0);   // + the owner's operands (6 slots) + second-tile layer-0 operands (5 x 3)
}

}  // namespace bbmpc

namespace bbmpc {

// =================================================================================================
// Quad mode: 4 particles per workgroup on v_mfma_f32_4x4x1_16b_f32.
//
// The 16x16x4 tiling needs 16 particles per workgroup, so a population of 1000 (BASELINE config 4) makes
// only 63 workgroups for 256 CUs.  The 16-block 4x4x1 form computes, per instruction, 16 independent
// (4 features x 4 particles) outer products: give all 16 blocks the same 4 particles and 64 different features
// and one wave produces 64 features x 4 particles per k.  A workgroup then needs just 4 particles:
// 250 workgroups at config 4.  With ceil(hidden/64) = 4 waves per workgroup every wave sits alone on its SIMD,
// owns the full 512-entry register file and keeps ALL of its A operands stationary (K0 + hidden + hidden/4
// registers: 26 + 200 + 50 for 26-200-200-20).
//   A operand, lane l : W[k][64*wave + l]                    (block l>>2, row l&3)
//   B operand, lane l : x[k][particle l&3]                   (same for every block)
//   D fragment, lane l: out[64*wave + 4*(l>>2) + r][l&3], r = 0..3
// Activations travel through LDS as [k/4][particle][4] so that a lane fetches 4 consecutive k of its particle
// with one ds_read_b128.  Last layer: K split over the waves, partial sums reduced in the epilogue.
// Two hidden layers of equal width; HG = hidden/4 groups, K0G = ceil((S+U)/4) groups, NWQ waves.
//
// The 200 layer-1 A operands of a lane do not fit next to everything else in the 256 architectural VGPRs, and left to
// itself the compiler parks them in AccVGPRs and copies each one back with v_accvgpr_read before the MFMA that uses it
// (112-144 copies per step, on the same issue port as the MFMAs).  MFMA can read SrcA straight from an AccVGPR, so
// layer 1 issues its MFMAs through inline asm with an "a" constraint on the weight: the weights live in AccVGPRs for
// the whole recurrence and nothing is copied.  The hazard recogniser does not look inside inline asm:
//  * the four accumulator chains are independent and interleaved, so a dependent MFMA is 4 issues (32 cycles) behind
//    its producer (the 4x4x1 form needs 12);
//  * mfma_operands_settled() separates the VALU writes that initialise the accumulators from the first MFMA;
//  * before the first VALU use of the results mfma_results_ready() spends the wait states the compiler would have
//    inserted (2-pass XDL write -> VALU read) and ties the accumulators to it.
__device__ __forceinline__ void mfma4x4_agpr_a(f32x4& acc, float w_in_agpr, float b) {
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w_in_agpr), "v"(b));
}
__device__ __forceinline__ void mfma_operands_settled(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    asm volatile("s_nop 3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));     // VALU-initialised accumulators -> first MFMA
}
__device__ __forceinline__ void mfma_results_ready(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
}

template <int HG, int K0G, int NWQ, int A0, int A1, int A2>
__global__ __launch_bounds__(NWQ * 64, 1) void k_rollout_mlp_q4(MlpRolloutArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    constexpr int NT = NWQ * 64, QP = 4;                 // threads, particles per workgroup
    constexpr int HK = HG * 4;                           // hidden width (multiple of 4)
    constexpr int KS = (HG + NWQ - 1) / NWQ;             // k groups per wave in the K-split last layer
    const int a = blockIdx.y, n0 = xcd_tile(blockIdx.x, gridDim.x, blockIdx.y) * QP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = p.S, U = p.U, H = p.H;
    const int Sp = (S + 3) & ~3;
    const bool normd = m.normalized != 0;
    // ---- LDS: xs[K0G][4][4] | h0[HG][4][4] | h1[HG][4][4] | part[NWQ][64][4] | st[2][4][Sp] | acts[H][4][U] | pen[4*U] | norm
    float* xs = smem;
    float* h0 = xs + K0G * 16;
    float* h1 = h0 + HG * 16;
    float* part = h1 + HG * 16;
    float* st = part + NWQ * 256;
    float* acts = st + 2 * QP * Sp;
    float* pens = acts + ((H * QP * U + 3) & ~3);
    float* nmean = pens + ((QP * U + 3) & ~3);
    float* ninv = nmean + (S + U);
    float* tmean = ninv + (S + U);
    float* tstd = tmean + S;
    float* lbias = tstd + S;
    float* xa = smem + (((int)(lbias - smem) + S + 3) & ~3);   // [H][K0G - S/4][4][4] normalised action groups (fast epilogue), 16-byte aligned
    float* zs = xa + H * K0G * 16;                         // [H][4]   0 * sum(a^2) of cost_func.py:21

    // ---- stationary A operands.  Packed by bbmpc_set_mlp as [k/4][Mp][4] (zero padded): a lane's four consecutive-k operands are one 16-byte
    // load, a wave's load is 1 KB contiguous -- 70 loads per lane instead of 278 dword loads (the prologue was a
    // fifth of the kernel).
    const int M1 = m.dims[1], M3 = m.dims[3];
    const int Mp1 = (M1 + 63) & ~63, Mp3 = (M3 + 63) & ~63;
    const float4* __restrict__ Q0 = reinterpret_cast<const float4*>(q.wq4[0]);
    const float4* __restrict__ Q1 = reinterpret_cast<const float4*>(q.wq4[1]);
    const float4* __restrict__ Q2 = reinterpret_cast<const float4*>(q.wq4[2]);
    const int f = wave * 64 + lane;                      // hidden feature this lane's A operands belong to (< Mp1)
    float wA0[K0G * 4], wA1[HK], wA2[KS * 4];
#pragma unroll
    for (int g = 0; g < K0G; ++g) {
        const float4 v = Q0[(size_t)g * Mp1 + f];
        wA0[4 * g + 0] = v.x; wA0[4 * g + 1] = v.y; wA0[4 * g + 2] = v.z; wA0[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int g = 0; g < HG; ++g) {
        const float4 v = Q1[(size_t)g * Mp1 + f];
        wA1[4 * g + 0] = v.x; wA1[4 * g + 1] = v.y; wA1[4 * g + 2] = v.z; wA1[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int g = 0; g < KS; ++g) {                       // last layer: my k range, output feature = lane
        const int gg = wave * KS + g;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (gg < HG) v = Q2[(size_t)gg * Mp3 + lane];
        wA2[4 * g + 0] = v.x; wA2[4 * g + 1] = v.y; wA2[4 * g + 2] = v.z; wA2[4 * g + 3] = v.w;
    }
    // bias of my 4 D rows: features 64*wave + 4*(lane>>2) + r
    f32x4 b0, b1;
    {
        const int fb = wave * 64 + 4 * (lane >> 2);
        b0.x = fb + 0 < M1 ? q.braw[0][fb + 0] : 0.0f; b0.y = fb + 1 < M1 ? q.braw[0][fb + 1] : 0.0f;
        b0.z = fb + 2 < M1 ? q.braw[0][fb + 2] : 0.0f; b0.w = fb + 3 < M1 ? q.braw[0][fb + 3] : 0.0f;
        b1.x = fb + 0 < M1 ? q.braw[1][fb + 0] : 0.0f; b1.y = fb + 1 < M1 ? q.braw[1][fb + 1] : 0.0f;
        b1.z = fb + 2 < M1 ? q.braw[1][fb + 2] : 0.0f; b1.w = fb + 3 < M1 ? q.braw[1][fb + 3] : 0.0f;
    }
    for (int i = tid; i < S + U; i += NT) {
        const float mu = normd ? (i < S ? m.mean_s[i] : m.mean_a[i - S]) : 0.0f;
        const float sd = normd ? (i < S ? m.std_s[i] : m.std_a[i - S]) : 1.0f;
        nmean[i] = mu;
        ninv[i] = normd ? 1.0f / (sd + 1e-7f) : 1.0f;
        if (i < S) {
            tmean[i] = normd ? m.mean_t[i] : 0.0f;
            tstd[i] = normd ? (m.std_t[i] + 1e-7f) : 1.0f;
            lbias[i] = q.braw[2][i];
        }
    }
    // ---- prologue: the 4 particles' action block + start state
    mlp_fill_actions<QP>(q, a, n0, tid, NT, acts, pens);
    for (int i = tid; i < K0G * 16; i += NT) xs[i] = 0.0f;
    for (int i = tid; i < QP * S; i += NT) st[(i / S) * Sp + (i % S)] = p.state[a * S + (i % S)];
    __syncthreads();
    // vector layout [k/4][p][4]
    auto vaddr = [](int k, int pp) { return ((k >> 2) * 4 + pp) * 4 + (k & 3); };
    for (int i = tid; i < QP * (S + U); i += NT) {
        const int k = i / QP, pp = i % QP;
        const float v = (k < S) ? st[pp * Sp + k] : acts[pp * U + (k - S)];
        xs[vaddr(k, pp)] = (v - nmean[k]) * ninv[k];
    }
    __syncthreads();

    // ---- fast epilogue (cheetah reward, S a multiple of 4): wave 0 keeps the state in D-fragment layout in
    // registers -- lane g*4+pp owns features 4g..4g+3 of particle pp, exactly the float4 the next step's layer-0 B
    // operand wants -- so a step's epilogue is four 16-byte partial reads, ~40 VALU and one 16-byte write instead of
    // 104 threads doing scalar LDS gathers.  The action part of the next input is pre-normalised once.
    const bool rew_none = p.reward_kind == REW_NONE;          // a user reward function scores the recorded trajectory afterwards
    const bool rew_inline0 = (p.reward_kind == REW_CHEETAH) && S > 17;
    const bool fast_epi = (rew_inline0 || rew_none) && (S & 3) == 0 && S <= 64 && (S + U) <= K0G * 4;
    const int SG = S >> 2, AG = K0G - SG;
    if (fast_epi) {
        for (int e = tid; e < H * AG * 16; e += NT) {
            const int c = e & 3, pp = (e >> 2) & 3, ga = (e >> 4) % AG, t = e / (16 * AG);
            const int k = S + ga * 4 + c;
            xa[e] = (k < S + U) ? (acts[(t * QP + pp) * U + (k - S)] - nmean[k]) * ninv[k] : 0.0f;
        }
        for (int e = tid; e < H * QP; e += NT) {
            const float* ac = acts + e * U;                // e = t*QP + pp
            float ss = 0.0f;
            for (int u = 0; u < U; ++u) ss = ss + ac[u] * ac[u];
            zs[e] = 0.0f * ss;
        }
    }
    f32x4 cur4 = {0.0f, 0.0f, 0.0f, 0.0f}, e_bias = cur4, e_tm = cur4, e_ts = {1.0f, 1.0f, 1.0f, 1.0f}, e_nm = cur4, e_ni = e_ts;
    if (fast_epi && wave == 0 && lane < S) {
        const int k0 = (lane >> 2) * 4;
        for (int r = 0; r < 4; ++r) {
            cur4[r] = p.state[a * S + k0 + r];
            e_bias[r] = lbias[k0 + r]; e_tm[r] = tmean[k0 + r]; e_ts[r] = tstd[k0 + r];
            e_nm[r] = nmean[k0 + r]; e_ni[r] = ninv[k0 + r];
        }
    }
    __syncthreads();

    const int pl = lane & 3;                              // my particle
    const int my_row = (wave * 16 + (lane >> 2)) * 16 + pl * 4;     // where my D fragment goes in h0/h1 (floats)
    const bool own_rows = (wave * 64 + 4 * (lane >> 2)) < HK;
    // HalfCheetah reward (cost_func.py:5-22) needs only cur[5..7], cur[17], nxt[17] and the action: the epilogue
    // thread of output feature 17 accumulates it in place (same operation order as reward_generic); any other
    // reward goes through reward_generic on threads 0..3.
    const bool rew_inline = (p.reward_kind == REW_CHEETAH) && S > 17;
    float total = 0.0f;                                   // rew_inline: thread (feature 17, particle pp); else threads 0..3
#ifdef BBMPC_KERNEL_DBG
    long long dbg_acc[6] = {0, 0, 0, 0, 0, 0}, dbg_t0 = 0;
#define Q4_MARK(i) do { const long long now_ = (long long)wall_clock64(); dbg_acc[i] += now_ - dbg_t0; dbg_t0 = now_; } while (0)
    dbg_t0 = (long long)wall_clock64();
    const long long dbg_start = dbg_t0;
#else
#define Q4_MARK(i) do {} while (0)
#endif
    for (int t = 0; t < H; ++t) {
        float* cur = st + (t & 1) * QP * Sp;
        float* nxt = st + ((t + 1) & 1) * QP * Sp;
        // ---- layer 0
        {
            f32x4 acc0 = b0, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 bq[K0G];                                  // all B operands in flight before the first MFMA
#pragma unroll
            for (int g = 0; g < K0G; ++g) bq[g] = *reinterpret_cast<const f32x4*>(xs + (g * 4 + pl) * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < K0G; ++g) {
                const f32x4 b = bq[g];
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA0[g * 4 + 0], b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA0[g * 4 + 1], b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA0[g * 4 + 2], b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA0[g * 4 + 3], b.w, acc1, 0, 0, 0);
            }
            f32x4 o;
            o.x = apply_act_ct<A0>(acc0.x + acc1.x); o.y = apply_act_ct<A0>(acc0.y + acc1.y);
            o.z = apply_act_ct<A0>(acc0.z + acc1.z); o.w = apply_act_ct<A0>(acc0.w + acc1.w);
            if (own_rows) *reinterpret_cast<f32x4*>(h0 + my_row) = o;
        }
        Q4_MARK(0);
        if (!rew_inline && t > 0 && tid < QP) {            // reward of step t-1 (both states complete)
            const float* c0 = st + ((t - 1) & 1) * QP * Sp;
            total = total + reward_generic(p.reward_kind, p.fix_q1 != 0, c0 + tid * Sp, acts + ((t - 1) * QP + tid) * U,
                                           cur + tid * Sp, S, U);
        }
        __syncthreads();
        Q4_MARK(1);
        // ---- layer 1: 4 independent accumulator chains
        {
            f32x4 c0 = b1, c1 = {0.0f, 0.0f, 0.0f, 0.0f}, c2 = c1, c3 = c1;
            // B operands are fetched one chunk ahead of the MFMAs that consume them: this wave is alone on its SIMD,
            // nothing else hides the LDS latency
            constexpr int CH = 10, NCH = (HG + CH - 1) / CH;
            f32x4 bq[2][CH];
#pragma unroll
            for (int g = 0; g < CH; ++g) bq[0][g] = *reinterpret_cast<const f32x4*>(h0 + (min(g, HG - 1) * 4 + pl) * 4);
            __builtin_amdgcn_sched_barrier(0);
            mfma_operands_settled(c0, c1, c2, c3);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + 1 < NCH) {
#pragma unroll
                    for (int g = 0; g < CH; ++g)
                        bq[(c + 1) & 1][g] = *reinterpret_cast<const f32x4*>(h0 + (min((c + 1) * CH + g, HG - 1) * 4 + pl) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < CH; ++g) {
                    const int gg = c * CH + g;
                    if (gg < HG) {
                        const f32x4 b = bq[c & 1][g];
                        mfma4x4_agpr_a(c0, wA1[gg * 4 + 0], b.x);
                        mfma4x4_agpr_a(c1, wA1[gg * 4 + 1], b.y);
                        mfma4x4_agpr_a(c2, wA1[gg * 4 + 2], b.z);
                        mfma4x4_agpr_a(c3, wA1[gg * 4 + 3], b.w);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_results_ready(c0, c1, c2, c3);
            f32x4 o;
            o.x = apply_act_ct<A1>((c0.x + c1.x) + (c2.x + c3.x)); o.y = apply_act_ct<A1>((c0.y + c1.y) + (c2.y + c3.y));
            o.z = apply_act_ct<A1>((c0.z + c1.z) + (c2.z + c3.z)); o.w = apply_act_ct<A1>((c0.w + c1.w) + (c2.w + c3.w));
            if (own_rows) *reinterpret_cast<f32x4*>(h1 + my_row) = o;
        }
        Q4_MARK(2);
        __syncthreads();
        Q4_MARK(3);
        // ---- last layer, K split: my k range, output feature = 4*(lane>>2)+r ... only features < S matter
        {
            f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = c0;
            f32x4 bq[KS];
#pragma unroll
            for (int g = 0; g < KS; ++g) {
                const int gg = min(wave * KS + g, HG - 1);               // clamped; out-of-range k has zero weights
                bq[g] = *reinterpret_cast<const f32x4*>(h1 + (gg * 4 + pl) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < KS; ++g) {
                const f32x4 b = bq[g];
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA2[g * 4 + 0], b.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA2[g * 4 + 1], b.y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA2[g * 4 + 2], b.z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wA2[g * 4 + 3], b.w, c1, 0, 0, 0);
            }
            f32x4 o = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
            *reinterpret_cast<f32x4*>(part + ((size_t)wave * 64 + lane) * 4) = o;
        }
        __syncthreads();
        Q4_MARK(4);
        if (fast_epi) {
            if (wave == 0) {
                if (lane < S) {
                    f32x4 pr[NWQ];
#pragma unroll
                    for (int w = 0; w < NWQ; ++w) pr[w] = *reinterpret_cast<const f32x4*>(part + ((size_t)w * 64 + lane) * 4);
                    f32x4 v4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float acc = e_bias[r];
#pragma unroll
                        for (int w = 0; w < NWQ; ++w) acc = acc + pr[w][r];
                        acc = apply_act_ct<A2>(acc);
                        const float dev = normd ? e_tm[r] + acc * e_ts[r] : acc;
                        v4[r] = dev + cur4[r];
                    }
                    // reward: flags live in the lanes of features 4..7, the progress term in those of 16..19
                    float fl = 0.0f;
                    if (cur4.y >= 0.2f) fl = fl + (-10.0f);            // cur[5]
                    if (cur4.z >= 0.0f) fl = fl + (-10.0f);            // cur[6]
                    if (cur4.w >= 0.0f) fl = fl + (-10.0f);            // cur[7]
                    const float f0 = __builtin_amdgcn_readlane(fl, 4), f1 = __builtin_amdgcn_readlane(fl, 5),
                                f2 = __builtin_amdgcn_readlane(fl, 6), f3 = __builtin_amdgcn_readlane(fl, 7);
                    if (q.traj && n0 + pl < p.n_pop)       // state after step t, features 4g..4g+3 of my particle
                        *reinterpret_cast<f32x4*>(q.traj + ((((size_t)t * p.A + a) * p.Nst) + n0 + pl) * S + 4 * (lane >> 2)) = v4;
                    if (!rew_none && (lane >> 2) == 4) {
                        float r = (pl == 0) ? f0 : (pl == 1) ? f1 : (pl == 2) ? f2 : f3;
                        r = r + (v4.y - cur4.y) / 0.01f;               // (nxt[17] - cur[17]) / 0.01
                        r = r - zs[t * QP + pl];
                        total = total + r;
                    }
                    cur4 = v4;
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (v4[r] - e_nm[r]) * e_ni[r];
                    *reinterpret_cast<f32x4*>(xs + (size_t)lane * 4) = o;          // lane = g*4 + pp
                } else if (lane < K0G * 4) {
                    const int tn = (t + 1 < H) ? t + 1 : t;
                    const int ga = (lane >> 2) - SG;
                    *reinterpret_cast<f32x4*>(xs + (size_t)lane * 4) =
                        *reinterpret_cast<const f32x4*>(xa + ((size_t)(tn * AG + ga) * 4 + pl) * 4);
                }
            }
        } else
        // ---- epilogue: thread (feature k, particle pp)
        for (int i = tid; i < QP * (S + U); i += NT) {
            const int k = i / QP, pp = i % QP;
            float v;
            if (k < S) {
                // D fragment of output feature k: lane (k>>2)*4 + pp, register k&3
                const int ln = (k >> 2) * 4 + pp, rg = k & 3;
                float acc = lbias[k];
#pragma unroll
                for (int w = 0; w < NWQ; ++w) acc = acc + part[((size_t)w * 64 + ln) * 4 + rg];
                acc = apply_act_ct<A2>(acc);
                const float dev = normd ? tmean[k] + acc * tstd[k] : acc;
                const float ck = cur[pp * Sp + k];
                v = dev + ck;
                nxt[pp * Sp + k] = v;
                if (q.traj && n0 + pp < p.n_pop) q.traj[((((size_t)t * p.A + a) * p.Nst) + n0 + pp) * S + k] = v;
                if (rew_inline && k == 17) {
                    const float c5 = cur[pp * Sp + 5], c6 = cur[pp * Sp + 6], c7 = cur[pp * Sp + 7];
                    const float* ac = acts + (t * QP + pp) * U;
                    float ss = 0.0f;
                    for (int u = 0; u < U; ++u) ss = ss + ac[u] * ac[u];
                    float r = 0.0f;
                    if (c5 >= 0.2f) r = r + (-10.0f);
                    if (c6 >= 0.0f) r = r + (-10.0f);
                    if (c7 >= 0.0f) r = r + (-10.0f);
                    r = r + (v - ck) / 0.01f;
                    r = r - 0.0f * ss;
                    total = total + r;
                }
            } else {
                const int tn = (t + 1 < H) ? t + 1 : t;
                v = acts[(tn * QP + pp) * U + (k - S)];
            }
            xs[vaddr(k, pp)] = (v - nmean[k]) * ninv[k];
        }
        __syncthreads();
        Q4_MARK(5);
    }
#ifdef BBMPC_KERNEL_DBG
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        printf("[q4dbg] H=%d total %lld | layer0 %lld bar %lld layer1 %lld bar %lld last+bar %lld epi+bar %lld (10ns units, summed over steps)\n",
               H, (long long)wall_clock64() - dbg_start, dbg_acc[0], dbg_acc[1], dbg_acc[2], dbg_acc[3], dbg_acc[4], dbg_acc[5]);
#endif
    const int rt = fast_epi ? tid - 16 : (rew_inline ? tid - 17 * QP : tid);      // particle whose total this thread holds
    if (rt >= 0 && rt < QP) {
        if (!rew_inline) {
            const float* c0 = st + ((H - 1) & 1) * QP * Sp;
            const float* n1 = st + (H & 1) * QP * Sp;
            total = total + reward_generic(p.reward_kind, p.fix_q1 != 0, c0 + rt * Sp, acts + ((H - 1) * QP + rt) * U,
                                           n1 + rt * Sp, S, U);
        }
        const int n = n0 + rt;
        if (n < p.n_pop) {
            if (total != total) total = -1.0e6f;
            if (q.pen) {
                float pen = 0.0f;
                for (int u = 0; u < U; ++u) pen = pen + pens[rt * U + u];
                const float nr = sqrtf(pen);
                pen = nr * nr;
                total = total - pen;
                if (p.penalty_out) p.penalty_out[(size_t)a * p.Nst + n] = pen;
            }
            p.rewards[(size_t)a * p.Nst + n] = total;
        }
    }
}

inline int mlp_q4_lds_floats(int HG, int K0G, int NWQ, int H, int U, int S) {
    const int Sp = (S + 3) & ~3;
    return K0G * 16 + 2 * HG * 16 + NWQ * 256 + 2 * 4 * Sp + ((H * 4 * U + 3) & ~3) + ((4 * U + 3) & ~3) + 2 * (S + U) + 3 * S + 16 +
           H * K0G * 16 + H * 4 + 8;
}

// =================================================================================================
// OPTIONAL bf16-input mode (BBMPC_MLP_BF16 = 1 | 3; never the default: the parity path is fp32 in / fp32 accumulate).
// v_mfma_f32_16x16x16_bf16 moves 4x the K per instruction in half the time of v_mfma_f32_16x16x4_f32
// (tools/microbench/mfma_valu_overlap.hip), so even the split form -- every operand x = hi + lo with hi, lo bf16,
// products hi*hi + hi*lo + lo*hi accumulated in fp32, i.e. ~16 mantissa bits per factor -- needs 24 ns of matrix time per
// 16-deep K tile instead of 61 ns.  NPROD = 1 drops the lo parts (plain bf16 inputs, ~8 mantissa bits).
// Same decomposition as SPEC 1 of rollout_mlp_body (2 hidden layers of equal width <= 256, one output tile per wave,
// weights stationary in VGPRs, last layer K-split): the D fragment of an output tile (rows 4g..4g+3 of column p) is,
// converted in place, exactly the 4-bf16 B operand of the next layer's K tile.  LDS tiles hold (hi | lo) = 16 bytes per
// lane, the size of the fp32 float4 they replace.  Tolerances are stated in tests/test_gpu_mlp.py.
typedef short bf16x4_t __attribute__((ext_vector_type(4)));

struct BfSplit { bf16x4_t hi, lo; };

__device__ __forceinline__ BfSplit bf_split4(const f32x4& v) {
    BfSplit o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t u = __float_as_uint(v[r]);
        const uint32_t h = u & 0xffff0000u;                                  // hi: truncated to bf16
        const float rem = v[r] - __uint_as_float(h);                         // exact
        const uint32_t l = __float_as_uint(rem) + 0x8000u;                   // lo: rounded to bf16
        o.hi[r] = (short)(h >> 16);
        o.lo[r] = (short)(l >> 16);
    }
    return o;
}

template <int NPROD>
__device__ __forceinline__ f32x4 bf_mma(const BfSplit& a, const BfSplit& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.hi, acc, 0, 0, 0);
    if (NPROD == 3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.hi, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.lo, b.hi, acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ BfSplit bf_from_u4(const uint4& w) {
    BfSplit o;
    o.hi[0] = (short)(w.x & 0xffffu); o.hi[1] = (short)(w.x >> 16); o.hi[2] = (short)(w.y & 0xffffu); o.hi[3] = (short)(w.y >> 16);
    o.lo[0] = (short)(w.z & 0xffffu); o.lo[1] = (short)(w.z >> 16); o.lo[2] = (short)(w.w & 0xffffu); o.lo[3] = (short)(w.w >> 16);
    return o;
}
__device__ __forceinline__ uint4 bf_to_u4(const BfSplit& s) {
    uint4 w;
    w.x = ((uint32_t)(unsigned short)s.hi[0]) | ((uint32_t)(unsigned short)s.hi[1] << 16);
    w.y = ((uint32_t)(unsigned short)s.hi[2]) | ((uint32_t)(unsigned short)s.hi[3] << 16);
    w.z = ((uint32_t)(unsigned short)s.lo[0]) | ((uint32_t)(unsigned short)s.lo[1] << 16);
    w.w = ((uint32_t)(unsigned short)s.lo[2]) | ((uint32_t)(unsigned short)s.lo[3] << 16);
    return w;
}

template <int NPROD>
__global__ void k_rollout_mlp_bf16(MlpRolloutArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    const int a = blockIdx.y;
    const int n0 = blockIdx.x * MLP_TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = q.nw, nthr = nw * 64;
    const int S = p.S, U = p.U, H = p.H, L = m.n_layers;
    const int Sp = (S + 3) & ~3;
    const MlpLds lay = mlp_lds_layout(m, H, U, S, nw);
    uint4* xs = reinterpret_cast<uint4*>(smem + lay.xs);          // [IT0][64] (hi|lo)
    uint4* hbuf = reinterpret_cast<uint4*>(smem + lay.actA);      // [HT][64]  (hi|lo)
    float* part = smem + lay.part;
    float* st = smem + lay.st;
    float* acts = smem + lay.acts;
    float* misc = smem + lay.misc;
    const bool normd = m.normalized != 0;
    float* nmean = smem + lay.norm;
    float* ninv = nmean + (S + U);
    float* tmean = ninv + (S + U);
    float* tstd = tmean + S;
    float* lbias = tstd + S;

    constexpr int HTM = 16, IT0M = 2, OTLM = 2;
    const int HT = m.tiles[1], IT0 = m.tiles[0], OTl = m.tiles[L];
    BfSplit w_in[IT0M], w_hid[HTM], w_out[OTLM];
#pragma unroll
    for (int it = 0; it < IT0M; ++it) w_in[it] = bf_from_u4(it < IT0 ? q.wbf[0][((size_t)wave * IT0 + it) * 64 + lane] : make_uint4(0, 0, 0, 0));
#pragma unroll
    for (int it = 0; it < HTM; ++it) w_hid[it] = bf_from_u4(it < HT ? q.wbf[1][((size_t)wave * HT + it) * 64 + lane] : make_uint4(0, 0, 0, 0));
#pragma unroll
    for (int ot = 0; ot < OTLM; ++ot) w_out[ot] = bf_from_u4(ot < OTl ? q.wbf[2][((size_t)ot * HT + wave) * 64 + lane] : make_uint4(0, 0, 0, 0));
    const f32x4 bias0 = *reinterpret_cast<const f32x4*>(m.bpack[0] + ((size_t)wave * 64 + lane) * 4);
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(m.bpack[1] + ((size_t)wave * 64 + lane) * 4);

    mlp_fill_actions<MLP_TP>(q, a, n0, tid, nthr, acts, misc);
    for (int f = tid; f < S + U; f += nthr) {
        const float mu = normd ? (f < S ? m.mean_s[f] : m.mean_a[f - S]) : 0.0f;
        const float sd = normd ? (f < S ? m.std_s[f] : m.std_a[f - S]) : 1.0f;
        nmean[f] = mu;
        ninv[f] = normd ? 1.0f / (sd + 1e-7f) : 1.0f;
        if (f < S) {
            tmean[f] = normd ? m.mean_t[f] : 0.0f;
            tstd[f] = normd ? (m.std_t[f] + 1e-7f) : 1.0f;
            lbias[f] = m.bpack[L - 1][((size_t)(f >> 4) * 64 + ((f & 15) >> 2) * 16) * 4 + (f & 3)];
        }
    }
    for (int i = tid; i < IT0 * 64; i += nthr) xs[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < MLP_TP * S; i += nthr) {
        const int pp = i / S, s = i % S;
        st[pp * Sp + s] = p.state[a * S + s];
    }
    __syncthreads();
    // element (feature f, particle pp) of an input tile: lane ((f&15)>>2)*16 + pp, bf16 slot f&3 of the hi / lo quads
    auto put_x = [&](int f, int pp, float v) {
        const uint32_t u = __float_as_uint(v);
        const uint32_t h = u & 0xffff0000u;
        const uint32_t l = __float_as_uint(v - __uint_as_float(h)) + 0x8000u;
        unsigned short* base = reinterpret_cast<unsigned short*>(xs + ((size_t)(f >> 4) * 64 + ((f & 15) >> 2) * 16 + pp));
        base[f & 3] = (unsigned short)(h >> 16);
        base[4 + (f & 3)] = (unsigned short)(l >> 16);
    };
    for (int i = tid; i < MLP_TP * (S + U); i += nthr) {
        const int f = i / MLP_TP, pp = i % MLP_TP;
        const float v = (f < S) ? st[pp * Sp + f] : acts[(0 * MLP_TP + pp) * U + (f - S)];
        put_x(f, pp, (v - nmean[f]) * ninv[f]);
    }
    __syncthreads();

    float total = 0.0f;
    for (int t = 0; t < H; ++t) {
        float* cur = st + (t & 1) * MLP_TP * Sp;
        float* nxt = st + ((t + 1) & 1) * MLP_TP * Sp;
        // ---- layer 0
        f32x4 acc = bias0;
#pragma unroll
        for (int it = 0; it < IT0M; ++it)
            if (it < IT0) acc = bf_mma<NPROD>(w_in[it], bf_from_u4(xs[(size_t)it * 64 + lane]), acc);
        acc.x = apply_act(acc.x, m.act[0]); acc.y = apply_act(acc.y, m.act[0]);
        acc.z = apply_act(acc.z, m.act[0]); acc.w = apply_act(acc.w, m.act[0]);
        hbuf[(size_t)wave * 64 + lane] = bf_to_u4(bf_split4(acc));               // all-gather through LDS
        __syncthreads();
        // ---- layer 1
        acc = bias1;
#pragma unroll
        for (int it = 0; it < HTM; ++it)
            if (it < HT) acc = bf_mma<NPROD>(w_hid[it], bf_from_u4(hbuf[(size_t)it * 64 + lane]), acc);
        acc.x = apply_act(acc.x, m.act[1]); acc.y = apply_act(acc.y, m.act[1]);
        acc.z = apply_act(acc.z, m.act[1]); acc.w = apply_act(acc.w, m.act[1]);
        // ---- last layer, K split: my own hidden tile (converted in registers) times my slab of W_last
        {
            const BfSplit hb = bf_split4(acc);
#pragma unroll
            for (int ot = 0; ot < OTLM; ++ot) {
                if (ot < OTl) {
                    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
                    o = bf_mma<NPROD>(w_out[ot], hb, o);
                    *reinterpret_cast<f32x4*>(part + (((size_t)wave * OTl + ot) * 64 + lane) * 4) = o;
                }
            }
        }
        __syncthreads();
        // ---- epilogue (fp32): reduce partials, bias, last activation, de-normalise, residual; stage step t+1's input
        const int nwp = min(nw, HT);
        for (int i = tid; i < MLP_TP * (S + U); i += nthr) {
            const int f = i / MLP_TP, pp = i % MLP_TP;
            float v;
            if (f < S) {
                const int ot = f >> 4, ln = ((f & 15) >> 2) * 16 + pp, rg = f & 3;
                const float* pp0 = part + (((size_t)ot) * 64 + ln) * 4 + rg;
                float s_ = lbias[f];
#pragma unroll 4
                for (int w = 0; w < nwp; ++w) s_ = s_ + pp0[(size_t)w * OTl * 256];
                s_ = apply_act(s_, m.act[L - 1]);
                const float dev = normd ? tmean[f] + s_ * tstd[f] : s_;
                const float ns = dev + cur[pp * Sp + f];
                nxt[pp * Sp + f] = ns;
                v = ns;
            } else {
                const int tn = (t + 1 < H) ? t + 1 : t;
                v = acts[(tn * MLP_TP + pp) * U + (f - S)];
            }
            put_x(f, pp, (v - nmean[f]) * ninv[f]);
        }
        __syncthreads();
        if (tid < MLP_TP) {
            const float r = reward_generic(p.reward_kind, p.fix_q1 != 0, cur + tid * Sp, acts + (t * MLP_TP + tid) * U,
                                           nxt + tid * Sp, S, U);
            total = total + r;
        }
    }
    __syncthreads();
    if (tid < MLP_TP) {
        const int n = n0 + tid;
        if (n < p.n_pop) {
            if (total != total) total = -1.0e6f;
            if (q.pen) {
                float pen = 0.0f;
                for (int u = 0; u < U; ++u) pen = pen + misc[tid * U + u];
                const float nr = sqrtf(pen);
                pen = nr * nr;
                total = total - pen;
                if (p.penalty_out) p.penalty_out[(size_t)a * p.Nst + n] = pen;
            }
            p.rewards[(size_t)a * p.Nst + n] = total;
        }
    }
}

}  // namespace bbmpc
