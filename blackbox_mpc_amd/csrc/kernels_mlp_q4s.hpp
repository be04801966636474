// Quad mode ("q4s"): the learned-model rollout for populations too small to give every CU a 16-particle tile -- 4 particles
// per workgroup on v_mfma_f32_4x4x1_16b_f32, four equal waves (one per SIMD), ALL weights stationary in registers, the state
// in registers, two barriers per model step.  The fused path (reference files under blackbox_mpc/):
//   SystemDynamicsHandler.process_input / process_output   dynamics_handlers/system_dynamics_handler.py:97-161
//   DeterministicMLP.__call__                              dynamics_functions/deterministic_mlp.py:27-51
//   reward_function                                        tutorials/mujoco/cost_func.py:5-22
//   DeterministicTrajectoryEvaluator.__call__              trajectory_evaluators/deterministic.py:26-77
//
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4 outer products (blocks of four lanes: A = 4 output features x 1 k, B = 1 k x
// 4 particles), and nothing says the 16 blocks have to work on the same k or the same features:
//  * layer 0: block b of wave w produces hidden quad 13w + b from the whole input (26 MFMAs per wave).  Every wave holds the
//    state: a lane keeps the normalised input groups "own quad" and 4 of its particle and takes the other three state groups
//    from its row neighbours by DPP rotation -- block g multiplies group (g - j) mod 4 in round j, its stationary A operands
//    are loaded in that order once.  The action part (independent of the state) is issued one step ahead.  h0 crosses LDS
//    behind the first barrier;
//  * layer 1 runs as "jobs" of 16 output features: block (row r, quad g) = features 16j + 4g .. +3 over the k slice of
//    16-lane row r (13 k groups), three jobs per wave sharing ONE set of B operands, the K split reduced in registers
//    (v_permlane32_swap / v_permlane16_swap as a reduce-scatter: 4 registers -> 1 in 6 instructions), after which lane
//    (row, quad g, particle p) holds the activation of feature 16*job + 4g + pr(row) -- which IS a B operand of the
//    instruction: one k per block, the four particles in the block's four lanes;
//  * so the last layer takes it from there: per job four MFMAs for output features 0..15 (block g produces output quad g;
//    round j multiplies the value of block g - j, fetched by a DPP row rotation, so that every block meets every k of its
//    row) and one for features 16..19 (every block its own k): 17 MFMAs per wave, no h1 in LDS, and the K split of the last
//    layer is across the WAVES' own hidden features -- what crosses LDS behind the second barrier is 80 partial sums per wave;
//  * every wave then adds the four partials and runs the epilogue (bias, de-normalisation, residual, normalisation,
//    all-gather by lane swaps) redundantly: 1-2 values per lane;
//  * the hidden features 192..199 (two quads nobody's three jobs cover) are one more chain on waves 0 and 1: block b takes
//    k = b, b + 16, ... (13 MFMAs, one riding in each round of the three jobs), all-reduced over the 16 blocks.
// Per wave and model step 26 + 169 + 17 = 212 MFMAs, 13 + 13 LDS reads behind the first barrier and 8 behind the second.
// Rounds 3-5 ran this shape as k_rollout_mlp_q4r (three state waves + a helper, every state wave computing the WHOLE 200 -> 20
// layer from a copy of h1 in LDS: 260 MFMAs per state wave and step, three barriers): 61.4 us per 1000 x 30-step launch against
// 57.0 us here (profiles/r5_cfg4pi2.md, r6_cfg4pi2.md; profiles/NOTES_r6.md has what was measured on the way).
// Requires dim_S <= 20 (five state groups: four rotate inside a row, the fifth is replicated; a shorter state runs padded: zero
// rows in the layer-0 pack wq4s0, zero columns in the last layer's, neutral constants), dim_U <= 8, two hidden layers of 200
// units (HG = 50) or of 256 (HG = 64: four jobs per wave, no tail chain), the HalfCheetah reward or none (a user function scores
// the recorded trajectory); activations at compile time for the tanh / relu networks, at run time (ACT_RT) for the rest;
// everything else keeps k_rollout_mlp_q4 or the 16-particle tilings.
#pragma once
#include "kernels_mlp.hpp"

namespace bbmpc {

constexpr int Q4S_MAX_ACTION_PAIRS = 512;               // (particle, 4 consecutive action-sequence elements) pairs a workgroup's threads keep in registers across the operand loads

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    // all lanes are sources here (row rotations), so no "old" value has to be materialised (bound_ctrl)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128, DPP_ROW_ROR12 = 0x12c;   // dst[i] = src[(i - n) & 15] in each 16-lane row
// a = [a.lo32 | b.lo32], b = [a.hi32 | b.hi32]              (tools/microbench/lane_ops.hip prints the maps)
__device__ __forceinline__ void swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
    a = __int_as_float(r[0]); b = __int_as_float(r[1]);
}
// rows of 16 lanes: a = [a.r0, b.r0, a.r2, b.r2], b = [a.r1, b.r1, a.r3, b.r3]
__device__ __forceinline__ void swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
    a = __int_as_float(r[0]); b = __int_as_float(r[1]);
}
// sum over the four 16-lane rows of four registers at once: row r of the result holds the total of v[{0,2,1,3}[r]]
__device__ __forceinline__ float rows_reduce_scatter(f32x4 v) {
    float a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
    swap32(a0, a1);
    swap32(a2, a3);
    float t01 = a0 + a1, t23 = a2 + a3;
    swap16(t01, t23);
    return t01 + t23;
}
// inverse: row r holds element {0,2,1,3}[r]; afterwards every row holds all four
__device__ __forceinline__ f32x4 rows_all_gather(float z) {
    float v0 = z, v1 = z;
    swap16(v0, v1);                 // v0 = [z0 z0 z2 z2], v1 = [z1 z1 z3 z3]   (z_r = row r's value)
    float c0 = v0, c1 = v1;
    swap32(v0, c0);                 // v0 = z0 everywhere, c0 = z2 everywhere
    swap32(v1, c1);                 // v1 = z1,            c1 = z3
    f32x4 o;
    o.x = v0; o.y = c0; o.z = v1; o.w = c1;       // rows hold elements 0, 2, 1, 3: z0 = e0, z2 = e1, z1 = e2, z3 = e3
    return o;
}
// ---- MFMAs through inline asm: accumulators stay in VGPRs (the compiler gives builtin MFMAs AccVGPR destinations as
// soon as the kernel touches AccVGPRs at all and then pays a v_accvgpr_read per result register), the stationary A
// operand comes from a VGPR ("v") or straight from an AccVGPR ("a"), and the first MFMA of a chain takes its C operand
// from the bias registers (or the inline constant 0) instead of a copy.  The compiler neither sees the instruction nor
// pads its hazards (cdna_hip_programming.md 5.7): every sequence below
//   * starts behind a sched_barrier + mfma4_operands_settled() (VALU-written operands -> MFMA read: 2 wait states),
//   * keeps consecutive MFMAs on one accumulator >= 3 issues apart (three or four chains, round robin),
//   * ends with mfma4_results_ready() (2-pass MFMA result -> VALU read: 4 wait states), which also ties the accumulators.
#define BBMPC_MFMA4 "v_mfma_f32_4x4x1_16b_f32 "
__device__ __forceinline__ void mfma4_v(f32x4& acc, float a, float b) {
    asm volatile(BBMPC_MFMA4 "%0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma4_v_c(f32x4& acc, float a, float b, const f32x4& c) {
    asm volatile(BBMPC_MFMA4 "%0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma4_v_0(f32x4& acc, float a, float b) {
    asm volatile(BBMPC_MFMA4 "%0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma4_a(f32x4& acc, float a_in_agpr, float b) {
    asm volatile(BBMPC_MFMA4 "%0, %1, %2, %0" : "+v"(acc) : "a"(a_in_agpr), "v"(b));
}
__device__ __forceinline__ void mfma4_a_0(f32x4& acc, float a_in_agpr, float b) {
    asm volatile(BBMPC_MFMA4 "%0, %1, %2, 0" : "=&v"(acc) : "a"(a_in_agpr), "v"(b));
}
__device__ __forceinline__ void mfma4_operands_settled() { asm volatile("s_nop 1"); }
__device__ __forceinline__ void mfma4_results_ready(f32x4& c0, f32x4& c1, f32x4& c2) {
    asm volatile("s_nop 4" : "+v"(c0), "+v"(c1), "+v"(c2));
}
__device__ __forceinline__ void mfma4_results_ready(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    asm volatile("s_nop 4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
}
__device__ __forceinline__ void mfma4_results_ready(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, f32x4& c4) {
    asm volatile("s_nop 4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4));
}

// One k-group of three layer-1 jobs + one MFMA of the tail chain = 13 MFMAs in ONE asm statement (four chains, round robin):
// between separate statements the compiler pads every re-use of an accumulator with an s_nop it cannot know to be unnecessary
__device__ __forceinline__ void mfma4_a_round3t(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& ct, const float* w0, const float* w1, const float* w2,
                                                const f32x4& b, float wt, float bt) {
    asm volatile(BBMPC_MFMA4 "%0, %4, %16, %0\n\t" BBMPC_MFMA4 "%1, %8, %16, %1\n\t" BBMPC_MFMA4 "%2, %12, %16, %2\n\t"
                 BBMPC_MFMA4 "%0, %5, %17, %0\n\t" BBMPC_MFMA4 "%1, %9, %17, %1\n\t" BBMPC_MFMA4 "%2, %13, %17, %2\n\t"
                 BBMPC_MFMA4 "%0, %6, %18, %0\n\t" BBMPC_MFMA4 "%1, %10, %18, %1\n\t" BBMPC_MFMA4 "%2, %14, %18, %2\n\t"
                 BBMPC_MFMA4 "%0, %7, %19, %0\n\t" BBMPC_MFMA4 "%1, %11, %19, %1\n\t" BBMPC_MFMA4 "%2, %15, %19, %2\n\t"
                 BBMPC_MFMA4 "%3, %20, %21, %3"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(ct)
                 : "a"(w0[0]), "a"(w0[1]), "a"(w0[2]), "a"(w0[3]), "a"(w1[0]), "a"(w1[1]), "a"(w1[2]), "a"(w1[3]),
                   "a"(w2[0]), "a"(w2[1]), "a"(w2[2]), "a"(w2[3]), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "a"(wt), "v"(bt));
}
__device__ __forceinline__ void mfma4_a_round3t_first(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& ct, const float* w0, const float* w1, const float* w2,
                                                      const f32x4& b, float wt, float bt) {
    asm volatile(BBMPC_MFMA4 "%0, %4, %16, 0\n\t"  BBMPC_MFMA4 "%1, %8, %16, 0\n\t"  BBMPC_MFMA4 "%2, %12, %16, 0\n\t"
                 BBMPC_MFMA4 "%0, %5, %17, %0\n\t" BBMPC_MFMA4 "%1, %9, %17, %1\n\t" BBMPC_MFMA4 "%2, %13, %17, %2\n\t"
                 BBMPC_MFMA4 "%0, %6, %18, %0\n\t" BBMPC_MFMA4 "%1, %10, %18, %1\n\t" BBMPC_MFMA4 "%2, %14, %18, %2\n\t"
                 BBMPC_MFMA4 "%0, %7, %19, %0\n\t" BBMPC_MFMA4 "%1, %11, %19, %1\n\t" BBMPC_MFMA4 "%2, %15, %19, %2\n\t"
                 BBMPC_MFMA4 "%3, %20, %21, 0"
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(ct)
                 : "a"(w0[0]), "a"(w0[1]), "a"(w0[2]), "a"(w0[3]), "a"(w1[0]), "a"(w1[1]), "a"(w1[2]), "a"(w1[3]),
                   "a"(w2[0]), "a"(w2[1]), "a"(w2[2]), "a"(w2[3]), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "a"(wt), "v"(bt));
}

// One k-group of FOUR layer-1 jobs (256 hidden units: no tail chain) = 16 MFMAs, four chains round robin
__device__ __forceinline__ void mfma4_a_round4(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const float* w0, const float* w1, const float* w2,
                                               const float* w3, const f32x4& b) {
    asm volatile(BBMPC_MFMA4 "%0, %4, %20, %0\n\t" BBMPC_MFMA4 "%1, %8, %20, %1\n\t" BBMPC_MFMA4 "%2, %12, %20, %2\n\t" BBMPC_MFMA4 "%3, %16, %20, %3\n\t"
                 BBMPC_MFMA4 "%0, %5, %21, %0\n\t" BBMPC_MFMA4 "%1, %9, %21, %1\n\t" BBMPC_MFMA4 "%2, %13, %21, %2\n\t" BBMPC_MFMA4 "%3, %17, %21, %3\n\t"
                 BBMPC_MFMA4 "%0, %6, %22, %0\n\t" BBMPC_MFMA4 "%1, %10, %22, %1\n\t" BBMPC_MFMA4 "%2, %14, %22, %2\n\t" BBMPC_MFMA4 "%3, %18, %22, %3\n\t"
                 BBMPC_MFMA4 "%0, %7, %23, %0\n\t" BBMPC_MFMA4 "%1, %11, %23, %1\n\t" BBMPC_MFMA4 "%2, %15, %23, %2\n\t" BBMPC_MFMA4 "%3, %19, %23, %3"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                 : "a"(w0[0]), "a"(w0[1]), "a"(w0[2]), "a"(w0[3]), "a"(w1[0]), "a"(w1[1]), "a"(w1[2]), "a"(w1[3]),
                   "a"(w2[0]), "a"(w2[1]), "a"(w2[2]), "a"(w2[3]), "a"(w3[0]), "a"(w3[1]), "a"(w3[2]), "a"(w3[3]),
                   "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
}
__device__ __forceinline__ void mfma4_a_round4_first(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const float* w0, const float* w1, const float* w2,
                                                     const float* w3, const f32x4& b) {
    asm volatile(BBMPC_MFMA4 "%0, %4, %20, 0\n\t"  BBMPC_MFMA4 "%1, %8, %20, 0\n\t"  BBMPC_MFMA4 "%2, %12, %20, 0\n\t"  BBMPC_MFMA4 "%3, %16, %20, 0\n\t"
                 BBMPC_MFMA4 "%0, %5, %21, %0\n\t" BBMPC_MFMA4 "%1, %9, %21, %1\n\t" BBMPC_MFMA4 "%2, %13, %21, %2\n\t" BBMPC_MFMA4 "%3, %17, %21, %3\n\t"
                 BBMPC_MFMA4 "%0, %6, %22, %0\n\t" BBMPC_MFMA4 "%1, %10, %22, %1\n\t" BBMPC_MFMA4 "%2, %14, %22, %2\n\t" BBMPC_MFMA4 "%3, %18, %22, %3\n\t"
                 BBMPC_MFMA4 "%0, %7, %23, %0\n\t" BBMPC_MFMA4 "%1, %11, %23, %1\n\t" BBMPC_MFMA4 "%2, %15, %23, %2\n\t" BBMPC_MFMA4 "%3, %19, %23, %3"
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
                 : "a"(w0[0]), "a"(w0[1]), "a"(w0[2]), "a"(w0[3]), "a"(w1[0]), "a"(w1[1]), "a"(w1[2]), "a"(w1[3]),
                   "a"(w2[0]), "a"(w2[1]), "a"(w2[2]), "a"(w2[3]), "a"(w3[0]), "a"(w3[1]), "a"(w3[2]), "a"(w3[3]),
                   "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
}

constexpr int ACT_RT = -1;                                // activation taken from MlpDesc::act at run time (a scalar branch per use)
template <int ACT>
__device__ __forceinline__ float apply_act_q4s(float x, int rt) {
    if constexpr (ACT == ACT_RT) return apply_act(x, rt);
    else return apply_act_ct<ACT>(x);
}

template <int HG, int K0G, int A0, int A1, int A2, int NE>
__global__ __launch_bounds__(256, 1) void k_rollout_mlp_q4s(MlpRolloutArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    constexpr int NT = 256, QP = 4, NW = 4;
    constexpr int SG = 5, AG = K0G - SG;                 // state / action input groups (dim_S == 20)
    constexpr int KA = (HG + 3) / 4;                     // k groups per 16-lane row when K is split over the rows
    constexpr int HP = 64;                               // h0 is padded to 64 groups (zero): clamp-free operand addresses
    constexpr int ZROW = HP - 1;                         // a zero group of the packed operands (bbmpc_set_mlp pads the k/4 axis to 64)
    constexpr int NJ = HG / (4 * NW);                    // layer-1 jobs of 16 output features per wave (3: 12 jobs = 192 features)
    constexpr int TQ = HG - 4 * NW * NJ;                 // hidden quads left over for the tail chains (2: waves 0 and 1)
    constexpr int KT = (HG + 3) / 4;                     // tail chain: MFMAs (block (row, g) takes k = 4*(row + 4m) + g, m < KT)
    constexpr int Q0W = (HG + NW - 1) / NW;              // layer 0: hidden quads per wave (13: blocks 0..12)
    static_assert(HG <= HP && KA * 4 <= HP && K0G > SG && (TQ == 0 || (HG < HP && 3 + 4 * (KT - 1) < HP)), "padding / input groups");
    static_assert(NT % (16 * AG) == 0 && ((NJ == 3 && TQ >= 0 && TQ <= NW) || (NJ == 4 && TQ == 0)) && KT == KA && Q0W <= 16 && AG == 2,
                  "written for 200 hidden units (three jobs per wave + tail chains) and 256 (four jobs), 20 + <= 8 inputs");
    const int a = blockIdx.y, n0 = xcd_tile(blockIdx.x, gridDim.x, blockIdx.y) * QP;     // an XCD's workgroups: contiguous particles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
    const int S = p.S, U = p.U, H = p.H;
    const bool normd = m.normalized != 0;
    const int rt0 = m.act[0], rt1 = m.act[1], rt2 = m.act[2];
    const int row = lane >> 4, blk = lane >> 2, fgq = blk & 3, pl = lane & 3;
    const int pr = ((row & 1) << 1) | (row >> 1);        // element of a reduce-scattered float4 this row holds
    // ---- LDS: h0[HP][4][4] | zx[4 waves][64 + 16] | acts[H][4][U] | pen[4U] | xa[H][AG][4][4] | zs[H][4] | rwd[H][4][4] | d2s[H][4][U]
    float* h0 = smem;
    float* zx = h0 + HP * 16;                             // the waves' partial sums of the last layer: [wave][lane] features 0..15, [wave][64 + row*4 + particle] features 16..19
    float* acts = zx + NW * 80;
    float* pens = acts + ((H * QP * U + 3) & ~3);
    float* xa = pens + ((QP * U + 3) & ~3);
    float* zs = xa + H * AG * 16;
    float* rwd = zs + H * QP;                             // per step and particle: (progress difference, flag 5, flag 6, flag 7)
    float* d2s = rwd + H * QP * 4;                        // [H][4][U] squared clip distances, summed per (particle, u) in step order at the end
#ifdef BBMPC_KERNEL_DBG
    // shader-clock phase counters; the scheduling fences keep the phases' instructions on their side of a mark
    long long dbg_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dbg_t0 = (long long)clock64();
#if BBMPC_KERNEL_DBG >= 2
#define Q4S_MARK(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = (long long)clock64(); dbg_acc[i] += now_ - dbg_t0; dbg_t0 = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define Q4S_MARK(i) do { (void)dbg_acc; (void)dbg_t0; } while (0)       // -DBBMPC_KERNEL_DBG=1: the loop's cycles and wall time only
#endif
#else
#define Q4S_MARK(i) do {} while (0)
#endif
    // ---- prologue.  Vector memory returns in order, so the SMALL loads (constants, start state, the sources of the 4
    // particles' action block) are issued first, then the ~260 KB of stationary operands, and only then is anything
    // consumed: the small results arrive first and are worked on while the operands stream in (the two waits used to
    // add up, 3.2 + 4.1 us of a 70 us kernel).
    // The action block is handled in (particle, 4 consecutive j) pairs, NE per thread (4 * ceil(H*U / 4) <= 256 * NE): the
    // four elements of a pair share one Philox block (rng.hpp: rng_block is keyed by j >> 2), so a thread draws once,
    // not four times -- 120 VALU instructions each at a lone wave's issue rate were a good microsecond of the prologue.
    const int HU = p.HU, a_pairs = QP * ((HU + 3) >> 2);
    // the constants of the two output features this lane finishes (fA = 4*fgq + pr, fB = 16 + pr) and of the action element
    // this thread normalises below (u_t: the same in every pass of that loop), straight into registers: staged through LDS
    // (round 3-5) they cost a barrier and ~40 dependent LDS reads at a lone wave's latency, a microsecond of every launch
    const int fA = 4 * fgq + pr, fB = 16 + pr, cs = min(tid, S - 1);
    const int u_t = ((tid >> 4) % AG) * 4 + (tid & 3);
    const bool u_on = u_t < U;
    // (dim_S < 20: the state features past dim_S do not exist -- zero weights in the padded operand packs, neutral constants
    // and a zero start state keep them at exactly 0 through every step)
    const bool onA = fA < S, onB = fB < S, nA = normd && onA, nB = normd && onB;
    const float g_msA = (nA ? m.mean_s : p.state)[nA ? fA : 0], g_msB = (nB ? m.mean_s : p.state)[nB ? fB : 0];
    const float g_ssA = (nA ? m.std_s : p.state)[nA ? fA : 0], g_ssB = (nB ? m.std_s : p.state)[nB ? fB : 0];
    const float g_mtA = (nA ? m.mean_t : p.state)[nA ? fA : 0], g_mtB = (nB ? m.mean_t : p.state)[nB ? fB : 0];
    const float g_stA = (nA ? m.std_t : p.state)[nA ? fA : 0], g_stB = (nB ? m.std_t : p.state)[nB ? fB : 0];
    const float g_mau = (normd ? m.mean_a : p.state)[(normd && u_on) ? u_t : 0], g_sau = (normd ? m.std_a : p.state)[(normd && u_on) ? u_t : 0];
    const float g_lbA = (onA ? q.braw[2] : p.state)[onA ? fA : 0], g_lbB = (onB ? q.braw[2] : p.state)[onB ? fB : 0];
    const float g_cA = p.state[onA ? a * S + fA : 0], g_cB = p.state[onB ? a * S + fB : 0];
    const float c_st = p.state[a * S + cs];
    // (branch-free: a load under a branch makes the compiler wait for everything in flight at the join, so elements that
    // do not exist / sources a mode does not have read word 0 of the state instead)
    const bool m_ref = q.mode == SRC_REF, m_buf = q.mode == SRC_BUF, m_uni = q.mode == SRC_UNIFORM;
    const bool has_raw = m_ref || m_buf || p.inj != nullptr, has_dist = !m_ref && !m_buf && !m_uni, has_bounds = q.pen || m_uni;
    const float* raw_base = m_ref ? p.seq : m_buf ? p.cand : (p.inj ? p.inj : p.state);
    const float* sg_base = has_dist ? p.sigma : p.state;
    const float* mn_base = has_dist ? p.mean : p.state;
    const float* lo_base = has_bounds ? p.lo : p.state;
    const float* hi_base = has_bounds ? p.hi : p.state;
    const bool has_rng = !has_raw;                         // draws made here (rng.hpp counters), SRC_UNIFORM or truncated normal
    const RngKey key_now = rng_key_now(p.key);               // (the control step from memory when the launch is a graph node)
    float a_raw[NE][4], a_sg[NE][4], a_mn[NE][4], a_lo[NE][4], a_hi[NE][4], a_f[NE][4], a_tq[NE][4];
    int a_n[NE], a_j0[NE], a_tu[NE];                     // particle (-1: no such pair / particle), first j, (t << 8) | u of the first element
    // (i) what does not wait for the generator: the elements' distribution and bounds
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int pi = tid + i * NT;
        const int pp = pi & (QP - 1), j0 = (pi >> 2) << 2;
        const int n = n0 + pp;
        const bool pvalid = pi < a_pairs && n < p.n_pop;
        const int t0 = j0 / U, u0 = j0 - t0 * U;
        a_n[i] = pvalid ? n : -1; a_j0[i] = j0; a_tu[i] = (t0 << 8) | u0;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int j = j0 + l;
            int u = u0 + l;
#pragma unroll
            for (int k = 0; k < 3; ++k) u = (u >= U) ? u - U : u;
            const bool valid = pvalid && j < HU;
            const int di = (valid && has_dist) ? a * HU + j : 0;
            a_sg[i][l] = sg_base[di]; a_mn[i][l] = mn_base[di];
            const int ui = (valid && has_bounds) ? u : 0;
            a_lo[i][l] = lo_base[ui]; a_hi[i][l] = hi_base[ui];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    Q4S_MARK(11);
    // ---- (ii) stationary A operands (packed [k/4][Mp][4] by bbmpc_set_mlp): ~215 KB per workgroup, the first 33 loads.  Operands a block or wave
    // does not use are read from a zero row (one cached 1 KB line) instead of being branched around: a load under a branch
    // is followed by a wait for everything issued so far.
    const int M1 = m.dims[1], M3 = m.dims[3];
    const int Mp1 = (M1 + 63) & ~63, Mp3 = (M3 + 63) & ~63;
    const float4* __restrict__ Q0 = reinterpret_cast<const float4*>(q.wq4s0);          // layer 0 with the state rows padded to 20 (bbmpc_set_mlp)
    const float4* __restrict__ Q1 = reinterpret_cast<const float4*>(q.wq4[1]);
    const float* __restrict__ Q1f = q.wq4[1];
    const float* __restrict__ Q2f = q.wq4[2];
    // layer 0: block b of wave w produces hidden quad 13w + b (b < 13, quad < HG); D register i = feature 4*quad + i
    const int q0 = Q0W * wave + blk;
    const bool l0_on = blk < Q0W && q0 < HG;
    const int f0 = l0_on ? 4 * q0 + (lane & 3) : 0;      // hidden feature of this lane's layer-0 A operands
    float wA0r[16], wA0b[4], wA0a[AG * 4], wJ[NJ][KA * 4], wT[KT], wLA[NJ][4], wLB[NJ], wLTA = 0.0f, wLTB = 0.0f;
#pragma unroll
    for (int g = 0; g < AG; ++g) {
        const float4 v = Q0[(size_t)(l0_on ? SG + g : ZROW) * Mp1 + f0];
        wA0a[4 * g + 0] = v.x; wA0a[4 * g + 1] = v.y; wA0a[4 * g + 2] = v.z; wA0a[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                        // round j of layer 0: my block multiplies state group (fgq - j) & 3
        const float4 v = Q0[(size_t)(l0_on ? ((fgq - j) & 3) : ZROW) * Mp1 + f0];
        wA0r[4 * j + 0] = v.x; wA0r[4 * j + 1] = v.y; wA0r[4 * j + 2] = v.z; wA0r[4 * j + 3] = v.w;
    }
    {
        const float4 v = Q0[(size_t)(l0_on ? 4 : ZROW) * Mp1 + f0];
        wA0b[0] = v.x; wA0b[1] = v.y; wA0b[2] = v.z; wA0b[3] = v.w;
    }
    // layer 1, K split over the rows: row r takes k groups [KA*r, KA*r + cnt)
    const int startA = KA * row;
    const int cntA = min(KA, HG - startA);
#pragma unroll
    for (int jj = 0; jj < NJ - 1; ++jj) {                // (ii) job = wave + 4*jj: output features 16*job + (lane & 15)
        const int o = 16 * (wave + NW * jj) + (lane & 15);
#pragma unroll
        for (int c = 0; c < KA; ++c) {
            const float4 v = Q1[(size_t)(c < cntA ? startA + c : ZROW) * Mp1 + o];
            wJ[jj][4 * c + 0] = v.x; wJ[jj][4 * c + 1] = v.y; wJ[jj][4 * c + 2] = v.z; wJ[jj][4 * c + 3] = v.w;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    Q4S_MARK(17);
    // (iii) the draws, while the first operands are on their way (215 KB per workgroup from every workgroup at once keep the
    // L2s busy for ~4 us: the generator's ~250 instructions used to run BEFORE the first operand load was issued).
    // word_to_trunc_normal (rng.hpp) split in two: the table entry is loaded here, the interpolation happens with the
    // other small results.  One unconditional pair of loads per element whatever the mode (a load inside a branch costs a
    // wait at the join).
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int pi = tid + i * NT;
        const int j0 = a_j0[i], n = n0 + (pi & (QP - 1));
        const bool pvalid = a_n[i] >= 0;
        U4 blk4 = {0u, 0u, 0u, 0u};
        if (has_rng) blk4 = rng_block(key_now, p.stream, p.iter, (uint32_t)(n + p.pop_offset), (uint32_t)(p.agent_offset + a), (uint32_t)j0);
        const bool tn = has_rng && !m_uni;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int j = j0 + l;
            const bool valid = pvalid && j < HU;
            const uint32_t w = l == 0 ? blk4.x : (l == 1 ? blk4.y : (l == 2 ? blk4.z : blk4.w));      // pick_word(blk4, j)
            const uint32_t v = w >> 9;
            a_f[i][l] = m_uni ? word_to_uniform(w)
                              : ((float)(v & ((1u << (23 - TNQ_BITS)) - 1u)) + 0.5f) * (1.0f / (float)(1u << (23 - TNQ_BITS)));
            const size_t ri = m_ref ? ((size_t)n * p.A + a) * HU + j : ((size_t)a * HU + j) * p.Nst + n;
            const float* tqp = reinterpret_cast<const float*>(g_tnq) + 2 * (v >> (23 - TNQ_BITS));
            const float* rp = tn ? tqp : raw_base + ((valid && has_raw) ? ri : (size_t)0);
            const float* tp = tn ? tqp + 1 : p.state;
            a_raw[i][l] = *rp; a_tq[i][l] = *tp;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    Q4S_MARK(18);
#pragma unroll
    for (int jj = NJ - 1; jj < NJ; ++jj) {              // (iv) the rest of the operands
        const int o = 16 * (wave + NW * jj) + (lane & 15);
#pragma unroll
        for (int c = 0; c < KA; ++c) {
            const float4 v = Q1[(size_t)(c < cntA ? startA + c : ZROW) * Mp1 + o];
            wJ[jj][4 * c + 0] = v.x; wJ[jj][4 * c + 1] = v.y; wJ[jj][4 * c + 2] = v.z; wJ[jj][4 * c + 3] = v.w;
        }
    }
    // the tail chain (hidden quad 4*NW*NJ + wave on the first TQ waves): MFMA m of block (row, g) takes k = 4*(row + 4m) + g
    // (groups 50, 51 of the packed operands and of h0 are zero), output feature 4*(4*NW*NJ + wave) + (lane & 3)
    const bool tail_on = wave < TQ;
    if constexpr (TQ > 0) {
#pragma unroll
        for (int mm = 0; mm < KT; ++mm)
            wT[mm] = Q1f[((size_t)(tail_on ? row + 4 * mm : ZROW) * Mp1 + 16 * NW * NJ + 4 * wave + (lane & 3)) * 4 + fgq];
    }
    // the last layer from registers.  After layer 1 lane (row, g, particle) holds hidden feature 16*job + 4g + pr: round j of
    // set A multiplies the value of block (g - j) & 3, i.e. k = 16*job + 4*((g - j) & 3) + pr, into output quad g; set B
    // multiplies the block's own k into output features 16..19.  The tail value (k = 4*(48 + wave) + pr, the same in the four
    // blocks of a row) goes into output quad g on every block and into features 16..19 on block 0 of the row only.
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int job = wave + NW * jj;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wLA[jj][j] = Q2f[((size_t)(4 * job + ((fgq - j) & 3)) * Mp3 + 4 * fgq + (lane & 3)) * 4 + pr];
        wLB[jj] = Q2f[((size_t)(4 * job + fgq) * Mp3 + 16 + (lane & 3)) * 4 + pr];
    }
    if constexpr (TQ > 0) {
        wLTA = Q2f[((size_t)(tail_on ? 4 * NW * NJ + wave : ZROW) * Mp3 + 4 * fgq + (lane & 3)) * 4 + pr];
        wLTB = Q2f[((size_t)((tail_on && fgq == 0) ? 4 * NW * NJ + wave : ZROW) * Mp3 + 16 + (lane & 3)) * 4 + pr];
    }
    // biases: layer 0 enters as the C operand of a chain's first MFMA (my 4 D rows: features 4*quad + r); the K-split
    // products get theirs after the reduction, where a lane holds ONE feature
    f32x4 b0;
    {
        const int fb = l0_on ? 4 * q0 : 0;
        b0.x = q.braw[0][fb + 0]; b0.y = q.braw[0][fb + 1]; b0.z = q.braw[0][fb + 2]; b0.w = q.braw[0][fb + 3];
    }
    float b1s[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) b1s[jj] = q.braw[1][16 * (wave + NW * jj) + 4 * fgq + pr];
    const float b1t = TQ > 0 ? q.braw[1][min(16 * NW * NJ + 4 * wave + pr, M1 - 1)] : 0.0f;                   // tail: feature 4*(48 + wave) + pr (waves < TQ)
    __builtin_amdgcn_sched_barrier(0);
    Q4S_MARK(5);

    // ---- the small results: constants to LDS; the action block (mlp_fill_actions' arithmetic, every thread clipping its
    // own elements; xa is scratch for the squared clip distances, summed per (particle, u) in step order below)
    if (q.state_copy && blockIdx.x == 0 && tid < S) q.state_copy[a * S + tid] = c_st;
    const float nmA = nA ? g_msA : 0.0f, nmB = nB ? g_msB : 0.0f;                             // un-normalised: (x - 0) * 1, 0 + z * 1 = z exactly
    const float niA = nA ? 1.0f / (g_ssA + 1e-7f) : 1.0f, niB = nB ? 1.0f / (g_ssB + 1e-7f) : 1.0f;
    const float tmA = nA ? g_mtA : 0.0f, tmB = nB ? g_mtB : 0.0f;
    const float tsA = nA ? (g_stA + 1e-7f) : 1.0f, tsB = nB ? (g_stB + 1e-7f) : 1.0f;
    const float lbA = onA ? g_lbA : 0.0f, lbB = onB ? g_lbB : 0.0f;
    float curA = onA ? g_cA : 0.0f, curB = onB ? g_cB : 0.0f;
    const float nmu = normd ? g_mau : 0.0f, niu = normd ? 1.0f / (g_sau + 1e-7f) : 1.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int l = 0; l < 4; ++l) asm volatile("" : "+v"(a_raw[i][l]), "+v"(a_tq[i][l]));   // the interpolation stays down here
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int pi = tid + i * NT;
        if (pi < a_pairs) {
            const int n = a_n[i], pp = pi & (QP - 1);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int j = a_j0[i] + l;
                int u = (a_tu[i] & 255) + l, t = a_tu[i] >> 8;
#pragma unroll
                for (int k = 0; k < 3; ++k) { t = (u >= U) ? t + 1 : t; u = (u >= U) ? u - U : u; }
                if (j < HU) {
                    float x = 0.0f, d2 = 0.0f;
                    if (n >= 0) {
                        if (m_ref || m_buf) x = a_raw[i][l];
                        else {
                            // word_to_trunc_normal's last line / the uniform draw / the injected draw
                            const float xi = has_rng ? (m_uni ? a_f[i][l] : fmaf(a_f[i][l], a_tq[i][l], a_raw[i][l])) : a_raw[i][l];
                            if (m_uni) x = xi * (a_hi[i][l] - a_lo[i][l]) + a_lo[i][l];
                            else x = xi * a_sg[i][l] + a_mn[i][l];
                        }
                        if (q.pen) {
                            const float xf = clipf(x, a_lo[i][l], a_hi[i][l]);
                            const float d = x - xf;
                            d2 = d * d;
                            x = xf;
                        }
                        if (p.samples) p.samples[((size_t)a * HU + j) * p.Nst + n] = x;
                    }
                    const int e = (t * QP + pp) * U + u;
                    acts[e] = x;
                    d2s[e] = d2;
                }
            }
        }
    }
    for (int i = tid; i < (HP - HG) * 16; i += NT) h0[HG * 16 + i] = 0.0f;
    __syncthreads();
    Q4S_MARK(6);

    // ---- prologue, part 2 (LDS only): normalised action groups, 0 * sum(a^2)
    for (int e = tid; e < H * AG * 16; e += NT) {         // (NT is a multiple of 16 * AG: e's action element is u_t in every pass)
        const int pp = (e >> 2) & 3, t = e / (16 * AG);
        xa[e] = u_on ? (acts[(t * QP + pp) * U + u_t] - nmu) * niu : 0.0f;
    }
    for (int e = tid; e < H * QP; e += NT) {
        const float* ac = acts + e * U;                    // e = t*QP + pp
        float ss = 0.0f;
        for (int u = 0; u < U; ++u) ss = ss + ac[u] * ac[u];
        zs[e] = 0.0f * ss;                                 // cost_func.py:21 (NaN / inf actions propagate)
    }
    // start state: the normalised input groups "own quad" and 4, by the loop's own all-gather
    f32x4 xA = rows_all_gather((curA - nmA) * niA), xB = rows_all_gather((curB - nmB) * niB);
    __syncthreads();
    Q4S_MARK(8);

    // where my layer-0 D fragment goes; blocks without a hidden quad (3 of 16, 5 on wave 3) store theirs into a pad group nobody
    // reads (60: the jobs read groups < 52 -- behind zero operand rows from 50 on --, the tail chains too): no exec-mask branch
    // around the activations (a branch inside the step costs more than the instructions it skips: NOTES_r6)
    float* const h0w = h0 + ((size_t)(l0_on ? q0 : 60) * 4 + pl) * 4;
    const bool rew_on = p.reward_kind == REW_CHEETAH;
    const float flag_thr = (pr == 1) ? 0.2f : 0.0f;       // cur[5] >= 0.2, cur[6] >= 0, cur[7] >= 0   cost_func.py:9-17
    // reward terms are parked per step and summed in step order after the loop (one IEEE division per (step, particle)
    // there instead of a ~12-instruction expansion inside every step): wave 0, quad 1 rows 1..3 own the flags of
    // cur[6], cur[5], cur[7]; wave 0, quad 0 of row 2 (feature 17) owns the progress difference
    const bool rwd_flag_lane = rew_on && wave == 0 && fgq == 1 && pr != 0;
    const bool rwd_prog_lane = rew_on && wave == 0 && fgq == 0 && pr == 1;
    float* rwd_f = rwd + pl * 4 + pr;                     // + t*16
    float* rwd_d = rwd + pl * 4;
    const float* hA0 = h0 + ((size_t)startA * 4 + pl) * 4;                   // B operands of the jobs: group startA + c, my particle
    const float* hT0 = h0 + ((size_t)row * 4 + pl) * 4 + fgq;                // of the tail chain: group row + 4m, element g
    float* const zxA = zx + lane;                                            // + 80 * wave: my partial of output feature 4g + pr
    float* const zxB = zx + 64 + row * 4 + pl;                               // + 80 * wave: of output feature 16 + pr
    // layer-0 accumulators (three chains), started with the bias and the action part of step 0
    f32x4 acc0, acc1, acc2;
    // the action part: 8 MFMAs in the chains 0 1 2 0 1 2 0 1
#define Q4S_ACTION_PART(ba0_, ba1_) do {                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        mfma4_operands_settled();                                                                                     \
        mfma4_v_c(acc0, wA0a[0], ba0_.x, b0); mfma4_v_0(acc1, wA0a[1], ba0_.y); mfma4_v_0(acc2, wA0a[2], ba0_.z);     \
        mfma4_v(acc0, wA0a[3], ba0_.w);       mfma4_v(acc1, wA0a[4], ba1_.x);   mfma4_v(acc2, wA0a[5], ba1_.y);       \
        mfma4_v(acc0, wA0a[6], ba1_.z);       mfma4_v(acc1, wA0a[7], ba1_.w);                                         \
    } while (0)
    {
        const f32x4 ba0 = *reinterpret_cast<const f32x4*>(xa + ((size_t)0 * 4 + pl) * 4);
        const f32x4 ba1 = *reinterpret_cast<const f32x4*>(xa + ((size_t)1 * 4 + pl) * 4);
        Q4S_ACTION_PART(ba0, ba1);
    }
#ifdef BBMPC_KERNEL_DBG
    Q4S_MARK(7);
    const long long dbg_cyc0 = (long long)clock64(), dbg_wall0 = (long long)wall_clock64();
#endif
    for (int t = 0; t < H; ++t) {
        // ---- layer 0, state part: groups 0..3 rotate through the row, group 4 is replicated
        {
            f32x4 r1, r2, r3;
            r1.x = dpp_mov<DPP_ROW_ROR4>(xA.x); r1.y = dpp_mov<DPP_ROW_ROR4>(xA.y);
            r1.z = dpp_mov<DPP_ROW_ROR4>(xA.z); r1.w = dpp_mov<DPP_ROW_ROR4>(xA.w);
            r2.x = dpp_mov<DPP_ROW_ROR8>(xA.x); r2.y = dpp_mov<DPP_ROW_ROR8>(xA.y);
            r2.z = dpp_mov<DPP_ROW_ROR8>(xA.z); r2.w = dpp_mov<DPP_ROW_ROR8>(xA.w);
            r3.x = dpp_mov<DPP_ROW_ROR12>(xA.x); r3.y = dpp_mov<DPP_ROW_ROR12>(xA.y);
            r3.z = dpp_mov<DPP_ROW_ROR12>(xA.z); r3.w = dpp_mov<DPP_ROW_ROR12>(xA.w);
            __builtin_amdgcn_sched_barrier(0);
            mfma4_operands_settled();
            // (the action part left the chains at acc1)
            mfma4_v(acc2, wA0r[0], xA.x);  mfma4_v(acc0, wA0r[1], xA.y);  mfma4_v(acc1, wA0r[2], xA.z);  mfma4_v(acc2, wA0r[3], xA.w);
            mfma4_v(acc0, wA0r[4], r1.x);  mfma4_v(acc1, wA0r[5], r1.y);  mfma4_v(acc2, wA0r[6], r1.z);  mfma4_v(acc0, wA0r[7], r1.w);
            mfma4_v(acc1, wA0r[8], r2.x);  mfma4_v(acc2, wA0r[9], r2.y);  mfma4_v(acc0, wA0r[10], r2.z); mfma4_v(acc1, wA0r[11], r2.w);
            mfma4_v(acc2, wA0r[12], r3.x); mfma4_v(acc0, wA0r[13], r3.y); mfma4_v(acc1, wA0r[14], r3.z); mfma4_v(acc2, wA0r[15], r3.w);
            mfma4_v(acc0, wA0b[0], xB.x);  mfma4_v(acc1, wA0b[1], xB.y);  mfma4_v(acc2, wA0b[2], xB.z);  mfma4_v(acc0, wA0b[3], xB.w);
            mfma4_results_ready(acc0, acc1, acc2);
            __builtin_amdgcn_sched_barrier(0);
            Q4S_MARK(12);
            f32x4 o;
            o.x = apply_act_q4s<A0>((acc0.x + acc1.x) + acc2.x, rt0); o.y = apply_act_q4s<A0>((acc0.y + acc1.y) + acc2.y, rt0);
            o.z = apply_act_q4s<A0>((acc0.z + acc1.z) + acc2.z, rt0); o.w = apply_act_q4s<A0>((acc0.w + acc1.w) + acc2.w, rt0);
            *reinterpret_cast<f32x4*>(h0w) = o;
        }
        Q4S_MARK(0);
        __syncthreads();                                   // h0 of step t is complete
        Q4S_MARK(1);
        // ---- layer 1: three 16-feature jobs per wave, K split over the rows, one shared set of B operands; the tail chain
        // rides along, one MFMA per round; stationary A operands in AccVGPRs, read by the MFMA directly
        float h1v[NJ], h1t = 0.0f;
        f32x4 ba0, ba1;
        {
            f32x4 bq[KA];
            // LDS returns in order and a round waits for what it needs only: group c of the jobs (1 KB per instruction: 104 LDS
            // cycles per wave for the 13, four waves at once) next to word c of the tail chain; the next step's action groups
            // (static data, for the action part behind the last layer) last
            if constexpr (NJ == 3) {
                float bt[KT];
                f32x4 cj0, cj1, cj2, ct;
                bq[0] = *reinterpret_cast<const f32x4*>(hA0);
                bt[0] = hT0[0]; bt[1] = hT0[64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 1; c < KA; ++c) {
                    bq[c] = *reinterpret_cast<const f32x4*>(hA0 + c * 16);
                    if (c + 1 < KT) bt[c + 1] = hT0[(c + 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    const int tn = (t + 1 < H) ? t + 1 : t;
                    ba0 = *reinterpret_cast<const f32x4*>(xa + ((size_t)(tn * AG + 0) * 4 + pl) * 4);
                    ba1 = *reinterpret_cast<const f32x4*>(xa + ((size_t)(tn * AG + 1) * 4 + pl) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                Q4S_MARK(13);
                mfma4_a_round3t_first(cj0, cj1, cj2, ct, wJ[0], wJ[1], wJ[2], bq[0], wT[0], bt[0]);
#pragma unroll
                for (int c = 1; c < KA; ++c) mfma4_a_round3t(cj0, cj1, cj2, ct, wJ[0] + 4 * c, wJ[1] + 4 * c, wJ[2] + 4 * c, bq[c], wT[c], bt[c]);
                mfma4_results_ready(cj0, cj1, cj2, ct);
                __builtin_amdgcn_sched_barrier(0);
                Q4S_MARK(2);
                h1v[0] = apply_act_q4s<A1>(rows_reduce_scatter(cj0) + b1s[0], rt1);
                h1v[1] = apply_act_q4s<A1>(rows_reduce_scatter(cj1) + b1s[1], rt1);
                h1v[2] = apply_act_q4s<A1>(rows_reduce_scatter(cj2) + b1s[2], rt1);
                // the tail chain: all 16 blocks hold partial sums -- the rows first (a reduce-scatter: four registers become one),
                // then the row's four blocks on that one register (8, then 4: every block adds the same pairs)
                float st = rows_reduce_scatter(ct);
                st = st + dpp_mov<DPP_ROW_ROR8>(st);
                st = st + dpp_mov<DPP_ROW_ROR4>(st);
                h1t = apply_act_q4s<A1>(st + b1t, rt1);
            } else {
                f32x4 cj0, cj1, cj2, cj3;
#pragma unroll
                for (int c = 0; c < KA; ++c) {
                    bq[c] = *reinterpret_cast<const f32x4*>(hA0 + c * 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    const int tn = (t + 1 < H) ? t + 1 : t;
                    ba0 = *reinterpret_cast<const f32x4*>(xa + ((size_t)(tn * AG + 0) * 4 + pl) * 4);
                    ba1 = *reinterpret_cast<const f32x4*>(xa + ((size_t)(tn * AG + 1) * 4 + pl) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                Q4S_MARK(13);
                mfma4_a_round4_first(cj0, cj1, cj2, cj3, wJ[0], wJ[1], wJ[2], wJ[3], bq[0]);
#pragma unroll
                for (int c = 1; c < KA; ++c) mfma4_a_round4(cj0, cj1, cj2, cj3, wJ[0] + 4 * c, wJ[1] + 4 * c, wJ[2] + 4 * c, wJ[3] + 4 * c, bq[c]);
                mfma4_results_ready(cj0, cj1, cj2, cj3);
                __builtin_amdgcn_sched_barrier(0);
                Q4S_MARK(2);
                h1v[0] = apply_act_q4s<A1>(rows_reduce_scatter(cj0) + b1s[0], rt1);
                h1v[1] = apply_act_q4s<A1>(rows_reduce_scatter(cj1) + b1s[1], rt1);
                h1v[2] = apply_act_q4s<A1>(rows_reduce_scatter(cj2) + b1s[2], rt1);
                h1v[3] = apply_act_q4s<A1>(rows_reduce_scatter(cj3) + b1s[3], rt1);
            }
        }
        Q4S_MARK(3);
        // ---- the last layer, my hidden features only, straight from the registers: set A (output quad g on block g; round j
        // takes the value of block g - j), set B (output features 16..19, every block its own k), the tail value last
        {
            f32x4 cA0, cA1, cA2, cA3, cB;
            float s1[NJ], s2[NJ], s3[NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                s1[jj] = dpp_mov<DPP_ROW_ROR4>(h1v[jj]); s2[jj] = dpp_mov<DPP_ROW_ROR8>(h1v[jj]); s3[jj] = dpp_mov<DPP_ROW_ROR12>(h1v[jj]);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma4_operands_settled();
            if constexpr (NJ == 3) {
                mfma4_a_0(cA0, wLA[0][0], h1v[0]); mfma4_a_0(cA1, wLA[1][0], h1v[1]); mfma4_a_0(cA2, wLA[2][0], h1v[2]); mfma4_a_0(cB, wLB[0], h1v[0]);
                mfma4_a(cA0, wLA[0][1], s1[0]);    mfma4_a(cA1, wLA[1][1], s1[1]);    mfma4_a(cA2, wLA[2][1], s1[2]);    mfma4_a(cB, wLB[1], h1v[1]);
                mfma4_a(cA0, wLA[0][2], s2[0]);    mfma4_a(cA1, wLA[1][2], s2[1]);    mfma4_a(cA2, wLA[2][2], s2[2]);    mfma4_a(cB, wLB[2], h1v[2]);
                mfma4_a(cA0, wLA[0][3], s3[0]);    mfma4_a(cA1, wLA[1][3], s3[1]);    mfma4_a(cA2, wLA[2][3], s3[2]);    mfma4_a(cB, wLTB, h1t);
                mfma4_a(cA0, wLTA, h1t);
            } else {
                mfma4_a_0(cA0, wLA[0][0], h1v[0]); mfma4_a_0(cA1, wLA[1][0], h1v[1]); mfma4_a_0(cA2, wLA[2][0], h1v[2]); mfma4_a_0(cA3, wLA[3][0], h1v[3]);
                mfma4_a_0(cB, wLB[0], h1v[0]);
                mfma4_a(cA0, wLA[0][1], s1[0]);    mfma4_a(cA1, wLA[1][1], s1[1]);    mfma4_a(cA2, wLA[2][1], s1[2]);    mfma4_a(cA3, wLA[3][1], s1[3]);
                mfma4_a(cB, wLB[1], h1v[1]);
                mfma4_a(cA0, wLA[0][2], s2[0]);    mfma4_a(cA1, wLA[1][2], s2[1]);    mfma4_a(cA2, wLA[2][2], s2[2]);    mfma4_a(cA3, wLA[3][2], s2[3]);
                mfma4_a(cB, wLB[2], h1v[2]);
                mfma4_a(cA0, wLA[0][3], s3[0]);    mfma4_a(cA1, wLA[1][3], s3[1]);    mfma4_a(cA2, wLA[2][3], s3[2]);    mfma4_a(cA3, wLA[3][3], s3[3]);
                mfma4_a(cB, wLB[3], h1v[3]);
            }
            Q4S_MARK(14);
            // the action part of the NEXT step's layer 0 (independent of everything here) covers the results' latency
            Q4S_ACTION_PART(ba0, ba1);
            if constexpr (NJ == 3) mfma4_results_ready(cA0, cA1, cA2, cB);
            else mfma4_results_ready(cA0, cA1, cA2, cA3, cB);
            __builtin_amdgcn_sched_barrier(0);
            Q4S_MARK(15);
            f32x4 sA = {(cA0.x + cA1.x) + cA2.x, (cA0.y + cA1.y) + cA2.y, (cA0.z + cA1.z) + cA2.z, (cA0.w + cA1.w) + cA2.w};
            if constexpr (NJ == 4) { sA.x = sA.x + cA3.x; sA.y = sA.y + cA3.y; sA.z = sA.z + cA3.z; sA.w = sA.w + cA3.w; }
            zxA[80 * wave] = rows_reduce_scatter(sA);      // output feature 4g + pr of my particle, my hidden features' share
            float pB = rows_reduce_scatter(cB);            // output feature 16 + pr: the rows first, then the row's four blocks
            pB = pB + dpp_mov<DPP_ROW_ROR8>(pB);
            pB = pB + dpp_mov<DPP_ROW_ROR4>(pB);
            if (fgq == 0) zxB[80 * wave] = pB;
        }
        Q4S_MARK(4);
        __syncthreads();                                   // every wave's partial sums are in zx
        Q4S_MARK(9);
        {
            // ---- every wave: the sum over the waves (one order everywhere: all four hold the same bits), then the epilogue on
            // the two features this lane finishes (process_output, then process_input of the next step)
            const float a0 = zxA[0], a1 = zxA[80], a2 = zxA[160], a3 = zxA[240];
            const float c0 = zxB[0], c1 = zxB[80], c2 = zxB[160], c3 = zxB[240];
            float zA = ((a0 + a1) + (a2 + a3)) + lbA;
            float zB = ((c0 + c1) + (c2 + c3)) + lbB;
            Q4S_MARK(16);
            zA = apply_act_q4s<A2>(zA, rt2); zB = apply_act_q4s<A2>(zB, rt2);
            const float vA = (tmA + zA * tsA) + curA, vB = (tmB + zB * tsB) + curB;
            if (rwd_flag_lane) rwd_f[t * 16] = (curA >= flag_thr) ? -10.0f : 0.0f;
            if (rwd_prog_lane) rwd_d[t * 16] = vB - curB;
            curA = vA; curB = vB;
            if (q.traj) {                                  // state after step t, for a user reward function
                const f32x4 rA = rows_all_gather(vA), rB = rows_all_gather(vB);
                if (n0 + pl < p.n_pop) {
                    float* dst = q.traj + ((((size_t)t * p.A + a) * p.Nst) + n0 + pl) * S;
                    if (S == 4 * SG) {
                        if (wave == 0 && row == 0) *reinterpret_cast<f32x4*>(dst + 4 * fgq) = rA;
                        if (wave == 0 && row == 1 && fgq == 0) *reinterpret_cast<f32x4*>(dst + 16) = rB;
                    } else if (wave == 0 && row == 0) {    // rows of dim_S floats are not 16-byte aligned
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (4 * fgq + c < S) dst[4 * fgq + c] = rA[c];
                            if (fgq == 0 && 16 + c < S) dst[16 + c] = rB[c];
                        }
                    }
                }
            }
            xA = rows_all_gather((vA - nmA) * niA);
            xB = rows_all_gather((vB - nmB) * niB);
        }
        Q4S_MARK(10);
    }
#undef Q4S_ACTION_PART
#ifdef BBMPC_KERNEL_DBG
    if (blockIdx.x == 0 && blockIdx.y == 0 && (tid == 0 || tid == 128)) {
        printf("[q4sdbg wave %d] H=%d | small loads issued %lld  operands 1 issued %lld  draws+table loads issued %lld  operands 2 issued %lld  constants+actions %lld  LDS part 2 %lld  first mfma %lld | loop %lld shader cycles in %lld x 10 ns, per step: rot+layer0 mfma %lld  sum+tanh+store %lld  bar1 %lld  lds reads issued %lld  layer1 mfma %lld  reduce+tanh %lld  rot+last mfma %lld  action part %lld  reduce+partials out %lld  bar2 %lld  partials in+sum %lld  epilogue+gather %lld\n",
               wave, H, dbg_acc[11], dbg_acc[17], dbg_acc[18], dbg_acc[5], dbg_acc[6], dbg_acc[8], dbg_acc[7], (long long)clock64() - dbg_cyc0, (long long)wall_clock64() - dbg_wall0, dbg_acc[12] / H, dbg_acc[0] / H, dbg_acc[1] / H, dbg_acc[13] / H,
               dbg_acc[2] / H, dbg_acc[3] / H, dbg_acc[14] / H, dbg_acc[15] / H, dbg_acc[4] / H, dbg_acc[9] / H, dbg_acc[16] / H, dbg_acc[10] / H);
    }
#endif
    // ---- rewards: cost_func.py:5-22 per step, summed in step order (deterministic.py:62-73)
    // (the H x 4 step rewards by all threads -- each has a division -- then four threads add them up in step order: one
    // thread per particle walking all of it was 1.5 us at the end of every launch, at a lone wave's issue rate)
    __syncthreads();
    float* rstep = h0;                                     // [H][QP]: the activations are dead
    if (rew_on)
        for (int e = tid; e < H * QP; e += NT) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(rwd + e * 4);   // (d17, flag 5, flag 6, flag 7) of (step, particle) e
            float r = 0.0f;
            r = r + w.y; r = r + w.z; r = r + w.w;
            r = r + w.x / 0.01f;
            r = r - zs[e];
            rstep[e] = r;
        }
    if (tid >= 64 && tid < 64 + QP * U) {                  // (wave 1: next to the step rewards, not in front of the loop)
        const int pp = (tid - 64) / U, u = (tid - 64) % U;
        float pen_part = 0.0f;
        if (q.pen && n0 + pp < p.n_pop)
            for (int t = 0; t < H; ++t) pen_part = pen_part + d2s[(t * QP + pp) * U + u];
        pens[tid - 64] = pen_part;
    }
    __syncthreads();
    if (tid < QP) {
        float total = 0.0f;
        if (rew_on)
            for (int t = 0; t < H; ++t) total = total + rstep[t * QP + tid];
        const int n = n0 + tid;
        if (n < p.n_pop) {
            if (total != total) total = -1.0e6f;
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
}

inline int mlp_q4s_lds_floats(int HG, int K0G, int H, int U) {
    (void)HG;
    return 64 * 16 + 4 * 80 + ((H * 4 * U + 3) & ~3) + ((4 * U + 3) & ~3) + H * (K0G - 5) * 16 + H * 4 + H * 16 + ((H * 4 * U + 3) & ~3) + 8;
}

}  // namespace bbmpc
