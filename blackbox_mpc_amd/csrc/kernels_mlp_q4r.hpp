// Quad mode, register-resident state ("q4r"): the 4-particle workgroup of k_rollout_mlp_q4 with TWO barriers per
// model step instead of four and a quarter fewer matrix instructions.  Same fused path (reference files under
// blackbox_mpc/):
//   SystemDynamicsHandler.process_input / process_output   dynamics_handlers/system_dynamics_handler.py:97-161
//   DeterministicMLP.__call__                              dynamics_functions/deterministic_mlp.py:27-51
//   reward_function                                        tutorials/mujoco/cost_func.py:5-22
//   DeterministicTrajectoryEvaluator.__call__              trajectory_evaluators/deterministic.py:26-77
//
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4 outer products, and nothing says the 16 blocks have to work on the
// same k.  k_rollout_mlp_q4 gives every block the same k and 64 different output features; a wave then issues 200
// MFMAs for layer 1 whatever its share of the 200 hidden units is (the fourth wave: 8 of 64 lanes useful), and the
// 200 -> 20 layer has to be K-split ACROSS the waves, which costs an exchange of partial sums plus a wave-0-only
// epilogue plus an exchange of the next input: four barrier / LDS round trips per model step, 0.86 us of 2.33.
// Here the K split is INSIDE the instruction:
//  * layer 1 runs as "jobs" of 16 output features: block (row r, quad g) = features 16j+4g..+3 over the k-slice of
//    16-lane row r (13 k-groups), 52 MFMAs per job, three jobs per wave sharing ONE set of B operands (13 LDS reads
//    per lane instead of 50) + an 8-feature tail job (2 feature blocks x 8 slices, 28 MFMAs) on wave 3:
//    156 / 184 MFMAs per wave instead of 200;
//  * the last layer is computed whole by every wave: set A (features 0..15: 4 quads x 4 row slices, 52 MFMAs) and
//    set B (features 16..19: block b takes k-groups b, b+16, b+32, b+48, 16 MFMAs);
//  * the slices are reduced in registers: inside a row with DPP rotations (8 then 4: every block adds the same pairs,
//    so all blocks hold the same bits), across the four rows with v_permlane32_swap / v_permlane16_swap used as a
//    reduce-scatter (one swap + one add folds TWO registers: 4 registers -> 1 in 6 instructions; row r ends up with
//    element {0,2,1,3}[r]).  Activation / bias / de-normalisation / residual / normalisation then touch 1-2 values
//    per lane, and for the state the same swaps run backwards as an all-gather;
//  * a lane then holds the normalised input groups "own quad" and 4 of its particle: layer 0 takes the other three
//    state groups from its row neighbours by DPP rotation -- block g multiplies group (g - j) mod 4 in round j, its
//    stationary A operands are loaded in that order once -- so the next step's layer 0 starts from registers.
// Barriers remain after h0 and after h1.  The action part of layer 0 (independent of the state) is issued one step
// ahead, behind the last layer's LDS reads, whose latency it fills.
// Requires dim_S == 20 (5 state groups: four rotate inside a row, the fifth is replicated), the HalfCheetah reward or
// none (a user function scores the recorded trajectory); everything else keeps k_rollout_mlp_q4.
#pragma once
#include "kernels_mlp.hpp"

namespace bbmpc {

constexpr int Q4R_MAX_ACTION_PAIRS = 512;               // (particle, 4 consecutive action-sequence elements) pairs a workgroup's threads keep in registers across the operand loads

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
__device__ __forceinline__ void mfma4_a_c(f32x4& acc, float a_in_agpr, float b, const f32x4& c) {
    asm volatile(BBMPC_MFMA4 "%0, %1, %2, %3" : "=&v"(acc) : "a"(a_in_agpr), "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma4_a_0(f32x4& acc, float a_in_agpr, float b) {
    asm volatile(BBMPC_MFMA4 "%0, %1, %2, 0" : "=&v"(acc) : "a"(a_in_agpr), "v"(b));
}
// One k-group of three layer-1 jobs = 12 MFMAs in ONE asm statement (three chains, round robin): between separate
// statements the compiler pads every re-use of an accumulator with an s_nop it cannot know to be unnecessary.
__device__ __forceinline__ void mfma4_a_round3(f32x4& c0, f32x4& c1, f32x4& c2, const float* w0, const float* w1, const float* w2, const f32x4& b) {
    asm volatile(BBMPC_MFMA4 "%0, %3, %15, %0\n\t" BBMPC_MFMA4 "%1, %7, %15, %1\n\t" BBMPC_MFMA4 "%2, %11, %15, %2\n\t"
                 BBMPC_MFMA4 "%0, %4, %16, %0\n\t" BBMPC_MFMA4 "%1, %8, %16, %1\n\t" BBMPC_MFMA4 "%2, %12, %16, %2\n\t"
                 BBMPC_MFMA4 "%0, %5, %17, %0\n\t" BBMPC_MFMA4 "%1, %9, %17, %1\n\t" BBMPC_MFMA4 "%2, %13, %17, %2\n\t"
                 BBMPC_MFMA4 "%0, %6, %18, %0\n\t" BBMPC_MFMA4 "%1, %10, %18, %1\n\t" BBMPC_MFMA4 "%2, %14, %18, %2"
                 : "+v"(c0), "+v"(c1), "+v"(c2)
                 : "a"(w0[0]), "a"(w0[1]), "a"(w0[2]), "a"(w0[3]), "a"(w1[0]), "a"(w1[1]), "a"(w1[2]), "a"(w1[3]),
                   "a"(w2[0]), "a"(w2[1]), "a"(w2[2]), "a"(w2[3]), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
}
// the first k-group: the chains start from the inline constant 0
__device__ __forceinline__ void mfma4_a_round3_first(f32x4& c0, f32x4& c1, f32x4& c2, const float* w0, const float* w1, const float* w2, const f32x4& b) {
    asm volatile(BBMPC_MFMA4 "%0, %3, %15, 0\n\t"  BBMPC_MFMA4 "%1, %7, %15, 0\n\t"  BBMPC_MFMA4 "%2, %11, %15, 0\n\t"
                 BBMPC_MFMA4 "%0, %4, %16, %0\n\t" BBMPC_MFMA4 "%1, %8, %16, %1\n\t" BBMPC_MFMA4 "%2, %12, %16, %2\n\t"
                 BBMPC_MFMA4 "%0, %5, %17, %0\n\t" BBMPC_MFMA4 "%1, %9, %17, %1\n\t" BBMPC_MFMA4 "%2, %13, %17, %2\n\t"
                 BBMPC_MFMA4 "%0, %6, %18, %0\n\t" BBMPC_MFMA4 "%1, %10, %18, %1\n\t" BBMPC_MFMA4 "%2, %14, %18, %2"
                 : "=&v"(c0), "=&v"(c1), "=&v"(c2)
                 : "a"(w0[0]), "a"(w0[1]), "a"(w0[2]), "a"(w0[3]), "a"(w1[0]), "a"(w1[1]), "a"(w1[2]), "a"(w1[3]),
                   "a"(w2[0]), "a"(w2[1]), "a"(w2[2]), "a"(w2[3]), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
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

template <int HG, int K0G, int A0, int A1, int A2, int NE>
__global__ __launch_bounds__(256, 1) void k_rollout_mlp_q4r(MlpRolloutArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RolloutArgs& p = q.r;
    const MlpDesc& m = q.m;
    constexpr int NT = 256, QP = 4;
    constexpr int SG = 5, AG = K0G - SG;                 // state / action input groups (dim_S == 20)
    constexpr int KA = (HG + 3) / 4;                     // k groups per 16-lane row when K is split over the rows
    constexpr int KB = (HG + 15) / 16;                   // last layer set B: k groups per 4-lane block
    constexpr int HP = 64;                               // h0 / h1 are padded to 64 groups (zero): clamp-free operand addresses
    constexpr int ZROW = HP - 1;                         // a zero group of the packed operands (bbmpc_set_mlp pads the k/4 axis to 64)
    constexpr int NJ = HG / 16;                          // layer-1 jobs of 16 output features per wave (12 jobs = 192 features)
    constexpr int TF = HG * 4 - NJ * 64;                 // features left for the tail job (8)
    constexpr int KT = (HG + 7) / 8;                     // tail job: k groups per slice (8 slices)
    static_assert(HG < HP && KA * 4 <= HP && (KB - 1) * 16 + 15 < HP && K0G > SG, "padding / input groups");
    static_assert(NJ == 3 && TF == 8 && KT <= KA && KB <= KA && AG == 2, "written for 200 hidden units, 20 + <= 8 inputs");
    const int a = blockIdx.y, n0 = xcd_tile(blockIdx.x, gridDim.x, blockIdx.y) * QP;     // an XCD's workgroups: contiguous particles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches
    const int S = p.S, U = p.U, H = p.H;
    const bool normd = m.normalized != 0;
    const int row = lane >> 4, blk = lane >> 2, fgq = blk & 3, pl = lane & 3;
    const int pr = ((row & 1) << 1) | (row >> 1);        // element of a reduce-scattered float4 this row holds
    // ---- LDS: h0[HP][4][4] | h1[HP][4][4] | acts[H][4][U] | pen[4U] | xa[H][AG][4][4] | zs[H][4] | rwd[H][4][4] | constants
    float* h0 = smem;
    float* h1 = h0 + HP * 16;
    float* acts = h1 + HP * 16;
    float* pens = acts + ((H * QP * U + 3) & ~3);
    float* xa = pens + ((QP * U + 3) & ~3);
    float* zs = xa + H * AG * 16;
    float* rwd = zs + H * QP;                             // per step and particle: (progress difference, flag 5, flag 6, flag 7)
    float* nmean = rwd + H * QP * 4;                      // [S+U] input mean, [S+U] 1/(std+1e-7), [S] target mean, [S] std+1e-7,
    float* ninv = nmean + 32;                             // [S] last bias, [S] start state   (S = 20, S+U <= 28)
    float* tmean = ninv + 32;
    float* tstd = tmean + 32;
    float* lbias = tstd + 32;
    float* st0 = lbias + 32;
#ifdef BBMPC_KERNEL_DBG
    long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dbg_t0 = (long long)wall_clock64();
#define Q4R_MARK(i) do { const long long now_ = (long long)wall_clock64(); dbg_acc[i] += now_ - dbg_t0; dbg_t0 = now_; } while (0)
    long long dbg_cyc[4] = {0, 0, 0, 0}, dbg_c0 = 0;
#define Q4R_CYC0() do { dbg_c0 = (long long)clock64(); } while (0)
#define Q4R_CYC(i) do { dbg_cyc[i] += (long long)clock64() - dbg_c0; } while (0)
#else
#define Q4R_MARK(i) do {} while (0)
#define Q4R_CYC0() do {} while (0)
#define Q4R_CYC(i) do {} while (0)
#endif

    // ---- prologue.  Vector memory returns in order, so the SMALL loads (constants, start state, the sources of the 4
    // particles' action block) are issued first, then the ~260 KB of stationary operands, and only then is anything
    // consumed: the small results arrive first and are worked on while the operands stream in (the two waits used to
    // add up, 3.2 + 4.1 us of a 70 us kernel).
    // The action block is handled in (particle, 4 consecutive j) pairs, NE per thread (4 * ceil(H*U / 4) <= 256 * NE): the
    // four elements of a pair share one Philox block (rng.hpp: rng_block is keyed by j >> 2), so a thread draws once,
    // not four times -- 120 VALU instructions each at a lone wave's issue rate were a good microsecond of the prologue.
    const int HU = p.HU, a_pairs = QP * ((HU + 3) >> 2);
    const int ci = min(tid, S + U - 1), cs = min(tid, S - 1);
    const float* cmu_base = !normd ? p.state : (ci < S ? m.mean_s : m.mean_a);
    const float* csd_base = !normd ? p.state : (ci < S ? m.std_s : m.std_a);
    const int cmi = !normd ? 0 : (ci < S ? ci : ci - S);
    const float c_mu = cmu_base[cmi], c_sd = csd_base[cmi];
    const float c_tm = (normd ? m.mean_t : p.state)[normd ? cs : 0], c_ts = (normd ? m.std_t : p.state)[normd ? cs : 0];
    const float c_lb = q.braw[2][cs], c_st = p.state[a * S + cs];
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
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int pi = tid + i * NT;
        const int pp = pi & (QP - 1), j0 = (pi >> 2) << 2;
        const int n = n0 + pp;
        const bool pvalid = pi < a_pairs && n < p.n_pop;
        const int t0 = j0 / U, u0 = j0 - t0 * U;
        a_n[i] = pvalid ? n : -1; a_j0[i] = j0; a_tu[i] = (t0 << 8) | u0;
        // word_to_trunc_normal (rng.hpp) split in two: the table entry is loaded here, the interpolation happens with the
        // other small results -- as one piece it would sit behind the operand loads.  One unconditional pair of loads
        // per element whatever the mode (a load inside a branch costs a wait at the join).
        U4 blk4 = {0u, 0u, 0u, 0u};
        if (has_rng) blk4 = rng_block(key_now, p.stream, p.iter, (uint32_t)(n + p.pop_offset), (uint32_t)(p.agent_offset + a), (uint32_t)j0);
        const bool tn = has_rng && !m_uni;
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
    Q4R_MARK(11);
    // ---- roles.  Waves 0..2 are STATE waves: layer 0 for their 64 hidden features, three layer-1 jobs, the whole last
    // layer, the state.  Wave 3 is the HELPER: layer-1 jobs 3, 7, 11 with the others and, after the second barrier,
    // job 12 (hidden features 192..199) while the state waves are already in the last layer, which takes job 12's
    // output last, behind a third barrier the helper reaches first.  The helper holds no state, so the eight layer-0
    // features that used to be its share (192..199) are a K-split "mini job" inside every state wave's layer-0
    // stream (zero weights except on wave MJ, which reduces and stores them).
    const bool helper = wave == 3;
    constexpr int MJ = 2;
    const bool mini = wave == MJ;
    // ---- stationary A operands (packed [k/4][Mp][4] by bbmpc_set_mlp): ~250 KB per workgroup.  Operands a role does
    // not use are read from a zero row (one cached 1 KB line) instead of being branched around: a load under a branch
    // is followed by a wait for everything issued so far.
    const int M1 = m.dims[1], M3 = m.dims[3];
    const int Mp1 = (M1 + 63) & ~63, Mp3 = (M3 + 63) & ~63;
    const float4* __restrict__ Q0 = reinterpret_cast<const float4*>(q.wq4[0]);
    const float4* __restrict__ Q1 = reinterpret_cast<const float4*>(q.wq4[1]);
    const float4* __restrict__ Q2 = reinterpret_cast<const float4*>(q.wq4[2]);
    const int f = wave * 64 + lane;                      // hidden feature of this lane's layer-0 A operands
    float wA0r[16], wA0b[4], wA0a[AG * 4], wM[4], wMa[4], wJ[NJ][KA * 4], wU[KA * 4], wA2b[KB * 4];
#pragma unroll
    for (int g = 0; g < AG; ++g) {
        const float4 v = Q0[(size_t)(helper ? ZROW : SG + g) * Mp1 + f];
        wA0a[4 * g + 0] = v.x; wA0a[4 * g + 1] = v.y; wA0a[4 * g + 2] = v.z; wA0a[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                        // round j of layer 0: my block multiplies state group (fgq - j) & 3
        const float4 v = Q0[(size_t)(helper ? ZROW : ((fgq - j) & 3)) * Mp1 + f];
        wA0r[4 * j + 0] = v.x; wA0r[4 * j + 1] = v.y; wA0r[4 * j + 2] = v.z; wA0r[4 * j + 3] = v.w;
    }
    {
        const float4 v = Q0[(size_t)(helper ? ZROW : 4) * Mp1 + f];
        wA0b[0] = v.x; wA0b[1] = v.y; wA0b[2] = v.z; wA0b[3] = v.w;
    }
    {   // mini job: block (row, quad g) works on hidden quad 48 + (row >> 1); even rows multiply state group g (the lane's own
        // x), odd rows state group 4 (g == 0) and, one step ahead, action groups 5 / 6 (g == 1 / 2)
        const int om = 64 * NJ + 4 * (row >> 1) + (lane & 3);
        const int gs = (row & 1) ? (fgq == 0 ? 4 : ZROW) : fgq;
        const int ga = ((row & 1) && (fgq == 1 || fgq == 2)) ? SG - 1 + fgq : ZROW;
        const float4 v = Q0[(size_t)(mini ? gs : ZROW) * Mp1 + om];
        const float4 u = Q0[(size_t)(mini ? ga : ZROW) * Mp1 + om];
        wM[0] = v.x; wM[1] = v.y; wM[2] = v.z; wM[3] = v.w;
        wMa[0] = u.x; wMa[1] = u.y; wMa[2] = u.z; wMa[3] = u.w;
    }
    // K split over the rows (layer-1 jobs, last-layer set A): row r takes k groups [KA*r, KA*r + cnt)
    const int startA = KA * row;
    const int cntA = min(KA, HG - startA);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {                    // job = wave + 4*jj: output features 16*job + (lane & 15)
        const int o = 16 * (wave + 4 * jj) + (lane & 15);
#pragma unroll
        for (int c = 0; c < KA; ++c) {
            const float4 v = Q1[(size_t)(c < cntA ? startA + c : ZROW) * Mp1 + o];
            wJ[jj][4 * c + 0] = v.x; wJ[jj][4 * c + 1] = v.y; wJ[jj][4 * c + 2] = v.z; wJ[jj][4 * c + 3] = v.w;
        }
    }
    // wU: the helper's job 12 (output features 192 + (lane & 15); the packed rows past the 200th are zero) or, on a state
    // wave, the last layer's set A (output feature = lane & 15, k slice of my row)
    {
        const float4* __restrict__ QU = helper ? Q1 : Q2;
        const size_t MpU = helper ? Mp1 : Mp3;
        const int oU = (helper ? 16 * 4 * NJ : 0) + (lane & 15);
#pragma unroll
        for (int c = 0; c < KA; ++c) {
            const float4 v = QU[(size_t)(c < cntA ? startA + c : ZROW) * MpU + oU];
            wU[4 * c + 0] = v.x; wU[4 * c + 1] = v.y; wU[4 * c + 2] = v.z; wU[4 * c + 3] = v.w;
        }
    }
    // set B: block b takes k groups b, b + 16, ...; output feature = 16 + (lane & 3)
#pragma unroll
    for (int c = 0; c < KB; ++c) {
        const int gg = blk + 16 * c;
        const float4 v = Q2[(size_t)((gg < HG && !helper) ? gg : ZROW) * Mp3 + 16 + (lane & 3)];
        wA2b[4 * c + 0] = v.x; wA2b[4 * c + 1] = v.y; wA2b[4 * c + 2] = v.z; wA2b[4 * c + 3] = v.w;
    }
    // biases: layer 0 enters as the C operand of a chain's first MFMA (my 4 D rows: features 64*wave + 4*blk + r); the
    // K-split products get theirs after the reduction, where a lane holds ONE feature -- one register per product instead
    // of a four-register C operand that is zero in 15 of 16 blocks (the kernel is short of registers, not of VALU slots)
    f32x4 b0;
    {
        const int fb = helper ? 0 : wave * 64 + 4 * blk;      // < 192 on a state wave
        b0.x = q.braw[0][fb + 0]; b0.y = q.braw[0][fb + 1]; b0.z = q.braw[0][fb + 2]; b0.w = q.braw[0][fb + 3];
    }
    const float bMs = q.braw[0][64 * NJ + 4 * (row >> 1) + 2 * (fgq >> 1) + (row & 1)];     // mini job: the feature this lane finishes
    float b1s[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) b1s[jj] = q.braw[1][16 * (wave + 4 * jj) + 4 * fgq + pr];
    const float b1xs = q.braw[1][min(16 * 4 * NJ + 4 * fgq + pr, M1 - 1)];                    // job 12: features 192 + 4*fgq + pr (< 200 for fgq < 2)
    __builtin_amdgcn_sched_barrier(0);
    Q4R_MARK(5);

    // ---- the small results: constants to LDS; the action block (mlp_fill_actions' arithmetic, every thread clipping its
    // own elements; xa is scratch for the squared clip distances, summed per (particle, u) in step order below)
    if (q.state_copy && blockIdx.x == 0 && tid < S) q.state_copy[a * S + tid] = c_st;
    if (tid < S + U) {
        nmean[tid] = normd ? c_mu : 0.0f;
        ninv[tid] = normd ? 1.0f / (c_sd + 1e-7f) : 1.0f;
        if (tid < S) {
            tmean[tid] = normd ? c_tm : 0.0f;              // un-normalised: 0 + z * 1 = z exactly
            tstd[tid] = normd ? (c_ts + 1e-7f) : 1.0f;
            lbias[tid] = c_lb;
            st0[tid] = c_st;
        }
    }
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
                    xa[e] = d2;
                }
            }
        }
    }
    for (int i = tid; i < (HP - HG) * 16; i += NT) { h0[HG * 16 + i] = 0.0f; h1[HG * 16 + i] = 0.0f; }
    __syncthreads();
    if (tid < QP * U) {
        const int pp = tid / U, u = tid % U;
        float pen_part = 0.0f;
        if (q.pen && n0 + pp < p.n_pop)
            for (int t = 0; t < H; ++t) pen_part = pen_part + xa[(t * QP + pp) * U + u];
        pens[tid] = pen_part;
    }
    __syncthreads();
    Q4R_MARK(6);

    // ---- prologue, part 2 (LDS only): normalised action groups, 0 * sum(a^2), constants
    for (int e = tid; e < H * AG * 16; e += NT) {
        const int c = e & 3, pp = (e >> 2) & 3, ga = (e >> 4) % AG, t = e / (16 * AG);
        const int u = ga * 4 + c;
        xa[e] = (u < U) ? (acts[(t * QP + pp) * U + u] - nmean[S + u]) * ninv[S + u] : 0.0f;
    }
    for (int e = tid; e < H * QP; e += NT) {
        const float* ac = acts + e * U;                    // e = t*QP + pp
        float ss = 0.0f;
        for (int u = 0; u < U; ++u) ss = ss + ac[u] * ac[u];
        zs[e] = 0.0f * ss;                                 // cost_func.py:21 (NaN / inf actions propagate)
    }
    // the two output features this lane finishes: fA = 4*fgq + pr, fB = 16 + pr
    const int fA = 4 * fgq + pr, fB = 16 + pr;
    const float tmA = tmean[fA], tsA = tstd[fA], tmB = tmean[fB], tsB = tstd[fB];
    const float nmA = nmean[fA], niA = ninv[fA], nmB = nmean[fB], niB = ninv[fB];
    const float lbA = lbias[fA], lbB = lbias[fB];
    // start state: the scattered copies (residual) and the normalised input groups "own quad" and 4
    float curA = st0[fA], curB = st0[fB];
    f32x4 xA, xB;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        xA[c] = (st0[4 * fgq + c] - nmean[4 * fgq + c]) * ninv[4 * fgq + c];
        xB[c] = (st0[16 + c] - nmean[16 + c]) * ninv[16 + c];
    }
    __syncthreads();
    Q4R_MARK(8);

    const int my_row = (wave * 16 + blk) * 16 + pl * 4;   // where my layer-0 D fragment goes in h0 (floats; state waves)
    // mini job: lane (row, g, pl) ends up with feature 192 + 4*(row >> 1) + 2*(g >> 1) + (row & 1) of particle pl
    float* h0m = h0 + ((size_t)(16 * NJ + (row >> 1)) * 4 + pl) * 4 + 2 * (fgq >> 1) + (row & 1);
    const bool rew_on = p.reward_kind == REW_CHEETAH;
    const float flag_thr = (pr == 1) ? 0.2f : 0.0f;       // cur[5] >= 0.2, cur[6] >= 0, cur[7] >= 0   cost_func.py:9-17
    // reward terms are parked per step and summed in step order after the loop (one IEEE division per (step, particle)
    // there instead of a ~12-instruction expansion inside every step): wave 0, quad 1 rows 1..3 own the flags of
    // cur[6], cur[5], cur[7]; wave 0, quad 0 of row 2 (feature 17) owns the progress difference
    const bool rwd_flag_lane = rew_on && wave == 0 && fgq == 1 && pr != 0;
    const bool rwd_prog_lane = rew_on && wave == 0 && fgq == 0 && pr == 1;
    float* rwd_f = rwd + pl * 4 + pr;                     // + t*16
    float* rwd_d = rwd + pl * 4;
    // where my layer-1 results go: job feature 16*job + 4*fgq + pr -> h1[group 4*job + fgq][pl][pr]
    float* h1w = h1 + ((size_t)(4 * wave + fgq) * 4 + pl) * 4 + pr;          // + jj * 16 groups
    float* h1x = h1 + ((size_t)(16 * NJ + fgq) * 4 + pl) * 4 + pr;           // job 12 (groups 50, 51 receive tanh(0) = 0)
    const float* hA0 = h0 + ((size_t)startA * 4 + pl) * 4;
    const float* hA1 = h1 + ((size_t)startA * 4 + pl) * 4;
    const float* hB1 = h1 + ((size_t)blk * 4 + pl) * 4;
    // rounds of the last layer that read job 12's output (h1 groups 48, 49): set A rounds LA0 .. LA0+1 (row 3), set B round KB-1
    constexpr int LA0 = 16 * NJ - KA * 3;                 // 48 - 39 = 9
    static_assert(LA0 + 2 <= KA - 2 && 16 * (KB - 1) == 16 * NJ, "late rounds of the last layer");
    // layer-0 accumulators (three chains + the mini job's), started with the bias and the action part of step 0
    f32x4 acc0, acc1, acc2, accM;
    // the action part: 8 MFMAs in the chains 0 1 2 0 1 2 0 1 with the mini job's four spread between them
#define Q4R_ACTION_PART(ba0_, ba1_) do {                                                                            \
        const f32x4 bm_ = (fgq == 1) ? ba0_ : ba1_;                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        mfma4_operands_settled();                                                                                     \
        mfma4_v_0(accM, wMa[0], bm_.x);                                                                               \
        mfma4_v_c(acc0, wA0a[0], ba0_.x, b0); mfma4_v_0(acc1, wA0a[1], ba0_.y); mfma4_v_0(acc2, wA0a[2], ba0_.z);     \
        mfma4_v(accM, wMa[1], bm_.y);                                                                                 \
        mfma4_v(acc0, wA0a[3], ba0_.w);       mfma4_v(acc1, wA0a[4], ba1_.x);   mfma4_v(acc2, wA0a[5], ba1_.y);       \
        mfma4_v(accM, wMa[2], bm_.z);                                                                                 \
        mfma4_v(acc0, wA0a[6], ba1_.z);       mfma4_v(acc1, wA0a[7], ba1_.w);                                         \
        mfma4_v(accM, wMa[3], bm_.w);                                                                                 \
    } while (0)
    if (!helper) {
        const f32x4 ba0 = *reinterpret_cast<const f32x4*>(xa + ((size_t)0 * 4 + pl) * 4);
        const f32x4 ba1 = *reinterpret_cast<const f32x4*>(xa + ((size_t)1 * 4 + pl) * 4);
        Q4R_ACTION_PART(ba0, ba1);
    }
#ifdef BBMPC_KERNEL_DBG
    Q4R_MARK(7);
    const long long dbg_start = dbg_t0, dbg_cyc0 = (long long)clock64();
#endif
    // Two loops, one per role, with the same three barriers per model step: as ONE loop with role branches every value a
    // role defines (the state, the layer-0 accumulators, ...) is live through the other role's path as far as the
    // register allocator can tell, which cost ~40 VGPR <-> AccVGPR copies per step.
    if (helper) {
        for (int t = 0; t < H; ++t) {
            __syncthreads();                               // h0 of step t is complete
            // ---- layer-1 jobs 3, 7, 11 (see the state loop)
            f32x4 bq[KA];
            {
                f32x4 cj0, cj1, cj2;
#pragma unroll
                for (int c = 0; c < KA; ++c) bq[c] = *reinterpret_cast<const f32x4*>(hA0 + c * 16);
                __builtin_amdgcn_sched_barrier(0);
                mfma4_a_round3_first(cj0, cj1, cj2, wJ[0], wJ[1], wJ[2], bq[0]);
#pragma unroll
                for (int c = 1; c < KA; ++c) mfma4_a_round3(cj0, cj1, cj2, wJ[0] + 4 * c, wJ[1] + 4 * c, wJ[2] + 4 * c, bq[c]);
                mfma4_results_ready(cj0, cj1, cj2);
                __builtin_amdgcn_sched_barrier(0);
                h1w[0 * 256] = apply_act_ct<A1>(rows_reduce_scatter(cj0) + b1s[0]);
                h1w[1 * 256] = apply_act_ct<A1>(rows_reduce_scatter(cj1) + b1s[1]);
                h1w[2 * 256] = apply_act_ct<A1>(rows_reduce_scatter(cj2) + b1s[2]);
            }
            __syncthreads();                               // every wave's jobs are in h1
            // ---- job 12 (hidden features 192..199) from the same B operands, three chains over its 52 MFMAs (MFMA number
            // i = 4c + e goes to chain i % 3), while the state waves run the part of the last layer that does not need it
            f32x4 cx0, cx1, cx2;
            __builtin_amdgcn_sched_barrier(0);
            mfma4_a_0(cx0, wU[0], bq[0].x); mfma4_a_0(cx1, wU[1], bq[0].y); mfma4_a_0(cx2, wU[2], bq[0].z);
            mfma4_a(cx0, wU[3], bq[0].w);
#pragma unroll
            for (int c = 1; c < KA; ++c) {
                f32x4& x0_ = ((4 * c + 0) % 3 == 0) ? cx0 : ((4 * c + 0) % 3 == 1) ? cx1 : cx2;
                f32x4& x1_ = ((4 * c + 1) % 3 == 0) ? cx0 : ((4 * c + 1) % 3 == 1) ? cx1 : cx2;
                f32x4& x2_ = ((4 * c + 2) % 3 == 0) ? cx0 : ((4 * c + 2) % 3 == 1) ? cx1 : cx2;
                f32x4& x3_ = ((4 * c + 3) % 3 == 0) ? cx0 : ((4 * c + 3) % 3 == 1) ? cx1 : cx2;
                mfma4_a(x0_, wU[c * 4 + 0], bq[c].x); mfma4_a(x1_, wU[c * 4 + 1], bq[c].y);
                mfma4_a(x2_, wU[c * 4 + 2], bq[c].z); mfma4_a(x3_, wU[c * 4 + 3], bq[c].w);
            }
            mfma4_results_ready(cx0, cx1, cx2);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 sx = {(cx0.x + cx1.x) + cx2.x, (cx0.y + cx1.y) + cx2.y, (cx0.z + cx1.z) + cx2.z, (cx0.w + cx1.w) + cx2.w};
            *h1x = apply_act_ct<A1>(rows_reduce_scatter(sx) + (fgq < 2 ? b1xs : 0.0f));
            __syncthreads();                               // job 12's output is in h1
        }
    } else {
    float zA = 0.0f, zB = 0.0f;
    for (int t = 0; t < H; ++t) {
        // ---- layer 0, state part: groups 0..3 rotate through the row, group 4 is replicated; the mini job rides along
        {
            f32x4 r1, r2, r3, bm;
            r1.x = dpp_mov<DPP_ROW_ROR4>(xA.x); r1.y = dpp_mov<DPP_ROW_ROR4>(xA.y);
            r1.z = dpp_mov<DPP_ROW_ROR4>(xA.z); r1.w = dpp_mov<DPP_ROW_ROR4>(xA.w);
            r2.x = dpp_mov<DPP_ROW_ROR8>(xA.x); r2.y = dpp_mov<DPP_ROW_ROR8>(xA.y);
            r2.z = dpp_mov<DPP_ROW_ROR8>(xA.z); r2.w = dpp_mov<DPP_ROW_ROR8>(xA.w);
            r3.x = dpp_mov<DPP_ROW_ROR12>(xA.x); r3.y = dpp_mov<DPP_ROW_ROR12>(xA.y);
            r3.z = dpp_mov<DPP_ROW_ROR12>(xA.z); r3.w = dpp_mov<DPP_ROW_ROR12>(xA.w);
            bm.x = (row & 1) ? xB.x : xA.x; bm.y = (row & 1) ? xB.y : xA.y;
            bm.z = (row & 1) ? xB.z : xA.z; bm.w = (row & 1) ? xB.w : xA.w;
            __builtin_amdgcn_sched_barrier(0);
            mfma4_operands_settled();
            // (the action part left the main chains at acc2.)  The mini job's four MFMAs go first; its reduction -- the 8
            // blocks of a quad (two rows) hold its partial sums: fold the rows (two registers per swap), then the row's
            // four blocks as a reduce-scatter (one value, one activation per lane, valid in the odd blocks) -- is a chain
            // of ~15 dependent VALU instructions, fed one link at a time between the remaining MFMAs, whose issue time
            // covers its latency (behind the MFMAs it cost the mini wave, and through the barrier everyone, ~230 cycles).
            // Every state wave runs it (zero weights off wave MJ); only wave MJ stores.
#define Q4R_FENCE() __builtin_amdgcn_sched_barrier(0)
            mfma4_v(accM, wM[0], bm.x);
            mfma4_v(acc2, wA0r[0], xA.x);  mfma4_v(acc0, wA0r[1], xA.y);
            mfma4_v(accM, wM[1], bm.y);
            mfma4_v(acc1, wA0r[2], xA.z);  mfma4_v(acc2, wA0r[3], xA.w);
            mfma4_v(accM, wM[2], bm.z);
            mfma4_v(acc0, wA0r[4], r1.x);  mfma4_v(acc1, wA0r[5], r1.y);
            mfma4_v(accM, wM[3], bm.w);
            mfma4_v(acc2, wA0r[6], r1.z);  mfma4_v(acc0, wA0r[7], r1.w);  mfma4_v(acc1, wA0r[8], r2.x);
            asm volatile("" : "+v"(accM));                 // three MFMAs (>= 24 cycles) behind its last one: safe to read
            Q4R_FENCE();
            float mx = accM.x, my = accM.y, mz = accM.z, mw = accM.w;
            swap16(mx, my);                                  // mx + my = [x.r0 + x.r1, y.r0 + y.r1, x.r2 + x.r3, y.r2 + y.r3]
            swap16(mz, mw);
            Q4R_FENCE();
            mfma4_v(acc2, wA0r[9], r2.y);  mfma4_v(acc0, wA0r[10], r2.z);
            Q4R_FENCE();
            const float s1 = mx + my, s2 = mz + mw;
            Q4R_FENCE();
            mfma4_v(acc1, wA0r[11], r2.w); mfma4_v(acc2, wA0r[12], r3.x);
            Q4R_FENCE();
            const float u1 = s1 + dpp_mov<DPP_ROW_ROR8>(s1), u2 = s2 + dpp_mov<DPP_ROW_ROR8>(s2);
            Q4R_FENCE();
            mfma4_v(acc0, wA0r[13], r3.y); mfma4_v(acc1, wA0r[14], r3.z);
            Q4R_FENCE();
            const float tt = (fgq < 2) ? u1 : u2;
            Q4R_FENCE();
            mfma4_v(acc2, wA0r[15], r3.w);
            Q4R_FENCE();
            // the other block of my pair sits one block down for the odd blocks, which therefore finish the sum and store (a
            // lane-dependent choice between two DPP sources would put the DPP under a partial exec mask)
            const float pre = (tt + dpp_mov<DPP_ROW_ROR4>(tt)) + bMs;
            Q4R_FENCE();
            mfma4_v(acc0, wA0b[0], xB.x);  mfma4_v(acc1, wA0b[1], xB.y);
            Q4R_FENCE();
            const float hv = apply_act_ct<A0>(pre);
            Q4R_FENCE();
            mfma4_v(acc2, wA0b[2], xB.z);  mfma4_v(acc0, wA0b[3], xB.w);
            mfma4_results_ready(acc0, acc1, acc2);
            Q4R_FENCE();
#undef Q4R_FENCE
            f32x4 o;
            o.x = apply_act_ct<A0>((acc0.x + acc1.x) + acc2.x); o.y = apply_act_ct<A0>((acc0.y + acc1.y) + acc2.y);
            o.z = apply_act_ct<A0>((acc0.z + acc1.z) + acc2.z); o.w = apply_act_ct<A0>((acc0.w + acc1.w) + acc2.w);
            *reinterpret_cast<f32x4*>(h0 + my_row) = o;
            if (mini && (fgq & 1)) *h0m = hv;
        }
        Q4R_MARK(0);
        __syncthreads();
        Q4R_MARK(1);
        f32x4 ba0, ba1;
        // ---- layer 1: three 16-feature jobs per wave (156 MFMAs on every wave), K split over the rows, one shared set of
        // B operands; stationary A operands in AccVGPRs, read by the MFMA directly
        {
            f32x4 bq[KA];
            f32x4 cj0, cj1, cj2;
#pragma unroll
            for (int c = 0; c < KA; ++c) bq[c] = *reinterpret_cast<const f32x4*>(hA0 + c * 16);
            {   // next step's normalised action groups: static data, fetched here so that the action part can start right
                // behind the second barrier
                const int tn = (t + 1 < H) ? t + 1 : t;
                ba0 = *reinterpret_cast<const f32x4*>(xa + ((size_t)(tn * AG + 0) * 4 + pl) * 4);
                ba1 = *reinterpret_cast<const f32x4*>(xa + ((size_t)(tn * AG + 1) * 4 + pl) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            Q4R_CYC0();
            mfma4_a_round3_first(cj0, cj1, cj2, wJ[0], wJ[1], wJ[2], bq[0]);
#pragma unroll
            for (int c = 1; c < KA; ++c) mfma4_a_round3(cj0, cj1, cj2, wJ[0] + 4 * c, wJ[1] + 4 * c, wJ[2] + 4 * c, bq[c]);
            mfma4_results_ready(cj0, cj1, cj2);
            __builtin_amdgcn_sched_barrier(0);
            Q4R_CYC(0);
            Q4R_CYC0();
            h1w[0 * 256] = apply_act_ct<A1>(rows_reduce_scatter(cj0) + b1s[0]);
            h1w[1 * 256] = apply_act_ct<A1>(rows_reduce_scatter(cj1) + b1s[1]);
            h1w[2 * 256] = apply_act_ct<A1>(rows_reduce_scatter(cj2) + b1s[2]);
            Q4R_CYC(1);
        }
        Q4R_MARK(2);
        __syncthreads();
        Q4R_MARK(3);
        // ---- the last layer, whole K in every wave, K-split over the MFMA's blocks (set A: features 0..15 in 4 quads x 4
        // row slices; set B: features 16..19, block b takes k groups b, b + 16, ...), its rounds ordered so that those
        // reading job 12's output come last; the action part of the NEXT step's layer 0 fills the LDS latency
        f32x4 cA0, cA1, cA2, cB0, cB1;
        f32x4 bqA[KA], bqB[KB];
        {
#pragma unroll
            for (int c = 0; c < KB - 1; ++c) bqB[c] = *reinterpret_cast<const f32x4*>(hB1 + c * 16 * 16);
#pragma unroll
            for (int c = 0; c < 6; ++c) bqA[c] = *reinterpret_cast<const f32x4*>(hA1 + c * 16);
            Q4R_ACTION_PART(ba0, ba1);
            __builtin_amdgcn_sched_barrier(0);
            Q4R_CYC0();
            // part 1, 36 MFMAs: rounds 0..2 of both sets (A0 B0 A1 B1 A2 B0 A0 B1 | A1 B0 A2 B1 A0 B0 A1 B1 | ...), rounds 3..5 of set A
            mfma4_a_0(cA0, wU[0], bqA[0].x);        mfma4_v_0(cB0, wA2b[0], bqB[0].x);
            mfma4_a_0(cA1, wU[1], bqA[0].y);        mfma4_v_0(cB1, wA2b[1], bqB[0].y);
            mfma4_a_0(cA2, wU[2], bqA[0].z);        mfma4_v(cB0, wA2b[2], bqB[0].z);
            mfma4_a(cA0, wU[3], bqA[0].w);          mfma4_v(cB1, wA2b[3], bqB[0].w);
#pragma unroll
            for (int c = 1; c < 6; ++c) {
                // set A's MFMA number i = 4c + e goes to chain i % 3 (compile-time after unrolling)
                f32x4& a0_ = ((4 * c + 0) % 3 == 0) ? cA0 : ((4 * c + 0) % 3 == 1) ? cA1 : cA2;
                f32x4& a1_ = ((4 * c + 1) % 3 == 0) ? cA0 : ((4 * c + 1) % 3 == 1) ? cA1 : cA2;
                f32x4& a2_ = ((4 * c + 2) % 3 == 0) ? cA0 : ((4 * c + 2) % 3 == 1) ? cA1 : cA2;
                f32x4& a3_ = ((4 * c + 3) % 3 == 0) ? cA0 : ((4 * c + 3) % 3 == 1) ? cA1 : cA2;
                if (c < KB - 1) {
                    mfma4_a(a0_, wU[c * 4 + 0], bqA[c].x); mfma4_v(cB0, wA2b[c * 4 + 0], bqB[c].x);
                    mfma4_a(a1_, wU[c * 4 + 1], bqA[c].y); mfma4_v(cB1, wA2b[c * 4 + 1], bqB[c].y);
                    mfma4_a(a2_, wU[c * 4 + 2], bqA[c].z); mfma4_v(cB0, wA2b[c * 4 + 2], bqB[c].z);
                    mfma4_a(a3_, wU[c * 4 + 3], bqA[c].w); mfma4_v(cB1, wA2b[c * 4 + 3], bqB[c].w);
                } else {
                    if (c == KB - 1) {                    // the B operands of part 2, into the registers rounds 0..2 have released
                        __builtin_amdgcn_sched_barrier(0);
                        bqA[6] = *reinterpret_cast<const f32x4*>(hA1 + 6 * 16);
                        bqA[7] = *reinterpret_cast<const f32x4*>(hA1 + 7 * 16);
                        bqA[8] = *reinterpret_cast<const f32x4*>(hA1 + 8 * 16);
                        bqA[LA0 + 2] = *reinterpret_cast<const f32x4*>(hA1 + (LA0 + 2) * 16);
                        bqA[LA0 + 3] = *reinterpret_cast<const f32x4*>(hA1 + (LA0 + 3) * 16);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    mfma4_a(a0_, wU[c * 4 + 0], bqA[c].x); mfma4_a(a1_, wU[c * 4 + 1], bqA[c].y);
                    mfma4_a(a2_, wU[c * 4 + 2], bqA[c].z); mfma4_a(a3_, wU[c * 4 + 3], bqA[c].w);
                }
            }
        }
        Q4R_MARK(9);
        __syncthreads();                                   // job 12's output is in h1 (the helper got here first)
        Q4R_MARK(10);
        {
            bqB[KB - 1] = *reinterpret_cast<const f32x4*>(hB1 + (KB - 1) * 16 * 16);
            bqA[LA0] = *reinterpret_cast<const f32x4*>(hA1 + LA0 * 16);
            bqA[LA0 + 1] = *reinterpret_cast<const f32x4*>(hA1 + (LA0 + 1) * 16);
            __builtin_amdgcn_sched_barrier(0);
            // part 2, 20 MFMAs that do not need them: set A rounds 6..8 and 11, 12 (chains keep rotating: 24 MFMAs so far)
            {
                constexpr int order[5] = {6, 7, 8, LA0 + 2, LA0 + 3};
                static_assert(LA0 == 9 && KA == 13, "round order of the last layer");
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int c = order[j];
                    f32x4& a0_ = ((4 * (6 + j) + 0) % 3 == 0) ? cA0 : ((4 * (6 + j) + 0) % 3 == 1) ? cA1 : cA2;
                    f32x4& a1_ = ((4 * (6 + j) + 1) % 3 == 0) ? cA0 : ((4 * (6 + j) + 1) % 3 == 1) ? cA1 : cA2;
                    f32x4& a2_ = ((4 * (6 + j) + 2) % 3 == 0) ? cA0 : ((4 * (6 + j) + 2) % 3 == 1) ? cA1 : cA2;
                    f32x4& a3_ = ((4 * (6 + j) + 3) % 3 == 0) ? cA0 : ((4 * (6 + j) + 3) % 3 == 1) ? cA1 : cA2;
                    mfma4_a(a0_, wU[c * 4 + 0], bqA[c].x); mfma4_a(a1_, wU[c * 4 + 1], bqA[c].y);
                    mfma4_a(a2_, wU[c * 4 + 2], bqA[c].z); mfma4_a(a3_, wU[c * 4 + 3], bqA[c].w);
                }
            }
            // part 3, 12 MFMAs: set A rounds 9, 10 (issue numbers 44..51 -> chains 2 0 1 2 0 1 2 0) and set B round 3
            mfma4_a(cA2, wU[LA0 * 4 + 0], bqA[LA0].x); mfma4_v(cB0, wA2b[(KB - 1) * 4 + 0], bqB[KB - 1].x);
            mfma4_a(cA0, wU[LA0 * 4 + 1], bqA[LA0].y); mfma4_v(cB1, wA2b[(KB - 1) * 4 + 1], bqB[KB - 1].y);
            mfma4_a(cA1, wU[LA0 * 4 + 2], bqA[LA0].z); mfma4_v(cB0, wA2b[(KB - 1) * 4 + 2], bqB[KB - 1].z);
            mfma4_a(cA2, wU[LA0 * 4 + 3], bqA[LA0].w); mfma4_v(cB1, wA2b[(KB - 1) * 4 + 3], bqB[KB - 1].w);
            mfma4_a(cA0, wU[(LA0 + 1) * 4 + 0], bqA[LA0 + 1].x); mfma4_a(cA1, wU[(LA0 + 1) * 4 + 1], bqA[LA0 + 1].y);
            mfma4_a(cA2, wU[(LA0 + 1) * 4 + 2], bqA[LA0 + 1].z); mfma4_a(cA0, wU[(LA0 + 1) * 4 + 3], bqA[LA0 + 1].w);
            mfma4_results_ready(cA0, cA1, cA2, cB0, cB1);
            __builtin_amdgcn_sched_barrier(0);
            Q4R_CYC(2);
            Q4R_CYC0();
            f32x4 sB = {cB0.x + cB1.x, cB0.y + cB1.y, cB0.z + cB1.z, cB0.w + cB1.w};
            const f32x4 sA = {(cA0.x + cA1.x) + cA2.x, (cA0.y + cA1.y) + cA2.y, (cA0.z + cA1.z) + cA2.z, (cA0.w + cA1.w) + cA2.w};
            // set B: the row's four blocks (8 first: every block then adds the same two pairs)
            sB.x = sB.x + dpp_mov<DPP_ROW_ROR8>(sB.x); sB.y = sB.y + dpp_mov<DPP_ROW_ROR8>(sB.y);
            sB.z = sB.z + dpp_mov<DPP_ROW_ROR8>(sB.z); sB.w = sB.w + dpp_mov<DPP_ROW_ROR8>(sB.w);
            sB.x = sB.x + dpp_mov<DPP_ROW_ROR4>(sB.x); sB.y = sB.y + dpp_mov<DPP_ROW_ROR4>(sB.y);
            sB.z = sB.z + dpp_mov<DPP_ROW_ROR4>(sB.z); sB.w = sB.w + dpp_mov<DPP_ROW_ROR4>(sB.w);
            zA = rows_reduce_scatter(sA) + lbA;           // output feature 4*fgq + pr of particle pl
            zB = rows_reduce_scatter(sB) + lbB;           // output feature 16 + pr
            // ---- epilogue on the two features this lane finishes (process_output, then process_input of the next step)
            zA = apply_act_ct<A2>(zA); zB = apply_act_ct<A2>(zB);
            const float vA = (tmA + zA * tsA) + curA, vB = (tmB + zB * tsB) + curB;
            if (rwd_flag_lane) rwd_f[t * 16] = (curA >= flag_thr) ? -10.0f : 0.0f;
            if (rwd_prog_lane) rwd_d[t * 16] = vB - curB;
            curA = vA; curB = vB;
            if (q.traj) {                                  // state after step t, for a user reward function
                const f32x4 rA = rows_all_gather(vA), rB = rows_all_gather(vB);
                if (n0 + pl < p.n_pop) {
                    float* dst = q.traj + ((((size_t)t * p.A + a) * p.Nst) + n0 + pl) * S;
                    if (wave == 0 && row == 0) *reinterpret_cast<f32x4*>(dst + 4 * fgq) = rA;
                    if (wave == 0 && row == 1 && fgq == 0) *reinterpret_cast<f32x4*>(dst + 16) = rB;
                }
            }
            xA = rows_all_gather((vA - nmA) * niA);
            xB = rows_all_gather((vB - nmB) * niB);
            Q4R_CYC(3);
        }
        Q4R_MARK(4);
    }
    }
#undef Q4R_ACTION_PART
#ifdef BBMPC_KERNEL_DBG
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        printf("[q4rdbg] H=%d | small loads issued %lld  operand loads issued %lld  constants+actions %lld  LDS part 2 %lld  first mfma %lld | loop %lld (%lld shader cycles): layer0 %lld bar %lld layer1 %lld bar %lld action+last part 1 %lld bar %lld last parts 2,3+epilogue %lld (10ns units)\n",
               H, dbg_acc[11], dbg_acc[5], dbg_acc[6], dbg_acc[8], dbg_acc[7], (long long)wall_clock64() - dbg_start, (long long)clock64() - dbg_cyc0, dbg_acc[0], dbg_acc[1], dbg_acc[2], dbg_acc[3], dbg_acc[9], dbg_acc[10], dbg_acc[4]);
        printf("[q4rdbg] shader cycles per step: layer-1 MFMA block (156) %lld | layer-1 reduce+tanh+store %lld | last-layer MFMA block (68) %lld | reduce+epilogue+gather %lld\n",
               dbg_cyc[0] / H, dbg_cyc[1] / H, dbg_cyc[2] / H, dbg_cyc[3] / H);
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

inline int mlp_q4r_lds_floats(int HG, int K0G, int H, int U) {
    return 2 * 64 * 16 + ((H * 4 * U + 3) & ~3) + ((4 * U + 3) & ~3) + H * (K0G - 5) * 16 + H * 4 + H * 16 + 6 * 32 + 8;
}

}  // namespace bbmpc
