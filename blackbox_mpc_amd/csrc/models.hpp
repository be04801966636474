// Leaf math of the hot path as device functors: analytic pendulum model and the
// two reward functions.  fp32, one rounding per reference TF op (the library is
// built with -ffp-contract=off so no multiply-add is fused behind our back).
#pragma once
#include <hip/hip_runtime.h>

#include "fastmath.hpp"

namespace bbmpc {

#define BBMPC_PI_F     3.14159274101257324f   /* float32(np.pi)   */
#define BBMPC_TWO_PI_F 6.28318548202514648f   /* float32(2*np.pi) */

// tf.clip_by_value semantics that let NaN through (min(max(x,lo),hi) on NaN stays NaN in TF/Eigen)
__device__ __forceinline__ float clipf(float x, float lo, float hi) {
    return (x < lo) ? lo : ((x > hi) ? hi : x);
}

// TF FloorMod on floats (used by `%` in utils/pendulum.py:7): fmod, then move into the divisor's sign.
__device__ __forceinline__ float floormodf(float x, float y) {
    if (y > 0.0f) return bb_floormod_pos(x, y);
    float r = fmodf(x, y);
    if (r != 0.0f && ((y < 0.0f) != (r < 0.0f))) r = r + y;
    return r;
}

// reward kinds (bbmpc.h)
constexpr int REW_NONE = 0;       // the reward is somebody else's business (a user device function evaluates it, kernels_user.hpp)
constexpr int REW_PENDULUM = 1;
constexpr int REW_CHEETAH = 2;

// pendulum_reward_function utils/pendulum.py:10-35 AS EXECUTED by
// trajectory_evaluators/deterministic.py:65-66 (quirk Q1): the third positional
// argument (named `actions`) receives next_state.  theta = atan2(cur[1],cur[0]).
__device__ __forceinline__ float pendulum_reward_from_theta(float theta, float thdot, float act_term_sumsq) {
    float ang = bb_floormod_pos(theta + BBMPC_PI_F, BBMPC_TWO_PI_F) - BBMPC_PI_F;   // :5-7
    float first = ang * ang + 0.1f * (thdot * thdot);
    return (-first) - 0.001f * act_term_sumsq;
}

// Generic-S reward dispatch used by the learned-dynamics path and the single-step API.
// cur/nxt have S entries, act has U entries.
__device__ __forceinline__ float reward_generic(int kind, bool fix_q1, const float* cur, const float* act,
                                                const float* nxt, int S, int U) {
    if (kind == REW_NONE) return 0.0f;
    if (kind == REW_PENDULUM) {
        float theta = bb_atan2f(cur[1], cur[0]);
        float ss = 0.0f;
        if (fix_q1) {
            for (int u = 0; u < U; ++u) ss = ss + act[u] * act[u];
        } else {
            for (int s = 0; s < S; ++s) ss = ss + nxt[s] * nxt[s];
        }
        return pendulum_reward_from_theta(theta, cur[2], ss);
    }
    // reward_function tutorials/mujoco/cost_func.py:5-22 (HalfCheetahEnvModified, S=20)
    float r = 0.0f;
    if (cur[5] >= 0.2f) r = r + (-10.0f);
    if (cur[6] >= 0.0f) r = r + (-10.0f);
    if (cur[7] >= 0.0f) r = r + (-10.0f);
    r = r + (nxt[17] - cur[17]) / 0.01f;
    float ss = 0.0f;
    for (int u = 0; u < U; ++u) ss = ss + act[u] * act[u];
    r = r - 0.0f * ss;
    return r;
}

// PendulumTrueModel.__call__ utils/pendulum.py:58-92 + true-model handler
// (process_input = concat, process_output = delta + state, system_dynamics_handler.py:116-118,149-151)
// + as-executed reward, fused.  State s = (cos th, sin th, thdot) is advanced in place;
// returns the step reward.  Quirk Q9: th integrates the unclipped speed, torque unclipped.
struct PendulumModel {
    static constexpr int S = 3;
    static constexpr int U = 1;
    bool fix_q1;

    __device__ __forceinline__ float step(float (&s)[3], const float (&a)[1]) const {
        const float u = a[0];
        const float theta = bb_atan2f(s[1], s[0]);              // :82 (the reward's atan2 has identical inputs)
        float acc = -15.0f * bb_sinf(theta + BBMPC_PI_F);       // -3g/(2l) = -15
        acc = acc + 3.0f * u;                                   // 3/(m l^2) = 3
        float nthd = s[2] + acc * 0.05f;                        // :83-85
        const float nth = theta + nthd * 0.05f;                 // :86 (unclipped speed)
        nthd = clipf(nthd, -8.0f, 8.0f);                        // :87
        float sn, cs;
        bb_sincosf(nth, &sn, &cs);
        // deviation = new - x[:, :3]; next = deviation + current   (:91, transforms.py:34)
        const float n0 = (cs - s[0]) + s[0];
        const float n1 = (sn - s[1]) + s[1];
        const float n2 = (nthd - s[2]) + s[2];
        float ss;
        if (fix_q1) ss = u * u;
        else ss = (n0 * n0 + n1 * n1) + n2 * n2;
        const float r = pendulum_reward_from_theta(theta, s[2], ss);
        s[0] = n0; s[1] = n1; s[2] = n2;
        return r;
    }
};

// The same recurrence carried as (theta, thdot) instead of (cos, sin, thdot).
//
// The reference re-derives theta = atan2(sin, cos) from the state every step (utils/pendulum.py:82 and
// again in the reward :27) after having just computed sin/cos of the new angle (:88-89).  Since
// atan2(sin x, cos x) is the principal value of x, the angle can be carried directly: per step this
// drops one atan2 and one sincos (about 3/4 of the instructions).  What changes numerically is only
// fp32 rounding noise of the 1e-7 class the libm choice already introduces (the reference's own
// (cos - s0) + s0 round trip perturbs the angle by up to 0.5 ulp(1)); sum(next_state**2) uses
// cos^2 + sin^2 = 1.  The strict, op-for-op PendulumModel above stays available
// (BBMPC_STRICT_MATH) and both are held to the same parity tolerances.
struct PendulumAngleModel {
    bool fix_q1;
    float theta, thd;

    __device__ __forceinline__ void init(float s0, float s1, float s2) {
        theta = bb_atan2f(s1, s0);
        thd = s2;
    }
    __device__ __forceinline__ float step(float u) {
        const float t1 = theta + BBMPC_PI_F;
        float acc = -15.0f * bb_sinf_fold_0_2pi(t1);            // the reference's sin(theta + pi), argument rounded as there
        acc = acc + 3.0f * u;
        float nthd = thd + acc * 0.05f;
        const float nth = theta + nthd * 0.05f;
        // one v_med3_f32 instead of two compare/select pairs.  (A NaN speed becomes -8 here, but theta is NaN in that
        // case as well and keeps the step reward NaN, which is all the evaluator's NaN guard looks at.)
        nthd = __builtin_amdgcn_fmed3f(nthd, -8.0f, 8.0f);
        const float n2 = (nthd - thd) + thd;
        const float ss = fix_q1 ? u * u : 1.0f + n2 * n2;
        // t1 is in [0, 2pi] (theta is a principal value): FloorMod reduces to one exact conditional subtract
        const float ang = ((t1 >= BBMPC_TWO_PI_F) ? t1 - BBMPC_TWO_PI_F : t1) - BBMPC_PI_F;
        const float first = ang * ang + 0.1f * (thd * thd);
        const float r = (-first) - 0.001f * ss;
        theta = bb_wrap_pi(nth);
        thd = n2;
        return r;
    }
};

// The carried angle in TURNS (phi = theta / 2pi, principal value in [-0.5, 0.5]) with the hardware sine: v_sin_f32 takes
// its argument in turns and is good to 1.25e-7 absolute on [-0.5, 0.5] (tools/microbench/hw_sin.hip: one ulp of 1 -- the
// class of the reference's own float32 rounding of theta + pi), the wrap to the principal value is the exact
// phi - rint(phi), and sin(theta + pi) = -sin(theta), angle_normalize(theta) = theta need no +-pi at all.  A model step's
// dependent chain is sin -> 2 x (mul, add) -> mul, add -> rint, sub: ~12 operations where the radian form (fold to
// [-pi/2, pi/2], degree-9 polynomial, two-constant wrap) has ~30 -- and the persistent pendulum kernels ARE that chain,
// 150 model steps per control step.  Same parity tolerances as the two forms above (tests/test_gpu_pendulum.py).
#ifndef BBMPC_PENDULUM_HW_SIN
#define BBMPC_PENDULUM_HW_SIN 2
#endif
#define BBMPC_INV_TWO_PI_F 0.15915494309189535f
#ifndef BBMPC_PENDULUM_SUMS
#define BBMPC_PENDULUM_SUMS 1
#endif
struct PendulumTurnModel {
    bool fix_q1;
    float phi, thd;
#if BBMPC_PENDULUM_HW_SIN == 2
    float sn;                            // sin(theta) of the CURRENT state: issued the moment the angle is known (see step)
    float q_ang, q_thd, q_act;           // step_acc: running sums of phi^2, thdot^2 and the action-cost term's squares
    int q_n;
#endif

    __device__ __forceinline__ static float sin_turns(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_sinf(x);
#else
        return sinf(x * BBMPC_TWO_PI_F);
#endif
    }
    __device__ __forceinline__ void init(float s0, float s1, float s2) {
        phi = bb_atan2f(s1, s0) * BBMPC_INV_TWO_PI_F;
        thd = s2;
#if BBMPC_PENDULUM_HW_SIN == 2
        sn = sin_turns(phi);
        q_ang = q_thd = q_act = 0.0f;
        q_n = 0;
#endif
    }
#if BBMPC_PENDULUM_HW_SIN == 2
    // One model step whose reward is not formed: the H-step sum  -sum(ang^2 + 0.1 thdot^2 + 0.001 ss)  is kept as three
    // sums of squares (one fma each per step) and put together once in total() -- nine instructions per step less in the
    // persistent kernels' rollout, which is instruction-issue bound (two waves per SIMD running this recurrence).  Float32
    // rounding differs from the step-by-step sum at the 1e-7 relative level, the class of the reference's own.
    __device__ __forceinline__ void step_acc(float u) {
        constexpr float C = 0.05f * BBMPC_INV_TWO_PI_F;
        const float b = fmaf(0.15f, u, thd);
        const float a = fmaf(b, C, phi);
        const float nphi = fmaf(0.75f * C, sn, a);
        const float phi0 = phi, sn0 = sn;
        phi = nphi - rintf(nphi);
        sn = sin_turns(phi);
        __builtin_amdgcn_sched_barrier(0);
        float nthd = fmaf(0.75f, sn0, b);
        nthd = __builtin_amdgcn_fmed3f(nthd, -8.0f, 8.0f);
        const float n2 = (nthd - thd) + thd;
        q_ang = fmaf(phi0, phi0, q_ang);
        q_thd = fmaf(thd, thd, q_thd);
        q_act = fix_q1 ? fmaf(u, u, q_act) : fmaf(n2, n2, q_act);
        q_n += 1;
        thd = n2;
    }
    __device__ __forceinline__ float total() const {
        const float ss = fix_q1 ? q_act : (float)q_n + q_act;               // sum(next_state^2) = 1 + newthdot^2 per step (quirk Q1)
        const float first = (BBMPC_TWO_PI_F * BBMPC_TWO_PI_F) * q_ang + 0.1f * q_thd;
        return (-first) - 0.001f * ss;
    }
#endif
    __device__ __forceinline__ float step(float u) {
#if BBMPC_PENDULUM_HW_SIN == 2
        // newthdot = thdot + (15 sin(theta) + 3u) dt and newth = theta + newthdot dt with everything that does not need the
        // sine formed beside it: behind the sine the angle is ONE fma + the wrap, and the next step's sine is issued right
        // there -- the recurrence a model step waits for is sin -> fma -> rint -> sub -> sin, the speed, the reward and
        // the next action fill the transcendental's latency
        constexpr float C = 0.05f * BBMPC_INV_TWO_PI_F;                    // dt in turns per radian
        const float b = fmaf(0.15f, u, thd);                                // thdot + 3 u dt
        const float a = fmaf(b, C, phi);                                    // theta + (thdot + 3 u dt) dt
        const float nphi = fmaf(0.75f * C, sn, a);                          // + 15 sin(theta) dt dt  (the unclipped speed, as the reference: quirk Q9)
        const float phi0 = phi, sn0 = sn;
        phi = nphi - rintf(nphi);                                           // principal value, exact
        sn = sin_turns(phi);
        __builtin_amdgcn_sched_barrier(0);                                  // (the chain first, in this order)
        float nthd = fmaf(0.75f, sn0, b);                                   // thdot + (15 sin(theta) + 3u) dt
        nthd = __builtin_amdgcn_fmed3f(nthd, -8.0f, 8.0f);                  // (a NaN speed: see PendulumAngleModel)
        const float n2 = (nthd - thd) + thd;
        const float ss = fix_q1 ? u * u : 1.0f + n2 * n2;
        const float ang = phi0 * BBMPC_TWO_PI_F;                            // angle_normalize(theta): theta is a principal value
        const float first = ang * ang + 0.1f * (thd * thd);
        const float r = (-first) - 0.001f * ss;
        thd = n2;
        return r;
#else
        const float sn1 = sin_turns(phi);                                   // sin(theta)
        float acc = 15.0f * sn1;                                            // -3g/(2l) sin(theta + pi)
        acc = acc + 3.0f * u;
        float nthd = thd + acc * 0.05f;
        const float nphi = phi + nthd * (0.05f * BBMPC_INV_TWO_PI_F);       // theta + newthdot * dt, in turns (before the clip, as the reference)
        nthd = __builtin_amdgcn_fmed3f(nthd, -8.0f, 8.0f);                  // (a NaN speed: see PendulumAngleModel)
        const float n2 = (nthd - thd) + thd;
        const float ss = fix_q1 ? u * u : 1.0f + n2 * n2;
        const float ang = phi * BBMPC_TWO_PI_F;                             // angle_normalize(theta): theta is a principal value
        const float first = ang * ang + 0.1f * (thd * thd);
        const float r = (-first) - 0.001f * ss;
        phi = nphi - rintf(nphi);                                           // principal value, exact
        thd = n2;
        return r;
#endif
    }
};

// Uniform rollout interface over the two formulations.
template <bool FASTM>
struct Roller;
template <>
struct Roller<false> {
    PendulumModel m;
    float s[3];
    __device__ __forceinline__ Roller() : m{false} {}
    __device__ __forceinline__ Roller(bool fix_q1, float s0, float s1, float s2) : m{fix_q1} {
        s[0] = s0; s[1] = s1; s[2] = s2;
    }
    __device__ __forceinline__ void init(bool fix_q1, float s0, float s1, float s2) {
        m.fix_q1 = fix_q1;
        s[0] = s0; s[1] = s1; s[2] = s2;
    }
    __device__ __forceinline__ float step(float u) {
        const float a[1] = {u};
        return m.step(s, a);
    }
    // the H-step sum kept by the roller (Roller<true> has a cheaper form of it)
    float acc_ = 0.0f;
    __device__ __forceinline__ void step_acc(float u) { acc_ = acc_ + step(u); }
    __device__ __forceinline__ float total() const { return acc_; }
};
template <>
struct Roller<true> {
#if BBMPC_PENDULUM_HW_SIN
    PendulumTurnModel m;
#else
    PendulumAngleModel m;
#endif
    __device__ __forceinline__ Roller() {}
    __device__ __forceinline__ Roller(bool fix_q1, float s0, float s1, float s2) {
        m.fix_q1 = fix_q1;
        m.init(s0, s1, s2);
    }
    __device__ __forceinline__ void init(bool fix_q1, float s0, float s1, float s2) {
        m.fix_q1 = fix_q1;
        m.init(s0, s1, s2);
    }
    __device__ __forceinline__ float step(float u) { return m.step(u); }
#if BBMPC_PENDULUM_HW_SIN == 2 && BBMPC_PENDULUM_SUMS
    __device__ __forceinline__ void step_acc(float u) { m.step_acc(u); }
    __device__ __forceinline__ float total() const { return m.total(); }
#else
    float acc_ = 0.0f;
    __device__ __forceinline__ void step_acc(float u) { acc_ = acc_ + m.step(u); }
    __device__ __forceinline__ float total() const { return acc_; }
#endif
};

}  // namespace bbmpc
