// Range-limited fp32 sin / cos / atan2 for the pendulum recurrence.
//
// The device libm (ocml) versions are general-purpose: argument reduction for
// |x| up to 2^128 (Payne-Hanek branches), a looping fmodf, etc.  On the hot path
// every angle is bounded (theta in [-pi,pi], theta+pi in [0,2pi], theta' within a
// few radians), so a 2-term Cody-Waite reduction + short polynomials do the same
// job in a fraction of the instructions.  Accuracy (tests/test_fastmath.py sweeps
// them on the HOST, where they compile bit-identically: only IEEE +,-,*,/ and
// fmaf are used):  sin/cos <= 1.6 ulp (mean 0.33) for |x| <= 8, atan2 <= 2 ulp --
// the same class as ocml / CUDA / Eigen's vectorised sin-cos that TF-CPU uses.
// Outside the fast domain each function falls back to the precise libm call, so
// results stay correct for any input (NaN/inf included).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD static inline
#endif

namespace bbmpc {

// pi/2 = C1 + C2 (+ ~1e-15)
#define BB_PIO2_HI 1.5707963705062866f
#define BB_PIO2_LO (-4.371138828673793e-08f)
#define BB_PI_HI 3.1415927410125732f
#define BB_PI_LO (-8.742277657347586e-08f)

// sin and cos of x; fast path for |x| <= 8.
BB_HD void bb_sincosf(float x, float* sn_out, float* cs_out) {
    if (!(fabsf(x) <= 8.0f)) {          // also catches NaN
        *sn_out = sinf(x);
        *cs_out = cosf(x);
        return;
    }
    const float kf = rintf(x * 0.6366197723675814f);          // 2/pi
    float r = fmaf(-kf, BB_PIO2_HI, x);
    r = fmaf(-kf, BB_PIO2_LO, r);
    const float s = r * r;
    // sin(r) = r + r*s*ps(s),  cos(r) = 1 - s/2 + s*s*pc(s)   on |r| <= pi/4
    float ps = 2.724304977164138e-06f;
    ps = fmaf(ps, s, -0.00019840050663333386f);
    ps = fmaf(ps, s, 0.008333331905305386f);
    ps = fmaf(ps, s, -0.1666666716337204f);
    const float sn = fmaf(r * s, ps, r);
    float pc = -3.619722122039093e-07f;
    pc = fmaf(pc, s, 2.490056249371264e-05f);
    pc = fmaf(pc, s, -0.0013889208203181624f);
    pc = fmaf(pc, s, 0.0416666679084301f);
    const float cs = fmaf(s * s, pc, fmaf(s, -0.5f, 1.0f));
    const int q = ((int)kf) & 3;
    const float a = (q & 1) ? cs : sn;      // sin(x)
    const float b = (q & 1) ? sn : cs;      // cos(x) up to sign
    *sn_out = (q & 2) ? -a : a;
    *cs_out = ((q + 1) & 2) ? -b : b;
}

// sin(x) for a principal-value angle x in [-pi-eps, pi+eps]: one reflection onto [-pi/2, pi/2]
// (sin(pi - x) = sin x, pi split hi+lo) and a single odd polynomial -- no quadrant bookkeeping, no cosine.
// <= 1.9 ulp (mean 0.32), tests/test_fastmath.py.
BB_HD float bb_sinf_pi(float x) {
    const float ax = fabsf(x);
    const float xf = (BB_PI_HI - ax) + BB_PI_LO;               // pi - |x|   (first subtraction exact)
    const float r = (ax > BB_PIO2_HI) ? ((x < 0.0f) ? -xf : xf) : x;
    const float s = r * r;
    float p = -2.408602561843054e-08f;
    p = fmaf(p, s, 2.7536809739103774e-06f);
    p = fmaf(p, s, -0.0001984109403565526f);
    p = fmaf(p, s, 0.00833333283662796f);
    p = fmaf(p, s, -0.1666666716337204f);
    return fmaf(r * s, p, r);
}

// sin(x) for x in [0, 2pi] (+- a few ulp): fold x itself onto [-pi/2, pi/2] -- pi - x on the middle half,
// x - 2pi on the last quarter; the leading subtraction is exact (Sterbenz), the lo-part adds ~1e-15 -- and
// evaluate the odd polynomial of bb_sinf_pi.  Same accuracy, and it keeps the rounding of the caller's argument.
BB_HD float bb_sinf_fold_0_2pi(float x) {
    const float a = (BB_PI_HI - x) + BB_PI_LO;                                  // pi - x
    const float b = (x - 6.2831854820251465f) - (-1.7484555314695172e-07f);     // x - 2pi
    const float r = (x <= BB_PIO2_HI) ? x : ((x <= 4.71238898038469f) ? a : b);
    const float s = r * r;
    float p = -2.408602561843054e-08f;
    p = fmaf(p, s, 2.7536809739103774e-06f);
    p = fmaf(p, s, -0.0001984109403565526f);
    p = fmaf(p, s, 0.00833333283662796f);
    p = fmaf(p, s, -0.1666666716337204f);
    return fmaf(r * s, p, r);
}

// Principal value of an angle: x - 2pi*rint(x/2pi) in [-pi, pi], |x| < ~100.  This is what
// atan2(sin x, cos x) returns, without evaluating either.
BB_HD float bb_wrap_pi(float x) {
    const float kf = rintf(x * 0.15915494309189535f);         // 1/(2pi)
    float r = fmaf(-kf, 6.2831854820251465f, x);               // 2pi hi
    r = fmaf(-kf, -1.7484555314695172e-07f, r);                // 2pi lo
    return r;
}

BB_HD float bb_sinf(float x) {
    float s, c;
    bb_sincosf(x, &s, &c);
    return s;
}

// atan2(y, x), any finite inputs; special values defer to libm.
BB_HD float bb_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float sm = ax + ay;                                  // NaN if either is NaN
    if (!(sm > 1e-30f && sm < 1e30f)) return atan2f(y, x);    // zeros, inf, NaN, extreme scales
    const float a = mn / mx;                                  // in [0,1]
    const float s = a * a;
    float q = -0.002447031904011965f;
    q = fmaf(q, s, 0.01375028770416975f);
    q = fmaf(q, s, -0.036270178854465485f);
    q = fmaf(q, s, 0.06284361332654953f);
    q = fmaf(q, s, -0.08673170953989029f);
    q = fmaf(q, s, 0.11037994176149368f);
    q = fmaf(q, s, -0.14279110729694366f);
    q = fmaf(q, s, 0.1999976634979248f);
    q = fmaf(q, s, -0.3333333134651184f);
    float r = fmaf(a * s, q, a);                              // atan(a)
    if (ay > ax) r = (BB_PIO2_HI - r) + BB_PIO2_LO;           // pi/2 - r
    if (x < 0.0f) r = (BB_PI_HI - r) + BB_PI_LO;              // pi - r
    return copysignf(r, y);
}

// TF FloorMod(x, y) for y > 0 (the `%` of utils/pendulum.py:7): exact for every input.
// For 0 <= x < 2y the answer is x or x - y (Sterbenz: the subtraction is exact), which is the only
// case the pendulum reward ever produces (x = theta + pi in [0, 2pi]); everything else takes fmodf.
BB_HD float bb_floormod_pos(float x, float y) {
    if (x >= 0.0f && x < 2.0f * y) return (x >= y) ? x - y : x;
    float r = fmodf(x, y);
    if (r != 0.0f && r < 0.0f) r = r + y;
    return r;
}

}  // namespace bbmpc
