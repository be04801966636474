// Counter-based RNG of the engine (Philox4x32-10) and the transforms that turn
// raw words into the *standard* draws the optimizers consume.
//
// The reference draws from TensorFlow's stateful, never-seeded Philox streams
// (tf.random.truncated_normal cem.py:90 / pi2.py:65 / pso.py:121 /
// optimizer_base.py:83, tf.random.uniform random_search.py:40 / pso.py:130 /
// spsa.py:73, tf.random.normal pso.py:108-109 / cma_es.py:139).  Those streams
// are not reproducible even run-to-run of the reference, so the engine defines
// its own scheme, keyed so that (a) any element can be regenerated anywhere
// (no sample storage needed for it) and (b) results do not depend on how agents
// are sharded over GPUs:
//
//   key     = (seed_lo, seed_hi)
//   counter = ( n,                       particle index
//               ga * Q + (j >> 2),       ga = GLOBAL agent id, j = h*U+u, Q = ceil(H*U/4)
//               control_step,            number of optimize() calls so far on this handle
//               (stream << 16) | iter )  stream = BBMPC_NOISE_* kind, iter = optimizer iteration
//   element j uses output word (j & 3).
//
// Standard draws from a word x:
//   U(0,1)      : u = ((x >> 9) + 0.5) * 2^-23           (23 random mantissa bits, never 0 or 1)
//   trunc normal: inverse-CDF sampling of N(0,1) conditioned on |z| < 2 (the distribution
//                 tf.random.truncated_normal samples by rejection): one uniform per draw, no loop.
//                 The quantile q(u) = sqrt(2)*erfinv((2u-1)*erf(sqrt 2)) is tabulated at u = i/2048 (float64 ->
//                 fp32) and interpolated linearly: with v = x >> 9 (23 bits), i = v >> 12, f = ((v & 0xFFF)+0.5)/4096,
//                 z = fma(f, q[i+1]-q[i], q[i]).  |z - q(u)| <= 2e-5 (tails; ~1e-6 in the bulk) -- a sampler
//                 definition, not an approximation the results are compared through: oracle and engine consume
//                 the same z.  ~7 instructions per draw instead of log + degree-8 polynomial.
//   normal      : Box-Muller on word pairs (x0,x1),(x2,x3).
//   rademacher  : +1 if top bit set else -1.
#pragma once
#include <stdexcept>
#include <string>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbmpc {

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

__host__ __device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
#ifndef BBMPC_PHILOX_ROUNDS
#define BBMPC_PHILOX_ROUNDS 10
#endif
    for (int r = 0; r < BBMPC_PHILOX_ROUNDS; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c.x, hi0, lo0);
        mulhilo(0xCD9E8D57u, c.z, hi1, lo1);
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__host__ __device__ __forceinline__ float word_to_uniform(uint32_t x) {
    return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f;   // 2^-23
}

constexpr int TNQ_BITS = 11;
constexpr int TNQ_SIZE = 1 << TNQ_BITS;                    // 2048 intervals
// (q[i], q[i+1]-q[i]) pairs, filled once per device by the host (Engine constructor)
static __device__ float2 g_tnq[TNQ_SIZE];     // one copy per translation unit (no relocatable device code): tnq_upload fills it
static inline void tnq_upload(const float2* table) {
    hipError_t e_ = hipMemcpyToSymbol(HIP_SYMBOL(g_tnq), table, sizeof(float2) * TNQ_SIZE);
    if (e_ != hipSuccess) throw std::runtime_error(std::string("hipMemcpyToSymbol(g_tnq): ") + hipGetErrorString(e_));
}

__device__ __forceinline__ float word_to_trunc_normal(uint32_t x) {
    const uint32_t v = x >> 9;                              // 23 random bits
    const uint32_t i = v >> (23 - TNQ_BITS);
    const float f = ((float)(v & ((1u << (23 - TNQ_BITS)) - 1u)) + 0.5f) * (1.0f / (float)(1u << (23 - TNQ_BITS)));
    const float2 e = g_tnq[i];
    return fmaf(f, e.y, e.x);                               // strictly inside (-2, 2): q[0] = -2, q[2048] = 2, 0 < f < 1
}

__device__ __forceinline__ float word_to_rademacher(uint32_t x) {
    return (x & 0x80000000u) ? 1.0f : -1.0f;
}

// Box-Muller: two words -> two N(0,1)
__device__ __forceinline__ void words_to_normal2(uint32_t x0, uint32_t x1, float& z0, float& z1) {
    float u1 = word_to_uniform(x0);
    float u2 = word_to_uniform(x1);
    float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

struct RngKey {
    uint32_t k0, k1;      // seed
    uint32_t step;        // control step
    uint32_t q_per_agent; // Q = ceil(H*U/4)
    // a control step replayed as a hipGraph (engine: step_graph) has its launch arguments frozen: the kernels on that path
    // take the control step from this word (device memory, advanced by the step's last kernel) instead
    const uint32_t* step_src;
};
__device__ __forceinline__ RngKey rng_key_now(const RngKey& k) {
    RngKey r = k;
    if (k.step_src) r.step = *k.step_src;
    return r;
}

// Raw words for the 4-element block containing element j of particle n, global agent ga.
__device__ __forceinline__ U4 rng_block(const RngKey& key, uint32_t stream, uint32_t iter, uint32_t n,
                                        uint32_t ga, uint32_t j) {
    U4 c;
    c.x = n;
    c.y = ga * key.q_per_agent + (j >> 2);
    c.z = key.step;
    c.w = (stream << 16) | iter;
    return philox4x32_10(c, key.k0, key.k1);
}

__device__ __forceinline__ uint32_t pick_word(const U4& r, uint32_t j) {
    uint32_t l = j & 3u;
    return l == 0 ? r.x : (l == 1 ? r.y : (l == 2 ? r.z : r.w));
}

}  // namespace bbmpc
