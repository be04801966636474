/* C restatement of the sampling-MPC hot path -- TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see oracle_np.py).
 *
 * Second, independent CPU restatement of the reference arithmetic (ossamaAhmed/blackbox_mpc v0.3), written from the
 * same reference lines as oracle/oracle_np.py and checked against it in tests/test_oracle_c.py.  It exists for two
 * reasons: (1) two independently written restatements that agree bit for bit are a stronger checker than one;
 * (2) bench.py's `cpu_baseline` leg needs a CPU path that is not dominated by Python interpreter overhead -- this
 * one is plain C with OpenMP over candidate trajectories (the reference's TF-CPU executor parallelises the same axis).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it; the product never does.
 *
 * Conventions (identical to oracle_np.py): every tensor fp32, one rounding per reference op, no FMA contraction
 * (build with -ffp-contract=off); transcendentals evaluated in fp64 and rounded once; Dense layers accumulate in
 * fp64 and round once; layouts are the reference's (samples [N,A,H,U], rewards [N,A], states [A,S]).
 * All path:line citations are relative to /root/reference/blackbox_mpc/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BBO_MAX_LAYERS 8
enum { BBO_DYN_PENDULUM = 1, BBO_DYN_MLP = 2 };
enum { BBO_REW_PENDULUM = 1, BBO_REW_CHEETAH = 2 };
enum { BBO_ACT_NONE = 0, BBO_ACT_TANH = 1, BBO_ACT_RELU = 2, BBO_ACT_SIGMOID = 3 };
enum { BBO_OPT_RS = 1, BBO_OPT_CEM = 2, BBO_OPT_PI2 = 3 };

typedef struct {
    int32_t dyn, rew, as_executed;            /* as_executed: quirk Q1 (deterministic.py:65-66) */
    int32_t N, A, H, U, S, iters, k;
    float alpha, lamda;
    const float* lo;                          /* [U] */
    const float* hi;                          /* [U] */
    /* learned dynamics (deterministic_mlp.py:27-51 + system_dynamics_handler.py:97-161) */
    int32_t n_layers, normalized;
    int32_t dims[BBO_MAX_LAYERS + 1];
    int32_t acts[BBO_MAX_LAYERS];
    const float* w[BBO_MAX_LAYERS];           /* Keras Dense kernels [in][out] */
    const float* b[BBO_MAX_LAYERS];
    const float *mean_s, *std_s, *mean_a, *std_a, *mean_t, *std_t;
} bbo_problem;

static const float PI32 = 3.14159274101257324f;       /* float32(pi) */
static const float TWO_PI32 = 6.28318548202514648f;   /* float32(2*pi) */

static inline float sin32(float x) { return (float)sin((double)x); }
static inline float cos32(float x) { return (float)cos((double)x); }
static inline float atan2_32(float y, float x) { return (float)atan2((double)y, (double)x); }
static inline float exp32(float x) { return (float)exp((double)x); }
static inline float tanh32(float x) { return (float)tanh((double)x); }
static inline float clip32(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }   /* NaN passes through */

/* TF FloorMod on floats (utils/pendulum.py:7): fmod, then shift into the divisor's sign */
static inline float floormod32(float x, float y) {
    float r = fmodf(x, y);
    if (r != 0.0f && ((y < 0.0f) != (r < 0.0f))) r = r + y;
    return r;
}

/* PendulumTrueModel.__call__ utils/pendulum.py:58-92 -> delta; true-model handler adds the state back
 * (system_dynamics_handler.py:149-151, transforms.py:34).  Quirk Q9: theta integrates the unclipped speed. */
static void pendulum_next(const float* s, float u, float* nxt) {
    const float theta = atan2_32(s[1], s[0]);                         /* :82 */
    float acc = -15.0f * sin32(theta + PI32);                         /* :83  -3g/(2l) = -15 */
    acc = acc + 3.0f * u;                                             /* :83-84  3/(m l^2) = 3 */
    float newthdot = s[2] + acc * 0.05f;                              /* :85 */
    const float newth = theta + newthdot * 0.05f;                     /* :86 */
    newthdot = clip32(newthdot, -8.0f, 8.0f);                         /* :87 */
    const float ns[3] = {cos32(newth), sin32(newth), newthdot};       /* :88-90 */
    for (int i = 0; i < 3; ++i) {
        const float delta = ns[i] - s[i];                             /* :91 */
        nxt[i] = delta + s[i];
    }
}

/* pendulum_reward_function utils/pendulum.py:10-35, called positionally as (cur, actions, next) => quirk Q1 */
static float pendulum_reward(const bbo_problem* P, const float* cur, const float* act, const float* nxt) {
    const float th = atan2_32(cur[1], cur[0]);
    const float ang = floormod32(th + PI32, TWO_PI32) - PI32;         /* :5-7 */
    const float a2 = ang * ang;
    const float v2 = cur[2] * cur[2];
    const float first = a2 + 0.1f * v2;
    float ssum = 0.0f;
    if (P->as_executed) for (int i = 0; i < P->S; ++i) ssum = ssum + nxt[i] * nxt[i];
    else for (int i = 0; i < P->U; ++i) ssum = ssum + act[i] * act[i];
    return (-first) - 0.001f * ssum;
}

/* reward_function tutorials/mujoco/cost_func.py:5-22 (HalfCheetahEnvModified, S=20) */
static float cheetah_reward(const bbo_problem* P, const float* cur, const float* act, const float* nxt) {
    float r = 0.0f;
    if (cur[5] >= 0.2f) r = r + (-10.0f);
    if (cur[6] >= 0.0f) r = r + (-10.0f);
    if (cur[7] >= 0.0f) r = r + (-10.0f);
    r = r + (nxt[17] - cur[17]) / 0.01f;
    float ss = 0.0f;
    for (int i = 0; i < P->U; ++i) ss = ss + act[i] * act[i];
    r = r - 0.0f * ss;
    return r;
}

static float reward_of(const bbo_problem* P, const float* cur, const float* act, const float* nxt) {
    return P->rew == BBO_REW_PENDULUM ? pendulum_reward(P, cur, act, nxt) : cheetah_reward(P, cur, act, nxt);
}

/* learned dynamics: process_input -> Dense stack -> process_output for one row */
static void mlp_next(const bbo_problem* P, const float* s, const float* a, float* nxt) {
    float x[512], y[512];
    const int S = P->S, U = P->U;
    for (int i = 0; i < S; ++i) x[i] = P->normalized ? (s[i] - P->mean_s[i]) / (P->std_s[i] + 1e-7f) : s[i];
    for (int i = 0; i < U; ++i) x[S + i] = P->normalized ? (a[i] - P->mean_a[i]) / (P->std_a[i] + 1e-7f) : a[i];
    for (int l = 0; l < P->n_layers; ++l) {
        const int K = P->dims[l], M = P->dims[l + 1];
        const float* W = P->w[l];
        for (int o = 0; o < M; ++o) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += (double)x[k] * (double)W[(size_t)k * M + o];
            float v = (float)acc;
            v = v + P->b[l][o];
            switch (P->acts[l]) {
                case BBO_ACT_TANH: v = tanh32(v); break;
                case BBO_ACT_RELU: v = v > 0.0f ? v : 0.0f; break;
                case BBO_ACT_SIGMOID: v = 1.0f / (1.0f + exp32(-v)); break;
                default: break;
            }
            y[o] = v;
        }
        memcpy(x, y, sizeof(float) * (size_t)M);
    }
    for (int i = 0; i < S; ++i) {
        const float dev = P->normalized ? P->mean_t[i] + x[i] * (P->std_t[i] + 1e-7f) : x[i];
        nxt[i] = dev + s[i];
    }
}

static void next_state(const bbo_problem* P, const float* s, const float* a, float* nxt) {
    if (P->dyn == BBO_DYN_PENDULUM) pendulum_next(s, a[0], nxt);
    else mlp_next(P, s, a, nxt);
}

/* DeterministicTrajectoryEvaluator.__call__ trajectory_evaluators/deterministic.py:26-77.
 * states [A,S], seq [N,A,H,U] -> rewards [N,A].  Rows are independent: OpenMP over them. */
int bbo_evaluate(const bbo_problem* P, const float* states, const float* seq, float* rewards) {
    const int N = P->N, A = P->A, H = P->H, U = P->U, S = P->S;
    if (S > 64 || U > 64) return -1;
#pragma omp parallel for schedule(static)
    for (int row = 0; row < N * A; ++row) {
        const int a = row % A;
        float cur[64], nxt[64];
        memcpy(cur, states + (size_t)a * S, sizeof(float) * (size_t)S);
        float total = 0.0f;
        for (int t = 0; t < H; ++t) {                                   /* :62-73 */
            const float* act = seq + ((size_t)row * H + t) * U;
            next_state(P, cur, act, nxt);
            total = total + reward_of(P, cur, act, nxt);
            memcpy(cur, nxt, sizeof(float) * (size_t)S);
        }
        if (total != total) total = -1.0e6f;                            /* :75-77 */
        rewards[row] = total;
    }
    return 0;
}

int bbo_predict_next_state(const bbo_problem* P, int rows, const float* s, const float* a, float* nxt) {
    for (int r = 0; r < rows; ++r) next_state(P, s + (size_t)r * P->S, a + (size_t)r * P->U, nxt + (size_t)r * P->S);
    return 0;
}

int bbo_evaluate_next_reward(const bbo_problem* P, int rows, const float* cur, const float* nxt, const float* act, float* out) {
    for (int r = 0; r < rows; ++r)
        out[r] = reward_of(P, cur + (size_t)r * P->S, act + (size_t)r * P->U, nxt + (size_t)r * P->S);
    return 0;
}

/* ---- standard noise for the timing leg (the reference draws inside its graph too).  xoshiro128++ per row,
 * truncated normal by rejection -- what tf.random.truncated_normal does. */
typedef struct { uint32_t s[4]; } rng_t;
static inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
static inline uint32_t rng_next(rng_t* r) {
    const uint32_t res = rotl(r->s[0] + r->s[3], 7) + r->s[0];
    const uint32_t t = r->s[1] << 9;
    r->s[2] ^= r->s[0]; r->s[3] ^= r->s[1]; r->s[1] ^= r->s[2]; r->s[0] ^= r->s[3];
    r->s[2] ^= t; r->s[3] = rotl(r->s[3], 11);
    return res;
}
static void rng_seed(rng_t* r, uint64_t seed) {
    for (int i = 0; i < 4; ++i) {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        r->s[i] = (uint32_t)((z ^ (z >> 31)) >> 16) | 1u;
    }
}
static inline float rng_uniform(rng_t* r) { return ((float)(rng_next(r) >> 9) + 0.5f) * 1.1920928955078125e-07f; }
static float rng_trunc_normal(rng_t* r) {
    for (;;) {
        const float u1 = rng_uniform(r), u2 = rng_uniform(r);
        const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
        if (fabsf(z) < 2.0f) return z;
    }
}
/* fills [N,A,H,U] with kind 0 = U[0,1), 1 = unit normal truncated to |z|<2 */
void bbo_fill_noise(int kind, uint64_t seed, int rows, int per_row, float* out) {
#pragma omp parallel for schedule(static)
    for (int row = 0; row < rows; ++row) {
        rng_t r;
        rng_seed(&r, seed * 0x100000001B3ull + (uint64_t)row);
        for (int j = 0; j < per_row; ++j) out[(size_t)row * per_row + j] = kind ? rng_trunc_normal(&r) : rng_uniform(&r);
    }
}

/* tf.nn.top_k(sorted=True) per agent: larger first, ties -> lower index first (cem.py:97-99) */
typedef struct { float v; int32_t i; } kv_t;
static int kv_cmp(const void* pa, const void* pb) {
    const kv_t* a = (const kv_t*)pa; const kv_t* b = (const kv_t*)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->i > b->i) - (a->i < b->i);
}

/* One optimizer call  OptimizerBase.__call__ optimizers/optimizer_base.py:55-95 (exploration noise off):
 *   RandomSearch random_search.py:38-48 | CEM cem.py:74-136 | PI2 pi2.py:58-96
 * state [A,S]; noise: [iters][N,A,H,U] standard draws or NULL (drawn here from `seed`);
 * prev_mean [A,H,U] in/out (CEM: read only, quirk Q2; PI2: shifted warm start :92-93);
 * outputs action [A,U], next_state [A,S], reward [A]; optional traces elites_out [iters][A,k],
 * mean_out/var_out [A,H,U] after the last iteration; forced_elites [iters][A,k] or NULL (lock-step parity runs). */
int bbo_optimize(const bbo_problem* P, int opt, const float* state, const float* noise, uint64_t seed,
                 float* prev_mean, float* action, float* nxt_out, float* rew_out,
                 int32_t* elites_out, float* mean_out, float* var_out, const int32_t* forced_elites) {
    const int N = P->N, A = P->A, H = P->H, U = P->U, HU = H * U, k = P->k;
    const int iters = opt == BBO_OPT_RS ? 1 : P->iters;
    const size_t cnt = (size_t)N * A * HU;
    float* samples = (float*)malloc(sizeof(float) * cnt);
    float* xi_own = noise ? NULL : (float*)malloc(sizeof(float) * cnt);
    float* rewards = (float*)malloc(sizeof(float) * (size_t)N * A);
    float* pen = (float*)malloc(sizeof(float) * (size_t)N * A);
    float* mean = (float*)malloc(sizeof(float) * (size_t)A * HU);
    float* var = (float*)malloc(sizeof(float) * (size_t)A * HU);
    float* sig = (float*)malloc(sizeof(float) * (size_t)A * HU);
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)N);
    float* prob = (float*)malloc(sizeof(float) * (size_t)N);
    int rc = 0;
    for (int a = 0; a < A; ++a)
        for (int j = 0; j < HU; ++j) {
            const int u = j % U;
            const float d = P->lo[u] - P->hi[u];
            mean[a * HU + j] = prev_mean[a * HU + j];
            var[a * HU + j] = (d * d) / 16.0f;                          /* cem.py:62-66 / pi2.py:53-55 */
        }
    for (int it = 0; it < iters && rc == 0; ++it) {
        const float* xi = noise ? noise + (size_t)it * cnt : xi_own;
        if (!noise) bbo_fill_noise(opt == BBO_OPT_RS ? 0 : 1, seed * 131u + (uint64_t)it, N * A, HU, xi_own);
        /* ---- sampling */
        for (int a = 0; a < A; ++a)
            for (int j = 0; j < HU; ++j) {
                const int u = j % U;
                const float m = mean[a * HU + j], v = var[a * HU + j];
                if (opt == BBO_OPT_CEM) {                               /* cem.py:79-88 */
                    const float lb = m - P->lo[u], ub = P->hi[u] - m;
                    const float l2 = (lb / 2.0f) * (lb / 2.0f), u2 = (ub / 2.0f) * (ub / 2.0f);
                    const float mn = l2 < u2 ? l2 : u2;
                    sig[a * HU + j] = sqrtf(mn < v ? mn : v);
                } else sig[a * HU + j] = sqrtf(v);
            }
#pragma omp parallel for schedule(static)
        for (int row = 0; row < N * A; ++row) {
            const int a = row % A;
            float ss = 0.0f;
            for (int j = 0; j < HU; ++j) {
                const int u = j % U;
                const size_t e = (size_t)row * HU + j;
                float x;
                if (opt == BBO_OPT_RS) x = xi[e] * (P->hi[u] - P->lo[u]) + P->lo[u];       /* random_search.py:40-41 */
                else x = xi[e] * sig[a * HU + j] + mean[a * HU + j];                       /* cem.py:90-94 / pi2.py:65-69 */
                if (opt == BBO_OPT_PI2) {                                                  /* pi2.py:70-75 */
                    const float xf = clip32(x, P->lo[u], P->hi[u]);
                    const float d = x - xf;
                    ss = ss + d * d;
                    x = xf;
                }
                samples[e] = x;
            }
            const float nr = sqrtf(ss);
            pen[row] = nr * nr;
        }
        rc = bbo_evaluate(P, state, samples, rewards);
        if (rc) break;
        /* ---- refit, per agent */
        for (int a = 0; a < A; ++a) {
            if (opt == BBO_OPT_RS) {                                    /* random_search.py:43-47: first maximum wins */
                int best = 0;
                for (int n = 1; n < N; ++n) if (rewards[n * A + a] > rewards[best * A + a]) best = n;
                for (int j = 0; j < HU; ++j) mean[a * HU + j] = samples[((size_t)best * A + a) * HU + j];
            } else if (opt == BBO_OPT_CEM) {
                for (int n = 0; n < N; ++n) { kv[n].v = rewards[n * A + a]; kv[n].i = n; }
                qsort(kv, (size_t)N, sizeof(kv_t), kv_cmp);
                if (forced_elites) for (int e = 0; e < k; ++e) kv[e].i = forced_elites[((size_t)it * A + a) * k + e];
                if (elites_out) for (int e = 0; e < k; ++e) elites_out[((size_t)it * A + a) * k + e] = kv[e].i;
                const float one_m = 1.0f - P->alpha;
                for (int j = 0; j < HU; ++j) {                          /* cem.py:112-125 */
                    float s = 0.0f;
                    for (int e = 0; e < k; ++e) s = s + samples[((size_t)kv[e].i * A + a) * HU + j];
                    const float em = s / (float)k;
                    float vs = 0.0f;
                    for (int e = 0; e < k; ++e) {
                        const float d = samples[((size_t)kv[e].i * A + a) * HU + j] - em;
                        vs = vs + d * d;
                    }
                    const float ev = vs / (float)k;
                    mean[a * HU + j] = P->alpha * mean[a * HU + j] + one_m * em;
                    var[a * HU + j] = P->alpha * var[a * HU + j] + one_m * ev;
                }
            } else {                                                    /* pi2.py:77-87 */
                float beta = INFINITY;
                for (int n = 0; n < N; ++n) {
                    const float c = -(rewards[n * A + a] - pen[n * A + a]);
                    prob[n] = c;
                    if (c < beta) beta = c;
                }
                const float inv = 1.0f / P->lamda;
                float eta = 0.0f;
                for (int n = 0; n < N; ++n) { prob[n] = exp32((-inv) * (prob[n] - beta)); eta = eta + prob[n]; }
                const float ieta = 1.0f / eta;
                for (int j = 0; j < HU; ++j) {
                    float s = 0.0f;
                    for (int n = 0; n < N; ++n) s = s + samples[((size_t)n * A + a) * HU + j] * (ieta * prob[n]);
                    mean[a * HU + j] = s;
                }
            }
        }
    }
    if (rc == 0) {
        for (int a = 0; a < A; ++a)
            for (int u = 0; u < U; ++u) action[a * U + u] = mean[a * HU + u];   /* cem.py:135 / pi2.py:94 / random_search.py:44-47 */
        if (opt == BBO_OPT_PI2)                                                  /* pi2.py:92-93 shift-left warm start */
            for (int a = 0; a < A; ++a)
                for (int j = 0; j < HU; ++j)
                    prev_mean[a * HU + j] = mean[a * HU + (j + U < HU ? j + U : HU - U + j % U)];
        if (mean_out) memcpy(mean_out, mean, sizeof(float) * (size_t)A * HU);
        if (var_out) memcpy(var_out, var, sizeof(float) * (size_t)A * HU);
        bbo_predict_next_state(P, A, state, action, nxt_out);                    /* optimizer_base.py:91-94 */
        bbo_evaluate_next_reward(P, A, state, nxt_out, action, rew_out);
    }
    free(samples); free(xi_own); free(rewards); free(pen); free(mean); free(var); free(sig); free(kv); free(prob);
    return rc;
}

int bbo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void bbo_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
