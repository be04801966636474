// Stand-alone timing of k_rollout_mlp_pair at the config-5 shape (N = 2000, A = 4, H = 50, 26-200-200-20) with random
// operands, plus per-wave clocks at the interval boundaries of one model step of workgroup 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iblackbox_mpc_amd/csrc -Iinclude
//         tools/microbench/pair_probe.hip -o /tmp/pair_probe && /tmp/pair_probe
#include <hip/hip_runtime.h>
__device__ long long g_pair_clk[16 * 24];
#define BBMPC_PAIR_CLK(slot) \
    do { if (blockIdx.x == 1 && blockIdx.y == 0 && t == 20 && lane == 0) g_pair_clk[wid * 24 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define BBMPC_PAIR_CLK2(slot, seq) \
    do { if (blockIdx.x == 1 && blockIdx.y == 0 && (seq) == 21 && lane == 0) g_pair_clk[wid * 24 + (slot)] = __builtin_readcyclecounter(); } while (0)
#include "engine.hpp"
using namespace bbmpc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static float* dev_rand(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.0f - 1.0f); }
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main() {
    const int N = 2000, A = 4, H = 50, S = 20, U = 6, HT = 13, Nst = 2048;
    MlpRolloutArgs q; memset(&q, 0, sizeof(q));
    RolloutArgs& r = q.r;
    r.n_pop = N; r.A = A; r.H = H; r.U = U; r.S = S; r.HU = H * U; r.Nst = Nst; r.reward_kind = REW_CHEETAH;
    r.state = dev_rand(A * S, 0.5f, 1); r.mean = dev_rand(A * H * U, 0.3f, 2); r.sigma = dev_rand(A * H * U, 0.2f, 3);
    r.lo = dev_rand(U, 0.0f, 4); r.hi = dev_rand(U, 0.0f, 5);
    r.samples = dev_rand((size_t)A * H * U * Nst, 0.0f, 6); r.rewards = dev_rand((size_t)A * Nst, 0.0f, 7);
    r.key = RngKey{}; r.stream = 1; r.iter = 0;
    MlpDesc& m = q.m;
    m.n_layers = 3; m.dims[0] = 26; m.dims[1] = 200; m.dims[2] = 200; m.dims[3] = 20;
    m.tiles[0] = 2; m.tiles[1] = 13; m.tiles[2] = 13; m.tiles[3] = 2;
    m.act[0] = ACT_TANH; m.act[1] = ACT_TANH; m.act[2] = ACT_NONE;
    m.wpack[0] = dev_rand(13 * 2 * 256, 0.2f, 8); m.wpack[1] = dev_rand(13 * 13 * 256, 0.07f, 9); m.wpack[2] = dev_rand(2 * 13 * 256, 0.07f, 10);
    m.bpack[0] = dev_rand(13 * 256, 0.1f, 11); m.bpack[1] = dev_rand(13 * 256, 0.1f, 12); m.bpack[2] = dev_rand(2 * 256, 0.1f, 13);
    m.half_tail[1] = 1; m.half_tail[2] = 1; m.normalized = 1;
    m.mean_s = dev_rand(S, 0.1f, 14); m.std_s = dev_rand(S, 0.0f, 15); m.mean_a = dev_rand(U, 0.1f, 16); m.std_a = dev_rand(U, 0.0f, 17);
    m.mean_t = dev_rand(S, 0.01f, 18); m.std_t = dev_rand(S, 0.0f, 19);
    // stds = 1
    { std::vector<float> one(32, 1.0f); hipMemcpy((void*)m.std_s, one.data(), S * 4, hipMemcpyHostToDevice); hipMemcpy((void*)m.std_a, one.data(), U * 4, hipMemcpyHostToDevice);
      std::vector<float> sm(32, 0.05f); hipMemcpy((void*)m.std_t, sm.data(), S * 4, hipMemcpyHostToDevice);
      std::vector<float> lo(8, -1.0f), hi(8, 1.0f); hipMemcpy((void*)r.lo, lo.data(), U * 4, hipMemcpyHostToDevice); hipMemcpy((void*)r.hi, hi.data(), U * 4, hipMemcpyHostToDevice); }
    q.mode = SRC_TRUNC; q.pen = 0; q.nw = 13;
    auto fn = k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2, 20, 6, 50, REW_CHEETAH, 1>;
    const size_t lds = (size_t)mlp_pair_lds_floats(HT, H, U, S, 2) * sizeof(float);
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    dim3 grid((N + 31) / 32, A), block(mlp_pair_waves(HT, 2) * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, grid, block, lds, 0, q);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, grid, block, lds, 0, q);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> rew(A * Nst); hipMemcpy(rew.data(), r.rewards, rew.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (int a = 0; a < A; ++a) for (int n = 0; n < N; ++n) cs += rew[a * Nst + n];
    printf("k_rollout_mlp_pair: %.1f us per launch (%d waves, lds %zu B), reward checksum %.6e\n", ms * 1000.0f / reps, (int)block.x / 64, lds, cs);
    long long clk[16 * 24]; hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_pair_clk), sizeof(clk));
    printf("wave: work1 wait1 | work2 wait2 | work3 wait3   (cycles of the shader clock, step 20 of workgroup (1,0))\n");
    for (int w = 0; w < (int)block.x / 64; ++w) {
        long long* c = clk + w * 24;
        printf("%2d: %6lld %6lld | %6lld %6lld | %6lld %6lld   total %lld\n", w, c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], c[5] - c[4], c[6] - c[5], c[6] - c[0]);
    }
    printf("inside the layer-1 stage (cycles since the interval's barrier release): loop entry | behind k tile 3 | 7 | 11 | loop exit | slab stored\n");
    for (int half = 0; half < 2; ++half)
        for (int w = 0; w < (int)block.x / 64; ++w) {
            long long* c = clk + w * 24;
            const long long t0 = c[half ? 4 : 2];
            long long* d = c + 8 + 8 * half;
            printf("I%d wave %2d: %6lld %6lld %6lld %6lld %6lld %6lld\n", half + 2, w, d[0] - t0, d[1] - t0, d[2] - t0, d[3] - t0, d[4] - t0, d[5] - t0);
        }
    return 0;
}
