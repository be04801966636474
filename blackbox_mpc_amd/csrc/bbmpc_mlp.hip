// Learned-dynamics half of the engine (DeterministicMLP, dynamics_functions/deterministic_mlp.py:5-51 behind
// SystemDynamicsHandler.process_input / process_output): weight packing and the choice + launch of the MFMA rollout
// kernels (kernels_mlp*.hpp).  A translation unit of its own: the template kernels are instantiated here only.
#define BBMPC_TU_MLP
#include "engine.hpp"
#include "engine_util.hpp"

namespace bbmpc {

void bbmpc_tu_mlp_upload_tnq(const float2* table) { tnq_upload(table); }

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
void Engine::set_mlp(int n_layers, const int32_t* dims, const int32_t* acts, const float* const* w, const float* const* b,
                     int is_normalized, const float* const* stats) {
    REQUIRE(cfg.dynamics == BBMPC_DYN_MLP, BBMPC_E_STATE, "handle was not created with BBMPC_DYN_MLP");
    REQUIRE(n_layers >= 1 && n_layers <= MLP_MAX_LAYERS, BBMPC_E_UNSUPPORTED, "1..8 Dense layers are supported");
    REQUIRE(dims && acts && w && b, BBMPC_E_INVALID, "null argument");
    REQUIRE(dims[0] == S + U && dims[n_layers] == S, BBMPC_E_INVALID, "MLP must map dim_S+dim_U -> dim_S");
    HIP_CHECK(hipStreamSynchronize(stream));
    memset(&mlp, 0, sizeof(mlp));
    mlp.n_layers = n_layers;
    int hidden_tiles = 1;
    for (int l = 0; l <= n_layers; ++l) {
        REQUIRE(dims[l] >= 1, BBMPC_E_INVALID, "layer width must be >= 1");
        mlp.dims[l] = dims[l];
        mlp.tiles[l] = (dims[l] + 15) / 16;
        if (l >= 1 && l < n_layers) hidden_tiles = std::max(hidden_tiles, mlp.tiles[l]);
    }
    REQUIRE(hidden_tiles <= 16 * MLP_TMAX, BBMPC_E_UNSUPPORTED, "hidden width > 512 not supported");
    REQUIRE(mlp.tiles[0] <= 8 && mlp.tiles[n_layers] <= 4, BBMPC_E_UNSUPPORTED, "dim_S+dim_U <= 128 and dim_S <= 64 supported");
    mlp_nw = std::min(16, hidden_tiles);
    for (int l = 1; l < n_layers; ++l) {
        const int tail = dims[l] - 16 * (mlp.tiles[l] - 1);
        mlp.half_tail[l] = (tail <= 8 && !sw.mlp_no_half_tail) ? 1 : 0;
    }
    for (int l = 0; l < n_layers; ++l) {
        REQUIRE(acts[l] >= BBMPC_ACT_NONE && acts[l] <= BBMPC_ACT_SIGMOID, BBMPC_E_INVALID, "unknown activation");
        REQUIRE(w[l] && b[l], BBMPC_E_INVALID, "null weight/bias pointer");
        mlp.act[l] = acts[l];
        const int K = dims[l], M = dims[l + 1], IT = mlp.tiles[l], OT = mlp.tiles[l + 1];
        std::vector<float> wp((size_t)OT * IT * 256, 0.0f), bp((size_t)OT * 256, 0.0f);
        // Hidden features are internal, so their order inside a tile is free.  When the last tile of a hidden layer
        // holds <= 8 features (200 units = 12 tiles + 8) they are put into the slots 4g + {0, 1}: as the next layer's
        // K tile that leaves MFMAs 2 and 3 (k = 4g + 2, 4g + 3) with nothing but zeros, and the pipelined kernel
        // skips them.  slot -> feature (or -1 = padding); inputs (layer 0) and outputs (last layer) keep their order.
        auto feature_of_slot = [&](int layer_of_feature, int slot) -> int {
            const int width = dims[layer_of_feature], tiles = (width + 15) / 16, t = slot >> 4, q = slot & 15;
            if (layer_of_feature >= 1 && layer_of_feature < n_layers && t == tiles - 1 && mlp.half_tail[layer_of_feature]) {
                if ((q & 3) >= 2) return -1;
                const int f = 16 * t + 2 * (q >> 2) + (q & 3);
                return f < width ? f : -1;
            }
            return slot < width ? slot : -1;
        };
        for (int ot = 0; ot < OT; ++ot) {
            for (int it = 0; it < IT; ++it)
                for (int s = 0; s < 4; ++s)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int k = feature_of_slot(l, it * 16 + 4 * (ln >> 4) + s), o = feature_of_slot(l + 1, ot * 16 + (ln & 15));
                        if (k >= 0 && o >= 0) wp[(((size_t)ot * IT + it) * 4 + s) * 64 + ln] = w[l][(size_t)k * M + o];
                    }
            for (int ln = 0; ln < 64; ++ln)
                for (int r = 0; r < 4; ++r) {
                    const int o = feature_of_slot(l + 1, ot * 16 + (ln >> 4) * 4 + r);
                    if (o >= 0) bp[((size_t)ot * 64 + ln) * 4 + r] = b[l][o];
                }
        }
        upload(d_wpack[l], wp);
        {   // the same operands with a lane's four A operands of a k tile side by side: [OT][IT][lane][s] (generic kernel)
            std::vector<float> w4(wp.size());
            for (size_t t = 0; t < (size_t)OT * IT; ++t)
                for (int s = 0; s < 4; ++s)
                    for (int ln = 0; ln < 64; ++ln) w4[(t * 64 + ln) * 4 + s] = wp[(t * 4 + s) * 64 + ln];
            upload(d_wpack4[l], w4);
        }
        upload(d_bpack[l], bp);
        upload(d_wraw[l], std::vector<float>(w[l], w[l] + (size_t)K * M));
        upload(d_braw[l], std::vector<float>(b[l], b[l] + M));
        {   // quad-mode operand order [k/4][Mp][4]; the k/4 axis is zero padded to a multiple of 64 groups so that the
            // kernels can load a fixed number of groups per lane without bounds (group 63 of a <= 252-input layer is a zero row)
            const int Mp = (M + 63) & ~63, KG = ((K + 3) / 4 + 63) & ~63;
            std::vector<float> wq((size_t)KG * Mp * 4, 0.0f);
            for (int kk = 0; kk < K; ++kk)
                for (int o = 0; o < M; ++o) wq[((size_t)(kk >> 2) * Mp + o) * 4 + (kk & 3)] = w[l][(size_t)kk * M + o];
            upload(d_wq4[l], wq);
            if (l == 0 && S <= 20 && U <= 8) {
                // k_rollout_mlp_q4s keeps five state groups and two action groups whatever dim_S is: the state rows padded to 20
                std::vector<float> ws((size_t)KG * Mp * 4, 0.0f);
                for (int kk = 0; kk < K; ++kk) {
                    const int kp = kk < S ? kk : 20 + (kk - S);
                    for (int o = 0; o < M; ++o) ws[((size_t)(kp >> 2) * Mp + o) * 4 + (kp & 3)] = w[l][(size_t)kk * M + o];
                }
                upload(d_wq4s0, ws);
            }
        }
        {   // optional bf16 mode operands: [OT][IT][64] x (4 bf16 hi | 4 bf16 lo), k = 16*it + 4*(lane>>4) + r, row = 16*ot + (lane&15)
            auto bf16_rne = [](float x) -> uint16_t {
                uint32_t u; memcpy(&u, &x, 4);
                u += 0x7fffu + ((u >> 16) & 1u);
                return (uint16_t)(u >> 16);
            };
            auto bf16_f = [](uint16_t h) -> float { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
            const int IT = (K + 15) / 16, OT = (M + 15) / 16;
            std::vector<float> wb((size_t)OT * IT * 64 * 4, 0.0f);          // 16 bytes per lane, kept in a float buffer
            uint16_t* hw = reinterpret_cast<uint16_t*>(wb.data());
            for (int ot = 0; ot < OT; ++ot)
                for (int it = 0; it < IT; ++it)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int r = 0; r < 4; ++r) {
                            // same slot -> feature map as the fp32 operands (the biases come from bpack)
                            const int kk = feature_of_slot(l, 16 * it + 4 * (ln >> 4) + r), o = feature_of_slot(l + 1, 16 * ot + (ln & 15));
                            const float v = (kk >= 0 && o >= 0) ? w[l][(size_t)kk * M + o] : 0.0f;
                            const uint16_t h = bf16_rne(v), lo = bf16_rne(v - bf16_f(h));
                            uint16_t* d = hw + (((size_t)ot * IT + it) * 64 + ln) * 8;
                            d[r] = h;
                            d[4 + r] = lo;
                        }
            upload(d_wbf[l], wb);
        }
        mlp.wpack[l] = d_wpack[l].p;
        mlp.bpack[l] = d_bpack[l].p;
    }
    {   // k_rollout_mlp_w4 (2..4 Dense layers of at most 32 units, dim_S <= 20, dim_U <= 8): operands in lane order
        bool small = n_layers >= 2 && n_layers <= 4 && S <= 20 && U <= 8;
        for (int l = 1; l < n_layers; ++l) small = small && dims[l] <= 32;
        d_w4pack.release();
        if (small) {
            std::vector<float> wp((size_t)n_layers * W4_OPS * 64, 0.0f);
            for (int l = 0; l < n_layers; ++l)
                for (int op = 0; op < W4_OPS; ++op)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int idx = mlp_w4_operand_index(l, op, ln, dims[l], dims[l + 1], S, U);
                        if (idx >= 0) wp[((size_t)l * W4_OPS + op) * 64 + ln] = op < 16 ? w[l][idx] : b[l][idx];
                    }
            upload(d_w4pack, wp);
        }
    }
    mlp.normalized = is_normalized ? 1 : 0;
    if (is_normalized) {
        REQUIRE(stats, BBMPC_E_INVALID, "normalisation statistics are required when is_normalized != 0");
        std::vector<float> st;
        const int lens[6] = {S, S, U, U, S, S};
        for (int i = 0; i < 6; ++i) {
            REQUIRE(stats[i], BBMPC_E_INVALID, "null statistics vector");
            st.insert(st.end(), stats[i], stats[i] + lens[i]);
        }
        upload(d_stats, st);
        float* q = d_stats.p;
        mlp.mean_s = q; q += S;
        mlp.std_s = q; q += S;
        mlp.mean_a = q; q += U;
        mlp.std_a = q; q += U;
        mlp.mean_t = q; q += S;
        mlp.std_t = q;
    }
    mlp_ready = true;
}

void Engine::launch_rollout_mlp(int mode, bool pen, RolloutArgs& ra, bool per_particle_state, float* final_state) {
    REQUIRE(mlp_ready, BBMPC_E_STATE, "learned dynamics: call bbmpc_set_mlp before computing");
    MlpRolloutArgs q;
    memset(&q, 0, sizeof(q));
    q.r = ra;
    q.m = mlp;
    q.mode = mode;
    q.pen = pen ? 1 : 0;
    q.per_particle_state = per_particle_state ? 1 : 0;
    q.final_state = final_state;
    q.nw = mlp_nw;
    q.traj = mlp_traj_out;
    const bool record = mlp_traj_out != nullptr;       // trajectory recording lives in rollout_mlp_body's epilogue only
    for (int l = 0; l < mlp.n_layers; ++l) { q.wraw[l] = d_wraw[l].p; q.braw[l] = d_braw[l].p; q.wq4[l] = d_wq4[l].p; q.wp4[l] = d_wpack4[l].p; q.wbf[l] = reinterpret_cast<const uint4*>(d_wbf[l].p); }
    q.wq4s0 = d_wq4s0.p;
    q.w4pack = d_w4pack.p;
    const MlpLds lay = mlp_lds_layout(mlp, ra.H, U, S, mlp_nw);
    const size_t lds = (size_t)lay.total * sizeof(float);
    REQUIRE(lds <= 159 * 1024, BBMPC_E_UNSUPPORTED,
            "learned-model rollout: a 16-particle tile's action block (planning_horizon x 16 x dim_u floats) plus the activation / "
            "partial-sum buffers of this network do not fit one CU's LDS (shorten the horizon or narrow the network)");
    // weights-stationary specialisations (kernels_mlp.hpp)
    int spec = 0;
    const bool small_io = mlp.tiles[0] <= 2 && mlp.tiles[mlp.n_layers] <= 2;
    if (small_io && mlp.n_layers == 3 && mlp.tiles[1] == mlp.tiles[2] && mlp.tiles[1] <= 16 && mlp_nw == mlp.tiles[1]) spec = 1;
    if (small_io && mlp.n_layers == 4 && mlp.tiles[1] == mlp.tiles[2] && mlp.tiles[2] == mlp.tiles[3] && mlp.tiles[1] <= 4 &&
        mlp_nw == mlp.tiles[1]) spec = 2;
    if (sw.mlp_generic) spec = 0;
    const bool single_step = per_particle_state && ra.H == 1;
    if (single_step) spec = 3;
    if (sw.mlp_bf16 && spec == 1 && !per_particle_state && !final_state && !record) {
        // opt-in reduced-precision mode (kernels_mlp.hpp): never selected automatically
        dim3 bgrid((ra.n_pop + MLP_TP - 1) / MLP_TP, A), bblock(mlp_nw * 64);
        dominant_kernel = sw.mlp_bf16 == 3 ? "k_rollout_mlp_bf16<3>" : "k_rollout_mlp_bf16<1>";
        const void* bfn = sw.mlp_bf16 == 3 ? (const void*)k_rollout_mlp_bf16<3> : (const void*)k_rollout_mlp_bf16<1>;
        if (lds > 64 * 1024) ensure_max_lds(bfn, 159 * 1024);
        prof_begin();
        if (sw.mlp_bf16 == 3) hipLaunchKernelGGL(k_rollout_mlp_bf16<3>, bgrid, bblock, lds, stream, q);
        else hipLaunchKernelGGL(k_rollout_mlp_bf16<1>, bgrid, bblock, lds, stream, q);
        HIP_CHECK(hipGetLastError());
        prof_end();
        return;
    }
    // small networks, one wave per four particles and nothing through LDS between layers (kernels_mlp_w4.hpp): 2..4 Dense
    // layers of at most 32 units, dim_S <= 20, dim_U <= 8 -- the reference tutorials' 4-32-32-32-3 and 26-32-32-32-20
    if (!sw.mlp_generic && sw.mlp_w4 != 0 && !single_step && d_w4pack.p != nullptr) {
        const bool small = true;                                   // (bbmpc_set_mlp packed the operands: the shape qualifies)
        bool tanh_net = mlp.act[mlp.n_layers - 1] == BBMPC_ACT_NONE;
        for (int l = 0; l + 1 < mlp.n_layers; ++l) tanh_net = tanh_net && mlp.act[l] == BBMPC_ACT_TANH;
        const size_t wlds = (size_t)mlp_w4_lds_layout(ra.H, U, S).total * sizeof(float);
        if (small && wlds <= 159 * 1024) {
            using KFn = void (*)(MlpRolloutArgs);
            static const KFn table[2][3] = {{k_rollout_mlp_w4<2, false>, k_rollout_mlp_w4<3, false>, k_rollout_mlp_w4<4, false>},
                                            {k_rollout_mlp_w4<2, true>, k_rollout_mlp_w4<3, true>, k_rollout_mlp_w4<4, true>}};
            const KFn wfn = table[tanh_net ? 1 : 0][mlp.n_layers - 2];
            if (wlds > 64 * 1024) ensure_max_lds((const void*)wfn, 159 * 1024);
            dim3 wgrid((ra.n_pop + W4_TP - 1) / W4_TP, per_particle_state ? 1 : A), wblock(256);
            dominant_kernel = "k_rollout_mlp_w4";
            snprintf(dominant_inst, sizeof(dominant_inst), "k_rollout_mlp_w4<%d, %s>", mlp.n_layers, tanh_net ? "true" : "false");
            prof_begin();
            hipLaunchKernelGGL(wfn, wgrid, wblock, wlds, stream, q);
            HIP_CHECK(hipGetLastError());
            prof_end();
            return;
        }
    }
    // small networks: one wave per 16-particle tile, the whole Dense stack in its registers (kernels_mlp_wave.hpp)
    if (!sw.mlp_generic && sw.mlp_wave != 0 && !single_step && small_io && mlp.n_layers >= 2 && mlp.n_layers <= 4 && mlp.tiles[1] <= 4) {
        bool same = true;
        for (int l = 2; l < mlp.n_layers; ++l) same = same && mlp.tiles[l] == mlp.tiles[1];
        if (same) {
            using KFn = void (*)(MlpRolloutArgs);
            static const KFn table[2][3][4] = {
                {{k_rollout_mlp_wave<1, 1, false>, k_rollout_mlp_wave<1, 2, false>, k_rollout_mlp_wave<1, 3, false>, k_rollout_mlp_wave<1, 4, false>},
                 {k_rollout_mlp_wave<2, 1, false>, k_rollout_mlp_wave<2, 2, false>, k_rollout_mlp_wave<2, 3, false>, k_rollout_mlp_wave<2, 4, false>},
                 {k_rollout_mlp_wave<3, 1, false>, k_rollout_mlp_wave<3, 2, false>, k_rollout_mlp_wave<3, 3, false>, k_rollout_mlp_wave<3, 4, false>}},
                {{k_rollout_mlp_wave<1, 1, true>, k_rollout_mlp_wave<1, 2, true>, k_rollout_mlp_wave<1, 3, true>, k_rollout_mlp_wave<1, 4, true>},
                 {k_rollout_mlp_wave<2, 1, true>, k_rollout_mlp_wave<2, 2, true>, k_rollout_mlp_wave<2, 3, true>, k_rollout_mlp_wave<2, 4, true>},
                 {k_rollout_mlp_wave<3, 1, true>, k_rollout_mlp_wave<3, 2, true>, k_rollout_mlp_wave<3, 3, true>, k_rollout_mlp_wave<3, 4, true>}}};
            bool tanh_net = mlp.act[mlp.n_layers - 1] == BBMPC_ACT_NONE;
            for (int l = 0; l + 1 < mlp.n_layers; ++l) tanh_net = tanh_net && mlp.act[l] == BBMPC_ACT_TANH;
            const KFn wfn = table[tanh_net ? 1 : 0][mlp.n_layers - 2][mlp.tiles[1] - 1];
            const int wht = mlp.tiles[1];
            const size_t wlds = (size_t)mlp_wave_lds_layout(ra.H, U, S, mlp.n_layers - 1, wht).total * sizeof(float);
            if (wlds <= 159 * 1024) {
                if (wlds > 64 * 1024) ensure_max_lds((const void*)wfn, 159 * 1024);
                dim3 wgrid((ra.n_pop + MLP_TP - 1) / MLP_TP, per_particle_state ? 1 : A), wblock(64 * mlp_wave_waves(wht));
                dominant_kernel = "k_rollout_mlp_wave";
                prof_begin();
                hipLaunchKernelGGL(wfn, wgrid, wblock, wlds, stream, q);
                HIP_CHECK(hipGetLastError());
                prof_end();
                return;
            }
        }
    }
    const void* fn = spec == 3 ? (const void*)k_step_mlp : spec == 1 ? (const void*)k_rollout_mlp<1> : (spec == 2 ? (const void*)k_rollout_mlp<2> : (const void*)k_rollout_mlp<0>);
    if (lds > 64 * 1024) ensure_max_lds(fn, 159 * 1024);
    // pair mode (two tiles per workgroup, software-pipelined) when there are more tiles than CUs can hold one each
    const long tiles_total = (long)((ra.n_pop + MLP_TP - 1) / MLP_TP) * A;
    // the 26-200-200-20 tanh/tanh/linear family has its own kernels (all of them can record the trajectory)
    const bool fam_dims = spec == 1 && mlp.tiles[1] == 13 && !per_particle_state;     // (spec == 1: two hidden layers of equal tile count, <= 32 inputs and outputs)
    const bool fam_ok = fam_dims && mlp.act[0] == BBMPC_ACT_TANH && mlp.act[1] == BBMPC_ACT_TANH && mlp.act[2] == BBMPC_ACT_NONE;
    const bool pair_ok = fam_ok;
    int pair = (pair_ok && tiles_total > 256) ? 1 : 0;
    if (sw.mlp_pair >= 0) pair = (sw.mlp_pair != 0 && pair_ok) ? 1 : 0;
    // quad mode (4 particles per workgroup, 4x4x1_16b MFMA, all weights in registers) when the population is too
    // small to give every CU a 16-particle tile
    {
        const bool q4_dims = mlp.dims[0] <= 28 && mlp.dims[1] == 200 && mlp.dims[2] == 200 && mlp.dims[3] <= 64;
        const bool q4_ok = fam_ok && q4_dims;
        // k_rollout_mlp_q4s: two hidden layers of 200 or of 256 units, dim_S <= 20, dim_U <= 8, any activations
        const bool wide = spec == 1 && mlp.tiles[1] == 16 && !per_particle_state && mlp.dims[0] <= 28 && mlp.dims[1] == 256 && mlp.dims[2] == 256;
        const bool q4s_ok = ((fam_dims && q4_dims) || wide) && S <= 20 && U <= 8 && mlp.dims[3] == S && d_wq4s0.p != nullptr &&
                            (ra.reward_kind == REW_CHEETAH || ra.reward_kind == REW_NONE);
        // measured on MI355X (tools/q4_sweep.py, PI2, H = 30, us per control step): a "wave" of 256 quad workgroups (one
        // per CU, 1024 particles) costs ~400 us, the 16-particle tiling ~850 us for anything up to 4096 particles:
        // quads win up to two waves (N*A <= 2048: 810 vs 860), lose from the third on (2500: 1177 vs 868)
        const long quads_total = (long)((ra.n_pop + 3) / 4) * A;
        int q4 = ((q4_ok || q4s_ok) && quads_total <= 512) ? 1 : 0;
        if (sw.mlp_q4 >= 0) q4 = (sw.mlp_q4 != 0 && (q4_ok || q4s_ok)) ? 1 : 0;
        // four equal waves, the state in registers, the last layer from registers, two barriers per model step (kernels_mlp_q4s.hpp)
        if (q4 && !sw.mlp_generic && sw.mlp_q4s && q4s_ok) {
            const size_t qlds = (size_t)mlp_q4s_lds_floats(wide ? 64 : 50, 7, ra.H, U) * sizeof(float);
            const int qpairs = 4 * ((ra.H * U + 3) / 4);
            if (qlds <= 160 * 1024 && qpairs <= Q4S_MAX_ACTION_PAIRS) {
                using KFn = void (*)(MlpRolloutArgs);
                const bool ne1 = qpairs <= 256;
                const bool tanh_net = mlp.act[0] == BBMPC_ACT_TANH && mlp.act[1] == BBMPC_ACT_TANH && mlp.act[2] == BBMPC_ACT_NONE;
                const bool relu_net = mlp.act[0] == BBMPC_ACT_RELU && mlp.act[1] == BBMPC_ACT_RELU && mlp.act[2] == BBMPC_ACT_NONE;
                const KFn fn = wide     ? (tanh_net ? (ne1 ? k_rollout_mlp_q4s<64, 7, ACT_TANH, ACT_TANH, ACT_NONE, 1> : k_rollout_mlp_q4s<64, 7, ACT_TANH, ACT_TANH, ACT_NONE, 2>)
                                                    : (ne1 ? k_rollout_mlp_q4s<64, 7, ACT_RT, ACT_RT, ACT_RT, 1> : k_rollout_mlp_q4s<64, 7, ACT_RT, ACT_RT, ACT_RT, 2>))
                             : tanh_net ? (ne1 ? k_rollout_mlp_q4s<50, 7, ACT_TANH, ACT_TANH, ACT_NONE, 1> : k_rollout_mlp_q4s<50, 7, ACT_TANH, ACT_TANH, ACT_NONE, 2>)
                             : relu_net ? (ne1 ? k_rollout_mlp_q4s<50, 7, ACT_RELU, ACT_RELU, ACT_NONE, 1> : k_rollout_mlp_q4s<50, 7, ACT_RELU, ACT_RELU, ACT_NONE, 2>)
                                        : (ne1 ? k_rollout_mlp_q4s<50, 7, ACT_RT, ACT_RT, ACT_RT, 1> : k_rollout_mlp_q4s<50, 7, ACT_RT, ACT_RT, ACT_RT, 2>);
                if (qlds > 64 * 1024) ensure_max_lds((const void*)fn, 160 * 1024);
                dim3 qgrid((ra.n_pop + 3) / 4, A), qblock(256);
                dominant_kernel = "k_rollout_mlp_q4s";
                {   // the instantiation as rocprofv3 prints it (bbmpc_profile_instantiation): HG, K0G, three activations, NE
                    const int rt = -1;
                    const int a0 = (tanh_net || (!wide && relu_net)) ? mlp.act[0] : rt, a1 = (tanh_net || (!wide && relu_net)) ? mlp.act[1] : rt,
                              a2 = (tanh_net || (!wide && relu_net)) ? mlp.act[2] : rt;
                    snprintf(dominant_inst, sizeof(dominant_inst), "k_rollout_mlp_q4s<%d, 7, %d, %d, %d, %d>", wide ? 64 : 50, a0, a1, a2, ne1 ? 1 : 2);
                }
                q.state_copy = mlp_state_copy;
                mlp_state_copy = nullptr;
                prof_begin();
                hipLaunchKernelGGL(fn, qgrid, qblock, qlds, stream, q);
                HIP_CHECK(hipGetLastError());
                prof_end();
                return;
            }
        }
        if (q4 && q4_ok && !sw.mlp_generic) {
            const size_t qlds = (size_t)mlp_q4_lds_floats(50, 7, 4, ra.H, U, S) * sizeof(float);
            if (qlds <= 160 * 1024) {
                auto fn = k_rollout_mlp_q4<50, 7, 4, ACT_TANH, ACT_TANH, ACT_NONE>;
                if (qlds > 64 * 1024) ensure_max_lds((const void*)fn, 160 * 1024);
                dim3 qgrid((ra.n_pop + 3) / 4, A), qblock(256);
                dominant_kernel = "k_rollout_mlp_q4";
                prof_begin();
                hipLaunchKernelGGL(fn, qgrid, qblock, qlds, stream, q);
                HIP_CHECK(hipGetLastError());
                prof_end();
                return;
            }
        }
    }
    if (pair_ok && !sw.mlp_generic) {
        // pipelined kernel for the 26-200-200-20 family: two tiles per workgroup when tiles outnumber the CUs,
        // one tile per workgroup otherwise (more workgroups beat better per-workgroup efficiency then)
        const int nt = pair ? 2 : 1;
        const size_t plds = (size_t)mlp_pair_lds_floats(13, ra.H, U, S, nt) * sizeof(float);
        if (plds <= 159 * 1024) {
            dim3 pgrid((ra.n_pop + nt * MLP_TP - 1) / (nt * MLP_TP), A), pblock(mlp_pair_waves(13, nt) * 64);
            dominant_kernel = "k_rollout_mlp_pair";
            prof_begin();
            if (nt == 2 && S == 20 && U == 6 && ra.H == 50 && ra.reward_kind == REW_CHEETAH && !q.traj && mlp.half_tail[1] && mlp.half_tail[2]) {
                // BASELINE config 5: dimensions and the reward at compile time (no register spills, kernels_mlp.hpp)
                ensure_max_lds((const void*)(k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2, 20, 6, 50, REW_CHEETAH, 1>), 159 * 1024);
                hipLaunchKernelGGL((k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2, 20, 6, 50, REW_CHEETAH, 1>), pgrid, pblock, plds, stream, q);
            } else if (nt == 2 && S == 20 && U == 6 && ra.H == 50) {
                ensure_max_lds((const void*)(k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2, 20, 6, 50>), 159 * 1024);
                hipLaunchKernelGGL((k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2, 20, 6, 50>), pgrid, pblock, plds, stream, q);
            } else if (nt == 2 && S == 20 && U == 6) {
                ensure_max_lds((const void*)(k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2, 20, 6>), 159 * 1024);
                hipLaunchKernelGGL((k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2, 20, 6>), pgrid, pblock, plds, stream, q);
            } else if (nt == 2) {
                ensure_max_lds((const void*)(k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2>), 159 * 1024);
                hipLaunchKernelGGL((k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 2>), pgrid, pblock, plds, stream, q);
            } else {
                ensure_max_lds((const void*)(k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 1>), 159 * 1024);
                hipLaunchKernelGGL((k_rollout_mlp_pair<13, ACT_TANH, ACT_TANH, ACT_NONE, 1>), pgrid, pblock, plds, stream, q);
            }
            HIP_CHECK(hipGetLastError());
            prof_end();
            return;
        }
    }
    dim3 grid((ra.n_pop + MLP_TP - 1) / MLP_TP, per_particle_state ? 1 : A), block(mlp_nw * 64);
    prof_begin();
    if (spec == 3) hipLaunchKernelGGL(k_step_mlp, grid, block, lds, stream, q);
    else if (spec == 1) hipLaunchKernelGGL(k_rollout_mlp<1>, grid, block, lds, stream, q);
    else if (spec == 2) hipLaunchKernelGGL(k_rollout_mlp<2>, grid, block, lds, stream, q);
    else hipLaunchKernelGGL(k_rollout_mlp<0>, grid, block, lds, stream, q);
    HIP_CHECK(hipGetLastError());
    prof_end();
}

}  // namespace bbmpc
