// User-supplied reward / dynamics functions as DEVICE code compiled at run time (hiprtc).
//
// The reference accepts any callable as `reward_function` / `dynamics_function`
// (trajectory_evaluators/deterministic.py:13-18, called at :65-66 and :99-100; plugin contracts in SURVEY.md 1-L1):
//     reward  r(current_state[B,S], actions[B,U], next_state[B,S]) -> [B]
//     dynamics f(x[B,S+U], train) -> delta[B,S]           (true model: next = delta + state, transforms.py:34)
// Python callables cannot run inside a kernel, so the counterpart here is a HIP source string that defines
//     __device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U);
//     __device__ void  bbmpc_user_dynamics(const float* x /*[S+U]*/, float* delta /*[S]*/, int S, int U);
// per row.  bbmpc_set_reward_source / bbmpc_set_dynamics_source compile it together with the two row kernels below and
// the engine calls them once per planning step from its step-wise evaluator (kernels_user.hpp).  libhiprtc is bound at
// run time, next to the HIP runtime the process already uses; compiling needs no GPU.
#pragma once
#include <dlfcn.h>
#include "../../include/bbmpc.h"   // bbmpc_rows_callback
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace bbmpc {

struct Hiprtc {
    typedef void* Program;
    int (*CreateProgram)(Program*, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*CompileProgram)(Program, int, const char**) = nullptr;
    int (*GetProgramLogSize)(Program, size_t*) = nullptr;
    int (*GetProgramLog)(Program, char*) = nullptr;
    int (*GetCodeSize)(Program, size_t*) = nullptr;
    int (*GetCode)(Program, char*) = nullptr;
    int (*DestroyProgram)(Program*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;

    static const Hiprtc& get() {
        static const Hiprtc api = load();
        return api;
    }

private:
    static Hiprtc load() {
        void* lib = dlopen("libhiprtc.so", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) {
            // the copy that ships with the HIP runtime this process runs on (PyTorch bundles its own pair)
            Dl_info info;
            if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
                std::string p(info.dli_fname);
                const size_t slash = p.rfind('/');
                if (slash != std::string::npos) lib = dlopen((p.substr(0, slash + 1) + "libhiprtc.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
            }
        }
        const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
        for (int i = 0; !lib && i < 3; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!lib) throw std::runtime_error("libhiprtc.so not found (needed to compile user reward / dynamics device functions)");
        Hiprtc a;
        auto sym = [&](const char* s) {
            void* p = dlsym(lib, s);
            if (!p) throw std::runtime_error(std::string("libhiprtc: missing symbol ") + s);
            return p;
        };
        a.CreateProgram = reinterpret_cast<decltype(a.CreateProgram)>(sym("hiprtcCreateProgram"));
        a.CompileProgram = reinterpret_cast<decltype(a.CompileProgram)>(sym("hiprtcCompileProgram"));
        a.GetProgramLogSize = reinterpret_cast<decltype(a.GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
        a.GetProgramLog = reinterpret_cast<decltype(a.GetProgramLog)>(sym("hiprtcGetProgramLog"));
        a.GetCodeSize = reinterpret_cast<decltype(a.GetCodeSize)>(sym("hiprtcGetCodeSize"));
        a.GetCode = reinterpret_cast<decltype(a.GetCode)>(sym("hiprtcGetCode"));
        a.DestroyProgram = reinterpret_cast<decltype(a.DestroyProgram)>(sym("hiprtcDestroyProgram"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("hiprtcGetErrorString"));
        return a;
    }
};

constexpr int USER_KIND_REWARD = 1, USER_KIND_DYNAMICS = 2;

// The row kernels the engine launches around the user's function.  BBMPC_S / BBMPC_U are compile-time so the per-row
// arrays live in registers; rows are [batch, S] / [batch, astride] row-major.
inline std::string user_program_source(const std::string& user_src, int kind) {
    std::string s;
    s += "// ---- user source ------------------------------------------------------------------\n";
    s += user_src;
    s += "\n// ---- row kernels (blackbox_mpc_amd/csrc/rtc.hpp) ------------------------------------\n";
    if (kind == USER_KIND_REWARD) {
        s += R"RTC(
extern "C" __global__ void bbmpc_user_reward_rows(const float* __restrict__ cur, const float* __restrict__ nxt,
                                                  const float* __restrict__ act, int astride, int batch,
                                                  float* __restrict__ total, int accumulate) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    float c[BBMPC_S], n[BBMPC_S], a[BBMPC_U];
    for (int i = 0; i < BBMPC_S; ++i) { c[i] = cur[(size_t)b * BBMPC_S + i]; n[i] = nxt[(size_t)b * BBMPC_S + i]; }
    for (int i = 0; i < BBMPC_U; ++i) a[i] = act[(size_t)b * astride + i];
    // reward_function(current_state, actions, next_state): the argument order of the CALL (deterministic.py:65-66)
    const float r = bbmpc_user_reward(c, a, n, BBMPC_S, BBMPC_U);
    total[b] = accumulate ? total[b] + r : r;
}

// The learned MLP's rollouts run on the matrix cores (kernels_mlp.hpp) and leave the state after every step in
// traj [H][A][Nst][S]; this scores the whole trajectory of one candidate per lane: sum_t r(s_t, a_t, s_t+1), NaN -> -1e6
// (deterministic.py:62-77).  `rewards` holds -(penalty) from the rollout kernel (0 without one): R - penalty.
extern "C" __global__ void bbmpc_user_reward_traj(int n_pop, int A, int H, int Nst, int from_ref,
                                                  const float* __restrict__ state, const float* __restrict__ traj,
                                                  const float* __restrict__ seq, const float* __restrict__ cand,
                                                  float* __restrict__ rewards) {
    const int a = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_pop) return;
    const int HU = H * BBMPC_U;
    float c[BBMPC_S], nx[BBMPC_S], ac[BBMPC_U];
    for (int i = 0; i < BBMPC_S; ++i) c[i] = state[a * BBMPC_S + i];
    float total = 0.0f;
    for (int t = 0; t < H; ++t) {
        const float* row = traj + ((((size_t)t * A + a) * Nst) + n) * BBMPC_S;
        for (int i = 0; i < BBMPC_S; ++i) nx[i] = row[i];
        for (int u = 0; u < BBMPC_U; ++u) {
            const int j = t * BBMPC_U + u;
            ac[u] = from_ref ? seq[((size_t)n * A + a) * HU + j] : cand[((size_t)a * HU + j) * Nst + n];
        }
        total = total + bbmpc_user_reward(c, ac, nx, BBMPC_S, BBMPC_U);
        for (int i = 0; i < BBMPC_S; ++i) c[i] = nx[i];
    }
    if (total != total) total = -1.0e6f;
    rewards[(size_t)a * Nst + n] = total + rewards[(size_t)a * Nst + n];
}
)RTC";
    } else {
        s += R"RTC(
extern "C" __global__ void bbmpc_user_dynamics_rows(const float* __restrict__ states, const float* __restrict__ act,
                                                    int astride, int batch, float* __restrict__ next_states) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    float x[BBMPC_S + BBMPC_U], d[BBMPC_S];
    for (int i = 0; i < BBMPC_S; ++i) x[i] = states[(size_t)b * BBMPC_S + i];                    // process_input: concat
    for (int i = 0; i < BBMPC_U; ++i) x[BBMPC_S + i] = act[(size_t)b * astride + i];
    bbmpc_user_dynamics(x, d, BBMPC_S, BBMPC_U);                                                // f(x, train=False) -> delta
    for (int i = 0; i < BBMPC_S; ++i) next_states[(size_t)b * BBMPC_S + i] = d[i] + x[i];        // transforms.py:34
}
)RTC";
    }
    return s;
}

#include "_embed.inc"      // k_embed_fastmath, k_embed_models: the leaf-math headers as text (generated by _build.py)

// The FUSED form: one lane per candidate trajectory, the whole H-step recurrence in registers, with the user's
// function(s) inlined next to the built-in model / rewards (the engine's own models.hpp, compiled from the same text
// with the same flags).  Used whenever the dynamics is not the learned MLP (whose rollouts live on the matrix cores);
// the step-wise evaluator (kernels_user.hpp) stays for MLP + user reward.
//   BBMPC_DYN_KIND 1 = PendulumTrueModel (op-for-op form), 3 = bbmpc_user_dynamics
//   BBMPC_REW_KIND 1 / 2 = built-in pendulum / cheetah reward, 3 = bbmpc_user_reward
inline std::string user_rollout_source(const std::string& reward_src, const std::string& dynamics_src) {
    std::string s = "#include \"models.hpp\"\n";
    s += "// ---- user reward --------------------------------------------------------------------\n" + reward_src + "\n";
    s += "// ---- user dynamics ------------------------------------------------------------------\n" + dynamics_src + "\n";
    s += R"RTC(
extern "C" __global__ void bbmpc_user_rollout(int n_pop, int A, int H, int Nst, int from_ref, int pen, int fix_q1,
                                              const float* __restrict__ state, const float* __restrict__ seq,
                                              const float* cand, float* samples, const float* __restrict__ lo,
                                              const float* __restrict__ hi, float* __restrict__ rewards,
                                              float* __restrict__ penalty_out) {
    constexpr int S = BBMPC_S, U = BBMPC_U;
    const int a = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_pop) return;
    const int HU = H * U;
    float x[S + U], nx[S];
    for (int i = 0; i < S; ++i) x[i] = state[a * S + i];                       // tf.tile(current_states, [nopt, 1])
    float total = 0.0f, pen_acc = 0.0f;
    for (int t = 0; t < H; ++t) {
        for (int u = 0; u < U; ++u) {
            const int j = t * U + u;
            float v = from_ref ? seq[((size_t)n * A + a) * HU + j] : cand[((size_t)a * HU + j) * Nst + n];
            if (pen) {
                const float xf = bbmpc::clipf(v, lo[u], hi[u]);
                const float d = v - xf;
                pen_acc = pen_acc + d * d;
                v = xf;
            }
            if (samples) samples[((size_t)a * HU + j) * Nst + n] = v;
            x[S + u] = v;
        }
#if BBMPC_DYN_KIND == 3
        {
            float d[S];
            bbmpc_user_dynamics(x, d, S, U);                                   // f(x, train=False) -> delta
            for (int i = 0; i < S; ++i) nx[i] = d[i] + x[i];                     // transforms.py:34
        }
#else
        {
            float ss[3] = {x[0], x[1], x[2]};
            const float ac[1] = {x[3]};
            const bbmpc::PendulumModel model{fix_q1 != 0};
            (void)model.step(ss, ac);
            nx[0] = ss[0]; nx[1] = ss[1]; nx[2] = ss[2];
        }
#endif
#if BBMPC_REW_KIND == 3
        total = total + bbmpc_user_reward(x, x + S, nx, S, U);                   // (current_state, actions, next_state)
#else
        total = total + bbmpc::reward_generic(BBMPC_REW_KIND, fix_q1 != 0, x, x + S, nx, S, U);
#endif
        for (int i = 0; i < S; ++i) x[i] = nx[i];
    }
    if (total != total) total = -1.0e6f;                                         // deterministic.py:75-77
    if (pen) {
        const float nr = sqrtf(pen_acc);                                         // tf.norm(...)**2  pi2.py:72-75
        const float pv = nr * nr;
        total = total - pv;
        if (penalty_out) penalty_out[(size_t)a * Nst + n] = pv;
    }
    rewards[(size_t)a * Nst + n] = total;
}
)RTC";
    return s;
}

// Compile for gfx950; returns the code object.  Throws std::runtime_error with the compiler log on failure.
inline std::vector<char> compile_rtc(const std::string& src, const char* name, const std::vector<std::string>& defines,
                                     bool with_engine_headers) {
    const Hiprtc& r = Hiprtc::get();
    Hiprtc::Program prog = nullptr;
    const char* hdr_text[] = {k_embed_fastmath, k_embed_models};
    const char* hdr_name[] = {"fastmath.hpp", "models.hpp"};
    int rc = r.CreateProgram(&prog, src.c_str(), name, with_engine_headers ? 2 : 0, with_engine_headers ? hdr_text : nullptr,
                             with_engine_headers ? hdr_name : nullptr);
    if (rc != 0) throw std::runtime_error(std::string("hiprtcCreateProgram: ") + r.GetErrorString(rc));
    // one rounding per source operation, as everywhere in the engine (and as the reference's TF ops round)
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
    for (const std::string& d : defines) opts.push_back(d.c_str());
    rc = r.CompileProgram(prog, (int)opts.size(), opts.data());
    std::string log;
    size_t n = 0;
    if (r.GetProgramLogSize(prog, &n) == 0 && n > 1) {
        log.resize(n);
        (void)r.GetProgramLog(prog, &log[0]);
    }
    if (rc != 0) {
        (void)r.DestroyProgram(&prog);
        throw std::runtime_error(std::string("user device function failed to compile (") + r.GetErrorString(rc) + "):\n" + log);
    }
    std::vector<char> code;
    if (r.GetCodeSize(prog, &n) != 0 || n == 0) {
        (void)r.DestroyProgram(&prog);
        throw std::runtime_error("hiprtcGetCodeSize failed");
    }
    code.resize(n);
    rc = r.GetCode(prog, code.data());
    (void)r.DestroyProgram(&prog);
    if (rc != 0) throw std::runtime_error(std::string("hiprtcGetCode: ") + r.GetErrorString(rc));
    return code;
}

inline std::vector<char> compile_user_program(const std::string& user_src, int kind, int S, int U) {
    return compile_rtc(user_program_source(user_src, kind), kind == USER_KIND_REWARD ? "bbmpc_user_reward.hip" : "bbmpc_user_dynamics.hip",
                       {"-DBBMPC_S=" + std::to_string(S), "-DBBMPC_U=" + std::to_string(U)}, false);
}

inline std::vector<char> compile_user_rollout(const std::string& reward_src, const std::string& dynamics_src, int dyn_kind, int rew_kind,
                                              int S, int U) {
    return compile_rtc(user_rollout_source(reward_src, dynamics_src), "bbmpc_user_rollout.hip",
                       {"-DBBMPC_S=" + std::to_string(S), "-DBBMPC_U=" + std::to_string(U), "-DBBMPC_DYN_KIND=" + std::to_string(dyn_kind),
                        "-DBBMPC_REW_KIND=" + std::to_string(rew_kind)}, true);
}

struct UserFunction {
    std::string source;
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    hipFunction_t fn_traj = nullptr;       // reward module only: bbmpc_user_reward_traj
    bbmpc_rows_callback cb = nullptr;      // or: a host callback working on device memory (bbmpc_set_*_callback)
    void* cb_user = nullptr;
    bool ready() const { return fn != nullptr || cb != nullptr; }
    void release() {
        if (module) (void)hipModuleUnload(module);
        module = nullptr;
        fn = nullptr;
        fn_traj = nullptr;
    }
};

}  // namespace bbmpc
