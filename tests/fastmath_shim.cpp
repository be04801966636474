// Host build of the engine's range-limited math (blackbox_mpc_amd/csrc/fastmath.hpp) so the accuracy
// sweep can run without a GPU.  The header uses only IEEE +,-,*,/ and fmaf, so host == device bitwise
// (the GPU test test_fastmath_device_matches_host checks that claim on the device).
#include "../blackbox_mpc_amd/csrc/fastmath.hpp"
extern "C" {
void shim_sincos(const float* x, int n, float* s, float* c) { for (int i = 0; i < n; ++i) bbmpc::bb_sincosf(x[i], s + i, c + i); }
void shim_sin_pi(const float* x, int n, float* o) { for (int i = 0; i < n; ++i) o[i] = bbmpc::bb_sinf_pi(x[i]); }
void shim_sin_fold(const float* x, int n, float* o) { for (int i = 0; i < n; ++i) o[i] = bbmpc::bb_sinf_fold_0_2pi(x[i]); }
void shim_atan2(const float* y, const float* x, int n, float* o) { for (int i = 0; i < n; ++i) o[i] = bbmpc::bb_atan2f(y[i], x[i]); }
void shim_floormod(const float* x, float y, int n, float* o) { for (int i = 0; i < n; ++i) o[i] = bbmpc::bb_floormod_pos(x[i], y); }
}
