#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    int a = l, b = 100 + l;
    // row_ror:4  dpp_ctrl = 0x120 + n
    out[l] = __builtin_amdgcn_update_dpp(0, a, 0x124, 0xf, 0xf, false);
    out[64 + l] = __builtin_amdgcn_update_dpp(0, a, 0x128, 0xf, 0xf, false);
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + l] = r32[0]; out[192 + l] = r32[1];
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[256 + l] = r16[0]; out[320 + l] = r16[1];
}
int main() {
    int* d; hipMalloc(&d, 384 * 4);
    k<<<1, 64>>>(d);
    int h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"row_ror:4", "row_ror:8", "pl32swap.0(vdst=a)", "pl32swap.1(src=b)", "pl16swap.0", "pl16swap.1"};
    for (int i = 0; i < 6; ++i) { printf("%s:", names[i]); for (int l = 0; l < 64; ++l) printf(" %d", h[i * 64 + l]); printf("\n"); }
    return 0;
}
