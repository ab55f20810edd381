// diagnostic: what the primitives of digat_reduce.inc do to lane-identifying inputs
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#include "digat_reduce.inc"
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float a = (float)l, b = 1000.f + l;
    out[0 * 64 + l] = rows_sum(a);
    float o[4];
    pair_add_bit3_x4(o, a, a, a, a, b, b, b, b);
    out[1 * 64 + l] = o[0];
    pair_add_bit2_x4(o, a, a, a, a, b, b, b, b);
    out[2 * 64 + l] = o[0];
    out[3 * 64 + l] = pair_add_bit1(a, b, (l & 2) != 0);
    out[4 * 64 + l] = pair_add_bit0(a, b, (l & 1) != 0);
    const unsigned u = __builtin_bit_cast(unsigned, a);
    const auto s16 = __builtin_amdgcn_permlane16_swap(u, __builtin_bit_cast(unsigned, b), false, false);
    out[5 * 64 + l] = __builtin_bit_cast(float, s16[0]);
    out[6 * 64 + l] = __builtin_bit_cast(float, s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(u, __builtin_bit_cast(unsigned, b), false, false);
    out[7 * 64 + l] = __builtin_bit_cast(float, s32[0]);
    out[8 * 64 + l] = __builtin_bit_cast(float, s32[1]);
}
int main() {
    float* d; float h[9 * 64];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[9] = {"rows_sum(l)", "bit3(l | 1000+l)", "bit2(l | 1000+l)", "bit1", "bit0", "pl16swap[0] (a=l,b=1000+l)", "pl16swap[1]", "pl32swap[0]", "pl32swap[1]"};
    for (int r = 0; r < 9; ++r) { printf("%-28s:", names[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
    return 0;
}
