// checks digat_reduce.inc's wave_sum_packed<4 / 8 / 16> against a host sum (round 6; run on the GPU box)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/eq8_packed/packed_reduce_test tools/exp/eq8_packed/packed_reduce_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#include "digat_reduce.inc"
template <int V>
__global__ void k(const float* in, float* out) {       // in [64][16], out [3][64]
    float v[V];
    for (int i = 0; i < V; ++i) v[i] = in[threadIdx.x * 16 + i];
    out[threadIdx.x] = wave_sum_packed<V>(v, threadIdx.x & 63);
}
int main() {
    float h[64 * 16], *d, *o, r[64];
    for (int i = 0; i < 64 * 16; ++i) h[i] = (float)((i * 2654435761u >> 8) % 2001) / 64.f - 15.f;     // exact in fp32; sums exact too
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    int bad = 0;
    for (int V = 4; V <= 16; V *= 2) {
        if (V == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, o);
        else if (V == 8) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, d, o);
        else hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, d, o);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            double s = 0; for (int t = 0; t < 64; ++t) s += h[t * 16 + (l & (V - 1))];
            if (fabs(s - r[l]) > 1e-3) { if (bad < 8) printf("V=%d lane %d: got %f want %f\n", V, l, r[l], s); ++bad; }
        }
    }
    printf("packed_reduce_test: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
