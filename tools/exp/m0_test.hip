// Does an LDS-DMA (global_load_lds_dwordx4) reach LDS addresses beyond 64 KiB through M0 on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__global__ void k(const float* src, float* out, unsigned off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float* f = (float*)smem;
    for (int i = threadIdx.x; i < 160 * 256; i += 64) f[i] = -1.f;
    __syncthreads();
    lds_dma16(src + threadIdx.x * 4, base + off);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report: where did the 256 floats land?
    int found = -1;
    for (int i = 0; i < 160 * 256; ++i) if (f[i] == 1000.f) { found = i * 4; break; }
    if (threadIdx.x == 0) { out[0] = (float)found; out[1] = f[off / 4 + 5]; }
}
int main() {
    float *src, *out; float h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000.f + i;
    hipMalloc(&src, 1024); hipMalloc(&out, 64); hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (unsigned off : {0u, 32768u, 65536u, 65536u + 4096u, 100000u / 16 * 16, 150u * 1024u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, src, out, off);
        float r[2]; hipMemcpy(r, out, 8, hipMemcpyDeviceToHost);
        printf("dst offset %u -> data found at byte %d (expected %u), f[off+5]=%g\n", off, (int)r[0], off, r[1]);
    }
    return 0;
}
