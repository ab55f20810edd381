// FETCH_SIZE / WRITE_SIZE calibration on known byte counts (round 6).
//
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other access shapes are
// uncalibrated.  bench.py's `traffic` doubled FETCH_SIZE for every kernel, which over-corrects the projection GEMM (raw FETCH ~ the A
// rows it must read).  Each kernel below reads (or writes) every byte of a 1 GiB buffer exactly once — far beyond the 256 MB Infinity
// Cache — in ONE access shape of the product kernels; tools/exp/fetch_calib.sh runs it under `rocprofv3 --pmc FETCH_SIZE` /
// `--pmc WRITE_SIZE` and prints factor = known bytes / (counter x 1024) per shape:
//   stream_f4      global_load_dwordx4, a wave reads 1 KB contiguous                          (the guide's calibrated case)
//   stream_dma     global_load_lds_dwordx4, 1 KB contiguous per wave instruction              (weight images of the GEMM)
//   rows128_dma    global_load_lds_dwordx4, 8 rows x 128 B per instruction, rows 1600 B apart (the GEMM's fp32 A tile)
//   rows1600_f4    global_load_dwordx4, whole 1600-B rows: 64 + 36 lanes                      (Eq. 8 / pooling kernels)
//   rows1600_f4g   the same rows through a random row index (gathered: the sparse Eq. 8 kernels' neighbour rows)
//   store_f4       global_store_dwordx4, 1 KB contiguous per wave instruction                 (WRITE_SIZE)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/fetch_calib tools/exp/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr size_t BYTES = 1ull << 30;

__global__ void __launch_bounds__(256) stream_f4(const float4* __restrict__ p, float* sink, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *sink = acc;
}
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__global__ void __launch_bounds__(256) stream_dma(const char* __restrict__ p, float* sink, size_t nkb) {
    __shared__ uint4 buf[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&buf[wave][0];
    for (size_t k = (size_t)blockIdx.x * 4 + wave; k < nkb; k += (size_t)gridDim.x * 4) lds_dma16(p + k * 1024 + lane * 16, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (buf[wave][lane].x == 0x12345678u) *sink = 1.f;
}
// pieces of 8 rows x 128 B, rows 1600 B apart: piece (rowblock, c) covers bytes [128 c, 128 c + 128) of rows 8 rowblock .. + 7 (c = 0 .. 11: 1536 of a
// row's 1600 bytes; the last 64 bytes of every row are read by a 13th, half-used piece in the product kernel: left out here, counted out below)
__global__ void __launch_bounds__(256) rows128_dma(const char* __restrict__ p, float* sink, size_t nrows) {
    __shared__ uint4 buf[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&buf[wave][0];
    const size_t pieces = (nrows / 8) * 12;
    for (size_t k = (size_t)blockIdx.x * 4 + wave; k < pieces; k += (size_t)gridDim.x * 4) {
        const size_t rb = k / 12, c = k % 12;
        lds_dma16(p + (rb * 8 + (lane >> 3)) * 1600 + c * 128 + (lane & 7) * 16, lds);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (buf[wave][lane].x == 0x12345678u) *sink = 1.f;
}
__global__ void __launch_bounds__(256) rows1600_f4(const float4* __restrict__ p, const int* __restrict__ idx, float* sink, size_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (size_t r = (size_t)blockIdx.x * 4 + wave; r < nrows; r += (size_t)gridDim.x * 4) {
        const size_t row = idx ? (size_t)idx[r] : r;
        const float4 a = p[row * 100 + lane];
        const float4 b = p[row * 100 + 64 + (lane < 36 ? lane : 35)];
        acc += a.x + b.y;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void __launch_bounds__(256) store_f4(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

int main() {
    char* buf; float* sink; int* idx;
    hipMalloc(&buf, BYTES); hipMalloc(&sink, 4);
    hipMemset(buf, 1, BYTES);
    const size_t nrows = BYTES / 1600;
    std::vector<int> perm(nrows);
    for (size_t i = 0; i < nrows; ++i) perm[i] = (int)i;
    unsigned s = 12345u;
    for (size_t i = nrows - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; const size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    hipMalloc(&idx, nrows * 4); hipMemcpy(idx, perm.data(), nrows * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_f4, grid, block, 0, 0, (const float4*)buf, sink, BYTES / 16);
        hipLaunchKernelGGL(stream_dma, grid, block, 0, 0, (const char*)buf, sink, BYTES / 1024);
        hipLaunchKernelGGL(rows128_dma, grid, block, 0, 0, (const char*)buf, sink, nrows);
        hipLaunchKernelGGL(rows1600_f4, grid, block, 0, 0, (const float4*)buf, (const int*)nullptr, sink, nrows);
        hipLaunchKernelGGL(rows1600_f4, grid, block, 0, 0, (const float4*)buf, (const int*)idx, sink, nrows);
        hipLaunchKernelGGL(store_f4, grid, block, 0, 0, (float4*)buf, BYTES / 16);
        hipDeviceSynchronize();
    }
    // known bytes per launch, in the order above (rows128_dma reads 1536 of every row's 1600 bytes)
    printf("known_bytes stream_f4 %zu stream_dma %zu rows128_dma %zu rows1600_f4 %zu rows1600_f4g %zu store_f4 %zu\n",
           BYTES, BYTES, (nrows / 8) * 8 * 1536, nrows * 1600, nrows * 1600, BYTES);
    return 0;
}
