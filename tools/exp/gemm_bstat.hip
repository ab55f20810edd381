// B-stationary projection GEMM, feasibility micro-benchmark (round 6; LAB).
//
// The shipped projection kernel streams BOTH operands through LDS-DMA rings with a barrier per 80-column strip and re-splits the fp32
// rows of A once per 240-column tile; its two waves per SIMD run in lockstep, so DMA issue, waits, operand split and MFMAs add up
// (DESIGN.md section 4).  This program measures the opposite decomposition on the same problem ([h|P|Q] = X [W|F1|F2]^T: M live rows,
// N = 1200, K = 400, fp16x3 = three v_mfma_f32_16x16x32_f16 products per fp32 product):
//   * a workgroup (8 waves, one per CU) keeps ONE 80-column strip of the split weights resident in LDS for the whole K (13 K tiles x
//     10 KB = 130 KB) and is persistent: it walks the row tiles of its group;
//   * the A rows arrive PRE-SPLIT ([8 fp16 hi | 8 fp16 lo] per 8 k: the layout the Eq. 8 kernels can store, GemmArgs.a_split) and go
//     global -> registers in MFMA fragment layout, one K tile ahead: no LDS write, no DMA, no operand split and NO BARRIER in the loop;
//   * the 15 strips of a row group sit on one XCD (blocks b, b + 8, ... share an XCD), so the 15-fold re-read of a row tile is
//     served by that XCD's L2.
// Wave tile 16 MT rows x 80 columns.  Prints the time of the 137 k-row launch and checks sampled outputs against a double reference.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/gemm_bstat tools/exp/gemm_bstat.hip ; run: tools/exp/gemm_bstat [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int K = 400, KT = 13, N = 1200, STRIPS = 15;
constexpr int IMG = 640;                         // 16-byte slots of one (strip, K tile) image: 2 planes x 4 k groups x 80 rows
constexpr int GROUPS = 16;                       // row groups: two per XCD (30 of an XCD's 32 CUs busy)

template <int MT>
__global__ void __launch_bounds__(512, 2) bstat_kernel(const unsigned char* __restrict__ As, const int* __restrict__ rowidx,
                                                       const uint4* __restrict__ wimg, float* __restrict__ out, int M) {
    extern __shared__ uint4 Bs[];                // [KT][IMG]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, lr = lane & 15;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;          // j = 0 .. 29
    const int group = xcd * 2 + j / STRIPS, strip = j % STRIPS;
    for (int i = tid; i < KT * IMG; i += 512) Bs[i] = wimg[(long)strip * KT * IMG + i];
    __syncthreads();
    constexpr int TR = 8 * 16 * MT;
    const int ntiles = (M + TR - 1) / TR;
    for (int tile = group; tile < ntiles; tile += GROUPS) {
        const unsigned char* ap[MT];
        int grow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int r = tile * TR + wave * 16 * MT + mt * 16 + lr;
            r = r < M ? r : M - 1;
            grow[mt] = rowidx ? rowidx[r] : r;
            ap[mt] = As + (long)grow[mt] * (K * 4) + kg * 32;
        }
        v4f acc[MT][5];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
        half8 a0[2][MT], a1[2][MT];              // [piece][row block], two K tiles
        auto load_a = [&](int kt, half8 (&dst)[2][MT]) {
            // the last K tile reaches past K = 400 (k groups 2, 3): those lanes read the row's start instead (finite; zero weights)
            const int off = (kt * 32 + kg * 8 < K) ? kt * 128 : -kg * 32;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                dst[0][mt] = *reinterpret_cast<const half8*>(ap[mt] + off);
                dst[1][mt] = *reinterpret_cast<const half8*>(ap[mt] + off + 16);
            }
        };
        auto step = [&](int kt, const half8 (&cur)[2][MT]) {
            const uint4* Bi = Bs + kt * IMG + kg * 80 + lr;
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                const half8 b1 = __builtin_bit_cast(half8, Bi[nt * 16]), b2 = __builtin_bit_cast(half8, Bi[320 + nt * 16]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, cur[1][mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2, cur[0][mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, cur[0][mt], acc[mt][nt], 0, 0, 0);
            }
        };
        load_a(0, a0);
#pragma nounroll
        for (int kt = 0; kt < KT - 1; kt += 2) {
            load_a(kt + 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            step(kt, a0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(kt + 2, a0);                  // kt + 2 <= KT - 1: KT is odd
            __builtin_amdgcn_sched_barrier(0);
            step(kt + 1, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
        step(KT - 1, a0);
        // lane (kg, lr): columns 4 kg .. 4 kg + 3 of each 16-column block, row lr of each row block
        constexpr float u = 1.f / (1024.f * 16.f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = tile * TR + wave * 16 * MT + mt * 16 + lr;
            if (r < M) {
                float* yrow = out + (long)grow[mt] * N + strip * 80 + kg * 4;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    const v4f a = acc[mt][nt];
                    *reinterpret_cast<float4*>(yrow + nt * 16) = make_float4(a[0] * u, a[1] * u, a[2] * u, a[3] * u);
                }
            }
        }
    }
}


// BS4: the same residency with FOUR waves (one per SIMD), 64-row wave tiles, and the pre-split A rows staged through wave-PRIVATE LDS
// buffers by LDS-DMA in full 128-byte lines (8 rows x 128 B per piece; a wave reads only what it requested itself, so its own vmcnt
// orders the reads: no workgroup barrier in the loop).  LDS: the strip trimmed to 12.5 K tiles (128 000 B) + 4 x 8 KB of staging.
__device__ __forceinline__ void lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
constexpr int BS4_B_SLOTS = 12 * IMG + 320;      // the last K tile holds k groups 0, 1 only: [plane][2][80]
__global__ void __launch_bounds__(256, 1) bs4_kernel(const unsigned char* __restrict__ As, const int* __restrict__ rowidx,
                                                     const uint4* __restrict__ wimg, float* __restrict__ out, int M, int dbg) {
    extern __shared__ uint4 lds_all[];           // staging [4 waves][64 rows][8 chunks] (in front: M0's LDS offset is 16 bits), then the strip
    uint4* const Bs = lds_all + 4 * 512;
    constexpr int MT = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, lr = lane & 15;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int group = xcd * 2 + j / STRIPS, strip = j % STRIPS;
    for (int i = tid; i < 12 * IMG; i += 256) Bs[i] = wimg[(long)strip * KT * IMG + i];
    for (int i = tid; i < 320; i += 256) {       // (plane, kq < 2, row)
        const int pl = i / 160, rem = i - pl * 160;
        Bs[12 * IMG + i] = wimg[(long)strip * KT * IMG + 12 * IMG + pl * 320 + rem];
    }
    __syncthreads();
    uint4* const stage = lds_all + wave * 512;
    const unsigned lds_stage = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)stage;
    constexpr int TR = 4 * 16 * MT;
    const int ntiles = (M + TR - 1) / TR;
    for (int tile = group; tile < ntiles; tile += GROUPS) {
        // DMA piece q (0..7): rows 8q .. 8q+7 of the wave tile; lane -> (row 8q + lane/8, chunk (lane&7) ^ swizzle(row))
        unsigned aoff[8];
        unsigned aclamp[8];                      // the last K tile's chunks 4..7 lie past K: read the row's first chunks instead
        const int row0 = tile * TR + wave * 16 * MT;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gr = row0 + r; gr = gr < M ? gr : M - 1;
            if (rowidx) gr = rowidx[gr];
            aoff[q] = (unsigned)gr * (K * 4) + c * 16;
            aclamp[q] = c >= 4 ? 12u * 128u : 0u;
        }
        int grow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { int r = row0 + mt * 16 + lr; r = r < M ? r : M - 1; grow[mt] = rowidx ? rowidx[r] : r; }
        v4f acc[MT][5];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
        auto issue = [&](int kt) {
            if (dbg & 1) return;
#pragma unroll
            for (int q = 0; q < 8; ++q) lds_dma16_s(As + kt * 128, aoff[q] - (kt == 12 ? aclamp[q] : 0u), lds_stage + q * 1024);
        };
        half8 af[2][MT];
        auto read_frags = [&]() {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 16 + lr, sw = (r >> 1) & 7;
                af[0][mt] = __builtin_bit_cast(half8, stage[r * 8 + ((kg * 2) ^ sw)]);
                af[1][mt] = __builtin_bit_cast(half8, stage[r * 8 + ((kg * 2 + 1) ^ sw)]);
            }
        };
        issue(0);
#pragma nounroll
        for (int kt = 0; kt < KT; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_frags();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < KT) issue(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            const bool tail = kt == 12;
            const uint4* Bi = tail ? Bs + 12 * IMG + (kg & 1) * 80 + lr : Bs + kt * IMG + kg * 80 + lr;
            const int p1 = tail ? 160 : 320;
            const bool live = !tail || kg < 2;
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                half8 b1 = __builtin_bit_cast(half8, Bi[nt * 16]), b2 = __builtin_bit_cast(half8, Bi[p1 + nt * 16]);
                if (!live) { b1 = (half8)(_Float16)0; b2 = (half8)(_Float16)0; }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, af[1][mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2, af[0][mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, af[0][mt], acc[mt][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr float u = 1.f / (1024.f * 16.f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = row0 + mt * 16 + lr;
            if (r < M && !(dbg & 2)) {
                float* yrow = out + (long)grow[mt] * N + strip * 80 + kg * 4;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    const v4f a = acc[mt][nt];
                    *reinterpret_cast<float4*>(yrow + nt * 16) = make_float4(a[0] * u, a[1] * u, a[2] * u, a[3] * u);
                }
            }
        }
    }
}

static unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static float rndf(unsigned s) { return ((int)(hash32(s) & 0xffffff) - 0x800000) * (1.f / 0x800000); }

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int M = argc > 1 ? atoi(argv[1]) : 137216;
    const int dbg = argc > 2 ? atoi(argv[2]) : 0;        // BS4 debugging: 1 = no DMA, 2 = no stores, 8 = BS4 only
    const int Mtot = 2 * M;                      // the live rows are every other row of the node buffer (a row list, as in the encoder)
    // host data: X [Mtot][K] in (-2, 2), W [N][K] in (-0.1, 0.1)
    std::vector<float> X((size_t)Mtot * K), W((size_t)N * K);
    for (size_t i = 0; i < X.size(); ++i) X[i] = 2.f * rndf((unsigned)i * 3u + 1u);
    for (size_t i = 0; i < W.size(); ++i) W[i] = 0.1f * rndf((unsigned)i * 7u + 5u);
    // split A rows: [8 hi | 8 lo] per 8 k, scaled by 16; weights: images [strip][kt][plane][kq][80][8], scaled by 1024, K padded to 416
    std::vector<_Float16> As((size_t)Mtot * K * 2), Wi((size_t)STRIPS * KT * IMG * 8);
    for (int m = 0; m < Mtot; ++m)
        for (int k = 0; k < K; ++k) {
            const float v = X[(size_t)m * K + k] * 16.f;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            const size_t base = (size_t)m * K * 2 + (size_t)(k >> 3) * 16 + (k & 7);
            As[base] = h; As[base + 8] = l;
        }
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < KT * 32; ++k) {
            const float v = k < K ? W[(size_t)n * K + k] * 1024.f : 0.f;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            const int strip = n / 80, r = n % 80, kt = k >> 5, kq = (k >> 3) & 3, e = k & 7;
            const size_t base = ((size_t)strip * KT + kt) * IMG * 8;
            Wi[base + ((0 * 4 + kq) * 80 + r) * 8 + e] = h;
            Wi[base + ((1 * 4 + kq) * 80 + r) * 8 + e] = l;
        }
    std::vector<int> rows(M);
    for (int i = 0; i < M; ++i) rows[i] = 2 * i + (hash32(i) & 1);
    unsigned char* dA; uint4* dW; float* dO; int* dR;
    hipMalloc(&dA, As.size() * 2); hipMalloc(&dW, Wi.size() * 2); hipMalloc(&dO, (size_t)Mtot * N * 4); hipMalloc(&dR, (size_t)M * 4);
    hipMemcpy(dA, As.data(), As.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, Wi.data(), Wi.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dR, rows.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    hipMemset(dO, 0, (size_t)Mtot * N * 4);
    const size_t lds = (size_t)KT * IMG * 16;
    hipFuncSetAttribute((const void*)bstat_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)bstat_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t lds4 = (size_t)(BS4_B_SLOTS + 4 * 512) * 16;
    hipFuncSetAttribute((const void*)bs4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
    auto bench = [&](int mt, const int* rl, const char* what) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
            if (mt == 44) hipLaunchKernelGGL(bs4_kernel, dim3(240), dim3(256), lds4, 0, dA, rl, dW, dO, M, dbg);
            else if (mt == 4) hipLaunchKernelGGL(bstat_kernel<4>, dim3(240), dim3(512), lds, 0, dA, rl, dW, dO, M);
            else hipLaunchKernelGGL(bstat_kernel<2>, dim3(240), dim3(512), lds, 0, dA, rl, dW, dO, M);
        };
        for (int w = 0; w < 3; ++w) launch();
        hipDeviceSynchronize();
        float best = 1e30f, sum = 0.f; const int reps = 10;
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        const double flops = 2.0 * M * N * (KT * 32.0) * 3.0;
        printf("%-58s %8.1f us (best %8.1f)  %7.1f TFLOP/s executed\n", what, sum / reps * 1e3, best * 1e3, flops / (sum / reps * 1e-3) / 1e12);
    };
    char name[128];
    if (!(dbg & 8)) {
    snprintf(name, sizeof name, "B-stationary, 64 x 80 wave tiles, %d listed rows", M); bench(4, dR, name);
    snprintf(name, sizeof name, "B-stationary, 32 x 80 wave tiles, %d listed rows", M); bench(2, dR, name);
    snprintf(name, sizeof name, "B-stationary, 64 x 80 wave tiles, %d contiguous rows", M); bench(4, nullptr, name);
    }
    snprintf(name, sizeof name, "BS4: 4 waves, A by wave-private LDS-DMA, %d contiguous", M); bench(44, nullptr, name);
    snprintf(name, sizeof name, "BS4: 4 waves, A by wave-private LDS-DMA, %d listed", M); bench(44, dR, name);
    // check (listed rows): BS4
    hipMemset(dO, 0, (size_t)Mtot * N * 4);
    hipLaunchKernelGGL(bs4_kernel, dim3(240), dim3(256), lds4, 0, dA, dR, dW, dO, M, dbg);
    hipDeviceSynchronize();
    std::vector<float> O((size_t)Mtot * N);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0; int bad = 0;
    for (int s = 0; s < 4000; ++s) {
        const int i = hash32(s * 11u + 3u) % M, n = hash32(s * 13u + 7u) % N, m = rows[i];
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)X[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
        const double err = fabs(ref - (double)O[(size_t)m * N + n]);
        worst = err > worst ? err : worst;
        if (err > 2e-5) ++bad;
    }
    // an unlisted row must stay zero
    int touched = 0;
    for (int i = 0; i < 1000; ++i) { const int m = 2 * i + 1 - (hash32(i) & 1); for (int n = 0; n < N; n += 97) touched += O[(size_t)m * N + n] != 0.f; }
    printf("check: worst |err| vs double %.3g over 4000 samples (%d over 2e-5), unlisted rows touched: %d\n", worst, bad, touched);
    return bad || touched ? 1 : 0;
}
