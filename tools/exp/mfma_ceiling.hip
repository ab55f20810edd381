// Ceiling of the projection GEMM's inner loop on gfx950 (LAB micro-benchmark, round 5).
//
// The shipped kernel (gemm_bf16x6s_kernel<3, true>: fp16x3 operands, wave tile 32 x 240, four waves per workgroup, two workgroups per
// CU) executes, per wave and 32-deep K tile, 90 v_mfma_f32_16x16x32_f16 (2 row blocks x 15 column blocks x 3 products) fed by 30 B
// fragments (ds_read_b128 from the strip images) and 4 A fragments.  This program runs EXACTLY that MFMA / fragment-read stream from
// operands that are already resident — no global loads, no LDS-DMA, no operand split, no row lists — at K = 400 (13 K tiles), and
// prints the TFLOP/s of each variant.  It answers: how fast could this tiling go if everything around the matrix pipe were free?
//
//   v0  MFMAs only, fragments in registers (the pure issue ceiling of 90 MFMAs per K tile at the clock the chip holds on random data)
//   v1  + the 30 B-fragment reads per K tile from LDS (conflict-free ds_read_b128, as the shipped kernel lays its images out)
//   v2  + one s_barrier per strip (3 per K tile), the shipped kernel's cadence
//   v3  v2 + the epilogue: every wave tile's 32 x 240 fp32 results stored (float4 per lane, rows of 1200 floats)
//   v4  v1 with 64-row "fat" wave tiles (180 MFMAs per K tile, one wave per SIMD): B fragments reused across four row blocks
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_ceiling tools/exp/mfma_ceiling.hip ; run: tools/exp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ half8 rnd8(unsigned seed) {
    half8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)(((int)(hash32(seed * 8u + i) & 0xffff) - 32768) * (1.f / 32768.f));
    return v;
}

constexpr int KT = 13;            // K = 400: 12.5 K tiles of 32 (the shipped kernel runs 13, the last half masked by zero weights)
constexpr int NT = 5;             // 16-column blocks per 80-column strip
constexpr int NSUB = 3;           // strips per 240-column tile

template <int MT, int VAR, int FILL = 0>        // MT: 16-row blocks per wave; VAR: see above; FILL: extra vector instructions per K tile (v12)
__global__ void __launch_bounds__(256, MT == 2 ? 2 : 1) ceiling_kernel(float* out, int tiles_per_wg, long ldo) {
    __shared__ uint4 Bs[3][2 * 320];                 // ring of three strip images: 2 planes (hi, lo) x 4 k groups x 80 rows, 16 B slots
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int kg = lane >> 4, lr = lane & 15;
    for (int i = tid; i < 3 * 2 * 320; i += 256) (&Bs[0][0])[i] = __builtin_bit_cast(uint4, rnd8(i * 977u + blockIdx.x));
    __syncthreads();
    half8 af[2][MT];                                 // A fragments (hi, lo) of this wave's row blocks: resident, refreshed per K tile by a cheap xor
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[p][mt] = rnd8(lane * 131u + p * 7u + mt * 3u + wm);
    half8 breg[2] = {rnd8(lane + 1000u), rnd8(lane + 2000u)};
    float sink = 0.f;
    float fill[4] = {1.f + lane, 2.f, 3.f, 4.f};
    for (int tile = 0; tile < tiles_per_wg; ++tile) {
        v4f acc[NSUB][MT][NT];
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[s][mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                if (VAR == 2 || VAR == 3) __builtin_amdgcn_s_barrier();
                else __builtin_amdgcn_sched_barrier(0);          // no barrier: still keep the compiler from hoisting a K tile's 30 reads (it spills)
                const uint4* Bi = Bs[(kt * NSUB + s) % 3];
                const int lslot = kg * 80 + lr;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    half8 b1, b2;
                    if (VAR == 0) { b1 = breg[0]; b2 = breg[1]; }
                    else { b1 = __builtin_bit_cast(half8, Bi[lslot + nt * 16]); b2 = __builtin_bit_cast(half8, Bi[320 + lslot + nt * 16]); }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[s][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, af[1][mt], acc[s][mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[s][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2, af[0][mt], acc[s][mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[s][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, af[0][mt], acc[s][mt][nt], 0, 0, 0);
                    if (FILL) {
#pragma unroll
                        for (int f = 0; f < (FILL + 14) / 15; ++f) fill[f & 3] = __builtin_fmaf(fill[f & 3], 1.0001f, fill[(f + 1) & 3]);
                    }
                }
            }
        }
        if (VAR == 3) {
            // the shipped epilogue's store pattern: lane (kg, lr) holds columns 4 kg .. 4 kg + 3 of each 16-column block of row lr
            const long row0 = ((long)(blockIdx.x * tiles_per_wg + tile) * 4 + wm) * (16 * MT);
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const v4f a = acc[s][mt][nt];
                        *reinterpret_cast<float4*>(out + ((row0 + mt * 16 + lr) % (1 << 17)) * ldo + s * 80 + nt * 16 + kg * 4) = make_float4(a[0], a[1], a[2], a[3]);
                    }
        } else {
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) sink += acc[s][mt][nt][0] + acc[s][mt][nt][3];
        }
    }
    if (sink + fill[0] + fill[1] + fill[2] + fill[3] == 123.456f) out[0] = sink;            // never true: keeps the accumulators alive
}

template <int MT, int VAR, int FILL = 0>
static void run(const char* what, float* out, int wgs, int tiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((ceiling_kernel<MT, VAR, FILL>), dim3(wgs), dim3(256), 0, 0, out, tiles, 1200L);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((ceiling_kernel<MT, VAR, FILL>), dim3(wgs), dim3(256), 0, 0, out, tiles, 1200L);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    // per wave tile: 16 MT rows x 240 columns x K' = KT * 32, three products
    const double flops = (double)wgs * tiles * 4 * (16.0 * MT) * 240.0 * (KT * 32.0) * 2.0 * 3.0;
    printf("%-64s  %8.3f ms (best %8.3f)  %8.1f TFLOP/s executed (best %8.1f)\n", what, sum / reps, best, flops / (sum / reps * 1e-3) / 1e12,
           flops / (best * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

// v6/v7: the A operand RESIDENT in registers.  A wave owns 32 rows for all five 240-column tiles of the product: its A fragments of
// the whole K = 400 (13 K tiles x 2 row blocks x 2 pieces x 4 registers = 208 registers) are made once and the loop over
// (column tile, K tile, strip) streams nothing but B fragments — one wave per SIMD (four waves, one workgroup per CU).  In the real
// kernel this would take the fp32 A tile's DMA, its LDS round trip and the operand split out of the K loop (today every row is
// fetched and split five times, once per column tile).  VAR 6: B from LDS, barrier per strip; 7: + epilogue stores.
template <int VAR>
__global__ void __launch_bounds__(256, 1) resident_a_kernel(float* out, int rowtiles_per_wg, long ldo) {
    __shared__ uint4 Bs[3][2 * 320];
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int kg = lane >> 4, lr = lane & 15;
    for (int i = tid; i < 3 * 2 * 320; i += 256) (&Bs[0][0])[i] = __builtin_bit_cast(uint4, rnd8(i * 977u + blockIdx.x));
    __syncthreads();
    float sink = 0.f;
    for (int rt = 0; rt < rowtiles_per_wg; ++rt) {
        half8 af[KT][2][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) af[kt][p][mt] = rnd8(lane * 131u + p * 7u + mt * 3u + wm + kt * 1009u + rt);
        for (int ntile = 0; ntile < 5; ++ntile) {
            v4f acc[NSUB][2][NT];
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[s][mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int s = 0; s < NSUB; ++s) {
                    __builtin_amdgcn_s_barrier();
                    const uint4* Bi = Bs[(kt * NSUB + s) % 3];
                    const int lslot = kg * 80 + lr;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const half8 b1 = __builtin_bit_cast(half8, Bi[lslot + nt * 16]), b2 = __builtin_bit_cast(half8, Bi[320 + lslot + nt * 16]);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[s][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, af[kt][1][mt], acc[s][mt][nt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[s][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2, af[kt][0][mt], acc[s][mt][nt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[s][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, af[kt][0][mt], acc[s][mt][nt], 0, 0, 0);
                    }
                }
            }
            if (VAR == 7) {
                const long row0 = ((long)(blockIdx.x * rowtiles_per_wg + rt) * 4 + wm) * 32;
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const v4f a = acc[s][mt][nt];
                            *reinterpret_cast<float4*>(out + ((row0 + mt * 16 + lr) % (1 << 17)) * ldo + ntile * 240 + s * 80 + nt * 16 + kg * 4) =
                                make_float4(a[0], a[1], a[2], a[3]);
                        }
            } else {
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) sink += acc[s][mt][nt][0] + acc[s][mt][nt][3];
            }
        }
    }
    if (sink == 123.456f) out[0] = sink;
}

template <int VAR>
static void run_resident(const char* what, float* out, int wgs, int rowtiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((resident_a_kernel<VAR>), dim3(wgs), dim3(256), 0, 0, out, rowtiles, 1200L);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((resident_a_kernel<VAR>), dim3(wgs), dim3(256), 0, 0, out, rowtiles, 1200L);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flops = (double)wgs * rowtiles * 4 * 32.0 * 1200.0 * (KT * 32.0) * 2.0 * 3.0;
    printf("%-64s  %8.3f ms (best %8.3f)  %8.1f TFLOP/s executed (best %8.1f)\n", what, sum / reps, best, flops / (sum / reps * 1e-3) / 1e12,
           flops / (best * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}


// v8 .. v11 (round 6): the same product on v_mfma_f32_32x32x16_f16.  Wave tile 32 rows x 256 columns (eight
// 32-column blocks; [W|F1|F2] padded 1200 -> 1280 columns: five 256-column tiles), four waves per workgroup stacked along M, two
// workgroups per CU.  Per wave and 32-deep K tile: 48 MFMAs (8 blocks x 2 k steps x 3 products) of 32 cycles each — the same
// matrix-pipe time per flop as the 90 of 16 cycles — fed by 32 B-fragment ds_read_b128 (8 x 2 k steps x 2 pieces) and 4 A fragments;
// an MFMA holds the SIMD's vector issue for 8 of its 32 cycles instead of 8 of 16, which is the resource the shipped loop is short of.
//   v8  MFMAs only (fragments in registers)            v9  + B fragments from LDS, one s_barrier per four blocks (2 per K tile)
//   v10 v9 + epilogue stores (32 x 256 fp32)            v11 v10 + FILL vector instructions per K tile (a stand-in for the operand split)
// v12 = v3 (the shipped 16x16x32 tiling with barriers and stores) + the same filler, for comparison.
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int NB = 8;             // 32-column blocks per wave tile
template <int VAR, int FILL>
__global__ void __launch_bounds__(256, 2) ceiling32_kernel(float* out, int tiles_per_wg, long ldo) {
    __shared__ uint4 Bs[3][2 * 4 * 128];             // ring of three images of FOUR blocks: [plane][k group 0..3][128 rows], 16 B slots (16 KB)
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    for (int i = tid; i < 3 * 2 * 4 * 128; i += 256) (&Bs[0][0])[i] = __builtin_bit_cast(uint4, rnd8(i * 977u + blockIdx.x));
    __syncthreads();
    half8 af[2][2];                                  // A fragments [piece][k step]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[p][ks] = rnd8(lane * 131u + p * 7u + ks * 3u + wm);
    half8 breg[4][2];                                // v8: distinct fragments per block (identical blocks would be merged by the compiler)
#pragma unroll
    for (int b = 0; b < 4; ++b) { breg[b][0] = rnd8(lane + 1000u + b * 77u); breg[b][1] = rnd8(lane + 2000u + b * 77u); }
    float sink = 0.f;
    float fill[4] = {1.f + lane, 2.f, 3.f, 4.f};
    for (int tile = 0; tile < tiles_per_wg; ++tile) {
        v16f acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {            // a step = four blocks
                if (VAR >= 9) __builtin_amdgcn_s_barrier();
                else __builtin_amdgcn_sched_barrier(0);
                const uint4* Bi = Bs[(kt * 2 + s) % 3];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        half8 b1, b2;
                        if (VAR == 8) { b1 = breg[b][0]; b2 = breg[b][1]; }
                        else {
                            b1 = __builtin_bit_cast(half8, Bi[(0 * 4 + 2 * ks + h) * 128 + b * 32 + r]);
                            b2 = __builtin_bit_cast(half8, Bi[(1 * 4 + 2 * ks + h) * 128 + b * 32 + r]);
                        }
                        v16f c = acc[s * 4 + b];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, af[1][ks], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, af[0][ks], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, af[0][ks], c, 0, 0, 0);
                        acc[s * 4 + b] = c;
                        if (FILL) {
#pragma unroll
                            for (int f = 0; f < FILL / 16; ++f) fill[f & 3] = __builtin_fmaf(fill[f & 3], 1.0001f, fill[(f + 1) & 3]);
                        }
                    }
                }
            }
        }
        if (VAR >= 10) {
            // lane (h, r): row r of the wave tile; register e of a block = columns 8 (e >> 2) + 4 h + (e & 3): float4 stores
            const long row0 = ((long)(blockIdx.x * tiles_per_wg + tile) * 4 + wm) * 32;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v16f a = acc[b];
                    *reinterpret_cast<float4*>(out + ((row0 + r) % (1 << 17)) * ldo + b * 32 + 8 * q + 4 * h) =
                        make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
                }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) sink += acc[b][0] + acc[b][15];
        }
    }
    if (sink + fill[0] + fill[1] + fill[2] + fill[3] == 123.456f) out[0] = sink;
}

template <int VAR, int FILL>
static void run32(const char* what, float* out, int wgs, int tiles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((ceiling32_kernel<VAR, FILL>), dim3(wgs), dim3(256), 0, 0, out, tiles, 1280L);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((ceiling32_kernel<VAR, FILL>), dim3(wgs), dim3(256), 0, 0, out, tiles, 1280L);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    // per wave tile: 32 rows x 256 columns x K' = KT * 32, three products; "useful" discounts the 1200 -> 1280 column padding
    const double flops = (double)wgs * tiles * 4 * 32.0 * 256.0 * (KT * 32.0) * 2.0 * 3.0;
    printf("%-64s  %8.3f ms (best %8.3f)  %8.1f TFLOP/s executed (best %8.1f), %8.1f useful\n", what, sum / reps, best, flops / (sum / reps * 1e-3) / 1e12,
           flops / (best * 1e-3) / 1e12, flops * (1200.0 / 1280.0) / (sum / reps * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* out;
    hipMalloc(&out, (size_t)(1 << 17) * 1280 * 4);        // 629 MB: the 137 k-row launch's result rows
    hipMemset(out, 0, (size_t)(1 << 17) * 1280 * 4);
    // 137 k rows x 1200 columns = 1072 row tiles of 128 x 5 column tiles = 5360 workgroup tiles (the shipped launch); here 512 workgroups
    // (two per CU) x 10 tiles each = 5120 tiles of the same size
    const int wgs = 512, tiles = 10;
    run<2, 0>("v0 32x240 wave tiles, MFMAs only (fragments in registers)", out, wgs, tiles);
    run<2, 1>("v1 + B fragments from LDS (30 ds_read_b128 per K tile)", out, wgs, tiles);
    run<2, 2>("v2 + one s_barrier per strip (the shipped cadence)", out, wgs, tiles);
    run<2, 3>("v3 + epilogue stores (32 x 240 fp32 per wave tile)", out, wgs, tiles);
    run<4, 1>("v4 64x240 fat wave tiles, one wave per SIMD, B from LDS", out, 256, tiles);
    run<4, 3>("v5 fat wave tiles + barriers + epilogue stores", out, 256, tiles);
    // 256 workgroups (one per CU) x 4 row tiles of 128 rows x all 1200 columns = 131 k rows
    run_resident<6>("v6 A resident in registers (32 rows x K 400), 1 wave/SIMD, barriers", out, 256, 4);
    run_resident<7>("v7 v6 + epilogue stores", out, 256, 4);
    // round 6: the 32x32x16 form, 32 x 256 wave tiles (N padded to 1280): 512 workgroups x 10 tiles of 128 x 256
    run32<8, 0>("v8 32x32x16: 32x256 wave tiles, MFMAs only", out, wgs, tiles);
    run32<9, 0>("v9 + B fragments from LDS (32 ds_read_b128), barrier per 4 blocks", out, wgs, tiles);
    run32<10, 0>("v10 + epilogue stores (32 x 256 fp32 per wave tile)", out, wgs, tiles);
    run32<11, 96>("v11 v10 + 96 dependent-free v_fma per K tile (operand-split stand-in)", out, wgs, tiles);
    run32<11, 192>("v11b v10 + 192 v_fma per K tile", out, wgs, tiles);
    run<2, 3, 96>("v12 v3 (16x16x32) + ~96 v_fma per K tile", out, wgs, tiles);
    run<2, 3, 192>("v12b v3 (16x16x32) + ~192 v_fma per K tile", out, wgs, tiles);
    hipFree(out);
    return 0;
}
