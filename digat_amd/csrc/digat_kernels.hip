// digat_kernels.hip — hand-written gfx950 (MI355X / CDNA4) kernels for DIGAT's dual-graph
// interaction hot path, and the C ABI declared in include/digat_hip.h.
//
// Kernels (DESIGN.md has the data layout and the roofline of each):
//   gemm_f32_kernel       nn.Linear on the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32),
//                         multi-segment N (W|ffn1|ffn2 in one pass over X), fused epilogues
//   xattn_fwd_kernel      Eq. 8: relu(K3+K1+K2).a -> leaky_relu -> -1e9 mask -> softmax_j
//                         -> relu(alpha @ h) + X, never materialising [B,n,n,d]
//   attn_pool_kernel      ScaledDotProductAttention with the key projection folded into the query
//   topic_pool_kernel     torch_scatter scatter_softmax + scatter_sum over history categories
//   build_user_nodes / row_logits   small glue kernels of DIGAT.inference / Model.inference
//
// gfx950 only: 64-wide wavefronts, 160 KiB LDS per CU, MFMA f32 16x16x4.  No CUDA shims.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/digat_hip.h"

typedef float v4f __attribute__((ext_vector_type(4)));

#define DIGAT_CHECK_LAUNCH()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return DIGAT_ERR_LAUNCH; \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- optional per-kernel event timing (bench.py's roofline leg) ---------------------------------
// Between digat_profile_start and digat_profile_stop every launch is bracketed by two hipEvents
// recorded on the stream the kernel is launched on; stop() synchronises once and sums elapsed time
// and the algorithmic work (flops for the MFMA kernels, bytes for the others) per kernel kind.
static struct {
    int enabled, cap, used;
    hipEvent_t* ev;
    int* kind;
    double* work;
    unsigned long long* rows_dev;      // [kinds] row-list GEMM launches add the rows they actually processed (device counters)
    double flops_per_row[16]; double rows_nominal[16];
} g_prof = {0, 0, 0, nullptr, nullptr, nullptr, nullptr, {0.0}, {0.0}};

struct ProfScope {
    hipStream_t st; int slot;
    ProfScope(int kind, double work, hipStream_t s) : st(s), slot(-1) {
        if (g_prof.enabled && g_prof.used < g_prof.cap) {
            slot = g_prof.used++;
            g_prof.kind[slot] = kind; g_prof.work[slot] = work;
            hipEventRecord(g_prof.ev[2 * slot], st);
        }
    }
    ~ProfScope() { if (slot >= 0) hipEventRecord(g_prof.ev[2 * slot + 1], st); }
};

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float f4_comp(const float4& v, int s) {
    return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// one 16-byte-per-lane global -> LDS copy; LDS address = lds_byte_addr (wave-uniform) + 16*lane.
// Invisible to hipcc's waitcnt bookkeeping: completion is counted by hand (wait_vmcnt below).
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void wait_vmcnt(int n) {      // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

// =================================================================================================
// 1. fp32 MFMA GEMM:  y_s[M, nseg] = A[M,K] @ w_s[nseg,K]^T (+ bias_s), s < nsegs, fused epilogue
// =================================================================================================
enum { EPI_NONE = 0, EPI_RELU_RES = 1, EPI_GATE = 2, EPI_ACCUM = 3 };   // ACCUM: y += result (backward sums)

struct GemmArgs {
    const float* a0; long lda0; int k0;      // columns [0,k0) of A come from a0 ...
    const float* a1; long lda1;              // ... columns [k0,K) from a1 (gate: [local ; global])
    const float* w[3]; const float* bias[3]; float* y[3]; long ldy;
    int nseg, nsegs, M, K, transW;           // transW: w_s stored [K, nseg] (y = A @ w)
    int epi;
    const float* e0; long lde0; const float* e1; long lde1; const float* e2; long lde2;
    int mtiles, ntiles;
    const unsigned short* wsplit;            // bf16x6 path: [3 planes][nsegs*nseg][K] bf16 of the weights
    const float* radd; int radd_seg, rows_per_b;   // segment radd_seg: y += radd[row / rows_per_b][col]  (K3 + K1 of Eq. 8)
    unsigned long long* exec_rows;                 // profiling only: += rows processed by a row-list launch
    const int* rowidx; const int* nrows_dev;       // bf16x6 kernel only: process rows rowidx[0 .. *nrows_dev) of A / y (live rows)
    int m_dispatch;                                // != 0: choose the kernel as if M were this (bit-identical results across batchings)
};

// LDS image: float4 tile[k4][row ^ k4]  (k4 = 4-float column group of the 32-deep K tile).
// A lane quarter q reads column group 4*kk+q with ds_read_b128 and feeds element s of it to the
// s-th MFMA k-step (A and B use the same k assignment, so the sum over k is complete); the XOR of
// the low row bits makes both that read and the staging ds_write_b128 bank-conflict free.
// TAG only names the instantiation (0 = nn.Linear, 1 = the Eq. 8 node projections) so that profiles
// list the projection launches under their own kernel symbol.
template <int BM, int BN, int WAVES_M, int WAVES_N, int PF, int TAG>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const GemmArgs g) {
    constexpr int BK4 = 8;
    constexpr int MT = BM / WAVES_M / 16;
    constexpr int NT = BN / WAVES_N / 16;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(BM % (WAVES_M * 16) == 0 && BN % (WAVES_N * 16) == 0, "wave tile");
    static_assert(BM % 8 == 0 && BN % 8 == 0, "xor swizzle needs 8-row groups");
    constexpr int A_PER_T = (BM * BK4 + 255) / 256;
    constexpr int B_PER_T = (BN * BK4 + 255) / 256;
    __shared__ float4 As[BK4 * BM];
    __shared__ float4 Bs[BK4 * BN];

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles; consecutive tiles share the same rows of A (its L2 keeps them).
    const int total = g.mtiles * g.ntiles;
    const int chunk = (total + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (tile >= total) return;
    const int mtile = tile / g.ntiles, ntile = tile - mtile * g.ntiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int Ntot = g.nseg * g.nsegs;
    // one weight segment per tile (launch_gemm guarantees tiles never straddle segments), chosen with
    // scalar arithmetic: a per-lane select of g.w[] makes hipcc fetch the pointer from kernarg memory
    // with a vector load and a vmcnt(0) in front of every tile load
    const int seg = n0 / g.nseg;
    const int nbase = n0 - seg * g.nseg;          // column of this tile inside its segment
    const float* const wseg = g.w[seg];
    const int ktiles = ((g.K >> 2) + BK4 - 1) / BK4;

    // PF register sets: PF = 1 overlaps one tile's loads with the MFMAs of the previous tile (enough when
    // a tile carries >= 2k MFMA cycles); PF = 2 keeps two tiles in flight for the small-M shapes whose
    // per-tile MFMA time is far below the load latency
    float4 ra0[A_PER_T], rb0[B_PER_T], ra1[PF == 2 ? A_PER_T : 1], rb1[PF == 2 ? B_PER_T : 1];

    auto load_tiles = [&](int kt, float4* ra, float4* rb) {
#pragma unroll
        for (int u = 0; u < A_PER_T; ++u) {
            const int i = tid + u * 256;
            float4 v = f4_zero();
            if (i < BM * BK4) {
                const int r = i / BK4, c4 = i % BK4;
                const int gm = m0 + r, k = (kt * BK4 + c4) * 4;
                if (gm < g.M && k < g.K) {
                    const float* p = (k < g.k0) ? g.a0 + (long)gm * g.lda0 + k
                                                : g.a1 + (long)gm * g.lda1 + (k - g.k0);
                    v = *reinterpret_cast<const float4*>(p);
                }
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < B_PER_T; ++u) {
            const int i = tid + u * 256;
            float4 v = f4_zero();
            if (i < BN * BK4) {
                const int r = g.transW ? i % BN : i / BK4;
                const int c4 = g.transW ? i / BN : i % BK4;
                const int gn = n0 + r, k = (kt * BK4 + c4) * 4;
                const int nn = nbase + r;
                if (nn < g.nseg && gn < Ntot && k < g.K) {
                    if (!g.transW) {
                        v = *reinterpret_cast<const float4*>(wseg + (long)nn * g.K + k);
                    } else {
                        const float* p = wseg + (long)k * g.nseg + nn;
                        v = make_float4(p[0], p[g.nseg], p[2 * (long)g.nseg], p[3 * (long)g.nseg]);
                    }
                }
            }
            rb[u] = v;
        }
    };
    auto store_tiles = [&](const float4* ra, const float4* rb) {
#pragma unroll
        for (int u = 0; u < A_PER_T; ++u) {
            const int i = tid + u * 256;
            if (i < BM * BK4) {
                const int r = i / BK4, c4 = i % BK4;
                As[c4 * BM + (r ^ c4)] = ra[u];
            }
        }
#pragma unroll
        for (int u = 0; u < B_PER_T; ++u) {
            const int i = tid + u * 256;
            if (i < BN * BK4) {
                const int r = g.transW ? i % BN : i / BK4;
                const int c4 = g.transW ? i / BN : i % BK4;
                Bs[c4 * BN + (r ^ c4)] = rb[u];
            }
        }
    };

    v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

    auto compute_tile = [&]() {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k4 = kk * 4 + (lane >> 4);
            float4 af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = wm * (MT * 16) + mt * 16 + (lane & 15);
                af[mt] = As[k4 * BM + (r ^ k4)];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int r = wn * (NT * 16) + nt * 16 + (lane & 15);
                bf[nt] = Bs[k4 * BN + (r ^ k4)];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            f4_comp(af[mt], s), f4_comp(bf[nt], s), acc[mt][nt], 0, 0, 0);
        }
    };

    if (PF == 1) {
        load_tiles(0, ra0, rb0);
        store_tiles(ra0, rb0);
        __syncthreads();
        for (int kt = 0; kt < ktiles; ++kt) {
            const bool more = kt + 1 < ktiles;
            if (more) load_tiles(kt + 1, ra0, rb0);          // global loads fly under this tile's MFMAs
            compute_tile();
            __syncthreads();
            if (more) {
                store_tiles(ra0, rb0);
                __syncthreads();
            }
        }
    } else {
        load_tiles(0, ra0, rb0);
        if (ktiles > 1) load_tiles(1, ra1, rb1);
        store_tiles(ra0, rb0);
        __syncthreads();
        for (int kt = 0; kt < ktiles; kt += 2) {
            // tile kt is in LDS, set 1 holds kt+1 (in flight), set 0 is free
            if (kt + 2 < ktiles) load_tiles(kt + 2, ra0, rb0);
            compute_tile();
            __syncthreads();
            if (kt + 1 >= ktiles) break;
            store_tiles(ra1, rb1);
            __syncthreads();
            // tile kt+1 is in LDS, set 0 holds kt+2 (in flight), set 1 is free
            if (kt + 3 < ktiles) load_tiles(kt + 3, ra1, rb1);
            compute_tile();
            __syncthreads();
            if (kt + 2 < ktiles) {
                store_tiles(ra0, rb0);
                __syncthreads();
            }
        }
    }

    // epilogue: D[row = 4*(lane>>4) + r][col = lane&15]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int nn = nbase + wn * (NT * 16) + nt * 16 + (lane & 15);
            if (nn >= g.nseg) continue;
            const float* bp = g.bias[seg];
            float* yp = g.y[seg];
            const float bv = bp ? bp[nn] : 0.f;
            const float* radd = seg == g.radd_seg ? g.radd : nullptr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * (MT * 16) + mt * 16 + (lane >> 4) * 4 + r;
                if (gm >= g.M) continue;
                float v = acc[mt][nt][r] + bv;
                if (radd) v = radd[(long)(gm / g.rows_per_b) * g.nseg + nn] + v;
                if (g.epi == EPI_RELU_RES) {
                    v = fmaxf(v, 0.f) + g.e0[(long)gm * g.lde0 + nn];
                } else if (g.epi == EPI_GATE) {
                    const float gate = 1.f / (1.f + expf(-v));
                    const float loc = g.e0[(long)gm * g.lde0 + nn];
                    const float glo = g.e1[(long)gm * g.lde1 + nn];
                    v = gate * loc + (1.f - gate) * glo;
                    if (g.e2) v = g.e2[(long)gm * g.lde2 + nn] + v;
                } else if (g.epi == EPI_ACCUM) {
                    v = yp[(long)gm * g.ldy + nn] + v;
                }
                yp[(long)gm * g.ldy + nn] = v;
            }
        }
    }
}

// =================================================================================================
// 1b. "bf16x6" GEMM: exact-fp32-quality product on the bf16 matrix cores
// =================================================================================================
// Every fp32 operand is split into three bf16 pieces (x = x1 + x2 + x3 exactly: 3 x 8 mantissa bits);
// the six partial products whose weight is >= 2^-16 (x1w1, x1w2, x2w1, x1w3, x2w2, x3w1) are summed in
// the fp32 MFMA accumulator (v_mfma_f32_16x16x32_bf16).  Dropped terms are <= 3 * 2^-24 relative, i.e.
// the result is as accurate as an fp32 fma chain (measured: mean error 0.6x that of an fp32 GEMM) at
// 6/16 of the fp32-MFMA cost.  Weights are split once per weight version (split_weights_tiled_kernel),
// activations while they pass through the kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Exact 3-way split by TRUNCATION: x1 = top 16 bits of v, r1 = v - x1 (exact, <= 16 significant bits),
// x2 = top 16 bits of r1, x3 = r1 - x2 (<= 8 significant bits, already a bf16).  Round-to-nearest is not
// needed for exactness and costs 2x the VALU work (measured: 217 vs ~110 VALU instructions per K tile).
struct Split3f { float a, b, c; };           // each value has its low 16 bits clear
__device__ __forceinline__ Split3f split3f(float v) {
    Split3f o;
    o.a = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
    const float r1 = v - o.a;
    o.b = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    o.c = r1 - o.b;
    return o;
}
// two bf16 (upper halves of lo and hi) packed into one dword: [hi16(lo) | hi16(hi) << 16], one v_perm_b32
__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
struct Split3 { unsigned short a, b, c; };
__device__ __forceinline__ Split3 split3(float v) {
    const Split3f f = split3f(v);
    Split3 o;
    o.a = (unsigned short)(__builtin_bit_cast(unsigned, f.a) >> 16);
    o.b = (unsigned short)(__builtin_bit_cast(unsigned, f.b) >> 16);
    o.c = (unsigned short)(__builtin_bit_cast(unsigned, f.c) >> 16);
    return o;
}

// Kernel design (what the counters asked for: a first version with 128x80 tiles and register-staged operands
// spent 3.8 non-MFMA vector instructions per MFMA — operand split, address arithmetic, bounds checks — while a
// 16-cycle MFMA leaves issue room for two; its dword epilogue cost another third of the run time):
//  * the four waves of a workgroup are stacked along M; a wave keeps its 32 rows of A for THREE 80-column
//    strips of the weights (wave tile 32 x 240, 120 accumulator registers), so the split is paid once per
//    180 MFMAs;
//  * the split weights are stored as ready-made LDS images, one per (80-column strip, 32-deep K tile):
//    [plane][k group][row] 16-byte slots, K zero-padded to 32.  An image is 15 KB and goes global -> LDS by
//    LDS-DMA (no registers, no address arithmetic, no bounds checks) into a ring of three: two images are in
//    flight behind the one the 60 MFMAs of a step read; the fp32 A tile (128 x 32) takes the same road once
//    per K tile and is split from LDS in the shadow of strips 1 and 2.  Every global read being a DMA, the
//    vector-memory queue is counted by hand: one "s_waitcnt vmcnt(4)" + one s_barrier per step;
//  * the weights are the MFMA's ROW operand and the activations its COLUMN operand, so a lane ends with four
//    consecutive output columns of one row: float4 bias/residual loads and float4 stores;
//  * tiles may span weight segments (bias/output picked per strip; segments are multiples of 80 columns).
// Measured (M=68 608, N=1 200, K=400): 373 us = 177 TFLOP/s fp32-equivalent = 1.06 PFLOP/s bf16; the 128x80
// version ran 467 us, the exact fp32-MFMA kernel (single segment) 770 us, rocBLAS fp32 addmm 680 us.
constexpr int WS_SLOTS = 960;            // 16-byte slots of one strip image: 3 planes x 4 k groups x 80 rows

__global__ void __launch_bounds__(256) split_weights_tiled_kernel(const float* w0, const float* w1, const float* w2,
                                                                  int nseg, int nsegs, int K, unsigned short* out) {
    const int KT = (K + 31) >> 5, Kp = KT * 32;
    const int Ntot = nseg * nsegs;
    const int strips = (Ntot + 79) / 80;
    const long total = (long)strips * 80 * Kp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
        float v = 0.f;
        if (n < Ntot && k < K) {
            const int seg = n / nseg;
            const float* w = seg == 0 ? w0 : (seg == 1 ? w1 : w2);
            v = w[(long)(n - seg * nseg) * K + k];
        }
        const Split3 sp = split3(v);
        const int strip = n / 80, r = n - strip * 80, kt = k >> 5, kq = (k >> 3) & 3, e = k & 7;
        const long base = ((long)strip * KT + kt) * WS_SLOTS * 8;
        out[base + ((0 * 4 + kq) * 80 + r) * 8 + e] = sp.a;
        out[base + ((1 * 4 + kq) * 80 + r) * 8 + e] = sp.b;
        out[base + ((2 * 4 + kq) * 80 + r) * 8 + e] = sp.c;
    }
}

template <int NSUB>
__global__ void __launch_bounds__(256, 2) gemm_bf16x6s_kernel(const GemmArgs g) {
    constexpr int MT = 2, NT = 5;
    constexpr int RING = 3;                  // strip images in LDS: two in flight behind the one being read
    constexpr int ABUF = NSUB == 1 ? 2 : 1;  // fp32 A tiles (128 rows x 32 k) in LDS
    __shared__ uint4 Bs[RING][WS_SLOTS];
    __shared__ uint4 As[ABUF][1024];         // slot r*8 + (c4 ^ ((r>>1)&7)): conflict-free 32-byte row pieces

    // With a row list (rowidx / *nrows_dev, written on the device) the grid is sized for all M rows and the tile
    // count comes from the live row count, so the live tiles still spread over the 8 XCDs evenly.
    const int Mv = g.nrows_dev ? *g.nrows_dev : g.M;
    if (g.exec_rows && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(g.exec_rows, (unsigned long long)Mv);
    const int total = ((Mv + 127) >> 7) * g.ntiles;
    const int chunk = (total + 7) >> 3;
    if ((int)(blockIdx.x >> 3) >= chunk) return;
    const int tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (tile >= total) return;
    const int mtile = tile / g.ntiles, ntile = tile - mtile * g.ntiles;

    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int kg = lane >> 4, lr = lane & 15;
    const int m0 = mtile * 128;
    const int strip0 = ntile * NSUB;
    const int KT = (g.K + 31) >> 5;
    const int nsteps = KT * NSUB;
    const unsigned ldsB = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&Bs[0][0];
    const unsigned ldsA = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&As[0][0];
    const char* const wimg = reinterpret_cast<const char*>(g.wsplit) + (long)lane * 16;
    const int wu = __builtin_amdgcn_readfirstlane(wm);          // the DMA's LDS address goes through M0: provably uniform

    // Every global read of the loop is an LDS-DMA, four wave-instructions per wave and image, so the vector-memory
    // queue holds whole images in issue order and "s_waitcnt vmcnt(4)" means "all but the youngest image".
    auto issue_b = [&](int step) {           // strip image of step (kt, s) -> Bs[step % RING]; 15 instructions + 1 repeat
        const int kt = step / NSUB, s = step - kt * NSUB;
        const int buf = step % RING;
        const char* src = wimg + ((long)(strip0 + s) * KT + kt) * (WS_SLOTS * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int q = wu + 4 * k;
            q = q < 15 ? q : 14;                                 // wave 3 copies piece 14 twice: four per wave, always
            lds_dma16(reinterpret_cast<const float*>(src + q * 1024), ldsB + (unsigned)((buf * WS_SLOTS + q * 64) * 16));
        }
    };
    const float* asrc[4];                    // this lane's source of A-tile piece q = wu + 4k at K tile 0
    int ac4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sl = (wm + 4 * k) * 64 + lane;
        const int r = sl >> 3, c4 = (sl & 7) ^ ((r >> 1) & 7);
        int gm = m0 + r;
        gm = gm < Mv ? gm : Mv - 1;                              // rows >= Mv are never stored
        if (g.rowidx) gm = g.rowidx[gm];
        asrc[k] = g.a0 + (long)gm * g.lda0 + c4 * 4;
        ac4[k] = c4 * 4;
    }
    auto issue_a = [&](int kt, int abuf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // K % 8 == 0, K >= 32; a float4 past K reads the row start instead (finite) and meets zero weights
            const int ko = kt * 32 + ac4[k] < g.K ? kt * 32 : 0;
            lds_dma16(asrc[k] + ko, ldsA + (unsigned)((abuf * 1024 + (wu + 4 * k) * 64) * 16));
        }
    };
    auto split_half = [&](const float4& v, unsigned (&o)[3][2]) {
        const Split3f s0 = split3f(v.x), s1 = split3f(v.y), s2 = split3f(v.z), s3 = split3f(v.w);
        o[0][0] = pack_hi16(s0.a, s1.a); o[0][1] = pack_hi16(s2.a, s3.a);
        o[1][0] = pack_hi16(s0.b, s1.b); o[1][1] = pack_hi16(s2.b, s3.b);
        o[2][0] = pack_hi16(s0.c, s1.c); o[2][1] = pack_hi16(s2.c, s3.c);
    };
    auto split_mt = [&](int mt, int abuf, bf16x8 (&dst)[3][MT]) {     // this lane's 8 k of row (mt, lr): LDS -> 3 bf16 fragments
        const int r = wm * 32 + mt * 16 + lr;
        const int sw = (r >> 1) & 7;
        const float4 v0 = __builtin_bit_cast(float4, As[abuf][r * 8 + ((kg * 2) ^ sw)]);
        const float4 v1 = __builtin_bit_cast(float4, As[abuf][r * 8 + ((kg * 2 + 1) ^ sw)]);
        unsigned lo[3][2], hi[3][2];
        split_half(v0, lo);
        split_half(v1, hi);
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p][mt] = __builtin_bit_cast(bf16x8, make_uint4(lo[p][0], lo[p][1], hi[p][0], hi[p][1]));
    };

    v4f acc[NSUB][MT][NT];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[s][mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

    bf16x8 af[3][MT], afn[3][MT];
    issue_a(0, 0);
    issue_b(0);
    if (nsteps > 1) issue_b(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    split_mt(0, 0, af);
    split_mt(1, 0, af);

    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int step = kt * NSUB + s;
            const int buf = step % RING;
            // queue, oldest first: image(step) | A tile requested a step ago | image(step+1): only the last may stay in flight
            if (step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // landed for every wave; image(step-1) and the old A tile are free
            if (NSUB == 1) {
                if (kt > 0) { split_mt(0, kt & 1, af); split_mt(1, kt & 1, af); }    // requested a step ago, into the other buffer
                if (more) issue_a(kt + 1, (kt + 1) & 1);
            } else if (s == 0 && more) {
                issue_a(kt + 1, 0);                                // read at strip 1, after the next barrier
            }
            if (step + 2 < nsteps) issue_b(step + 2);
            const uint4* Bi = Bs[buf];
            const int lslot = kg * 80 + lr;
            bf16x8 bq[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[0][p] = __builtin_bit_cast(bf16x8, Bi[p * 320 + lslot]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt + 1 < NT) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) bq[(nt + 1) & 1][p] = __builtin_bit_cast(bf16x8, Bi[p * 320 + lslot + (nt + 1) * 16]);
                }
                const bf16x8 b1 = bq[nt & 1][0], b2 = bq[nt & 1][1], b3 = bq[nt & 1][2];
                v4f c0 = acc[s][0][nt], c1 = acc[s][1][nt];
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, af[2][0], c0, 0, 0, 0);   // x3 w1
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, af[2][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, af[1][0], c0, 0, 0, 0);   // x2 w2
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, af[1][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b3, af[0][0], c0, 0, 0, 0);   // x1 w3
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b3, af[0][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, af[1][0], c0, 0, 0, 0);   // x2 w1
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, af[1][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, af[0][0], c0, 0, 0, 0);   // x1 w2
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, af[0][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, af[0][0], c0, 0, 0, 0);   // x1 w1
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, af[0][1], c1, 0, 0, 0);
                acc[s][0][nt] = c0; acc[s][1][nt] = c1;
                if (NSUB > 1 && more) {      // the next K tile's rows landed at this step's wait (requested at strip 0)
                    if (s == 1 && nt == 1) split_mt(0, 0, afn);
                    if (s == 1 && nt == 3) split_mt(1, 0, afn);
                }
            }
        }
        if (NSUB > 1 && more) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[p][mt] = afn[p][mt];
        }
    }

    // The weights are the MFMA's row operand, the activations its column operand, so a lane ends up with four
    // CONSECUTIVE output columns (4*kg .. 4*kg+3 of each 16-column block) of one row (lr): float4 bias / residual
    // loads and float4 stores, a quarter of the instructions of the row-major accumulator layout.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int gv = m0 + wm * 32 + mt * 16 + lr;
        if (gv >= Mv) continue;
        const int gm = g.rowidx ? g.rowidx[gv] : gv;
        const long rb = g.radd ? (long)(gm / g.rows_per_b) * g.nseg : 0;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int ncol0 = (strip0 + s) * 80;         // first column of this strip in the stacked output
            const int seg = ncol0 / g.nseg;
            const int nbase = ncol0 - seg * g.nseg + kg * 4;
            const float* bp = g.bias[seg];
            const float* rrow = (g.radd && seg == g.radd_seg) ? g.radd + rb : nullptr;
            const float* erow = g.epi == EPI_RELU_RES ? g.e0 + (long)gm * g.lde0 : nullptr;
            float* yrow = g.y[seg] + (long)gm * g.ldy;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nn = nbase + nt * 16;
                const v4f a = acc[s][mt][nt];
                float4 v = make_float4(a[0], a[1], a[2], a[3]);
                if (bp) v = f4_add(v, *reinterpret_cast<const float4*>(bp + nn));
                if (rrow) v = f4_add(*reinterpret_cast<const float4*>(rrow + nn), v);
                if (erow) {
                    const float4 x = *reinterpret_cast<const float4*>(erow + nn);
                    v = make_float4(fmaxf(v.x, 0.f) + x.x, fmaxf(v.y, 0.f) + x.y, fmaxf(v.z, 0.f) + x.z, fmaxf(v.w, 0.f) + x.w);
                }
                *reinterpret_cast<float4*>(yrow + nn) = v;
            }
        }
    }
}

// 1c. skinny linear for the [B,d] projections (M < 2048): a latency chain, not a throughput problem
// =================================================================================================
// 32x80 output tile per workgroup; the four waves split K (16-wide k blocks, wave w takes blocks
// w, w+4, ...) and load their MFMA operand fragments straight from global memory as float4 — lane
// (row l&15, quarter q) holds k = 16*kb + 4q .. +3 of its row and feeds element s to the s-th
// v_mfma_f32_16x16x4_f32 k-step, for A and B alike — with the next block's loads in flight under the
// current block's 40 MFMAs.  No LDS staging, no barrier in the K loop; one LDS reduction at the end.
// A may be two K-segments split at a multiple of 16 (the gate's [local ; global]); epilogues as gemm_f32.
__global__ void __launch_bounds__(256) gemm_skinny_kernel(const GemmArgs g) {
    __shared__ float red[4][40][64];
    const int tile = blockIdx.x;
    const int mtile = tile / g.ntiles, ntile = tile - mtile * g.ntiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int m0 = mtile * 32, n0 = ntile * 80;
    const int seg = n0 / g.nseg;
    const int nbase = n0 - seg * g.nseg;
    const float* const wseg = g.w[seg];
    const int nkb = (g.K + 15) >> 4;

    const float* arow[2][2];     // [mt][A segment]
    bool aok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int gm = m0 + mt * 16 + lr;
        aok[mt] = gm < g.M;
        arow[mt][0] = g.a0 + (long)(aok[mt] ? gm : 0) * g.lda0;
        arow[mt][1] = g.a1 ? g.a1 + (long)(aok[mt] ? gm : 0) * g.lda1 : arow[mt][0];
    }
    const float* brow[5];
    bool bok[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
        const int nn = nbase + nt * 16 + lr;
        bok[nt] = nn < g.nseg;
        brow[nt] = wseg + (long)(bok[nt] ? nn : 0) * g.K;
    }
    auto load_block = [&](int kb, float4* a, float4* b) {
        const int k = kb * 16 + 4 * lq;
        const bool kin = k < g.K;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float* p = k < g.k0 ? arow[mt][0] + k : arow[mt][1] + (k - g.k0);
            a[mt] = (kin && aok[mt]) ? *reinterpret_cast<const float4*>(p) : f4_zero();
        }
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
            b[nt] = (kin && bok[nt]) ? *reinterpret_cast<const float4*>(brow[nt] + k) : f4_zero();
    };

    v4f acc[2][5];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

    float4 a0[2], b0[5], a1[2], b1[5];
    int kb = wave;
    if (kb < nkb) load_block(kb, a0, b0);
    while (kb < nkb) {
        const int kn = kb + 4;
        if (kn < nkb) load_block(kn, a1, b1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_comp(a0[mt], s), f4_comp(b0[nt], s), acc[mt][nt], 0, 0, 0);
        kb = kn;
        if (kb >= nkb) break;
        const int kn2 = kb + 4;
        if (kn2 < nkb) load_block(kn2, a0, b0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_comp(a1[mt], s), f4_comp(b1[nt], s), acc[mt][nt], 0, 0, 0);
        kb = kn2;
    }
    // K-split reduction in wave order 0..3 (deterministic); wave w finalises accumulator registers [10w, 10w+10)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(mt * 5 + nt) * 4 + r][lane] = acc[mt][nt][r];
    __syncthreads();
    const float* bp = g.bias[seg];
    float* yp = g.y[seg];
    for (int q = 0; q < 10; ++q) {
        const int reg = wave * 10 + q;
        const int mt = reg / 20, nt = (reg / 4) % 5, r = reg & 3;
        const int gm = m0 + mt * 16 + lq * 4 + r;
        const int nn = nbase + nt * 16 + lr;
        if (gm >= g.M || nn >= g.nseg) continue;
        float v = ((red[0][reg][lane] + red[1][reg][lane]) + red[2][reg][lane]) + red[3][reg][lane];
        v += bp ? bp[nn] : 0.f;
        if (g.epi == EPI_RELU_RES) {
            v = fmaxf(v, 0.f) + g.e0[(long)gm * g.lde0 + nn];
        } else if (g.epi == EPI_GATE) {
            const float gate = 1.f / (1.f + expf(-v));
            const float loc = g.e0[(long)gm * g.lde0 + nn];
            const float glo = g.e1[(long)gm * g.lde1 + nn];
            v = gate * loc + (1.f - gate) * glo;
            if (g.e2) v = g.e2[(long)gm * g.lde2 + nn] + v;
        } else if (g.epi == EPI_ACCUM) {
            v = yp[(long)gm * g.ldy + nn] + v;
        }
        yp[(long)gm * g.ldy + nn] = v;
    }
}

// the strip-mined bf16x6 kernel serves this launch (the only kernel that takes a row list)
static bool gemm_is_bf16x6(const GemmArgs& g) {
    const int Md = g.m_dispatch > 0 ? g.m_dispatch : g.M;
    return g.wsplit && Md >= 2048 && g.nseg % 80 == 0 && g.K % 8 == 0 && g.K >= 32 && g.ldy % 4 == 0 && g.lde0 % 4 == 0 &&
           (g.epi == EPI_NONE || g.epi == EPI_RELU_RES) && g.k0 == g.K && !g.transW;
}

static int launch_gemm(GemmArgs g, hipStream_t st, int kind = DIGAT_KERNEL_LINEAR) {
    if (g.M <= 0) return DIGAT_OK;
    if (g.rowidx && !gemm_is_bf16x6(g)) return DIGAT_ERR_ARG;
    const int Ntot = g.nseg * g.nsegs;
    const int Md = g.m_dispatch > 0 ? g.m_dispatch : g.M;
    if (Md < 2048 && g.nseg % 80 == 0 && !g.transW && g.K % 4 == 0 && g.k0 % 16 == 0 && !g.radd) {
        ProfScope prof(kind, 2.0 * g.M * (double)Ntot * g.K, st);
        g.mtiles = (g.M + 31) / 32;
        g.ntiles = Ntot / 80;
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3(g.mtiles * g.ntiles), dim3(256), 0, st, g);
        DIGAT_CHECK_LAUNCH();
        return DIGAT_OK;
    }
    // tile configuration: 128x80 for the big projections; below 2048 rows 32x64 (most workgroups), or
    // 64x80 for multi-segment launches whose segments are multiples of 80 columns (d = 400); the small-M
    // shapes keep two K tiles in flight
    const int cfg = Md >= 2048 ? 0 : ((g.nsegs > 1 && g.nseg % 80 == 0) ? 1 : 2);
    const int bn = cfg == 2 ? 64 : 80;
    if (g.nsegs > 1 && g.nseg % bn != 0) {
        // a tile must lie inside one weight segment; when the tile width does not divide the segment
        // (only small test shapes), run the segments one launch each
        for (int sgm = 0; sgm < g.nsegs; ++sgm) {
            GemmArgs one = g;
            one.w[0] = g.w[sgm]; one.bias[0] = g.bias[sgm]; one.y[0] = g.y[sgm]; one.nsegs = 1;
            if (sgm == g.radd_seg) one.radd_seg = 0; else one.radd = nullptr;
            const int rc = launch_gemm(one, st, kind);
            if (rc) return rc;
        }
        return DIGAT_OK;
    }
    // a row-list launch does the work of its live rows only: the kernel adds that count to a device counter and
    // digat_profile_stop prices it; the launch itself is recorded with zero work
    const bool listed = g.rowidx != nullptr;
    if (listed && g_prof.enabled) {
        g.exec_rows = g_prof.rows_dev + kind;
        g_prof.flops_per_row[kind] = 2.0 * (double)Ntot * g.K; g_prof.rows_nominal[kind] += g.M;
    }
    ProfScope prof(kind, listed ? 0.0 : 2.0 * g.M * (double)Ntot * g.K, st);
    if (g.wsplit && cfg == 0 && g.nseg % 80 == 0 && g.K % 8 == 0 && g.K >= 32 && g.ldy % 4 == 0 && g.lde0 % 4 == 0 && (g.epi == EPI_NONE || g.epi == EPI_RELU_RES) &&
        g.k0 == g.K && !g.transW) {
        const int strips = Ntot / 80;
        g.mtiles = (g.M + 127) / 128;
        if (strips % 3 == 0) {           // 240-column tiles: the operand split is paid once per three strips
            g.ntiles = strips / 3;
            hipLaunchKernelGGL((gemm_bf16x6s_kernel<3>), dim3((unsigned)(((g.mtiles * g.ntiles + 7) / 8) * 8)), dim3(256), 0, st, g);
        } else {
            g.ntiles = strips;
            hipLaunchKernelGGL((gemm_bf16x6s_kernel<1>), dim3((unsigned)(((g.mtiles * g.ntiles + 7) / 8) * 8)), dim3(256), 0, st, g);
        }
        DIGAT_CHECK_LAUNCH();
        return DIGAT_OK;
    }
    const int bm = cfg == 0 ? 128 : (cfg == 1 ? 64 : 32);
    g.mtiles = (g.M + bm - 1) / bm;
    g.ntiles = (Ntot + bn - 1) / bn;
    const dim3 grid((unsigned)(((g.mtiles * g.ntiles + 7) / 8) * 8));
    if (cfg == 0 && kind == DIGAT_KERNEL_PROJ) hipLaunchKernelGGL((gemm_f32_kernel<128, 80, 4, 1, 1, 1>), grid, dim3(256), 0, st, g);
    else if (cfg == 0) hipLaunchKernelGGL((gemm_f32_kernel<128, 80, 4, 1, 1, 0>), grid, dim3(256), 0, st, g);
    else if (cfg == 1) hipLaunchKernelGGL((gemm_f32_kernel<64, 80, 4, 1, 2, 0>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<32, 64, 1, 4, 2, 0>), grid, dim3(256), 0, st, g);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

static GemmArgs gemm_plain(const float* x, long ldx, const float* w, const float* b, float* y, long ldy,
                           int M, int N, int K, int transW) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a0 = x; g.lda0 = ldx; g.k0 = K; g.a1 = nullptr; g.lda1 = 0;
    g.w[0] = w; g.bias[0] = b; g.y[0] = y; g.ldy = ldy;
    g.nseg = N; g.nsegs = 1; g.M = M; g.K = K; g.transW = transW; g.epi = EPI_NONE;
    return g;
}

// =================================================================================================
// 2. Eq. 8 (a1/a2 after the projections): score kernel (VALU) + aggregation kernel (fp32 MFMA)
// =================================================================================================
// Score kernel.  One thread owns a 4x4 tile of (centre i, neighbour j) pairs and runs the whole sum
// over the d channels for it in registers (16 accumulators), so the scores need no cross-lane
// reduction.  P' = K3 + K1 (written by the projection GEMM's epilogue, the reference's left-to-right
// order) and Q stream through LDS in channel chunks by LDS-DMA (global_load_lds_dwordx4: no register
// round trip), a ring of XA_RING chunk images with two chunks in flight behind hand-counted
// s_waitcnt vmcnt and ONE raw s_barrier per chunk.  The DMA destination is lane-linear, so the
// permutation lives in the per-lane SOURCE address: image slot [pos][c4], pos = (node%4)*NT + node/4;
// with an odd chunk width (5 float4 at d = 400) the 16 lanes of a ds_read_b128 group hit distinct
// 16-B slots.  Slots of padding positions re-read a real node (finite, never used).
// Tiles whose 16 adjacency bytes are all zero are never computed: a per-workgroup list of the
// non-empty tiles is built from the adjacency image and threads are dealt tiles from that list, so
// whole waves drop out on sparse graphs (masked scores are replaced by -1e9 whatever their value).
// To make empty tiles common, the nodes of every row are first ORDERED by their first neighbour
// (stable rank sort in LDS): same-category history items and their topic node share that key, so the
// adjacency becomes block-diagonal.  The order only changes which node a tile slot holds (DMA source
// address, adjacency lookup); alpha is written in the caller's node order and its values do not depend
// on the order (each score is one thread's sequential channel sum).
// The kernel is VALU-bound (PMC: SQ_ACTIVE_INST_VALU ~ 96 % of the kernel, 4 cycles per wave64
// instruction), so the inner loop uses a.relu(x) = (a.x + a.|x|)/2: per (i,j,c) one v_add (x = p + q) and
// one v_fma with the free |x| source modifier instead of add + max + fma; the separable linear part
// a.P'_j + a.Q_i is accumulated once per node by the first 8*NT*RB threads.
// Scores go to LDS, one wave per (row, centre) does the masked softmax with shuffles and writes
// alpha [B,n,n]; [B,n,n,d] is never materialised.
struct ScoreArgs {
    const float* P;    // P' = r + P  (K3 + K1)
    const float* Q; const float* a; const uint8_t* A; float* alpha;
    float* s_out;      // optional [B,n,n]: scores before leaky_relu / mask on the edges (training)
    const uint8_t* live;   // optional [B,n]: 0 = dead node (see user_live_flags_kernel): its alpha row is not computed
    const int* qgroup;     // optional [B]: row b's Q lives at Q[qgroup[b]] (rows of one impression share the centre-side projection)
    int B, n, d, d4;
    int NT, SN, CC4, nchunks, RB;
    int img_slots;     // float4 slots of one operand image of one chunk = RB * 4*NT * CC4
    int ring_slots;    // slots of one ring buffer: both operands, padded to a multiple of 64
    int ninstr;        // DMA wave-instructions per chunk = ring_slots / 64
    int am_off;        // byte offset of the adjacency BIT rows in LDS: [RB*n][NW] words, NW = ceil(n/32)
    int a_off;         // byte offset of a (d floats) in LDS
    int tl_off;        // byte offset of the non-empty-tile list (ints) + per-wave counters
    int ld_off;        // byte offset of the per-node linear terms a.P'_j, a.Q_i (2 * RB * 4*NT floats)
    int pm_off;        // byte offset of the node permutation: keys [RB*n] ints, then node_of [RB*n] ints
    int skip;          // ablation only (env DIGAT_XATTN_SKIP, 0 in production): 1 no score loop,
                       // 2 no aggregation launch, 8 no softmax, 16 no empty-tile skipping, 32 no score launch
};
constexpr int XA_RING = 3;   // chunk images in the LDS ring
constexpr int XA_KMAX = 4;   // DMA wave-instructions one wave issues per chunk (upper bound)


__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 8))) xattn_score_kernel(const ScoreArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
    const int b0 = blockIdx.x * g.RB;
    const int rows_here = min(g.RB, g.B - b0);
    const int n = g.n, NT = g.NT, CC4 = g.CC4, SN = g.SN, d4 = g.d4;

    float4* ring = reinterpret_cast<float4*>(smem);        // [XA_RING][ring_slots]: P' image, then Q image
    float* Ss = reinterpret_cast<float*>(smem);            // [RB][n][SN], aliases the ring after the chunk loop
    unsigned* Ab = reinterpret_cast<unsigned*>(smem + g.am_off);   // adjacency bit rows [RB*n][NW]
    const int NW = (n + 31) >> 5;
    float4* a_lds = reinterpret_cast<float4*>(smem + g.a_off);
    int* tl = reinterpret_cast<int*>(smem + g.tl_off);     // [RB*NT*NT] tile list, then [16] wave counts
    int* wcnt = tl + g.RB * NT * NT;
    float* lin = reinterpret_cast<float*>(smem + g.ld_off);   // [2][RB][4*NT]: a.P'_node, a.Q_node by image position
    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- phase 0: the adjacency bytes pass through the (still idle) ring area with wide loads and are
    // condensed to bit rows; the score vector a goes to LDS
    {
        // 16-byte loads from the enclosing aligned window
        const uint8_t* src = g.A + (long)b0 * n * n;
        const int mis = (int)(reinterpret_cast<uintptr_t>(src) & 15);
        const uint4* src16 = reinterpret_cast<const uint4*>(src - mis);
        const int nvec = (rows_here * n * n + mis + 15) >> 4;
        uint4* dst16 = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < nvec; i += nthreads) dst16[i] = src16[i];
        const float4* a4 = reinterpret_cast<const float4*>(g.a);
        for (int i = tid; i < d4; i += nthreads) a_lds[i] = a4[i];
        __syncthreads();
        const uint8_t* bytes = smem + mis;
        for (int e = tid; e < rows_here * n * NW; e += nthreads) {
            const int row = e / NW, w = e - row * NW;
            unsigned bits = 0;
            const uint8_t* br = bytes + row * n + w * 32;
            const int lim = min(32, n - w * 32);
            for (int k = 0; k < lim; ++k) bits |= (br[k] != 0 ? 1u : 0u) << k;
            if (g.live && !g.live[(long)b0 * n + row]) bits = 0;      // a dead node's row: no pair of it is ever needed
            Ab[e] = bits;
        }
    }
    __syncthreads();
    auto edge = [&](int rb, int i, int j) -> bool { return (Ab[(rb * n + i) * NW + (j >> 5)] >> (j & 31)) & 1u; };

    // ---- node order: stable sort of each row's nodes by their first neighbour (block-diagonalises the
    // category structure of user graphs; any order is legal)
    int* keys = reinterpret_cast<int*>(smem + g.pm_off);       // [RB*n]
    int* node_of = keys + g.RB * n;                            // [RB*n]: node held by sorted slot sig
    for (int e = tid; e < rows_here * n; e += nthreads) {
        const int rb = e / n, i = e - rb * n;
        int key = g.live ? n + i : i;        // rows without any bit (dead nodes, when a list is given) go last, together
        if (!(g.skip & 16)) {
            for (int w = 0; w < NW; ++w) {
                const unsigned bits = Ab[e * NW + w];
                if (bits) { key = w * 32 + __ffs(bits) - 1; break; }
            }
        }
        keys[e] = key;
    }
    __syncthreads();
    for (int e = tid; e < rows_here * n; e += nthreads) {
        const int rb = e / n, i = e - rb * n;
        const int key = keys[e];
        int rank = 0;
        for (int k = 0; k < n; ++k) {
            const int kk = keys[rb * n + k];
            rank += (kk < key) || (kk == key && k < i);
        }
        node_of[rb * n + rank] = i;
    }
    __syncthreads();

    // ---- this wave's DMA pieces: instruction q = wave + nwaves*k covers image slots [64q, 64q+64)
    const float* Pblk = g.P + (long)b0 * n * g.d;
    const float* Qblk = g.qgroup ? g.Q : g.Q + (long)b0 * n * g.d;      // grouped: offsets below are absolute rows of Q
    int src_off[XA_KMAX];        // float offset of this lane's source at chunk 0; bit 31: Q operand
    int kw = 0;                  // wave-uniform: instructions this wave issues per chunk
#pragma unroll
    for (int k = 0; k < XA_KMAX; ++k) {
        const int q = wave + nwaves * k;
        src_off[k] = 0;
        if (q < g.ninstr) {
            kw = k + 1;
            int slot = q * 64 + lane;
            if (slot >= 2 * g.img_slots) slot = 2 * g.img_slots - 1;           // padding lanes re-read the last slot
            const int op = slot >= g.img_slots;
            const int e = slot - op * g.img_slots;
            const int per_row = 4 * NT * CC4;
            int rb = e / per_row;
            const int rem = e - rb * per_row;
            const int pos = rem / CC4, c4 = rem - pos * CC4;
            int sig = (pos % NT) * 4 + pos / NT;                               // sorted slot held by this image position
            if (sig >= n) sig = n - 1;                                         // padding positions: any real node
            if (rb >= rows_here) rb = rows_here - 1;
            int node = node_of[rb * n + sig];
            if (g.live && !g.live[(long)(b0 + rb) * n + node]) node = node_of[rb * n];      // dead: re-read the first (live) node, never used
            const int brow = (op && g.qgroup) ? g.qgroup[b0 + rb] : rb;
            src_off[k] = ((brow * n + node) * g.d + c4 * 4) | (op << 31);
        }
    }
    auto issue = [&](int ch, int buf) {
#pragma unroll
        for (int k = 0; k < XA_KMAX; ++k) {
            if (k < kw) {
                const int q = wave + nwaves * k;
                const float* base = (src_off[k] < 0) ? Qblk : Pblk;
                lds_dma16(base + (src_off[k] & 0x7fffffff) + ch * CC4 * 4,
                          ring_lds + (unsigned)((buf * g.ring_slots + q * 64) * 16));
            }
        }
    };
    issue(0, 0);
    if (g.nchunks > 1) issue(1, 1);

    // ---- non-empty tiles -> compact list (order = tile id, deterministic)
    const int tiles = NT * NT;
    int my_tile = -1;
    {
        const int t = tid;
        bool flag = false;
        if (t < rows_here * tiles) {
            const int rb = t / tiles, tt = t - rb * tiles;
            const int ti = tt / NT, tj = tt - ti * NT;
            if (g.skip & 16) {
                flag = true;
            } else {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int si = 4 * ti + ii;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int sj = 4 * tj + jj;
                        if (si < n && sj < n && edge(rb, node_of[rb * n + si], node_of[rb * n + sj])) flag = true;
                    }
                }
            }
        }
        const unsigned long long mask = __ballot(flag);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < nwaves; ++w) {
            const int c = wcnt[w];
            if (w < wave) base += c;
            total += c;
        }
        if (flag) tl[base + before] = t;
        __syncthreads();
        if (tid < total) my_tile = tl[tid];
    }
    const bool active = my_tile >= 0;
    const int rb_t = active ? my_tile / tiles : 0;
    const int tt = active ? my_tile - rb_t * tiles : 0;
    const int ti = tt / NT, tj = tt - ti * NT;

    // ---- phase 1: scores
    float acc[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.f;

    const int lin_rows = 2 * g.RB * 4 * NT;       // image rows: P' rows of every row-block, then Q rows
    float ldot = 0.f;                               // thread tid < lin_rows: a . (image row tid)
    for (int ch = 0; ch < g.nchunks; ++ch) {
        // chunk ch has landed once at most the newer chunk's pieces of THIS wave are outstanding ...
        wait_vmcnt(ch + 1 < g.nchunks ? kw : 0);
        // ... and every wave has said so: the barrier publishes chunk ch and retires chunk ch-1's readers
        __builtin_amdgcn_s_barrier();
        if (ch + 2 < g.nchunks) issue(ch + 2, (ch + 2) % XA_RING);      // into the image read in iteration ch-1
        if (active && !(g.skip & 1)) {
            const float4* Pb = ring + (ch % XA_RING) * g.ring_slots + rb_t * 4 * NT * CC4;
            const float4* Qb = Pb + g.img_slots;
            const float4* av = a_lds + ch * CC4;
            for (int c4 = 0; c4 < CC4; ++c4) {
                const float4 a4 = av[c4];                     // same address in every lane: LDS broadcast
                float4 p[4], q[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) p[jj] = Pb[(jj * NT + tj) * CC4 + c4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) q[ii] = Qb[(ii * NT + ti) * CC4 + c4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        float sc = acc[ii][jj];          // accumulates sum_c a_c |x_c|
                        sc = fmaf(a4.x, fabsf(p[jj].x + q[ii].x), sc);
                        sc = fmaf(a4.y, fabsf(p[jj].y + q[ii].y), sc);
                        sc = fmaf(a4.z, fabsf(p[jj].z + q[ii].z), sc);
                        sc = fmaf(a4.w, fabsf(p[jj].w + q[ii].w), sc);
                        acc[ii][jj] = sc;
                    }
            }
        }
        if (tid < lin_rows) {   // the separable part sum_c a_c (P'_jc + Q_ic), once per node
            const float4* row = ring + (ch % XA_RING) * g.ring_slots + tid * CC4;      // P' rows then Q rows: contiguous
            const float4* av = a_lds + ch * CC4;
            for (int c4 = 0; c4 < CC4; ++c4) {
                const float4 v = row[c4], a4 = av[c4];
                ldot = fmaf(a4.w, v.w, fmaf(a4.z, v.z, fmaf(a4.y, v.y, fmaf(a4.x, v.x, ldot))));
            }
        }
    }
    if (tid < lin_rows) lin[tid] = ldot;
    __syncthreads();            // every wave is done with the ring: Ss may alias it

    // ---- phase 2a: every score starts masked (-1e9, not -inf)
    for (int i = tid; i < rows_here * n * SN; i += nthreads) Ss[i] = -1e9f;
    __syncthreads();
    // leaky_relu(0.2) of the computed scores where the adjacency has an edge
    if (active) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            if (4 * ti + ii >= n) continue;
            const int i = node_of[rb_t * n + 4 * ti + ii];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (4 * tj + jj >= n) continue;
                const int j = node_of[rb_t * n + 4 * tj + jj];
                if (edge(rb_t, i, j)) {
                    // a.relu(x) = (a.x + a.|x|) / 2
                    const float e = 0.5f * (acc[ii][jj] + (lin[rb_t * 4 * NT + jj * NT + tj]
                                                          + lin[(g.RB + rb_t) * 4 * NT + ii * NT + ti]));
                    if (g.s_out) g.s_out[((long)(b0 + rb_t) * n + i) * n + j] = e;
                    Ss[(rb_t * n + i) * SN + j] = e > 0.f ? e : 0.2f * e;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2b: softmax over the neighbours j, one wave per (row, centre i); a row without any
    // edge is all -1e9 and comes out uniform, as in the reference
    for (int rho = wave; rho < ((g.skip & 8) ? 0 : rows_here * n); rho += nwaves) {
        if (g.live && !g.live[(long)b0 * n + rho]) continue;           // dead centre: its alpha row keeps its (finite) old values
        const float* srow = Ss + (long)rho * SN;
        const float v0 = lane < n ? srow[lane] : -INFINITY;
        const float v1 = lane + 64 < n ? srow[lane + 64] : -INFINITY;
        const float m = wave_max(fmaxf(v0, v1));
        const float e0 = lane < n ? expf(v0 - m) : 0.f;
        const float e1 = lane + 64 < n ? expf(v1 - m) : 0.f;
        const float inv = wave_sum(e0 + e1);
        float* arow = g.alpha + ((long)b0 * n + rho) * n;
        if (lane < n) arow[lane] = e0 / inv;
        if (lane + 64 < n) arow[lane + 64] = e1 / inv;
    }
}

// Score kernel for SMALL graphs (n <= 16: the news graph, N = 10 by default).  The tile kernel above would
// keep 72 of 256 threads busy there and pay a barrier + DMA round trip per 20-channel chunk; with so little
// arithmetic the job is a latency problem.  Here one WAVE owns one (row, centre i): the lanes span the channels
// (float4 each, coalesced), Q_i and a stay in registers, every neighbour's P'_j streams through once, the
// per-pair channel sum is a wave reduction, lane j keeps score j, and the masked softmax is the same
// shuffle code as above.  No LDS, no barrier; B*n waves (10 240 for the default batch).
template <int U>   // float4 pieces per lane: d <= 256 * U
__global__ void __launch_bounds__(256) xattn_score_small_kernel(const ScoreArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rho = (long)blockIdx.x * 4 + wave;
    const int n = g.n, d4 = g.d4;
    if (rho >= (long)g.B * n) return;
    const long b = rho / n;
    const float4* a4 = reinterpret_cast<const float4*>(g.a);
    const float4* Qi = reinterpret_cast<const float4*>(g.Q) + (g.qgroup ? (long)g.qgroup[b] * n + (rho - b * n) : rho) * d4;
    const float4* Pb = reinterpret_cast<const float4*>(g.P) + b * n * d4;
    float4 q[U], av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c4 = lane + 64 * u;
        q[u] = c4 < d4 ? Qi[c4] : f4_zero();
        av[u] = c4 < d4 ? a4[c4] : f4_zero();          // a = 0 on the padding channels: they add nothing
    }
    const bool edge = lane < n && g.A[rho * n + lane] != 0;
    float mine = 0.f;
    auto pair_sum = [&](const float4 (&p)[U]) -> float {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            part = fmaf(av[u].x, fmaxf(p[u].x + q[u].x, 0.f), part);
            part = fmaf(av[u].y, fmaxf(p[u].y + q[u].y, 0.f), part);
            part = fmaf(av[u].z, fmaxf(p[u].z + q[u].z, 0.f), part);
            part = fmaf(av[u].w, fmaxf(p[u].w + q[u].w, 0.f), part);
        }
        return wave_sum(part);
    };
    int j = 0;
    for (; j + 2 <= n; j += 2) {                        // two neighbours' rows in flight
        float4 p0[U], p1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c4 = lane + 64 * u;
            p0[u] = c4 < d4 ? Pb[(long)j * d4 + c4] : f4_zero();
            p1[u] = c4 < d4 ? Pb[(long)(j + 1) * d4 + c4] : f4_zero();
        }
        const float s0 = pair_sum(p0), s1 = pair_sum(p1);
        if (lane == j) mine = s0;
        if (lane == j + 1) mine = s1;
    }
    if (j < n) {
        float4 p0[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c4 = lane + 64 * u;
            p0[u] = c4 < d4 ? Pb[(long)j * d4 + c4] : f4_zero();
        }
        const float s0 = pair_sum(p0);
        if (lane == j) mine = s0;
    }
    if (edge && g.s_out) g.s_out[rho * n + lane] = mine;
    const float lk = mine > 0.f ? mine : 0.2f * mine;
    const float v0 = lane < n ? (edge ? lk : -1e9f) : -INFINITY;     // masked: -1e9, not -inf (a row without edges -> uniform)
    const float m = wave_max(v0);
    const float e0 = lane < n ? expf(v0 - m) : 0.f;
    const float inv = wave_sum(e0);
    if (lane < n) g.alpha[rho * n + lane] = e0 / inv;
}

// Aggregation kernel: out[b] = relu(alpha[b] @ h[b]) + X[b] on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32 is an exact k-ordered fma chain, so this equals a sequential sum over the
// neighbours j).  One workgroup owns one row b: alpha[b] (n*n floats) is staged in LDS once, each wave
// owns 64 channels: MFMA column l&15 of channel tile s stands for channel c0 + 4*(l&15) + s, so a lane
// loads h as one float4 and stores its 4 results as one float4.  h rows are requested AG_PF neighbour
// steps ahead; alpha blocks (16 centres x 4 neighbours) that are entirely zero — masked pairs — skip
// their MFMAs.
struct AggArgs {
    const float* alpha; const float* Hh; const float* X; float* out; int B, n, d, groups, sa;
    const uint8_t* live;     // optional [B,n]: rows of dead nodes (see user_live_flags_kernel) are neither read nor written
    const int* hgroup;       // optional [B]: row b's h lives at Hh[hgroup[b]]
};
constexpr int AG_IT = 5;     // 16-row centre tiles per pass (80 centres)
constexpr int AG_PF = 3;     // neighbour steps of h in flight

__global__ void __launch_bounds__(1024) xattn_agg_kernel(const AggArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* As = reinterpret_cast<float*>(smem);            // [n][sa], sa odd: conflict-free column reads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int n = g.n, d = g.d, sa = g.sa;
    unsigned char* lv = smem + (size_t)n * sa * 4;         // [n] live flags of this row's nodes (all 1 without a list)
    for (int i = tid; i < n; i += blockDim.x) lv[i] = g.live ? g.live[(long)b * n + i] : 1;
    __syncthreads();
    {
        // a dead node's alpha row is never used (its output is not written) and its column is zero for every live
        // centre (it has no edge to them): zero rows in the image let whole MFMA blocks drop out
        const float* Ab = g.alpha + (long)b * n * n;
        for (int e = tid; e < n * n; e += blockDim.x) {
            const int i = e / n, j = e - i * n;
            As[i * sa + j] = lv[i] ? Ab[e] : 0.f;
        }
    }
    __syncthreads();
    if (wave >= g.groups) return;
    const int lr = lane & 15, lq = lane >> 4;
    const int ch = wave * 64 + 4 * lr;
    const bool ch_ok = ch < d;
    const float* Hb = g.Hh + (long)(g.hgroup ? g.hgroup[b] : b) * n * d;
    const float* Xb = g.X + (long)b * n * d;
    float* Ob = g.out + (long)b * n * d;
    const int nit = (n + 15) >> 4;
    const int nsteps = (n + 3) >> 2;

    auto load_h = [&](int step) -> float4 {
        const int j = step * 4 + lq;
        return (j < n && ch_ok && lv[j]) ? *reinterpret_cast<const float4*>(Hb + (long)j * d + ch) : f4_zero();
    };

    for (int it0 = 0; it0 < nit; it0 += AG_IT) {
        v4f acc[AG_IT][4];
#pragma unroll
        for (int it = 0; it < AG_IT; ++it)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[it][s] = (v4f){0.f, 0.f, 0.f, 0.f};

        float4 hq[AG_PF];
#pragma unroll
        for (int u = 0; u < AG_PF; ++u) hq[u] = load_h(u);
        for (int step = 0; step < nsteps; step += AG_PF) {
#pragma unroll
            for (int u = 0; u < AG_PF; ++u) {
                const int st = step + u;
                if (st < nsteps) {
                    const float4 hc = hq[u];
                    hq[u] = load_h(st + AG_PF);
                    const int j = st * 4 + lq;
#pragma unroll
                    for (int it = 0; it < AG_IT; ++it) {
                        const int i = (it0 + it) * 16 + lr;
                        const float av = (i < n && j < n) ? As[i * sa + j] : 0.f;
                        if (it0 + it < nit && __any(av != 0.f)) {
                            acc[it][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hc.x, acc[it][0], 0, 0, 0);
                            acc[it][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hc.y, acc[it][1], 0, 0, 0);
                            acc[it][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hc.z, acc[it][2], 0, 0, 0);
                            acc[it][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, hc.w, acc[it][3], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (ch_ok) {
#pragma unroll
            for (int it = 0; it < AG_IT; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = (it0 + it) * 16 + 4 * lq + r;
                    if (i < n && lv[i]) {
                        const float4 x = *reinterpret_cast<const float4*>(Xb + (long)i * d + ch);
                        *reinterpret_cast<float4*>(Ob + (long)i * d + ch) =
                            make_float4(fmaxf(acc[it][0][r], 0.f) + x.x, fmaxf(acc[it][1][r], 0.f) + x.y,
                                        fmaxf(acc[it][2][r], 0.f) + x.z, fmaxf(acc[it][3][r], 0.f) + x.w);
                    }
                }
            }
        }
    }
}

struct XattnPlan { ScoreArgs g; int threads; size_t lds; int blocks; };

static int plan_xattn(int B, int n, int d, XattnPlan* pl) {
    if (B < 0 || n <= 0 || d <= 0) return DIGAT_ERR_ARG;
    if (d % 4 != 0 || n > DIGAT_MAX_NODES) return DIGAT_ERR_SHAPE;
    ScoreArgs& g = pl->g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.n = n; g.d = d; g.d4 = d / 4;
    g.NT = (n + 3) / 4;
    g.SN = (n + 3) / 4 * 4;
    const int tiles = g.NT * g.NT;
    int threads, rbmax;
    if (tiles >= 128) { threads = (tiles + 63) / 64 * 64; rbmax = 1; }
    else { threads = 256; rbmax = 256 / tiles; }
    if (rbmax > B && B > 0) rbmax = B;
    const int nwaves = threads / 64;
    int bestRB = 0, bestCC = 0;
    // <= 40 KiB keeps 4 workgroups per CU (1024 rows = one round on 256 CUs); graphs too large for
    // that may take up to 150 KiB.  The DMA ring must sit in the first 64 KiB of LDS (M0 addressing).
    for (int pass = 0; pass < 3 && !bestRB; ++pass) {
        const size_t lds_budget = pass == 0 ? 40 * 1024 : (pass == 1 ? 64 * 1024 : 150 * 1024);
        for (int rb = rbmax; rb >= 1 && !bestRB; --rb) {
            int cc_ok = 0;
            // odd chunk widths first (conflict-free LDS reads), widest first
            for (int odd = 1; odd >= 0 && !cc_ok; --odd) {
                for (int cc = g.d4; cc >= 1; --cc) {
                    if (g.d4 % cc || (cc & 1) != odd) continue;
                    const long img = (long)rb * 4 * g.NT * cc;
                    const long ring_slots = (2 * img + 63) / 64 * 64;
                    if (ring_slots / 64 > (long)XA_KMAX * nwaves) continue;
                    if ((size_t)XA_RING * ring_slots * 16 > 60 * 1024) continue;
                    const size_t sc = (size_t)rb * n * g.SN * 4;
                    const size_t ringb = (size_t)XA_RING * ring_slots * 16;
                    if ((size_t)rb * n * n + 32 > ringb) continue;          // the byte image passes through the ring area
                    const size_t tot = align_up(ringb > sc ? ringb : sc, 16) + (size_t)rb * n * ((n + 31) / 32) * 4
                                       + (size_t)d * 4 + ((size_t)rb * tiles + 16) * 4 + (size_t)rb * 8 * g.NT * 4
                                       + (size_t)2 * rb * n * 4;
                    if (2 * rb * 4 * g.NT > threads) continue;           // one thread per image row for the linear terms
                    if (tot > lds_budget) continue;
                    cc_ok = cc;
                    break;
                }
            }
            const int want = g.d4 < 5 ? 1 : 5;
            if (cc_ok >= want || (rb == 1 && cc_ok >= 1)) { bestRB = rb; bestCC = cc_ok; }
        }
    }
    if (!bestRB) return DIGAT_ERR_SHAPE;
    g.RB = bestRB; g.CC4 = bestCC; g.nchunks = g.d4 / g.CC4;
    g.img_slots = g.RB * 4 * g.NT * g.CC4;
    g.ring_slots = (2 * g.img_slots + 63) / 64 * 64;
    g.ninstr = g.ring_slots / 64;
    const size_t ringb = (size_t)XA_RING * g.ring_slots * 16;
    const size_t sc = (size_t)g.RB * n * g.SN * 4;
    g.am_off = (int)align_up(ringb > sc ? ringb : sc, 16);
    g.a_off = g.am_off + (int)align_up((size_t)g.RB * n * ((n + 31) / 32) * 4, 16);
    g.tl_off = g.a_off + d * 4;
    g.ld_off = g.tl_off + (g.RB * tiles + 16) * 4;
    g.pm_off = g.ld_off + g.RB * 8 * g.NT * 4;
    pl->lds = g.pm_off + (size_t)2 * g.RB * n * 4;
    pl->threads = threads;
    pl->blocks = (B + g.RB - 1) / g.RB;
    return DIGAT_OK;
}

// the score launch of a filled-in plan: the wave-per-centre kernel for small graphs, the tile kernel otherwise
static int launch_score(const XattnPlan& pl, hipStream_t st) {
    const ScoreArgs& g = pl.g;
    if (g.n <= 16 && g.d4 <= 256 && !g.skip) {
        const unsigned blocks = (unsigned)(((long)g.B * g.n + 3) / 4);
        if (g.d4 <= 64) hipLaunchKernelGGL(xattn_score_small_kernel<1>, dim3(blocks), dim3(256), 0, st, g);
        else if (g.d4 <= 128) hipLaunchKernelGGL(xattn_score_small_kernel<2>, dim3(blocks), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(xattn_score_small_kernel<4>, dim3(blocks), dim3(256), 0, st, g);
        DIGAT_CHECK_LAUNCH();
        return DIGAT_OK;
    }
    if (pl.lds > 64 * 1024) {
        static int raised = 0;     // benign race: the attribute is idempotent
        if (!raised) {
            if (hipFuncSetAttribute((const void*)xattn_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess) return DIGAT_ERR_LAUNCH;
            raised = 1;
        }
    }
    hipLaunchKernelGGL(xattn_score_kernel, dim3(pl.blocks), dim3(pl.threads), pl.lds, st, pl.g);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

// Pr = K3 + K1 (r already added to the neighbour-side projection), see xattn_core
static int launch_xattn_pairwise(const float* Pr, const float* Q, const float* h, const float* X,
                                 const float* a, const uint8_t* A, float* out, float* alpha,
                                 int B, int n, int d, hipStream_t st, const uint8_t* live = nullptr, const int* group = nullptr) {
    XattnPlan pl;
    const int rc = plan_xattn(B, n, d, &pl);
    if (rc) return rc;
    if (B == 0) return DIGAT_OK;
    pl.g.P = Pr; pl.g.Q = Q; pl.g.a = a; pl.g.A = A; pl.g.alpha = alpha; pl.g.live = live; pl.g.qgroup = group;
    {
        static int skip = -1;
        if (skip < 0) { const char* e = getenv("DIGAT_XATTN_SKIP"); skip = e ? atoi(e) : 0; }
        pl.g.skip = skip;
    }
    if (!(pl.g.skip & 32)) {
        // algorithmic bytes of the score launch: P', Q in (2 n d floats), adjacency, alpha out, a
        ProfScope prof(DIGAT_KERNEL_XATTN, (double)B * (2.0 * n * d * 4 + (double)n * n * 5.0) + 4.0 * d, st);
        const int rc2 = launch_score(pl, st);
        if (rc2) return rc2;
    }
    if (!(pl.g.skip & 2)) {
        AggArgs ag{alpha, h, X, out, B, n, d, (d + 63) / 64, n | 1, live, group};
        if (ag.groups > 16) return DIGAT_ERR_SHAPE;        // d <= 1024
        // algorithmic bytes of the aggregation launch: h, X in + out (3 n d floats), alpha in;
        // flops 2 n^2 d per row run on the MFMA pipe
        ProfScope prof(DIGAT_KERNEL_AGG, (double)B * (3.0 * n * d * 4 + (double)n * n * 4.0), st);
        hipLaunchKernelGGL(xattn_agg_kernel, dim3(B), dim3(64 * ag.groups), (size_t)n * ag.sa * 4 + n, st, ag);
        DIGAT_CHECK_LAUNCH();
    }
    return DIGAT_OK;
}

// =================================================================================================
// 3. ScaledDotProductAttention pooling (a6) with the key projection folded into the query:
//    (K x_j).q = x_j.(K^T q)  ->  one [B,d] vector kq, then a_j = x_j.kq / sqrt(d)
// =================================================================================================
struct PoolArgs {
    const float* feat; long ld_b;        // feat[b] = feat + b*ld_b, nodes are d floats apart
    const float* kq; const uint8_t* mask; const float* addend; float* out;
    int B, n, d; float sqrt_d;
    float* alpha_out;                    // optional [B,n]: the attention weights (training)
};

__global__ void __launch_bounds__(256) attn_pool_kernel(const PoolArgs g) {
    __shared__ float sc[DIGAT_MAX_NODES];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d4 = g.d >> 2, n = g.n;
    const float4* F4 = reinterpret_cast<const float4*>(g.feat + (long)b * g.ld_b);
    const float4* K4 = reinterpret_cast<const float4*>(g.kq + (long)b * g.d);
    for (int j = wave; j < n; j += 4) {
        float part = 0.f;
        for (int c4 = lane; c4 < d4; c4 += 64) {
            const float4 x = F4[(long)j * d4 + c4], k = K4[c4];
            part = fmaf(x.w, k.w, fmaf(x.z, k.z, fmaf(x.y, k.y, fmaf(x.x, k.x, part))));
        }
        part = wave_sum(part);
        if (lane == 0) sc[j] = g.mask[(long)b * n + j] == 0 ? -1e9f : part / g.sqrt_d;
    }
    __syncthreads();
    if (wave == 0) {
        const float v0 = lane < n ? sc[lane] : -INFINITY;
        const float v1 = lane + 64 < n ? sc[lane + 64] : -INFINITY;
        const float m = wave_max(fmaxf(v0, v1));
        const float e0 = lane < n ? expf(v0 - m) : 0.f;
        const float e1 = lane + 64 < n ? expf(v1 - m) : 0.f;
        const float s = wave_sum(e0 + e1);
        if (lane < n) sc[lane] = e0 / s;
        if (lane + 64 < n) sc[lane + 64] = e1 / s;
        if (g.alpha_out) {
            if (lane < n) g.alpha_out[(long)b * n + lane] = e0 / s;
            if (lane + 64 < n) g.alpha_out[(long)b * n + lane + 64] = e1 / s;
        }
    }
    __syncthreads();
    for (int c4 = tid; c4 < d4; c4 += 256) {
        float4 o = f4_zero();
        for (int j = 0; j < n; ++j) {
            const float al = sc[j];
            const float4 x = F4[(long)j * d4 + c4];
            o.x = fmaf(al, x.x, o.x); o.y = fmaf(al, x.y, o.y); o.z = fmaf(al, x.z, o.z); o.w = fmaf(al, x.w, o.w);
        }
        if (g.addend) o = f4_add(reinterpret_cast<const float4*>(g.addend + (long)b * g.d)[c4], o);
        reinterpret_cast<float4*>(g.out + (long)b * g.d)[c4] = o;
    }
}

static int launch_pool(const float* feat, long ld_b, const float* kq, const uint8_t* mask, const float* addend,
                       float* out, int B, int n, int d, hipStream_t st) {
    if (n > DIGAT_MAX_NODES || d % 4) return DIGAT_ERR_SHAPE;
    if (B == 0) return DIGAT_OK;
    PoolArgs g{feat, ld_b, kq, mask, addend, out, B, n, d, sqrtf((float)d), nullptr};
    ProfScope prof(DIGAT_KERNEL_POOL, (double)B * ((double)n * d * 4 + 2.0 * d * 4 + n), st);
    hipLaunchKernelGGL(attn_pool_kernel, dim3(B), dim3(256), 0, st, g);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

// =================================================================================================
// 4. topic pooling: scatter_softmax over history positions grouped by category + scatter_sum
// =================================================================================================
struct TopicArgs {
    const float* Xu; long ld_b; const float* kq; const int64_t* idx; float* out;
    int B, H, C1, d; float sqrt_d;
    float* alpha_out;                    // optional [B,H]: the segment-softmax weights (training)
    int skip;                            // ablation only (env DIGAT_TOPIC_SKIP, 0 in production)
};
constexpr int TOPIC_MAX_H = 256;
constexpr int TOPIC_MAX_CT = 4;          // 16-row category tiles: category_num + 1 <= 64

// One workgroup per row, one wave per 64 channels (7 waves at d = 400).
//  1. scores a_t = x_t . kq / sqrt(d): a wave takes four history rows at a time (all their loads in flight),
//     lanes span the channels, one wave reduction per row;
//  2. segment softmax over the rows of equal category (H threads, O(H^2) LDS reads) -> the [C1, H] matrix
//     M[c][t] = alpha_t if idx_t == c else 0 in LDS;
//  3. out = M @ X on the fp32 matrix cores: v_mfma_f32_16x16x4_f32 is an exact k-ordered fma chain and
//     fma(0, x, acc) = acc, so out[c] is the sum over the rows of category c in ascending t — the CPU
//     scatter_add order — while X streams through once, coalesced, prefetched (the scalar version of this
//     phase walked H dependent, branchy loads per output).  Lane layout as in xattn_agg_kernel: MFMA column
//     l&15 of channel tile s is channel c0 + 4*(l&15) + s, so X loads and output stores are float4.
__global__ void __launch_bounds__(1024) topic_pool_kernel(const TopicArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = g.H, hs = H | 1;
    const int ct = (g.C1 + 15) >> 4;
    float* M = reinterpret_cast<float*>(smem);               // [ct*16][hs]
    float* sa = M + ct * 16 * hs;                            // [H]
    int* sidx = reinterpret_cast<int*>(sa + H);              // [H]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nw = nthreads >> 6;
    const int d4 = g.d >> 2;
    const float* Xb = g.Xu + (long)b * g.ld_b;
    const float4* F4 = reinterpret_cast<const float4*>(Xb);
    const float4* K4 = reinterpret_cast<const float4*>(g.kq + (long)b * g.d);
    for (int t = tid; t < H; t += nthreads) {
        const long v = g.idx[(long)b * H + t];
        sidx[t] = (v >= 0 && v < g.C1) ? (int)v : -1;
    }
    for (int i = tid; i < ct * 16 * hs; i += nthreads) M[i] = 0.f;
    for (int t0 = wave; t0 < ((g.skip & 1) ? 0 : H); t0 += 4 * nw) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c4 = lane; c4 < d4; c4 += 64) {
            const float4 k = K4[c4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + r * nw;
                if (t < H) {
                    const float4 x = F4[(long)t * d4 + c4];
                    part[r] = fmaf(x.w, k.w, fmaf(x.z, k.z, fmaf(x.y, k.y, fmaf(x.x, k.x, part[r]))));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + r * nw;
            const float s = wave_sum(part[r]);
            if (lane == 0 && t < H) sa[t] = s / g.sqrt_d;
        }
    }
    __syncthreads();
    for (int t = tid; t < ((g.skip & 2) ? 0 : H); t += nthreads) {
        const int s = sidx[t];
        float m = -INFINITY;
        for (int u = 0; u < H; ++u) if (sidx[u] == s) m = fmaxf(m, sa[u]);
        float den = 0.f;
        for (int u = 0; u < H; ++u) if (sidx[u] == s) den += expf(sa[u] - m);
        const float al = s >= 0 ? expf(sa[t] - m) / den : 0.f;
        if (s >= 0) M[s * hs + t] = al;
        if (g.alpha_out) g.alpha_out[(long)b * H + t] = al;
    }
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    const int ch = wave * 64 + 4 * lr;
    const bool ch_ok = ch < g.d;             // no early exit: the MFMAs need every lane's operand rows
    const int nsteps = (g.skip & 4) ? 0 : (H + 3) >> 2;
    auto load_x = [&](int step) -> float4 {
        const int j = step * 4 + lq;
        return (j < H && ch_ok) ? *reinterpret_cast<const float4*>(Xb + (long)j * g.d + ch) : f4_zero();
    };
    constexpr int PF = 4;
    v4f acc[TOPIC_MAX_CT][4];
#pragma unroll
    for (int it = 0; it < TOPIC_MAX_CT; ++it)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[it][s] = (v4f){0.f, 0.f, 0.f, 0.f};
    float4 xq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) xq[u] = load_x(u);
    for (int step = 0; step < nsteps; step += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int st = step + u;
            if (st < nsteps) {
                const float4 xc = xq[u];
                xq[u] = load_x(st + PF);
                const int j = st * 4 + lq;
#pragma unroll
                for (int it = 0; it < TOPIC_MAX_CT; ++it) {
                    if (it < ct) {
                        const float av = j < H ? M[(it * 16 + lr) * hs + j] : 0.f;
                        acc[it][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.x, acc[it][0], 0, 0, 0);
                        acc[it][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.y, acc[it][1], 0, 0, 0);
                        acc[it][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.z, acc[it][2], 0, 0, 0);
                        acc[it][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.w, acc[it][3], 0, 0, 0);
                    }
                }
            }
        }
    }
    float* Ob = g.out + (long)b * g.C1 * g.d;
#pragma unroll
    for (int it = 0; it < TOPIC_MAX_CT; ++it) {
        if (it < ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = it * 16 + 4 * lq + r;
                if (c < g.C1 && ch_ok)
                    *reinterpret_cast<float4*>(Ob + (long)c * g.d + ch) =
                        make_float4(acc[it][0][r], acc[it][1][r], acc[it][2][r], acc[it][3][r]);
            }
        }
    }
}

// H <= 64 (the reference's max_history_num is 50): the history rows are read ONCE.  A wave owns 64 channels and
// keeps its slice of every history row in registers, in the MFMA operand layout of phase 3 (lane (lr, lq) holds
// channels c0+4lr..+3 of rows 4*step+lq): all 16 loads of a lane are in flight together, the scores are
// partial dot products over the wave's channels (reduced over the 16 lr lanes, then over the waves through LDS),
// the segment softmax is one masked wave reduction per category, and the MFMAs take X straight from the
// registers.  Same results as the kernel above up to the summation order of the scores.
constexpr int TOPIC_RES_STEPS = 16;      // 4-row steps held in registers

__global__ void __launch_bounds__(1024) topic_pool_resident_kernel(const TopicArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = g.H, hs = H | 1;
    const int ct = (g.C1 + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nw = nthreads >> 6;
    float* M = reinterpret_cast<float*>(smem);               // [ct*16][hs]
    float* part = M + ct * 16 * hs;                          // [nw][64] partial scores
    float* sa = part + nw * 64;                              // [64]
    int* sidx = reinterpret_cast<int*>(sa + 64);             // [64]
    const int b = blockIdx.x;
    const float* Xb = g.Xu + (long)b * g.ld_b;
    const int lr = lane & 15, lq = lane >> 4;
    const int ch = wave * 64 + 4 * lr;
    const bool ch_ok = ch < g.d;             // no early exit: the MFMAs need every lane's operand rows
    const int nsteps = (H + 3) >> 2;

    float4 xq[TOPIC_RES_STEPS];
#pragma unroll
    for (int s = 0; s < TOPIC_RES_STEPS; ++s) {
        const int j = s * 4 + lq;
        xq[s] = (s < nsteps && j < H && ch_ok) ? *reinterpret_cast<const float4*>(Xb + (long)j * g.d + ch) : f4_zero();
    }
    const float4 k4 = ch_ok ? *reinterpret_cast<const float4*>(g.kq + (long)b * g.d + ch) : f4_zero();
    if (tid < 64) {
        long v = -1;
        if (tid < H) v = g.idx[(long)b * H + tid];
        sidx[tid] = (v >= 0 && v < g.C1) ? (int)v : -1;
    }
    for (int i = tid; i < ct * 16 * hs; i += nthreads) M[i] = 0.f;

    // partial scores of this wave's 64 channels
#pragma unroll
    for (int s = 0; s < TOPIC_RES_STEPS; ++s) {
        float p = fmaf(xq[s].w, k4.w, fmaf(xq[s].z, k4.z, fmaf(xq[s].y, k4.y, xq[s].x * k4.x)));
        p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 8, 64);
        if (lr == 0) part[wave * 64 + s * 4 + lq] = p;
    }
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += part[w * 64 + tid];
        sa[tid] = a / g.sqrt_d;
    }
    __syncthreads();
    // segment softmax: wave w takes the categories c = w, w + nw, ...; lane t is history row t
    {
        const int my = lane < H ? sidx[lane] : -2;
        const float v = lane < H ? sa[lane] : 0.f;
        for (int c = wave; c < g.C1; c += nw) {
            const bool in = my == c;
            const float m = wave_max(in ? v : -INFINITY);
            const float e = in ? expf(v - m) : 0.f;
            const float den = wave_sum(e);
            if (in) {
                const float al = e / den;
                M[c * hs + lane] = al;
                if (g.alpha_out) g.alpha_out[(long)b * H + lane] = al;
            }
        }
        if (wave == 0 && g.alpha_out && lane < H && my < 0) g.alpha_out[(long)b * H + lane] = 0.f;
    }
    __syncthreads();

    v4f acc[TOPIC_MAX_CT][4];
#pragma unroll
    for (int it = 0; it < TOPIC_MAX_CT; ++it)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[it][s] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < TOPIC_RES_STEPS; ++s) {
        if (s < nsteps) {
            const int j = s * 4 + lq;
            const float4 xc = xq[s];
#pragma unroll
            for (int it = 0; it < TOPIC_MAX_CT; ++it) {
                if (it < ct) {
                    const float av = j < H ? M[(it * 16 + lr) * hs + j] : 0.f;
                    acc[it][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.x, acc[it][0], 0, 0, 0);
                    acc[it][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.y, acc[it][1], 0, 0, 0);
                    acc[it][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.z, acc[it][2], 0, 0, 0);
                    acc[it][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc.w, acc[it][3], 0, 0, 0);
                }
            }
        }
    }
    float* Ob = g.out + (long)b * g.C1 * g.d;
#pragma unroll
    for (int it = 0; it < TOPIC_MAX_CT; ++it) {
        if (it < ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = it * 16 + 4 * lq + r;
                if (c < g.C1 && ch_ok)
                    *reinterpret_cast<float4*>(Ob + (long)c * g.d + ch) =
                        make_float4(acc[it][0][r], acc[it][1][r], acc[it][2][r], acc[it][3][r]);
            }
        }
    }
}

static int launch_topic_args(TopicArgs g, hipStream_t st) {
    static int skip = -1;
    if (skip < 0) { const char* e = getenv("DIGAT_TOPIC_SKIP"); skip = e ? atoi(e) : 0; }
    g.skip = skip;
    const int groups = (g.d + 63) / 64;
    const int ct = (g.C1 + 15) / 16;
    if (g.H > TOPIC_MAX_H || g.d % 4 || groups > 16 || ct > TOPIC_MAX_CT) return DIGAT_ERR_SHAPE;
    const size_t lds = ((size_t)ct * 16 * (g.H | 1) + 2 * (size_t)g.H) * 4;
    if (lds > 64 * 1024) {
        static int raised = 0;
        if (!raised) {
            if (hipFuncSetAttribute((const void*)topic_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess) return DIGAT_ERR_LAUNCH;
            raised = 1;
        }
    }
    if (g.H <= 64 && !(g.skip & 64)) {
        const size_t ldsr = ((size_t)ct * 16 * (g.H | 1) + (size_t)groups * 64 + 128) * 4;
        hipLaunchKernelGGL(topic_pool_resident_kernel, dim3(g.B), dim3(64 * groups), ldsr, st, g);
    } else {
        hipLaunchKernelGGL(topic_pool_kernel, dim3(g.B), dim3(64 * groups), lds, st, g);
    }
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

static int launch_topic(const float* Xu, long ld_b, const float* kq, const int64_t* idx, float* out,
                        int B, int H, int C1, int d, hipStream_t st) {
    if (B == 0) return DIGAT_OK;
    TopicArgs g{Xu, ld_b, kq, idx, out, B, H, C1, d, sqrtf((float)d), nullptr, 0};
    ProfScope prof(DIGAT_KERNEL_TOPIC, (double)B * ((double)H * d * 4 + d * 4.0 + H * 8.0 + (double)C1 * d * 4), st);
    return launch_topic_args(g, st);
}

// =================================================================================================
// 5. glue kernels
// =================================================================================================
// Xu[b] = [user_news_embedding[b] (H rows) | topic_node_embedding (C rows)]   (graphEncoders.py:191)
// group != NULL: row b takes the history of user group[b] (rows of one impression share the user side)
__global__ void __launch_bounds__(256) build_user_nodes_kernel(const float4* ue, const float4* topic, float4* Xu,
                                                               long B, int H, int C, int d4, const int* group) {
    const long per_row = (long)(H + C) * d4;
    const long total = B * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / per_row;
        const long rem = i - b * per_row;
        const long hist = (long)H * d4;
        const long src = group ? group[b] : b;
        Xu[i] = rem < hist ? ue[src * hist + rem] : topic[rem - hist];
    }
}

// ---- live user-graph nodes --------------------------------------------------------------------------------
// A node that has no edge to any other node and whose pooled contribution is masked cannot reach the encoder's
// outputs: padding slots of the history (MIND_corpus.py:153-156: identity row, category index C, and
// category_mask[C] is never set) and topic nodes of categories the user never read (identity row, never pooled).
// About half of the 67 nodes of a MIND-shaped user graph are such nodes.  Their rows are left out of the
// projections of layers >= 1 (layer 0 computes every row, so the buffers stay finite; a dead node keeps
// evolving from stale-but-finite values that nothing reads).  Live means:
//   any off-diagonal entry in the node's adjacency row or column, or
//   a history slot whose category is unmasked, or any history slot of a row with NO unmasked category (the
//   context attention is then uniform over all C+1 buckets: util E2, the empty-history user), or
//   any node at all when some adjacency row of the graph has no entry, not even the self loop (E5: that centre's
//   scores are all -1e9, its softmax is uniform over EVERY node).
// One wave per row b.  Lane i builds the bit mask of row i of the adjacency (diagonal cleared); a node has an
// off-diagonal entry in its row iff its mask is non-zero, and in its column iff its bit is set in the OR of all
// the masks.  flags [B,U] bytes and cnt [B] are written; after the scan, live_list_kernel writes the row indices
// b*U + i of the live nodes in ascending order.
__global__ void __launch_bounds__(256) user_live_flags_kernel(const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx,
                                                              int B, int U, int H, int C1, uint8_t* flags, int* cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const uint8_t* A = Au + (long)b * U * U;
    bool any_cat = false;
    for (int c = lane; c < C1; c += 64) any_cat |= cat_mask[(long)b * C1 + c] != 0;
    any_cat = __any(any_cat);
    // U <= DIGAT_MAX_NODES = 128: two nodes per lane, masks of 4 x 32 bits
    unsigned rm[2][4];
    unsigned cm[4] = {0u, 0u, 0u, 0u};
    bool empty_row = false;          // a row without ANY entry (not even the self loop) is all -1e9: uniform over every node
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
#pragma unroll
        for (int w = 0; w < 4; ++w) rm[h][w] = 0u;
        if (i < U) {
            const uint8_t* row = A + (long)i * U;
            bool any = false;
            for (int j = 0; j < U; ++j) {
                any |= row[j] != 0;
                if (row[j] && j != i) rm[h][j >> 5] |= 1u << (j & 31);
            }
            empty_row |= !any;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) cm[w] |= rm[h][w];
    }
    const bool all_live = __any(empty_row);     // such a centre reads every node of the row: nothing may be left out
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cm[w] |= __shfl_xor(cm[w], o, 64);
    int count = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        bool live = false;
        if (i < U) {
            live = all_live || (rm[h][0] | rm[h][1] | rm[h][2] | rm[h][3]) != 0u || ((cm[i >> 5] >> (i & 31)) & 1u);
            if (i < H) {
                const long c = cat_idx[(long)b * H + i];
                live |= !any_cat || (c >= 0 && c < C1 && cat_mask[(long)b * C1 + c] != 0);
            }
            flags[(long)b * U + i] = live ? 1 : 0;
        }
        count += __popcll(__ballot(live));
    }
    if (lane == 0) cnt[b] = count;
}

// The pooled topic buckets [B, C+1] feed featureAffine and then the user attention, which masks every category the
// user never read (weight exactly 0): only the unmasked buckets of a row are live — or all of them when no category
// is unmasked (uniform attention).
__global__ void __launch_bounds__(256) bucket_live_flags_kernel(const uint8_t* cat_mask, int B, int C1, uint8_t* flags, int* cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    bool any_cat = false;
    for (int c = lane; c < C1; c += 64) any_cat |= cat_mask[(long)b * C1 + c] != 0;
    any_cat = __any(any_cat);
    int count = 0;
    for (int c0 = 0; c0 < C1; c0 += 64) {
        const int c = c0 + lane;
        const bool live = c < C1 && (!any_cat || cat_mask[(long)b * C1 + c] != 0);
        if (c < C1) flags[(long)b * C1 + c] = live ? 1 : 0;
        count += __popcll(__ballot(live));
    }
    if (lane == 0) cnt[b] = count;
}

__global__ void __launch_bounds__(256) live_list_kernel(const uint8_t* flags, const int* off, int B, int U, int* rowidx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    int base = off[b];
    for (int i0 = 0; i0 < U; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < U && flags[(long)b * U + i] != 0;
        const unsigned long long m = __ballot(live);
        if (live) rowidx[base + __popcll(m & ((1ull << lane) - 1ull))] = b * U + i;
        base += __popcll(m);
    }
}

// off[b] = sum of cnt[0..b), off[B] = total: one workgroup, B <= 2^20
__global__ void __launch_bounds__(1024) exclusive_scan_kernel(const int* cnt, int* off, int B) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int s = tid * per, e = s + per < B ? s + per : B;
    int a = 0;
    for (int i = s; i < e; ++i) a += cnt[i];
    part[tid] = a;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (int i = s; i < e; ++i) { off[i] = run; run += cnt[i]; }
    if (tid == 1023) off[B] = part[1023];
}

// out[b] = in[group[b]] for rows of `row_bytes` bytes (16-byte multiple or byte-wise)
__global__ void __launch_bounds__(256) gather_rows_kernel(const uint8_t* in, uint8_t* out, const int* group, long B, long row_bytes) {
    const long total = B * row_bytes;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / row_bytes;
        out[i] = in[(long)group[b] * row_bytes + (i - b * row_bytes)];
    }
}

// layer-0 user graph of grouped rows: h[b] = h0[g], P'[b] = r[b] + P0[g] (K3 + K1), Q[b] = Q0[g]
__global__ void __launch_bounds__(256) expand_proj_kernel(const float4* P0, const float4* r, const int* group, float4* Pr,
                                                          long B, int n, int d4) {
    const long per_row = (long)n * d4;
    const long total = B * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / per_row;
        const long rem = i - b * per_row;
        const int c4 = (int)(rem % d4);
        Pr[i] = f4_add(r[b * d4 + c4], P0[(long)group[b] * per_row + rem]);
    }
}

__global__ void __launch_bounds__(256) row_logits_kernel(const float* nc, const float* uc, float* logits, int B, int d) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    float part = 0.f;
    for (int c = lane; c < d; c += 64) part = fmaf(uc[(long)b * d + c], nc[(long)b * d + c], part);
    part = wave_sum(part);
    if (lane == 0) logits[b] = part;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int digat_version(void) { return DIGAT_ABI_VERSION; }

const char* digat_error_string(int code) {
    switch (code) {
        case DIGAT_OK: return "ok";
        case DIGAT_ERR_ARG: return "bad argument (null pointer or negative size)";
        case DIGAT_ERR_SHAPE: return "unsupported shape (d % 4 != 0, graph larger than DIGAT_MAX_NODES, depth too large)";
        case DIGAT_ERR_WORKSPACE: return "workspace too small";
        case DIGAT_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown error";
    }
}

int digat_linear_f32(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy,
                     int M, int N, int K, void* stream) {
    if (!x || !w || !y || M < 0 || N <= 0 || K <= 0) return DIGAT_ERR_ARG;
    if (K % 4 || ldx % 4) return DIGAT_ERR_SHAPE;
    return launch_gemm(gemm_plain(x, ldx, w, b, y, ldy, M, N, K, 0), (hipStream_t)stream);
}

// ---- a1 / a2 ------------------------------------------------------------------------------------
size_t digat_xattn_workspace_bytes(int B, int n, int d) {
    // h, P, Q [B,n,d] + r [B,d] + alpha [B,n,n]
    return align_up((size_t)3 * B * n * d * 4, 256) + align_up((size_t)B * d * 4, 256)
           + align_up((size_t)B * n * n * 4, 256);
}

int digat_xattn_pairwise_fwd(const float* Pr, const float* Q, const float* h, const float* X,
                             const float* a, const uint8_t* A, float* out, float* alpha,
                             int B, int n, int d, void* stream) {
    if (!Pr || !Q || !h || !X || !a || !A || !out || !alpha) return DIGAT_ERR_ARG;
    return launch_xattn_pairwise(Pr, Q, h, X, a, A, out, alpha, B, n, d, (hipStream_t)stream);
}

// Eq. 8 layer with K3 (r = ctx F3^T + b3) already computed; `r_given` may live anywhere
static int xattn_core(const float* X, const uint8_t* A, const float* r_given,
                      const float* W, const float* bW, const float* F1, const float* F2, const float* a,
                      float* out, float* alpha_out, int B, int n, int d, void* workspace, hipStream_t st,
                      const void* wsplit = nullptr, const int* rowidx = nullptr, const int* nrows_dev = nullptr,
                      const uint8_t* live = nullptr) {
    const size_t nd = (size_t)B * n * d;
    float* h = (float*)workspace;
    float* P = h + nd;
    float* Q = P + nd;
    float* alpha = alpha_out ? alpha_out
                             : (float*)((char*)workspace + align_up(3 * nd * 4, 256) + align_up((size_t)B * d * 4, 256));
    // [h | P | Q] = X [W | F1 | F2]^T (+ bW on h): one pass over X on the matrix cores
    GemmArgs g = gemm_plain(X, d, W, bW, h, d, B * n, d, d, 0);
    g.w[1] = F1; g.bias[1] = nullptr; g.y[1] = P;
    g.w[2] = F2; g.bias[2] = nullptr; g.y[2] = Q;
    g.nsegs = 3;
    g.wsplit = (const unsigned short*)wsplit;          // non-NULL: bf16x6 on the bf16 matrix cores
    g.radd = r_given; g.radd_seg = 1; g.rows_per_b = n; // P' = K3 + K1: the reference's left-to-right order
    const bool listed = rowidx && gemm_is_bf16x6(g);
    if (listed) { g.rowidx = rowidx; g.nrows_dev = nrows_dev; }                         // live rows only (see user_live_flags_kernel)
    const int rc = launch_gemm(g, st, DIGAT_KERNEL_PROJ);
    if (rc) return rc;
    return launch_xattn_pairwise(P, Q, h, X, a, A, out, alpha, B, n, d, st, listed ? live : nullptr);
}

int digat_xattn_fwd(const float* X, const uint8_t* A, const float* ctx,
                    const float* W, const float* bW, const float* F1, const float* F2,
                    const float* F3, const float* b3, const float* a,
                    float* out, float* alpha_out, int B, int n, int d,
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !A || !ctx || !W || !F1 || !F2 || !F3 || !a || !out || !workspace) return DIGAT_ERR_ARG;
    if (B < 0 || n <= 0 || d <= 0) return DIGAT_ERR_ARG;
    if (d % 4 || n > DIGAT_MAX_NODES) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_xattn_workspace_bytes(B, n, d)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    hipStream_t st = (hipStream_t)stream;
    float* r = (float*)((char*)workspace + align_up((size_t)3 * B * n * d * 4, 256));
    // r = ctx F3^T + b3   (K3)
    const int rc = launch_gemm(gemm_plain(ctx, d, F3, b3, r, d, B, d, d, 0), st);
    if (rc) return rc;
    return xattn_core(X, A, r, W, bW, F1, F2, a, out, alpha_out, B, n, d, workspace, st);
}

// ---- bf16x6 weight preparation + a directly callable linear (tests, micro-benchmarks) --------------
size_t digat_split_weights_bytes(int rows, int K) {
    return (size_t)((rows + 79) / 80) * ((K + 31) / 32) * WS_SLOTS * 16;       // one 15 KB image per (80-row strip, K tile)
}

static int launch_split(const float* w0, const float* w1, const float* w2, int nseg, int nsegs, int K, void* wsplit, hipStream_t st) {
    const long total = (long)nseg * nsegs * K;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_weights_tiled_kernel, dim3(blocks), dim3(256), 0, st, w0, w1, w2, nseg, nsegs, K, (unsigned short*)wsplit);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

int digat_split_proj_weights(const float* W, const float* F1, const float* F2, int d, void* wsplit, void* stream) {
    if (!W || !F1 || !F2 || !wsplit || d <= 0) return DIGAT_ERR_ARG;
    return launch_split(W, F1, F2, d, 3, d, wsplit, (hipStream_t)stream);
}

int digat_split_weights(const float* W, int N, int K, void* wsplit, void* stream) {
    if (!W || !wsplit || N <= 0 || K <= 0) return DIGAT_ERR_ARG;
    return launch_split(W, W, W, N, 1, K, wsplit, (hipStream_t)stream);
}

int digat_linear_f32x3(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy,
                       int M, int N, int K, void* wsplit, void* stream) {
    if (!x || !w || !y || !wsplit || M < 0 || N <= 0 || K <= 0) return DIGAT_ERR_ARG;
    if (K % 8 || ldx % 4 || N % 80) return DIGAT_ERR_SHAPE;
    const int rcs = launch_split(w, w, w, N, 1, K, wsplit, (hipStream_t)stream);
    if (rcs) return rcs;
    GemmArgs g = gemm_plain(x, ldx, w, b, y, ldy, M, N, K, 0);
    g.wsplit = (const unsigned short*)wsplit;
    if (M < 2048) return DIGAT_ERR_SHAPE;      // the bf16x6 kernel serves the big projections only
    return launch_gemm(g, (hipStream_t)stream, DIGAT_KERNEL_PROJ);
}

// ---- a3 -----------------------------------------------------------------------------------------
size_t digat_news_ctx_workspace_bytes(int B, int N, int d) {
    (void)N;
    return 3 * align_up((size_t)B * d * 4, 256);
}

int digat_news_ctx_fwd(const float* X, const uint8_t* mask, const float* Kc, const float* Qc, const float* bQc,
                       const float* Wg, const float* bg, const float* addend, float* out, int B, int N, int d,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !mask || !Kc || !Qc || !Wg || !out || !workspace || B < 0 || N <= 0 || d <= 0) return DIGAT_ERR_ARG;
    if (d % 4 || N > DIGAT_MAX_NODES) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_news_ctx_workspace_bytes(B, N, d)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t slot = align_up((size_t)B * d * 4, 256);
    float* qv = (float*)workspace;
    float* kq = (float*)((char*)workspace + slot);
    float* glob = (float*)((char*)workspace + 2 * slot);
    const long ldx = (long)N * d;      // node 0 of every row: the local context (graphEncoders.py:110)
    int rc;
    rc = launch_gemm(gemm_plain(X, ldx, Qc, bQc, qv, d, B, d, d, 0), st);           // Q(query)
    if (rc) return rc;
    rc = launch_gemm(gemm_plain(qv, d, Kc, nullptr, kq, d, B, d, d, 1), st);         // K^T q
    if (rc) return rc;
    rc = launch_pool(X, ldx, kq, mask, nullptr, glob, B, N, d, st);                  // global context
    if (rc) return rc;
    GemmArgs g = gemm_plain(X, ldx, Wg, bg, out, d, B, d, 2 * d, 0);                 // gate([local ; global])
    g.k0 = d; g.a1 = glob; g.lda1 = d;
    g.epi = EPI_GATE; g.e0 = X; g.lde0 = ldx; g.e1 = glob; g.lde1 = d; g.e2 = addend; g.lde2 = d;
    return launch_gemm(g, st);
}

// ---- a4 -----------------------------------------------------------------------------------------
size_t digat_user_ctx_workspace_bytes(int B, int U, int H, int C1, int d) {
    (void)U; (void)H;
    return 2 * align_up((size_t)B * d * 4, 256) + 2 * align_up((size_t)B * C1 * d * 4, 256);
}

int digat_topic_pool_fwd(const float* Xu, const float* kq, const int64_t* cat_idx, float* out,
                         int B, int U, int H, int C1, int d, void* stream) {
    if (!Xu || !kq || !cat_idx || !out || B < 0 || H < 0 || U < H || C1 <= 0 || d <= 0) return DIGAT_ERR_ARG;
    return launch_topic(Xu, (long)U * d, kq, cat_idx, out, B, H, C1, d, (hipStream_t)stream);
}

int digat_user_ctx_fwd(const float* Xu, const uint8_t* cat_mask, const int64_t* cat_idx, const float* c_n,
                       const float* Ku, const float* Qu, const float* bQu, const float* Fa, const float* bFa,
                       const float* Kua, const float* Qua, const float* bQua, const float* addend, float* out,
                       int B, int U, int H, int C1, int d, void* workspace, size_t workspace_bytes, void* stream) {
    if (!Xu || !cat_mask || !cat_idx || !c_n || !Ku || !Qu || !Fa || !Kua || !Qua || !out || !workspace)
        return DIGAT_ERR_ARG;
    if (B < 0 || H < 0 || U < H || C1 <= 0 || d <= 0) return DIGAT_ERR_ARG;
    if (d % 4 || C1 > DIGAT_MAX_NODES || H > TOPIC_MAX_H) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_user_ctx_workspace_bytes(B, U, H, C1, d)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t s1 = align_up((size_t)B * d * 4, 256), s2 = align_up((size_t)B * C1 * d * 4, 256);
    float* qv = (float*)workspace;
    float* kq = (float*)((char*)workspace + s1);
    float* T = (float*)((char*)workspace + 2 * s1);
    float* T2 = (float*)((char*)workspace + 2 * s1 + s2);
    int rc;
    // topic-level attention (:126-130)
    rc = launch_gemm(gemm_plain(c_n, d, Qu, bQu, qv, d, B, d, d, 0), st);
    if (rc) return rc;
    rc = launch_gemm(gemm_plain(qv, d, Ku, nullptr, kq, d, B, d, d, 1), st);
    if (rc) return rc;
    rc = launch_topic(Xu, (long)U * d, kq, cat_idx, T, B, H, C1, d, st);
    if (rc) return rc;
    // featureAffine + relu + residual (:131)
    GemmArgs g = gemm_plain(T, d, Fa, bFa, T2, d, B * C1, d, d, 0);
    g.epi = EPI_RELU_RES; g.e0 = T; g.lde0 = d;
    rc = launch_gemm(g, st);
    if (rc) return rc;
    // user-level attention (:133)
    rc = launch_gemm(gemm_plain(c_n, d, Qua, bQua, qv, d, B, d, d, 0), st);
    if (rc) return rc;
    rc = launch_gemm(gemm_plain(qv, d, Kua, nullptr, kq, d, B, d, d, 1), st);
    if (rc) return rc;
    return launch_pool(T2, (long)C1 * d, kq, cat_mask, addend, out, B, C1, d, st);
}

// ---- folded attention queries (inference): (K x).(Q c + b) = x.(K^T Q c + K^T b) ------------------
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int d) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int y = by + r, x = bx + threadIdx.x;
        if (y < d && x < d) tile[r][threadIdx.x] = in[(long)y * d + x];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int y = bx + r, x = by + threadIdx.x;
        if (y < d && x < d) out[(long)y * d + x] = tile[threadIdx.x][r];
    }
}

size_t digat_fold_workspace_bytes(int d) { return 2 * align_up((size_t)d * d * 4, 256); }

int digat_fold_attention(const float* K, const float* Q, const float* bQ, float* Wf, float* bf, int d,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!K || !Q || !Wf || !bf || !workspace || d <= 0) return DIGAT_ERR_ARG;
    if (d % 4) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_fold_workspace_bytes(d)) return DIGAT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* Kt = (float*)workspace;
    float* Qt = (float*)((char*)workspace + align_up((size_t)d * d * 4, 256));
    const dim3 grid((d + 31) / 32, (d + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, K, Kt, d);
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, Q, Qt, d);
    DIGAT_CHECK_LAUNCH();
    // Wf[c][i] = sum_o K[o][c] Q[o][i]  = linear(x = K^T [c,o], w = Q^T [i,o])
    int rc = launch_gemm(gemm_plain(Kt, d, Qt, nullptr, Wf, d, d, d, d, 0), st);
    if (rc) return rc;
    // bf[c] = sum_o K[o][c] bQ[o]       = linear(x = bQ [1,o], w = K^T [c,o])
    if (bQ) return launch_gemm(gemm_plain(bQ, d, Kt, nullptr, bf, d, 1, d, d, 0), st);
    return hipMemsetAsync(bf, 0, (size_t)d * 4, st) == hipSuccess ? DIGAT_OK : DIGAT_ERR_LAUNCH;
}

// ---- a5 -----------------------------------------------------------------------------------------
static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// Inference fast path with folded attention queries (digat_fold_attention): per layer the [B,d]
// linears shrink from 9 launches to 4 — {topic query, user query, next layer's K3 of the user graph}
// all read the same c_n and go out as ONE three-segment launch.
// Within a layer the news-graph update and the user-graph update read only the PREVIOUS contexts
// (graphEncoders.py:194-195), so the news chain — K3, projection, score, aggregation, context pooling, gate, and
// the queries derived from the new c_n — is independent of the user graph's Eq. 8 until the user context is
// pooled.  The news kernels are small (N = 10 nodes, [B,d] linears: tens of workgroups, latency chains) and run
// on a side stream under the user graph's projection / score / aggregation, which fill the chip; fork and join
// are two events per layer (a pattern hipGraph capture accepts).  DIGAT_SINGLE_STREAM=1 keeps everything on the
// caller's stream.
struct SideStream { hipStream_t s; hipEvent_t fork, join; int ok; };
static int g_live_rows_on = getenv("DIGAT_NO_SKIP") && atoi(getenv("DIGAT_NO_SKIP")) ? 0 : 1;
static int g_side_stream_on = getenv("DIGAT_SINGLE_STREAM") && atoi(getenv("DIGAT_SINGLE_STREAM")) ? 0 : 1;
static SideStream* side_stream() {
    static SideStream tab[16];
    static int state[16];                    // 0 = untried, 1 = ready, -1 = unavailable
    int dev = 0;
    if (!g_side_stream_on || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (state[dev] == 0) {                   // one process per GPU; a race here would only create a spare stream
        SideStream& x = tab[dev];
        const bool ok = hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess;
        state[dev] = ok ? 1 : -1;
    }
    return state[dev] == 1 ? &tab[dev] : nullptr;
}

static int encoder_fwd_folded(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                              const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx,
                              float* c_n, float* c_u, int B, int N, int H, float* const Xu[2], float* const Xn[2],
                              void* xws, void* xws_news, void* cws, float* kq_t, float* kq_u, float* const r_user2[2],
                              float* r_news, int* live_ws, hipStream_t st, const int* row_group, int G, const float* ue_groups) {
    const int d = p->d, C = p->category_num, L = p->depth, U = H + C, C1 = C + 1;
    const size_t s2 = align_up((size_t)B * C1 * d * 4, 256);
    float* T = (float*)cws;                       // [B,C1,d] pooled topics
    float* T2 = (float*)((char*)cws + s2);        // after featureAffine
    float* glob = (float*)((char*)cws + 2 * s2);  // [B,d]; cws holds >= 2*s2 + 2*[B,d] (user-context layout)
    const int* bucket_idx = nullptr;              // live topic buckets (set by find_live_rows during layer 0)
    const int* nbuckets_dev = nullptr;
    int rc;
    // the user-side queries + (optionally) the next user-graph K3, all from c_n
    auto from_c_n = [&](int next_layer, hipStream_t sq) -> int {
        GemmArgs g = gemm_plain(c_n, d, p->user_news_fold_W, p->user_news_fold_b, kq_t, d, B, d, d, 0);
        g.w[1] = p->userAtt_fold_W; g.bias[1] = p->userAtt_fold_b; g.y[1] = kq_u;
        g.nsegs = 2;
        if (next_layer < L) {
            g.w[2] = p->user[next_layer].F3; g.bias[2] = p->user[next_layer].b3; g.y[2] = r_user2[next_layer & 1];
            g.nsegs = 3;
        }
        return launch_gemm(g, sq);
    };
    auto user_ctx_tail = [&](const float* Xu_cur, const float* addend) -> int {
        int e = launch_topic(Xu_cur, (long)U * d, kq_t, cat_idx, T, B, H, C1, d, st);
        if (e) return e;
        GemmArgs g = gemm_plain(T, d, p->featureAffine_W, p->featureAffine_b, T2, d, B * C1, d, d, 0);
        g.epi = EPI_RELU_RES; g.e0 = T; g.lde0 = d;
        g.wsplit = (const unsigned short*)p->featureAffine_wsplit;       // non-NULL: bf16x6
        if (bucket_idx && gemm_is_bf16x6(g)) { g.rowidx = bucket_idx; g.nrows_dev = nbuckets_dev; }   // unmasked buckets only
        e = launch_gemm(g, st);
        if (e) return e;
        return launch_pool(T2, (long)C1 * d, kq_u, cat_mask, addend, c_u, B, C1, d, st);
    };
    auto news_ctx = [&](const float* Xn_cur, hipStream_t sq) -> int {
        const long ldx = (long)N * d;
        float* kq = kq_t;                          // free here: the previous user context has consumed it
        int e = launch_gemm(gemm_plain(Xn_cur, ldx, p->cand_fold_W, p->cand_fold_b, kq, d, B, d, d, 0), sq);
        if (e) return e;
        e = launch_pool(Xn_cur, ldx, kq, Mn, nullptr, glob, B, N, d, sq);
        if (e) return e;
        GemmArgs g = gemm_plain(Xn_cur, ldx, p->news_graph_W, p->news_graph_b, c_n, d, B, d, 2 * d, 0);
        g.k0 = d; g.a1 = glob; g.lda1 = d;
        g.epi = EPI_GATE; g.e0 = Xn_cur; g.lde0 = ldx; g.e1 = glob; g.lde1 = d; g.e2 = c_n; g.lde2 = d;
        return launch_gemm(g, sq);
    };

    // live rows of the user graph for the projections of layers >= 1 (DIGAT_NO_SKIP=1: every row)
    const int* rowidx = nullptr;
    const int* nrows_dev = nullptr;
    uint8_t* live_flags = nullptr;
    auto find_live_rows = [&](hipStream_t sq) -> int {     // needed after layer 0's Eq. 8: runs beside it
        int* cnt = live_ws;
        int* off = cnt + align_up((size_t)B, 64);
        int* idx = off + align_up((size_t)B + 1, 64);
        int* cnt2 = idx + align_up((size_t)B * U, 64);
        int* off2 = cnt2 + align_up((size_t)B, 64);
        int* idx2 = off2 + align_up((size_t)B + 1, 64);
        live_flags = (uint8_t*)(idx2 + align_up((size_t)B * C1, 64));
        uint8_t* flags2 = live_flags + align_up((size_t)B * U, 256);
        ProfScope prof(DIGAT_KERNEL_GLUE, (double)B * ((double)U * U + 2.0 * C1 + H * 8.0) + (double)B * (U + C1) * 6, sq);
        hipLaunchKernelGGL(user_live_flags_kernel, dim3((B + 3) / 4), dim3(256), 0, sq, Au, cat_mask, cat_idx, B, U, H, C1,
                           live_flags, cnt);
        DIGAT_CHECK_LAUNCH();
        hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, sq, (const int*)cnt, off, B);
        DIGAT_CHECK_LAUNCH();
        hipLaunchKernelGGL(live_list_kernel, dim3((B + 3) / 4), dim3(256), 0, sq, (const uint8_t*)live_flags, (const int*)off, B, U, idx);
        DIGAT_CHECK_LAUNCH();
        hipLaunchKernelGGL(bucket_live_flags_kernel, dim3((B + 3) / 4), dim3(256), 0, sq, cat_mask, B, C1, flags2, cnt2);
        DIGAT_CHECK_LAUNCH();
        hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, sq, (const int*)cnt2, off2, B);
        DIGAT_CHECK_LAUNCH();
        hipLaunchKernelGGL(live_list_kernel, dim3((B + 3) / 4), dim3(256), 0, sq, (const uint8_t*)flags2, (const int*)off2, B, C1, idx2);
        DIGAT_CHECK_LAUNCH();
        rowidx = idx; nrows_dev = off + B;
        bucket_idx = idx2; nbuckets_dev = off2 + B;
        return DIGAT_OK;
    };
    const bool want_live = L > 0 && g_live_rows_on && live_ws;
    SideStream* side = side_stream();
    rc = from_c_n(0, st);
    if (rc) return rc;
    rc = user_ctx_tail(Xu[0], nullptr);            // c_u (:192)
    if (rc) return rc;
    const float* xn_cur = Xn_in;
    int un = 0, nn = 0;
    for (int i = 0; i < L; ++i) {
        const digat_layer_params& ln = p->news[i];
        const digat_layer_params& lu = p->user[i];
        const float* r_user = r_user2[i & 1];     // K3 of the user graph, from the previous c_n
        hipStream_t sn = side ? side->s : st;
        if (side) {
            if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(sn, side->fork, 0) != hipSuccess)
                return DIGAT_ERR_LAUNCH;
        }
        // ---- user graph, Eq. 8 (caller's stream)
        if (i == 0 && row_group) {
            // layer 0 of grouped rows: every row of a group has the same user nodes, so project the G groups once
            // ([G*U] rows instead of [B*U]).  h and Q of a group go straight to the h / Q slots of the Eq. 8 workspace
            // and are read through the group index by the aggregation / score kernels (the 37 rows of an impression
            // share them: they stay in L2); only P' = K3_b + P depends on the row and is expanded.
            const size_t ndg = (size_t)G * U * d;
            const size_t nd = (size_t)B * U * d;
            float* Xg = Xu[1];                                  // group nodes and P of the groups: free until this
            float* P0 = Xg + ndg;                               // layer's output is written (2 ndg <= nd)
            float* h0 = (float*)xws;
            float* P = h0 + nd;
            float* Q0 = P + nd;
            {
                const long total4 = (long)ndg / 4;
                int blocks = (int)((total4 + 255) / 256);
                if (blocks > 2048) blocks = 2048;
                hipLaunchKernelGGL(build_user_nodes_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)ue_groups,
                                   (const float4*)p->topic_node_embedding, (float4*)Xg, (long)G, H, C, d / 4, (const int*)nullptr);
                DIGAT_CHECK_LAUNCH();
            }
            GemmArgs gg = gemm_plain(Xg, d, lu.W, lu.bW, h0, d, G * U, d, d, 0);
            gg.w[1] = lu.F1; gg.bias[1] = nullptr; gg.y[1] = P0;
            gg.w[2] = lu.F2; gg.bias[2] = nullptr; gg.y[2] = Q0;
            gg.nsegs = 3;
            gg.wsplit = (const unsigned short*)lu.wsplit;
            gg.m_dispatch = B * U;                              // the kernel the per-row path would pick: same bits
            rc = launch_gemm(gg, st, DIGAT_KERNEL_PROJ);
            if (rc) return rc;
            float* alpha = (float*)((char*)xws + align_up(3 * nd * 4, 256) + align_up((size_t)B * d * 4, 256));
            {
                const long total4 = (long)nd / 4;
                int blocks = (int)((total4 + 255) / 256);
                if (blocks > 4096) blocks = 4096;
                ProfScope prof(DIGAT_KERNEL_GLUE, (double)nd * 4, st);
                hipLaunchKernelGGL(expand_proj_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)P0, (const float4*)r_user,
                                   row_group, (float4*)P, (long)B, U, d / 4);
                DIGAT_CHECK_LAUNCH();
            }
            rc = launch_xattn_pairwise(P, Q0, h0, Xu[0], lu.a, Au, Xu[1], alpha, B, U, d, st, nullptr, row_group);
        } else {
            // layer 0 computes every row (the buffers then hold finite values everywhere); later layers only the live ones
            rc = xattn_core(Xu[un], Au, r_user, lu.W, lu.bW, lu.F1, lu.F2, lu.a, Xu[un ^ 1], nullptr, B, U, d, xws, st, lu.wsplit,
                            i > 0 ? rowidx : nullptr, i > 0 ? nrows_dev : nullptr, i > 0 ? live_flags : nullptr);
        }
        if (rc) return rc;
        // ---- news graph, Eq. 8 + context + the queries that follow from the new c_n (side stream)
        if (i == 0 && want_live) {
            rc = find_live_rows(sn);
            if (rc) return rc;
        }
        rc = launch_gemm(gemm_plain(c_u, d, ln.F3, ln.b3, r_news, d, B, d, d, 0), sn);     // K3 of the news graph
        if (rc) return rc;
        rc = xattn_core(xn_cur, An, r_news, ln.W, ln.bW, ln.F1, ln.F2, ln.a, Xn[nn], nullptr, B, N, d, xws_news, sn, ln.wsplit);
        if (rc) return rc;
        xn_cur = Xn[nn]; nn ^= 1; un ^= 1;
        rc = news_ctx(xn_cur, sn);                 // c_n += ... (:196)
        if (rc) return rc;
        rc = from_c_n(i + 1, sn);                  // queries (+ next K3, into the other r_user buffer) from the UPDATED c_n
        if (rc) return rc;
        if (side) {
            if (hipEventRecord(side->join, sn) != hipSuccess || hipStreamWaitEvent(st, side->join, 0) != hipSuccess)
                return DIGAT_ERR_LAUNCH;
        }
        rc = user_ctx_tail(Xu[un], c_u);           // c_u += ... (:197)
        if (rc) return rc;
    }
    return DIGAT_OK;
}

size_t digat_encoder_workspace_bytes(int B, int N, int H, int C, int d, int depth) {
    (void)depth;
    const int U = H + C;
    const int nmax = N > U ? N : U;
    size_t tot = 0;
    tot += 2 * align_up((size_t)B * U * d * 4, 256);     // user nodes, ping-pong
    tot += 2 * align_up((size_t)B * N * d * 4, 256);     // news nodes, ping-pong
    tot += digat_xattn_workspace_bytes(B, nmax, d);
    tot += max_sz(digat_news_ctx_workspace_bytes(B, N, d), digat_user_ctx_workspace_bytes(B, U, H, C + 1, d));
    tot += 5 * align_up((size_t)B * d * 4, 256);         // folded path: kq_topic, kq_user, r_user x2, r_news
    tot += digat_xattn_workspace_bytes(B, N, d);         // the news graph's own Eq. 8 workspace (side stream)
    // live-node and live-bucket counts, offsets, lists (int) and flags (bytes)
    tot += (2 * align_up((size_t)B, 64) + 2 * align_up((size_t)B + 1, 64) + align_up((size_t)B * U, 64)
            + align_up((size_t)B * (C + 1), 64)) * 4 + align_up((size_t)B * U, 256) + align_up((size_t)B * (C + 1), 256);
    return tot;
}

// row_group == NULL: user tensors are per row.  Otherwise they are per group (G of them) and row_group[b]
// names the group of row b; the caller (digat_encoder_fwd_grouped) has expanded the small per-row byte /
// index arrays, so Au / cat_mask / cat_idx are per row in both cases and only ue [G,H,d] is per group.
static int encoder_fwd_impl(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                            const float* ue, const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx,
                            const float* c_n0, float* out_news, float* out_user, int B, int N, int H,
                            void* workspace, size_t workspace_bytes, void* stream, const int* row_group, int G) {
    if (!p || !Xn_in || !An || !Mn || !ue || !Au || !cat_mask || !cat_idx || !out_news || !out_user || !workspace)
        return DIGAT_ERR_ARG;
    if (B < 0 || N <= 0 || H < 0) return DIGAT_ERR_ARG;
    const int d = p->d, C = p->category_num, L = p->depth, U = H + C;
    if (d <= 0 || d % 4 || L < 0 || L > DIGAT_MAX_DEPTH || N > DIGAT_MAX_NODES || U > DIGAT_MAX_NODES || C < 0)
        return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_encoder_workspace_bytes(B, N, H, C, d, L)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    hipStream_t st = (hipStream_t)stream;

    char* ws = (char*)workspace;
    const size_t su = align_up((size_t)B * U * d * 4, 256), sn = align_up((size_t)B * N * d * 4, 256);
    float* Xu[2] = {(float*)ws, (float*)(ws + su)};
    ws += 2 * su;
    float* Xn[2] = {(float*)ws, (float*)(ws + sn)};
    ws += 2 * sn;
    const int nmax = N > U ? N : U;
    void* xws = ws;
    const size_t xws_bytes = digat_xattn_workspace_bytes(B, nmax, d);
    ws += xws_bytes;
    void* cws = ws;
    const size_t cws_bytes = max_sz(digat_news_ctx_workspace_bytes(B, N, d), digat_user_ctx_workspace_bytes(B, U, H, C + 1, d));
    ws += cws_bytes;
    const size_t sb = align_up((size_t)B * d * 4, 256);
    float* kq_t = (float*)ws;
    float* kq_u = (float*)(ws + sb);
    float* r_user = (float*)(ws + 2 * sb);
    float* r_news = (float*)(ws + 3 * sb);
    float* const r_user2[2] = {r_user, (float*)(ws + 4 * sb)};
    void* xws_news = ws + 5 * sb;
    int* live_ws = (int*)((char*)xws_news + digat_xattn_workspace_bytes(B, N, d));

    int rc;
    // user graph nodes = [history | topic nodes]  (:191)
    {
        const long total4 = (long)B * U * (d / 4);
        int blocks = (int)((total4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        ProfScope prof(DIGAT_KERNEL_GLUE, (double)B * ((double)H * d * 8 + (double)C * d * 4), st);
        hipLaunchKernelGGL(build_user_nodes_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)ue,
                           (const float4*)p->topic_node_embedding, (float4*)Xu[0], (long)B, H, C, d / 4, row_group);
        DIGAT_CHECK_LAUNCH();
    }
    // c_n: given (inference, :189) or computed (forward, :180); it lives in out_news from here on
    if (c_n0) {
        if (hipMemcpyAsync(out_news, c_n0, (size_t)B * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return DIGAT_ERR_LAUNCH;
    } else {
        rc = digat_news_ctx_fwd(Xn_in, Mn, p->cand_K, p->cand_Q, p->cand_bQ, p->news_graph_W, p->news_graph_b,
                                nullptr, out_news, B, N, d, cws, cws_bytes, stream);
        if (rc) return rc;
    }
    if (p->cand_fold_W && p->user_news_fold_W && p->userAtt_fold_W)
        return encoder_fwd_folded(p, Xn_in, An, Mn, Au, cat_mask, cat_idx, out_news, out_user, B, N, H, Xu, Xn, xws,
                                  xws_news, cws, kq_t, kq_u, r_user2, r_news, live_ws, st, row_group, G, ue);
    // c_u (:192)
    rc = digat_user_ctx_fwd(Xu[0], cat_mask, cat_idx, out_news, p->user_news_K, p->user_news_Q, p->user_news_bQ,
                            p->featureAffine_W, p->featureAffine_b, p->userAtt_K, p->userAtt_Q, p->userAtt_bQ,
                            nullptr, out_user, B, U, H, C + 1, d, cws, cws_bytes, stream);
    if (rc) return rc;

    const float* xn_cur = Xn_in;
    int un = 0, nn = 0;
    for (int i = 0; i < L; ++i) {
        const digat_layer_params& ln = p->news[i];
        const digat_layer_params& lu = p->user[i];
        // both graph updates read the PREVIOUS contexts (:194-195)
        rc = launch_gemm(gemm_plain(out_user, d, ln.F3, ln.b3, r_news, d, B, d, d, 0), st);
        if (rc) return rc;
        rc = xattn_core(xn_cur, An, r_news, ln.W, ln.bW, ln.F1, ln.F2, ln.a, Xn[nn], nullptr, B, N, d, xws, st, ln.wsplit);
        if (rc) return rc;
        rc = launch_gemm(gemm_plain(out_news, d, lu.F3, lu.b3, r_user, d, B, d, d, 0), st);
        if (rc) return rc;
        rc = xattn_core(Xu[un], Au, r_user, lu.W, lu.bW, lu.F1, lu.F2, lu.a, Xu[un ^ 1], nullptr, B, U, d, xws, st, lu.wsplit);
        if (rc) return rc;
        xn_cur = Xn[nn]; nn ^= 1; un ^= 1;
        // c_n += news context (:196); c_u += user context with the UPDATED c_n (:197)
        rc = digat_news_ctx_fwd(xn_cur, Mn, p->cand_K, p->cand_Q, p->cand_bQ, p->news_graph_W, p->news_graph_b,
                                out_news, out_news, B, N, d, cws, cws_bytes, stream);
        if (rc) return rc;
        rc = digat_user_ctx_fwd(Xu[un], cat_mask, cat_idx, out_news, p->user_news_K, p->user_news_Q, p->user_news_bQ,
                                p->featureAffine_W, p->featureAffine_b, p->userAtt_K, p->userAtt_Q, p->userAtt_bQ,
                                out_user, out_user, B, U, H, C + 1, d, cws, cws_bytes, stream);
        if (rc) return rc;
    }
    return DIGAT_OK;
}

int digat_set_live_row_skipping(int enabled) {
    const int prev = g_live_rows_on;
    g_live_rows_on = enabled ? 1 : 0;
    return prev;
}

int digat_set_side_stream(int enabled) {
    const int prev = g_side_stream_on;
    g_side_stream_on = enabled ? 1 : 0;
    return prev;
}

int digat_encoder_fwd(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                      const float* ue, const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx,
                      const float* c_n0, float* out_news, float* out_user, int B, int N, int H,
                      void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_fwd_impl(p, Xn_in, An, Mn, ue, Au, cat_mask, cat_idx, c_n0, out_news, out_user, B, N, H, workspace,
                            workspace_bytes, stream, nullptr, 0);
}

size_t digat_encoder_grouped_workspace_bytes(int B, int N, int H, int C, int d, int depth) {
    const int U = H + C;
    return digat_encoder_workspace_bytes(B, N, H, C, d, depth) + align_up((size_t)B * U * U, 256)
           + align_up((size_t)B * (C + 1), 256) + align_up((size_t)B * H * 8, 256);
}

int digat_encoder_fwd_grouped(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                              const float* ue_g, const uint8_t* Au_g, const uint8_t* cat_mask_g, const int64_t* cat_idx_g,
                              const int32_t* row_group, const float* c_n0, float* out_news, float* out_user,
                              int B, int G, int N, int H, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !ue_g || !Au_g || !cat_mask_g || !cat_idx_g || !row_group || !workspace || G <= 0) return DIGAT_ERR_ARG;
    const int d = p->d, C = p->category_num, U = H + C;
    if (!p->cand_fold_W || !p->user_news_fold_W || !p->userAtt_fold_W) return DIGAT_ERR_ARG;   // grouped = folded path
    if ((size_t)4 * G > (size_t)B) return DIGAT_ERR_SHAPE;       // the group-level projections reuse one [B,U,d] buffer
    if (workspace_bytes < digat_encoder_grouped_workspace_bytes(B, N, H, C, d, p->depth)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    hipStream_t st = (hipStream_t)stream;
    // expand the small per-user byte / index arrays to rows (4.6 MB for 1024 rows of 67x67 adjacency)
    const size_t base = digat_encoder_workspace_bytes(B, N, H, C, d, p->depth);
    uint8_t* Au = (uint8_t*)workspace + base;
    uint8_t* cm = Au + align_up((size_t)B * U * U, 256);
    uint8_t* ci = cm + align_up((size_t)B * (C + 1), 256);
    struct { const uint8_t* in; uint8_t* out; long bytes; } jobs[3] = {
        {Au_g, Au, (long)U * U}, {cat_mask_g, cm, (long)(C + 1)}, {(const uint8_t*)cat_idx_g, ci, (long)H * 8}};
    for (auto& j : jobs) {
        const long total = (long)B * j.bytes;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, st, j.in, j.out, row_group, (long)B, j.bytes);
        DIGAT_CHECK_LAUNCH();
    }
    return encoder_fwd_impl(p, Xn_in, An, Mn, ue_g, Au, cm, (const int64_t*)ci, c_n0, out_news, out_user, B, N, H, workspace,
                            base, stream, row_group, G);
}

static double g_prof_last_live_fraction = -1.0;
double digat_profile_live_row_fraction(void) { return g_prof_last_live_fraction; }

int digat_profile_start(int max_launches) {
    if (max_launches <= 0) return DIGAT_ERR_ARG;
    if (g_prof.ev) return DIGAT_ERR_ARG;          // already running
    g_prof.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * max_launches);
    g_prof.kind = (int*)malloc(sizeof(int) * max_launches);
    g_prof.work = (double*)malloc(sizeof(double) * max_launches);
    if (!g_prof.ev || !g_prof.kind || !g_prof.work) return DIGAT_ERR_ARG;
    for (int i = 0; i < 2 * max_launches; ++i)
        if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return DIGAT_ERR_LAUNCH;
    if (hipMalloc((void**)&g_prof.rows_dev, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(g_prof.rows_dev, 0, 16 * sizeof(unsigned long long)) != hipSuccess) return DIGAT_ERR_LAUNCH;
    for (int k = 0; k < 16; ++k) { g_prof.flops_per_row[k] = 0.0; g_prof.rows_nominal[k] = 0.0; }
    g_prof.cap = max_launches; g_prof.used = 0; g_prof.enabled = 1;
    return DIGAT_OK;
}

int digat_profile_stop(double* ms_per_kind, double* work_per_kind, int* launches_per_kind) {
    if (!g_prof.ev) return DIGAT_ERR_ARG;
    g_prof.enabled = 0;
    for (int k = 0; k < DIGAT_KERNEL_KINDS; ++k) {
        if (ms_per_kind) ms_per_kind[k] = 0;
        if (work_per_kind) work_per_kind[k] = 0;
        if (launches_per_kind) launches_per_kind[k] = 0;
    }
    int rc = DIGAT_OK;
    for (int i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) { rc = DIGAT_ERR_LAUNCH; continue; }
        const int k = g_prof.kind[i];
        if (ms_per_kind) ms_per_kind[k] += ms;
        if (work_per_kind) work_per_kind[k] += g_prof.work[i];
        if (launches_per_kind) launches_per_kind[k] += 1;
    }
    g_prof_last_live_fraction = -1.0;
    if (g_prof.rows_dev) {
        unsigned long long rows[16];
        if (hipMemcpy(rows, g_prof.rows_dev, sizeof(rows), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int k = 0; k < DIGAT_KERNEL_KINDS; ++k)
                if (work_per_kind) work_per_kind[k] += (double)rows[k] * g_prof.flops_per_row[k];
            if (g_prof.rows_nominal[DIGAT_KERNEL_PROJ] > 0)
                g_prof_last_live_fraction = (double)rows[DIGAT_KERNEL_PROJ] / g_prof.rows_nominal[DIGAT_KERNEL_PROJ];
        }
        hipFree(g_prof.rows_dev);
        g_prof.rows_dev = nullptr;
    }
    for (int i = 0; i < 2 * g_prof.cap; ++i) hipEventDestroy(g_prof.ev[i]);
    free(g_prof.ev); free(g_prof.kind); free(g_prof.work);
    g_prof.ev = nullptr; g_prof.kind = nullptr; g_prof.work = nullptr; g_prof.cap = g_prof.used = 0;
    return rc;
}

int digat_row_logits(const float* news_ctx, const float* user_ctx, float* logits, int B, int d, void* stream) {
    if (!news_ctx || !user_ctx || !logits || B < 0 || d <= 0) return DIGAT_ERR_ARG;
    if (B == 0) return DIGAT_OK;
    hipLaunchKernelGGL(row_logits_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, news_ctx, user_ctx,
                       logits, B, d);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

}  // extern "C"

#include "digat_train.inc"
#include "digat_eval.inc"
