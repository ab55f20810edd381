// digat_kernels.hip — hand-written gfx950 (MI355X / CDNA4) kernels for DIGAT's dual-graph
// interaction hot path, and the C ABI declared in include/digat_hip.h.
//
// One translation unit; the kernels live in the .inc files next to this one (DESIGN.md has the data layout and
// the roofline of each):
//   digat_gemm.inc     nn.Linear on the matrix cores: exact fp32 (v_mfma_f32_16x16x4_f32) with fused epilogues, the
//                      strip-mined "bf16x6" kernel of the node projections (LDS-DMA operands, optional row list),
//                      the skinny [B,d] linears
//   digat_xattn.inc    Eq. 8: relu(K3+K1+K2).a -> leaky_relu -> -1e9 mask -> softmax_j -> relu(alpha @ h) + X,
//                      never materialising [B,n,n,d]
//   digat_context.inc  ScaledDotProductAttention pooling (key projection folded into the query); torch_scatter's
//                      scatter_softmax + scatter_sum over history categories
//   digat_glue.inc     user-node build, group expansion, live-row lists, row logits
//   digat_train.inc    backward / training kernels;  digat_eval.inc  per-impression ranking + metrics
//   digat_news.inc     MSA news encoder (inference);  digat_gat.inc  vanilla-GAT layer of the ablation encoders
// This file: shared helpers, the per-kernel profiler, and the C ABI (encoder orchestration included).
//
// gfx950 only: 64-wide wavefronts, 160 KiB LDS per CU, MFMA f32 16x16x4.  No CUDA shims.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <unordered_map>

#include "../../include/digat_hip.h"

typedef float v4f __attribute__((ext_vector_type(4)));

#define DIGAT_CHECK_LAUNCH()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return DIGAT_ERR_LAUNCH; \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Development knobs read from the environment — A/B switches between two kernels that give the same results, timing ablations
// that give WRONG results (DIGAT_*_SKIP), phase timers — exist in LAB builds only (-DDIGAT_LAB: tools/exp/build_variant.sh).  In
// the product library LAB_ENV is its default and the variable's name is not even in the binary: no environment variable can
// change what a scoring run computes (tests/test_abi_cpu.py looks for the names).
#ifdef DIGAT_LAB
static int lab_env_value(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define LAB_ENV(name, dflt) lab_env_value(name, dflt)
#else
#define LAB_ENV(name, dflt) (dflt)
#endif

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
    unsigned kind_mask;                // only launches of these kinds are bracketed (digat_profile_set_kinds)
    double* bytes;                     // MFMA kinds: the operand + result bytes of the launch (digat_profile_gemm_bytes)
    double bytes_per_row[16];
    int* part;                         // DIGAT_KERNEL_XATTN launches: which Eq. 8 kernel (XPART_*; digat_profile_xattn_parts)
} g_prof = {0, 0, 0, nullptr, nullptr, nullptr, nullptr, {0.0}, {0.0}, ~0u, nullptr, {0.0}, nullptr};

// The Eq. 8 launches are three unlike kernels with their own byte budgets: they are timed and priced apart (bench.py:
// roofline_xattn.parts).  Device-side byte counts of a part (live-row lists: known on the device only) go to rows_dev[8 + part].
enum { XPART_TWIN = 0,      // user graph, layers >= 1: row-list launches (twin kernel / wave per live centre)
       XPART_L0 = 1,        // user graph, layer 0 of grouped rows (chunk kernel / wave per live centre through the group index)
       XPART_NEWS = 2,      // news graphs of <= 16 nodes, the graph in LDS (fused)
       XPART_OTHER = 3,     // everything else: dense score launches, larger news graphs on the sparse kernel
       XATTN_PARTS = 4 };

struct ProfScope {
    hipStream_t st; int slot;
    ProfScope(int kind, double work, hipStream_t s, double bytes = 0.0, int part = XPART_OTHER) : st(s), slot(-1) {
        if (g_prof.enabled && ((g_prof.kind_mask >> kind) & 1u) && g_prof.used < g_prof.cap) {
            slot = g_prof.used++;
            g_prof.kind[slot] = kind; g_prof.work[slot] = work; g_prof.bytes[slot] = bytes; g_prof.part[slot] = part;
            (void)hipEventRecord(g_prof.ev[2 * slot], st);
        }
    }
    ~ProfScope() { if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st); }
};

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float f4_comp(const float4& v, int s) {
    return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w));
}
// Wave-wide reductions, result in every lane.  Within a row of 16 lanes the operands move by DPP (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror: VALU modifiers, no LDS crossbar round trip as __shfl_xor / ds_bpermute has); the
// four row results are read back with v_readlane and combined.  Fixed order -> deterministic; every lane sees the same
// bits (each step adds the same two values in both partners; float + is commutative).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// Round 5: the four row sums S0..S3 meet by two more DPP steps instead of four v_readlane + two v_mov + three v_add on uniform
// values (9 vector instructions -> 3; the Eq. 8 kernels are vector-issue-bound and reduce once per (centre, neighbour) pair):
// row_bcast:15 adds lane 15 of row k to every lane of row k + 1 (rows 1 and 3: S0 + S1, S2 + S3), row_bcast:31 adds lane 31
// (= S0 + S1) to rows 2 and 3, so lane 63 holds (S3 + S2) + (S1 + S0) — THE BITS of the old (S0 + S1) + (S2 + S3): float + is
// commutative, the association is the same.  Disabled rows keep their value (row_mask).  The s_nop's are the DPP read-after-write
// wait states hipcc would insert itself for a builtin (inline asm is not padded: cdna_hip_programming.md section 5.7).
#ifndef DIGAT_WAVE_REDUCE_BCAST
#define DIGAT_WAVE_REDUCE_BCAST 1
#endif
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);      // row_half_mirror
    v += dpp_mov<0x140>(v);      // row_mirror: every lane holds the sum of its row of 16
#if DIGAT_WAVE_REDUCE_BCAST
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return lane_bcast(v, 63);
#else
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
#endif
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
#if DIGAT_WAVE_REDUCE_BCAST
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return lane_bcast(v, 63);
#else
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
#endif
}

// one 16-byte-per-lane global -> LDS copy; LDS address = lds_byte_addr (wave-uniform) + 16*lane.
// Invisible to hipcc's waitcnt bookkeeping: completion is counted by hand (wait_vmcnt below).
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// The same copy with the source as (wave-uniform 64-bit base in SGPRs) + (32-bit byte offset per lane): no 64-bit vector add per
// piece, and M0 is written but neither saved nor restored (round 5: the GEMM's main loop is bound by instruction ISSUE — 1.5
// non-MFMA instructions per MFMA — and a third of those were this helper's M0 save / restore and its callers' address adds).
// Only for kernels in which nothing else depends on M0 (gfx9 LDS instructions do not; the ISA of every user is checked for it).
__device__ __forceinline__ void lds_dma16_s(const void* sbase_any, unsigned voff, unsigned lds_byte_addr) {
    // the base IS wave-uniform; where the compiler cannot prove it (a row count loaded from memory) it is made so explicitly
    const unsigned long long bits = (unsigned long long)(uintptr_t)sbase_any;
    const void* sbase = (const void*)(uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bits >> 32)) << 32) |
                                                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)bits));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {        // s_waitcnt vmcnt(N) with a compile-time N (0 .. 63)
    // gfx9 encoding of the s_waitcnt immediate: vmcnt low bits [3:0], expcnt [6:4] = 7, lgkmcnt [11:8] = 15, vmcnt high bits [15:14]
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void wait_vmcnt(int n) {      // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;      // callers pass 0 .. 6; anything else waits for more than asked
    }
}

// Round 5: TWO twins per wave at five waves per SIMD (94 registers) instead of four at four (114): 280 against 305 us per 4 096-row
// launch alone, 460-510 against 535-570 inside the overlapped region (tools/exp/ab.py, alternating runs; three at five waves spill)
#ifndef DIGAT_TWIN_R
#define DIGAT_TWIN_R 2
#endif
// leaky_relu(0.2) of a WAVE-UNIFORM score (a wave_sum result): e > 0 ? e : 0.2 e = max(e, 0.2 e), bit for bit (also for -0, inf, NaN),
// as one multiply and one v_max_f32 with the score as the scalar operand (fmaxf costs a third instruction that canonicalises e)
__device__ __forceinline__ float leaky02_uniform(float e) {
    const float t = 0.2f * e;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(e), "v"(t));
    return r;
}
// Dropout of the training path: a counter-based hash of (seed, flat element index) against floor(p 2^32) — every kernel that applies a
// dropout (dropout_fwd_kernel and the fused sites: the Eq. 8 scores, the gate, the pooled topics) draws element e's bit from here,
// so a fused site lands on the elements the stand-alone launch would have (oracle/digat_oracle.py restates it for the tests)
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned drop_threshold(float p) { return (unsigned)(p * 4294967296.0); }
__device__ __forceinline__ bool drop_keep(unsigned seed, long e, unsigned thr) {
    return hash32((unsigned)e * 0x9E3779B9U + hash32(seed + (unsigned)(e >> 32))) >= thr;
}
constexpr int TWIN_R = DIGAT_TWIN_R;   // centres with equal adjacency rows served by one wave (user_live_flags_kernel, xattn_sparse_twin_kernel)
#include "digat_gemm.inc"
#include "digat_xattn.inc"
#include "digat_context.inc"
#include "digat_ctxfused.inc"
#include "digat_glue.inc"
#ifdef DIGAT_LAB
#include "digat_staged.inc"       // Eq. 8 of the user graph from LDS-staged rows: five variants, all measured slower (docs/REJECTED.md, row 7)
#else
// the product library carries the evidence (profiles/), not the code path
struct PlanBuffers {};
static size_t plan_bytes(int, int) { return 0; }
static bool staged_ok(int, int) { return false; }
static PlanBuffers plan_carve(void*, int, int) { return PlanBuffers{}; }
static int launch_staged(const SparseArgs&, const PlanBuffers&, int, int, hipStream_t) { return DIGAT_ERR_ARG; }
static int launch_plan(const uint8_t*, const uint8_t*, const int64_t*, const int*, int, int, int, int, int, int, bool, const PlanBuffers&, hipStream_t) { return DIGAT_ERR_ARG; }
#endif

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

struct TwinLists { const unsigned* word; const int* list; const int* count; };     // user_live_flags_kernel's twins (see there)
// pre-split node rows between two Eq. 8 layers (fp16x3 format): `in` = the rows of X as the previous layer's Eq. 8 kernel stored
// them for this layer's projection GEMM (GemmArgs.a_split), `out` = where this layer's Eq. 8 kernel stores its output rows for the next
struct SplitIO { const void* in; unsigned char* out; unsigned* range; };

// Eq. 8 layer with K3 (r = ctx F3^T + b3) already computed; `r_given` may live anywhere
static int xattn_core(const float* X, const uint8_t* A, const float* r_given,
                      const float* W, const float* bW, const float* F1, const float* F2, const float* a,
                      float* out, float* alpha_out, int B, int n, int d, void* workspace, hipStream_t st,
                      const void* wsplit = nullptr, const int* rowidx = nullptr, const int* nrows_dev = nullptr,
                      const uint8_t* live = nullptr, int sparse_mode = DIGAT_XATTN_DENSE, const int* sparse_flag = nullptr,
                      int pq_x3 = 0, const PlanBuffers* plan = nullptr, int plan_slot = 0, int pq_mode = 0, int centre_limit = 0,
                      int gemm_format = 0, unsigned* range_flag = nullptr, const TwinLists* tw = nullptr, const SplitIO* sio = nullptr,
                      int prof_part = 0) {
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
    g.wsplit = (const unsigned short*)wsplit;          // non-NULL: split operands on the bf16 / fp16 matrix cores
    g.format = gemm_format; g.range_flag = range_flag;
    g.radd = r_given; g.radd_seg = 1; g.rows_per_b = n; // P' = K3 + K1: the reference's left-to-right order
    g.x3_segs = pq_x3 ? 6 : 0;                          // DIGAT_PROJ_PQ_X3: P and Q (segments 1, 2) with three products
    // DIGAT_PQ_BF16 (pq_mode & 1): P' and Q stored in bf16, read by the wave-per-centre sparse kernel; & 2: one product for them
    const bool pq16 = (pq_mode & 1) && gemm_is_bf16x6(g) && sparse_mode == DIGAT_XATTN_SPARSE && !alpha_out && !plan && n > 16 &&
                      d / 4 <= 256 && d % 8 == 0 && (long)B * n >= 2048;
    // DIGAT_PQ_FP8 (pq_mode & 4): P' and Q stored as block-scaled e4m3 rows (one fp32 scale per 80-channel strip), same reader
    const bool pq8 = (pq_mode & 4) && !(pq_mode & 1) && gemm_is_bf16x6(g) && sparse_mode == DIGAT_XATTN_SPARSE && !alpha_out && !plan && n > 16 &&
                     d / 4 <= 256 && d % 80 == 0 && (long)B * n >= 2048;
    const long ld8 = (long)align_up((size_t)d + 4 * (size_t)(d / 80), 64);      // [d codes | d / 80 scales | pad]: whole 64-byte lines
    if (pq16) { g.bf16_segs = 6; if (pq_mode & 2) g.x1_segs = 6; }
    if (pq8) { g.fp8_segs = 6; g.ldy8 = ld8; if (pq_mode & 2) g.x1_segs = 6; }
    const bool listed = rowidx && gemm_is_bf16x6(g);
    if (listed) { g.rowidx = rowidx; g.nrows_dev = nrows_dev; }                         // live rows only (see user_live_flags_kernel)
    if (sio && sio->in && listed && gemm_format == 1 && d % 8 == 0) g.a_split = sio->in;    // the rows arrive split: no operand split in the GEMM
    const int rc = launch_gemm(g, st, DIGAT_KERNEL_PROJ);
    if (rc) return rc;
    const int* skip_if = nullptr;
    if (sparse_mode != DIGAT_XATTN_DENSE && !alpha_out && n > 16 && d / 4 <= 256) {      // see xattn_sparse_kernel
        SparseArgs sg{P, Q, h, X, a, A, out, nullptr, nullptr, listed ? live : nullptr,
                      sparse_mode == DIGAT_XATTN_AUTO ? sparse_flag : nullptr, B, n, d / 4, 0, nullptr, nullptr,
                      listed && live ? rowidx : nullptr, listed && live ? nrows_dev : nullptr, 0, nullptr, pq8 ? 2 : (pq16 ? 1 : 0), plan ? 0 : centre_limit};
        sg.ld8 = pq8 ? ld8 : 0;
        sg.prof_part = prof_part;
        if (tw && listed && live && !plan) { sg.twin = tw->word; sg.twlist = tw->list; sg.twcount = tw->count; }
        if (sio && sio->out && listed && live && !plan && sparse_mode == DIGAT_XATTN_SPARSE) { sg.xsplit = sio->out; sg.xsplit_range = sio->range; }
        // with a plan of the batch (encoder entry points): the LDS-staged kernel, each needed row read once (digat_staged.inc)
        const int rcs = plan ? launch_staged(sg, *plan, listed && live ? 1 : 0, plan_slot, st) : launch_sparse(sg, st);
        if (rcs || sparse_mode == DIGAT_XATTN_SPARSE) return rcs;
        skip_if = sparse_flag;
    }
    return launch_xattn_pairwise(P, Q, h, X, a, A, out, alpha, B, n, d, st, listed ? live : nullptr, nullptr, nullptr,
                                 alpha_out != nullptr, skip_if);
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

int digat_xattn_fwd_mode(const float* X, const uint8_t* A, const float* ctx,
                         const float* W, const float* bW, const float* F1, const float* F2,
                         const float* F3, const float* b3, const float* a,
                         float* out, int B, int n, int d, int mode,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (mode != DIGAT_XATTN_DENSE && mode != DIGAT_XATTN_SPARSE) return DIGAT_ERR_ARG;
    if (!X || !A || !ctx || !W || !F1 || !F2 || !F3 || !a || !out || !workspace) return DIGAT_ERR_ARG;
    if (B < 0 || n <= 0 || d <= 0) return DIGAT_ERR_ARG;
    if (d % 4 || n > DIGAT_MAX_NODES) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_xattn_workspace_bytes(B, n, d)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    hipStream_t st = (hipStream_t)stream;
    float* r = (float*)((char*)workspace + align_up((size_t)3 * B * n * d * 4, 256));
    const int rc = launch_gemm(gemm_plain(ctx, d, F3, b3, r, d, B, d, d, 0), st);
    if (rc) return rc;
    return xattn_core(X, A, r, W, bW, F1, F2, a, out, nullptr, B, n, d, workspace, st, nullptr, nullptr, nullptr, nullptr, mode);
}

int digat_xattn_fwd_lowprec(const float* X, const uint8_t* A, const float* ctx,
                            const float* W, const float* bW, const float* F1, const float* F2,
                            const float* F3, const float* b3, const float* a, const void* wsplit, int format,
                            float* out, int B, int n, int d, int pq,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (pq < 0 || pq > 2 || (format != 0 && format != 1)) return DIGAT_ERR_ARG;
    if (!X || !A || !ctx || !W || !F1 || !F2 || !F3 || !a || !wsplit || !out || !workspace) return DIGAT_ERR_ARG;
    if (B < 0 || n <= 0 || d <= 0) return DIGAT_ERR_ARG;
    if (d % 80 || d > 1024 || n <= 16 || n > DIGAT_MAX_NODES || (long)B * n < 2048) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_xattn_workspace_bytes(B, n, d)) return DIGAT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* r = (float*)((char*)workspace + align_up((size_t)3 * B * n * d * 4, 256));
    const int rc = launch_gemm(gemm_plain(ctx, d, F3, b3, r, d, B, d, d, 0), st);
    if (rc) return rc;
    return xattn_core(X, A, r, W, bW, F1, F2, a, out, nullptr, B, n, d, workspace, st, wsplit, nullptr, nullptr, nullptr, DIGAT_XATTN_SPARSE,
                      nullptr, 0, nullptr, 0, pq == 1 ? 1 : (pq == 2 ? 4 : 0), 0, format, nullptr);
}

// ---- bf16x6 weight preparation + a directly callable linear (tests, micro-benchmarks) --------------
size_t digat_split_weights_bytes(int rows, int K) {
    return (size_t)((rows + 79) / 80) * ((K + 31) / 32) * WS_SLOTS * 16;       // one 15 KB image per (80-row strip, K tile)
}

// A training entry that was handed a ready-made image of its weights (digat_split_jobs: every image of a step in one launch) passes
// the image where its helpers expect their split destination and names it here: the split launch for exactly that pointer is skipped.
static thread_local const void* tl_premade_image = nullptr;
struct PremadeImage {
    const void* prev;
    explicit PremadeImage(const void* image) : prev(tl_premade_image) { tl_premade_image = image; }
    ~PremadeImage() { tl_premade_image = prev; }
};
// format: DIGAT_GEMM_BF16X6 (three bf16 pieces; what every training entry uses) or DIGAT_GEMM_F16X3 (two scaled fp16 pieces)
static int launch_split(const float* w0, const float* w1, const float* w2, int nseg, int nsegs, int K, void* wsplit, hipStream_t st,
                        int transposed = 0, int format = 0) {
    if (format != 0 && format != 1) return DIGAT_ERR_ARG;
#ifdef DIGAT_LAB
    if (wsplit && wsplit == tl_premade_image && format == 0) return DIGAT_OK;      // (LAB, DIGAT_TRAIN_F16: an fp16x3 image is split over the ready-made one)
#else
    if (wsplit && wsplit == tl_premade_image) return format == 0 ? DIGAT_OK : DIGAT_ERR_ARG;
#endif
    const long total = (long)nseg * nsegs * K;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    wsplit_note(wsplit, format);
    hipLaunchKernelGGL(split_weights_tiled_kernel, dim3(blocks), dim3(256), 0, st, w0, w1, w2, nseg, nsegs, K, (unsigned short*)wsplit, transposed,
                       format == 1 ? 2 : 3);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

// every split image of a training step in ONE launch (blockIdx.y = job): 20 split launches per 64 x 5-row step before
size_t digat_split_job_bytes(int rows, int cols, int layout, int matrices) {
    if (rows <= 0 || cols <= 0 || (matrices != 1 && matrices != 3)) return 0;
    if (layout == 0) return digat_split_weights_bytes(rows * matrices, cols);
    if (layout == 1) return matrices == 1 ? digat_split_weights_bytes(cols, rows) : digat_split_weights_bytes(cols, 3 * rows);
    return 0;
}
int digat_split_jobs(const digat_split_job* jobs, int njobs, void* stream) {
    if (!jobs || njobs < 0 || njobs > SPLIT_MAX_JOBS) return DIGAT_ERR_ARG;
    if (njobs == 0) return DIGAT_OK;
    SplitJobsDev all;
    memset(&all, 0, sizeof(all));
    long most = 0;
    for (int k = 0; k < njobs; ++k) {
        const digat_split_job& j = jobs[k];
        const bool three = j.w1 != nullptr || j.w2 != nullptr;
        if (!j.w0 || !j.image || j.rows <= 0 || j.cols <= 0 || (three && (!j.w1 || !j.w2)) || (j.layout != 0 && j.layout != 1)) return DIGAT_ERR_ARG;
        SplitJobDev& o = all.j[k];
        o.w0 = j.w0; o.w1 = three ? j.w1 : j.w0; o.w2 = three ? j.w2 : j.w0; o.out = (unsigned short*)j.image;
        if (j.layout == 0) { o.nseg = j.rows; o.nsegs = three ? 3 : 1; o.K = j.cols; o.transposed = 0; }
        else { o.nseg = j.cols; o.nsegs = 1; o.K = three ? 3 * j.rows : j.rows; o.transposed = three ? 2 : 1; }
        const long total = (long)((o.nseg * o.nsegs + 79) / 80) * 80 * ((o.K + 31) / 32) * 4;      // 16-byte slots of the image's three planes / 3
        if (total > most) most = total;
        wsplit_note(j.image, 0);
    }
    long bx = (most + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(split_weights_jobs_kernel, dim3((unsigned)bx, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, all);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

int digat_forget_split_image(const void* wsplit) { return wsplit && wsplit_forget(wsplit) ? DIGAT_OK : DIGAT_ERR_ARG; }

int digat_gather_tables(const digat_gather_job* jobs, int njobs, void* stream) {
    if (!jobs || njobs < 0 || njobs > GATHER_MAX_JOBS) return DIGAT_ERR_ARG;
    if (njobs == 0) return DIGAT_OK;
    GatherTableJobs all;
    memset(&all, 0, sizeof(all));
    long most = 0;
    for (int k = 0; k < njobs; ++k) {
        const digat_gather_job& j = jobs[k];
        if (j.rows < 0 || j.row_bytes <= 0 || (j.rows > 0 && (!j.src || !j.dst || !j.idx)) || (j.idx2 && j.inner <= 0)) return DIGAT_ERR_ARG;
        all.job[k] = j;
        if (!j.idx2) all.job[k].inner = 1;
        const long units = (j.row_bytes & 15) == 0 ? j.rows * (j.row_bytes >> 4) : j.rows * j.row_bytes;
        if (units > most) most = units;
    }
    if (most == 0) return DIGAT_OK;
    long bx = (most + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(gather_tables_kernel, dim3((unsigned)bx, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, all);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

int digat_user_row_runs(const float* ue, const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx, int B, int H, int U, int C1, int d,
                        int32_t* row_group, int64_t* leaders, int32_t* n_runs, void* workspace, size_t workspace_bytes, void* stream) {
    if (!ue || !Au || !cat_mask || !cat_idx || !row_group || !leaders || !n_runs || !workspace || B <= 0 || H < 0 || U <= 0 || C1 <= 0 || d <= 0)
        return DIGAT_ERR_ARG;
    if (d % 4 || ((long)H * d) % 4) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < (size_t)B) return DIGAT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint8_t* same = (uint8_t*)workspace;
    ProfScope prof(DIGAT_KERNEL_GLUE, (double)B * ((double)H * d * 4 + (double)U * U + C1 + 8.0 * H), st);
    hipLaunchKernelGGL(user_rows_same_kernel, dim3(B), dim3(256), 0, st, (const uint4*)ue, Au, cat_mask, cat_idx, B, (long)H * d / 4, U * U, C1, H, same);
    DIGAT_CHECK_LAUNCH();
    hipLaunchKernelGGL(user_row_runs_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)same, B, row_group, leaders, n_runs);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

int digat_split_proj_weights(const float* W, const float* F1, const float* F2, int d, void* wsplit, int format, void* stream) {
    if (!W || !F1 || !F2 || !wsplit || d <= 0) return DIGAT_ERR_ARG;
    return launch_split(W, F1, F2, d, 3, d, wsplit, (hipStream_t)stream, 0, format);
}

size_t digat_split_ctx_fused_bytes(int d) { return d > 0 ? ctxfused_image_bytes(d) : 0; }
int digat_split_ctx_fused_weights(const float* W, int d, void* image, void* stream) {
    if (!W || !image || d <= 0) return DIGAT_ERR_ARG;
    if (d % 8) return DIGAT_ERR_SHAPE;
    const long total = (long)(ctxfused_image_bytes(d) / 16);
    hipLaunchKernelGGL(split_weights_ctxfused_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, d, (uint4*)image);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

int digat_split_weights(const float* W, int N, int K, void* wsplit, int format, void* stream) {
    if (!W || !wsplit || N <= 0 || K <= 0) return DIGAT_ERR_ARG;
    return launch_split(W, W, W, N, 1, K, wsplit, (hipStream_t)stream, 0, format);
}

#ifdef DIGAT_CF_TIMERS
extern "C" int digat_debug_cf_timers(double* out16) {
    unsigned long long h[16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cf_timers), sizeof(h)) != hipSuccess) return DIGAT_ERR_LAUNCH;
    for (int k = 0; k < 16; ++k) { out16[k] = (double)h[k]; h[k] = 0; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cf_timers), h, sizeof(h)) == hipSuccess ? DIGAT_OK : DIGAT_ERR_LAUNCH;
}
#endif
#ifdef DIGAT_GEMM_TIMERS
int digat_debug_gemm_timers(double* out8) {
    unsigned long long h[8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_timers), sizeof(h)) != hipSuccess) return DIGAT_ERR_LAUNCH;
    for (int k = 0; k < 8; ++k) { out8[k] = (double)h[k]; h[k] = 0; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_timers), h, sizeof(h)) == hipSuccess ? DIGAT_OK : DIGAT_ERR_LAUNCH;
}
#endif

// BASELINE configs[4], training half: the >= 2048-row GEMMs of the training path (projections, featureAffine, input
// gradients) with ONE bf16 product per fp32 product — plain bf16 mixed precision: fp32 master weights and activations, bf16
// matrix-core operands, fp32 accumulation — instead of the six of the fp32-grade split (the >= 2048-row weight gradients included: gemm_tn_bf16x6_kernel<true>).
static std::atomic<int> g_train_bf16{0};      // set once per training run (Trainer.__init__); read by every training GEMM
int digat_set_train_precision(int bf16) { return g_train_bf16.exchange(bf16 ? 1 : 0); }

int digat_linear_f32x3(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy,
                       int M, int N, int K, void* wsplit, int format, void* stream) {
    if (!x || !w || !y || !wsplit || M < 0 || N <= 0 || K <= 0) return DIGAT_ERR_ARG;
    if (K % 8 || ldx % 4 || N % 80) return DIGAT_ERR_SHAPE;
    const int rcs = launch_split(w, w, w, N, 1, K, wsplit, (hipStream_t)stream, 0, format);
    if (rcs) return rcs;
    GemmArgs g = gemm_plain(x, ldx, w, b, y, ldy, M, N, K, 0);
    g.wsplit = (const unsigned short*)wsplit; g.format = format;
    if (g_train_bf16) g.x1_segs = 7;
    // M >= 2048: the strip-mined kernel; below: the skinny kernel on the same split image (gemm_skinny_split_kernel)
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
static size_t l0_chunk_bytes(int B, int U);      // the extra lists of the folded path (defined with the workspace sizes below)
// group-level outputs of user_live_flags_kernel (grouped entries: at most B / 4 groups), expanded to rows by live_expand_kernel
static size_t live_group_bytes(int B, int U, int C1) {
    const size_t Gm = (size_t)B / 4 + 1;
    return 2 * align_up(Gm * U, 256) + align_up(Gm * U * 4, 256) + align_up(Gm * C1, 256) + 5 * align_up(Gm * 4, 256);
}
// live nodes of the news graphs (news_live_flags_kernel): cnt [B], off [B + 1], list [B N] (int) and flags [B N] (bytes)
static size_t news_live_bytes(int B, int N) {
    return (align_up((size_t)B, 64) + align_up((size_t)B + 1, 64) + align_up((size_t)B * N, 64)) * 4 + align_up((size_t)B * N, 256);
}
struct SideStream { hipStream_t s; hipEvent_t fork, join, early; int ok; };
// Nothing below is mutable: live-row lists and the side stream are chosen PER CALL through digat_params.flags
// (DIGAT_PARAMS_NO_LIVE_ROWS, DIGAT_PARAMS_SIDE_STREAM_OFF / _ON), so two host threads with different settings cannot flip each
// other's (round 3 had process-wide setters for them).
static const int g_sparse_per_node = LAB_ENV("DIGAT_SPARSE_PER_NODE", 20);
// 0 = never, 1 = always, 2 = by pass size (default): below 2 048 rows — there the news kernels are a few waves of workgroups each;
// from 2 048 rows up every kernel fills the chip by itself and the second stream only makes launches share it (4 096 rows,
// three passes in flight: 3.21 vs 3.28 ms per pass; stress 16.4 vs 16.8, MIND-large shape 4.52 vs 4.65).  DIGAT_SINGLE_STREAM=1 / 0
// forces never / always.
static const int g_side_stream_lab = LAB_ENV("DIGAT_SINGLE_STREAM", -1);       // LAB builds: 1 / 0 force never / always
// layer 0 of the user graph on the live nodes only (A/B switch for measurements; DIGAT_L0_LIVE=0: every node at layer 0)
static const int g_l0_live_on = LAB_ENV("DIGAT_L0_LIVE", 1);
// 0 (default): the wave-per-centre sparse kernel; 1: the LDS-staged kernels of digat_staged.inc (compulsory HBM traffic, measured
// slower in round 2: DESIGN.md section 4)
#ifdef DIGAT_LAB
static int g_staged_on = LAB_ENV("DIGAT_XATTN_STAGED", 0);
#else
static constexpr int g_staged_on = 0;
#endif
// One side stream (and its three events) per CALLER stream: consecutive batches issued on alternating caller streams
// (util.batch_streams) then overlap their side work too, and two host threads driving two streams never touch the same
// events.  A caller stream is expected to be driven by one thread at a time (include/digat_hip.h, threading contract); the
// table itself is guarded by a mutex.  Entries live for the life of the process (streams are few and long-lived).
static SideStream* side_stream(hipStream_t caller) {
    struct Entry { int dev; hipStream_t caller; SideStream side; int state; };     // state: 1 = ready, -1 = unavailable
    static Entry tab[64];
    static int used = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used; ++i)
        if (tab[i].dev == dev && tab[i].caller == caller) return tab[i].state == 1 ? &tab[i].side : nullptr;
    if (used == 64) return nullptr;            // more caller streams than anyone has: those run single-stream
    Entry& e = tab[used++];
    e.dev = dev; e.caller = caller;
    SideStream& x = e.side;
    // LAB builds: DIGAT_SIDE_PRIO=1 makes the side stream a high-priority queue (its small news-side kernels are dispatched ahead of the
    // caller stream's long-running Eq. 8 / GEMM workgroups whenever a slot frees up)
    static const int side_prio = LAB_ENV("DIGAT_SIDE_PRIO", 0);
    int prio_lo = 0, prio_hi = 0;
    if (side_prio) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const bool ok = (side_prio ? hipStreamCreateWithPriority(&x.s, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking)) == hipSuccess &&
                    hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&x.early, hipEventDisableTiming) == hipSuccess;
    e.state = ok ? 1 : -1;
    return ok ? &x : nullptr;
}

static int encoder_fwd_folded(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                              const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx,
                              float* c_n, float* c_u, int B, int N, int H, float* const Xu[2], float* const Xn[2],
                              void* xws, void* xws_news, void* cws, float* kq_t, float* kq_u, float* const r_user2[2],
                              float* r_news, int* live_ws, hipStream_t st, const int* row_group, int G, const float* ue_groups,
                              const float* Xg0, const float* news_hpq0, const float* hist_hpq0, const float* topic_hpq0, void* plan_ws,
                              void* chunk_ws, unsigned char* xsplit_ws, const uint8_t* Au_g, const uint8_t* cm_g, const int64_t* ci_g, const float* ctxq0,
                              const int64_t* news_index, int64_t news_rows, const float* c_n_src, void* live_g_ws = nullptr,
                              const uint8_t* run_leader = nullptr, const uint8_t* run_lead = nullptr, void* news_live_ws = nullptr) {
    // SHARED-USER RUNS (digat_encoder_fwd_shared; round 5): the user tensors are given per ROW, as the reference's driver hands them
    // over (util.py:57-67), and consecutive rows with identical users were found on the device: row_group[b] = the ROW that leads
    // row b's run, run_leader[b] = 1 for those rows, run_lead[b] = rows of the layer-0 chunk row b leads.  Layer 0's group-level
    // data (user nodes, [h|P|Q], live flags) then lives in the LEADING ROW's slots of the full-size buffers, every count stays on
    // the device (no host read of the number of runs), and the group-indexed kernels work as they are.  G is not used.
    const bool shared = run_leader != nullptr && run_lead != nullptr && row_group != nullptr;
    // c_n_src: where the news context stands BEFORE layer 0 — c_n itself, or (depth >= 1, context given) the caller's c_n0, read
    // in place by the two consumers that precede the first update instead of being copied into c_n first
    const bool xu0_grouped = Xg0 != nullptr;       // layer-0 user nodes exist once per group, at Xg0 [G,U,d]
    const int d = p->d, C = p->category_num, L = p->depth, U = H + C, C1 = C + 1;
    const int fmt = (p->flags & DIGAT_PARAMS_GEMM_F16X3) ? 1 : 0;       // the format every wsplit image of `p` was split in
    unsigned* const rflag = fmt ? (unsigned*)p->range_flag : nullptr;
    // the kernel of the [B,d] linears is named by the caller, not chosen from B: a row's bits must not depend on the batch it sits in
    // (nor on whether its context queries come from the per-news table)
    const int bd_disp = (p->flags & DIGAT_PARAMS_BD_TILED) ? (1 << 30) : 1;
    const size_t s2 = align_up((size_t)B * C1 * d * 4, 256);
    float* T = (float*)cws;                       // [B,C1,d] pooled topics
    float* T2 = (float*)((char*)cws + s2);        // after featureAffine
    float* glob = (float*)((char*)cws + 2 * s2);  // [B,d]; cws holds >= 2*s2 + 2*[B,d] (user-context layout)
    const int* bucket_idx = nullptr;              // live topic buckets (set by find_live_rows during layer 0)
    const int* nbuckets_dev = nullptr;
    const int* hist_last = nullptr;               // [B]: 1 + the last live history slot of every row (published with the lists)
    int rc;
    // the user-side queries + (optionally) the next user-graph K3, all from c_n
    auto from_c_n = [&](int next_layer, hipStream_t sq) -> int {
        GemmArgs g = gemm_plain(next_layer == 0 ? c_n_src : c_n, d, p->user_news_fold_W, p->user_news_fold_b, kq_t, d, B, d, d, 0);
        g.w[1] = p->userAtt_fold_W; g.bias[1] = p->userAtt_fold_b; g.y[1] = kq_u;
        g.nsegs = 2;
        if (next_layer < L) {
            g.w[2] = p->user[next_layer].F3; g.bias[2] = p->user[next_layer].b3; g.y[2] = r_user2[next_layer & 1];
            g.nsegs = 3;
        }
        g.wsplit = (const unsigned short*)p->ctx_wsplit[next_layer]; g.format = fmt; g.range_flag = rflag;   // NULL: fp32 MFMA
        g.m_dispatch = bd_disp;
        return launch_gemm(g, sq);
    };
    // `live` (layers' outputs): the rows of dead nodes were never written — the topic pooling takes them as zero
    // kq_topic / kq_user: the two queries derived from the news context (the workspace ones, or — initial context — the rows of the
    // per-news table the caller gathered: ctxq0)
    auto user_ctx_tail = [&](const float* Xu_cur, const float* addend, hipStream_t sq, const int* xgroup = nullptr,
                             const uint8_t* live = nullptr, const float* kq_topic = nullptr, const float* kq_user = nullptr) -> int {
        if (!kq_topic) { kq_topic = kq_t; kq_user = kq_u; }
        // ONE launch (digat_ctxfused.inc) when the weight version carries the fused image and the shape fits; T, T2 stay unused then
        if (p->featureAffine_fsplit && fmt == 1 && ctxfused_ok(H, C1, d) && LAB_ENV("DIGAT_CTX_FUSED", 1) != 0) {
            const CtxFusedArgs fa{Xu_cur, (long)U * d, xgroup, live, U, live ? hist_last : nullptr, kq_topic, kq_user, cat_idx, cat_mask, addend, c_u,
                                  (const uint4*)p->featureAffine_fsplit, p->featureAffine_b, rflag, B, H, C1, d, sqrtf((float)d), LAB_ENV("DIGAT_CF_DBG", 0)};
            return launch_user_ctx_fused(fa, sq);
        }
        int e = launch_topic(Xu_cur, (long)U * d, kq_topic, cat_idx, T, B, H, C1, d, sq, xgroup, live, U, live ? hist_last : nullptr);
        if (e) return e;
        GemmArgs g = gemm_plain(T, d, p->featureAffine_W, p->featureAffine_b, T2, d, B * C1, d, d, 0);
        g.epi = EPI_RELU_RES; g.e0 = T; g.lde0 = d;
        g.wsplit = (const unsigned short*)p->featureAffine_wsplit;       // non-NULL: split operands on the matrix cores
        g.format = fmt; g.range_flag = rflag;
        if (bucket_idx && gemm_is_bf16x6(g)) { g.rowidx = bucket_idx; g.nrows_dev = nbuckets_dev; }   // unmasked buckets only
        e = launch_gemm(g, sq);
        if (e) return e;
        return launch_pool(T2, (long)C1 * d, kq_user, cat_mask, addend, c_u, B, C1, d, sq);
    };
    auto news_ctx = [&](const float* Xn_cur, hipStream_t sq, bool first) -> int {
        const long ldx = (long)N * d;
        float* kq = kq_t;                          // free here: the previous user context has consumed it
        GemmArgs gq = gemm_plain(Xn_cur, ldx, p->cand_fold_W, p->cand_fold_b, kq, d, B, d, d, 0);
        gq.wsplit = (const unsigned short*)p->cand_fold_wsplit; gq.format = fmt; gq.range_flag = rflag;
        gq.m_dispatch = bd_disp;
        int e = launch_gemm(gq, sq);
        if (e) return e;
        e = launch_pool(Xn_cur, ldx, kq, Mn, nullptr, glob, B, N, d, sq);
        if (e) return e;
        GemmArgs g = gemm_plain(Xn_cur, ldx, p->news_graph_W, p->news_graph_b, c_n, d, B, d, 2 * d, 0);
        g.k0 = d; g.a1 = glob; g.lda1 = d;
        g.epi = EPI_GATE; g.e0 = Xn_cur; g.lde0 = ldx; g.e1 = glob; g.lde1 = d; g.e2 = first ? c_n_src : c_n; g.lde2 = d;
        g.wsplit = (const unsigned short*)p->gate_wsplit; g.format = fmt; g.range_flag = rflag;
        g.m_dispatch = 1;          // two-operand input: the split-image [B,d] kernel at every row count (the tiled one does not take it)
        return launch_gemm(g, sq);
    };

    const bool want_live = L > 0 && !(p->flags & DIGAT_PARAMS_NO_LIVE_ROWS) && LAB_ENV("DIGAT_NO_SKIP", 0) == 0 && live_ws;
    // Eq. 8 of the user graph: the sparse kernel, the dense pair, or both with the device choosing (p->flags; the choice
    // comes out of the adjacency pass of find_live_rows)
    int sparse_mode = p->flags & 3;
    if (sparse_mode == 3 || (sparse_mode == DIGAT_XATTN_AUTO && !(L > 0 && live_ws))) sparse_mode = DIGAT_XATTN_DENSE;
    const int* sparse_flag = nullptr;
    const int pq_x3 = (p->flags & DIGAT_PROJ_PQ_X3) ? 1 : 0;
    const bool want_scan = want_live || sparse_mode == DIGAT_XATTN_AUTO;      // the adjacency pass: live lists and / or the decision
    // sparse Eq. 8 of the user graph from LDS-staged rows (digat_staged.inc): needs the plan of the batch (units of centres)
    const bool use_staged = L > 0 && g_staged_on && plan_ws && sparse_mode != DIGAT_XATTN_DENSE && U > 16 && staged_ok(U, d / 4);
    const PlanBuffers plan = use_staged ? plan_carve(plan_ws, B, U) : PlanBuffers{};
    // live rows of the user graph for the projections of layers >= 1 (DIGAT_NO_SKIP=1: every row)
    const int* rowidx = nullptr;
    const int* nrows_dev = nullptr;
    uint8_t* live_flags = nullptr;
    // launches the list kernels on `sq`; `publish` hands the lists to the code below (the initial user context, issued on
    // the caller's stream at the same time, must not see them: only a join orders the caller's stream after the side stream)
    const int *pend_rowidx = nullptr, *pend_nrows = nullptr, *pend_bidx = nullptr, *pend_nb = nullptr, *pend_hlast = nullptr;

    uint8_t* pend_flags = nullptr;
    // layer 0 of grouped rows on the chunk kernel (R rows of an impression per wave: xattn_sparse_l0_kernel): its list — the live
    // centres of the rows that lead a chunk — is made with the other two
    static const int l0_chunks_on = LAB_ENV("DIGAT_L0_CHUNKS", 1);
    const bool l0_chunked = l0_chunks_on && chunk_ws && row_group && (Xg0 || shared) && want_live && g_l0_live_on && !use_staged &&
                            sparse_mode == DIGAT_XATTN_SPARSE && d / 4 <= 128 && U <= 128;
    int* const l0_gs = (int*)chunk_ws;
    int* const l0_off = l0_gs + align_up((size_t)B + 64, 64);
    uint8_t* const l0_lead_own = (uint8_t*)(l0_off + align_up((size_t)B + 64, 64));
    const uint8_t* const l0_lead = shared ? run_lead : l0_lead_own;          // shared runs: the chunk sizes came with the runs
    int* const l0_idx = (int*)(l0_lead_own + align_up((size_t)B, 256));
    // twins: centres of a graph with equal adjacency rows, served together in layers >= 1 (xattn_sparse_twin_kernel)
    static const int twins_on = LAB_ENV("DIGAT_SPARSE_TWINS", 1);
    const bool twins = twins_on && chunk_ws && want_live && !use_staged && sparse_mode == DIGAT_XATTN_SPARSE && d / 4 <= 128 && U <= 128 && L > 1;
    unsigned* const tw_word = (unsigned*)((char*)l0_idx + align_up((size_t)B * U * 4, 256));
    int* const tw_list = (int*)((char*)tw_word + align_up((size_t)B * U * 4, 256));
    uint8_t* const tw_flags = (uint8_t*)tw_list + align_up((size_t)B * U * 4, 256);
    int* const tw_cnt = (int*)(tw_flags + align_up((size_t)B * U, 256));
    int* const tw_off = tw_cnt + align_up((size_t)B + 64, 64);
    // shared runs: the live nodes of the LEADING rows (the rows layer 0's group projection has to make), offsets [B + 64] + list [B U]
    int* const gl_off = tw_off + align_up((size_t)B + 64, 64);
    int* const gl_idx = gl_off + align_up((size_t)B + 64, 64);
    TwinLists tw_pub{nullptr, nullptr, nullptr};
    // Node rows between two layers also as ready-split fp16 pairs (fp16x3 format, sparse Eq. 8 on the live lists): the Eq. 8 kernel of
    // layer i stores them, the projection GEMM of layer i + 1 reads its A fragments as they are (GemmArgs.a_split; same bits)
    // MEASURED (round 4): the shipped GEMM loses 5-8 % of its time with it (its step is bound by the lockstep of its two workgroups,
    // not by the split), the Eq. 8 kernels lose as much storing the second copy: LAB builds only, off by default.
    static const int presplit_on = LAB_ENV("DIGAT_PRESPLIT", 0);
    const bool presplit = presplit_on && xsplit_ws && fmt == 1 && want_live && !use_staged && sparse_mode == DIGAT_XATTN_SPARSE &&
                          d % 8 == 0 && d / 4 <= 256 && U > 16 && (long)B * U >= 2048 && L > 1;
    bool xs_ready = false;            // the split rows of the CURRENT user nodes exist (written by the previous layer's Eq. 8 kernel)
    auto find_live_rows = [&](hipStream_t sq) -> int {
        int* cnt = live_ws;
        int* off = cnt + align_up((size_t)B, 64);
        int* idx = off + align_up((size_t)B + 1, 64);
        int* cnt2 = idx + align_up((size_t)B * U, 64);
        int* off2 = cnt2 + align_up((size_t)B, 64);
        int* idx2 = off2 + align_up((size_t)B + 1, 64);
        int* entries = idx2 + align_up((size_t)B * C1, 64);
        int* flag = entries + align_up((size_t)B, 64);
        int* hlast = flag + 64;
        uint8_t* flags1 = (uint8_t*)(hlast + align_up((size_t)B, 64));
        uint8_t* flags2 = flags1 + align_up((size_t)B * U, 256);
        ProfScope prof(DIGAT_KERNEL_GLUE, (double)B * ((double)U * U + 2.0 * C1 + H * 8.0) + (double)B * (U + C1) * 6, sq);
        if (l0_chunked && !shared) {
            hipLaunchKernelGGL(sparse_l0_chunks_kernel, dim3(1), dim3(1024), 0, sq, row_group, B, G, SPARSE_L0_ROWS, l0_gs, l0_lead_own);
            DIGAT_CHECK_LAUNCH();
        }
        if (shared) {
            // the adjacency pass for the leading rows only; every other row takes its leader's results (in place)
            const bool want_entries = sparse_mode == DIGAT_XATTN_AUTO;
            hipLaunchKernelGGL(user_live_flags_kernel, dim3((B + 3) / 4), dim3(256), (size_t)4 * ((U * U + 63) & ~63), sq, Au, cat_mask, cat_idx, B, U, H, C1,
                               flags1, cnt, want_entries ? entries : (int*)nullptr, hlast, flags2, cnt2,
                               twins ? tw_word : (unsigned*)nullptr, twins ? tw_flags : (uint8_t*)nullptr, twins ? tw_cnt : (int*)nullptr,
                               run_leader);
            DIGAT_CHECK_LAUNCH();
            const LiveExpand le{flags1, twins ? tw_word : nullptr, tw_flags, flags2, cnt, entries, hlast, cnt2, tw_cnt,
                                flags1, twins ? tw_word : nullptr, tw_flags, flags2,
                                cnt, want_entries ? entries : nullptr, hlast, cnt2, twins ? tw_cnt : nullptr};
            hipLaunchKernelGGL(live_expand_kernel, dim3((B + 3) / 4), dim3(256), 0, sq, le, row_group, B, U, C1);
            DIGAT_CHECK_LAUNCH();
        } else if (live_g_ws && row_group && Au_g && cm_g && ci_g && 4 * (long)G <= B) {
            // the user side is given per group: the adjacency pass once per GROUP, its results handed to the group's rows
            const size_t Gm = (size_t)B / 4 + 1;
            uint8_t* fg = (uint8_t*)live_g_ws;
            uint8_t* tfg = fg + align_up(Gm * U, 256);
            unsigned* twg = (unsigned*)(tfg + align_up(Gm * U, 256));
            uint8_t* bfg = (uint8_t*)twg + align_up(Gm * U * 4, 256);
            int* cg = (int*)(bfg + align_up(Gm * C1, 256));
            const size_t gi = align_up(Gm * 4, 256) / 4;
            int *eg = cg + gi, *hg = eg + gi, *bcg = hg + gi, *tcg = bcg + gi;
            const bool want_entries = sparse_mode == DIGAT_XATTN_AUTO;
            hipLaunchKernelGGL(user_live_flags_kernel, dim3((G + 3) / 4), dim3(256), (size_t)4 * ((U * U + 63) & ~63), sq, Au_g, cm_g, ci_g, G, U, H, C1,
                               fg, cg, want_entries ? eg : (int*)nullptr, hg, bfg, bcg,
                               twins ? twg : (unsigned*)nullptr, twins ? tfg : (uint8_t*)nullptr, twins ? tcg : (int*)nullptr);
            DIGAT_CHECK_LAUNCH();
            const LiveExpand le{fg, twins ? twg : nullptr, tfg, bfg, cg, eg, hg, bcg, tcg,
                                flags1, twins ? tw_word : nullptr, tw_flags, flags2,
                                cnt, want_entries ? entries : nullptr, hlast, cnt2, twins ? tw_cnt : nullptr};
            hipLaunchKernelGGL(live_expand_kernel, dim3((B + 3) / 4), dim3(256), 0, sq, le, row_group, B, U, C1);
            DIGAT_CHECK_LAUNCH();
        } else {
            hipLaunchKernelGGL(user_live_flags_kernel, dim3((B + 3) / 4), dim3(256), (size_t)4 * ((U * U + 63) & ~63), sq, Au, cat_mask, cat_idx, B, U, H, C1,
                               flags1, cnt, sparse_mode == DIGAT_XATTN_AUTO ? entries : (int*)nullptr, hlast, flags2, cnt2,
                               twins ? tw_word : (unsigned*)nullptr, twins ? tw_flags : (uint8_t*)nullptr, twins ? tw_cnt : (int*)nullptr);
            DIGAT_CHECK_LAUNCH();
        }
        if (sparse_mode == DIGAT_XATTN_AUTO) {
            hipLaunchKernelGGL(sparse_decide_kernel, dim3(1), dim3(1024), 0, sq, (const int*)entries, B, U, g_sparse_per_node, flag);
            DIGAT_CHECK_LAUNCH();
            sparse_flag = flag;
        }
        // node list and bucket list: both scans in one launch, both lists in one launch
        // up to four lists in the two launches: live nodes, live buckets, [layer-0 chunk leads | twin leads] as wanted
        ScanPair sp{{cnt, cnt2}, {off, off2}, {nullptr, nullptr}};
        ListPair lp{{flags1, flags2}, {off, off2}, {U, C1, U, U, U, U}, {idx, idx2}, {nullptr, nullptr}};
        int jobs = 2;
        if (l0_chunked) {
            sp.cnt[jobs] = cnt; sp.off[jobs] = l0_off; sp.rowmask[jobs] = l0_lead;
            lp.flags[jobs] = flags1; lp.off[jobs] = l0_off; lp.out[jobs] = l0_idx; lp.rowmask[jobs] = l0_lead;
            ++jobs;
        }
        if (twins) {
            sp.cnt[jobs] = tw_cnt; sp.off[jobs] = tw_off;
            lp.flags[jobs] = tw_flags; lp.off[jobs] = tw_off; lp.out[jobs] = tw_list;
            lp.width[jobs] = U;
            ++jobs;
        }
        if (shared) {            // the live nodes of the leading rows: what layer 0's group projection makes
            sp.cnt[jobs] = cnt; sp.off[jobs] = gl_off; sp.rowmask[jobs] = run_leader;
            lp.flags[jobs] = flags1; lp.off[jobs] = gl_off; lp.out[jobs] = gl_idx; lp.rowmask[jobs] = run_leader;
            lp.width[jobs] = U;
            ++jobs;
        }
        hipLaunchKernelGGL(exclusive_scan2_kernel, dim3(jobs), dim3(1024), 0, sq, sp, B);
        DIGAT_CHECK_LAUNCH();
        hipLaunchKernelGGL(live_list2_kernel, dim3((B + 3) / 4, jobs), dim3(256), 0, sq, lp, B);
        DIGAT_CHECK_LAUNCH();
        pend_rowidx = idx; pend_nrows = off + B; pend_bidx = idx2; pend_nb = off2 + B; pend_flags = flags1; pend_hlast = hlast;
        return DIGAT_OK;
    };
    // the plan follows the live flags on the same stream (its second list is the live centres)
    // one plan per distinct user graph: per group when the caller gave the user side per group, else per row
    auto make_plan = [&](hipStream_t sq) -> int {
        const bool per_group = row_group && Au_g && cm_g && ci_g;
        return launch_plan(per_group ? Au_g : Au, per_group ? cm_g : cat_mask, per_group ? ci_g : cat_idx, per_group ? row_group : nullptr,
                           B, per_group ? G : B, U, H, C1, d / 4, want_live, plan, sq);
    };
    auto publish_live_rows = [&]() {
        rowidx = pend_rowidx; nrows_dev = pend_nrows; bucket_idx = pend_bidx; nbuckets_dev = pend_nb; live_flags = pend_flags;
        hist_last = pend_hlast;
        if (twins) tw_pub = TwinLists{tw_word, tw_list, tw_off + B};
    };
    // side stream: by pass size unless the caller says (flags): never / always
    int side_mode = (p->flags & DIGAT_PARAMS_SIDE_STREAM_OFF) ? 0 : ((p->flags & DIGAT_PARAMS_SIDE_STREAM_ON) ? 1 : 2);
    if (g_side_stream_lab >= 0) side_mode = g_side_stream_lab ? 0 : 1;
    SideStream* side = (side_mode == 0 || (side_mode == 2 && B >= 2048)) ? nullptr : side_stream(st);
    // Small news graphs (the wave-per-centre score kernel adds K3 itself): the node projections of a layer depend only on
    // the news nodes, so they are issued on the side stream a phase early — layer 0's under the initial user context,
    // layer i+1's under the pooling of user context i — instead of waiting for c_u.
    const bool news_early = L > 0 && N <= 16 && d / 4 <= 256;       // same arithmetic with and without the side stream
    // The padding slots of a news graph are dead nodes — neither projected nor (on the sparse kernel) scored or written in any layer
    // (news_live_flags_kernel); the context pooling masks them and skips zero weights, nobody else reads them.  Larger graphs
    // (N > 16, sparse kernel): projection and Eq. 8 run on the live list.  Small graphs: the projections of layers >= 1 do; the
    // one-workgroup-per-graph Eq. 8 kernel still computes every centre (a live centre visits its adjacency entries only: live nodes).
    const bool news_lists_on = L > 0 && news_live_ws && d / 4 <= 256 && !(p->flags & DIGAT_PARAMS_NO_LIVE_ROWS) && LAB_ENV("DIGAT_NEWS_LIVE", 1) != 0;
    const bool news_lists = news_lists_on && ((!news_early && (p->flags & DIGAT_NEWS_XATTN_SPARSE) && N > 16) || (news_early && L > 1 && B >= 2048));
    // (small graphs below 2 048 rows: the three list launches sit on the news chain's critical path and cost what two smaller projections save)
    const int* news_rowidx = nullptr; const int* news_nrows = nullptr; const uint8_t* news_flags = nullptr;
    auto news_project = [&](int layer, const float* Xn_cur, hipStream_t sq) -> int {
        const digat_layer_params& ln = p->news[layer];
        const size_t ndn = (size_t)B * N * d;
        float* hn = (float*)xws_news;
        GemmArgs gp = gemm_plain(Xn_cur, d, ln.W, ln.bW, hn, d, B * N, d, d, 0);
        gp.w[1] = ln.F1; gp.bias[1] = nullptr; gp.y[1] = hn + ndn;
        gp.w[2] = ln.F2; gp.bias[2] = nullptr; gp.y[2] = hn + 2 * ndn;
        gp.nsegs = 3;
        gp.x3_segs = pq_x3 ? 6 : 0;
        gp.m_dispatch = 1 << 30;       // always the large-M kernel: a row's bits then do not depend on the batch it sits in
                                       // (digat_news_project0 makes the same launch per news, once)
        gp.wsplit = (const unsigned short*)ln.wsplit;
        gp.format = fmt; gp.range_flag = rflag;
        if (news_rowidx && gemm_is_bf16x6(gp)) { gp.rowidx = news_rowidx; gp.nrows_dev = news_nrows; }      // live nodes only (layers >= 1)
        return launch_gemm(gp, sq, DIGAT_KERNEL_PROJ);
    };
    // layer 0 of grouped rows: every row of a group has the same user nodes, so the G groups are projected once
    // ([G*U] rows instead of [B*U]).  h and Q of a group go straight to the h / Q slots of the Eq. 8 workspace and are
    // read through the group index by the aggregation / score kernels (the 37 rows of an impression share them: they
    // stay in L2); only P' = K3_b + P depends on the row and is expanded later.  Needs nothing but the inputs.
    auto group_project = [&](hipStream_t sq) -> int {
        const digat_layer_params& lu = p->user[0];
        if (shared) {
            // shared runs: [h|P|Q] of the LEADING rows' live nodes, in place in the full-size planes (the per-row launch of layer 0
            // restricted to the rows whose results anybody reads; K3 joins in the Eq. 8 kernel, as for the groups below)
            const size_t ndf = (size_t)B * U * d;
            float* hs = (float*)xws;
            GemmArgs gs = gemm_plain(Xu[0], d, lu.W, lu.bW, hs, d, B * U, d, d, 0);
            gs.w[1] = lu.F1; gs.bias[1] = nullptr; gs.y[1] = hs + ndf;
            gs.w[2] = lu.F2; gs.bias[2] = nullptr; gs.y[2] = hs + 2 * ndf;
            gs.nsegs = 3;
            gs.x3_segs = pq_x3 ? 6 : 0;
            gs.wsplit = (const unsigned short*)lu.wsplit;
            gs.format = fmt; gs.range_flag = rflag;
            gs.m_dispatch = B * U;
            if (want_live && gemm_is_bf16x6(gs)) { gs.rowidx = gl_idx; gs.nrows_dev = gl_off + B; }
            return launch_gemm(gs, sq, DIGAT_KERNEL_PROJ);
        }
        const size_t ndg = (size_t)G * U * d;
        const size_t nd = (size_t)B * U * d;
        float* Xg = xu0_grouped ? const_cast<float*>(Xg0) : Xu[1];     // group nodes: built by the caller, or here (Xu[1] is free until layer 0 writes it)
        float* h0 = (float*)xws;
        float* P0 = h0 + ndg;                               // behind the groups' h in the h slot (2 ndg <= nd)
        float* Q0 = h0 + 2 * nd;
        const long total4 = (long)ndg / 4;
        int blocks = (int)((total4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (!xu0_grouped) {
            hipLaunchKernelGGL(build_user_nodes_kernel, dim3(blocks), dim3(256), 0, sq, (const float4*)ue_groups,
                               (const float4*)p->topic_node_embedding, (float4*)Xg, (long)G, H, C, d / 4, (const int*)nullptr);
            DIGAT_CHECK_LAUNCH();
        }
        if (hist_hpq0 && topic_hpq0 && (long)B * U >= 2048) {
            // [h|P|Q] of a history node depend on that news alone and those of a topic node on nothing: the caller keeps them
            // per news / per topic (digat_user_project0) and hands over the groups' history rows; the projection GEMM of the
            // groups becomes three assemblies [history rows | topic rows] (same kernel, same bits: rows are independent)
            float* dst[3] = {h0, P0, Q0};
            UserNodes3 un3;
            for (int t = 0; t < 3; ++t) {
                un3.hist[t] = (const float4*)(hist_hpq0 + (size_t)t * G * H * d);
                un3.topic[t] = (const float4*)(topic_hpq0 + (size_t)t * C * d);
                un3.dst[t] = (float4*)dst[t];
            }
            hipLaunchKernelGGL(build_user_nodes3_kernel, dim3(blocks, 3), dim3(256), 0, sq, un3, (long)G, H, C, d / 4);
            DIGAT_CHECK_LAUNCH();
            return DIGAT_OK;
        }
        GemmArgs gg = gemm_plain(Xg, d, lu.W, lu.bW, h0, d, G * U, d, d, 0);
        gg.w[1] = lu.F1; gg.bias[1] = nullptr; gg.y[1] = P0;
        gg.w[2] = lu.F2; gg.bias[2] = nullptr; gg.y[2] = Q0;
        gg.nsegs = 3;
        gg.x3_segs = pq_x3 ? 6 : 0;
        gg.wsplit = (const unsigned short*)lu.wsplit;
        gg.format = fmt; gg.range_flag = rflag;
        gg.m_dispatch = B * U;                              // the kernel the per-row path would pick: same bits
        return launch_gemm(gg, sq, DIGAT_KERNEL_PROJ);
    };
    // Work that depends on the inputs alone goes out on the side stream at once, under the initial user context:
    // the group projections of layer 0 and (small news graphs) the news projections of layer 0.
    const bool group_early = side && L > 0 && row_group;
    const bool live_early = side && want_scan;
    if (side && (news_early || group_early || live_early)) {
        if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess)
            return DIGAT_ERR_LAUNCH;
    }
    const bool plan_early = side && use_staged;
    if (side && plan_early && !(news_early || group_early || live_early)) {
        if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess)
            return DIGAT_ERR_LAUNCH;
    }
    if (live_early) {                  // first: layer 0 may need the sparse / dense decision
        rc = find_live_rows(side->s);
        if (rc) return rc;
    }
    if (plan_early) {
        rc = make_plan(side->s);
        if (rc) return rc;
    }
    if (group_early) {
        rc = group_project(side->s);
        if (rc) return rc;
    }
    if (group_early || plan_early || live_early) {
        if (hipEventRecord(side->early, side->s) != hipSuccess) return DIGAT_ERR_LAUNCH;
    }
    if (news_early && !news_hpq0) {      // news_hpq0: the caller kept layer 0's news projections per news (digat_news_project0)
        rc = news_project(0, Xn_in, side ? side->s : st);
        if (rc) return rc;
    }
    // [kq_topic | kq_user | K3 of the user graph's layer 0] are functions of the candidate's cached c_n0: a caller that keeps them
    // per news (digat_news_context_queries, next to c_n0 itself) hands over the batch's rows and the first link of the chain goes
    const size_t bd = (size_t)B * d;
    if (!ctxq0) {
        rc = from_c_n(0, st);
        if (rc) return rc;
    }
    const bool xu0_shared = shared && sparse_mode == DIGAT_XATTN_SPARSE;      // only the leading rows of Xu[0] were built: read through row_group
    rc = user_ctx_tail(xu0_grouped ? Xg0 : Xu[0], nullptr, st, (xu0_grouped || xu0_shared) ? row_group : nullptr, nullptr,
                       ctxq0 ? ctxq0 : nullptr, ctxq0 ? ctxq0 + bd : nullptr);        // c_u (:192)
    if (rc) return rc;
    const float* xn_cur = Xn_in;
    int un = 0, nn = 0;
    for (int i = 0; i < L; ++i) {
        const digat_layer_params& ln = p->news[i];
        const digat_layer_params& lu = p->user[i];
        const float* r_user = (i == 0 && ctxq0 && L > 0) ? ctxq0 + 2 * bd : r_user2[i & 1];     // K3 of the user graph, from the previous c_n
        hipStream_t sn = side ? side->s : st;
        if (side && i == 0) {                      // the news chain starts from the initial c_u (caller's stream)
            if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(sn, side->fork, 0) != hipSuccess)
                return DIGAT_ERR_LAUNCH;
        }
        if (side && i > 0) {                       // K3 of this layer's user graph + the live lists come from the side stream
            if (hipStreamWaitEvent(st, side->join, 0) != hipSuccess) return DIGAT_ERR_LAUNCH;
        }
        // ---- user graph, Eq. 8 (caller's stream)
        if (i == 0 && want_scan && !live_early) {              // no side stream: the lists and the sparse / dense decision first
            rc = find_live_rows(st);
            if (rc) return rc;
        }
        if (i == 0 && use_staged && !plan_early) {
            rc = make_plan(st);
            if (rc) return rc;
        }
        if (i == 0 && side && (group_early || plan_early || live_early)) {
            if (hipStreamWaitEvent(st, side->early, 0) != hipSuccess) return DIGAT_ERR_LAUNCH;
        }
        // The live lists are in force from layer 0 on: a dead node (a history padding slot, the topic node of an unread category:
        // only its self loop, pooled with weight 0) is never projected, scored or written, in ANY layer.  Its rows of the two
        // node buffers therefore hold whatever the workspace held; the one reader that walks all history rows — the topic
        // pooling — takes them as zero through the flags (a select: no bits of such a row can reach a result;
        // test_uninitialised_workspace_cannot_reach_the_outputs fills the scratch with NaN patterns).
        if (i == 0 && want_live) publish_live_rows();
        if (i == 0 && row_group) {
            const size_t nd = (size_t)B * U * d;
            const size_t ndg = shared ? nd : (size_t)G * U * d;      // shared runs: full-size planes, a group's rows sit in its leading row's slots
            float* h0 = (float*)xws;
            float* P0 = h0 + ndg;
            float* P = h0 + nd;
            float* Q0 = P + nd;
            if (!group_early) {
                rc = group_project(st);
                if (rc) return rc;
            }
            rc = DIGAT_OK;
            if (sparse_mode != DIGAT_XATTN_DENSE && d / 4 <= 256) {
                // P' = K1 (the groups' P0) + K3 (this layer's r_user) is formed inside the kernel: nothing is expanded
                // live centres only (the list of find_live_rows), P / Q / h / X read through the group index.  The staged kernels
                // (opt-in) keep the older arrangement: every centre computed, and the dead rows of the other buffer filled with X_i
                const bool l0_live = want_live && live_flags && !use_staged && g_l0_live_on;
                SparseArgs sg{P0, Q0, h0, xu0_grouped ? Xg0 : Xu[0], lu.a, Au, Xu[1], r_user, row_group, l0_live ? live_flags : nullptr,
                                    sparse_mode == DIGAT_XATTN_AUTO ? sparse_flag : nullptr, B, U, d / 4, (xu0_grouped || xu0_shared) ? 1 : 0,
                                    (!l0_live && xu0_grouped && want_live) ? (const uint8_t*)pend_flags : nullptr, Xu[0],
                                    l0_live ? rowidx : nullptr, l0_live ? nrows_dev : nullptr, G, nullptr, 0, 0};
                if (presplit && l0_live && L > 1) { sg.xsplit = xsplit_ws; sg.xsplit_range = rflag; xs_ready = true; }
                if (l0_chunked && l0_live && sparse_l0_ok(sg)) {
                    // R rows of an impression per wave: every neighbour row fetched serves R rows (xattn_sparse_l0_kernel; same bits)
                    rc = launch_sparse_l0(sg, l0_lead, l0_idx, l0_off + B, shared ? (B + SPARSE_L0_ROWS - 1) / SPARSE_L0_ROWS : G, st);
                } else
                rc = use_staged ? launch_staged(sg, plan, 0, 0, st) : launch_sparse(sg, st);
            }
            if (!rc && !(sparse_mode == DIGAT_XATTN_SPARSE && d / 4 <= 256)) {
                const int* skip_if = sparse_mode == DIGAT_XATTN_AUTO && d / 4 <= 256 ? sparse_flag : nullptr;
                float* alpha = (float*)((char*)xws + align_up(3 * nd * 4, 256) + align_up((size_t)B * d * 4, 256));
                {
                    const long total4 = (long)nd / 4;
                    int blocks = (int)((total4 + 255) / 256);
                    if (blocks > 4096) blocks = 4096;
                    ProfScope prof(DIGAT_KERNEL_GLUE, skip_if ? 0.0 : (double)nd * 4, st);
                    hipLaunchKernelGGL(expand_proj_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)P0, (const float4*)r_user,
                                       row_group, (float4*)P, (long)B, U, d / 4, skip_if);
                    DIGAT_CHECK_LAUNCH();
                }
                rc = launch_xattn_pairwise(P, Q0, h0, Xu[0], lu.a, Au, Xu[1], alpha, B, U, d, st, nullptr, row_group, nullptr, true,
                                           skip_if);
            }
        } else {
            // every layer projects, scores and writes the live nodes only (layer 0 too: see publish_live_rows above)
            const bool lv_on = i > 0 || (g_l0_live_on && !use_staged);
            const bool xs_out = presplit && lv_on && rowidx && live_flags && i + 1 < L;
            const SplitIO sio{xs_ready && lv_on && rowidx ? (const void*)xsplit_ws : nullptr, xs_out ? xsplit_ws : nullptr, rflag};
            rc = xattn_core(Xu[un], Au, r_user, lu.W, lu.bW, lu.F1, lu.F2, lu.a, Xu[un ^ 1], nullptr, B, U, d, xws, st, lu.wsplit,
                            lv_on ? rowidx : nullptr, lv_on ? nrows_dev : nullptr, lv_on ? live_flags : nullptr, sparse_mode,
                            sparse_flag, pq_x3, use_staged ? &plan : nullptr, i,
                            i > 0 ? ((p->flags & DIGAT_PQ_BF16) ? 1 : 0) | ((p->flags & DIGAT_PQ_X1) ? 2 : 0) | ((p->flags & DIGAT_PQ_FP8) ? 4 : 0) : 0,
                            // after the last layer only the history rows are read (the user context's topic pooling, :124):
                            // the topic nodes' own Eq. 8 is not computed there (wave-per-centre sparse kernel)
                            (i > 0 && i == L - 1 && sparse_mode == DIGAT_XATTN_SPARSE) ? H : 0, fmt, rflag,
                            (i > 0 && lv_on && tw_pub.word) ? &tw_pub : nullptr, &sio);
            xs_ready = sio.out != nullptr;
        }
        if (rc) return rc;
        if (side && hipEventRecord(side->fork, st) != hipSuccess) return DIGAT_ERR_LAUNCH;      // this layer's user nodes are written
        // ---- news graph, Eq. 8 + context + the queries that follow from the new c_n (side stream)
        {
            GemmArgs g3 = gemm_plain(c_u, d, ln.F3, ln.b3, r_news, d, B, d, d, 0);              // K3 of the news graph
            g3.wsplit = (const unsigned short*)ln.f3_wsplit; g3.format = fmt; g3.range_flag = rflag;
            g3.m_dispatch = bd_disp;
            rc = launch_gemm(g3, sn);
        }
        if (rc) return rc;
        if (i == 0 && news_lists) {            // the live nodes of the news graphs, once per pass (the graphs do not change with the layers)
            int* cnt_n = (int*)news_live_ws;
            int* off_n = cnt_n + align_up((size_t)B, 64);
            int* idx_n = off_n + align_up((size_t)B + 1, 64);
            uint8_t* flags_n = (uint8_t*)(idx_n + align_up((size_t)B * N, 64));
            ProfScope prof(DIGAT_KERNEL_GLUE, (double)B * ((double)N * N + 6.0 * N), sn);
            hipLaunchKernelGGL(news_live_flags_kernel, dim3((B + 3) / 4), dim3(256), (size_t)4 * ((N * N + 15) & ~15), sn, An, Mn, B, N, flags_n, cnt_n);
            DIGAT_CHECK_LAUNCH();
            ScanPair sp{{cnt_n}, {off_n}, {nullptr}};
            ListPair lp{{flags_n}, {off_n}, {N, N, N, N, N, N}, {idx_n}, {nullptr}};
            hipLaunchKernelGGL(exclusive_scan2_kernel, dim3(1), dim3(1024), 0, sn, sp, B);
            DIGAT_CHECK_LAUNCH();
            hipLaunchKernelGGL(live_list2_kernel, dim3((B + 3) / 4, 1), dim3(256), 0, sn, lp, B);
            DIGAT_CHECK_LAUNCH();
            news_rowidx = idx_n; news_nrows = off_n + B; news_flags = flags_n;
        }
        if (news_early) {        // projections already done (news_project below): K3 joins in the score kernel
            const size_t ndn = (size_t)B * N * d;
            const float* hn = (i == 0 && news_hpq0) ? news_hpq0 : (const float*)xws_news;
            // layer 0 with news_index: h | P | Q and the nodes themselves are the per-news TABLES, read in place through the index
            const bool tab = i == 0 && news_hpq0 && news_index;
            const size_t plane = tab ? (size_t)news_rows * N * d : ndn;
            float* alpha_n = (float*)((char*)xws_news + align_up(3 * ndn * 4, 256) + align_up((size_t)B * d * 4, 256));
            rc = launch_xattn_pairwise(hn + plane, hn + 2 * plane, hn, xn_cur, ln.a, An, Xn[nn], alpha_n, B, N, d, sn, nullptr, nullptr, r_news,
                                       false, nullptr, tab ? news_index : nullptr);
        } else if (i == 0 && news_hpq0 && news_index) {
            // Larger news graphs, layer 0 from the per-news TABLES (round 4): [h | P | Q] of a news graph depend on the news alone, so
            // the projection GEMM of layer 0 (B N rows: at N = 26 the largest launch of a MIND-large step) is replaced by reading the
            // candidates' rows of the table IN PLACE — the sparse kernel's group indirection with the candidate id as the "group" —
            // and adding K3 in the kernel, in the GEMM epilogue's order (K3 + K1): the bits of the in-batch launch.
            const size_t plane = (size_t)news_rows * N * d;
            int* idx32 = (int*)((char*)chunk_ws + l0_chunk_bytes(B, U) - align_up((size_t)B * 4, 256));
            hipLaunchKernelGGL(index_to_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, sn, news_index, idx32, B);
            DIGAT_CHECK_LAUNCH();
            SparseArgs sgn{news_hpq0 + plane, news_hpq0 + 2 * plane, news_hpq0, xn_cur, ln.a, An, Xn[nn], r_news, idx32, news_flags,
                           nullptr, B, N, d / 4, 1, nullptr, nullptr, news_rowidx, news_nrows, B, nullptr, 0, 0};      // G (profiling: distinct rows behind the index): at most B candidates
            if (news_rowidx) sgn.prof_part = XPART_NEWS + 1;
            rc = launch_sparse(sgn, sn);
        } else {
            // larger news graphs (N = 26 / 65: the breadth-first SAG, a few entries per node) take the sparse kernel when the
            // caller says so (flags bit 3); there is no device-side decision for this graph
            rc = xattn_core(xn_cur, An, r_news, ln.W, ln.bW, ln.F1, ln.F2, ln.a, Xn[nn], nullptr, B, N, d, xws_news, sn, ln.wsplit,
                            news_rowidx, news_nrows, news_flags, (p->flags & DIGAT_NEWS_XATTN_SPARSE) ? DIGAT_XATTN_SPARSE : DIGAT_XATTN_DENSE,
                            nullptr, pq_x3, nullptr, 0,
                            // the news graph's P' always carries K3 from the GEMM epilogue: bf16 storage applies at every layer
                            ((p->flags & DIGAT_PQ_BF16) ? 1 : 0) | ((p->flags & DIGAT_PQ_X1) ? 2 : 0) | ((p->flags & DIGAT_PQ_FP8) ? 4 : 0), 0, fmt, rflag,
                            nullptr, nullptr, news_rowidx ? XPART_NEWS + 1 : 0);
        }
        if (rc) return rc;
        xn_cur = Xn[nn]; nn ^= 1; un ^= 1;
        rc = news_ctx(xn_cur, sn, i == 0);         // c_n += ... (:196)
        if (rc) return rc;
        rc = from_c_n(i + 1, sn);                  // queries (+ next K3, into the other r_user buffer) from the UPDATED c_n
        if (rc) return rc;
        if (side && hipEventRecord(side->join, sn) != hipSuccess) return DIGAT_ERR_LAUNCH;      // the next user-graph update may start
        if (news_early && i + 1 < L) {             // the next layer's news projections need only Xn
            rc = news_project(i + 1, xn_cur, sn);
            if (rc) return rc;
        }
        // The user context of this layer feeds the next NEWS update and the result, not the next user-graph update: it runs
        // on the side stream once the caller's stream has written the user nodes, under the next layer's projection GEMM.
        if (side && hipStreamWaitEvent(sn, side->fork, 0) != hipSuccess) return DIGAT_ERR_LAUNCH;
        rc = user_ctx_tail(Xu[un], c_u, sn, nullptr, live_flags);       // c_u += ... (:197)
        if (rc) return rc;
    }
    if (side && L > 0) {
        if (hipEventRecord(side->join, side->s) != hipSuccess || hipStreamWaitEvent(st, side->join, 0) != hipSuccess)
            return DIGAT_ERR_LAUNCH;
    }
    return DIGAT_OK;
}

#ifdef DIGAT_LAB
static size_t xsplit_bytes(int B, int U, int d) { return align_up((size_t)B * U * d * 4, 256); }
#else
static size_t xsplit_bytes(int, int, int) { return 0; }
#endif
// layer 0 of grouped rows (xattn_sparse_l0_kernel): group starts [B + 64] int, rows led by each row [B] bytes, offsets [B + 64] int
// and list [B U] int of the live centres of the chunk-leading rows
// + twins (xattn_sparse_twin_kernel): twin words [B U] u32, lead flags [B U] bytes, leads per row [B + 64] int, offsets [B + 64] int, list [B U] int
// + [B] int: the candidate ids as 32-bit indices (layer 0 of larger news graphs from the per-news tables)
// + shared-user runs (digat_encoder_fwd_shared): offsets [B + 64] int and list [B U] int of the live nodes of the run-leading rows
static size_t l0_chunk_bytes(int B, int U) {
    return 2 * align_up((size_t)(B + 64) * 4, 256) + align_up((size_t)B, 256) + align_up((size_t)B * U * 4, 256)
           + 2 * align_up((size_t)B * U * 4, 256) + align_up((size_t)B * U, 256) + 2 * align_up((size_t)(B + 64) * 4, 256)
           + align_up((size_t)(B + 64) * 4, 256) + align_up((size_t)B * U * 4, 256)
           + align_up((size_t)B * 4, 256);
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
    // + adjacency entries per row and the sparse / dense decision (int)
    tot += align_up((4 * align_up((size_t)B, 64) + 2 * align_up((size_t)B + 1, 64) + align_up((size_t)B * U, 64)
                     + align_up((size_t)B * (C + 1), 64) + 64) * 4 + align_up((size_t)B * U, 256) + align_up((size_t)B * (C + 1), 256), 256);
    tot += news_live_bytes(B, N);                         // larger news graphs on the sparse kernel: live-node flags, counts, offsets, list
    tot += xsplit_bytes(B, U, d);                         // LAB builds: the user nodes as split fp16 pairs between two layers (SplitIO)
    tot += l0_chunk_bytes(B, U);                          // layer 0 of grouped rows: group starts + rows led by each row (xattn_sparse_l0_kernel)
    tot += plan_bytes(B, U);                             // the staged Eq. 8 kernel's plan of the batch (digat_staged.inc)
    return tot;
}

// row_group == NULL: user tensors are per row.  Otherwise they are per group (G of them) and row_group[b]
// names the group of row b; the caller (digat_encoder_fwd_grouped) has expanded the small per-row byte /
// index arrays, so Au / cat_mask / cat_idx are per row in both cases and only ue [G,H,d] is per group.
static int encoder_fwd_impl(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                            const float* ue, const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx,
                            const float* c_n0, float* out_news, float* out_user, int B, int N, int H,
                            void* workspace, size_t workspace_bytes, void* stream, const int* row_group, int G,
                            const float* news_hpq0 = nullptr, const float* hist_hpq0 = nullptr, const float* topic_hpq0 = nullptr,
                            const uint8_t* Au_g = nullptr, const uint8_t* cm_g = nullptr, const int64_t* ci_g = nullptr,
                            const float* ctxq0 = nullptr, const int64_t* news_index = nullptr, int64_t news_rows = 0,
                            void* live_g_ws = nullptr, const uint8_t* run_leader = nullptr, const uint8_t* run_lead = nullptr) {
    if (!p || !Xn_in || !An || !Mn || !ue || !Au || !cat_mask || !cat_idx || !out_news || !out_user || !workspace)
        return DIGAT_ERR_ARG;
    if (B < 0 || N <= 0 || H < 0) return DIGAT_ERR_ARG;
    const int d = p->d, C = p->category_num, L = p->depth, U = H + C;
    if (d <= 0 || d % 4 || L < 0 || L > DIGAT_MAX_DEPTH || N > DIGAT_MAX_NODES || U > DIGAT_MAX_NODES || C < 0)
        return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_encoder_workspace_bytes(B, N, H, C, d, L)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    // per-news tables read in place: only where layer 0 of the news graph is the one reader of the node table (given c_n0,
    // cached projections, the small-graph kernel)
    // ... or, for larger news graphs (round 4), the sparse Eq. 8 kernel through the candidate ids (flags: DIGAT_NEWS_XATTN_SPARSE)
    if (news_index && !(c_n0 && news_hpq0 && L > 0 && d / 4 <= 256 && news_rows > 0 && news_rows <= 0x7fffffffLL &&
                        (N <= 16 || ((p->flags & DIGAT_NEWS_XATTN_SPARSE) && N <= DIGAT_MAX_NODES)) &&
                        p->cand_fold_W && p->user_news_fold_W && p->userAtt_fold_W)) return DIGAT_ERR_ARG;
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
    void* plan_ws = (char*)workspace + digat_encoder_workspace_bytes(B, N, H, C, d, L) - plan_bytes(B, U);
    void* chunk_ws = (char*)plan_ws - l0_chunk_bytes(B, U);
    unsigned char* xsplit_ws = xsplit_bytes(B, U, d) ? (unsigned char*)chunk_ws - xsplit_bytes(B, U, d) : nullptr;
    void* news_live_ws = (char*)chunk_ws - xsplit_bytes(B, U, d) - news_live_bytes(B, N);

    int rc;
    const bool folded = p->cand_fold_W && p->user_news_fold_W && p->userAtt_fold_W;
    // Rows of one impression share the user nodes.  When every reader of the layer-0 nodes can go through the group index
    // (the sparse Eq. 8 kernel and the topic pooling can; the dense tile / aggregation kernels cannot) they are built once
    // per GROUP: 3 MB instead of a 110 MB expansion that the first two kernels would read back.
    // Shared-user runs (digat_encoder_fwd_shared): ue is per ROW, row_group[b] = the row that leads row b's run.  Taken when the
    // group-indexed kernels of layer 0 apply (sparse Eq. 8 on the live lists); otherwise the runs are ignored: the plain per-row path.
    const bool shared = run_leader && run_lead && row_group && folded && L > 0 && (p->flags & 3) == DIGAT_XATTN_SPARSE &&
                        !(p->flags & DIGAT_PARAMS_NO_LIVE_ROWS) && d / 4 <= 128 && U <= 128 && U > 16;
    if ((run_leader || run_lead) && !shared) { row_group = nullptr; G = 0; run_leader = run_lead = nullptr; }
    const bool xu0_grouped = !shared && folded && row_group && L > 0 && (p->flags & 3) == DIGAT_XATTN_SPARSE && d / 4 <= 256 && U > 16 && 3 * (long)G <= B;
    // user graph nodes = [history | topic nodes]  (:191)
    float* const Xg0 = xu0_grouped ? (float*)xws + 2 * (size_t)G * U * d : nullptr;      // behind the groups' h and P in the h slot (3 G <= B)
    {
        const long nrows = xu0_grouped ? G : B;
        const long total4 = nrows * U * (d / 4);
        int blocks = (int)((total4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        ProfScope prof(DIGAT_KERNEL_GLUE, (double)nrows * ((double)H * d * 8 + (double)C * d * 4), st);
        // shared runs: ue is per row already, and only the run-leading rows are ever read (through row_group)
        hipLaunchKernelGGL(build_user_nodes_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)ue,
                           (const float4*)p->topic_node_embedding, (float4*)(xu0_grouped ? Xg0 : Xu[0]), nrows, H, C, d / 4,
                           (xu0_grouped || shared) ? (const int*)nullptr : row_group, shared ? run_leader : (const uint8_t*)nullptr);
        DIGAT_CHECK_LAUNCH();
    }
    // c_n: given (inference, :189) or computed (forward, :180); it lives in out_news from here on
    const bool c_n0_in_place = c_n0 && folded && L > 0;        // layer 0's news context update writes out_news from c_n0 directly
    if (c_n0_in_place) {
    } else if (c_n0) {
        if (hipMemcpyAsync(out_news, c_n0, (size_t)B * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return DIGAT_ERR_LAUNCH;
    } else {
        rc = digat_news_ctx_fwd(Xn_in, Mn, p->cand_K, p->cand_Q, p->cand_bQ, p->news_graph_W, p->news_graph_b,
                                nullptr, out_news, B, N, d, cws, cws_bytes, stream);
        if (rc) return rc;
    }
    if (folded)
        return encoder_fwd_folded(p, Xn_in, An, Mn, Au, cat_mask, cat_idx, out_news, out_user, B, N, H, Xu, Xn, xws,
                                  xws_news, cws, kq_t, kq_u, r_user2, r_news, live_ws, st, row_group, G, ue, Xg0, news_hpq0, hist_hpq0, topic_hpq0, plan_ws, chunk_ws, xsplit_ws, Au_g, cm_g, ci_g,
                                  c_n0 ? ctxq0 : nullptr, news_index, news_rows, c_n0_in_place ? c_n0 : out_news, live_g_ws,
                                  run_leader, run_lead, news_live_ws);
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
        const int fmt = (p->flags & DIGAT_PARAMS_GEMM_F16X3) ? 1 : 0;
        unsigned* const rflag = fmt ? (unsigned*)p->range_flag : nullptr;
        rc = xattn_core(xn_cur, An, r_news, ln.W, ln.bW, ln.F1, ln.F2, ln.a, Xn[nn], nullptr, B, N, d, xws, st, ln.wsplit,
                        nullptr, nullptr, nullptr, DIGAT_XATTN_DENSE, nullptr, 0, nullptr, 0, 0, 0, fmt, rflag);
        if (rc) return rc;
        rc = launch_gemm(gemm_plain(out_news, d, lu.F3, lu.b3, r_user, d, B, d, d, 0), st);
        if (rc) return rc;
        rc = xattn_core(Xu[un], Au, r_user, lu.W, lu.bW, lu.F1, lu.F2, lu.a, Xu[un ^ 1], nullptr, B, U, d, xws, st, lu.wsplit,
                        nullptr, nullptr, nullptr, DIGAT_XATTN_DENSE, nullptr, 0, nullptr, 0, 0, 0, fmt, rflag);
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

#ifdef DIGAT_LAB
int digat_set_staged_xattn(int mode) {        // LAB builds only: process-wide, one host thread
    (void)staged_cfg();
    const int prev = g_staged_on ? 1 + g_staged_cfg : 0;
    g_staged_on = mode > 0 ? 1 : 0;
    if (mode > 0 && mode <= 5) g_staged_cfg = mode - 1;
    return prev;
}
#endif

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
           + align_up((size_t)B * (C + 1), 256) + align_up((size_t)B * H * 8, 256) + live_group_bytes(B, U, C + 1);
}

static int encoder_fwd_grouped_impl(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                                    const float* ue_g, const uint8_t* Au_g, const uint8_t* cat_mask_g, const int64_t* cat_idx_g,
                                    const int32_t* row_group, const float* c_n0, float* out_news, float* out_user,
                                    int B, int G, int N, int H, void* workspace, size_t workspace_bytes, void* stream,
                                    const float* news_hpq0, const float* hist_hpq0 = nullptr, const float* topic_hpq0 = nullptr,
                                    const float* ctxq0 = nullptr, const int64_t* news_index = nullptr, int64_t news_rows = 0) {
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
    {
        const GatherJobs jobs{{Au_g, cat_mask_g, (const uint8_t*)cat_idx_g}, {Au, cm, ci}, {(long)U * U, (long)(C + 1), (long)H * 8}};
        hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, st, jobs, row_group, (long)B);
        DIGAT_CHECK_LAUNCH();
    }
    void* live_g_ws = ci + align_up((size_t)B * H * 8, 256);
    return encoder_fwd_impl(p, Xn_in, An, Mn, ue_g, Au, cm, (const int64_t*)ci, c_n0, out_news, out_user, B, N, H, workspace,
                            base, stream, row_group, G, news_hpq0, hist_hpq0, topic_hpq0, Au_g, cat_mask_g, cat_idx_g, ctxq0, news_index, news_rows,
                            live_g_ws);
}

int digat_encoder_fwd_grouped(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                              const float* ue_g, const uint8_t* Au_g, const uint8_t* cat_mask_g, const int64_t* cat_idx_g,
                              const int32_t* row_group, const float* c_n0, float* out_news, float* out_user,
                              int B, int G, int N, int H, void* workspace, size_t workspace_bytes, void* stream) {
    return encoder_fwd_grouped_impl(p, Xn_in, An, Mn, ue_g, Au_g, cat_mask_g, cat_idx_g, row_group, c_n0, out_news, out_user, B, G, N, H,
                                    workspace, workspace_bytes, stream, nullptr);
}

int digat_encoder_fwd_grouped_cached(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn,
                                     const float* ue_g, const uint8_t* Au_g, const uint8_t* cat_mask_g, const int64_t* cat_idx_g,
                                     const int32_t* row_group, const float* c_n0, const float* news_hpq0,
                                     const float* hist_hpq0, const float* topic_hpq0, const float* ctxq0,
                                     const int64_t* news_index, int64_t news_rows, float* out_news,
                                     float* out_user, int B, int G, int N, int H, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if ((hist_hpq0 == nullptr) != (topic_hpq0 == nullptr)) return DIGAT_ERR_ARG;
    if (ctxq0 && !c_n0) return DIGAT_ERR_ARG;           // the queries belong to a given news context
    return encoder_fwd_grouped_impl(p, Xn_in, An, Mn, ue_g, Au_g, cat_mask_g, cat_idx_g, row_group, c_n0, out_news, out_user, B, G, N, H,
                                    workspace, workspace_bytes, stream, news_hpq0, hist_hpq0, topic_hpq0, ctxq0, news_index, news_rows);
}

// ---- shared-user runs: the per-row signature of digat_encoder_fwd, the grouped arithmetic of layer 0 --------------------------
size_t digat_encoder_shared_workspace_bytes(int B, int N, int H, int C, int d, int depth) {
    // + same[B] bytes, leader_of[B] int, is_leader[B] bytes, lead[B] bytes
    return digat_encoder_workspace_bytes(B, N, H, C, d, depth) + 3 * align_up((size_t)B, 256) + align_up((size_t)B * 4, 256);
}

int digat_encoder_fwd_shared(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn, const float* ue,
                             const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx, const float* c_n0, float* out_news,
                             float* out_user, int B, int N, int H, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !ue || !Au || !cat_mask || !cat_idx || !workspace) return DIGAT_ERR_ARG;
    if (B < 0 || N <= 0 || H < 0) return DIGAT_ERR_ARG;
    const int d = p->d, C = p->category_num, U = H + C;
    if (d <= 0 || d % 4 || C < 0) return DIGAT_ERR_SHAPE;
    if (workspace_bytes < digat_encoder_shared_workspace_bytes(B, N, H, C, d, p->depth)) return DIGAT_ERR_WORKSPACE;
    if (B == 0) return DIGAT_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t base = digat_encoder_workspace_bytes(B, N, H, C, d, p->depth);
    uint8_t* same = (uint8_t*)workspace + base;
    uint8_t* is_leader = same + align_up((size_t)B, 256);
    uint8_t* lead = is_leader + align_up((size_t)B, 256);
    int* leader_of = (int*)(lead + align_up((size_t)B, 256));
    {
        ProfScope prof(DIGAT_KERNEL_GLUE, (double)B * ((double)H * d * 4 + (double)U * U + (C + 1) + 8.0 * H), st);
        hipLaunchKernelGGL(user_rows_same_kernel, dim3(B), dim3(256), 0, st, (const uint4*)ue, Au, cat_mask, cat_idx, B, (long)H * d / 4, U * U,
                           C + 1, H, same);
        DIGAT_CHECK_LAUNCH();
        hipLaunchKernelGGL(shared_runs_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)same, B, SPARSE_L0_ROWS, leader_of, is_leader, lead);
        DIGAT_CHECK_LAUNCH();
    }
    return encoder_fwd_impl(p, Xn_in, An, Mn, ue, Au, cat_mask, cat_idx, c_n0, out_news, out_user, B, N, H, workspace, base, stream,
                            leader_of, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, is_leader, lead);
}

int digat_news_context_queries(const digat_params* p, const float* c_n, float* out, int M, void* stream) {
    if (!p || !c_n || !out || M < 0) return DIGAT_ERR_ARG;
    if (!p->user_news_fold_W || !p->userAtt_fold_W) return DIGAT_ERR_ARG;          // folded inference path only
    const int d = p->d, L = p->depth;
    if (d <= 0 || d % 4) return DIGAT_ERR_SHAPE;
    if (M == 0) return DIGAT_OK;
    const size_t md = (size_t)M * d;
    // exactly the launch the encoder makes from the batch's c_n0 before layer 0 (rows are independent of the batch they sit in)
    GemmArgs g = gemm_plain(c_n, d, p->user_news_fold_W, p->user_news_fold_b, out, d, M, d, d, 0);
    g.w[1] = p->userAtt_fold_W; g.bias[1] = p->userAtt_fold_b; g.y[1] = out + md;
    g.nsegs = 2;
    if (L > 0) { g.w[2] = p->user[0].F3; g.bias[2] = p->user[0].b3; g.y[2] = out + 2 * md; g.nsegs = 3; }
    g.wsplit = (const unsigned short*)p->ctx_wsplit[0];
    g.format = (p->flags & DIGAT_PARAMS_GEMM_F16X3) ? 1 : 0; g.range_flag = g.format ? (unsigned*)p->range_flag : nullptr;
    g.m_dispatch = (p->flags & DIGAT_PARAMS_BD_TILED) ? (1 << 30) : 1;      // the kernel the encoder uses for this linear inside a batch: same bits
    return launch_gemm(g, (hipStream_t)stream);
}

int digat_user_project0(const digat_params* p, const float* X, float* hpq, int M, void* stream) {
    if (!p || !X || !hpq || M < 0 || p->depth <= 0) return DIGAT_ERR_ARG;
    const int d = p->d;
    if (d <= 0 || d % 4) return DIGAT_ERR_SHAPE;
    if (M == 0) return DIGAT_OK;
    const digat_layer_params& lu = p->user[0];
    const size_t nd = (size_t)M * d;
    GemmArgs gg = gemm_plain(X, d, lu.W, lu.bW, hpq, d, M, d, d, 0);       // the groups' projection launch of layer 0, row by row
    gg.w[1] = lu.F1; gg.bias[1] = nullptr; gg.y[1] = hpq + nd;
    gg.w[2] = lu.F2; gg.bias[2] = nullptr; gg.y[2] = hpq + 2 * nd;
    gg.nsegs = 3;
    gg.x3_segs = (p->flags & DIGAT_PROJ_PQ_X3) ? 6 : 0;
    gg.wsplit = (const unsigned short*)lu.wsplit;
    gg.format = (p->flags & DIGAT_PARAMS_GEMM_F16X3) ? 1 : 0; gg.range_flag = gg.format ? (unsigned*)p->range_flag : nullptr;
    gg.m_dispatch = 1 << 30;                                               // the large-M kernel whatever M is (C topic rows)
#ifdef DIGAT_LAB
    if (LAB_ENV("DIGAT_LAB_FAKE_PRESPLIT", 0) && gg.format == 1) gg.a_split = X;     // timing only (tools/exp/gemm_lab.py): X read as if already split
#endif
    return launch_gemm(gg, (hipStream_t)stream, DIGAT_KERNEL_PROJ);
}

int digat_news_project0(const digat_params* p, const float* Xn, float* hpq, int M, int N, void* stream) {
    if (!p || !Xn || !hpq || M < 0 || N <= 0 || p->depth <= 0) return DIGAT_ERR_ARG;
    const int d = p->d;
    if (d <= 0 || d % 4 || (long)M * N > 0x7fffffffL / 4) return DIGAT_ERR_SHAPE;
    if (M == 0) return DIGAT_OK;
    const digat_layer_params& ln = p->news[0];
    const size_t ndn = (size_t)M * N * d;
    // exactly the launch the encoder makes for layer 0 of the news graph (rows are independent of the batch they sit in)
    GemmArgs gp = gemm_plain(Xn, d, ln.W, ln.bW, hpq, d, M * N, d, d, 0);
    gp.w[1] = ln.F1; gp.bias[1] = nullptr; gp.y[1] = hpq + ndn;
    gp.w[2] = ln.F2; gp.bias[2] = nullptr; gp.y[2] = hpq + 2 * ndn;
    gp.nsegs = 3;
    gp.x3_segs = (p->flags & DIGAT_PROJ_PQ_X3) ? 6 : 0;
    gp.wsplit = (const unsigned short*)ln.wsplit;
    gp.format = (p->flags & DIGAT_PARAMS_GEMM_F16X3) ? 1 : 0; gp.range_flag = gp.format ? (unsigned*)p->range_flag : nullptr;
    gp.m_dispatch = 1 << 30;
    return launch_gemm(gp, (hipStream_t)stream, DIGAT_KERNEL_PROJ);
}

static double g_prof_last_live_fraction = -1.0;
double digat_profile_live_row_fraction(void) { return g_prof_last_live_fraction; }
static double g_prof_last_gemm_bytes[DIGAT_KERNEL_KINDS] = {0.0};
static double g_prof_last_part_ms[XATTN_PARTS] = {0.0}, g_prof_last_part_bytes[XATTN_PARTS] = {0.0};
static int g_prof_last_part_launches[XATTN_PARTS] = {0};
// After digat_profile_stop: the DIGAT_KERNEL_XATTN launches by kernel — [0] user graph layers >= 1 (row lists: twin kernel),
// [1] user graph layer 0 of grouped rows, [2] news graphs in LDS (n <= 16), [3] everything else; ms, algorithmic bytes, launches.
int digat_profile_xattn_parts(double* ms, double* bytes, int* launches) {
    for (int p = 0; p < XATTN_PARTS; ++p) {
        if (ms) ms[p] = g_prof_last_part_ms[p];
        if (bytes) bytes[p] = g_prof_last_part_bytes[p];
        if (launches) launches[p] = g_prof_last_part_launches[p];
    }
    return DIGAT_OK;
}
int digat_profile_gemm_bytes(double* bytes_per_kind) {
    if (!bytes_per_kind) return DIGAT_ERR_ARG;
    for (int k = 0; k < DIGAT_KERNEL_KINDS; ++k) bytes_per_kind[k] = g_prof_last_gemm_bytes[k];
    return DIGAT_OK;
}

// recording costs two hipEventRecord calls per launch on the host: a caller that wants per-kernel times over a long
// region without slowing it down samples it — pause(1) ... pause(0) around the steps it does not want recorded
int digat_profile_pause(int paused) {
    if (!g_prof.ev) return DIGAT_ERR_ARG;
    g_prof.enabled = paused ? 0 : 1;
    return DIGAT_OK;
}

// A one-thread kernel whose only purpose is to be visible in a kernel trace (rocprofv3 --kernel-trace): bench.py launches one
// at each end of its timed region, so that per-kernel averages of exactly that region can be cut out of the trace
// (tools/trace_region.py) and held against the library's own event timings.
__global__ void digat_region_marker_kernel(int id, int* sink) { if (sink && id < 0) *sink = id; }
int digat_profile_marker(int id, void* stream) {
    hipLaunchKernelGGL(digat_region_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, id, (int*)nullptr);
    DIGAT_CHECK_LAUNCH();
    return DIGAT_OK;
}

// Which kernel kinds get their two events: every event pair is a pair of marker packets in the launch's queue and costs the
// overlapped encoder about 1 % of a step per 10 pairs; a measurement that only needs the dominant kernels says so.
int digat_profile_set_kinds(unsigned mask) {
    const unsigned prev = g_prof.kind_mask;
    g_prof.kind_mask = mask;
    return (int)prev;
}

int digat_profile_start(int max_launches) {
    if (max_launches <= 0) return DIGAT_ERR_ARG;
    if (g_prof.ev) return DIGAT_ERR_ARG;          // already running
    g_prof.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * max_launches);
    g_prof.kind = (int*)malloc(sizeof(int) * max_launches);
    g_prof.work = (double*)malloc(sizeof(double) * max_launches);
    g_prof.bytes = (double*)malloc(sizeof(double) * max_launches);
    g_prof.part = (int*)malloc(sizeof(int) * max_launches);
    if (!g_prof.ev || !g_prof.kind || !g_prof.work || !g_prof.bytes || !g_prof.part) return DIGAT_ERR_ARG;
    for (int i = 0; i < 2 * max_launches; ++i)
        if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return DIGAT_ERR_LAUNCH;
    if (hipMalloc((void**)&g_prof.rows_dev, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(g_prof.rows_dev, 0, 16 * sizeof(unsigned long long)) != hipSuccess) return DIGAT_ERR_LAUNCH;
    for (int k = 0; k < 16; ++k) { g_prof.flops_per_row[k] = 0.0; g_prof.rows_nominal[k] = 0.0; g_prof.bytes_per_row[k] = 0.0; }
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
        g_prof_last_gemm_bytes[k] = 0.0;
    }
    for (int p = 0; p < XATTN_PARTS; ++p) { g_prof_last_part_ms[p] = 0.0; g_prof_last_part_bytes[p] = 0.0; g_prof_last_part_launches[p] = 0; }
    int rc = DIGAT_OK;
    for (int i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) { rc = DIGAT_ERR_LAUNCH; continue; }
        const int k = g_prof.kind[i];
        if (ms_per_kind) ms_per_kind[k] += ms;
        if (work_per_kind) work_per_kind[k] += g_prof.work[i];
        if (launches_per_kind) launches_per_kind[k] += 1;
        g_prof_last_gemm_bytes[k] += g_prof.bytes[i];
        if (k == DIGAT_KERNEL_XATTN) {
            const int p = g_prof.part[i] >= 0 && g_prof.part[i] < XATTN_PARTS ? g_prof.part[i] : XPART_OTHER;
            g_prof_last_part_ms[p] += ms; g_prof_last_part_bytes[p] += g_prof.work[i]; g_prof_last_part_launches[p] += 1;
        }
    }
    g_prof_last_live_fraction = -1.0;
    if (g_prof.rows_dev) {
        unsigned long long rows[16];
        if (hipMemcpy(rows, g_prof.rows_dev, sizeof(rows), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int k = 0; k < DIGAT_KERNEL_KINDS; ++k) {
                if (work_per_kind) work_per_kind[k] += (double)rows[k] * g_prof.flops_per_row[k];
                if (k == DIGAT_KERNEL_PROJ || k == DIGAT_KERNEL_LINEAR) g_prof_last_gemm_bytes[k] += (double)rows[k] * g_prof.bytes_per_row[k];
            }
            for (int p = 0; p < XATTN_PARTS; ++p) {         // Eq. 8 launches on live-row lists: BYTES counted on the device
                if (work_per_kind) work_per_kind[DIGAT_KERNEL_XATTN] += (double)rows[8 + p];
                g_prof_last_part_bytes[p] += (double)rows[8 + p];
            }
            if (g_prof.rows_nominal[DIGAT_KERNEL_PROJ] > 0)
                g_prof_last_live_fraction = (double)rows[DIGAT_KERNEL_PROJ] / g_prof.rows_nominal[DIGAT_KERNEL_PROJ];
        }
        (void)hipFree(g_prof.rows_dev);
        g_prof.rows_dev = nullptr;
    }
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    free(g_prof.ev); free(g_prof.kind); free(g_prof.work); free(g_prof.bytes); free(g_prof.part);
    g_prof.ev = nullptr; g_prof.kind = nullptr; g_prof.work = nullptr; g_prof.bytes = nullptr; g_prof.part = nullptr; g_prof.cap = g_prof.used = 0;
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
#include "digat_train_abi.inc"
#include "digat_eval.inc"
#include "digat_news.inc"
#include "digat_news_train.inc"
#include "digat_gat.inc"
#include "digat_sag.inc"
