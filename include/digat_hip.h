/*
 * digat_hip.h — C ABI of the MI355X (gfx950) DIGAT dual-graph interaction hot path.
 *
 * The reference (Veason-silverbullet/DIGAT) has no FFI: its boundary for this path is the Python
 * class contract of graphEncoders.DIGAT (graphEncoders.py:48-198), selected by
 * `--graph_encoder=DIGAT` (model.py:18-19).  This header is the boundary a native implementation
 * of that class binds to; every entry point names the reference function it replaces.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to contiguous row-major data, 16-byte aligned;
 *     fp32 unless stated; adjacency / masks are one byte per element (torch.bool), 0 = absent;
 *     category indices are int64 (MIND_corpus.py:149);
 *   - nn.Linear weights are [out, in] exactly as in the reference's state_dict;
 *   - d (news_embedding_dim) must be a multiple of 4; graph sizes n <= DIGAT_MAX_NODES;
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no allocation, no host sync:
 *     scratch comes from the caller (`workspace`, size from the matching *_workspace_bytes);
 *   - inputs are never written; outputs may alias the `addend` argument where one exists;
 *   - returns DIGAT_OK (0) or a DIGAT_ERR_* code, never throws.  Eval-mode semantics
 *     (dropout = identity), i.e. what DIGAT.inference / model.eval() computes;
 *   - threading: calls on DIFFERENT streams may be in flight together — from one host thread or several — as long as each stream
 *     has its own workspace and is driven by one thread at a time (digat_amd.util.score_rows alternates three).  Everything a
 *     call's behaviour depends on travels WITH the call (digat_params.flags: Eq. 8 variant, operand format, side stream, live-row
 *     lists); the library keeps no mutable process-wide switch.  Per-caller-stream side streams live in a mutex-guarded table.
 *     Exceptions, both diagnostics: the launch profiler (digat_profile_*: one measuring thread) and digat_set_train_precision
 *     (set once per training run).
 *   - no environment variable changes what the product library computes or which kernels it runs: the development knobs of
 *     earlier rounds (A/B switches, the wrong-result timing ablations DIGAT_*_SKIP, phase timers, the LDS-staged Eq. 8 variants)
 *     exist only in LAB builds (-DDIGAT_LAB, tools/exp/build_variant.sh).
 */
#ifndef DIGAT_HIP_H
#define DIGAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): digat_xattn_fwd_train / digat_xattn_bwd / digat_user_ctx_fwd_train / digat_user_ctx_bwd take a ready-made split image
 * (or NULL) before `stream`; the Eq. 8 pair takes the layer's input dropout (p_in, seed_in).  3: digat_params grew featureAffine_fsplit; digat_news_ctx_bwd / digat_user_ctx_bwd took
 * `accumulate_params` in round 5 and entry points were added without a bump: a loader built for one version must refuse a library
 * of another (digat_amd/_lib.py does) rather than shift arguments. */
#define DIGAT_ABI_VERSION 4
#define DIGAT_MAX_NODES 128
#define DIGAT_MAX_DEPTH 16

enum {
    DIGAT_OK = 0,
    DIGAT_ERR_ARG = 1,        /* null pointer / negative size */
    DIGAT_ERR_SHAPE = 2,      /* d % 4 != 0, n > DIGAT_MAX_NODES, depth > DIGAT_MAX_DEPTH ... */
    DIGAT_ERR_WORKSPACE = 3,  /* workspace too small */
    DIGAT_ERR_LAUNCH = 4      /* the HIP runtime refused a launch */
};

int digat_version(void);
const char* digat_error_string(int code);

/* ---- nn.Linear on the matrix cores: y[M,N] = x[M,K] @ w[N,K]^T + b[N]  (b may be NULL) --------
 * Replaces the aten::addmm / mm calls of graphEncoders.py:146-149,166-169 (exact fp32:
 * v_mfma_f32_16x16x4_f32).  ldx / ldy are row strides in floats. */
int digat_linear_f32(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy,
                     int M, int N, int K, void* stream);

/* The same product on the bf16 / fp16 matrix cores: every fp32 operand is split exactly into three bf16
 * pieces and the six significant partial products are summed in the fp32 accumulator (format DIGAT_GEMM_BF16X6), or into
 * two scaled fp16 pieces with three products (DIGAT_GEMM_F16X3; range and accuracy: see the enum below).  `wsplit`
 * (digat_split_weights_bytes(N, K) bytes, enough for either format) receives the split weights as ready-made LDS images
 * in the format the caller names; the library remembers the format of every image it has split (by address) until
 * digat_forget_split_image(wsplit) — call it before the buffer is freed or reused for anything else — and refuses a launch
 * that names the other one (DIGAT_ERR_ARG).  N % 80 == 0, K % 8 == 0; any M (from 2048 rows up the strip-mined tiled kernel,
 * below the split-image [B,d] kernel — same split operands, same accuracy). */
size_t digat_split_weights_bytes(int rows, int K);
int digat_forget_split_image(const void* wsplit);      /* 0 = forgotten, DIGAT_ERR_ARG = not an image the library knows */
int digat_split_proj_weights(const float* W, const float* F1, const float* F2, int d, void* wsplit, int format, void* stream);
int digat_split_weights(const float* W, int N, int K, void* wsplit, int format, void* stream);    /* one [N,K] matrix */
int digat_linear_f32x3(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy,
                       int M, int N, int K, void* wsplit, int format, void* stream);

/* ---- a1 / a2: DIGAT.compute_news_graph_embeddings / compute_user_graph_embeddings -------------
 * graphEncoders.py:143-154 / :163-174 (Eq. 8).  X [B,n,d], A [B,n,n] bytes, ctx [B,d] (the OTHER
 * graph's context); W [d,d]+bW, F1 (ffn1, neighbour j), F2 (ffn2, centre i), F3 [d,d]+b3, a [d]
 * (the [1,d] weight of *_graph_attention_a).  out [B,n,d] = relu(alpha @ h) + X.
 * alpha_out [B,n,n] is optional (NULL to skip). */
size_t digat_xattn_workspace_bytes(int B, int n, int d);
int digat_xattn_fwd(const float* X, const uint8_t* A, const float* ctx,
                    const float* W, const float* bW, const float* F1, const float* F2,
                    const float* F3, const float* b3, const float* a,
                    float* out, float* alpha_out, int B, int n, int d,
                    void* workspace, size_t workspace_bytes, void* stream);
/* The same without alpha_out and with the Eq. 8 variant named (DIGAT_XATTN_DENSE or DIGAT_XATTN_SPARSE, below): the sparse
 * kernel visits only the adjacency entries (graphs with a few entries per node; n > 16, d <= 1024), results equal to
 * fp32 summation order. */
int digat_xattn_fwd_mode(const float* X, const uint8_t* A, const float* ctx,
                         const float* W, const float* bW, const float* F1, const float* F2,
                         const float* F3, const float* b3, const float* a,
                         float* out, int B, int n, int d, int mode,
                         void* workspace, size_t workspace_bytes, void* stream);

/* One Eq. 8 layer on the reduced-precision operand path of BASELINE configs[4], directly callable (tests, micro-benchmarks):
 * the projection GEMM on the split image `wsplit` (digat_split_proj_weights(W, F1, F2, d, wsplit, format, ...)) stores P' = K3 + K1
 * and Q = K2 as bf16 (pq = 1, DIGAT_PQ_BF16) or as block-scaled e4m3 rows (pq = 2, DIGAT_PQ_FP8: the layout documented there),
 * pq = 0: fp32; the sparse Eq. 8 kernel reads them.  Needs B n >= 2048, n > 16, d % 80 == 0, d <= 1024 (DIGAT_ERR_SHAPE otherwise);
 * workspace as digat_xattn_fwd.  (graphEncoders.py:143-154 / :163-174; README.md:62-66) */
int digat_xattn_fwd_lowprec(const float* X, const uint8_t* A, const float* ctx,
                            const float* W, const float* bW, const float* F1, const float* F2,
                            const float* F3, const float* b3, const float* a, const void* wsplit, int format,
                            float* out, int B, int n, int d, int pq,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The Eq. 8 pairwise part alone, on already-projected inputs: h = X W^T + bW, Q = X F2^T and
 * Pr = (ctx F3^T + b3) + X F1^T, i.e. K3 + K1 already summed in the reference's left-to-right order:
 * score -> leaky_relu(0.2) -> -1e9 mask -> softmax_j (written to alpha [B,n,n], required)
 * -> out = relu(alpha @ h) + X.  Two launches: the score kernel bench.py prices against the HBM
 * roofline, and the aggregation on the matrix cores. */
int digat_xattn_pairwise_fwd(const float* Pr, const float* Q, const float* h, const float* X,
                             const float* a, const uint8_t* A,
                             float* out, float* alpha, int B, int n, int d, void* stream);

/* ---- a3: DIGAT.compute_news_graph_context  (graphEncoders.py:109-114, layers.py:199-206) ------
 * X [B,N,d], mask [B,N] bytes; Kc/Qc/bQc = candidate_attention.{K.weight,Q.weight,Q.bias},
 * Wg [d,2d]/bg = news_graph_W.  out [B,d] = addend + context (addend may be NULL or == out). */
size_t digat_news_ctx_workspace_bytes(int B, int N, int d);
int digat_news_ctx_fwd(const float* X, const uint8_t* mask,
                       const float* Kc, const float* Qc, const float* bQc,
                       const float* Wg, const float* bg,
                       const float* addend, float* out, int B, int N, int d,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- a4 (+a7): DIGAT.compute_user_graph_context  (graphEncoders.py:123-134) -------------------
 * Xu [B,U,d] (first H nodes = history), cat_mask [B,C1] bytes, cat_idx [B,H] int64 in [0,C1),
 * c_n [B,d].  C1 = category_num + 1.  Contains the torch_scatter scatter_softmax / scatter_sum
 * replacement (topic pooling).  out [B,d] = addend + context. */
size_t digat_user_ctx_workspace_bytes(int B, int U, int H, int C1, int d);
int digat_user_ctx_fwd(const float* Xu, const uint8_t* cat_mask, const int64_t* cat_idx, const float* c_n,
                       const float* Ku, const float* Qu, const float* bQu,
                       const float* Fa, const float* bFa,
                       const float* Kua, const float* Qua, const float* bQua,
                       const float* addend, float* out, int B, int U, int H, int C1, int d,
                       void* workspace, size_t workspace_bytes, void* stream);

/* topic pooling alone: out [B,C1,d] = scatter_sum(scatter_softmax(a) * hist) with
 * a_t = hist_t . kq / sqrt(d); kq [B,d] = (c_n Qu^T + bQu) Ku.  (graphEncoders.py:126-130) */
int digat_topic_pool_fwd(const float* Xu, const float* kq, const int64_t* cat_idx, float* out,
                         int B, int U, int H, int C1, int d, void* stream);

/* ---- a5: DIGAT.forward / DIGAT.inference  (graphEncoders.py:177-198) ---------------------------
 * Parameter block: the reference's state_dict tensors by name (prefix graph_encoder. in Model). */
/* Eq. 8 of the USER graph inside the encoder entry points (digat_params.flags, bits 0-1).  MIND user graphs hold ~4 entries
 * per adjacency row; xattn_sparse_kernel visits only those (score, softmax and aggregation in one launch), the dense pair
 * (tile score kernel + MFMA aggregation) streams whole rows.  AUTO: both are launched and a device-side count of the
 * adjacency entries (<= 20 per node on average: sparse) lets one of them return at its first instruction — no host
 * synchronisation; results differ between the two only by fp32 summation order.  Callers that know their graphs choose. */
enum { DIGAT_XATTN_AUTO = 0, DIGAT_XATTN_DENSE = 1, DIGAT_XATTN_SPARSE = 2 };
/* digat_params.flags bit 2 (folded inference path, bf16x6 projections only): the P and Q projections of Eq. 8 — which only
 * feed the attention score — with the three leading bf16 products (hi*hi, mid*hi, hi*mid: relative error ~2^-16) instead
 * of six (~2^-24); h, which carries the values, keeps six.  A third fewer matrix instructions in the projection GEMM. */
enum { DIGAT_PROJ_PQ_X3 = 4 };
/* digat_params.flags bit 3: the NEWS graph's Eq. 8 with the sparse kernel too (news graphs of more than 16 nodes; smaller
 * ones always take the wave-per-centre small-graph kernel).  SAG news graphs are breadth-first trees plus a few cross
 * edges: 3-4 entries per node at N = 26 or 65. */
enum { DIGAT_NEWS_XATTN_SPARSE = 8 };
/* digat_params.flags bit 4 (BASELINE configs[4]; folded inference path, bf16x6 projections, sparse Eq. 8: the user graph's
 * layers >= 1 — the launches that carry most of the bytes — and news graphs of more than 16 nodes at every layer): the
 * projection GEMM stores P' = K3 + K1 and Q = K2 in bf16
 * (round to nearest even) and the Eq. 8 kernel reads them as such; the score is still accumulated in fp32, h (the values), X
 * and every output stay fp32.  This is the reference's own "faster inference" idea — a quantised K3 + K1 + K2
 * (README.md:62-66).  Ranking metrics move by < 1e-4 (tests/test_hip_lowprec.py); element-wise the contexts move by ~1e-4.
 * Bit 5: the P and Q segments with the leading bf16 product alone (2^-8 per product) instead of the number bit 2 selects. */
enum { DIGAT_PQ_BF16 = 16, DIGAT_PQ_X1 = 32 };
/* digat_params.flags bit 8 (BASELINE configs[4], "fp8 Eq.-8 inference path"; same launches as bit 4, which wins when both are
 * set): the projection GEMM stores P' = K3 + K1 and Q = K2 as BLOCK-SCALED OCP e4m3 (v_cvt_pk_fp8_f32) — one block per (row,
 * 80-column strip of the GEMM's wave tile), scale = the block's absmax / 448 as fp32, laid out per row as
 * [d codes | d / 80 scales | pad to 64 bytes] (d = 400: 448 bytes instead of 1 600) — and the Eq. 8 kernel reads 8-code pieces,
 * rescales in registers (one fma per channel: p s_p + q) and accumulates the score in fp32; h, X and every output stay fp32.
 * Requires d % 80 == 0.  Measured drift of AUC / MRR / nDCG on the trained reference-pinned dev set: tests/test_hip_lowprec.py. */
enum { DIGAT_PQ_FP8 = 256 };
/* digat_params.flags bit 6: every wsplit image of this parameter block (layers' [W|ffn1|ffn2], featureAffine) was split with
 * format DIGAT_GEMM_F16X3 (two scaled fp16 pieces, three products); clear = DIGAT_GEMM_BF16X6.  See digat_split_proj_weights. */
enum { DIGAT_PARAMS_GEMM_F16X3 = 64 };
/* digat_params.flags bit 7: which kernel runs the [B,d] linears of the folded inference path (context queries, candidate query, K3 of the
 * news graph) and digat_news_context_queries' table — set: the tiled split-operand kernel (right for passes of >= 2 048 rows), clear:
 * the split-image [B,d] kernel (right below that).  The CALLER names it, the row count does not choose: a row's bits then do not
 * depend on the batch it sits in, nor on whether its queries were computed in the batch or read from the per-news table. */
enum { DIGAT_PARAMS_BD_TILED = 128 };

typedef struct digat_layer_params {
    const float *W, *bW;      /* {g}_graph_attention_W.i.{weight,bias}    */
    const float *F1;          /* {g}_graph_attention_ffn1.i.weight        */
    const float *F2;          /* {g}_graph_attention_ffn2.i.weight        */
    const float *F3, *b3;     /* {g}_graph_attention_ffn3.i.{weight,bias} */
    const float *a;           /* {g}_graph_attention_a.i.weight  [1,d]    */
    const void  *wsplit;      /* optional: [W|ffn1|ffn2] pre-split by digat_split_proj_weights (opaque: the
                                 three bf16 pieces of every weight, arranged as LDS images per 80-row strip and
                                 32-deep K tile; digat_split_weights_bytes(3d, d) bytes).  Non-NULL runs the
                                 node projections as six bf16 MFMA products per fp32 product ("bf16x6":
                                 fp32-equivalent accuracy, 6/16 of the fp32-MFMA cost); NULL = fp32 MFMA. */
    const void  *f3_wsplit;   /* optional: ffn3.weight split by digat_split_weights(F3, d, d, ...) in the block's format: K3 = ctx F3^T + b3
                                 ([B,d] rows) then runs on the split image too (news graph; the user graph's F3 travels inside
                                 digat_params.ctx_wsplit).  NULL = fp32 MFMA */
} digat_layer_params;

typedef struct digat_params {
    int32_t d;                /* news_embedding_dim                        */
    int32_t depth;            /* graph_depth                               */
    int32_t category_num;     /* C (topic_node_embedding rows)             */
    int32_t flags;            /* bits 0-1: Eq. 8 of the user graph, DIGAT_XATTN_AUTO / _DENSE / _SPARSE; bit 2: DIGAT_PROJ_PQ_X3; bit 3: DIGAT_NEWS_XATTN_SPARSE (see below) */
    const float *topic_node_embedding;                      /* [C,d]        */
    const float *cand_K, *cand_Q, *cand_bQ;                 /* candidate_attention */
    const float *news_graph_W, *news_graph_b;               /* [d,2d], [d]  */
    const float *user_news_K, *user_news_Q, *user_news_bQ;
    const float *featureAffine_W, *featureAffine_b;
    const float *userAtt_K, *userAtt_Q, *userAtt_bQ;        /* userAttention */
    digat_layer_params news[DIGAT_MAX_DEPTH];
    digat_layer_params user[DIGAT_MAX_DEPTH];
    /* Optional (all three pairs or none; NULL = unfolded path): attention queries with the key
     * projection folded in by digat_fold_attention — Wf = K^T Q [d,d], bf = K^T bQ [d] — for
     * candidate_attention, user_news_{K,Q} and userAttention.  Weight-only preprocessing, valid
     * while the weights do not change (inference). */
    const float *cand_fold_W, *cand_fold_b;
    const float *user_news_fold_W, *user_news_fold_b;
    const float *userAtt_fold_W, *userAtt_fold_b;
    const void  *featureAffine_wsplit;   /* optional: featureAffine.weight split by digat_split_weights (same format as the layers') */
    /* Optional split images (the block's format) of the [B,d] linears of the folded inference path; NULL = fp32 MFMA for that one:
     * cand_fold_wsplit: cand_fold_W [d,d]; gate_wsplit: news_graph_W [d,2d];
     * ctx_wsplit[l], l = 0 .. depth: the three matrices applied to the news context before layer l, stacked
     * [user_news_fold_W ; userAtt_fold_W ; user[l].F3] (3d rows; the last entry, l = depth, has no F3: 2d rows are read),
     * each made by digat_split_proj_weights(user_news_fold_W, userAtt_fold_W, user[l].F3, d, ...). */
    const void  *cand_fold_wsplit, *gate_wsplit;
    const void  *ctx_wsplit[DIGAT_MAX_DEPTH + 1];
    uint32_t    *range_flag;             /* optional, DIGAT_PARAMS_GEMM_F16X3 only: one device word the fp16x3 GEMMs OR 1 into when an
                                            activation is at or beyond the format's range (|x| >= 4094 or inf; a NaN input does not raise it — it reaches the outputs as NaN, as in
                                            the reference): the caller zeroes it
                                            before a scoring run and reads it after (digat_amd/util.py re-scores in bf16x6 when set) */
    const void  *featureAffine_fsplit;   /* optional (round 6), DIGAT_PARAMS_GEMM_F16X3 only: featureAffine.weight as the lane-ordered image of
                                            digat_split_ctx_fused_weights — with it (and the folded queries) compute_user_graph_context
                                            (graphEncoders.py:123-134) runs as ONE launch: topic pooling, featureAffine and the SDPA pooling
                                            without T / T' leaving the CU (H <= 52, 192 < d <= 448, C + 1 <= 20; other shapes keep the three launches) */
} digat_params;

/* featureAffine.weight [d,d] -> the fused user-context kernel's image (fp16x3 pieces in the kernel's lane order). */
size_t digat_split_ctx_fused_bytes(int d);
int digat_split_ctx_fused_weights(const float* W, int d, void* image, void* stream);

/* (K x).(Q c + bQ) = x.(Wf c + bf): fold one ScaledDotProductAttention / user_news pair.
 * K, Q [d,d] nn.Linear weights, bQ [d] or NULL; outputs Wf [d,d] (as an nn.Linear weight), bf [d]. */
size_t digat_fold_workspace_bytes(int d);
int digat_fold_attention(const float* K, const float* Q, const float* bQ, float* Wf, float* bf, int d,
                         void* workspace, size_t workspace_bytes, void* stream);

/* news_graph_embeddings [B,N,d], news_graph [B,N,N], news_graph_mask [B,N],
 * user_news_embedding [B,H,d], user_graph [B,U,U] (U = H + C), user_category_mask [B,C+1],
 * user_category_indices [B,H] int64, news_graph_context [B,d] or NULL.
 * NULL context = DIGAT.forward in eval mode (c_n0 computed here, :180); non-NULL = DIGAT.inference.
 * Outputs: the 2-tuple (news_graph_context, user_graph_context), each [B,d]. */
size_t digat_encoder_workspace_bytes(int B, int N, int H, int C, int d, int depth);
int digat_encoder_fwd(const digat_params* params,
                      const float* news_graph_embeddings, const uint8_t* news_graph,
                      const uint8_t* news_graph_mask, const float* user_news_embedding,
                      const uint8_t* user_graph, const uint8_t* user_category_mask,
                      const int64_t* user_category_indices, const float* news_graph_context,
                      float* out_news_context, float* out_user_context,
                      int B, int N, int H,
                      void* workspace, size_t workspace_bytes, void* stream);

/* With the folded-query fields set, digat_encoder_fwd / _grouped can run the news-graph kernels of a layer (small: N nodes,
 * [B,d] linears) on an internal side stream — one per caller stream — under the user graph's Eq. 8 and join before the user
 * context is pooled.  digat_params.flags bits 9-10 choose: neither = by pass size (on below 2 048 rows, where those kernels
 * are a few waves of workgroups each; off from 2 048 rows up, where every kernel fills the chip by itself),
 * DIGAT_PARAMS_SIDE_STREAM_OFF = never, DIGAT_PARAMS_SIDE_STREAM_ON = always (right for a driver that keeps a single pass in
 * flight).  Results do not depend on it, bit for bit.
 * Bit 11, DIGAT_PARAMS_NO_LIVE_ROWS: project, score and write EVERY user-graph node in every layer.  Default (clear): nodes that
 * cannot reach an output — history padding slots, topic nodes of unread categories: only their self loop, pooled with weight 0 —
 * are found on the device and skipped (about half of a MIND user graph).  Outputs are bit-identical either way. */
enum { DIGAT_PARAMS_SIDE_STREAM_OFF = 512, DIGAT_PARAMS_SIDE_STREAM_ON = 1024, DIGAT_PARAMS_NO_LIVE_ROWS = 2048 };

#ifdef DIGAT_LAB
/* LAB builds only (process-wide, one host thread): sparse Eq. 8 of the user graph from LDS-staged rows (digat_staged.inc; five
 * variants, mode 1-5; 0 = the wave-per-centre kernel).  Measured slower in round 2: the product
 * library does not carry the code path. */
int digat_set_staged_xattn(int mode);
#endif

/* The same inference for rows that SHARE users: in dev/test scoring the ~37 candidate rows of one
 * impression carry identical user tensors (util.py:57-67 expands them per row).  Here the user side is
 * passed once per group — user_news_embedding [G,H,d], user_graph [G,U,U], user_category_mask [G,C+1],
 * user_category_indices [G,H] — with row_group [B] (int32, values in [0,G), 4*G <= B) naming each row's
 * group; the news side and the context stay per row.  Layer 0's user projections then run on G*U rows
 * instead of B*U.  Results are bit-identical to digat_encoder_fwd on the expanded tensors.  Needs the
 * folded-query fields of digat_params. */
size_t digat_encoder_grouped_workspace_bytes(int B, int N, int H, int C, int d, int depth);
int digat_encoder_fwd_grouped(const digat_params* params,
                              const float* news_graph_embeddings, const uint8_t* news_graph,
                              const uint8_t* news_graph_mask, const float* user_news_embedding_groups,
                              const uint8_t* user_graph_groups, const uint8_t* user_category_mask_groups,
                              const int64_t* user_category_indices_groups, const int32_t* row_group,
                              const float* news_graph_context, float* out_news_context, float* out_user_context,
                              int B, int G, int N, int H, void* workspace, size_t workspace_bytes, void* stream);

/* Layer 0's projections of the NEWS graph ([h|P|Q] = X_n [W|ffn1|ffn2]^T of news[0]) depend on the candidate news alone, like
 * the news representations and c_n0 the reference's driver caches per news (util.py:24-44).  digat_news_project0 computes
 * them for M news graphs: Xn [M,N,d] -> hpq [3][M,N,d] (the launch the encoder itself makes; M*N >= 2048 rows for the bf16x6
 * kernel).  digat_encoder_fwd_grouped_cached takes the batch's rows of that table (news_hpq0 [3][B,N,d], gathered by the
 * caller; NULL = compute as usual) and skips the projection GEMM of layer 0 when N <= 16.  Results are bit-identical.
 * With news_index (the candidate ids; news_graph_embeddings and news_hpq0 are then the whole per-news tables, news_rows rows) the
 * rows are read IN PLACE: by the graph-in-LDS kernel for N <= 16, and — round 4 — by the sparse Eq. 8 kernel for larger news graphs
 * when digat_params.flags has DIGAT_NEWS_XATTN_SPARSE (N = 26, 65: the layer-0 projection GEMM of B N rows, the largest launch of a
 * MIND-large step, goes away; K3 joins in the kernel in the GEMM epilogue's order: the bits of the in-batch launch). */
int digat_news_project0(const digat_params* params, const float* Xn, float* hpq, int M, int N, void* stream);
/* The USER graph's layer-0 projections are row-wise too: those of a history node depend on that news alone, those of a topic
 * node on nothing.  digat_user_project0: X [M,d] -> hpq [3][M,d] with user[0]'s [W|ffn1|ffn2] (for the news table, M = number
 * of news; for the topic table, X = topic_node_embedding, M = C).  The cached entry takes the groups' history rows
 * hist_hpq0 [3][G,H,d] (gathered by the caller with the history ids) and topic_hpq0 [3][C,d] — both or neither — and assembles
 * the groups' [h|P|Q] from them instead of running the projection GEMM (used when B*U >= 2048: below that the in-batch
 * launch is another kernel).  Bit-identical. */
int digat_user_project0(const digat_params* params, const float* X, float* hpq, int M, void* stream);
/* The first link of the [B,d] chain — the topic query, the user-attention query and the user graph's layer-0 K3, all three linear
 * maps of the candidate's cached news context c_n0 (graphEncoders.py:126,133,169 with news_graph_context given, :189) — depends on
 * the news alone.  digat_news_context_queries: c_n [M,d] -> out [3][M,d] (the encoder's own launch, row by row; folded path), kept
 * per news next to c_n0; the cached entry takes the batch's rows ctxq0 [3][B,d] (or NULL) and skips that launch.  Bit-identical. */
int digat_news_context_queries(const digat_params* params, const float* c_n, float* out, int M, void* stream);
/* news_index [B] int64 (or NULL) with news_rows: the per-news tables are read IN PLACE — news_graph_embeddings is then the table
 * [news_rows, N, d] and news_hpq0 the table [3][news_rows, N, d], row b using their row news_index[b] — instead of the caller
 * gathering the batch's rows (130 MB of copies per 1024-row batch at N = 10).  Needs news_graph_context, news_hpq0 and N <= 16
 * (layer 0 of the news graph is then the only reader of the node table).  Bit-identical. */
int digat_encoder_fwd_grouped_cached(const digat_params* params,
                                     const float* news_graph_embeddings, const uint8_t* news_graph, const uint8_t* news_graph_mask,
                                     const float* user_news_embedding_g, const uint8_t* user_graph_g,
                                     const uint8_t* user_category_mask_g, const int64_t* user_category_indices_g,
                                     const int32_t* row_group, const float* news_graph_context, const float* news_hpq0,
                                     const float* hist_hpq0, const float* topic_hpq0, const float* ctxq0,
                                     const int64_t* news_index, int64_t news_rows,
                                     float* out_news, float* out_user, int B, int G, int N, int H,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* H1: Model.inference's last line (model.py:89): logits[b] = sum_c user_ctx[b,c] * news_ctx[b,c]. */
int digat_row_logits(const float* news_ctx, const float* user_ctx, float* logits, int B, int d, void* stream);

/* ---- training: native forward/backward pairs for every op on the path ----------------------------
 * The reference trains by autograd through DIGAT.forward (trainer.py:98-102).  digat_amd/training.py
 * composes the same forward from these primitives (each an autograd.Function); Eq. 8's backward
 * recomputes relu'(K3+K1+K2) from the saved projections instead of saving [B,n,n,d].  All reductions
 * are ordered (no atomics), so gradients are run-to-run reproducible.  `accumulate` != 0 adds into the
 * destination. */
int digat_linear_bwd_input(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx,
                           int M, int N, int K, int accumulate, void* stream);
/* The same on the bf16 matrix cores (bf16x6: fp32-grade).  W^T is split into `wsplit` (digat_split_weights_bytes(K, N)
 * bytes) by every call — the weights change every optimiser step.  M >= 2048, K % 80 == 0, N % 8 == 0, N >= 32. */
int digat_linear_bwd_input_x3(const float* dy, int64_t lddy, const float* w, float* dx, int64_t lddx,
                              int M, int N, int K, int accumulate, void* wsplit, void* stream);           /* dx[M,K] = dy[M,N] @ w[N,K] */
size_t digat_linear_bwd_weight_workspace(int M, int No, int Ni);
int digat_linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dW, float* db,
                            int M, int No, int Ni, int accumulate, void* workspace, size_t workspace_bytes,
                            void* stream);                                                /* dW = dy^T x, db = colsum(dy) */
int digat_colsum(const float* x, int64_t ldx, float* out, int M, int N, int accumulate, void* stream);
int digat_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint32_t seed, void* stream);
int digat_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream);
int digat_gate_fwd(const float* z, const float* l, int64_t ldl, const float* g, float* out, int B, int d, void* stream);
int digat_gate_bwd(const float* dout, const float* z, const float* l, int64_t ldl, const float* g,
                   float* dz, float* dl, float* dg, int B, int d, void* stream);
int digat_relu_res_fwd(const float* y, const float* t, float* out, int64_t n, void* stream);   /* relu(y) + t */
int digat_relu_mask(const float* dout, const float* y, float* dy, int64_t n, void* stream);    /* dout * [y > 0] */
int digat_attn_pool_fwd(const float* feat, int64_t ld_b, const float* kq, const uint8_t* mask, float* out,
                        float* alpha_out, int B, int n, int d, void* stream);
int digat_attn_pool_bwd(const float* feat, int64_t ld_b, const float* kq, const uint8_t* mask, const float* alpha,
                        const float* dout, float* dfeat, int64_t ldd_b, float* dkq, int B, int n, int d,
                        int accumulate_dfeat, void* stream);
int digat_topic_pool_fwd_train(const float* Xu, const float* kq, const int64_t* cat_idx, float* out, float* alpha_out,
                               int B, int U, int H, int C1, int d, void* stream);
int digat_topic_pool_bwd(const float* Xu, const float* kq, const int64_t* cat_idx, const float* alpha, const float* dT,
                         float* dXu, float* dkq, int B, int U, int H, int C1, int d, void* stream);
int digat_xattn_project(const float* X, const float* r, const float* W, const float* bW, const float* F1,
                        const float* F2, float* h, float* Pr, float* Q, int B, int n, int d, void* stream);
/* ... with the bf16x6 kernel: [W|ffn1|ffn2] is split into `wsplit` (digat_split_weights_bytes(3 d, d) bytes) first.
 * B*n >= 2048, d % 80 == 0. */
int digat_xattn_project_x3(const float* X, const float* r, const float* W, const float* bW, const float* F1, const float* F2,
                           float* h, float* Pr, float* Q, int B, int n, int d, void* wsplit, void* stream);
int digat_xattn_pairwise_fwd_train(const float* Pr, const float* Q, const float* h, const float* X, const float* a,
                                   const uint8_t* A, float* out, float* alpha, float* s_pre, float* alpha_drop,
                                   uint8_t* amask, float p, uint32_t seed, int B, int n, int d, void* stream);
size_t digat_xattn_pairwise_bwd_workspace(int B, int n, int d);
int digat_xattn_pairwise_bwd(const float* dOut, const float* out, const float* Xres, const float* Pr, const float* Q,
                             const float* h, const float* a, const uint8_t* A, const float* alpha, const float* s_pre,
                             const uint8_t* amask, float p, float* dPr, float* dQ, float* dh, float* da,
                             int accumulate_da, int B, int n, int d, void* workspace, size_t workspace_bytes,
                             void* stream);
int digat_sum_nodes(const float* dP, float* dr, int B, int n, int d, void* stream);          /* dr[b] = sum_j dP[b,j] */

/* BASELINE configs[4], training half.  1: the >= 2048-row GEMMs of the training path (Eq. 8 projections, featureAffine,
 * their input and weight gradients) run with ONE bf16 product per fp32 product — bf16 mixed precision: fp32 master weights and
 * activations, bf16 matrix-core operands, fp32 accumulation; 0 (default): the fp32-grade six-product split.  The [B,d] linears
 * (and their weight gradients) and every reduction stay fp32.  Returns the previous setting. */
int digat_set_train_precision(int bf16);

/* The operand format of the >= 2048-row matrix-core GEMMs (node projections, featureAffine): DIGAT_GEMM_BF16X6 = every fp32
 * value as three bf16 pieces, six products — as accurate as an fp32 fma chain, no range limit; DIGAT_GEMM_F16X3 = two fp16
 * pieces (22-23 significant bits), three products on v_mfma_f32_16x16x32_f16: 0.7x the time.  The weights are scaled by 2^10 and
 * the activations by 2^4 on their way in (exact; undone in the epilogue) so that the low pieces of ordinary values are normal
 * fp16 numbers: against fp64 its mean and largest errors are at or below an fp32 fma chain's (tests/test_hip_lowprec.py) for
 * |w| < 63 and 1e-4 < |x| < 4094; far below the lower bound the pieces are subnormal (an absolute floor of 2^-24 per term);
 * at the upper bound the high piece saturates (precision degrades to ~2^-15, then inf) — the kernel reports that through
 * digat_params.range_flag.
 * There is NO process-wide setting: the format is a property of a split image.  It is chosen by whoever splits the weights
 * (the `format` argument of digat_split_*), the encoder entry points are told through digat_params.flags & DIGAT_PARAMS_GEMM_F16X3 (bit 6 = 64; NOT the enum value DIGAT_GEMM_F16X3 = 1)
 * (every wsplit image of one digat_params has the same format), and the library remembers the format of every image it has
 * split: a launch that names the other format returns DIGAT_ERR_ARG instead of misreading the image.  The training entries
 * (digat_*_fwd_train, digat_*_bwd, digat_linear_bwd_input_x3, digat_xattn_project_x3) and the MSA encoder split their weights in
 * DIGAT_GEMM_BF16X6: gradients have no lower bound (1e-6 and below is ordinary) and fp16 pieces of such values are subnormal or
 * zero; bf16 pieces keep fp32's exponent range.  Two host threads on two streams may therefore run an fp16x3 evaluation and a
 * training step at the same time. */
enum { DIGAT_GEMM_BF16X6 = 0, DIGAT_GEMM_F16X3 = 1 };

/* The step's head and tail around the encoder (round 6): the backward of digat_row_logits (model.py:75: logits = sum_d user_ctx news_ctx)
 * and the loss of trainer.py:100 — mean over the B impressions of -log_softmax(logits [B,K])[:, 0] — with its gradient d loss / d logits
 * [B,K] (written by the same launch: the caller scales it by the loss's incoming gradient). */
int digat_row_logits_bwd(const float* dlogits, const float* news_ctx, const float* user_ctx, float* dnews, float* duser, int B, int d,
                         void* stream);
int digat_click_loss(const float* logits, int B, int K, float* loss, float* dlogits, void* stream);

/* The optimiser's step (trainer.py:30, :103-105: clip_grad_norm_ + Adam) as three launches: the squared gradient norm per chunk, its
 * total, and one pass over (p, g, m, v) that applies the clipping coefficient min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no
 * clipping) and torch.optim.Adam's update (coupled weight decay per tensor; bias corrections from `step` >= 1).  `tensors`: an array
 * of digat_opt_tensor in DEVICE memory; chunk c covers the digat_opt_chunk() elements of tensor chunk_tensor[c] from chunk_off[c];
 * scratch: nchunks + 1 floats.  The gradients are read, not rewritten. */
typedef struct digat_opt_tensor { float* p; const float* g; float* m; float* v; int64_t n; float weight_decay; int32_t pad; } digat_opt_tensor;
int digat_opt_chunk(void);
int digat_clip_adam_step(const void* tensors, const int32_t* chunk_tensor, const int64_t* chunk_off, int nchunks, float* scratch, float max_norm,
                         float lr, float beta1, float beta2, float eps, int step, void* stream);

/* ---- training: the three functions of the path as one forward and one backward call each ----------------
 * Composed on the C++ side from the primitives above (digat_train_abi.inc); digat_amd/training.py wraps each pair in one
 * autograd.Function.  `save` is a caller-owned buffer carrying what the backward needs from the forward (private layout,
 * *_save_bytes); `workspace` is scratch (*_workspace_bytes, the same size for both directions).  Dropout: the library's
 * counter-hash generator with the caller's seed; p = 0 disables it.  Gradients are WRITTEN (not accumulated) and
 * bit-reproducible (ordered reductions).
 *
 * a1 / a2 (graphEncoders.py:143-154, :163-174).  X is the layer input AFTER its input dropout drop_{p/2} (the residual uses
 * the dropped input); out = relu(drop_p(alpha) h) + X.  The backward recomputes relu'(K3+K1+K2) from the saved projections. */
/* Split images made ahead (round 6).  The matrix-core products of these entries read their weights as bf16x6 images; by default
 * every call splits its own (the weights change every optimiser step): 20 split launches per 64 x 5-row step.  digat_split_jobs
 * writes any number of images (<= 24) in ONE launch — once per step, after the optimiser's update — and the entries take the
 * image of THEIR weights in THEIR layout through the *_image arguments (NULL: split inside, as before).  An image is only a
 * function of the weight values: the caller keeps it valid (same weights, not freed) until the calls that read it have been
 * enqueued on the same stream.
 *   layout 0: y = x [w0 | w1 | w2]^T, the forward product of nn.Linear weights [rows, cols] (w1 = w2 = NULL: one matrix);
 *   layout 1: dx = dy w0, the input gradient ([dy0 | dy1 | dy2] [w0; w1; w2] with three matrices). */
typedef struct digat_split_job {
    const float *w0, *w1, *w2;   /* [rows, cols] each; w1, w2 both NULL or both given */
    int32_t rows, cols, layout, reserved;
    void* image;                 /* digat_split_job_bytes(rows, cols, layout, 1 or 3) bytes */
} digat_split_job;
size_t digat_split_job_bytes(int rows, int cols, int layout, int matrices);
int digat_split_jobs(const digat_split_job* jobs, int njobs, void* stream);

size_t digat_xattn_train_save_bytes(int B, int n, int d);
size_t digat_xattn_train_workspace_bytes(int B, int n, int d);
/* proj_image: NULL, or the layout-0 image of (W, F1, F2); bwd_image: NULL, or their layout-1 image.
 * p_in > 0 (round 6): X is the layer input BEFORE its input dropout; the entry applies drop_{p_in} with seed_in itself (the dropped
 * input and its keep bytes travel in `save`), out = relu(drop_p(alpha) h) + drop(X), and the backward's dX is the gradient of the
 * UNDROPPED X — the keep bytes are applied in the epilogue of the input-gradient product, no dropout launch.  p_in = 0: X is used
 * as given (callers that drop their input themselves).  The backward takes the same p_in.
 * xattn_mode (graphs of more than 16 nodes; both directions take the same value): which kernels evaluate Eq. 8 — a choice of speed
 * only, the function is the same.  DIGAT_TRAIN_XATTN_AUTO: decided on the device per batch from a sample of the adjacency (the
 * entry-wise kernels when it holds at most ~20 entries per node, the all-pairs kernels otherwise; the launches of the side not
 * taken return at once); _SPARSE: the entry-wise kernels (a wave per centre / per node over the adjacency's entries), _DENSE: the
 * all-pairs kernels — unguarded: a caller that knows its corpus saves the decision and the empty launches. */
enum { DIGAT_TRAIN_XATTN_AUTO = 0, DIGAT_TRAIN_XATTN_SPARSE = 1, DIGAT_TRAIN_XATTN_DENSE = 2 };
int digat_xattn_fwd_train(const float* X, const uint8_t* A, const float* ctx, const float* W, const float* bW, const float* F1,
                          const float* F2, const float* F3, const float* b3, const float* a, float* out, float p_alpha,
                          uint32_t seed, float p_in, uint32_t seed_in, int B, int n, int d, void* save, size_t save_bytes, void* workspace,
                          size_t workspace_bytes, const void* proj_image, int xattn_mode, void* stream);
int digat_xattn_bwd(const float* dOut, const float* out, const float* X, const uint8_t* A, const float* ctx, const float* W,
                    const float* F1, const float* F2, const float* F3, const float* a, float p_alpha, float p_in, const void* save,
                    size_t save_bytes, float* dX, float* dctx, float* dW, float* dbW, float* dF1, float* dF2, float* dF3,
                    float* db3, float* da, int B, int n, int d, void* workspace, size_t workspace_bytes, const void* bwd_image,
                    int xattn_mode, void* stream);
/* a3 (graphEncoders.py:109-114): out = gate(drop_{p_gate}(W_g [l ; g] + b_g), l, g), l = X[:,0], g = candidate_attention(X, l). */
size_t digat_news_ctx_train_save_bytes(int B, int N, int d);
size_t digat_news_ctx_train_workspace_bytes(int B, int N, int d);
int digat_news_ctx_fwd_train(const float* X, const uint8_t* mask, const float* Kc, const float* Qc, const float* bQc,
                             const float* Wg, const float* bg, float* out, float p_gate, uint32_t seed, int B, int N, int d,
                             void* save, size_t save_bytes, void* workspace, size_t workspace_bytes,
                             const float* prev /* NULL, or [B,d]: out = prev + context (graphEncoders.py:185; d prev = dout) */, void* stream);
/* accumulate_params != 0: the PARAMETER gradients (dKc, dQc, dbQc, dWg, dbg) are added to what their buffers hold instead of
 * overwriting it — the context functions' weights are shared by the depth + 1 calls of a step (graphEncoders.py:177-187: one
 * candidate_attention / news_graph_W for every layer), so a caller can sum a step's gradients in one set of buffers inside the
 * weight-gradient launches themselves (training.py's StepSink) instead of one element-wise add per weight and call.  dX is always
 * written. */
int digat_news_ctx_bwd(const float* dout, const float* X, const uint8_t* mask, const float* Kc, const float* Qc, const float* Wg,
                       float p_gate, const void* save, size_t save_bytes, float* dX, float* dKc, float* dQc, float* dbQc,
                       float* dWg, float* dbg, int B, int N, int d, int accumulate_params, void* workspace, size_t workspace_bytes,
                       void* stream);
/* a4 (+a6, a7) (graphEncoders.py:123-134): topic pooling (scatter_softmax + scatter_sum), featureAffine + relu + residual,
 * drop_{p_topic}, userAttention.  dXu [B,U,d]: history rows get the pooling's gradient, topic rows zero. */
size_t digat_user_ctx_train_save_bytes(int B, int U, int H, int C1, int d);
size_t digat_user_ctx_train_workspace_bytes(int B, int U, int H, int C1, int d);
int digat_user_ctx_fwd_train(const float* Xu, const uint8_t* cat_mask, const int64_t* cat_idx, const float* c_n, const float* Ku,
                             const float* Qu, const float* bQu, const float* Fa, const float* bFa, const float* Kua,
                             const float* Qua, const float* bQua, float* out, float p_topic, uint32_t seed, int B, int U, int H,
                             int C1, int d, void* save, size_t save_bytes, void* workspace, size_t workspace_bytes,
                             const void* fa_image /* NULL, or the layout-0 image of Fa */,
                             const float* prev /* NULL, or [B,d]: out = prev + context (graphEncoders.py:186) */, void* stream);
int digat_user_ctx_bwd(const float* dout, const float* Xu, const uint8_t* cat_mask, const int64_t* cat_idx, const float* c_n,
                       const float* Ku, const float* Qu, const float* Fa, const float* Kua, const float* Qua, float p_topic,
                       const void* save, size_t save_bytes, float* dXu, float* dc_n, float* dKu, float* dQu, float* dbQu,
                       float* dFa, float* dbFa, float* dKua, float* dQua, float* dbQua, int B, int U, int H, int C1, int d,
                       int accumulate_params /* as digat_news_ctx_bwd: the eight parameter gradients; dXu, dc_n are always written */,
                       void* workspace, size_t workspace_bytes, const void* fa_bwd_image /* NULL, or the layout-1 image of Fa */, void* stream);

/* ---- H2: ranking + metrics of the dev/test driver  (util.py:70-80, evaluate.py:32-89) -------------------
 * scores [R] f32 in impression-major row order; impression_start [I+1] int64 (row offsets; impression i owns
 * rows [start[i], start[i+1])).  ranks [R] int32 receives the 1-based rank of every candidate after a STABLE
 * descending sort of its impression's scores (what util.py:73-79 writes to the rank file).  With labels [R]
 * (bytes, non-zero = clicked) and per_impression [I,4] f64 it also writes AUC, MRR, nDCG@5, nDCG@10 of every
 * impression on 1/rank scores (evaluate.py:45-60), and with mean4 [4] their means over the I impressions
 * (evaluate.py:85-89).  labels / per_impression / mean4 may be NULL (ranks only). */
int digat_rank_metrics(const float* scores, const uint8_t* labels, const int64_t* impression_start, int num_impressions,
                       int32_t* ranks, double* per_impression, double* mean4, void* stream);

/* The rank file of util.py:74-84 as bytes, formatted on the HOST (no GPU work): "<impression id> [r1,r2,...]" per line for ids
 * 1 .. impressions ("[]" for an id without rows), lines joined by '\n', no trailing newline.  ranks [R] int64 (1-based ranks in
 * row order), starts [impressions + 1] int64 row ranges.  Returns the number of bytes written, or — when out is NULL or cap is too
 * small — the capacity needed. */
int64_t digat_format_rank_file(const int64_t* ranks, const int64_t* starts, int64_t impressions, char* out, int64_t cap);

/* ---- vanilla-GAT update layer of the ablation encoders -----------------------------------------
 * graphEncoders.py:493-519 (wo_interaction), :641-651 (News_graph_wo_inter), :788-798 (User_graph_wo_inter), eval mode:
 * h = X W^T + bW; e_ij = leaky_relu_0.2(a1.h_j + a2.h_i); -1e9 where A_ij = 0; alpha = softmax_j; out = relu(alpha h) + X.
 * X, out [B,n,d]; A [B,n,n] bytes; W [d,d], bW [d] or NULL, a1, a2 [d]. */
size_t digat_gat_workspace_bytes(int B, int n, int d);
int digat_gat_fwd(const float* X, const uint8_t* A, const float* W, const float* bW, const float* a1, const float* a2,
                  float* out, int B, int n, int d, void* workspace, size_t workspace_bytes, void* stream);
/* The same layer in training mode (dropout live: graphEncoders.py:495 on the input is the caller's, :500 on alpha is
 * p_alpha here), one call per direction like digat_xattn_fwd_train / digat_xattn_bwd: `save` carries h, alpha, the scores
 * before leaky_relu and the dropout mask from the forward to the backward.  dX, dW, dbW, da1, da2 are written, not
 * accumulated; sums run in index order (bit-reproducible). */
size_t digat_gat_train_save_bytes(int B, int n, int d);
size_t digat_gat_train_workspace_bytes(int B, int n, int d);
int digat_gat_fwd_train(const float* X, const uint8_t* A, const float* W, const float* bW, const float* a1, const float* a2, float* out,
                        float p_alpha, uint32_t seed, int B, int n, int d, void* save, size_t save_bytes, void* workspace,
                        size_t workspace_bytes, void* stream);
int digat_gat_bwd(const float* dOut, const float* out, const float* X, const uint8_t* A, const float* W, const float* a1, const float* a2,
                  float p_alpha, const void* save, size_t save_bytes, float* dX, float* dW, float* dbW, float* da1, float* da2,
                  int B, int n, int d, void* workspace, size_t workspace_bytes, void* stream);

/* ---- semantic-augmented-graph construction: construct_SAG.py ------------------------------------
 * generate_cos_similarities (construct_SAG.py:112-162), one category: title/content [n, dim], corpus_title/corpus_content
 * [m, dim] fp32 sentence embeddings; k = min(top_M, m - 1) + 1 <= 32, dim % 16 == 0.  values / indices [5, n, k]
 * (fp32 / int32), kinds in the reference's return order: title-title, content-content, title-content (title query vs
 * corpus contents), content-title, and the mean of the four; every row sorted descending as torch.topk returns it
 * (equal values: lower corpus index first). */
size_t digat_sag_cos_topk_workspace_bytes(int64_t n, int64_t m, int dim);
int digat_sag_cos_topk(const float* title, const float* content, int64_t n, const float* corpus_title, const float* corpus_content,
                       int64_t m, int dim, int k, float* values, int32_t* indices, void* workspace, size_t workspace_bytes,
                       void* stream);
/* generate_news_graph (construct_SAG.py:449-485): sim_index / sim_cos [news_num, top_M] = every news's similar-news list
 * (news indices, cosines; sim_len [news_num] entries are valid), hop, news_node_num <= 256, threshold = the reference's
 * similarity_threshold (0.5, :10).  Outputs (all rewritten): news_node_ID [news_num, nn] int32, news_graph
 * [news_num, nn, nn] bytes, news_graph_mask [news_num, nn] bytes.  *overflow (device int32) is set to 1 when a walk needs
 * more than news_node_num nodes (the reference raises IndexError there). */
int digat_sag_news_graph(const int32_t* sim_index, const float* sim_cos, const int32_t* sim_len, int64_t news_num, int top_M, int hop,
                         int news_node_num, float threshold, int32_t* news_node_ID, uint8_t* news_graph, uint8_t* news_graph_mask,
                         int32_t* overflow, void* stream);

/* ---- news encoder (upstream of the path): newsEncoders.MSA.forward in eval mode ------------------
 * newsEncoders.py:70-82 with layers.MultiHeadAttention (layers.py:50-88) and layers.Attention (:91-115).
 * title_text [T, Lw] int32 token ids (rows of word_embedding), title_mask [T, Lw] bytes (0 = padding: masked only in
 * the pooling, as in the reference), out [T, head_num * head_dim].  Lw <= 64, head_dim <= 32.
 * qkv_wsplit: [W_Q|W_K|W_V] split by digat_split_proj_weights-style call digat_split_msa_weights (optional: NULL =
 * fp32 MFMA path with a materialised embedding gather). */
typedef struct digat_msa_params {
    int32_t word_embedding_dim, head_num, head_dim, attention_dim;
    const float *word_embedding;                 /* word_embedding.weight [V, word_embedding_dim]            */
    const float *W_Q, *b_Q, *W_K, *W_V, *b_V;    /* multiheadSelfattention.W_{Q,K,V}.{weight,bias}            */
    const float *A1, *b1, *a2;                   /* attention.affine1.{weight,bias}, attention.affine2.weight */
    const void  *qkv_wsplit;                     /* digat_split_msa_weights output, or NULL                   */
    const void  *a1_wsplit;                      /* digat_split_weights(affine1.weight, attention_dim, hd) output, or NULL */
} digat_msa_params;
size_t digat_msa_split_bytes(int word_embedding_dim, int head_num, int head_dim);
int digat_split_msa_weights(const float* W_Q, const float* W_K, const float* W_V, int word_embedding_dim, int hd,
                            void* wsplit, void* stream);
size_t digat_msa_workspace_bytes(int T, int Lw, int word_embedding_dim, int head_num, int head_dim, int attention_dim);
int digat_msa_fwd(const digat_msa_params* params, const int32_t* title_text, const uint8_t* title_mask, float* out,
                  int T, int Lw, void* workspace, size_t workspace_bytes, void* stream);
/* The same encoder in training mode (newsEncoders.py:70-82 with dropout live), one call per direction.  p_drop is the dropout
 * on the embedded tokens (:77).  `save` carries the dropped embeddings, Q|K|V, h, the affine1 product and the pooling weights to the
 * backward; the attention scores are recomputed there.  The *_wsplit fields of params are ignored (the weights change every
 * optimiser step: they are split inside, into the workspace).  Lw <= 32, attention_dim % 4 == 0.
 * digat_msa_bwd writes (does not accumulate) row_grad [T*Lw, word_embedding_dim], rows ld_row_grad floats apart
 * (word_embedding_dim, or digat_msa_row_grad_ld(): padded to a multiple of 80 so that one matrix-core product computes it) — the
 * gradient at the embedding rows the tokens looked up — and dW_Q dW_K dW_V [hd, word_embedding_dim], db_Q db_V [hd], dA1 [attention_dim, hd], db1 da2 [attention_dim].
 * digat_embedding_bwd sums row_grad per token into table_grad [V, word_embedding_dim] (zero-filled by the caller) in a fixed
 * order — bit-reproducible, no atomics: `order` [M] = the row indices stably sorted by token, sorted_tokens [M] the tokens in that order. */
size_t digat_msa_train_save_bytes(int T, int Lw, int word_embedding_dim, int head_num, int head_dim, int attention_dim);
size_t digat_msa_train_workspace_bytes(int T, int Lw, int word_embedding_dim, int head_num, int head_dim, int attention_dim);
int digat_msa_fwd_train(const digat_msa_params* params, const int32_t* title_text, const uint8_t* title_mask, float* out, float p_drop,
                        uint32_t seed, int T, int Lw, void* save, size_t save_bytes, void* workspace, size_t workspace_bytes,
                        void* stream);
int digat_msa_bwd(const digat_msa_params* params, const int32_t* title_text, const uint8_t* title_mask, const float* dout, float p_drop,
                  const void* save, size_t save_bytes, float* row_grad, int64_t ld_row_grad, float* dW_Q, float* db_Q, float* dW_K,
                  float* dW_V, float* db_V, float* dA1, float* db1, float* da2, int T, int Lw, void* workspace, size_t workspace_bytes,
                  void* stream);
int64_t digat_msa_row_grad_ld(int T, int Lw, int word_embedding_dim);
size_t digat_embedding_bwd_workspace_bytes(int64_t M, int dm);
int digat_embedding_bwd(const float* row_grad, int64_t ld_row_grad, const int32_t* order, const int32_t* sorted_tokens, int64_t M, int dm,
                        float* table_grad, void* workspace, size_t workspace_bytes, void* stream);
/* The same sum for a few thousand looked-up rows without a sorted order (the backward of torch.nn.functional.embedding on a wide
 * table, trainer.py:98-102 through the news encoder's table): ids0 [M0] / ids1 [M1] int64 (two lookups of ONE table; ids1 may be NULL
 * with M1 = 0), their row gradients g0 / g1 (row strides ld0 / ld1); table_grad [V, dm] zero-filled by the caller; an id outside
 * [0, V) receives nothing.  Two launches (and a memset of 8 V bytes): per chunk of 64 positions the first position of an id adds the
 * chunk's rows of that id in ascending order, then the first position overall adds those partial rows in ascending chunk order —
 * bit-reproducible (the only atomics are an integer min and an integer count per id).
 * dm % 4 == 0, dm <= 1024, M0 + M1 <= 2^24 (the id scans are quadratic in M: meant for M of a few thousand). */
size_t digat_embedding_bwd_unsorted_workspace_bytes(int64_t M, int dm, int64_t V);
int digat_embedding_bwd_unsorted(const int64_t* ids0, const float* g0, int64_t ld0, int64_t M0, const int64_t* ids1, const float* g1,
                                 int64_t ld1, int64_t M1, int dm, int64_t V, float* table_grad, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* ---- runs of identical consecutive user rows (drop-in path) -------------------------------------------------------------------
 * The reference's driver hands Model.inference the user tensors EXPANDED per row — an impression's history embeddings, user graph,
 * category mask and indices repeated once per candidate (util.py:57-67, model.py:87-90) — so consecutive rows are bit-identical
 * runs.  This finds them on the device: row b belongs to the run of row b - 1 iff EVERY byte of its four user tensors is equal
 * (ue [B,H,d] f32, Au [B,U,U] u8, cat_mask [B,C1] u8, cat_idx [B,H] i64).  row_group[b] = index of row b's run (ascending, runs are
 * consecutive: the layout digat_encoder_fwd_grouped's layer 0 exploits), leaders[g] = first row of run g (int64), *n_runs = G.
 * workspace: B bytes.  DIGAT.inference reads *n_runs (one host synchronisation), gathers the leaders' user tensors and calls the
 * grouped entry when 4 G <= B: bit-identical to digat_encoder_fwd on the expanded rows. */
int digat_user_row_runs(const float* ue, const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx, int B, int H, int U, int C1, int d,
                        int32_t* row_group, int64_t* leaders, int32_t* n_runs, void* workspace, size_t workspace_bytes, void* stream);

/* digat_encoder_fwd with the search for shared users INSIDE (DIGAT.inference's default; round 5): the same per-row arguments as
 * digat_encoder_fwd (graphEncoders.py:189-198) — the reference's driver expands an impression's user tensors once per candidate
 * (util.py:57-67) — and layer 0 of the user graph computed once per RUN of identical consecutive rows: the runs are found on the
 * device (every byte of the four user tensors compared with the previous row's), every count stays there (no host read), the
 * group-level data of a run lives in its leading row's slots of the full-size buffers.  Bit-identical to digat_encoder_fwd; rows
 * that share nothing cost the comparison pass (B x 85 KB read once) on top of it.  Needs the folded inference path and the sparse
 * Eq. 8 variant of the user graph (flags), else it IS digat_encoder_fwd.  c_n0 may be NULL (forward). */
size_t digat_encoder_shared_workspace_bytes(int B, int N, int H, int C, int d, int depth);
int digat_encoder_fwd_shared(const digat_params* p, const float* Xn_in, const uint8_t* An, const uint8_t* Mn, const float* ue,
                             const uint8_t* Au, const uint8_t* cat_mask, const int64_t* cat_idx, const float* c_n0, float* out_news,
                             float* out_user, int B, int N, int H, void* workspace, size_t workspace_bytes, void* stream);

/* ---- batch assembly (the device-side counterpart of MIND_dataset.py's __getitem__ + collate) ------------------
 * All table gathers of one scoring batch in one launch.  Job k copies `rows` rows of `row_bytes` bytes: dst row r = src row
 * idx[r] (idx2 == NULL, inner = 1) or src row idx2[idx[r / inner] * inner + r % inner] (two-level: e.g. the representation of
 * the r % H-th history item of impression idx[r / H], idx2 = the [impressions, H] history table).  `jobs` is a HOST array
 * (njobs <= 24), copied into the launch; every pointer inside is a device pointer. */
typedef struct digat_gather_job {
    const void* src; void* dst;
    int64_t row_bytes, rows;
    const int64_t* idx; const int64_t* idx2;
    int64_t inner;
} digat_gather_job;
int digat_gather_tables(const digat_gather_job* jobs, int njobs, void* stream);

/* ---- measurement aid (not on the reference's surface): per-kernel HIP-event timing ---------------
 * Between start and stop every kernel launch of this library is bracketed by two events recorded
 * on the stream it is launched on.  stop() synchronises and returns, per kernel kind, the summed
 * duration [ms], the summed algorithmic work (flops for LINEAR/PROJ, bytes for the others) and the
 * launch count; the three arrays have DIGAT_KERNEL_KINDS entries (any may be NULL). */
enum {
    DIGAT_KERNEL_PROJ = 0,    /* [h|P|Q] = X [W|ffn1|ffn2]^T, the Eq. 8 node projections (MFMA) */
    DIGAT_KERNEL_LINEAR = 1,  /* every other nn.Linear (MFMA)                                    */
    DIGAT_KERNEL_XATTN = 2,   /* fused Eq. 8 score + mask + softmax -> alpha                     */
    DIGAT_KERNEL_POOL = 3,    /* ScaledDotProductAttention pooling                               */
    DIGAT_KERNEL_TOPIC = 4,   /* scatter_softmax + scatter_sum topic pooling                     */
    DIGAT_KERNEL_GLUE = 5,    /* user-node concat                                                */
    DIGAT_KERNEL_AGG = 6,     /* relu(alpha @ h) + X on the matrix cores                         */
    DIGAT_KERNEL_KINDS = 7
};
int digat_profile_start(int max_launches);
/* launches a one-thread kernel (digat_region_marker_kernel) on `stream`: a landmark in a rocprofv3 kernel trace */
int digat_profile_marker(int id, void* stream);
/* bit k set = launches of kind k are recorded (default: all).  Returns the previous mask. */
int digat_profile_set_kinds(unsigned mask);
int digat_profile_pause(int paused);   /* between start and stop: 1 = launches are not recorded, 0 = recorded again (sampling) */
/* after digat_profile_stop: rows processed / rows nominal over the row-list node-projection launches of the profiled
 * region (the encoder leaves the user-graph nodes that cannot reach its outputs out of the projections of
 * layers >= 1), or -1 if there was none */
double digat_profile_live_row_fraction(void);
int digat_profile_stop(double* ms_per_kind, double* work_per_kind, int* launches_per_kind);
/* After digat_profile_stop: the compulsory HBM bytes (operand rows in, result rows out, epilogue row operands; weights not
 * counted) of the recorded launches of the two MFMA kinds, DIGAT_KERNEL_PROJ and DIGAT_KERNEL_LINEAR, whose `work` is flops —
 * for a whole-step roofline (bench.py: roofline_step).  DIGAT_KERNEL_KINDS entries; the byte-priced kinds read 0 here. */
int digat_profile_gemm_bytes(double* bytes_per_kind);
/* After digat_profile_stop: the DIGAT_KERNEL_XATTN launches by kernel, DIGAT_XATTN_PARTS entries each (any may be NULL):
 * [0] user graph, layers >= 1 (row-list launches: xattn_sparse_twin_kernel), [1] user graph, layer 0 of grouped rows
 * (xattn_sparse_l0_kernel), [2] news graphs of <= 16 nodes with the graph in LDS (xattn_small_lds_kernel), [3] every other
 * Eq. 8 launch.  ms = summed launch durations, bytes = algorithmic HBM bytes (SURVEY 8d's bytes_B on the live rows: each
 * distinct row once), launches = count.  Replaces nothing in the reference: measurement only (bench.py roofline_xattn.parts). */
#define DIGAT_XATTN_PARTS 4
int digat_profile_xattn_parts(double* ms, double* bytes, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* DIGAT_HIP_H */
