// digat_torch_ext — the thin torch extension over the C ABI of include/digat_hip.h (BASELINE north_star: "kernels bound through a thin
// C-ABI torch extension").
//
// Nothing is computed here.  Each function takes the at::Tensors of one call of the reference's plugin surface (graphEncoders.py:177-198,
// model.py:73,88), checks what the C ABI assumes about them — same CUDA device, dtype, contiguity, shapes — takes PyTorch's CURRENT
// HIP stream, and forwards raw device pointers to libdigat_hip.so.  The parameter block (digat_params: pointers into the module's
// own nn.Parameters and the split weight images) is built once per weight version on the Python side and arrives as an address.
// digat_amd/_lib.py loads this module when it has been built (digat_amd/build.py) and binds the same entry points through ctypes
// otherwise: both roads end in the same shared object, there is no CPU path on either.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/digat_hip.h"

namespace {

void need(bool ok, const char* what) { TORCH_CHECK(ok, "digat_torch_ext: ", what); }

const at::Tensor& on_gpu(const at::Tensor& t, const at::Tensor& like, const char* name) {
    TORCH_CHECK(t.is_cuda(), "digat_torch_ext: ", name, " must be a GPU tensor (there is no CPU path)");
    TORCH_CHECK(t.device() == like.device(), "digat_torch_ext: ", name, " lives on another device");
    TORCH_CHECK(t.is_contiguous(), "digat_torch_ext: ", name, " must be contiguous");
    return t;
}
const float* f32(const at::Tensor& t, const at::Tensor& like, const char* name) {
    on_gpu(t, like, name);
    TORCH_CHECK(t.scalar_type() == at::kFloat, "digat_torch_ext: ", name, " must be float32");
    return t.data_ptr<float>();
}
const uint8_t* bytes(const at::Tensor& t, const at::Tensor& like, const char* name) {     // torch.bool / torch.uint8: one byte per element
    on_gpu(t, like, name);
    TORCH_CHECK(t.scalar_type() == at::kBool || t.scalar_type() == at::kByte, "digat_torch_ext: ", name, " must be bool or uint8");
    return static_cast<const uint8_t*>(t.data_ptr());
}
const int64_t* i64(const at::Tensor& t, const at::Tensor& like, const char* name) {
    on_gpu(t, like, name);
    TORCH_CHECK(t.scalar_type() == at::kLong, "digat_torch_ext: ", name, " must be int64");
    return t.data_ptr<int64_t>();
}
const int32_t* i32(const at::Tensor& t, const at::Tensor& like, const char* name) {
    on_gpu(t, like, name);
    TORCH_CHECK(t.scalar_type() == at::kInt, "digat_torch_ext: ", name, " must be int32");
    return t.data_ptr<int32_t>();
}
void* stream_of(const at::Tensor& t) {
    TORCH_CHECK(t.is_cuda(), "digat_torch_ext: GPU tensors only (there is no CPU path)");
    return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}
// shapes the raw-pointer kernels assume (round 6, ADVICE r05: a wrong-shaped tensor used to reach them unchecked)
void shape(const at::Tensor& t, std::initializer_list<int64_t> want, const char* name) {
    bool ok = t.dim() == (int64_t)want.size();
    int64_t k = 0;
    if (ok) for (int64_t w : want) { ok = ok && t.size(k) == w; ++k; }
    TORCH_CHECK(ok, "digat_torch_ext: ", name, " has shape ", t.sizes(), ", expected ", at::IntArrayRef(want.begin(), want.size()));
}
void check(int rc, const char* what) {
    TORCH_CHECK(rc == DIGAT_OK, "digat_torch_ext: ", what, " failed: [", rc, "] ", digat_error_string(rc));
}
const digat_params* params_at(int64_t addr) {
    need(addr != 0, "null parameter block");
    return reinterpret_cast<const digat_params*>(static_cast<uintptr_t>(addr));
}

// graphEncoders.DIGAT.forward (c_n0 undefined) / .inference (c_n0 given): digat_encoder_fwd, or — shared = true — digat_encoder_fwd_shared
// (the same arguments; runs of identical consecutive user rows are found on the device and layer 0 is computed once per run)
void encoder_fwd(int64_t params, bool shared, const at::Tensor& Xn, const at::Tensor& An, const at::Tensor& Mn, const at::Tensor& ue, const at::Tensor& Au,
                 const at::Tensor& cm, const at::Tensor& ci, const c10::optional<at::Tensor>& c_n0, at::Tensor& out_n, at::Tensor& out_u,
                 at::Tensor& ws) {
    need(Xn.dim() == 3 && ue.dim() == 3, "news_graph_embeddings [B,N,d] and user_news_embedding [B,H,d] expected");
    const int B = (int)Xn.size(0), N = (int)Xn.size(1), H = (int)ue.size(1);
    need(ue.size(0) == B && An.size(0) == B && Mn.size(0) == B && Au.size(0) == B && cm.size(0) == B && ci.size(0) == B, "batch sizes differ");
    need(out_n.size(0) == B && out_u.size(0) == B && out_n.size(1) == Xn.size(2) && out_u.size(1) == Xn.size(2), "outputs must be [B,d]");
    const auto fn = shared ? digat_encoder_fwd_shared : digat_encoder_fwd;
    check(fn(params_at(params), f32(Xn, Xn, "news_graph_embeddings"), bytes(An, Xn, "news_graph"), bytes(Mn, Xn, "news_graph_mask"),
             f32(ue, Xn, "user_news_embedding"), bytes(Au, Xn, "user_graph"), bytes(cm, Xn, "user_category_mask"),
             i64(ci, Xn, "user_category_indices"), c_n0.has_value() ? f32(*c_n0, Xn, "news_graph_context") : nullptr,
             const_cast<float*>(f32(out_n, Xn, "out_news")), const_cast<float*>(f32(out_u, Xn, "out_user")), B, N, H,
             on_gpu(ws, Xn, "workspace").data_ptr(), (size_t)ws.nbytes(), stream_of(Xn)),
          shared ? "digat_encoder_fwd_shared" : "digat_encoder_fwd");
}

// DIGAT.inference_grouped: digat_encoder_fwd_grouped / _grouped_cached (the per-news tables are optional, any subset)
void encoder_fwd_grouped(int64_t params, const at::Tensor& Xn, const at::Tensor& An, const at::Tensor& Mn, const at::Tensor& ue_g,
                         const at::Tensor& Au_g, const at::Tensor& cm_g, const at::Tensor& ci_g, const at::Tensor& row_group,
                         const at::Tensor& c_n0, const c10::optional<at::Tensor>& news_hpq0, const c10::optional<at::Tensor>& hist_hpq0,
                         const c10::optional<at::Tensor>& topic_hpq0, const c10::optional<at::Tensor>& ctxq0,
                         const c10::optional<at::Tensor>& news_index, at::Tensor& out_n, at::Tensor& out_u, at::Tensor& ws) {
    need(Xn.dim() == 3 && ue_g.dim() == 3 && An.dim() == 3, "news_graph_embeddings [.,N,d], news_graph [B,N,N], user_news_embedding [G,H,d] expected");
    const int B = (int)An.size(0), N = (int)Xn.size(1), G = (int)ue_g.size(0), H = (int)ue_g.size(1);
    need(row_group.numel() == B && Mn.size(0) == B && c_n0.size(0) == B, "row_group, news_graph_mask and news_graph_context are per row");
    need(Au_g.size(0) == G && cm_g.size(0) == G && ci_g.size(0) == G, "the user tensors are per group");
    need(news_index.has_value() || Xn.size(0) == B, "news_graph_embeddings: one graph per row, or the per-news table with news_index");
    const bool cached = news_hpq0.has_value() || hist_hpq0.has_value() || ctxq0.has_value() || news_index.has_value();
    const float* xn = f32(Xn, Xn, "news_graph_embeddings");
    float* on = const_cast<float*>(f32(out_n, Xn, "out_news"));
    float* ou = const_cast<float*>(f32(out_u, Xn, "out_user"));
    void* wsp = on_gpu(ws, Xn, "workspace").data_ptr();
    if (!cached) {
        check(digat_encoder_fwd_grouped(params_at(params), xn, bytes(An, Xn, "news_graph"), bytes(Mn, Xn, "news_graph_mask"),
                                        f32(ue_g, Xn, "user_news_embedding"), bytes(Au_g, Xn, "user_graph"), bytes(cm_g, Xn, "user_category_mask"),
                                        i64(ci_g, Xn, "user_category_indices"), i32(row_group, Xn, "row_group"), f32(c_n0, Xn, "news_graph_context"),
                                        on, ou, B, G, N, H, wsp, (size_t)ws.nbytes(), stream_of(Xn)),
              "digat_encoder_fwd_grouped");
        return;
    }
    check(digat_encoder_fwd_grouped_cached(params_at(params), xn, bytes(An, Xn, "news_graph"), bytes(Mn, Xn, "news_graph_mask"),
                                           f32(ue_g, Xn, "user_news_embedding"), bytes(Au_g, Xn, "user_graph"), bytes(cm_g, Xn, "user_category_mask"),
                                           i64(ci_g, Xn, "user_category_indices"), i32(row_group, Xn, "row_group"), f32(c_n0, Xn, "news_graph_context"),
                                           news_hpq0.has_value() ? f32(*news_hpq0, Xn, "news_hpq0") : nullptr,
                                           hist_hpq0.has_value() ? f32(*hist_hpq0, Xn, "hist_hpq0") : nullptr,
                                           topic_hpq0.has_value() ? f32(*topic_hpq0, Xn, "topic_hpq0") : nullptr,
                                           ctxq0.has_value() ? f32(*ctxq0, Xn, "ctxq0") : nullptr,
                                           news_index.has_value() ? i64(*news_index, Xn, "news_index") : nullptr,
                                           news_index.has_value() ? (int64_t)Xn.size(0) : 0, on, ou, B, G, N, H, wsp, (size_t)ws.nbytes(),
                                           stream_of(Xn)),
          "digat_encoder_fwd_grouped_cached");
}

// model.py:75,90: logits = sum_d(user_ctx * news_ctx)
void row_logits(const at::Tensor& news_ctx, const at::Tensor& user_ctx, at::Tensor& logits) {
    need(news_ctx.dim() == 2 && user_ctx.sizes() == news_ctx.sizes() && logits.numel() == news_ctx.size(0), "contexts [B,d], logits [B]");
    check(digat_row_logits(f32(news_ctx, news_ctx, "news_ctx"), f32(user_ctx, news_ctx, "user_ctx"),
                           const_cast<float*>(f32(logits, news_ctx, "logits")), (int)news_ctx.size(0), (int)news_ctx.size(1), stream_of(news_ctx)),
          "digat_row_logits");
}

// the drop-in path's search for runs of identical consecutive user rows (digat_user_row_runs)
void user_row_runs(const at::Tensor& ue, const at::Tensor& Au, const at::Tensor& cm, const at::Tensor& ci, at::Tensor& row_group,
                   at::Tensor& leaders, at::Tensor& n_runs, at::Tensor& ws) {
    need(ue.dim() == 3 && Au.dim() == 3 && cm.dim() == 2 && ci.dim() == 2, "ue [B,H,d], user_graph [B,U,U], mask [B,C1], indices [B,H]");
    const int B = (int)ue.size(0);
    check(digat_user_row_runs(f32(ue, ue, "user_news_embedding"), bytes(Au, ue, "user_graph"), bytes(cm, ue, "user_category_mask"),
                              i64(ci, ue, "user_category_indices"), B, (int)ue.size(1), (int)Au.size(1), (int)cm.size(1), (int)ue.size(2),
                              const_cast<int32_t*>(i32(row_group, ue, "row_group")), const_cast<int64_t*>(i64(leaders, ue, "leaders")),
                              const_cast<int32_t*>(i32(n_runs, ue, "n_runs")), on_gpu(ws, ue, "workspace").data_ptr(), (size_t)ws.nbytes(),
                              stream_of(ue)),
          "digat_user_row_runs");
}

// ---- training: one call per function and direction (digat_*_fwd_train / digat_*_bwd, include/digat_hip.h "training") ---------------
// The output, `save` and gradient tensors are allocated here and the scratch buffer is cached per (device, stream): a 64 x 5-row
// step makes 45 of these calls with 20-35 arguments each, and the step was host-bound through their marshalling (round 5).

at::Tensor scratch(const at::Tensor& like, size_t bytes) {       // stream-ordered reuse, as digat_amd/_lib.workspace
    static std::mutex mu;
    // never destroyed: a tensor released by a static destructor at interpreter exit would reach the allocator after it is gone
    static auto* cache = new std::map<std::pair<int, void*>, at::Tensor>();
    const std::pair<int, void*> key{(int)like.device().index(), stream_of(like)};
    std::lock_guard<std::mutex> lock(mu);
    at::Tensor& t = (*cache)[key];
    if (!t.defined() || (size_t)t.numel() < bytes) t = at::empty({(int64_t)std::max<size_t>(bytes, 256)}, like.options().dtype(at::kByte));
    return t;
}
at::Tensor byte_buffer(const at::Tensor& like, size_t bytes) { return at::empty({(int64_t)std::max<size_t>(bytes, 256)}, like.options().dtype(at::kByte)); }
float* out_f32(at::Tensor& t) { return t.data_ptr<float>(); }
// the context accumulated so far ([B,d], added to the call's output), or nothing
const float* prev_of(const c10::optional<at::Tensor>& prev, const at::Tensor& like, int B, int d) {
    if (!prev.has_value()) return nullptr;
    shape(*prev, {B, d}, "prev");
    return f32(*prev, like, "prev");
}
// a ready-made split image (digat_split_jobs), or nothing: a byte tensor on the inputs' device, at least `bytes` long
const void* image_of(const c10::optional<at::Tensor>& image, const at::Tensor& like, size_t bytes) {
    if (!image.has_value()) return nullptr;
    need(image->is_cuda() && image->device() == like.device() && image->scalar_type() == at::kByte && image->is_contiguous() &&
         (size_t)image->numel() >= bytes, "image: a contiguous uint8 tensor on the inputs' device holding the whole split image");
    return image->data_ptr();
}

std::tuple<at::Tensor, at::Tensor> xattn_fwd_train(const at::Tensor& Xd, const at::Tensor& A, const at::Tensor& cvec, const at::Tensor& W,
                                                   const at::Tensor& bW, const at::Tensor& F1, const at::Tensor& F2, const at::Tensor& F3,
                                                   const at::Tensor& b3, const at::Tensor& a, double p, int64_t seed, double p_in, int64_t seed_in,
                                                   const c10::optional<at::Tensor>& image, int64_t xattn_mode) {
    need(Xd.dim() == 3, "Xd [B,n,d] expected");
    const int B = (int)Xd.size(0), n = (int)Xd.size(1), d = (int)Xd.size(2);
    const c10::DeviceGuard guard(Xd.device());          // allocations and the launch on the input's device, whatever the current one is
    shape(A, {B, n, n}, "A"); shape(cvec, {B, d}, "ctx"); shape(W, {d, d}, "W"); shape(F1, {d, d}, "F1"); shape(F2, {d, d}, "F2"); shape(F3, {d, d}, "F3");
    shape(bW, {d}, "bW"); shape(b3, {d}, "b3"); need(a.numel() == d, "a [1,d] expected");
    at::Tensor out = at::empty_like(Xd);
    const size_t nsave = digat_xattn_train_save_bytes(B, n, d), nws = digat_xattn_train_workspace_bytes(B, n, d);
    at::Tensor save = byte_buffer(Xd, nsave), ws = scratch(Xd, nws);
    check(digat_xattn_fwd_train(f32(Xd, Xd, "Xd"), bytes(A, Xd, "A"), f32(cvec, Xd, "ctx"), f32(W, Xd, "W"), f32(bW, Xd, "bW"), f32(F1, Xd, "F1"),
                                f32(F2, Xd, "F2"), f32(F3, Xd, "F3"), f32(b3, Xd, "b3"), f32(a, Xd, "a"), out_f32(out), (float)p, (uint32_t)seed, (float)p_in,
                                (uint32_t)seed_in, B, n, d,
                                save.data_ptr(), nsave, ws.data_ptr(), nws, image_of(image, Xd, digat_split_job_bytes(d, d, 0, 3)), (int)xattn_mode,
                                stream_of(Xd)), "digat_xattn_fwd_train");
    return {out, save};
}

// -> dX, dctx, dW3 ([3,d,d]: dW, dF1, dF2 written in place as one product), dbW, dF3, db3, da
std::vector<at::Tensor> xattn_bwd(const at::Tensor& dOut, const at::Tensor& out, const at::Tensor& Xd, const at::Tensor& A, const at::Tensor& cvec,
                                  const at::Tensor& W, const at::Tensor& F1, const at::Tensor& F2, const at::Tensor& F3, const at::Tensor& a, double p,
                                  double p_in, const at::Tensor& save, const c10::optional<at::Tensor>& image, int64_t xattn_mode) {
    need(Xd.dim() == 3, "Xd [B,n,d] expected");
    const int B = (int)Xd.size(0), n = (int)Xd.size(1), d = (int)Xd.size(2);
    const c10::DeviceGuard guard(Xd.device());
    shape(dOut, {B, n, d}, "dOut"); shape(out, {B, n, d}, "out"); shape(A, {B, n, n}, "A"); shape(cvec, {B, d}, "ctx");
    shape(W, {d, d}, "W"); shape(F1, {d, d}, "F1"); shape(F2, {d, d}, "F2"); shape(F3, {d, d}, "F3"); need(a.numel() == d, "a [1,d] expected");
    const size_t nsave = digat_xattn_train_save_bytes(B, n, d), nws = digat_xattn_train_workspace_bytes(B, n, d);
    need(save.is_cuda() && save.device() == Xd.device() && save.scalar_type() == at::kByte && save.is_contiguous() && (size_t)save.numel() >= nsave,
         "save: the uint8 buffer the forward call returned (same device, same shapes)");
    at::Tensor ws = scratch(Xd, nws);
    at::Tensor dX = at::empty_like(Xd), dc = at::empty_like(cvec), dW3 = at::empty({3, W.size(0), W.size(1)}, W.options()), dF3 = at::empty_like(W);
    at::Tensor dbW = at::empty({d}, W.options()), db3 = at::empty({d}, W.options()), da = at::empty({d}, W.options());
    float* w3 = out_f32(dW3);
    const size_t dd = (size_t)W.size(0) * W.size(1);
    check(digat_xattn_bwd(f32(dOut, Xd, "dOut"), f32(out, Xd, "out"), f32(Xd, Xd, "Xd"), bytes(A, Xd, "A"), f32(cvec, Xd, "ctx"), f32(W, Xd, "W"),
                          f32(F1, Xd, "F1"), f32(F2, Xd, "F2"), f32(F3, Xd, "F3"), f32(a, Xd, "a"), (float)p, (float)p_in, save.data_ptr(), nsave, out_f32(dX),
                          out_f32(dc), w3, out_f32(dbW), w3 + dd, w3 + 2 * dd, out_f32(dF3), out_f32(db3), out_f32(da), B, n, d, ws.data_ptr(), nws,
                          image_of(image, Xd, digat_split_job_bytes(d, d, 1, 3)), (int)xattn_mode, stream_of(Xd)), "digat_xattn_bwd");
    return {dX, dc, dW3, dbW, dF3, db3, da};
}

std::tuple<at::Tensor, at::Tensor> news_ctx_fwd_train(const at::Tensor& X, const at::Tensor& mask, const at::Tensor& Kc, const at::Tensor& Qc,
                                                      const at::Tensor& bQc, const at::Tensor& Wg, const at::Tensor& bg, double p, int64_t seed,
                                                      const c10::optional<at::Tensor>& prev) {
    need(X.dim() == 3, "X [B,N,d] expected");
    const int B = (int)X.size(0), N = (int)X.size(1), d = (int)X.size(2);
    const c10::DeviceGuard guard(X.device());
    shape(mask, {B, N}, "mask"); shape(Kc, {d, d}, "Kc"); shape(Qc, {d, d}, "Qc"); shape(bQc, {d}, "bQc"); shape(Wg, {d, 2 * d}, "Wg"); shape(bg, {d}, "bg");
    at::Tensor out = at::empty({B, d}, X.options());
    const size_t nsave = digat_news_ctx_train_save_bytes(B, N, d), nws = digat_news_ctx_train_workspace_bytes(B, N, d);
    at::Tensor save = byte_buffer(X, nsave), ws = scratch(X, nws);
    check(digat_news_ctx_fwd_train(f32(X, X, "X"), bytes(mask, X, "mask"), f32(Kc, X, "Kc"), f32(Qc, X, "Qc"), f32(bQc, X, "bQc"), f32(Wg, X, "Wg"),
                                   f32(bg, X, "bg"), out_f32(out), (float)p, (uint32_t)seed, B, N, d, save.data_ptr(), nsave, ws.data_ptr(), nws,
                                   prev_of(prev, X, B, d), stream_of(X)), "digat_news_ctx_fwd_train");
    return {out, save};
}

// grads: dKc, dQc, dbQc, dWg, dbg (the caller's: written, or added to when accumulate); -> dX
at::Tensor news_ctx_bwd(const at::Tensor& dout, const at::Tensor& X, const at::Tensor& mask, const at::Tensor& Kc, const at::Tensor& Qc,
                        const at::Tensor& Wg, double p, const at::Tensor& save, std::vector<at::Tensor> grads, bool accumulate) {
    need(grads.size() == 5, "five parameter-gradient tensors expected");
    need(X.dim() == 3, "X [B,N,d] expected");
    const int B = (int)X.size(0), N = (int)X.size(1), d = (int)X.size(2);
    const c10::DeviceGuard guard(X.device());
    shape(dout, {B, d}, "dout"); shape(mask, {B, N}, "mask"); shape(Kc, {d, d}, "Kc"); shape(Qc, {d, d}, "Qc"); shape(Wg, {d, 2 * d}, "Wg");
    shape(grads[0], {d, d}, "dKc"); shape(grads[1], {d, d}, "dQc"); shape(grads[2], {d}, "dbQc"); shape(grads[3], {d, 2 * d}, "dWg"); shape(grads[4], {d}, "dbg");
    const size_t nsave = digat_news_ctx_train_save_bytes(B, N, d), nws = digat_news_ctx_train_workspace_bytes(B, N, d);
    need(save.is_cuda() && save.device() == X.device() && save.scalar_type() == at::kByte && save.is_contiguous() && (size_t)save.numel() >= nsave,
         "save: the uint8 buffer the forward call returned (same device, same shapes)");
    at::Tensor ws = scratch(X, nws), dX = at::empty_like(X);
    for (auto& g : grads) f32(g, X, "parameter gradient");
    check(digat_news_ctx_bwd(f32(dout, X, "dout"), f32(X, X, "X"), bytes(mask, X, "mask"), f32(Kc, X, "Kc"), f32(Qc, X, "Qc"), f32(Wg, X, "Wg"), (float)p,
                             save.data_ptr(), nsave, out_f32(dX), out_f32(grads[0]), out_f32(grads[1]), out_f32(grads[2]), out_f32(grads[3]),
                             out_f32(grads[4]), B, N, d, accumulate ? 1 : 0, ws.data_ptr(), nws, stream_of(X)), "digat_news_ctx_bwd");
    return dX;
}

std::tuple<at::Tensor, at::Tensor> user_ctx_fwd_train(const at::Tensor& Xu, const at::Tensor& cat_mask, const at::Tensor& cat_idx, const at::Tensor& c_n,
                                                      const at::Tensor& Ku, const at::Tensor& Qu, const at::Tensor& bQu, const at::Tensor& Fa,
                                                      const at::Tensor& bFa, const at::Tensor& Kua, const at::Tensor& Qua, const at::Tensor& bQua,
                                                      int64_t H, int64_t C1, double p, int64_t seed, const c10::optional<at::Tensor>& image,
                                                      const c10::optional<at::Tensor>& prev) {
    need(Xu.dim() == 3, "Xu [B,U,d] expected");
    const int B = (int)Xu.size(0), U = (int)Xu.size(1), d = (int)Xu.size(2);
    const c10::DeviceGuard guard(Xu.device());
    need(H >= 0 && H <= U && C1 >= 1, "H history rows of the U nodes, C1 = category_num + 1 buckets");
    shape(cat_mask, {B, C1}, "cat_mask"); shape(cat_idx, {B, H}, "cat_idx"); shape(c_n, {B, d}, "c_n");
    shape(Ku, {d, d}, "Ku"); shape(Qu, {d, d}, "Qu"); shape(bQu, {d}, "bQu"); shape(Fa, {d, d}, "Fa"); shape(bFa, {d}, "bFa");
    shape(Kua, {d, d}, "Kua"); shape(Qua, {d, d}, "Qua"); shape(bQua, {d}, "bQua");
    at::Tensor out = at::empty({B, d}, Xu.options());
    const size_t nsave = digat_user_ctx_train_save_bytes(B, U, (int)H, (int)C1, d), nws = digat_user_ctx_train_workspace_bytes(B, U, (int)H, (int)C1, d);
    at::Tensor save = byte_buffer(Xu, nsave), ws = scratch(Xu, nws);
    check(digat_user_ctx_fwd_train(f32(Xu, Xu, "Xu"), bytes(cat_mask, Xu, "cat_mask"), i64(cat_idx, Xu, "cat_idx"), f32(c_n, Xu, "c_n"), f32(Ku, Xu, "Ku"),
                                   f32(Qu, Xu, "Qu"), f32(bQu, Xu, "bQu"), f32(Fa, Xu, "Fa"), f32(bFa, Xu, "bFa"), f32(Kua, Xu, "Kua"), f32(Qua, Xu, "Qua"),
                                   f32(bQua, Xu, "bQua"), out_f32(out), (float)p, (uint32_t)seed, B, U, (int)H, (int)C1, d, save.data_ptr(), nsave,
                                   ws.data_ptr(), nws, image_of(image, Xu, digat_split_job_bytes(d, d, 0, 1)), prev_of(prev, Xu, B, d), stream_of(Xu)),
          "digat_user_ctx_fwd_train");
    return {out, save};
}

// grads: dKu, dQu, dFa, dKua, dQua, dbQu, dbFa, dbQua (the caller's); -> dXu, dc_n
std::tuple<at::Tensor, at::Tensor> user_ctx_bwd(const at::Tensor& dout, const at::Tensor& Xu, const at::Tensor& cat_mask, const at::Tensor& cat_idx,
                                                const at::Tensor& c_n, const at::Tensor& Ku, const at::Tensor& Qu, const at::Tensor& Fa,
                                                const at::Tensor& Kua, const at::Tensor& Qua, double p, const at::Tensor& save,
                                                std::vector<at::Tensor> grads, bool accumulate, int64_t H, int64_t C1,
                                                const c10::optional<at::Tensor>& image) {
    need(grads.size() == 8, "eight parameter-gradient tensors expected");
    need(Xu.dim() == 3, "Xu [B,U,d] expected");
    const int B = (int)Xu.size(0), U = (int)Xu.size(1), d = (int)Xu.size(2);
    const c10::DeviceGuard guard(Xu.device());
    need(H >= 0 && H <= U && C1 >= 1, "H history rows of the U nodes, C1 = category_num + 1 buckets");
    shape(dout, {B, d}, "dout"); shape(cat_mask, {B, C1}, "cat_mask"); shape(cat_idx, {B, H}, "cat_idx"); shape(c_n, {B, d}, "c_n");
    shape(Ku, {d, d}, "Ku"); shape(Qu, {d, d}, "Qu"); shape(Fa, {d, d}, "Fa"); shape(Kua, {d, d}, "Kua"); shape(Qua, {d, d}, "Qua");
    for (int k = 0; k < 5; ++k) shape(grads[k], {d, d}, "weight gradient");
    for (int k = 5; k < 8; ++k) shape(grads[k], {d}, "bias gradient");
    const size_t nsave = digat_user_ctx_train_save_bytes(B, U, (int)H, (int)C1, d), nws = digat_user_ctx_train_workspace_bytes(B, U, (int)H, (int)C1, d);
    need(save.is_cuda() && save.device() == Xu.device() && save.scalar_type() == at::kByte && save.is_contiguous() && (size_t)save.numel() >= nsave,
         "save: the uint8 buffer the forward call returned (same device, same shapes)");
    at::Tensor ws = scratch(Xu, nws), dXu = at::empty_like(Xu), dc = at::empty_like(c_n);
    for (auto& g : grads) f32(g, Xu, "parameter gradient");
    check(digat_user_ctx_bwd(f32(dout, Xu, "dout"), f32(Xu, Xu, "Xu"), bytes(cat_mask, Xu, "cat_mask"), i64(cat_idx, Xu, "cat_idx"), f32(c_n, Xu, "c_n"),
                             f32(Ku, Xu, "Ku"), f32(Qu, Xu, "Qu"), f32(Fa, Xu, "Fa"), f32(Kua, Xu, "Kua"), f32(Qua, Xu, "Qua"), (float)p, save.data_ptr(), nsave,
                             out_f32(dXu), out_f32(dc), out_f32(grads[0]), out_f32(grads[1]), out_f32(grads[5]), out_f32(grads[2]), out_f32(grads[6]),
                             out_f32(grads[3]), out_f32(grads[4]), out_f32(grads[7]), B, U, (int)H, (int)C1, d, accumulate ? 1 : 0, ws.data_ptr(), nws,
                             image_of(image, Xu, digat_split_job_bytes(d, d, 1, 1)), stream_of(Xu)), "digat_user_ctx_bwd");
    return {dXu, dc};
}

std::tuple<at::Tensor, at::Tensor> dropout_fwd(const at::Tensor& x, double p, int64_t seed) {
    const c10::DeviceGuard guard(x.device());
    at::Tensor y = at::empty_like(x), mask = at::empty(x.sizes(), x.options().dtype(at::kByte));
    check(digat_dropout_fwd(f32(x, x, "x"), out_f32(y), static_cast<uint8_t*>(mask.data_ptr()), x.numel(), (float)p, (uint32_t)seed, stream_of(x)),
          "digat_dropout_fwd");
    return {y, mask};
}
at::Tensor dropout_bwd(const at::Tensor& dy, const at::Tensor& mask, double p) {
    const c10::DeviceGuard guard(dy.device());
    need(mask.numel() == dy.numel(), "mask: one byte per element of dy");
    at::Tensor dx = at::empty_like(dy);
    check(digat_dropout_bwd(f32(dy, dy, "dy"), bytes(mask, dy, "mask"), out_f32(dx), dy.numel(), (float)p, stream_of(dy)), "digat_dropout_bwd");
    return dx;
}

}  // namespace

PYBIND11_MODULE(digat_torch_ext, m) {
    m.doc() = "thin torch extension over libdigat_hip.so's C ABI (include/digat_hip.h)";
    m.def("abi_version", []() { return digat_version(); });
    m.def("encoder_fwd", &encoder_fwd);
    m.def("encoder_fwd_grouped", &encoder_fwd_grouped);
    m.def("row_logits", &row_logits);
    m.def("user_row_runs", &user_row_runs);
    m.def("xattn_fwd_train", &xattn_fwd_train);
    m.def("xattn_bwd", &xattn_bwd);
    m.def("news_ctx_fwd_train", &news_ctx_fwd_train);
    m.def("news_ctx_bwd", &news_ctx_bwd);
    m.def("user_ctx_fwd_train", &user_ctx_fwd_train);
    m.def("user_ctx_bwd", &user_ctx_bwd);
    m.def("dropout_fwd", &dropout_fwd);
    m.def("dropout_bwd", &dropout_bwd);
}
