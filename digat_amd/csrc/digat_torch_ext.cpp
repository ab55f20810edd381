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

#include <string>

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

}  // namespace

PYBIND11_MODULE(digat_torch_ext, m) {
    m.doc() = "thin torch extension over libdigat_hip.so's C ABI (include/digat_hip.h)";
    m.def("abi_version", []() { return digat_version(); });
    m.def("encoder_fwd", &encoder_fwd);
    m.def("encoder_fwd_grouped", &encoder_fwd_grouped);
    m.def("row_logits", &row_logits);
    m.def("user_row_runs", &user_row_runs);
}
