"""Training-mode forward of the DIGAT encoder: autograd through native HIP forward/backward pairs.

The reference trains by plain autograd through ``graphEncoders.DIGAT.forward`` (trainer.py:98-102).
Here the same forward (graphEncoders.py:177-187, with its three dropouts live) is composed from a small
set of ``torch.autograd.Function``s whose forward AND backward are calls into ``libdigat_hip.so``:

    XattnFused            Eq. 8 layer: K3, projections, score / softmax / attention dropout / aggregation
                          (``digat_xattn_fwd_train`` / ``digat_xattn_bwd``); the backward RECOMPUTES
                          relu'(K3+K1+K2) from the saved projections, [B,n,n,d] is never stored
    NewsCtxFused          compute_news_graph_context (``digat_news_ctx_fwd_train`` / ``digat_news_ctx_bwd``)
    UserCtxFused          compute_user_graph_context (``digat_user_ctx_fwd_train`` / ``digat_user_ctx_bwd``)
    Dropout               counter-hash dropout (own RNG: masks differ from torch's, statistics do not)

One library call per function and direction: a training step is ~45 calls from the host; the
finer-grained Functions below (Linear, MatmulW, AttnPool, TopicPool, GateMix, ReluRes, XattnLayer) wrap the primitives
those calls are composed of and are kept for tests and for callers that build other encoders from them.

PyTorch does what it does for the reference too: owns the tensors, records the graph, and runs the
few pure data-movement ops (``cat``, ``select``, ``expand``, ``+`` of two contexts).  There is no
eager fallback for any arithmetic of the path: every Function raises through ``_lib.check``.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _lib

L = _lib.lib
S = _lib.stream_ptr


def _seed() -> int:
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if not t.is_contiguous() else t


def _rows(x: torch.Tensor):
    """(tensor, M, K, ld): x viewed as M rows of K contiguous floats, row stride ld."""
    if x.dim() == 2 and x.stride(1) == 1:
        return x, x.shape[0], x.shape[1], x.stride(0)
    x = _f(x)
    K = x.shape[-1]
    return x, x.numel() // K, K, K


def _x3_ok(M, n_out, k_in):
    """Shapes the bf16x6 matrix-core kernel serves (fp32-grade: exact 3-way bf16 split, 6 products, fp32 accumulate)."""
    return M >= 2048 and n_out % 80 == 0 and k_in % 8 == 0 and k_in >= 32


def _wsplit(n_out, k_in, dev):
    nb = L().digat_split_weights_bytes(n_out, k_in)
    return _lib.workspace(nb, dev, "wsplit")     # rewritten by every call (weights change each step); stream-ordered reuse


def _linear_fwd(x_ptr, ld, W, b_ptr, y_ptr, M, N, K, dev, what):
    """y[M,N] = x[M,K] W[N,K]^T (+ b)"""
    if _x3_ok(M, N, K) and ld % 4 == 0:
        # the training path's operand format is bf16x6: gradients have no lower bound, fp16 pieces of 1e-6 are subnormal or zero
        _lib.check(L().digat_linear_f32x3(x_ptr, ld, W.data_ptr(), b_ptr, y_ptr, N, M, N, K, _wsplit(N, K, dev).data_ptr(), _lib.GEMM_BF16X6, S()), what)
    else:
        _lib.check(L().digat_linear_f32(x_ptr, ld, W.data_ptr(), b_ptr, y_ptr, N, M, N, K, S()), what)


def _linear_bwd_input(dy_ptr, W, dx_ptr, M, N, K, accumulate, dev, what):
    """dx[M,K] (+)= dy[M,N] W[N,K]"""
    if _x3_ok(M, K, N):
        _lib.check(L().digat_linear_bwd_input_x3(dy_ptr, N, W.data_ptr(), dx_ptr, K, M, N, K, accumulate,
                                                 _wsplit(K, N, dev).data_ptr(), S()), what)
    else:
        _lib.check(L().digat_linear_bwd_input(dy_ptr, N, W.data_ptr(), dx_ptr, K, M, N, K, accumulate, S()), what)


class Linear(Function):
    """y = x W^T + b  (W [N,K] as in nn.Linear)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x, M, K, ld = _rows(x)
        N = W.shape[0]
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        if M:
            _linear_fwd(x.data_ptr(), ld, W, _lib.ptr(b), y.data_ptr(), M, N, K, x.device, "digat_linear_f32")
        ctx.save_for_backward(x, W)
        ctx.dims = (M, N, K, ld, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        M, N, K, ld, has_b = ctx.dims
        dy = _f(dy)
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        dW = torch.empty_like(W)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if has_b else None
        if M:
            _linear_bwd_input(dy.data_ptr(), W, dx.data_ptr(), M, N, K, 0, x.device, "digat_linear_bwd_input")
            nb = L().digat_linear_bwd_weight_workspace(M, N, K)
            ws = _lib.workspace(nb, x.device, "dW")
            _lib.check(L().digat_linear_bwd_weight(dy.data_ptr(), N, x.data_ptr(), ld, dW.data_ptr(), _lib.ptr(db), M, N, K, 0,
                                                   ws.data_ptr(), nb, S()), "digat_linear_bwd_weight")
        else:
            dx.zero_(); dW.zero_()
            if db is not None:
                db.zero_()
        return dx, dW, db


class MatmulW(Function):
    """y[M,K] = x[M,N] @ W[N,K]   (kq = q @ K.weight: the key projection moved onto the query)."""

    @staticmethod
    def forward(ctx, x, W):
        x = _f(x)
        M, N = x.shape
        K = W.shape[1]
        y = torch.empty((M, K), dtype=torch.float32, device=x.device)
        _linear_bwd_input(x.data_ptr(), W, y.data_ptr(), M, N, K, 0, x.device, "matmul_w")
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _f(dy)
        M, N = x.shape
        K = W.shape[1]
        dx = torch.empty_like(x)
        dW = torch.empty_like(W)
        # dx = dy @ W^T  -> linear(dy, W);  dW[n][k] = sum_m x[m][n] dy[m][k]
        _linear_fwd(dy.data_ptr(), K, W, None, dx.data_ptr(), M, N, K, x.device, "matmul_w dx")
        nb = L().digat_linear_bwd_weight_workspace(M, N, K)
        ws = _lib.workspace(nb, x.device, "dW")
        _lib.check(L().digat_linear_bwd_weight(x.data_ptr(), N, dy.data_ptr(), K, dW.data_ptr(), None, M, N, K, 0,
                                               ws.data_ptr(), nb, S()), "matmul_w dW")
        return dx, dW


class Dropout(Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _f(x)
        E = _lib.ext()
        if E is not None:
            y, mask = E.dropout_fwd(x, float(p), _seed())
            ctx.save_for_backward(mask)
            ctx.p = float(p)
            return y
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        _lib.check(L().digat_dropout_fwd(x.data_ptr(), y.data_ptr(), mask.data_ptr(), x.numel(), float(p), _seed(), S()),
                   "digat_dropout_fwd")
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _f(dy)
        E = _lib.ext()
        if E is not None:
            return E.dropout_bwd(dy, mask, ctx.p), None
        dx = torch.empty_like(dy)
        _lib.check(L().digat_dropout_bwd(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), ctx.p, S()),
                   "digat_dropout_bwd")
        return dx, None


class RowLogits(Function):
    """logits[b] = sum_d user_ctx[b] news_ctx[b] (model.py:75), one launch each way (``digat_row_logits`` / ``digat_row_logits_bwd``)."""

    @staticmethod
    def forward(ctx, news_ctx, user_ctx):
        news_ctx, user_ctx = _f(news_ctx), _f(user_ctx)
        B, d = news_ctx.shape
        logits = torch.empty(B, dtype=torch.float32, device=news_ctx.device)
        if B:
            _lib.check(L().digat_row_logits(news_ctx.data_ptr(), user_ctx.data_ptr(), logits.data_ptr(), B, d, S()), "digat_row_logits")
        ctx.save_for_backward(news_ctx, user_ctx)
        return logits

    @staticmethod
    def backward(ctx, dl):
        news_ctx, user_ctx = ctx.saved_tensors
        B, d = news_ctx.shape
        dl = _f(dl)
        dn, du = torch.empty_like(news_ctx), torch.empty_like(user_ctx)
        _lib.check(L().digat_row_logits_bwd(dl.data_ptr(), news_ctx.data_ptr(), user_ctx.data_ptr(), dn.data_ptr(), du.data_ptr(), B, d, S()),
                   "digat_row_logits_bwd")
        return dn, du


class ClickLoss(Function):
    """trainer.py:100: mean(-log_softmax(logits [B,K], dim=1)[:, 0]) and its gradient from one launch (``digat_click_loss``)."""

    @staticmethod
    def forward(ctx, logits):
        logits = _f(logits)
        B, K = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        _lib.check(L().digat_click_loss(logits.data_ptr(), B, K, loss.data_ptr(), dlogits.data_ptr(), S()), "digat_click_loss")
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dloss


class TableLookup2(Function):
    """Rows of ONE embedding table for two id lists — the candidate graphs' nodes and the histories of a training batch
    (model.py:72-73 through a table-backed news encoder) — with ONE dense table gradient from one launch
    (``digat_embedding_bwd_unsorted``: equal ids added in a fixed order, no sort, two launches), where
    ``F.embedding``'s backward sorts, scans and scatters per lookup (18 launches each at these sizes) and autograd then adds the
    two 104 MB gradients."""

    @staticmethod
    def forward(ctx, table, ids_a, ids_b):
        ids_a, ids_b = ids_a.long().contiguous(), ids_b.long().contiguous()
        ctx.save_for_backward(ids_a, ids_b)
        ctx.table_shape = tuple(table.shape)
        return torch.nn.functional.embedding(ids_a, table), torch.nn.functional.embedding(ids_b, table)

    @staticmethod
    def backward(ctx, ga, gb):
        ids_a, ids_b = ctx.saved_tensors
        V, dm = ctx.table_shape
        dev = ids_a.device
        dtable = torch.zeros((V, dm), dtype=torch.float32, device=dev)
        ga = _f(ga) if ga is not None else None
        gb = _f(gb) if gb is not None else None
        Ma, Mb = (ids_a.numel() if ga is not None else 0), (ids_b.numel() if gb is not None else 0)
        nb = L().digat_embedding_bwd_unsorted_workspace_bytes(Ma + Mb, dm, V)
        ws = _lib.workspace(nb, dev, "embedding_bwd")
        _lib.check(L().digat_embedding_bwd_unsorted(ids_a.data_ptr() if Ma else None, _lib.ptr(ga), dm, Ma, ids_b.data_ptr() if Mb else None,
                                                    _lib.ptr(gb), dm, Mb, dm, V, dtable.data_ptr(), ws.data_ptr(), nb, S()),
                   "digat_embedding_bwd_unsorted")
        return dtable, None, None


def dropout(x, p, training=True):
    return Dropout.apply(x, p) if (training and p > 0) else x


class AttnPool(Function):
    """ScaledDotProductAttention pooling with the folded query kq: out = sum_j softmax(x_j.kq/sqrt(d)) x_j."""

    @staticmethod
    def forward(ctx, feat, kq, mask):
        feat, kq = _f(feat), _f(kq)
        B, n, d = feat.shape
        out = torch.empty((B, d), dtype=torch.float32, device=feat.device)
        alpha = torch.empty((B, n), dtype=torch.float32, device=feat.device)
        _lib.check(L().digat_attn_pool_fwd(feat.data_ptr(), n * d, kq.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                           alpha.data_ptr(), B, n, d, S()), "digat_attn_pool_fwd")
        ctx.save_for_backward(feat, kq, mask, alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, kq, mask, alpha = ctx.saved_tensors
        B, n, d = feat.shape
        dout = _f(dout)
        dfeat = torch.empty_like(feat)
        dkq = torch.empty_like(kq)
        _lib.check(L().digat_attn_pool_bwd(feat.data_ptr(), n * d, kq.data_ptr(), mask.data_ptr(), alpha.data_ptr(),
                                           dout.data_ptr(), dfeat.data_ptr(), n * d, dkq.data_ptr(), B, n, d, 0, S()),
                   "digat_attn_pool_bwd")
        return dfeat, dkq, None


class TopicPool(Function):
    """scatter_softmax + scatter_sum over the first H rows of Xu grouped by category (graphEncoders.py:126-130)."""

    @staticmethod
    def forward(ctx, Xu, kq, idx, H, C1):
        Xu, kq = _f(Xu), _f(kq)
        B, U, d = Xu.shape
        T = torch.empty((B, C1, d), dtype=torch.float32, device=Xu.device)
        alpha = torch.empty((B, H), dtype=torch.float32, device=Xu.device)
        _lib.check(L().digat_topic_pool_fwd_train(Xu.data_ptr(), kq.data_ptr(), idx.data_ptr(), T.data_ptr(), alpha.data_ptr(),
                                                  B, U, H, C1, d, S()), "digat_topic_pool_fwd_train")
        ctx.save_for_backward(Xu, kq, idx, alpha)
        ctx.dims = (H, C1)
        return T

    @staticmethod
    def backward(ctx, dT):
        Xu, kq, idx, alpha = ctx.saved_tensors
        H, C1 = ctx.dims
        B, U, d = Xu.shape
        dT = _f(dT)
        dXu = torch.zeros_like(Xu)              # topic rows receive no gradient from the pooling
        dkq = torch.empty_like(kq)
        _lib.check(L().digat_topic_pool_bwd(Xu.data_ptr(), kq.data_ptr(), idx.data_ptr(), alpha.data_ptr(), dT.data_ptr(),
                                            dXu.data_ptr(), dkq.data_ptr(), B, U, H, C1, d, S()), "digat_topic_pool_bwd")
        return dXu, dkq, None, None, None


class GateMix(Function):
    """out = sigmoid(z) * l + (1 - sigmoid(z)) * g   (graphEncoders.py:112-113)."""

    @staticmethod
    def forward(ctx, z, l, g):
        z, l, g = _f(z), _f(l), _f(g)
        B, d = z.shape
        out = torch.empty_like(z)
        _lib.check(L().digat_gate_fwd(z.data_ptr(), l.data_ptr(), d, g.data_ptr(), out.data_ptr(), B, d, S()), "digat_gate_fwd")
        ctx.save_for_backward(z, l, g)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, l, g = ctx.saved_tensors
        B, d = z.shape
        dout = _f(dout)
        dz, dl, dg = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        _lib.check(L().digat_gate_bwd(dout.data_ptr(), z.data_ptr(), l.data_ptr(), d, g.data_ptr(), dz.data_ptr(), dl.data_ptr(),
                                      dg.data_ptr(), B, d, S()), "digat_gate_bwd")
        return dz, dl, dg


class ReluRes(Function):
    """relu(y) + t   (graphEncoders.py:131)."""

    @staticmethod
    def forward(ctx, y, t):
        y, t = _f(y), _f(t)
        out = torch.empty_like(y)
        _lib.check(L().digat_relu_res_fwd(y.data_ptr(), t.data_ptr(), out.data_ptr(), y.numel(), S()), "digat_relu_res_fwd")
        ctx.save_for_backward(y)
        return out

    @staticmethod
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        dout = _f(dout)
        dy = torch.empty_like(y)
        _lib.check(L().digat_relu_mask(dout.data_ptr(), y.data_ptr(), dy.data_ptr(), y.numel(), S()), "digat_relu_mask")
        return dy, dout


class XattnLayer(Function):
    """Eq. 8 layer on already-dropped-out inputs Xd: out = relu(dropout_p(alpha) @ h) + Xd."""

    @staticmethod
    def forward(ctx, Xd, A, r, W, bW, F1, F2, a, p_alpha):
        Xd, r = _f(Xd), _f(r)
        B, n, d = Xd.shape
        dev = Xd.device
        h, Pr, Q = (torch.empty_like(Xd) for _ in range(3))
        if _x3_ok(B * n, d, d):
            _lib.check(L().digat_xattn_project_x3(Xd.data_ptr(), r.data_ptr(), W.data_ptr(), bW.data_ptr(), F1.data_ptr(),
                                                  F2.data_ptr(), h.data_ptr(), Pr.data_ptr(), Q.data_ptr(), B, n, d,
                                                  _wsplit(3 * d, d, dev).data_ptr(), S()), "digat_xattn_project_x3")
        else:
            _lib.check(L().digat_xattn_project(Xd.data_ptr(), r.data_ptr(), W.data_ptr(), bW.data_ptr(), F1.data_ptr(),
                                               F2.data_ptr(), h.data_ptr(), Pr.data_ptr(), Q.data_ptr(), B, n, d, S()),
                       "digat_xattn_project")
        out = torch.empty_like(Xd)
        alpha = torch.empty((B, n, n), dtype=torch.float32, device=dev)
        s_pre = torch.zeros((B, n, n), dtype=torch.float32, device=dev)
        p = float(p_alpha)
        adrop = torch.empty_like(alpha) if p > 0 else None
        amask = torch.empty((B, n, n), dtype=torch.uint8, device=dev) if p > 0 else None
        _lib.check(L().digat_xattn_pairwise_fwd_train(Pr.data_ptr(), Q.data_ptr(), h.data_ptr(), Xd.data_ptr(), a.data_ptr(),
                                                      A.data_ptr(), out.data_ptr(), alpha.data_ptr(), s_pre.data_ptr(),
                                                      _lib.ptr(adrop), _lib.ptr(amask), p, _seed() if p > 0 else 0, B, n, d, S()),
                   "digat_xattn_pairwise_fwd_train")
        ctx.save_for_backward(Xd, A, W, F1, F2, a, h, Pr, Q, out, alpha, s_pre, amask if amask is not None else A)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, dOut):
        Xd, A, W, F1, F2, a, h, Pr, Q, out, alpha, s_pre, amask = ctx.saved_tensors
        B, n, d = Xd.shape
        dev = Xd.device
        dOut = _f(dOut)
        M = B * n
        dPr, dQ, dh = (torch.empty_like(Xd) for _ in range(3))
        da = torch.empty_like(a)
        nb = L().digat_xattn_pairwise_bwd_workspace(B, n, d)
        ws = _lib.workspace(nb, dev, "xattn_bwd")
        _lib.check(L().digat_xattn_pairwise_bwd(dOut.data_ptr(), out.data_ptr(), Xd.data_ptr(), Pr.data_ptr(), Q.data_ptr(),
                                                h.data_ptr(), a.data_ptr(), A.data_ptr(), alpha.data_ptr(), s_pre.data_ptr(),
                                                amask.data_ptr() if ctx.p > 0 else None, ctx.p, dPr.data_ptr(), dQ.data_ptr(),
                                                dh.data_ptr(), da.data_ptr(), 0, B, n, d, ws.data_ptr(), nb, S()),
                   "digat_xattn_pairwise_bwd")
        # projections: dXd = dOut (residual) + dh W + dP' F1 + dQ F2
        dXd = dOut.clone()
        for g_, w_ in ((dh, W), (dPr, F1), (dQ, F2)):
            _linear_bwd_input(g_.data_ptr(), w_, dXd.data_ptr(), M, d, d, 1, dev, "digat_linear_bwd_input")
        dW, dF1, dF2 = torch.empty_like(W), torch.empty_like(F1), torch.empty_like(F2)
        dbW = torch.empty(d, dtype=torch.float32, device=dev)
        nbw = L().digat_linear_bwd_weight_workspace(M, d, d)
        wsw = _lib.workspace(nbw, dev, "dW")
        for g_, dw_, db_ in ((dh, dW, dbW), (dPr, dF1, None), (dQ, dF2, None)):
            _lib.check(L().digat_linear_bwd_weight(g_.data_ptr(), d, Xd.data_ptr(), d, dw_.data_ptr(), _lib.ptr(db_), M, d, d, 0,
                                                   wsw.data_ptr(), nbw, S()), "digat_linear_bwd_weight")
        dr = torch.empty((B, d), dtype=torch.float32, device=dev)
        _lib.check(L().digat_sum_nodes(dPr.data_ptr(), dr.data_ptr(), B, n, d, S()), "digat_sum_nodes")
        return dXd, None, dr, dW, dbW, dF1, dF2, da.view_as(a), None


# --------------------------------------------------------------------------------------------------
# the three functions of the path, one library call per direction (include/digat_hip.h: digat_*_fwd_train / digat_*_bwd)
# --------------------------------------------------------------------------------------------------
def _save_buffer(nbytes, dev):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


class XattnFused(Function):
    """Eq. 8 layer on already-dropped-out inputs Xd (graphEncoders.py:143-154): K3, the projections, score / softmax /
    attention dropout / aggregation in ``digat_xattn_fwd_train``; the whole backward — pairwise (recomputing
    relu'(K3+K1+K2)), the three projection backwards, K3's — in ``digat_xattn_bwd``."""

    @staticmethod
    def forward(ctx, Xd, A, cvec, W, bW, F1, F2, F3, b3, a, p_alpha, images=None, p_in=0.0, xattn_mode=0):
        """``images``: None, or (forward image, backward image) of (W, F1, F2) from ``split_images`` (either may be None).
        ``p_in`` > 0: ``Xd`` is the layer input BEFORE its input dropout and the library applies drop_{p_in} itself (its backward rides
        in the epilogue of the input-gradient product); 0: the caller has dropped the input.
        ``xattn_mode``: 0 = the library decides per batch on the device between the entry-wise and the all-pairs Eq. 8 kernels (graphs of
        more than 16 nodes), 1 = entry-wise (sparse corpus), 2 = all-pairs — the same function, a choice of speed."""
        Xd, cvec = _f(Xd), _f(cvec)
        B, n, d = Xd.shape
        dev = Xd.device
        p, p_in = float(p_alpha), float(p_in)
        img_f, ctx.img_b = images if images is not None else (None, None)
        ctx.mode = int(xattn_mode)
        seed_in = _seed() if p_in > 0 else 0          # the input dropout's site comes first (graphEncoders.py:145 before :152)
        seed = _seed() if p > 0 else 0
        E = _lib.ext()
        if E is not None:          # the thin torch extension: tensors in, out / save allocated there (the same C entry)
            out, save = E.xattn_fwd_train(Xd, A, cvec, W, bW, F1, F2, F3, b3, a, p, seed, p_in, seed_in, img_f, ctx.mode)
            ctx.save_for_backward(Xd, A, cvec, W, F1, F2, F3, a, out, save)
            ctx.p, ctx.p_in, ctx.sizes = p, p_in, None
            return out
        out = torch.empty_like(Xd)
        nsave, nws = L().digat_xattn_train_save_bytes(B, n, d), L().digat_xattn_train_workspace_bytes(B, n, d)
        save, ws = _save_buffer(nsave, dev), _lib.workspace(nws, dev, "train")
        _lib.check(L().digat_xattn_fwd_train(Xd.data_ptr(), A.data_ptr(), cvec.data_ptr(), W.data_ptr(), bW.data_ptr(), F1.data_ptr(),
                                             F2.data_ptr(), F3.data_ptr(), b3.data_ptr(), a.data_ptr(), out.data_ptr(), p,
                                             seed, p_in, seed_in, B, n, d, save.data_ptr(), nsave, ws.data_ptr(), nws, _lib.ptr(img_f), ctx.mode, S()),
                   "digat_xattn_fwd_train")
        ctx.save_for_backward(Xd, A, cvec, W, F1, F2, F3, a, out, save)
        ctx.p, ctx.p_in, ctx.sizes = p, p_in, (nsave, nws)
        return out

    @staticmethod
    def backward(ctx, dOut):
        Xd, A, cvec, W, F1, F2, F3, a, out, save = ctx.saved_tensors
        B, n, d = Xd.shape
        dev = Xd.device
        dOut = _f(dOut)
        E = _lib.ext()
        if E is not None and ctx.sizes is None:
            dX, dc, dW3, dbW, dF3, db3, da = E.xattn_bwd(dOut, out, Xd, A, cvec, W, F1, F2, F3, a, ctx.p, ctx.p_in, save, ctx.img_b, ctx.mode)
            return dX, None, dc, dW3[0], dbW, dW3[1], dW3[2], dF3, db3, da.view_as(a), None, None, None, None
        nsave, nws = ctx.sizes
        ws = _lib.workspace(nws, dev, "train")
        dX, dc = torch.empty_like(Xd), torch.empty_like(cvec)
        # dW, dF1, dF2 as the three blocks of one [3 d, d] buffer: the library writes its single [3 d, d] weight-gradient product in place
        dW3 = torch.empty((3,) + tuple(W.shape), dtype=torch.float32, device=dev)
        dW, dF1, dF2 = dW3[0], dW3[1], dW3[2]
        dF3 = torch.empty_like(W)
        dbW, db3, da = (torch.empty(d, dtype=torch.float32, device=dev) for _ in range(3))
        _lib.check(L().digat_xattn_bwd(dOut.data_ptr(), out.data_ptr(), Xd.data_ptr(), A.data_ptr(), cvec.data_ptr(), W.data_ptr(),
                                       F1.data_ptr(), F2.data_ptr(), F3.data_ptr(), a.data_ptr(), ctx.p, ctx.p_in, save.data_ptr(), nsave,
                                       dX.data_ptr(), dc.data_ptr(), dW.data_ptr(), dbW.data_ptr(), dF1.data_ptr(), dF2.data_ptr(),
                                       dF3.data_ptr(), db3.data_ptr(), da.data_ptr(), B, n, d, ws.data_ptr(), nws, _lib.ptr(ctx.img_b), ctx.mode, S()),
                   "digat_xattn_bwd")
        return dX, None, dc, dW, dbW, dF1, dF2, dF3, db3, da.view_as(a), None, None, None, None


class GatFused(Function):
    """Vanilla-GAT update layer of the ablation encoders on already-dropped-out inputs (graphEncoders.py:493-503):
    ``digat_gat_fwd_train`` / ``digat_gat_bwd``."""

    @staticmethod
    def forward(ctx, Xd, A, W, bW, a1, a2, p_alpha):
        Xd = _f(Xd)
        B, n, d = Xd.shape
        dev = Xd.device
        out = torch.empty_like(Xd)
        nsave, nws = L().digat_gat_train_save_bytes(B, n, d), L().digat_gat_train_workspace_bytes(B, n, d)
        save, ws = _save_buffer(nsave, dev), _lib.workspace(nws, dev, "train")
        p = float(p_alpha)
        _lib.check(L().digat_gat_fwd_train(Xd.data_ptr(), A.data_ptr(), W.data_ptr(), bW.data_ptr(), a1.data_ptr(), a2.data_ptr(),
                                           out.data_ptr(), p, _seed() if p > 0 else 0, B, n, d, save.data_ptr(), nsave,
                                           ws.data_ptr(), nws, S()), "digat_gat_fwd_train")
        ctx.save_for_backward(Xd, A, W, a1, a2, out, save)
        ctx.p, ctx.sizes = p, (nsave, nws)
        return out

    @staticmethod
    def backward(ctx, dOut):
        Xd, A, W, a1, a2, out, save = ctx.saved_tensors
        B, n, d = Xd.shape
        dev = Xd.device
        dOut = _f(dOut)
        nsave, nws = ctx.sizes
        ws = _lib.workspace(nws, dev, "train")
        dX, dW = torch.empty_like(Xd), torch.empty_like(W)
        dbW, da1, da2 = (torch.empty(d, dtype=torch.float32, device=dev) for _ in range(3))
        _lib.check(L().digat_gat_bwd(dOut.data_ptr(), out.data_ptr(), Xd.data_ptr(), A.data_ptr(), W.data_ptr(), a1.data_ptr(),
                                     a2.data_ptr(), ctx.p, save.data_ptr(), nsave, dX.data_ptr(), dW.data_ptr(), dbW.data_ptr(),
                                     da1.data_ptr(), da2.data_ptr(), B, n, d, ws.data_ptr(), nws, S()), "digat_gat_bwd")
        return dX, None, dW, dbW, da1.view_as(a1), da2.view_as(a2), None


class StepSink:
    """Sums the PARAMETER gradients of the context functions over the depth + 1 calls of one training forward inside the library's
    weight-gradient launches (``accumulate_params``) instead of one element-wise add per weight and call (36 launches of a step).

    The reference uses one candidate_attention / news_graph_W / user_news_K,Q / featureAffine / userAttention for every layer
    (graphEncoders.py:177-187), so autograd adds four gradients per weight.  With a sink, the first backward call of a kind
    allocates the gradient buffers, the later ones add into them in the library, and only the LAST call of the kind returns them to
    autograd (the others return None for the parameters): AccumulateGrad — and DDP's hooks — fire once per weight with the sum.
    One sink per forward, made by ``digat_forward_train`` ONLY (the ablation encoders' ``ablation_forward_train`` passes none: their
    context calls return per-call gradients and autograd adds them): the count of calls is the forward's own, and every call's output
    reaches the loss through the c_n / c_u sums.
    RESTRICTION: the sum is complete only when EVERY context call of the forward takes part in the backward pass.  A partial backward —
    a loss on ``c_n`` alone, ``torch.autograd.grad`` with respect to inputs that prune a context node — ends with a kind incomplete,
    and the pass then fails loudly (``_check``) rather than handing autograd a partial sum under the full name.  Such callers set
    ``encoder.sum_shared_gradients_in_library = False`` (no sink: one gradient per call, summed by autograd; 36 more element-wise
    launches per step).  Under DistributedDataParallel the shared weights' hooks fire once, at the LAST backward call of their kind
    (the layer-0 call, i.e. near the end of the backward pass): their bucket's all-reduce overlaps less of the backward than per-call
    gradients would allow — 2.4 M of the 5.3 M parameters; the per-layer Eq. 8 weights keep their per-layer hooks."""

    def __init__(self):
        self.calls, self.done, self.bufs = {}, {}, {}
        self._armed = False

    def enter(self, kind):
        self.calls[kind] = self.calls.get(kind, 0) + 1

    def begin(self, kind, make):
        """(buffers, accumulate) for this backward call of ``kind``."""
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._check)
        bufs = self.bufs.get(kind)
        if bufs is None:
            bufs = self.bufs[kind] = make()
            return bufs, 0
        return bufs, 1

    def end(self, kind):
        """True when this was the kind's last call: the caller hands the buffers to autograd."""
        self.done[kind] = self.done.get(kind, 0) + 1
        if self.done[kind] == self.calls.get(kind, 0):
            del self.bufs[kind]
            self.done[kind] = 0            # a second backward over a retained graph starts a new sum
            return True
        return False

    def _check(self):
        self._armed = False
        pending = sorted(self.bufs)
        if pending:
            self.bufs.clear(); self.done.clear()
            raise RuntimeError(f"digat_amd.training.StepSink: the backward pass ended before every {pending} call had run (an output of "
                               "the graph encoder did not reach the loss): parameter gradients would be incomplete")


class NewsCtxFused(Function):
    """compute_news_graph_context (graphEncoders.py:109-114), training mode."""

    @staticmethod
    def forward(ctx, X, mask, Kc, Qc, bQc, Wg, bg, p_gate, sink=None, prev=None):
        """``prev``: None, or the context accumulated so far [B,d]: the call returns prev + context (graphEncoders.py:185) from the gate's
        launch instead of a separate element-wise add; its gradient is the output's."""
        X = _f(X)
        prev = _f(prev) if prev is not None else None
        ctx.sink, ctx.has_prev = sink, prev is not None
        if sink is not None:
            sink.enter("news_ctx")
        B, N, d = X.shape
        dev = X.device
        E = _lib.ext()
        if E is not None:
            p = float(p_gate)
            out, save = E.news_ctx_fwd_train(X, mask, Kc, Qc, bQc, Wg, bg, p, _seed() if p > 0 else 0, prev)
            ctx.save_for_backward(X, mask, Kc, Qc, Wg, save)
            ctx.p, ctx.sizes = p, None
            return out
        out = torch.empty((B, d), dtype=torch.float32, device=dev)
        nsave, nws = L().digat_news_ctx_train_save_bytes(B, N, d), L().digat_news_ctx_train_workspace_bytes(B, N, d)
        save, ws = _save_buffer(nsave, dev), _lib.workspace(nws, dev, "train")
        p = float(p_gate)
        _lib.check(L().digat_news_ctx_fwd_train(X.data_ptr(), mask.data_ptr(), Kc.data_ptr(), Qc.data_ptr(), bQc.data_ptr(),
                                                Wg.data_ptr(), bg.data_ptr(), out.data_ptr(), p, _seed() if p > 0 else 0, B, N, d,
                                                save.data_ptr(), nsave, ws.data_ptr(), nws, _lib.ptr(prev), S()), "digat_news_ctx_fwd_train")
        ctx.save_for_backward(X, mask, Kc, Qc, Wg, save)
        ctx.p, ctx.sizes = p, (nsave, nws)
        return out

    @staticmethod
    def backward(ctx, dout):
        X, mask, Kc, Qc, Wg, save = ctx.saved_tensors
        B, N, d = X.shape
        dev = X.device
        dout = _f(dout)

        def make():
            return (torch.empty_like(Kc), torch.empty_like(Qc), torch.empty(d, dtype=torch.float32, device=dev),
                    torch.empty_like(Wg), torch.empty(d, dtype=torch.float32, device=dev))
        sink = ctx.sink
        grads, acc = sink.begin("news_ctx", make) if sink is not None else (make(), 0)
        dKc, dQc, dbQc, dWg, dbg = grads
        E = _lib.ext()
        if E is not None and ctx.sizes is None:
            dX = E.news_ctx_bwd(dout, X, mask, Kc, Qc, Wg, ctx.p, save, list(grads), bool(acc))
            dprev = dout if ctx.has_prev else None
            if sink is not None and not sink.end("news_ctx"):
                return dX, None, None, None, None, None, None, None, None, dprev
            return dX, None, dKc, dQc, dbQc, dWg, dbg, None, None, dprev
        nsave, nws = ctx.sizes
        ws = _lib.workspace(nws, dev, "train")
        dX = torch.empty_like(X)
        _lib.check(L().digat_news_ctx_bwd(dout.data_ptr(), X.data_ptr(), mask.data_ptr(), Kc.data_ptr(), Qc.data_ptr(), Wg.data_ptr(),
                                          ctx.p, save.data_ptr(), nsave, dX.data_ptr(), dKc.data_ptr(), dQc.data_ptr(), dbQc.data_ptr(),
                                          dWg.data_ptr(), dbg.data_ptr(), B, N, d, acc, ws.data_ptr(), nws, S()), "digat_news_ctx_bwd")
        dprev = dout if ctx.has_prev else None
        if sink is not None and not sink.end("news_ctx"):
            return dX, None, None, None, None, None, None, None, None, dprev
        return dX, None, dKc, dQc, dbQc, dWg, dbg, None, None, dprev


class UserCtxFused(Function):
    """compute_user_graph_context (graphEncoders.py:123-134), training mode."""

    @staticmethod
    def forward(ctx, Xu, cat_mask, cat_idx, c_n, Ku, Qu, bQu, Fa, bFa, Kua, Qua, bQua, H, C1, p_topic, sink=None, images=None, prev=None):
        """``images``: None, or (forward image, backward image) of featureAffine.weight from ``split_images``.
        ``prev``: None, or the context accumulated so far [B,d]: the call returns prev + context (graphEncoders.py:186)."""
        Xu, c_n = _f(Xu), _f(c_n)
        prev = _f(prev) if prev is not None else None
        ctx.sink, ctx.has_prev = sink, prev is not None
        img_f, ctx.img_b = images if images is not None else (None, None)
        if sink is not None:
            sink.enter("user_ctx")
        B, U, d = Xu.shape
        dev = Xu.device
        E = _lib.ext()
        if E is not None:
            p = float(p_topic)
            out, save = E.user_ctx_fwd_train(Xu, cat_mask, cat_idx, c_n, Ku, Qu, bQu, Fa, bFa, Kua, Qua, bQua, int(H), int(C1), p,
                                             _seed() if p > 0 else 0, img_f, prev)
            ctx.save_for_backward(Xu, cat_mask, cat_idx, c_n, Ku, Qu, Fa, Kua, Qua, save)
            ctx.p, ctx.sizes, ctx.dims = p, None, (H, C1)
            return out
        out = torch.empty((B, d), dtype=torch.float32, device=dev)
        nsave = L().digat_user_ctx_train_save_bytes(B, U, H, C1, d)
        nws = L().digat_user_ctx_train_workspace_bytes(B, U, H, C1, d)
        save, ws = _save_buffer(nsave, dev), _lib.workspace(nws, dev, "train")
        p = float(p_topic)
        _lib.check(L().digat_user_ctx_fwd_train(Xu.data_ptr(), cat_mask.data_ptr(), cat_idx.data_ptr(), c_n.data_ptr(), Ku.data_ptr(),
                                                Qu.data_ptr(), bQu.data_ptr(), Fa.data_ptr(), bFa.data_ptr(), Kua.data_ptr(),
                                                Qua.data_ptr(), bQua.data_ptr(), out.data_ptr(), p, _seed() if p > 0 else 0,
                                                B, U, H, C1, d, save.data_ptr(), nsave, ws.data_ptr(), nws, _lib.ptr(img_f), _lib.ptr(prev), S()),
                   "digat_user_ctx_fwd_train")
        ctx.save_for_backward(Xu, cat_mask, cat_idx, c_n, Ku, Qu, Fa, Kua, Qua, save)
        ctx.p, ctx.sizes, ctx.dims = p, (nsave, nws), (H, C1)
        return out

    @staticmethod
    def backward(ctx, dout):
        Xu, cat_mask, cat_idx, c_n, Ku, Qu, Fa, Kua, Qua, save = ctx.saved_tensors
        B, U, d = Xu.shape
        H, C1 = ctx.dims
        dev = Xu.device
        dout = _f(dout)

        def make():
            return tuple(torch.empty_like(Ku) for _ in range(5)) + tuple(torch.empty(d, dtype=torch.float32, device=dev) for _ in range(3))
        sink = ctx.sink
        grads, acc = sink.begin("user_ctx", make) if sink is not None else (make(), 0)
        dKu, dQu, dFa, dKua, dQua, dbQu, dbFa, dbQua = grads
        E = _lib.ext()
        if E is not None and ctx.sizes is None:
            dXu, dc = E.user_ctx_bwd(dout, Xu, cat_mask, cat_idx, c_n, Ku, Qu, Fa, Kua, Qua, ctx.p, save, list(grads), bool(acc), int(H), int(C1),
                                     ctx.img_b)
            dprev = dout if ctx.has_prev else None
            if sink is not None and not sink.end("user_ctx"):
                return (dXu, None, None, dc) + (None,) * 13 + (dprev,)
            return dXu, None, None, dc, dKu, dQu, dbQu, dFa, dbFa, dKua, dQua, dbQua, None, None, None, None, None, dprev
        nsave, nws = ctx.sizes
        ws = _lib.workspace(nws, dev, "train")
        dXu, dc = torch.empty_like(Xu), torch.empty_like(c_n)
        _lib.check(L().digat_user_ctx_bwd(dout.data_ptr(), Xu.data_ptr(), cat_mask.data_ptr(), cat_idx.data_ptr(), c_n.data_ptr(),
                                          Ku.data_ptr(), Qu.data_ptr(), Fa.data_ptr(), Kua.data_ptr(), Qua.data_ptr(), ctx.p,
                                          save.data_ptr(), nsave, dXu.data_ptr(), dc.data_ptr(), dKu.data_ptr(), dQu.data_ptr(),
                                          dbQu.data_ptr(), dFa.data_ptr(), dbFa.data_ptr(), dKua.data_ptr(), dQua.data_ptr(),
                                          dbQua.data_ptr(), B, U, H, C1, d, acc, ws.data_ptr(), nws, _lib.ptr(ctx.img_b), S()), "digat_user_ctx_bwd")
        dprev = dout if ctx.has_prev else None
        if sink is not None and not sink.end("user_ctx"):
            return (dXu, None, None, dc) + (None,) * 13 + (dprev,)
        return dXu, None, None, dc, dKu, dQu, dbQu, dFa, dbFa, dKua, dQua, dbQua, None, None, None, None, None, dprev


# --------------------------------------------------------------------------------------------------
# the reference's four functions, training mode
# --------------------------------------------------------------------------------------------------
def split_images(jobs, dev):
    """Every split image of a step's matrix-core weights in ONE launch (``digat_split_jobs``).  ``jobs``: a list of
    ((w0, w1, w2) or (w0,), layout) with nn.Linear weights [rows, cols]; returns one uint8 view per job, in order (all views of one
    buffer: kept alive by whoever holds a view)."""
    L_ = L()
    sizes = []
    for ws, layout in jobs:
        rows, cols = ws[0].shape
        sizes.append((L_.digat_split_job_bytes(rows, cols, layout, len(ws)) + 255) // 256 * 256)
    buf = torch.empty(max(sum(sizes), 256), dtype=torch.uint8, device=dev)
    arr = (_lib.SplitJob * len(jobs))()
    views, off = [], 0
    for k, ((ws, layout), nb) in enumerate(zip(jobs, sizes)):
        rows, cols = ws[0].shape
        view = buf[off:off + nb]
        arr[k] = _lib.SplitJob(ws[0].data_ptr(), ws[1].data_ptr() if len(ws) == 3 else None, ws[2].data_ptr() if len(ws) == 3 else None,
                               rows, cols, layout, 0, view.data_ptr())
        views.append(view)
        off += nb
    _lib.check(L_.digat_split_jobs(arr, len(jobs), S()), "digat_split_jobs")
    return views


def step_images(enc, rows_news, rows_user, rows_topics, dev):
    """The split images of one training forward + backward of ``enc`` (the Eq. 8 projections of every layer and graph, featureAffine),
    for the products that run on the bf16x6 kernel at these row counts: {(graph, layer): (fwd, bwd)}, {"fa": (fwd, bwd)}."""
    d = enc.news_embedding_dim
    jobs, keys = [], []
    for g, rows in (("news", rows_news), ("user", rows_user)):
        if not (_x3_ok(rows, d, d) and g in getattr(enc, "EQ8", ("news", "user"))):
            continue
        for i in range(enc.graph_depth):
            ws = tuple(getattr(enc, f"{g}_graph_attention_{nm}")[i].weight for nm in ("W", "ffn1", "ffn2"))
            jobs += [(ws, 0), (ws, 1)]
            keys += [(g, i, 0), (g, i, 1)]
    if _x3_ok(rows_topics, d, d):
        jobs += [((enc.featureAffine.weight,), 0), ((enc.featureAffine.weight,), 1)]
        keys += [("fa", 0, 0), ("fa", 0, 1)]
    if not jobs or len(jobs) > 24:
        return {}
    views = dict(zip(keys, split_images(jobs, dev)))
    out = {}
    for (g, i, _), _v in views.items():
        out[(g, i)] = (views[(g, i, 0)], views[(g, i, 1)])
    return out


def news_graph_context(enc, X, mask_bytes, p, training=True, sink=None, prev=None):
    ca, g = enc.candidate_attention, enc.news_graph_W
    return NewsCtxFused.apply(X, mask_bytes, ca.K.weight, ca.Q.weight, ca.Q.bias, g.weight, g.bias, p / 2 if training else 0.0, sink, prev)


def user_graph_context(enc, Xu, cat_mask_bytes, cat_idx, c_n, p, training=True, sink=None, images=None, prev=None):
    ua, fa = enc.userAttention, enc.featureAffine
    return UserCtxFused.apply(Xu, cat_mask_bytes, cat_idx, c_n, enc.user_news_K.weight, enc.user_news_Q.weight, enc.user_news_Q.bias,
                              fa.weight, fa.bias, ua.K.weight, ua.Q.weight, ua.Q.bias, enc.max_history_num, enc.category_num,
                              p if training else 0.0, sink, images, prev)


def graph_embeddings(enc, g, i, X, A_bytes, ctx_vec, p, training=True, images=None):
    # the layer's input dropout (p/2, :145 / :165) is applied inside the library call, its backward inside the input-gradient product
    F3 = getattr(enc, f"{g}_graph_attention_ffn3")[i]
    W = getattr(enc, f"{g}_graph_attention_W")[i]
    return XattnFused.apply(X, A_bytes, ctx_vec, W.weight, W.bias,
                            getattr(enc, f"{g}_graph_attention_ffn1")[i].weight,
                            getattr(enc, f"{g}_graph_attention_ffn2")[i].weight, F3.weight, F3.bias,
                            getattr(enc, f"{g}_graph_attention_a")[i].weight, p if training else 0.0, images, p / 2 if training else 0.0,
                            {"auto": 0, "sparse": 1, "dense": 2}[enc.resolved_xattn_mode(g)] if hasattr(enc, "resolved_xattn_mode") else 0)


def gat_embeddings(enc, g, i, X, A_bytes, p, training=True):
    """Vanilla-GAT update layer (graphEncoders.py:493-503 / :509-519), training mode."""
    Xd = dropout(X, p / 2, training)
    W = getattr(enc, f"{g}_graph_attention_W")[i]
    return GatFused.apply(Xd, A_bytes, W.weight, W.bias, getattr(enc, f"{g}_graph_attention_a1")[i].weight,
                          getattr(enc, f"{g}_graph_attention_a2")[i].weight, p if training else 0.0)


def _train_inputs(enc, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph, user_category_mask,
                  user_category_indices):
    p = enc.dropout_rate
    Xn = _lib.f32(news_graph_embeddings)
    ue = _lib.f32(user_news_embedding)
    _lib.require_device(Xn, news_graph, news_graph_mask, ue, user_graph, user_category_mask, user_category_indices)
    An, Mn = _lib.as_bytes(news_graph), _lib.as_bytes(news_graph_mask)
    Au, cm = _lib.as_bytes(user_graph), _lib.as_bytes(user_category_mask)
    ci = user_category_indices.to(torch.int64).contiguous()
    topic = dropout(enc.topic_node_embedding.unsqueeze(0).expand(Xn.shape[0], -1, -1), p / 2)
    return p, Xn, An, Mn, torch.cat([ue, topic], dim=1), Au, cm, ci


def ablation_forward_train(enc, *inputs):
    """Training-mode forward of the five ablation encoders (graphEncoders.py:264-271 wo_SA, :374-382 Seq_SA, :523-535
    wo_interaction, :672-683 News_graph_wo_inter, :817-829 User_graph_wo_inter), dropout placed as in the reference: p/2 on
    every layer input, the topic nodes and the news-context gate; p on every alpha and the pooled topics."""
    p, Xn, An, Mn, Xu, Au, cm, ci = _train_inputs(enc, *inputs)
    kind = type(enc).__name__

    def layer(g, i, X, A, cvec):
        return graph_embeddings(enc, g, i, X, A, cvec, p) if g in enc.EQ8 else gat_embeddings(enc, g, i, X, A, p)

    if kind == "wo_SA":
        c = Xn[:, 0].contiguous()
        for i in range(enc.graph_depth):
            Xu = layer("user", i, Xu, Au, c)
        return c, user_graph_context(enc, Xu, cm, ci, c, p)
    c_n = news_graph_context(enc, Xn, Mn, p)
    c_u = user_graph_context(enc, Xu, cm, ci, c_n, p)
    for i in range(enc.graph_depth):
        if kind == "Seq_SA":                      # the news side is a sequence: pooled once, never updated
            Xu = layer("user", i, Xu, Au, c_n)
        else:
            Xn_next = layer("news", i, Xn, An, c_u)
            Xu = layer("user", i, Xu, Au, c_n)
            Xn = Xn_next
            c_n = c_n + news_graph_context(enc, Xn, Mn, p)
        c_u = c_u + user_graph_context(enc, Xu, cm, ci, c_n, p)
    return c_n, c_u


def digat_forward_train(enc, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                        user_category_mask, user_category_indices):
    """graphEncoders.py:177-187 with dropout live (p, p, p/2 as in :22-24).

    The shared context weights' gradients are summed inside the library (``StepSink``): BOTH returned contexts must reach the loss of
    the backward pass (they do in ``Model.forward``: logits = user_ctx . news_ctx); for a partial backward set
    ``enc.sum_shared_gradients_in_library = False`` first (see StepSink)."""
    p, Xn, An, Mn, Xu, Au, cm, ci = _train_inputs(enc, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding,
                                                  user_graph, user_category_mask, user_category_indices)
    sink = StepSink() if getattr(enc, "sum_shared_gradients_in_library", True) else None
    # every split image the step's matrix-core products read — forward and backward, each layer and graph, featureAffine — in ONE
    # launch now (round 6: each call split its own before, 20 launches per step); the weights do not change before the backward
    img = step_images(enc, Xn.shape[0] * Xn.shape[1], Xu.shape[0] * Xu.shape[1], Xu.shape[0] * enc.category_num, Xn.device) \
        if getattr(enc, "split_weights_once_per_step", True) else {}
    fa = img.get(("fa", 0))
    c_n = news_graph_context(enc, Xn, Mn, p, sink=sink)
    c_u = user_graph_context(enc, Xu, cm, ci, c_n, p, sink=sink, images=fa)
    for i in range(enc.graph_depth):
        Xn_next = graph_embeddings(enc, "news", i, Xn, An, c_u, p, images=img.get(("news", i)))
        Xu_next = graph_embeddings(enc, "user", i, Xu, Au, c_n, p, images=img.get(("user", i)))
        Xn, Xu = Xn_next, Xu_next
        c_n = news_graph_context(enc, Xn, Mn, p, sink=sink, prev=c_n)                   # c_n + ... (:185), added where the context is produced
        c_u = user_graph_context(enc, Xu, cm, ci, c_n, p, sink=sink, images=fa, prev=c_u)   # c_u + ... (:186), of the UPDATED c_n
    return c_n, c_u
