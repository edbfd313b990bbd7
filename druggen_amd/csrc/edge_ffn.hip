// dg_edge_ffn_ln_fwd / dg_edge_ffn_ln_bwd: the edge (and node) feed-forward half of an
// Encoder_Block -- reference src/model/layers.py:191-192 with MLP.forward (:40-54):
//
//     y = LayerNorm( x + fc2( relu( fc1(x) ) ) ) * gamma + beta
//
// as ONE call per direction.  The entry points sequence the row-GEMM / LayerNorm / weight-gradient
// kernels of this library on the caller's stream (every elementwise op lives in a GEMM epilogue):
//   fwd : h  = relu(x W1^T + b1)            [dg_row_gemm, ReLU + packed-mask epilogue]
//         y  = LN(x + h W2^T + b2)          [dg_row_gemm, bias + residual + LayerNorm epilogue]
//   bwd : dz = LN'(dy)                      [dg_ln_residual_bwd on the saved pre-LN sum]
//         dh = (dz W2) * (h > 0)            [dg_row_gemm, packed ReLU mask in the epilogue]
//         dx = dz + dh W1                   [dg_row_gemm, residual epilogue]
//         dW2, db2 = dz^T h, sum dz ; dW1, db1 = dh^T x, sum dh      [dg_linear_wgrad]
#include "bf16.h"

using namespace dg;

extern "C" size_t dg_edge_ffn_ln_workspace_bytes(int64_t R, int C, int H) {
    const size_t ln = dg_ln_workspace_bytes(R, C);
    const size_t w1 = dg_linear_wgrad_workspace_bytes(R, H, C), w2 = dg_linear_wgrad_workspace_bytes(R, C, H);
    if (!ln || !w1 || !w2) return 0;
    // LayerNorm partials and both weight gradients' partial sums live side by side (one reduce launch for all three)
    return ((ln + 255) / 256 + (w1 + 255) / 256 + (w2 + 255) / 256) * 256;
}

extern "C" int dg_edge_ffn_ln_fwd(const void* x, const void* w1_packed, const float* b1, const void* w2_packed,
                                  const float* b2, const float* gamma, const float* beta, void* y, void* h,
                                  unsigned* relu_bits, void* pre_ln, float* mean, float* rstd, int64_t R, int C,
                                  int H, float eps, int dtype, dg_stream_t stream) {
    if (!x || !w1_packed || !b1 || !w2_packed || !b2 || !gamma || !beta || !y || !h || !mean || !rstd)
        return fail(DG_E_ARG, "dg_edge_ffn_ln_fwd: null pointer");
    if (C != 128 || H != 384) return fail(DG_E_SHAPE, "dg_edge_ffn_ln_fwd: needs dim 128, hidden 384 (got %d, %d)", C, H);
    int st = dg_row_gemm(x, w1_packed, h, R, C, H, b1, 1, relu_bits, nullptr, nullptr, nullptr, nullptr, nullptr,
                         nullptr, nullptr, 0.f, dtype, stream);
    if (st) return st;
    return dg_row_gemm(h, w2_packed, y, R, H, C, b2, 0, nullptr, nullptr, x, gamma, beta, mean, rstd, pre_ln, eps,
                       dtype, stream);
}

extern "C" int dg_edge_ffn_ln_bwd(const void* x, const void* h, const unsigned* relu_bits, const void* pre_ln,
                                  const float* mean, const float* rstd, const float* gamma,
                                  const void* w1_dgrad_packed, const void* w2_dgrad_packed, const void* dy,
                                  const void* dz_add, void* dz, void* dh, void* dx, float* dgamma, float* dbeta, float* dw1,
                                  float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes,
                                  int64_t R, int C, int H, int dtype, dg_stream_t stream) {
    if (!x || !h || !relu_bits || !pre_ln || !mean || !rstd || !gamma || !w1_dgrad_packed || !w2_dgrad_packed ||
        !dz || !dh || !workspace)
        return fail(DG_E_ARG, "dg_edge_ffn_ln_bwd: null pointer");
    if (!dy && dz_add != dz)
        return fail(DG_E_ARG, "dg_edge_ffn_ln_bwd: dy == NULL means dz already holds the LayerNorm input gradient (pass it as dz_add too)");
    if (C != 128 || H != 384) return fail(DG_E_SHAPE, "dg_edge_ffn_ln_bwd: needs dim 128, hidden 384 (got %d, %d)", C, H);
    if (workspace_bytes < dg_edge_ffn_ln_workspace_bytes(R, C, H))
        return fail(DG_E_WORKSPACE, "dg_edge_ffn_ln_bwd: workspace too small");
    int st = 0;
    const size_t lnb = (dg_ln_workspace_bytes(R, C) + 255) / 256 * 256;
    const bool batch = dw2 && dw1;      // LayerNorm's dgamma / dbeta and both weight gradients: one reduce launch
    if (batch) dg_linear_wgrad_batch_begin();
    // dy == NULL: the LayerNorm backward ran in the epilogue of the GEMM that produced its output gradient
    // (dg_row_gemm_ln_bwd) and dz holds the result; dgamma / dbeta are not touched
    if (dy)
        st = dg_ln_residual_bwd_add(pre_ln, nullptr, gamma, mean, rstd, dy, dz_add, dz, dgamma, dbeta, workspace,
                                    lnb, R, C, act_dtype(dtype), stream);
    if (st) {
        if (batch) dg_linear_wgrad_batch_end(stream);
        return st;
    }
    workspace = static_cast<char*>(workspace) + lnb;
    workspace_bytes -= lnb;
    // dh = (dz @ W2) masked by the forward's ReLU bits
    const int dt_h = ffn_h_dtype(dtype), dt_dh = ffn_dh_dtype(dtype);      // storage of h / of dh (DG_DTYPE_F32_DH16 / _DH24 differ)
    st = dg_row_gemm(dz, w2_dgrad_packed, dh, R, C, H, nullptr, 0, nullptr, relu_bits, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, 0.f, dt_dh, stream);
    if (!st && dx)   // dx = dz + dh @ W1 (residual path folded into the epilogue)
        st = dg_row_gemm(dh, w1_dgrad_packed, dx, R, H, C, nullptr, 0, nullptr, nullptr, dz, nullptr, nullptr,
                         nullptr, nullptr, nullptr, 0.f, dt_dh, stream);
    if (st) {
        if (batch) dg_linear_wgrad_batch_end(stream);
        return st;
    }
    if (batch) {      // two split-K kernels into separate parts of the workspace
        const size_t w2b = (dg_linear_wgrad_workspace_bytes(R, C, H) + 255) / 256 * 256;
        st = dg_linear_wgrad(dz, nullptr, h, dw2, db2, workspace, w2b, R, C, H, dt_h, stream);
        if (!st)
            st = dg_linear_wgrad(dh, nullptr, x, dw1, db1, static_cast<char*>(workspace) + w2b, workspace_bytes - w2b, R, H, C,
                                 dt_dh, stream);
        const int st2 = dg_linear_wgrad_batch_end(stream);
        return st ? st : st2;
    }
    if (dw2) {
        st = dg_linear_wgrad(dz, nullptr, h, dw2, db2, workspace, workspace_bytes, R, C, H, dt_h, stream);
        if (st) return st;
    }
    if (dw1) {
        st = dg_linear_wgrad(dh, nullptr, x, dw1, db1, workspace, workspace_bytes, R, H, C, dt_dh, stream);
        if (st) return st;
    }
    return 0;
}

// ---- node + edge feed-forward of a block in one call (riding launches, pair.h) ---------------------------------------
namespace {

int ffn_fwd_check(const dg_ffn_fwd_args* a, const char* who) {
    if (!a || !a->x || !a->w1_packed || !a->b1 || !a->w2_packed || !a->b2 || !a->gamma || !a->beta || !a->y || !a->h ||
        !a->mean || !a->rstd)
        return fail(DG_E_ARG, "dg_edge_ffn_ln_fwd_pair: null pointer (%s)", who);
    return 0;
}

// one step of dg_edge_ffn_ln_bwd's sequence for one problem; the workspace is laid out as in that function
int ffn_bwd_phase(const dg_ffn_bwd_args& a, int phase, int C, int H, int dtype, dg_stream_t stream) {
    const int64_t R = a.R;
    const size_t lnb = (dg_ln_workspace_bytes(R, C) + 255) / 256 * 256;
    const size_t w2b = (dg_linear_wgrad_workspace_bytes(R, C, H) + 255) / 256 * 256;
    char* ws = static_cast<char*>(a.workspace);
    switch (phase) {
        case 0:      // dz = LN'(dy) (+ dz_add); dy == NULL: dz already holds it
            if (!a.dy) return 0;
            return dg_ln_residual_bwd_add(a.pre_ln, nullptr, a.gamma, a.mean, a.rstd, a.dy, a.dz_add, a.dz, a.dgamma, a.dbeta, ws,
                                          lnb, R, C, act_dtype(dtype), stream);
        case 1:      // dh = (dz W2) masked by the forward's ReLU bits
            return dg_row_gemm(a.dz, a.w2_dgrad_packed, a.dh, R, C, H, nullptr, 0, nullptr, a.relu_bits, nullptr, nullptr,
                               nullptr, nullptr, nullptr, nullptr, 0.f, ffn_dh_dtype(dtype), stream);
        case 2:      // dx = dz + dh W1
            if (!a.dx) return 0;
            return dg_row_gemm(a.dh, a.w1_dgrad_packed, a.dx, R, H, C, nullptr, 0, nullptr, nullptr, a.dz, nullptr, nullptr,
                               nullptr, nullptr, nullptr, 0.f, ffn_dh_dtype(dtype), stream);
        case 3:      // dW2, db2 = dz^T h, sum dz
            if (!a.dw2) return 0;
            return dg_linear_wgrad(a.dz, nullptr, a.h, a.dw2, a.db2, ws + lnb, w2b, R, C, H, ffn_h_dtype(dtype), stream);
        default:     // dW1, db1 = dh^T x, sum dh
            if (!a.dw1) return 0;
            return dg_linear_wgrad(a.dh, nullptr, a.x, a.dw1, a.db1, ws + lnb + w2b, a.workspace_bytes - lnb - w2b, R, H, C,
                                   ffn_dh_dtype(dtype), stream);
    }
}

}  // namespace

extern "C" int dg_edge_ffn_ln_fwd_pair(const dg_ffn_fwd_args* node, const dg_ffn_fwd_args* edge, int C, int H, int dtype,
                                       dg_stream_t stream) {
    if (int st = ffn_fwd_check(node, "node")) return st;
    if (int st = ffn_fwd_check(edge, "edge")) return st;
    if (C != 128 || H != 384) return fail(DG_E_SHAPE, "dg_edge_ffn_ln_fwd_pair: needs dim 128, hidden 384 (got %d, %d)", C, H);
    const dg_ffn_fwd_args* ps[2] = {node, edge};
    dg_launch_pair_begin();
    int st = 0;
    for (int i = 0; i < 2 && !st; ++i)      // h = relu(x W1^T + b1)
        st = dg_row_gemm(ps[i]->x, ps[i]->w1_packed, ps[i]->h, ps[i]->R, C, H, ps[i]->b1, 1, ps[i]->relu_bits, nullptr, nullptr,
                         nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, dtype, stream);
    for (int i = 0; i < 2 && !st; ++i)      // y = LN(x + h W2^T + b2)
        st = dg_row_gemm(ps[i]->h, ps[i]->w2_packed, ps[i]->y, ps[i]->R, H, C, ps[i]->b2, 0, nullptr, nullptr, ps[i]->x,
                         ps[i]->gamma, ps[i]->beta, ps[i]->mean, ps[i]->rstd, ps[i]->pre_ln, ps[i]->eps, dtype, stream);
    const int st2 = dg_launch_pair_end(stream);
    return st ? st : st2;
}

extern "C" int dg_edge_ffn_ln_bwd_pair(const dg_ffn_bwd_args* node, const dg_ffn_bwd_args* edge, int C, int H, int dtype,
                                       dg_stream_t stream) {
    const dg_ffn_bwd_args* ps[2] = {node, edge};
    for (const dg_ffn_bwd_args* a : ps) {
        if (!a || !a->x || !a->h || !a->relu_bits || !a->pre_ln || !a->mean || !a->rstd || !a->gamma || !a->w1_dgrad_packed ||
            !a->w2_dgrad_packed || !a->dz || !a->dh || !a->workspace)
            return fail(DG_E_ARG, "dg_edge_ffn_ln_bwd_pair: null pointer");
        if (!a->dy && a->dz_add != a->dz)
            return fail(DG_E_ARG, "dg_edge_ffn_ln_bwd_pair: dy == NULL means dz already holds the LayerNorm input gradient (pass it as dz_add too)");
        if (a->workspace_bytes < dg_edge_ffn_ln_workspace_bytes(a->R, C, H))
            return fail(DG_E_WORKSPACE, "dg_edge_ffn_ln_bwd_pair: workspace too small");
    }
    if (C != 128 || H != 384) return fail(DG_E_SHAPE, "dg_edge_ffn_ln_bwd_pair: needs dim 128, hidden 384 (got %d, %d)", C, H);
    // every reduction of the call (LayerNorm dgamma / dbeta of up to two problems, up to four weight gradients) in one launch
    dg_linear_wgrad_batch_begin();
    dg_launch_pair_begin();
    int st = 0;
    for (int phase = 0; phase < 5 && !st; ++phase)
        for (int i = 0; i < 2 && !st; ++i) st = ffn_bwd_phase(*ps[i], phase, C, H, dtype, stream);
    const int st2 = dg_launch_pair_end(stream);
    const int st3 = dg_linear_wgrad_batch_end(stream);
    return st ? st : (st2 ? st2 : st3);
}
