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
#include "common.h"

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
                                    lnb, R, C, dtype, stream);
    if (st) {
        if (batch) dg_linear_wgrad_batch_end(stream);
        return st;
    }
    workspace = static_cast<char*>(workspace) + lnb;
    workspace_bytes -= lnb;
    // dh = (dz @ W2) masked by the forward's ReLU bits
    st = dg_row_gemm(dz, w2_dgrad_packed, dh, R, C, H, nullptr, 0, nullptr, relu_bits, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, 0.f, dtype, stream);
    if (!st && dx)   // dx = dz + dh @ W1 (residual path folded into the epilogue)
        st = dg_row_gemm(dh, w1_dgrad_packed, dx, R, H, C, nullptr, 0, nullptr, nullptr, dz, nullptr, nullptr,
                         nullptr, nullptr, nullptr, 0.f, dtype, stream);
    if (st) {
        if (batch) dg_linear_wgrad_batch_end(stream);
        return st;
    }
    if (batch) {      // two split-K kernels into separate parts of the workspace
        const size_t w2b = (dg_linear_wgrad_workspace_bytes(R, C, H) + 255) / 256 * 256;
        st = dg_linear_wgrad(dz, nullptr, h, dw2, db2, workspace, w2b, R, C, H, dtype, stream);
        if (!st)
            st = dg_linear_wgrad(dh, nullptr, x, dw1, db1, static_cast<char*>(workspace) + w2b, workspace_bytes - w2b, R, H, C,
                                 dtype, stream);
        const int st2 = dg_linear_wgrad_batch_end(stream);
        return st ? st : st2;
    }
    if (dw2) {
        st = dg_linear_wgrad(dz, nullptr, h, dw2, db2, workspace, workspace_bytes, R, C, H, dtype, stream);
        if (st) return st;
    }
    if (dw1) {
        st = dg_linear_wgrad(dh, nullptr, x, dw1, db1, workspace, workspace_bytes, R, H, C, dtype, stream);
        if (st) return st;
    }
    return 0;
}
