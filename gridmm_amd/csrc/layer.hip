// gridmm_xattn_layer_fwd: one GraphLXRTXLayer (reference map_nav_src/models/vilmodel.py:399-414 with graph_sprels = None:
// BertXAttention over a context -> BertAttention over the tokens themselves -> BertIntermediate / BertOutput) as ONE
// C call -- the entry point SURVEY.md §8b names for the cross-modal layers.  It owns no arithmetic of its own: it
// sequences the library's kernels (6 plane GEMMs, 2 attention_rows, 3 LayerNorms) on the caller's stream through a
// caller-provided workspace, so a C / C++ host (or the Python module, which uses it for its inference path) drives a
// whole layer without touching intermediate tensors.  Results are bit-identical to issuing the eleven calls one by one
// (pinned by tests/test_hip_kernels.py).
#include <cstdlib>

#include "common.h"

namespace {
inline size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

// One plane GEMM of the layer: the tiled weight planes when the layer carries them and the shape's tile reads them, else the
// row-major ones.
inline int lin(const gridmm_linear_t& W, const void* A_hi, const void* A_lo, int lda, const float* R, int ldr, float* C, int ldc,
               void* C_hi, void* C_lo, int ldp, int M, int act, gridmm_stream_t stream) {
  if (W.wt_hi && W.wt_lo) {
    const int rc = gridmm_linear_planes_map(A_hi, A_lo, lda, 0, 0, W.wt_hi, W.wt_lo, W.Kp, GRIDMM_W_TILED, W.bias, R, ldr, C, ldc,
                                            C_hi, C_lo, ldp, M, W.N, W.K, act, stream);
    if (rc != GRIDMM_EUNSUPPORTED) return rc;
  }
  return gridmm_linear_planes(A_hi, A_lo, lda, W.w_hi, W.w_lo, W.Kp, W.bias, R, ldr, C, ldc, C_hi, C_lo, ldp, M, W.N, W.K, act,
                              stream);
}
}

extern "C" size_t gridmm_xattn_layer_workspace(int B, int Sq, int H, int I) {
  const size_t M = (size_t)B * Sq;
  // planes: q (H), attention context (H), x-attn out (H), qkv (3H), self context (H), self out (H), ffn (I): hi + lo
  // fp32 : pre-LN sums (H) x1 (reused), post-LN a (H), post-LN b (H)
  return a256(M * H * 4) * 6 + a256(M * 3 * H * 4) + a256(M * (size_t)I * 4) + a256(M * H * 4) * 3 + 4096;
}

extern "C" int gridmm_xattn_layer_fwd(const gridmm_xlayer_t* L, const float* X, const void* X_hi, const void* X_lo,
                                      const void* KV_hi, const void* KV_lo, int64_t kv_bs, int kv_rs, int k_col, int v_col,
                                      int Sk1, const void* KV2_hi, const void* KV2_lo, int64_t kv2_bs, int kv2_rs,
                                      int k2_col, int v2_col, const uint8_t* ctx_mask, int ctx_mask_bs, const uint8_t* self_mask, int self_mask_bs,
                                      float* Y, void* Y_hi, void* Y_lo, int y_p_rpb, int64_t y_p_bs, void* workspace,
                                      size_t workspace_bytes, int B, int Sq, int Sk, int heads,
                                      gridmm_stream_t stream) {
  if (!L || !X || !X_hi || !X_lo || !KV_hi || !KV_lo || !workspace || B <= 0 || Sq <= 0 || Sk <= 0 || heads <= 0)
    return GRIDMM_EINVAL;
  if (!KV2_hi) Sk1 = Sk;                                     // one context buffer
  if (Sk1 < 0 || Sk1 > Sk || (Sk1 < Sk && !KV2_lo)) return GRIDMM_EINVAL;
  const int H = L->xq.N, I = L->ffn_i.N, M = B * Sq;
  if (H != heads * 64 || L->xq.K != H || L->xo.N != H || L->xo.K != H || L->sqkv.N != 3 * H || L->sqkv.K != H ||
      L->so.N != H || L->so.K != H || L->ffn_i.K != H || L->ffn_o.N != H || L->ffn_o.K != I || (!Y && !Y_hi))
    return GRIDMM_EINVAL;
  if (workspace_bytes < gridmm_xattn_layer_workspace(B, Sq, H, I)) return GRIDMM_EINVAL;
  char* w = (char*)workspace;
  auto take = [&](size_t bytes) { char* p = w; w += a256(bytes); return p; };
  const size_t pl = (size_t)M * H * 2;                       // one bf16 plane of (M, H)
  unsigned short *q_hi = (unsigned short*)take(2 * pl), *q_lo = q_hi + (size_t)M * H;
  unsigned short *c_hi = (unsigned short*)take(2 * pl), *c_lo = c_hi + (size_t)M * H;
  unsigned short *a_hi = (unsigned short*)take(2 * pl), *a_lo = a_hi + (size_t)M * H;
  unsigned short *qkv_hi = (unsigned short*)take(6 * pl), *qkv_lo = qkv_hi + (size_t)M * 3 * H;
  unsigned short *s_hi = (unsigned short*)take(2 * pl), *s_lo = s_hi + (size_t)M * H;
  unsigned short *b_hi = (unsigned short*)take(2 * pl), *b_lo = b_hi + (size_t)M * H;
  unsigned short *f_hi = (unsigned short*)take((size_t)M * I * 4), *f_lo = f_hi + (size_t)M * I;
  float* h = (float*)take((size_t)M * H * 4);
  float* a = (float*)take((size_t)M * H * 4);
  float* bb = (float*)take((size_t)M * H * 4);
  const float scale = 0.125f;                                // 1 / sqrt(64)
  int rc;
#define GRIDMM_TRY(call) do { rc = (call); if (rc != GRIDMM_OK) return rc; } while (0)
  // ---- cross attention over the context (vilmodel.py:370-379): q = query(x); a = LN(dense(attn) + x)
  GRIDMM_TRY(lin(L->xq, X_hi, X_lo, H, nullptr, 0, nullptr, 0, q_hi, q_lo, H, M, GRIDMM_ACT_NONE, stream));
  const unsigned short *k2h = (const unsigned short*)KV2_hi, *k2l = (const unsigned short*)KV2_lo;
  GRIDMM_TRY(gridmm_attention_rows_seg(q_hi, q_lo, (int64_t)Sq * H, H, (const unsigned short*)KV_hi + k_col,
                                       (const unsigned short*)KV_lo + k_col, kv_bs, kv_rs, (const unsigned short*)KV_hi + v_col,
                                       (const unsigned short*)KV_lo + v_col, kv_bs, kv_rs, Sk1, k2h ? k2h + k2_col : nullptr,
                                       k2h ? k2l + k2_col : nullptr, k2h ? k2h + v2_col : nullptr, k2h ? k2l + v2_col : nullptr,
                                       kv2_bs, kv2_rs, ctx_mask, ctx_mask_bs, nullptr, 0, 0, c_hi, c_lo, (int64_t)Sq * H, H, B,
                                       heads, Sq, Sk, scale, stream));
  // dense + residual + LayerNorm: GEMM (pre-LayerNorm sums in h), then the LayerNorm launch.  (Two fused forms -- a
  // rendezvous of the row block's column tiles inside the GEMM, and a LayerNorm deferred into its consumers -- were built in
  // round 4, measured slower than the two launches and removed in round 5: profiles/r4_layernorm_fusion_experiments.txt.)
  auto dense_ln = [&](const gridmm_linear_t& W, const gridmm_ln_t& ln, const void* in_hi, const void* in_lo, int K,
                      const float* res, float* out, void* out_hi, void* out_lo, int p_rpb, int64_t p_bs) -> int {
    const int r = lin(W, in_hi, in_lo, K, res, H, h, H, nullptr, nullptr, 0, M, GRIDMM_ACT_NONE, stream);
    if (r != GRIDMM_OK) return r;
    return gridmm_layernorm_map(h, H, nullptr, 0, ln.gamma, ln.beta, ln.eps, out, H, nullptr, 0, nullptr, nullptr, out_hi,
                                out_lo, H, p_rpb, p_bs, M, H, stream);
  };
  GRIDMM_TRY(dense_ln(L->xo, L->x_ln, c_hi, c_lo, H, X, a, a_hi, a_lo, 0, 0));
  // ---- self attention (vilmodel.py:172-182)
  GRIDMM_TRY(lin(L->sqkv, a_hi, a_lo, H, nullptr, 0, nullptr, 0, qkv_hi, qkv_lo, 3 * H, M, GRIDMM_ACT_NONE, stream));
  GRIDMM_TRY(gridmm_attention_rows(qkv_hi, qkv_lo, (int64_t)Sq * 3 * H, 3 * H, qkv_hi + H, qkv_lo + H, (int64_t)Sq * 3 * H,
                                   3 * H, qkv_hi + 2 * H, qkv_lo + 2 * H, (int64_t)Sq * 3 * H, 3 * H, self_mask, self_mask_bs,
                                   nullptr, 0, 0, s_hi, s_lo, (int64_t)Sq * H, H, B, heads, Sq, Sq, scale, stream));
  GRIDMM_TRY(dense_ln(L->so, L->s_ln, s_hi, s_lo, H, a, bb, b_hi, b_lo, 0, 0));
  // ---- feed forward (vilmodel.py:184-209): LN(dense(gelu(dense(b))) + b)
  GRIDMM_TRY(lin(L->ffn_i, b_hi, b_lo, H, nullptr, 0, nullptr, 0, f_hi, f_lo, I, M, GRIDMM_ACT_GELU, stream));
  GRIDMM_TRY(dense_ln(L->ffn_o, L->f_ln, f_hi, f_lo, I, bb, Y, Y_hi, Y_lo, y_p_rpb, y_p_bs));
#undef GRIDMM_TRY
  return GRIDMM_OK;
}
