// gridmm_xattn_layer_train_fwd / gridmm_xattn_layer_bwd: one GraphLXRTXLayer of the DIFFERENTIABLE path (fine-tune loop
// map_nav_src/r2r/agent_base.py:164-211 through models/vilmodel.py:399-414; pre-training pretrain_src/model/
// vilmodel.py:404-415) -- forward that keeps what the backward needs in one caller-provided block, and the whole backward of
// the layer (3 LayerNorm backwards, 6 Linear backwards = 18 GEMMs + 6 transposing splits, 2 attention backwards, GELU') as
// ONE C call: SURVEY.md 8b's gridmm_xattn_layer_bwd.  Like gridmm_xattn_layer_fwd these functions own no arithmetic: they
// sequence the library's kernels on the caller's stream, in exactly the order (and with the same tile choices) as the
// op-by-op autograd path of gridmm_amd/autograd.py, so outputs and gradients are bit-identical to it
// (tests/test_hip_layer_train.py).  Hidden-state dropout rides on the LayerNorm kernels, attention-probability dropout
// inside the attention kernels; the backward regenerates both masks from the seeds.
#include "common.h"

namespace {
inline size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int mp32(int M) { return (M + 31) / 32 * 32; }

// Layout of the saved block (forward writes, backward reads).  T planes: [C][Mp] hi then lo.
struct Saved {
  unsigned short *xT, *cT, *a1T, *c2T, *a2T, *gT;   // transposed planes of every Linear input
  float *q, *c, *lse_x, *h1, *a1, *qkv, *c2, *lse_s, *h2, *a2, *f1, *h3, *qkv_shift;
  size_t bytes;
};

Saved carve_saved(char* base, int B, int Sq, int H, int I, int heads) {
  const size_t M = (size_t)B * Sq, Mp = mp32((int)M), Sqp = (Sq + 15) / 16 * 16;
  char* w = base;
  auto take = [&](size_t bytes) { char* p = w; w += a256(bytes); return p; };
  Saved s;
  s.xT = (unsigned short*)take((size_t)H * Mp * 4);
  s.cT = (unsigned short*)take((size_t)H * Mp * 4);
  s.a1T = (unsigned short*)take((size_t)H * Mp * 4);
  s.c2T = (unsigned short*)take((size_t)H * Mp * 4);
  s.a2T = (unsigned short*)take((size_t)H * Mp * 4);
  s.gT = (unsigned short*)take((size_t)I * Mp * 4);
  s.q = (float*)take(M * H * 4);
  s.c = (float*)take(M * H * 4);
  s.lse_x = (float*)take((size_t)B * heads * Sqp * 4);
  s.h1 = (float*)take(M * H * 4);
  s.a1 = (float*)take(M * H * 4);
  s.qkv = (float*)take(M * 3 * H * 4);
  s.c2 = (float*)take(M * H * 4);
  s.lse_s = (float*)take((size_t)B * heads * Sqp * 4);
  s.h2 = (float*)take(M * H * 4);
  s.a2 = (float*)take(M * H * 4);
  s.f1 = (float*)take(M * (size_t)I * 4);
  s.h3 = (float*)take(M * H * 4);
  s.qkv_shift = (float*)take((size_t)B * 3 * H * 4);   // row 0 of every episode's q | k | v projection (shifted K / V planes)
  s.bytes = (size_t)(w - base);
  return s;
}

bool shapes_ok(const gridmm_xlayer_train_t* L, int H, int I, bool cross = true) {
  const gridmm_linear_train_t* ls[6] = {&L->xq, &L->xo, &L->sqkv, &L->so, &L->ffn_i, &L->ffn_o};
  const int N[6] = {H, H, 3 * H, H, I, H}, K[6] = {H, H, H, H, H, I};
  for (int i = cross ? 0 : 2; i < 6; ++i)
    if (ls[i]->N != N[i] || ls[i]->K != K[i] || !ls[i]->w_hi || !ls[i]->w_lo || ls[i]->Kp < K[i]) return false;
  return H % 32 == 0 && I % 32 == 0;
}

// y (M, N) fp32 = x W^T + b (+ R): the forward of autograd._Linear -- one pass over x writes its row planes, zero padded
// to Mp rows, into the SAVED block (`xP`: hi [Mp][K] then lo): the A operand of this GEMM and, in the backward, an operand
// of dW = dY^T X through gridmm_linear_planes_tn (no transposed copy of x).
// (yP != NULL: the result also -- or, with y == NULL, only -- as bf16 planes hi [M][N] then lo: what the attention kernels on
// the bf16 matrix pipe read)
int linear_fwd(const gridmm_linear_train_t& l, const float* x, unsigned short* xP, unsigned short* /*rows*/, const float* R,
               float* y, int M, int act, gridmm_stream_t st, unsigned short* yP = nullptr) {
  const int K = l.K, Mp = mp32(M);
  int rc = gridmm_split_rows_pad(x, K, xP, xP + (size_t)K * Mp, K, nullptr, nullptr, M, K, Mp, st);
  if (rc != GRIDMM_OK) return rc;
  return gridmm_linear_planes(xP, xP + (size_t)K * Mp, K, l.w_hi, l.w_lo, l.Kp, l.bias, R, R ? l.N : 0, y, l.N, yP,
                              yP ? yP + (size_t)M * l.N : nullptr, l.N, M, l.N, K, act, st);
}

// The same when the producer of x (LayerNorm, GELU, attention) already wrote its planes into the saved block.
int linear_fwd_planes(const gridmm_linear_train_t& l, const unsigned short* xP, const float* R, float* y, int M, int act,
                      gridmm_stream_t st, unsigned short* yP = nullptr, unsigned short* yPlo = nullptr) {
  const int K = l.K, Mp = mp32(M);
  return gridmm_linear_planes(xP, xP + (size_t)K * Mp, K, l.w_hi, l.w_lo, l.Kp, l.bias, R, R ? l.N : 0, y, l.N, yP,
                              yP ? (yPlo ? yPlo : yP + (size_t)M * l.N) : nullptr, l.N, M, l.N, K, act, st);
}

// The cross attention runs on the bf16 matrix pipe (gridmm_attention_rows_train / _bwd) when the caller hands the planes of
// the context's K / V projections; the self attention always does (its q | k | v planes come from this layer's own GEMM).
// The fp32 regions `q` / `qkv` of the saved block then hold the PLANES of q / qkv (hi [M][N] then lo: the same bytes).
bool planes_ok(const void* KV_hi, const void* KV_lo, int64_t kv_bs, int kv_rs, int k_col, int v_col, int Sk) {
  return KV_hi && KV_lo && Sk <= 2048 && !(kv_bs & 7) && !(kv_rs & 7) && !(k_col & 7) && !(v_col & 7);
}

// Backward of one Linear (autograd._Linear.backward): dY -> [one pass: zero-padded row planes + column sums = db],
// dX = dY W (+ Radd: the gradient arriving over another branch, summed in the GEMM epilogue), dW = dY^T X from the row
// planes of dY and of the saved x.
struct LinWs { unsigned short *rows, *yT; float *cs_ws, *splitk; };

// The weight gradients of a layer's backward, collected and issued as ONE grouped launch at the end of the layer
// (gridmm_linear_planes_tn_grouped): each Linear keeps the planes of its dY, its column-sum partials and its split-K
// workspace in a region of its own (bump-allocated from `area`) until then.
struct TnBatch {
  gridmm_tn_problem_t p[8];
  int n = 0;
  char* area = nullptr;
  size_t used = 0, cap = 0;
  char* take(size_t bytes) {
    char* q = area + used;
    used += a256(bytes);
    return used <= cap ? q : nullptr;
  }
};
// bytes of that area for a layer whose Linears have the given (N, K) and M rows
size_t tn_area_bytes(int M, const int* N, const int* K, int n) {
  const size_t Mp = mp32(M);
  size_t b = 0;
  for (int i = 0; i < n; ++i) {
    const int splits = gridmm_linear_planes_tn_splits(M, N[i], K[i]);
    b += a256((size_t)N[i] * Mp * 4) + a256((size_t)8 * N[i] * 4);
    if (splits > 1) b += a256((size_t)splits * N[i] * K[i] * 4);
  }
  return b;
}
int tn_flush(TnBatch& tb, gridmm_stream_t st) {
  if (!tb.n) return GRIDMM_OK;
  const int rc = gridmm_linear_planes_tn_grouped(tb.p, tb.n, st);
  tb.n = 0;
  return rc;
}

// The dY-plane region of the NEXT Linear backward of the layer (hi [Mp][N] then lo), handed out ahead of time so that the
// kernel that PRODUCES dY (LayerNorm backward, GELU backward, the dropout pass) writes the planes itself and the Linear's
// own split pass is skipped (pass the pointer back to linear_bwd as yP_ready).
unsigned short* reserve_y(TnBatch& tb, int N, int M) { return (unsigned short*)tb.take((size_t)N * mp32(M) * 4); }

int linear_bwd(const gridmm_linear_train_t& l, const float* dY, const unsigned short* xP, const float* Radd, float* dX,
               float* dW, float* db, int M, const LinWs& ws, gridmm_stream_t st, TnBatch* tb = nullptr,
               unsigned short* yP_ready = nullptr) {
  const int N = l.N, K = l.K, Mp = mp32(M);
  const bool defer = tb && dW;
  const int splits = dW ? gridmm_linear_planes_tn_splits(M, N, K) : 1;   // <= 8: ws.splitk holds 8 partial tiles
  unsigned short* yh = yP_ready ? yP_ready : ws.yT;
  float *db_ws = ws.cs_ws, *splitk = ws.splitk;
  if (defer) {
    if (tb->n >= 8) return GRIDMM_EINVAL;
    if (!yP_ready) yh = (unsigned short*)tb->take((size_t)N * Mp * 4);
    db_ws = db ? (float*)tb->take((size_t)splits * N * 4) : nullptr;
    splitk = splits > 1 ? (float*)tb->take((size_t)splits * N * K * 4) : nullptr;
    if (!yh || (db && !db_ws) || (splits > 1 && !splitk)) return GRIDMM_EINVAL;
  }
  unsigned short* yl = yh + (size_t)N * Mp;
  // db: with a weight gradient the TN GEMM computes the column sums of dY from the planes itself (gridmm_linear_planes_tn_db);
  // without one the split pass does (its own reduction launch)
  const bool fold = db && dW;
  int rc = GRIDMM_OK;
  if (!yP_ready) rc = gridmm_split_rows_pad(dY, N, yh, yl, N, fold ? nullptr : db, (db && !fold) ? ws.cs_ws : nullptr, M, N, Mp, st);
  else if (db && !fold) return GRIDMM_EINVAL;
  if (rc != GRIDMM_OK) return rc;
  if (dX) {
    if (!l.wt_hi || !l.wt_lo || l.Np < N) return GRIDMM_EINVAL;
    rc = gridmm_linear_planes(yh, yl, N, l.wt_hi, l.wt_lo, l.Np, nullptr, Radd, Radd ? K : 0, dX, K, nullptr, nullptr, 0, M, K, N,
                              GRIDMM_ACT_NONE, st);
    if (rc != GRIDMM_OK) return rc;
  }
  if (defer) {
    gridmm_tn_problem_t& q = tb->p[tb->n++];
    q.A_hi = yh; q.A_lo = yl; q.lda = N;
    q.B_hi = xP; q.B_lo = xP + (size_t)K * Mp; q.ldb = K;
    q.C = dW; q.workspace = splitk; q.M = M; q.N = N; q.K = K; q.splits = splits;
    q.db_ws = fold ? db_ws : nullptr; q.db = fold ? db : nullptr;
  } else if (dW) {
    rc = gridmm_linear_planes_tn_db(yh, yl, N, xP, xP + (size_t)K * Mp, K, dW, splitk, M, N, K, splits, fold ? db_ws : nullptr,
                                    fold ? db : nullptr, st);
  }
  return rc;
}
}  // namespace

extern "C" size_t gridmm_xattn_layer_train_saved_bytes(int B, int Sq, int H, int I) {
  return carve_saved(nullptr, B, Sq, H, I, H / 64).bytes;
}

// forward scratch: row planes of the widest Linear input (M x I, hi + lo); backward: see below
extern "C" size_t gridmm_xattn_layer_train_workspace(int B, int Sq, int H, int I) {
  const size_t M = (size_t)B * Sq, Mp = mp32((int)M), W = (size_t)(I > 3 * H ? I : 3 * H);
  const size_t rows = a256(M * W * 4), yT = a256(W * Mp * 4), cs = a256(((Mp + 255) / 256) * W * 4);
  const size_t splitk = a256((size_t)8 * H * (I > 3 * H ? I : 3 * H) * 4);
  const size_t lnws = a256((M + 3) / 4 * 2 * H * 4), delta = a256((size_t)B * (H / 64) * ((Sq + 15) / 16 * 16) * 4);
  // gradients in flight: dh (M,H) x2, da (M,H) x2, dc (M,H), dqkv (M,3H), dg (M,I), df1 (M,I)
  const size_t grads = a256(M * H * 4) * 5 + a256(M * 3 * H * 4) + a256(M * (size_t)I * 4) * 2;
  const int Ns[6] = {H, H, 3 * H, H, I, H}, Ks[6] = {H, H, H, H, H, I};
  return rows + yT + cs + splitk + lnws + delta + grads + a256(gridmm_attention_rows_bwd_workspace(B, H / 64, Sq)) +
         tn_area_bytes((int)M, Ns, Ks, 6) + 4096;
}

extern "C" int gridmm_xattn_layer_train_fwd(const gridmm_xlayer_train_t* L, const float* X, const float* KV, const void* KV_hi,
                                            const void* KV_lo, const float* KV_shift, int64_t kv_shift_bs, int64_t kv_bs,
                                            int kv_rs, int k_col, int v_col,
                                            const uint8_t* ctx_mask, int ctx_mask_bs,
                                            const uint8_t* self_mask, int self_mask_bs, float* Y, void* saved,
                                            size_t saved_bytes, void* workspace, size_t workspace_bytes, int B, int Sq,
                                            int Sk, int heads, gridmm_stream_t stream) {
  // KV == NULL: a BertLayer (vilmodel.py:214-231: self attention + feed forward, no cross-attention block; the xq / xo / x_ln
  // members of L are not read)
  const bool cross = KV != nullptr;
  if (!L || !X || !Y || !saved || !workspace || B <= 0 || Sq <= 0 || (cross && Sk <= 0) || heads <= 0) return GRIDMM_EINVAL;
  const int H = heads * 64, I = L->ffn_i.N, M = B * Sq, Sqp = (Sq + 15) / 16 * 16;
  if (!shapes_ok(L, H, I, cross)) return GRIDMM_EINVAL;
  if (saved_bytes < gridmm_xattn_layer_train_saved_bytes(B, Sq, H, I) ||
      workspace_bytes < gridmm_xattn_layer_train_workspace(B, Sq, H, I))
    return GRIDMM_EINVAL;
  Saved s = carve_saved((char*)saved, B, Sq, H, I, heads);
  unsigned short* rows = (unsigned short*)workspace;          // row planes of the current Linear input (scratch)
  float* g = (float*)((char*)workspace + a256((size_t)M * (I > 3 * H ? I : 3 * H) * 4));   // gelu output (M, I)
  const float scale = 0.125f;
  const float ph = L->p_hidden, pa = L->p_attn;
  int rc;
#define GRIDMM_TRY(call) do { rc = (call); if (rc != GRIDMM_OK) return rc; } while (0)
  const int Mp = mp32(M);
  // Every producer (LayerNorm, attention, GELU) also writes the bf16 planes of its output straight into the SAVED block of
  // the Linear that consumes it: that Linear's A operand now, an operand of its weight gradient in the backward -- one split
  // pass per layer (over the layer input X) instead of six.
  auto ln = [&](const float* x, const float* r, const gridmm_ln_t& p, unsigned long long seed, float* y, unsigned short* yP) {
    unsigned short* yl = yP ? yP + (size_t)H * Mp : nullptr;
    if (ph > 0.f)
      return gridmm_layernorm_dropout_planes(x, r, H, p.gamma, p.beta, p.eps, y, yP, yl, ph, seed, L->seed_dev, M, H, stream);
    return gridmm_layernorm(x, H, r, H, p.gamma, p.beta, p.eps, y, H, nullptr, 0, nullptr, nullptr, yP, yl, H, M, H, stream);
  };
  const float* a1 = cross ? s.a1 : X;             // input of the self-attention block (and its LayerNorm's residual)
  if (!cross) {                                   // the layer input straight into the self-attention block: its planes
    GRIDMM_TRY(gridmm_split_rows_pad(X, H, s.a1T, s.a1T + (size_t)H * Mp, H, nullptr, nullptr, M, H, Mp, stream));
  } else {
  // ---- cross attention (vilmodel.py:370-379)
  if (!L->attention_fp32 && planes_ok(KV_hi, KV_lo, kv_bs, kv_rs, k_col, v_col, Sk)) {
    unsigned short* qP = (unsigned short*)s.q;
    const unsigned short *kvh = (const unsigned short*)KV_hi, *kvl = (const unsigned short*)KV_lo;
    GRIDMM_TRY(linear_fwd(L->xq, X, s.xT, rows, nullptr, nullptr, M, GRIDMM_ACT_NONE, stream, qP));
    GRIDMM_TRY(gridmm_attention_rows_train(qP, qP + (size_t)M * H, (int64_t)Sq * H, H, kvh + k_col, kvl + k_col, kv_bs, kv_rs,
                                           kvh + v_col, kvl + v_col, kv_bs, kv_rs, ctx_mask, ctx_mask_bs, s.c, (int64_t)Sq * H, H,
                                           s.cT, s.cT + (size_t)H * Mp, (int64_t)Sq * H, H, s.lse_x, Sqp,
                                           KV_shift ? KV_shift + v_col : nullptr, kv_shift_bs, B, heads, Sq, Sk, scale, pa,
                                           L->seed[0], L->seed_dev, stream));
  } else {
    GRIDMM_TRY(linear_fwd(L->xq, X, s.xT, rows, nullptr, s.q, M, GRIDMM_ACT_NONE, stream));
    GRIDMM_TRY(gridmm_attention_train_planes(s.q, (int64_t)Sq * H, H, KV + k_col, kv_bs, kv_rs, KV + v_col, kv_bs, kv_rs,
                                             ctx_mask, ctx_mask_bs, s.c, (int64_t)Sq * H, H, s.cT, s.cT + (size_t)H * Mp,
                                             (int64_t)Sq * H, H, s.lse_x, Sqp, B, heads, Sq, Sk, scale, pa, L->seed[0],
                                             L->seed_dev, stream));
  }
  GRIDMM_TRY(linear_fwd_planes(L->xo, s.cT, nullptr, s.h1, M, GRIDMM_ACT_NONE, stream));
  GRIDMM_TRY(ln(s.h1, X, L->x_ln, L->seed[1], s.a1, s.a1T));
  }
  // ---- self attention (vilmodel.py:172-182)
  if (L->attention_fp32) {
    GRIDMM_TRY(linear_fwd_planes(L->sqkv, s.a1T, nullptr, s.qkv, M, GRIDMM_ACT_NONE, stream));
    GRIDMM_TRY(gridmm_attention_train_planes(s.qkv, (int64_t)Sq * 3 * H, 3 * H, s.qkv + H, (int64_t)Sq * 3 * H, 3 * H,
                                             s.qkv + 2 * H, (int64_t)Sq * 3 * H, 3 * H, self_mask, self_mask_bs, s.c2,
                                             (int64_t)Sq * H, H, s.c2T, s.c2T + (size_t)H * Mp, (int64_t)Sq * H, H, s.lse_s,
                                             Sqp, B, heads, Sq, Sq, scale, pa, L->seed[2], L->seed_dev, stream));
  } else {
    unsigned short *qh = (unsigned short*)s.qkv, *ql = qh + (size_t)M * 3 * H;
    const int64_t bs = (int64_t)Sq * 3 * H;
    // row 0 of every episode through the projection (B rows, read in place through the row map), then the projection itself
    // with the k | v planes shifted by it (csrc/attention_train.hip "SHIFTED K / V")
    const unsigned short *ah = s.a1T, *al = s.a1T + (size_t)H * Mp;
    GRIDMM_TRY(gridmm_linear_planes_map(ah, al, H, 1, (int64_t)Sq * H, L->sqkv.w_hi, L->sqkv.w_lo, L->sqkv.Kp, GRIDMM_W_ROWMAJOR,
                                        L->sqkv.bias, nullptr, 0, s.qkv_shift, 3 * H, nullptr, nullptr, 0, B, 3 * H, H,
                                        GRIDMM_ACT_NONE, stream));
    GRIDMM_TRY(gridmm_linear_planes_shift(ah, al, H, L->sqkv.w_hi, L->sqkv.w_lo, L->sqkv.Kp, L->sqkv.bias, nullptr, 0, nullptr, 0, qh,
                                          ql, 3 * H, s.qkv_shift, Sq, H, M, 3 * H, H, GRIDMM_ACT_NONE, stream));
    GRIDMM_TRY(gridmm_attention_rows_train(qh, ql, bs, 3 * H, qh + H, ql + H, bs, 3 * H, qh + 2 * H, ql + 2 * H, bs, 3 * H, self_mask,
                                           self_mask_bs, s.c2, (int64_t)Sq * H, H, s.c2T, s.c2T + (size_t)H * Mp, (int64_t)Sq * H,
                                           H, s.lse_s, Sqp, s.qkv_shift + 2 * H, (int64_t)3 * H, B, heads, Sq, Sq, scale, pa,
                                           L->seed[2], L->seed_dev, stream));
  }
  GRIDMM_TRY(linear_fwd_planes(L->so, s.c2T, nullptr, s.h2, M, GRIDMM_ACT_NONE, stream));
  GRIDMM_TRY(ln(s.h2, a1, L->s_ln, L->seed[3], s.a2, s.a2T));
  // ---- feed forward (vilmodel.py:184-209)
  // linear1 and its GELU as one launch: s.f1 = the pre-activation (for the GELU backward), s.gT = the planes of gelu(s.f1)
  GRIDMM_TRY(linear_fwd_planes(L->ffn_i, s.a2T, nullptr, s.f1, M, GRIDMM_ACT_GELU_PLANES, stream, s.gT, s.gT + (size_t)I * Mp));
  (void)g;
  GRIDMM_TRY(linear_fwd_planes(L->ffn_o, s.gT, nullptr, s.h3, M, GRIDMM_ACT_NONE, stream));
  GRIDMM_TRY(ln(s.h3, s.a2, L->f_ln, L->seed[4], Y, nullptr));
#undef GRIDMM_TRY
  return GRIDMM_OK;
}

extern "C" int gridmm_xattn_layer_bwd(const gridmm_xlayer_train_t* L, const float* X, const float* KV, const void* KV_hi,
                                      const void* KV_lo, const float* KV_shift, int64_t kv_shift_bs, int64_t kv_bs, int kv_rs,
                                      int k_col, int v_col,
                                      const uint8_t* ctx_mask, int ctx_mask_bs,
                                      const uint8_t* self_mask, int self_mask_bs, const void* saved, size_t saved_bytes,
                                      const float* dY, float* dX, float* dKV, int64_t dkv_bs, int dkv_rs,
                                      const gridmm_xlayer_grads_t* G, void* workspace, size_t workspace_bytes, int B, int Sq,
                                      int Sk, int heads, gridmm_stream_t stream) {
  const bool cross = KV != nullptr;               // KV == NULL: the BertLayer form (see the forward); dKV is not written
  if (!L || !X || !saved || !dY || !dX || (cross && !dKV) || !G || !workspace || B <= 0 || Sq <= 0 || (cross && Sk <= 0) ||
      heads <= 0)
    return GRIDMM_EINVAL;
  const int H = heads * 64, I = L->ffn_i.N, M = B * Sq, Mp = mp32(M), Sqp = (Sq + 15) / 16 * 16;
  if (!shapes_ok(L, H, I, cross)) return GRIDMM_EINVAL;
  if (saved_bytes < gridmm_xattn_layer_train_saved_bytes(B, Sq, H, I) ||
      workspace_bytes < gridmm_xattn_layer_train_workspace(B, Sq, H, I))
    return GRIDMM_EINVAL;
  Saved s = carve_saved((char*)saved, B, Sq, H, I, heads);
  const size_t W = (size_t)(I > 3 * H ? I : 3 * H);
  char* w = (char*)workspace;
  auto take = [&](size_t bytes) { char* p = w; w += a256(bytes); return p; };
  LinWs lw;
  lw.rows = (unsigned short*)take((size_t)M * W * 4);
  lw.yT = (unsigned short*)take(W * Mp * 4);
  lw.cs_ws = (float*)take(((Mp + 255) / 256) * W * 4);
  lw.splitk = (float*)take((size_t)8 * H * W * 4);
  float* lnws = (float*)take((size_t)(M + 3) / 4 * 2 * H * 4);
  float* delta = (float*)take((size_t)B * heads * Sqp * 4);
  float* dh = (float*)take((size_t)M * H * 4);      // gradient of a pre-LayerNorm sum
  float* dr = (float*)take((size_t)M * H * 4);      // gradient of the LayerNorm's residual input
  float* da = (float*)take((size_t)M * H * 4);      // gradient of a block output (both branches summed)
  float* da_b = (float*)take((size_t)M * H * 4);
  float* dc = (float*)take((size_t)M * H * 4);      // gradient of an attention output
  float* dqkv = (float*)take((size_t)M * 3 * H * 4);
  float* dg = (float*)take((size_t)M * I * 4);
  float* df1 = (float*)take((size_t)M * I * 4);
  const size_t att_bytes = gridmm_attention_rows_bwd_workspace(B, heads, Sq);
  void* att_ws = take(att_bytes);                   // delta + planes of dO of the attention backward on the bf16 matrix pipe
  TnBatch tb;                                       // the layer's weight gradients: one grouped launch at the end
  {
    const int Ns[6] = {H, H, 3 * H, H, I, H}, Ks[6] = {H, H, H, H, H, I};
    tb.cap = tn_area_bytes(M, Ns, Ks, 6);
    tb.area = take(tb.cap);
  }
  const float scale = 0.125f;
  const float ph = L->p_hidden, pa = L->p_attn;
  int rc;
#define GRIDMM_TRY(call) do { rc = (call); if (rc != GRIDMM_OK) return rc; } while (0)
  // LN backward: dx = gradient of the (dropped-out) dense output, dres = gradient of the residual input
  // (yP: the reserved dY-plane region of the Linear in front of the LayerNorm -- the kernel writes the planes of dx there)
  auto ln_bwd = [&](const float* x, const float* r, const gridmm_ln_t& p, unsigned long long seed, const float* dy, float* dx,
                    float* dres, float* dgm, float* dbt, unsigned short* yP) {
    unsigned short* yl = yP ? yP + (size_t)H * Mp : nullptr;
    if (ph > 0.f)
      return gridmm_layernorm_dropout_bwd_planes(x, r, H, p.gamma, p.eps, dy, dx, yP, yl, dres, dgm, dbt, lnws, ph, seed,
                                                 L->seed_dev, M, H, stream);
    return gridmm_layernorm_bwd_planes(x, H, r, H, p.gamma, p.eps, dy, H, dx, H, yP, yl, dgm, dbt, lnws, M, H, stream);
  };
  unsigned short* yP;
  // without dropout the residual's gradient IS dx (one buffer)
  auto res_of = [&](float* dx, float* dres) { return ph > 0.f ? dres : dx; };

  // ---- feed forward
  // (the producers of a Linear's dY -- LayerNorm backward, GELU backward -- write its planes into that Linear's region of the
  // batch: no split pass for ffn_o, ffn_i, so, xo; the attention backward leaves fp32 gradients: sqkv and xq still split)
  if (!(yP = reserve_y(tb, H, M))) return GRIDMM_EINVAL;
  GRIDMM_TRY(ln_bwd(s.h3, s.a2, L->f_ln, L->seed[4], dY, dh, dr, G->f_ln_g, G->f_ln_b, yP));
  GRIDMM_TRY(linear_bwd(L->ffn_o, dh, s.gT, nullptr, dg, G->ffn_o_w, G->ffn_o_b, M, lw, stream, &tb, yP));
  if (!(yP = reserve_y(tb, I, M))) return GRIDMM_EINVAL;
  GRIDMM_TRY(gridmm_activation_planes(s.f1, dg, df1, yP, yP + (size_t)I * Mp, (int64_t)M * I, 1, stream));
  GRIDMM_TRY(linear_bwd(L->ffn_i, df1, s.a2T, res_of(dh, dr), da, G->ffn_i_w, G->ffn_i_b, M, lw, stream, &tb, yP));   // da = d a2
  // ---- self attention
  if (!(yP = reserve_y(tb, H, M))) return GRIDMM_EINVAL;
  GRIDMM_TRY(ln_bwd(s.h2, cross ? s.a1 : X, L->s_ln, L->seed[3], da, dh, dr, G->s_ln_g, G->s_ln_b, yP));
  GRIDMM_TRY(linear_bwd(L->so, dh, s.c2T, nullptr, dc, G->so_w, G->so_b, M, lw, stream, &tb, yP));
  if (L->attention_fp32) {
    GRIDMM_TRY(gridmm_attention_bwd(s.qkv, (int64_t)Sq * 3 * H, 3 * H, s.qkv + H, (int64_t)Sq * 3 * H, 3 * H, s.qkv + 2 * H,
                                    (int64_t)Sq * 3 * H, 3 * H, self_mask, self_mask_bs, s.c2, (int64_t)Sq * H, H, dc,
                                    (int64_t)Sq * H, H, s.lse_s, delta, dqkv, (int64_t)Sq * 3 * H, 3 * H, dqkv + H,
                                    (int64_t)Sq * 3 * H, 3 * H, dqkv + 2 * H, (int64_t)Sq * 3 * H, 3 * H, B, heads, Sq, Sq, Sqp,
                                    scale, pa, L->seed[2], L->seed_dev, stream));
  } else {
    const unsigned short *qh = (const unsigned short*)s.qkv, *ql = qh + (size_t)M * 3 * H;
    const int64_t bs = (int64_t)Sq * 3 * H;
    // the kernels also write the planes of dq | dk | dv (strides of dqkv): the dY planes of the q | k | v projection
    if (!(yP = reserve_y(tb, 3 * H, M))) return GRIDMM_EINVAL;
    unsigned short *gh = yP, *gl = yP + (size_t)3 * H * Mp;
    GRIDMM_TRY(gridmm_attention_rows_bwd_planes(qh, ql, bs, 3 * H, qh + H, ql + H, bs, 3 * H, qh + 2 * H, ql + 2 * H, bs, 3 * H, self_mask,
                                         self_mask_bs, s.c2, (int64_t)Sq * H, H, dc, (int64_t)Sq * H, H, s.lse_s,
                                         s.qkv_shift + 2 * H, (int64_t)3 * H, att_ws, att_bytes, dqkv, bs, 3 * H, dqkv + H, bs, 3 * H, dqkv + 2 * H, bs, 3 * H,
                                         gh, gl, gh + H, gl + H, gh + 2 * H, gl + 2 * H, B, heads, Sq, Sq, Sqp, scale,
                                         pa, L->seed[2], L->seed_dev, stream));
  }
  GRIDMM_TRY(linear_bwd(L->sqkv, dqkv, s.a1T, res_of(dh, dr), cross ? da_b : dX, G->sqkv_w, G->sqkv_b, M, lw, stream, &tb,
                        L->attention_fp32 ? nullptr : yP));   // d a1
  if (!cross) return tn_flush(tb, stream);
  // ---- cross attention
  if (!(yP = reserve_y(tb, H, M))) return GRIDMM_EINVAL;
  GRIDMM_TRY(ln_bwd(s.h1, X, L->x_ln, L->seed[1], da_b, dh, dr, G->x_ln_g, G->x_ln_b, yP));
  GRIDMM_TRY(linear_bwd(L->xo, dh, s.cT, nullptr, dc, G->xo_w, G->xo_b, M, lw, stream, &tb, yP));
  float* dq = da;                                   // (M, H) scratch: the gradient of the query projection
  unsigned short* yqP = nullptr;                    // its planes, when the attention backward on the matrix pipe wrote them
  if (!L->attention_fp32 && planes_ok(KV_hi, KV_lo, kv_bs, kv_rs, k_col, v_col, Sk)) {
    const unsigned short* qP = (const unsigned short*)s.q;
    const unsigned short *kvh = (const unsigned short*)KV_hi, *kvl = (const unsigned short*)KV_lo;
    if (!(yqP = reserve_y(tb, H, M))) return GRIDMM_EINVAL;
    GRIDMM_TRY(gridmm_attention_rows_bwd_planes(qP, qP + (size_t)M * H, (int64_t)Sq * H, H, kvh + k_col, kvl + k_col, kv_bs, kv_rs,
                                         kvh + v_col, kvl + v_col, kv_bs, kv_rs, ctx_mask, ctx_mask_bs, s.c, (int64_t)Sq * H, H, dc,
                                         (int64_t)Sq * H, H, s.lse_x, KV_shift ? KV_shift + v_col : nullptr, kv_shift_bs, att_ws,
                                         att_bytes, dq, (int64_t)Sq * H, H, dKV + k_col, dkv_bs,
                                         dkv_rs, dKV + v_col, dkv_bs, dkv_rs, yqP, yqP + (size_t)H * Mp, nullptr, nullptr, nullptr,
                                         nullptr, B, heads, Sq, Sk, Sqp, scale, pa, L->seed[0], L->seed_dev, stream));
  } else {
    GRIDMM_TRY(gridmm_attention_bwd(s.q, (int64_t)Sq * H, H, KV + k_col, kv_bs, kv_rs, KV + v_col, kv_bs, kv_rs, ctx_mask,
                                    ctx_mask_bs, s.c, (int64_t)Sq * H, H, dc, (int64_t)Sq * H, H, s.lse_x, delta, dq,
                                    (int64_t)Sq * H, H, dKV + k_col, dkv_bs, dkv_rs, dKV + v_col, dkv_bs, dkv_rs, B, heads, Sq,
                                    Sk, Sqp, scale, pa, L->seed[0], L->seed_dev, stream));
  }
  GRIDMM_TRY(linear_bwd(L->xq, dq, s.xT, res_of(dh, dr), dX, G->xq_w, G->xq_b, M, lw, stream, &tb, yqP));
#undef GRIDMM_TRY
  return tn_flush(tb, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// One PRE-LayerNorm transformer layer of the differentiable path (the panorama encoder's and the grid encoder's layers:
// TransformerEncoderLayer.forward_pre, map_nav_src/models/transformer.py:170-182 through create_transformer_encoder,
// models/ops.py:11-16): x1 = x + drop(out_proj(attn(qkv(LN1(x)))));  y = x1 + drop(linear2(drop(gelu(linear1(LN2(x1)))))).
// Forward and whole backward as one C call each; like the calls above they own no arithmetic and issue the kernels of
// vilmodel_train.pre_ln_encoder's op-by-op form with the same tile choices (bit-identical: tests/test_hip_layer_train.py).
// The residual sums ride on the dropout pass (gridmm_dropout_add); the activation's dropout writes the planes that
// linear2 reads (no split pass).
namespace {
struct SavedP {
  unsigned short *h1T, *cT, *h2T, *fT;     // planes of every Linear input (hi [Mp][K] then lo)
  float *qkv, *c, *lse, *x1, *f1, *qkv_shift;
  size_t bytes;
};
SavedP carve_saved_p(char* base, int B, int S, int H, int I, int heads) {
  const size_t M = (size_t)B * S, Mp = mp32((int)M), Sp = (S + 15) / 16 * 16;
  char* w = base;
  auto take = [&](size_t bytes) { char* p = w; w += a256(bytes); return p; };
  SavedP s;
  s.h1T = (unsigned short*)take((size_t)H * Mp * 4);
  s.cT = (unsigned short*)take((size_t)H * Mp * 4);
  s.h2T = (unsigned short*)take((size_t)H * Mp * 4);
  s.fT = (unsigned short*)take((size_t)I * Mp * 4);
  s.qkv = (float*)take(M * 3 * H * 4);            // the PLANES of q | k | v (hi [M][3H] then lo: the same bytes)
  s.c = (float*)take(M * H * 4);
  s.lse = (float*)take((size_t)B * heads * Sp * 4);
  s.x1 = (float*)take(M * H * 4);
  s.f1 = (float*)take(M * (size_t)I * 4);
  s.qkv_shift = (float*)take((size_t)B * 3 * H * 4);
  s.bytes = (size_t)(w - base);
  return s;
}
bool shapes_ok_p(const gridmm_preln_layer_t* L, int H, int I) {
  const gridmm_linear_train_t* ls[4] = {&L->qkv, &L->out, &L->ffn1, &L->ffn2};
  const int N[4] = {3 * H, H, I, H}, K[4] = {H, H, H, I};
  for (int i = 0; i < 4; ++i)
    if (ls[i]->N != N[i] || ls[i]->K != K[i] || !ls[i]->w_hi || !ls[i]->w_lo || ls[i]->Kp < K[i]) return false;
  return H % 32 == 0 && I % 32 == 0 && L->p >= 0.f && L->p < 1.f;
}
}  // namespace

extern "C" size_t gridmm_preln_layer_saved_bytes(int B, int S, int H, int I) {
  return carve_saved_p(nullptr, B, S, H, I, H / 64).bytes;
}

extern "C" size_t gridmm_preln_layer_workspace(int B, int S, int H, int I) {
  const size_t M = (size_t)B * S, Mp = mp32((int)M), W = (size_t)(I > 3 * H ? I : 3 * H);
  const size_t lin = a256(M * W * 4) + a256(W * Mp * 4) + a256(((Mp + 255) / 256) * W * 4) + a256((size_t)8 * H * W * 4);
  const size_t lnws = a256((M + 3) / 4 * 2 * H * 4);
  // forward: LayerNorm output, dense output, activation; backward: d_o / dG, dF, dF1, dH, dxln, dx1, dC, dqkv
  const size_t bufs = a256(M * H * 4) * 5 + a256(M * 3 * H * 4) + a256(M * (size_t)I * 4) * 3;
  const int Ns[4] = {3 * H, H, I, H}, Ks[4] = {H, H, H, I};
  return lin + lnws + bufs + a256(gridmm_attention_rows_bwd_workspace(B, H / 64, S)) + tn_area_bytes((int)M, Ns, Ks, 4) + 4096;
}

extern "C" int gridmm_preln_layer_train_fwd(const gridmm_preln_layer_t* L, const float* X, const uint8_t* mask, int mask_bs,
                                            float* Y, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                            int B, int S, int heads, gridmm_stream_t stream) {
  if (!L || !X || !Y || !saved || !workspace || B <= 0 || S <= 0 || heads <= 0 || S > 2048) return GRIDMM_EINVAL;
  const int H = heads * 64, I = L->ffn1.N, M = B * S, Mp = mp32(M), Sp = (S + 15) / 16 * 16;
  if (!shapes_ok_p(L, H, I)) return GRIDMM_EINVAL;
  if (saved_bytes < gridmm_preln_layer_saved_bytes(B, S, H, I) || workspace_bytes < gridmm_preln_layer_workspace(B, S, H, I))
    return GRIDMM_EINVAL;
  SavedP s = carve_saved_p((char*)saved, B, S, H, I, heads);
  char* w = (char*)workspace;
  auto take = [&](size_t bytes) { char* p = w; w += a256(bytes); return p; };
  float* hbuf = (float*)take((size_t)M * H * 4);          // fp32 output of a LayerNorm (its planes are what is read)
  float* o = (float*)take((size_t)M * H * 4);             // dense output in front of a residual sum
  float* g = (float*)take((size_t)M * I * 4);             // activation
  const float p = L->p, scale = 0.125f;
  const int64_t nH = (int64_t)M * H, nI = (int64_t)M * I;
  int rc;
#define GRIDMM_TRY(call) do { rc = (call); if (rc != GRIDMM_OK) return rc; } while (0)
  // ---- self attention block: LN1 -> q | k | v (k | v planes shifted by row 0 of the episode) -> attention -> out_proj
  GRIDMM_TRY(gridmm_layernorm(X, H, nullptr, 0, L->ln1.gamma, L->ln1.beta, L->ln1.eps, hbuf, H, nullptr, 0, nullptr, nullptr,
                              s.h1T, s.h1T + (size_t)H * Mp, H, M, H, stream));
  {
    unsigned short *qh = (unsigned short*)s.qkv, *ql = qh + (size_t)M * 3 * H;
    const int64_t bs = (int64_t)S * 3 * H;
    const unsigned short *ah = s.h1T, *al = s.h1T + (size_t)H * Mp;
    GRIDMM_TRY(gridmm_linear_planes_map(ah, al, H, 1, (int64_t)S * H, L->qkv.w_hi, L->qkv.w_lo, L->qkv.Kp, GRIDMM_W_ROWMAJOR,
                                        L->qkv.bias, nullptr, 0, s.qkv_shift, 3 * H, nullptr, nullptr, 0, B, 3 * H, H,
                                        GRIDMM_ACT_NONE, stream));
    GRIDMM_TRY(gridmm_linear_planes_shift(ah, al, H, L->qkv.w_hi, L->qkv.w_lo, L->qkv.Kp, L->qkv.bias, nullptr, 0, nullptr, 0, qh,
                                          ql, 3 * H, s.qkv_shift, S, H, M, 3 * H, H, GRIDMM_ACT_NONE, stream));
    GRIDMM_TRY(gridmm_attention_rows_train(qh, ql, bs, 3 * H, qh + H, ql + H, bs, 3 * H, qh + 2 * H, ql + 2 * H, bs, 3 * H, mask,
                                           mask_bs, s.c, (int64_t)S * H, H, s.cT, s.cT + (size_t)H * Mp, (int64_t)S * H, H, s.lse,
                                           Sp, s.qkv_shift + 2 * H, (int64_t)3 * H, B, heads, S, S, scale, p, L->seed[0],
                                           L->seed_dev, stream));
  }
  GRIDMM_TRY(linear_fwd_planes(L->out, s.cT, nullptr, o, M, GRIDMM_ACT_NONE, stream));
  GRIDMM_TRY(gridmm_dropout_add(o, X, s.x1, nullptr, nullptr, nH, p, L->seed[1], L->seed_dev, stream));
  // ---- feed forward block: LN2 -> linear1 -> gelu -> dropout (writes the planes linear2 reads) -> linear2
  GRIDMM_TRY(gridmm_layernorm(s.x1, H, nullptr, 0, L->ln2.gamma, L->ln2.beta, L->ln2.eps, hbuf, H, nullptr, 0, nullptr, nullptr,
                              s.h2T, s.h2T + (size_t)H * Mp, H, M, H, stream));
  if (p > 0.f) {
    GRIDMM_TRY(linear_fwd_planes(L->ffn1, s.h2T, nullptr, s.f1, M, GRIDMM_ACT_NONE, stream));
    GRIDMM_TRY(gridmm_activation(s.f1, nullptr, g, nI, 0, stream));
    GRIDMM_TRY(gridmm_dropout_add(g, nullptr, nullptr, s.fT, s.fT + (size_t)I * Mp, nI, p, L->seed[2], L->seed_dev, stream));
  } else {   // no dropout between the activation and linear2: linear1 + GELU as one launch (pre-activation + planes of gelu)
    GRIDMM_TRY(linear_fwd_planes(L->ffn1, s.h2T, nullptr, s.f1, M, GRIDMM_ACT_GELU_PLANES, stream, s.fT, s.fT + (size_t)I * Mp));
  }
  GRIDMM_TRY(linear_fwd_planes(L->ffn2, s.fT, nullptr, o, M, GRIDMM_ACT_NONE, stream));
  GRIDMM_TRY(gridmm_dropout_add(o, s.x1, Y, nullptr, nullptr, nH, p, L->seed[3], L->seed_dev, stream));
#undef GRIDMM_TRY
  return GRIDMM_OK;
}

extern "C" int gridmm_preln_layer_bwd(const gridmm_preln_layer_t* L, const float* X, const uint8_t* mask, int mask_bs,
                                      const void* saved, size_t saved_bytes, const float* dY, float* dX,
                                      const gridmm_preln_grads_t* G, void* workspace, size_t workspace_bytes, int B, int S,
                                      int heads, gridmm_stream_t stream) {
  if (!L || !X || !saved || !dY || !dX || !G || !workspace || B <= 0 || S <= 0 || heads <= 0 || S > 2048) return GRIDMM_EINVAL;
  const int H = heads * 64, I = L->ffn1.N, M = B * S, Mp = mp32(M), Sp = (S + 15) / 16 * 16;
  if (!shapes_ok_p(L, H, I)) return GRIDMM_EINVAL;
  if (saved_bytes < gridmm_preln_layer_saved_bytes(B, S, H, I) || workspace_bytes < gridmm_preln_layer_workspace(B, S, H, I))
    return GRIDMM_EINVAL;
  SavedP s = carve_saved_p((char*)saved, B, S, H, I, heads);
  const size_t W = (size_t)(I > 3 * H ? I : 3 * H);
  char* w = (char*)workspace;
  auto take = [&](size_t bytes) { char* p = w; w += a256(bytes); return p; };
  LinWs lw;
  lw.rows = (unsigned short*)take((size_t)M * W * 4);
  lw.yT = (unsigned short*)take(W * Mp * 4);
  lw.cs_ws = (float*)take(((Mp + 255) / 256) * W * 4);
  lw.splitk = (float*)take((size_t)8 * H * W * 4);
  float* lnws = (float*)take((size_t)(M + 3) / 4 * 2 * H * 4);
  float* dd = (float*)take((size_t)M * I * 4);      // a gradient after its dropout (M x H or M x I)
  float* dF = (float*)take((size_t)M * I * 4);
  float* dF1 = (float*)take((size_t)M * I * 4);
  float* dH = (float*)take((size_t)M * H * 4);      // gradient of a LayerNorm output
  float* dln = (float*)take((size_t)M * H * 4);     // gradient of a LayerNorm input (its own branch)
  float* dx1 = (float*)take((size_t)M * H * 4);     // gradient of x1 (both branches summed)
  float* dC = (float*)take((size_t)M * H * 4);
  float* dqkv = (float*)take((size_t)M * 3 * H * 4);
  const size_t att_bytes = gridmm_attention_rows_bwd_workspace(B, heads, S);
  void* att_ws = take(att_bytes);
  TnBatch tb;                                       // the layer's four weight gradients: one grouped launch at the end
  {
    const int Ns[4] = {3 * H, H, I, H}, Ks[4] = {H, H, H, I};
    tb.cap = tn_area_bytes(M, Ns, Ks, 4);
    tb.area = take(tb.cap);
  }
  const float p = L->p, scale = 0.125f;
  const int64_t nH = (int64_t)M * H, nI = (int64_t)M * I;
  int rc;
#define GRIDMM_TRY(call) do { rc = (call); if (rc != GRIDMM_OK) return rc; } while (0)
  // the gradient of drop(t) from the gradient of its output: the same mask on the gradient (p == 0: the gradient itself)
  // (yP: the reserved dY-plane region of the Linear whose output was dropped: the pass writes the planes -- with p == 0 it is
  // run as a plain split, one launch either way)
  auto undrop = [&](const float* dy, int width, unsigned long long seed, const float** out, unsigned short* yP) {
    const int64_t n = (int64_t)M * width;
    unsigned short* yl = yP + (size_t)width * Mp;
    *out = p > 0.f ? dd : dy;
    return gridmm_dropout_add(dy, nullptr, p > 0.f ? dd : nullptr, yP, yl, n, p, seed, L->seed_dev, stream);
  };
  const float* t;
  unsigned short* yP;
  // ---- feed forward block
  if (!(yP = reserve_y(tb, H, M))) return GRIDMM_EINVAL;
  GRIDMM_TRY(undrop(dY, H, L->seed[3], &t, yP));
  GRIDMM_TRY(linear_bwd(L->ffn2, t, s.fT, nullptr, dF, G->ffn2_w, G->ffn2_b, M, lw, stream, &tb, yP));
  if (p > 0.f) {
    GRIDMM_TRY(gridmm_dropout_add(dF, nullptr, dd, nullptr, nullptr, nI, p, L->seed[2], L->seed_dev, stream));
    t = dd;
  } else {
    t = dF;
  }
  if (!(yP = reserve_y(tb, I, M))) return GRIDMM_EINVAL;
  GRIDMM_TRY(gridmm_activation_planes(s.f1, t, dF1, yP, yP + (size_t)I * Mp, nI, 1, stream));
  GRIDMM_TRY(linear_bwd(L->ffn1, dF1, s.h2T, nullptr, dH, G->ffn1_w, G->ffn1_b, M, lw, stream, &tb, yP));
  GRIDMM_TRY(gridmm_layernorm_bwd(s.x1, H, nullptr, 0, L->ln2.gamma, L->ln2.eps, dH, H, dln, H, G->ln2_g, G->ln2_b, lnws, M, H,
                                  stream));
  GRIDMM_TRY(gridmm_dropout_add(dln, dY, dx1, nullptr, nullptr, nH, 0.f, 0, nullptr, stream));       // x1 feeds LN2 and the sum
  // ---- self attention block
  if (!(yP = reserve_y(tb, H, M))) return GRIDMM_EINVAL;
  GRIDMM_TRY(undrop(dx1, H, L->seed[1], &t, yP));
  GRIDMM_TRY(linear_bwd(L->out, t, s.cT, nullptr, dC, G->out_w, G->out_b, M, lw, stream, &tb, yP));
  {
    const unsigned short *qh = (const unsigned short*)s.qkv, *ql = qh + (size_t)M * 3 * H;
    const int64_t bs = (int64_t)S * 3 * H;
    if (!(yP = reserve_y(tb, 3 * H, M))) return GRIDMM_EINVAL;
    unsigned short *gh = yP, *gl = yP + (size_t)3 * H * Mp;
    GRIDMM_TRY(gridmm_attention_rows_bwd_planes(qh, ql, bs, 3 * H, qh + H, ql + H, bs, 3 * H, qh + 2 * H, ql + 2 * H, bs, 3 * H, mask,
                                         mask_bs, s.c, (int64_t)S * H, H, dC, (int64_t)S * H, H, s.lse, s.qkv_shift + 2 * H,
                                         (int64_t)3 * H, att_ws, att_bytes, dqkv, bs, 3 * H, dqkv + H, bs, 3 * H, dqkv + 2 * H, bs,
                                         3 * H, gh, gl, gh + H, gl + H, gh + 2 * H, gl + 2 * H, B, heads, S, S, Sp, scale, p,
                                         L->seed[0], L->seed_dev, stream));
  }
  GRIDMM_TRY(linear_bwd(L->qkv, dqkv, s.h1T, nullptr, dH, G->qkv_w, G->qkv_b, M, lw, stream, &tb, yP));
  GRIDMM_TRY(gridmm_layernorm_bwd(X, H, nullptr, 0, L->ln1.gamma, L->ln1.eps, dH, H, dln, H, G->ln1_g, G->ln1_b, lnws, M, H,
                                  stream));
  GRIDMM_TRY(gridmm_dropout_add(dln, dx1, dX, nullptr, nullptr, nH, 0.f, 0, nullptr, stream));        // x feeds LN1 and the sum
#undef GRIDMM_TRY
  return tn_flush(tb, stream);
}
