/*
 * gridmm.h — C-ABI of libgridmm_hip.so: hand-written HIP/CDNA4 (gfx950) kernels for the
 * GridMM grid-memory forward path.
 *
 * The reference (MrZihan/GridMM) is 100 % Python: there is no FFI to mirror.  Each entry
 * point below replaces the stock-op sequence cited next to it (paths relative to the
 * reference tree).  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless a
 *     parameter is marked [host];
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it;
 *   - no allocation inside: outputs / workspaces are caller-provided;
 *   - returns 0 on success, a negative GRIDMM_E* code otherwise (never throws);
 *   - re-entrant, no global state (tuning overrides and in-kernel profilers exist only in the development build,
 *     `make debug` -> libgridmm_hip_dbg.so, declared under GRIDMM_DEBUG_HOOKS at the end of this header).
 */
#ifndef GRIDMM_H
#define GRIDMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIDMM_OK 0
#define GRIDMM_EINVAL (-1)   /* bad shape / unsupported size */
#define GRIDMM_EUNSUPPORTED (-2)   /* a fused form that this shape / device cannot take: issue the unfused calls */
#define GRIDMM_ELAUNCH (-1000) /* launch failed: status = -1000 - hipError_t */

#define GRIDMM_GRID 14
#define GRIDMM_CELLS 196

typedef void* gridmm_stream_t;

/* ABI version of this header (bumped on any signature change). */
int gridmm_abi_version(void);

/* Measurement helper ("a measured stream peak next to the specification"): reads `bytes` (% 16 == 0, 16-byte
 * aligned) at p exactly once with 16-byte loads from 2048 workgroups; out >= 2048 floats, zeroed by the caller (its content is
 * a by-product).  bench.py times it over a 4 GiB window -- 16x the Infinity Cache -- and reports the rate as
 * roofline_grid_aggregate.peak_measured. */
int gridmm_hbm_read_probe(const void* p, size_t bytes, float* out, gridmm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Grid memory ("fill_gridmap")
 * ---------------------------------------------------------------------------------------- */

/* Back-project ONE new observation per episode into world XY and append it to the
 * device-resident point history; update the running bounding box; derive the map scale and
 * the 196 cell-centre position features.
 * Replaces: get_rel_position  map_nav_src/r2r/env.py:115-121,
 *           EnvBatch.getGlobalMap :289-294 (projection), :312-331 (bbox, half_len),
 *           EnvBatch.get_gridmap_pos_fts :242-265.
 * Arithmetic is fp32 in the reference's operation order without FMA contraction (bit-exact XY).
 *   depth      [B][n_pts] uint16, sampled patch-centre depth, view-major (n_pts = n_views*ppv)
 *   x_off      [ppv] f32   lateral offsets * tan(fov/2)           (host computes, env.py:118)
 *   view_cos/sin [n_views] f32, cos/sin of the python-double view angle rounded to f32
 *   pose       [B][2] f32  (x, y) of the current viewpoint rounded to f32
 *   n_pts      [B] int32   in: points already in the history of each episode; out: += n_views*ppv
 *                          (kept on the device so that a whole step can be replayed from a hipGraph)
 *   hist_x/y   [B][cap] f32, hist_valid [B][cap] uint8: history (new points written at n_old[b])
 *   bbox       [B][4] f32  running (max_x, min_x, max_y, min_y); init (-10000,10000,-10000,10000)
 *   half_len   [B] f32 out; pos_fts [B][196][5] f32 out
 *   active     [B] uint8 or NULL: episodes with 0 are skipped entirely
 * VLN-CE twin (VLN_CE/vlnce_baselines/models/Policy_ViewSelection_GridMap.py:632-641, 689-825), flags bit 0:
 *   depth_f32 = 1: depth is float32 metres used as is; view_stride = n_views: per-episode view_cos/sin
 *   [B][n_views] of (v*pi/6 - heading); gy = -ry + y; cell features through the (x, Z, y) reading of
 *   vlnce_baselines/models/utils.py:125-144; max_dist 25 (R2R-CE) / 40 (RxR-CE) instead of 30.
 */
#define GRIDMM_FLAG_VLNCE 1
int gridmm_grid_project(const void* depth, int depth_f32, const float* x_off, const float* view_cos,
                        const float* view_sin, int view_stride, const float* pose, int32_t* n_pts,
                        float* hist_x, float* hist_y, uint8_t* hist_valid, float* bbox,
                        float* half_len, float* pos_fts, const uint8_t* active,
                        int B, int n_views, int ppv, int cap, float depth_div, int flags, float max_dist,
                        gridmm_stream_t stream);

/* Re-bin the WHOLE history of every episode into the current egocentric 14x14 frame and
 * build the per-cell point lists (stable counting sort by cell id).
 * Replaces: EnvBatch.getGlobalMap map_nav_src/r2r/env.py:337-369 (rotate, scale, truncate,
 *           clamp, 196-iteration mask loop).   Cell ids are bit-exact with the reference.
 *   n_pts      [B] int32   points in each history (after the append)
 *   head_cs    [B][2] f32  cos/sin of (-heading) rounded to f32 (host)
 *   cell_id    [B][cap] int16 out: x*14+y, or -1 for invalid depth
 *   perm       [B][cap] int32 out: point indices sorted by (cell, index); invalid points last
 *   cell_start [B][198] int32 out: perm range of cell c is [cell_start[c], cell_start[c+1]);
 *              cell_start[196] = #valid points, cell_start[197] = n_pts
 */
int gridmm_grid_bin(const float* hist_x, const float* hist_y, const uint8_t* hist_valid,
                    const int32_t* n_pts, const float* pose, const float* head_cs,
                    const float* half_len, int16_t* cell_id, int32_t* perm, int32_t* cell_start,
                    int B, int cap, int flags, gridmm_stream_t stream);
/* (flags bit 0 = GRIDMM_FLAG_VLNCE: head_cs = cos/sin(-heading + pi) and map_x = -(tx cos + ty sin)) */

/* cmax[0] = the batch's largest occupied-cell count after a binning call: the reference's max_cell_num
 * (map_nav_src/models/vilmodel.py:809-823, a python max() over the batch), on the device in one launch so that a caller can
 * fetch it with one small asynchronous copy behind the binning kernels.   cell_start [B][198] as gridmm_grid_bin wrote it. */
int gridmm_grid_cell_count_max(const int32_t* cell_start, int32_t* cmax, int B, gridmm_stream_t stream);

/* Same counting sort for caller-provided cell ids (the reference's `grid_map` list form,
 * map_nav_src/r2r/agent.py:168): ids are int16 in {-1, 0..195}. */
int gridmm_grid_sort_ids(const int16_t* cell_id, const int32_t* n_pts, int32_t* perm,
                         int32_t* cell_start, int B, int cap, gridmm_stream_t stream);

/* gridmm_grid_bin for deep memories: the same result (bit-exact cell ids, the same stable order), with every episode
 * cut into `slices` contiguous parts handled by their own workgroups (histogram | scan | scatter; 3 launches).
 *   workspace  [B][slices][17][197] int32 scratch;  slices = 1 (or workspace NULL) runs gridmm_grid_bin.
 * Replaces: the same lines as gridmm_grid_bin (map_nav_src/r2r/env.py:337-369). */
int gridmm_grid_bin_sliced(const float* hist_x, const float* hist_y, const uint8_t* hist_valid,
                           const int32_t* n_pts, const float* pose, const float* head_cs,
                           const float* half_len, int16_t* cell_id, int32_t* perm, int32_t* cell_start,
                           int32_t* workspace, int slices, int B, int cap, int flags, gridmm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Instruction-relevance grid aggregation
 * ---------------------------------------------------------------------------------------- */

/* Split text_fts = text_proj(txt_embeds) (fp32) into fp16 hi + lo planes laid out as MFMA
 * B-fragments.   text [B][L][D] f32 -> frag [B][2][Lt][D/32][64][8] fp16, Lt = ceil(L/16). */
int gridmm_text_fragments(const float* text, void* frag, int B, int L, int D,
                          gridmm_stream_t stream);

/* One pass over the fp16 slab: per point relevance w_j = max_l <x_j, text_l> (all L columns,
 * padded tokens included, vilmodel.py:798) on MFMA f16 tiles, then per cell
 * out[c] = sum_j softmax_j(w_j) x_j  (online softmax, fp32) and occ[c] = cell non-empty.
 * Replaces: map_nav_src/models/vilmodel.py:797-807 (the 196*B python loop).  grid_proj is
 * applied AFTER the reduction (W (sum_j a_j x_j) + b, since sum_j a_j = 1).
 *   slab       [B][cap][D] fp16      perm/cell_start as produced by gridmm_grid_bin
 *   cells      [B][196][D] f32 out (zeros for empty cells); occ [B][196] uint8 out
 *   relevance  [B][cap] f32 out or NULL: w of the point at SORTED position p (slot perm[b][p]); saved for the backward
 *              (D = 768: also the intermediate of the two-pass path -- relevance pass, then accumulation pass; with NULL
 *              that shape runs on the slower single-kernel fallback)
 *   workspace  gridmm_grid_aggregate_workspace(B, D, n_chunks) bytes of device memory (contents undefined on entry):
 *              an episode's sorted points are cut into n_chunks equal shares, one workgroup each; a cell that a cut
 *              splits leaves its pieces there (sum, denominator, maximum) and a merge kernel combines them
 */
size_t gridmm_grid_aggregate_workspace(int B, int D, int n_chunks);
int gridmm_grid_aggregate(const void* slab, const int32_t* perm, const int32_t* cell_start,
                          const void* text_frag, float* cells, uint8_t* occ, float* relevance,
                          void* workspace, int B, int cap, int D, int L, int n_chunks,
                          gridmm_stream_t stream);

/* The same with the routing of the backward as a second by-product (fine-tune / pre-training forward):
 *   amax [B][cap] int32 out: arg-max instruction token of the point at sorted position p (first maximum, as torch.max).
 * Returns GRIDMM_OK with amax written, 1 when the shape ran on the generic kernel (amax untouched: use
 * gridmm_grid_aggregate_bwd, which recomputes it), < 0 on error.  Replaces: vilmodel.py:797-807. */
int gridmm_grid_aggregate_train(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                const void* text_frag, float* cells, uint8_t* occ, float* relevance, int32_t* amax,
                                void* workspace, int B, int cap, int D, int L, int n_chunks, gridmm_stream_t stream);

/* The two-pass aggregation (D = 768; instructions outside 33..96 tokens at D <= 512) for a memory that stays on the device:
 * the relevance w_j = max_l <x_j, text_l> of a point depends only on its slab row and on the instruction, both constant over
 * an episode, while the reference recomputes all of them at every step (vilmodel.py:797-798 inside the per-step forward).
 * Here they are kept in history order and only the points without a value are computed (normally the observation just
 * appended); the accumulation pass then reads the slab ONCE.  Same cells / occ / relevance as gridmm_grid_aggregate, bit for bit.
 *   n_pts      [B] int32 device: points per episode AFTER this step's append (the grid memory's counter)
 *   active     [B] uint8 device or NULL (all): episodes that appended n_new rows in this step -- those rows are recomputed
 *              whatever rel_valid says (a rewound memory may have re-written them)
 *   rel_hist   [B][cap] f32 in/out: relevance by HISTORY index;  rel_valid [B] int32 in/out: leading points of the episode
 *              that have a value.  The caller clears rel_valid when the instruction (text_frag) changes or rows are recycled.
 *   scratch    gridmm_grid_aggregate_incremental_scratch(B, cap) bytes;  workspace / n_chunks as for gridmm_grid_aggregate
 *   relevance  [B][cap] f32 out (by sorted position, as gridmm_grid_aggregate writes it): required
 *   full       != 0: the caller knows that no point has a value yet (first step of an episode): the plain relevance pass over
 *              all points, its values filed into rel_hist (one launch more than gridmm_grid_aggregate, two fewer than the
 *              general sequence); always correct, whatever rel_valid holds.  Points without a cell (id -1: no depth,
 *              env.py:283-285) get no value from this form: whether a point has depth must not change over its lifetime
 * Returns GRIDMM_EINVAL (nothing computed) for the one-pass shapes and for shapes outside the two-pass kernels' range: use
 * gridmm_grid_aggregate there.  Device-side decisions only: replayable from a hipGraph. */
size_t gridmm_grid_aggregate_incremental_scratch(int B, int cap);
int gridmm_grid_aggregate_incremental(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                      const void* text_frag, const int32_t* n_pts, const uint8_t* active, int n_new,
                                      float* rel_hist, int32_t* rel_valid, void* scratch, float* cells, uint8_t* occ,
                                      float* relevance, void* workspace, int B, int cap, int D, int L, int n_chunks,
                                      int full, gridmm_stream_t stream);

/* Compact non-empty cells to the front (cell order), add the position embedding, build the
 * key mask exactly as vilmodel.py:813-823 does (including its in-place view quirk).
 *   proj [B][196][H] f32 = grid_proj(cells)+bias; pos_emb [B][196][H] f32
 *   out  rows [0,196) of a [B][S_pad][H] buffer (row stride H, batch stride S_pad*H)
 *   mask rows [0,196) of a [B][S_pad] uint8 buffer;  n_cells [B] int32 out; cmax [1] int32 out
 */
int gridmm_cells_compact(const float* proj, const float* pos_emb, const uint8_t* occ, float* out,
                         uint8_t* mask, int32_t* n_cells, int32_t* cmax, int B, int H, int S_pad,
                         gridmm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Encoder building blocks (QKV / FFN GEMMs on MFMA bf16, fp32 everywhere else)
 * ---------------------------------------------------------------------------------------- */

/* W [N][K] f32 -> bf16 hi / lo planes [N][Kp], Kp = roundup(K,32), zero padded. */
int gridmm_split_weight(const float* W, void* hi, void* lo, int N, int K, int Kp,
                        gridmm_stream_t stream);

#define GRIDMM_ACT_NONE 0
#define GRIDMM_ACT_GELU 1  /* exact erf gelu (vilmodel.py:47-53, transformer.py:472) */
#define GRIDMM_ACT_RELU 2
#define GRIDMM_ACT_QUICKGELU 3  /* x * sigmoid(1.702 x): CLIP's MLP (VLN_CE/.../gridmap/clip.py:26-28); gridmm_linear_planes only */
#define GRIDMM_ACT_GELU_PLANES 4 /* gridmm_linear_planes family, training: C receives the PRE-activation x W^T + b (what the GELU
                                  * backward reads), the planes C_hi / C_lo those of gelu(C) (what the next Linear reads): the
                                  * feed-forward's first Linear and its activation as one launch.  Needs C and the planes, no residual. */

/* C[M][N] = act(A[M][K] * W^T + bias) (+ residual), fp32 in / fp32 out, the contraction on
 * MFMA bf16 16x16x32 tiles as a 3-term split (a_hi w_hi + a_lo w_hi + a_hi w_lo, fp32
 * accumulate): ~2^-16 relative, inside the 1e-3 logit tolerance where plain bf16 is not.
 * Replaces every nn.Linear on the path (vilmodel.py:124-126, 165, 187, 201, 345-347, ...;
 * transformer.py in_proj/out_proj/linear1/linear2).
 *   lda/ldc/ldr in elements; residual may be NULL; bias may be NULL. */
int gridmm_linear(const float* A, int lda, const void* W_hi, const void* W_lo, int Kp,
                  const float* bias, const float* residual, int ldr, float* C, int ldc,
                  int M, int N, int K, int act, gridmm_stream_t stream);

/* Y = LayerNorm(X (+ R)) * gamma + beta, row-wise over H, fp32 two-pass statistics.
 * Optionally Y += add1 (+ table[idx[row]]).   Replaces BertLayerNorm / nn.LayerNorm uses. */
int gridmm_layernorm(const float* X, int ldx, const float* R, int ldr, const float* gamma,
                     const float* beta, float eps, float* Y, int ldy, const float* add1, int ld1,
                     const float* table, const int64_t* idx, void* Y_hi, void* Y_lo, int ldp,
                     int M, int H, gridmm_stream_t stream);
/* (Y may be NULL when only the bf16 hi/lo planes Y_hi/Y_lo [M][ldp] -- the next GEMM's A operand -- are wanted) */

/* fp32 rows -> bf16 hi/lo planes [M][ldp], zero padded to ldp (ldp % 8 == 0). */
int gridmm_split_rows(const float* X, int ldx, void* hi, void* lo, int ldp, int M, int K,
                      gridmm_stream_t stream);

/* gridmm_layernorm / gridmm_split_rows with the bf16 PLANES written through a batched row map (p_rpb rows per
 * episode, episodes p_bs elements apart; p_rpb <= 0: plain [M][ldp]): the producer of a sequence writes it straight into
 * its place inside a longer one (the [map | txt] context of the local encoder, vilmodel.py:846-848) -- no torch.cat. */
int gridmm_layernorm_map(const float* X, int ldx, const float* R, int ldr, const float* gamma, const float* beta,
                         float eps, float* Y, int ldy, const float* add1, int ld1, const float* table,
                         const int64_t* idx, void* Y_hi, void* Y_lo, int ldp, int p_rpb, int64_t p_bs, int M, int H,
                         gridmm_stream_t stream);
int gridmm_split_rows_map(const float* X, int ldx, void* hi, void* lo, int ldp, int p_rpb, int64_t p_bs, int M, int K,
                          gridmm_stream_t stream);

/* gridmm_linear with BOTH operands as pre-split bf16 planes (the hot-path GEMM: LDS-DMA tile pipeline,
 * no conversion in the loop).  K % 32 == 0, lda % 8 == 0, N % 4 == 0.  Output as fp32 (C) and/or as
 * bf16 hi/lo planes (C_hi/C_lo, row stride ldp) for the next GEMM. */
int gridmm_linear_planes(const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo,
                         int Kp, const float* bias, const float* residual, int ldr, float* C, int ldc,
                         void* C_hi, void* C_lo, int ldp, int M, int N, int K, int act,
                         gridmm_stream_t stream);

/* Same with an explicit tile configuration (cfg 0 = the heuristic; the shipping library answers the tiles the heuristic can
 * choose -- 2, 4, 13, 15, 16, 36, 43, 57 -- and GRIDMM_EINVAL for the rest of the development build's experiment table). */
int gridmm_linear_planes_cfg(const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo,
                             int Kp, const float* bias, const float* residual, int ldr, float* C, int ldc,
                             void* C_hi, void* C_lo, int ldp, int M, int N, int K, int act, int cfg,
                             gridmm_stream_t stream);

/* Layout of a pair of weight planes handed to gridmm_linear_planes_map. */
#define GRIDMM_W_ROWMAJOR 0   /* [N][Kp]: what gridmm_split_weight writes */
#define GRIDMM_W_TILED 1      /* [roundup(N,16) / 16][Kp / 32][16 rows][32 k] blocks, rows past N zero (gridmm_linear_t.wt_hi):
                               * every 1-KiB LDS-DMA piece of a BK = 32 tile is one contiguous KiB of memory.  Shapes whose tile
                               * choice is not a BK = 32 tile return GRIDMM_EUNSUPPORTED: pass the row-major planes then. */
/* gridmm_linear_planes in its general form.  The A rows are taken through a batched row map: GEMM row m = row (m % a_rpb)
 * of episode (m / a_rpb) in a buffer whose episodes lie a_bs elements apart (a_rpb <= 0: plain rows, a_bs ignored;
 * a_bs % 8 == 0).  A sub-sequence of a longer padded sequence is multiplied in place -- the instruction rows of the local
 * encoder's [map | txt] context (vilmodel.py:846-848), the map-node rows of [cells | nodes] (:843, :872) -- instead of being
 * copied out first.  The W planes arrive in the layout `w_layout` names (explicit: no sign-encoded arguments). */
int gridmm_linear_planes_map(const void* A_hi, const void* A_lo, int lda, int a_rpb, int64_t a_bs, const void* W_hi,
                             const void* W_lo, int Kp, int w_layout, const float* bias, const float* residual, int ldr,
                             float* C, int ldc, void* C_hi, void* C_lo, int ldp, int M, int N, int K, int act,
                             gridmm_stream_t stream);

/* gridmm_linear_planes whose PLANE output is shifted per episode: planes[m][n] = split(x[m][n] - shift[(m / shift_rpb) * N + n])
 * for the columns n >= shift_c0 (x = act(A W^T + bias) + residual; the fp32 output C stays x).  For the K / V projections of
 * the differentiable path: shift = the projection of row 0 of every episode (a B-row gridmm_linear_planes_map call with
 * a_rpb = 1), the attention kernels on the bf16 matrix pipe read the shifted planes and get V[row 0] as `vbar`
 * (gridmm_attention_rows_train).  C_hi / C_lo required; shift_c0 % 4 == 0. */
int gridmm_linear_planes_shift(const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo, int Kp,
                               const float* bias, const float* residual, int ldr, float* C, int ldc, void* C_hi, void* C_lo,
                               int ldp, const float* shift, int shift_rpb, int shift_c0, int M, int N, int K, int act,
                               gridmm_stream_t stream);

/* Several small plane GEMMs C_i = act_i(A_i W_i^T + b_i) in ONE launch (64x64 tiles, K % 64 == 0): the ClsPrediction
 * heads at the end of forward('navigation') (vilmodel.py:859-877, 903-905) are 32..1824 rows each.  `problems` is a
 * [host] array of n_problems <= GRIDMM_MAX_GROUPED records; A rows through the row map of gridmm_linear_planes_map;
 * act in {NONE, GELU, RELU}; outputs fp32 (C, ldc) and/or bf16 planes (C_hi / C_lo, ldp). */
#define GRIDMM_MAX_GROUPED 8
typedef struct {
  const void *A_hi, *A_lo; int lda; int a_rpb; int64_t a_bs;
  const void *W_hi, *W_lo; int Kp; const float* bias;
  float* C; int ldc; void *C_hi, *C_lo; int ldp;
  int M, N, K, act;
} gridmm_gemm_problem_t;
int gridmm_linear_planes_grouped(const gridmm_gemm_problem_t* problems, int n_problems, gridmm_stream_t stream);

/* Multi-head attention core, head_dim 64, fp32 (MFMA f32 16x16x4), online softmax.
 * O[b][i][h*64+d] = sum_j softmax_j(scale * <Q[b,i,h], K[b,j,h]>  over keys with kmask=1) V[b,j,h,d]
 * Masked keys contribute exactly 0 (both mask conventions of the reference, vilmodel.py:136,
 * 354 (-10000 additive) and transformer.py:176 (key_padding_mask), give 0 in fp32).
 *   Q/K/V element (b,i,h,d) at ptr[b*bs + i*rs + h*64 + d]  (strides in elements)
 *   kmask [B][Sk] uint8 (row stride mask_bs) */
int gridmm_attention(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs, int k_rs,
                     const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs,
                     float* O, int64_t o_bs, int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs,
                     int B, int heads, int Sq, int Sk, float scale, gridmm_stream_t stream);
/* (O may be NULL when only the bf16 hi/lo planes O_hi/O_lo, strides p_bs/p_rs in elements, are wanted) */

/* bf16x3 variant (the one on the hot path): Q and K as the bf16 hi/lo planes the QKV GEMM emits, V as
 * per-head RE-TILED planes VT[b][h][key tile of 32][d][32 slots] (Skp = roundup(Sk, 32) keys, zero padded;
 * slot 8g+e <-> key 4g+e for e<4, 16+4g+(e-4) otherwise) built by gridmm_transpose_v.  Both matmuls run on MFMA bf16 16x16x32 with the 3-term split, softmax in fp32. */
int gridmm_transpose_v(const void* V_hi, const void* V_lo, int64_t v_bs, int v_rs, void* T_hi, void* T_lo,
                       int B, int heads, int Sk, int Skp, gridmm_stream_t stream);
int gridmm_attention_planes(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                            const void* K_lo, int64_t k_bs, int k_rs, const void* T_hi, const void* T_lo, int Skp,
                            const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs, int o_rs, void* O_hi,
                            void* O_lo, int64_t p_bs, int p_rs, int B, int heads, int Sq, int Sk, float scale,
                            gridmm_stream_t stream);

/* bf16x3 attention with K AND V taken as the ROW-MAJOR hi/lo planes the QKV / KV GEMMs emit (no re-tiling pass):
 * a workgroup stages its head's K / V rows once in LDS (LDS-DMA, source-side swizzle) and reads V^T fragments
 * through the hardware transpose read.  Replaces gridmm_transpose_v + gridmm_attention_planes on the hot path of
 * BertSelfAttention / BertOutAttention (map_nav_src/models/vilmodel.py:317-379) and of the grid encoder's
 * nn.MultiheadAttention (map_nav_src/models/transformer.py:176-177).  Strides in elements, rows 16-byte aligned;
 * Sk <= 2048; kmask (B, Sk) bytes, 0 = masked (contributes exactly 0); a fully masked query row yields 0.
 * _cfg: cfg = 0 picks the launch shape, cfg > 0 forces one (tools/bench_attn2.py). */
int gridmm_attention_rows(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                          const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo, int64_t v_bs,
                          int v_rs, const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs, int o_rs, void* O_hi,
                          void* O_lo, int64_t p_bs, int p_rs, int B, int heads, int Sq, int Sk, float scale,
                          gridmm_stream_t stream);
int gridmm_attention_rows_cfg(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                              const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo,
                              int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs,
                              int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs, int B, int heads, int Sq,
                              int Sk, float scale, int cfg, gridmm_stream_t stream);

/* gridmm_attention_rows over a context that lives in TWO plane buffers: keys [0, S1) are rows of K / V, keys [S1, Sk)
 * rows (key - S1) of K2 / V2 (row stride kv2_rs, episode stride kv2_bs, both % 8 == 0); kmask spans all Sk keys.  Key
 * by key the same arithmetic as one concatenated buffer (bit-identical).  The local encoder's [map | txt] context
 * (map_nav_src/models/vilmodel.py:846-853): the K / V projections of the instruction rows do not change during an
 * episode and are kept apart from the per-step projections of the map rows. */
int gridmm_attention_rows_seg(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi,
                              const void* K_lo, int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo, int64_t v_bs,
                              int v_rs, int S1, const void* K2_hi, const void* K2_lo, const void* V2_hi,
                              const void* V2_lo, int64_t kv2_bs, int kv2_rs, const uint8_t* kmask, int mask_bs, float* O,
                              int64_t o_bs, int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs, int B, int heads,
                              int Sq, int Sk, float scale, gridmm_stream_t stream);

/* ---- one cross-modal layer as one call -------------------------------------------------------------------------
 * GraphLXRTXLayer.forward with graph_sprels = None (map_nav_src/models/vilmodel.py:399-414; pretrain / VLN-CE twins
 * identical): cross attention of the Sq tokens over a context whose K / V projections the caller has already computed
 * (KV planes (B, Sk, .), K at column k_col, V at v_col: the local encoder shares one K/V GEMM over its 4 layers,
 * vilmodel.py:843-853), self attention, feed forward; every block = dense + residual + LayerNorm.  Weights arrive as
 * the bf16 hi/lo planes of gridmm_split_weight.  All intermediates live in `workspace` (>= gridmm_xattn_layer_workspace
 * bytes, 256-byte aligned); outputs: Y fp32 (M, H) and/or its bf16 planes.  heads * 64 == H. */
typedef struct {                 /* nn.Linear(K, N) */
  const void *w_hi, *w_lo; const float* bias; int N, K, Kp;
  const void *wt_hi, *wt_lo;     /* optional (NULL: absent): the same planes TILED as [roundup(N,16) / 16][Kp / 32][16][32]
                                  * blocks, rows past N zero (GRIDMM_W_TILED): every 1-KiB LDS-DMA piece of a BK = 32 tile is then one
                                  * contiguous KiB of memory */
} gridmm_linear_t;
typedef struct { const float *gamma, *beta; float eps; } gridmm_ln_t;
typedef struct {
  gridmm_linear_t xq, xo;        /* visual_attention.att.query, visual_attention.output.dense */
  gridmm_linear_t sqkv, so;      /* visn_self_att.self.{query|key|value} stacked (3H, H), visn_self_att.output.dense */
  gridmm_linear_t ffn_i, ffn_o;  /* visn_inter.dense (H -> I, gelu), visn_output.dense (I -> H) */
  gridmm_ln_t x_ln, s_ln, f_ln;  /* the three output LayerNorms */
} gridmm_xlayer_t;
size_t gridmm_xattn_layer_workspace(int B, int Sq, int H, int I);
int gridmm_xattn_layer_fwd(const gridmm_xlayer_t* L, const float* X, const void* X_hi, const void* X_lo,
                           const void* KV_hi, const void* KV_lo, int64_t kv_bs, int kv_rs, int k_col, int v_col,
                           int Sk1, const void* KV2_hi, const void* KV2_lo, int64_t kv2_bs, int kv2_rs, int k2_col,
                           int v2_col, const uint8_t* ctx_mask, int ctx_mask_bs, const uint8_t* self_mask,
                           int self_mask_bs, float* Y, void* Y_hi, void* Y_lo, int y_p_rpb, int64_t y_p_bs,
                           void* workspace, size_t workspace_bytes, int B, int Sq, int Sk, int heads, gridmm_stream_t stream);
/* (KV2_hi != NULL: the context rows [Sk1, Sk) come from a second K / V plane buffer, gridmm_attention_rows_seg; NULL: all
 * Sk rows from KV) */
/* (y_p_rpb > 0: the output PLANES go through the row map of gridmm_layernorm_map -- Sq rows per episode into a buffer
 * whose episodes are y_p_bs elements apart, e.g. the [map | txt] context of the next encoder; 0: plain [M][H]) */

/* Patch tokens of a vision tower -> grid-memory slab: X (B * n_views, T, D) fp32 token rows, token 0 (class token)
 * dropped; episode b's slot `slab + b * slab_bs` receives n_views * (T-1) rows of D fp16, view-major.  The device-side
 * replacement of the GPU -> CPU -> GPU round trip at VLN_CE/vlnce_baselines/models/Policy_ViewSelection_GridMap.py:
 * 340-357, 496 (CLIP tokens to numpy, per-episode python lists, torch.tensor(...).cuda() again). */
int gridmm_tokens_to_slab(const float* X, int T, int D, void* slab, int64_t slab_bs, int B, int n_views,
                          gridmm_stream_t stream);

/* out[m] = <LayerNorm(X[m]) * gamma + beta, w> + b0      (tail of ClsPrediction,
 * vilmodel.py:663-674: Linear -> ReLU -> LN -> Linear(H,1)). */
int gridmm_ln_dot(const float* X, int ldx, const float* gamma, const float* beta, float eps,
                  const float* w, const float* b0, float* out, int M, int H,
                  gridmm_stream_t stream);

/* Logit masking + global/local fusion (vilmodel.py:859-907) with integer index maps.
 *   g_raw [B][G], l_raw [B][V], grid_raw [B][G] f32: head outputs; fuse_raw [B] f32 (pre-sigmoid) or NULL (0.5)
 *   gmap_masks, gmap_visited [B][G] uint8; vp_nav_masks [B][V] uint8
 *   cand_of_node [B][G] int32: j>0 unvisited node -> index k of the same vpid among the
 *       candidates, or -1 (add the sum of visited candidates' local logits); host-built from
 *       the python vpid lists.   cand_visited [B][V] uint8: candidate k>0 is a visited node.
 *   outputs global/grid/fused [B][G], local [B][V] f32 (-inf where masked) */
int gridmm_fuse_logits(const float* g_raw, const float* l_raw, const float* grid_raw,
                       const float* fuse_raw, const uint8_t* gmap_masks,
                       const uint8_t* gmap_visited, const uint8_t* vp_nav_masks,
                       const int32_t* cand_of_node, const uint8_t* cand_visited,
                       float* global_logits, float* local_logits, float* grid_logits,
                       float* fused_logits, int B, int G, int V, gridmm_stream_t stream);

/* ---- fused row-wise stages of forward('navigation') (launch-count reduction; gridmm_amd/csrc/navfuse.hip) -------------
 * gridmm_cells_embed = gridmm_cells_compact with grid_pos_embeddings (Linear(K=5, H) + LayerNorm, vilmodel.py:697-700,
 * 816) evaluated inside: out row p < n_b = proj[src cell] + LN(W_pos pos_fts[src cell] + b_pos) * gamma + beta, rows
 * [n_b, 196) zero; the key mask as gridmm_cells_compact (quirk included) at row stride mask_bs, followed by n_tail bytes
 * copied from tail_mask [B][n_tail] (the map-node mask of the [cells | nodes] sequence; NULL to skip).
 *   W_pos [K][H] f32: the nn.Linear(K, H) weight TRANSPOSED (built once per weight version), pos_fts [B][196][K] f32 */
int gridmm_cells_embed(const float* proj, const float* pos_fts, int K, const float* W_pos, const float* b_pos,
                       const float* gamma, const float* beta, float eps, const uint8_t* occ, float* out,
                       uint8_t* mask, int mask_bs, const uint8_t* tail_mask, int n_tail, int32_t* n_cells,
                       int32_t* cmax, int B, int H, int S_pad, int c_pad, gridmm_stream_t stream);
/* (c_pad: cell rows of the padded sequence, <= 196; 0 = 196.  The reference cuts the sequence to the batch's largest
 * occupied-cell count, vilmodel.py:809-823 / ops.py:46-68 pad_tensors_wgrad; a caller that knows a bound c_pad >= cmax --
 * e.g. the previous step's cmax rounded up to a bucket -- gets the same result on c_pad + n_tail rows: rows [0, c_pad) and
 * mask columns [0, c_pad) are written, the tail mask follows at column c_pad; cmax > c_pad truncates: check cmax and redo.) */

/* Position embeddings of graph nodes / candidate views: out[m] = LN(W pos[m] + b) * gamma + beta (+ add1[m])
 * (+ table[idx[m]]) for up to two row segments in one launch (vilmodel.py:828-833: gmap_pos_embeddings + image embeds +
 * step embedding; vp_pos_embeddings + image embeds), fp32 and/or bf16 planes, output rows through a batched row map
 * (out_rpb rows per episode, episodes out_bs elements apart, row stride H; out_rpb <= 0: plain).  `segs` is a [host] array.
 * The same launch assembles the byte masks of the two sequences the encoders attend over (vilmodel.py:846-851):
 *   kv_masks[b][kv_col0 ..] = [gmap_masks[b] (G) | txt_masks[b] (L)]   (row stride kv_bs; NULL to skip)
 *   q_masks [b]             = [gmap_masks[b] (G) | vp_masks[b] (V)]    (contiguous [B][G+V]; NULL to skip) */
typedef struct {
  const float* pos; int K;                       /* [M][K] position features, K <= 16 */
  const float *W, *bias, *gamma, *beta; float eps; /* nn.Linear(K, H) weight TRANSPOSED [K][H] + bias, LayerNorm */
  const float* add1; int ld1;                    /* [M][ld1] or NULL */
  const float* table; const int64_t* idx;        /* embedding table [.][H] + row index [M], or NULL */
  float* out; void *out_hi, *out_lo; int out_rpb; int64_t out_bs;
  int M;
} gridmm_embed_seg_t;
int gridmm_node_embed(const gridmm_embed_seg_t* segs, int n_segs, int H, const uint8_t* gmap_masks, int G,
                      const uint8_t* vp_masks, int V, const uint8_t* txt_masks, int L, uint8_t* kv_masks, int kv_bs,
                      int kv_col0, uint8_t* q_masks, int B, gridmm_stream_t stream);

/* The tails of the ClsPrediction heads (LayerNorm . w + b0 after Linear + ReLU, vilmodel.py:663-674) for one step, then
 * gridmm_fuse_logits' masking / fusion (vilmodel.py:859-907): one wave per head row over the batch, then one workgroup
 * per episode (two launches; the raw head outputs pass through `workspace`, gridmm_nav_heads_workspace bytes):
 *   h_gl   [B*(G+V)][ld_gl] f32: relu(Linear) of global_sap_head at columns [0,H) (node rows), local_sap_head at [H,2H)
 *          (view rows), og_head at [2H,3H) (view rows; only read when obj_logits != NULL)
 *   fuse_a, fuse_b [B][H] f32: the two K-halves of sap_fuse_linear's Linear (gmap[:,0] and vp[:,0] parts, no bias, no
 *          activation), fuse_bias [H]; NULL -> fusion weight 0.5 (vilmodel.py:864)
 *   h_grid [B*G][H] f32: relu(Linear) of grid_sap_head over the pre-local-encoder node rows
 *   tails  [host] 5 records: fuse, global, local, grid, object
 *   masks / index maps / logit outputs as gridmm_fuse_logits; obj_logits [B][V] masked by vp_obj_masks (or NULL) */
typedef struct { const float *gamma, *beta; float eps; const float* w; const float* b0; } gridmm_cls_tail_t;
size_t gridmm_nav_heads_workspace(int B, int G, int V);
int gridmm_nav_heads(const float* h_gl, int ld_gl, const float* fuse_a, const float* fuse_b, const float* fuse_bias,
                     const float* h_grid, const gridmm_cls_tail_t* tails, const uint8_t* gmap_masks,
                     const uint8_t* gmap_visited, const uint8_t* vp_nav_masks, const uint8_t* vp_obj_masks,
                     const int32_t* cand_of_node, const uint8_t* cand_visited, float* global_logits,
                     float* local_logits, float* grid_logits, float* fused_logits, float* obj_logits, void* workspace,
                     int B, int G, int V, int H, gridmm_stream_t stream);

/* Strided row copy / gather used to assemble [cells | gmap | txt] sequences without torch.cat:
 * dst[b][dst_row0 + i][:] = src[b][i][:] for i < rows.  H floats per row. */
int gridmm_copy_rows(const float* src, int64_t src_bs, int src_rs, float* dst, int64_t dst_bs,
                     int dst_rs, int B, int rows, int H, gridmm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training (backward) entry points -- SURVEY.md §8 rows a11 / a13 / a14: the fine-tune loop
 * (map_nav_src/r2r/agent_base.py:164-211, loss.backward at :199) and the pre-training loop
 * (pretrain_src/train_r2r.py:231-327) back-propagate through the same encoders; in the reference
 * that is torch autograd over nn.Linear / LayerNorm / softmax attention / GELU.
 * ---------------------------------------------------------------------------------------- */

/* X fp32 [M][C] (row stride ldx) -> transposed bf16 hi/lo planes T [C][Mp] (Mp % 32 == 0, zero padded),
 * plus colsum[C] = sum_m X[m][c] (NULL to skip).  With gridmm_linear_planes (C = A B^T) this gives
 *   dW [N][K] = dY^T X:  A = T(dY) [N][Mp], B = T(X) [K][Mp];   db = colsum(dY)
 * (backward of nn.Linear, e.g. vilmodel.py:84-86,128,144). */
int gridmm_transpose_split(const float* X, int ldx, void* T_hi, void* T_lo, float* colsum, float* colsum_ws, void* R_hi,
                           void* R_lo, int ldp, int M, int C, int Mp, gridmm_stream_t stream);
/* (colsum != NULL needs colsum_ws >= ceil(Mp / 256) * C floats: one partial per 256-row block, summed in a fixed
 * order -- no float atomics, db is bit-reproducible from run to run.  colsum = NULL with colsum_ws != NULL: the partials
 * only.  With a weight gradient the bias gradient comes from gridmm_linear_planes_tn_db instead, and no column sums are needed) */
/* (R_hi / R_lo, optional: the row-major planes [M][ldp] of the same X from the same pass -- the A operand of the
 * forward / dX GEMM -- so an activation or a gradient is read ONCE for both of its GEMM roles) */

/* Weight gradient WITHOUT transposed copies: C (N x K fp32, contiguous) = A^T B over the M rows of the row-major bf16
 * hi/lo planes A [M][lda >= N] (dY) and B [M][ldb >= K] (X), any M (the last 32-row step clamps its row reads to M - 1 and
 * zeroes A's copies in LDS).  The kernel stages row-major panels in LDS and reads the MFMA fragments through the hardware
 * transpose read (ds_read_b64_tr_b16).  splits > 1: contraction cut into `splits` ranges, partials in `workspace`
 * (splits x N x K floats), summed in order (deterministic).  Backward of nn.Linear as above. */
int gridmm_linear_planes_tn(const void* A_hi, const void* A_lo, int lda, const void* B_hi, const void* B_lo, int ldb,
                            float* C, float* workspace, int M, int N, int K, int splits, gridmm_stream_t stream);
/* The same, plus the bias gradient of that Linear: db [N] = column sums of A (= of dY), computed by the GEMM itself from the
 * planes (two more MFMAs per A tile and k-step against an all-ones operand; hi + lo = dY to 2^-17 relative, fp32 accumulate) --
 * one partial row per contraction range in db_ws (>= splits x N floats), summed in range order by the summing pass
 * (deterministic).  The split pass of dY need not produce column sums, and a producer that emits the planes of its dX
 * (LayerNorm / GELU backward, dropout) makes the split pass itself unnecessary.  db = NULL: exactly gridmm_linear_planes_tn. */
int gridmm_linear_planes_tn_db(const void* A_hi, const void* A_lo, int lda, const void* B_hi, const void* B_lo, int ldb,
                               float* C, float* workspace, int M, int N, int K, int splits, float* db_ws, float* db,
                               gridmm_stream_t stream);
/* n <= 8 such weight gradients (with their bias gradients) as at most two GEMM launches (one per tile class) + one summing launch
 * -- the six of a cross-modal layer's backward, the four of a BertLayer's: each member exactly as gridmm_linear_planes_tn_db
 * would compute it (same tiles, same ranges, same order of the sums).  probs: HOST array (read during the call). */
typedef struct {
  const void *A_hi, *A_lo; int lda;
  const void *B_hi, *B_lo; int ldb;
  float *C, *workspace;
  int M, N, K, splits;
  float *db_ws, *db;
} gridmm_tn_problem_t;
int gridmm_linear_planes_tn_grouped(const gridmm_tn_problem_t* probs, int n, gridmm_stream_t stream);
/* the number of ranges (1 .. 8) that fills the chip for this problem: what the library's own callers pass as `splits` */
int gridmm_linear_planes_tn_splits(int M, int N, int K);
/* X fp32 [M][C] -> row-major planes [Mp][ldp] with rows [M, Mp) zero [+ colsum as gridmm_transpose_split]: one pass per
 * activation / gradient for BOTH of its GEMM roles (forward or dX: first M rows; dW through gridmm_linear_planes_tn). */
int gridmm_split_rows_pad(const float* X, int ldx, void* R_hi, void* R_lo, int ldp, float* colsum, float* colsum_ws,
                          int M, int C, int Mp, gridmm_stream_t stream);

/* Backward of y = LayerNorm(X (+ R)) * gamma + beta (BertLayerNorm / nn.LayerNorm, vilmodel.py:33,131,147).
 *   dX [M][H] (same gradient flows to R); dgamma, dbeta [H]; workspace >= ceil(M/4) * 2 * H floats. */
int gridmm_layernorm_bwd(const float* X, int ldx, const float* R, int ldr, const float* gamma, float eps,
                         const float* dY, int ldy, float* dX, int lddx, float* dgamma, float* dbeta,
                         float* workspace, int M, int H, gridmm_stream_t stream);
/* (_planes: also the bf16 hi / lo planes of dX, rows of H contiguous -- the dY operand of the Linear in front of this LayerNorm
 * for its dX GEMM and its weight gradient: that Linear's split pass is skipped) */
int gridmm_layernorm_bwd_planes(const float* X, int ldx, const float* R, int ldr, const float* gamma, float eps,
                                const float* dY, int ldy, float* dX, int lddx, void* dX_hi, void* dX_lo, float* dgamma,
                                float* dbeta, float* workspace, int M, int H, gridmm_stream_t stream);

/* Training: y = LayerNorm(dropout(X) + R) and its backward, with the hidden-state dropout of BertSelfOutput / BertOutput
 * (vilmodel.py:160-170, 199-211: dense -> dropout -> LayerNorm(. + input)) applied inside the LayerNorm kernels:
 * keep(seed, row * H + col) ? x / (1 - p) : 0 -- the mask gridmm_dropout derives for the contiguous (M, H) tensor, so the
 * fused form equals gridmm_dropout followed by gridmm_layernorm bit for bit.  X, Y, dY, dX, dR contiguous (M, H);
 * dX = gradient of X (masked, rescaled), dR = gradient of R; workspace as gridmm_layernorm_bwd; seed_dev as gridmm_dropout. */
int gridmm_layernorm_dropout(const float* X, const float* R, int ldr, const float* gamma, const float* beta, float eps,
                             float* Y, float p, unsigned long long seed, const unsigned long long* seed_dev, int M, int H,
                             gridmm_stream_t stream);
/* (_planes: also the bf16 hi/lo planes (M, H) of Y, the next Linear's A operand and an operand of its TN weight gradient) */
int gridmm_layernorm_dropout_planes(const float* X, const float* R, int ldr, const float* gamma, const float* beta, float eps,
                                    float* Y, void* Y_hi, void* Y_lo, float p, unsigned long long seed,
                                    const unsigned long long* seed_dev, int M, int H, gridmm_stream_t stream);
int gridmm_layernorm_dropout_bwd(const float* X, const float* R, int ldr, const float* gamma, float eps, const float* dY,
                                 float* dX, float* dR, float* dgamma, float* dbeta, float* workspace, float p,
                                 unsigned long long seed, const unsigned long long* seed_dev, int M, int H,
                                 gridmm_stream_t stream);
int gridmm_layernorm_dropout_bwd_planes(const float* X, const float* R, int ldr, const float* gamma, float eps, const float* dY,
                                        float* dX, void* dX_hi, void* dX_lo, float* dR, float* dgamma, float* dbeta,
                                        float* workspace, float p, unsigned long long seed, const unsigned long long* seed_dev,
                                        int M, int H, gridmm_stream_t stream);   /* (planes of the masked dX, as above) */

/* Elementwise activations for training.  mode 0: out = gelu(X) (erf form, vilmodel.py:37-43);
 * 1: out = dY * gelu'(X); 2: out = relu(X); 3: out = dY * (X > 0).  n % 4 == 0, contiguous. */
int gridmm_activation(const float* X, const float* dY, float* out, int64_t n, int mode, gridmm_stream_t stream);
/* (_planes, forward modes only: also the bf16 hi/lo planes of `out`, same element order) */
int gridmm_activation_planes(const float* X, const float* dY, float* out, void* out_hi, void* out_lo, int64_t n, int mode,
                             gridmm_stream_t stream);

/* gridmm_attention that also returns lse [B][heads][Sqp] (Sqp = roundup(Sq,16)): log-sum-exp of the scaled,
 * masked scores per query -- the only statistic the backward needs. */
int gridmm_attention_train(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs, int k_rs,
                           const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, float* O,
                           int64_t o_bs, int o_rs, float* lse, int Sqp, int B, int heads, int Sq, int Sk,
                           float scale, float dropout_p, unsigned long long seed, const unsigned long long* seed_dev,
                           gridmm_stream_t stream);
/* (_planes: also the bf16 hi/lo planes of O, row stride p_rs / episode stride p_bs in elements) */
int gridmm_attention_train_planes(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs, int k_rs,
                                  const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, float* O,
                                  int64_t o_bs, int o_rs, void* O_hi, void* O_lo, int64_t p_bs, int p_rs, float* lse, int Sqp,
                                  int B, int heads, int Sq, int Sk, float scale, float dropout_p, unsigned long long seed,
                                  const unsigned long long* seed_dev, gridmm_stream_t stream);
/* dropout_p > 0: dropout on the attention probabilities (vilmodel.py:143,362; transformer.py MultiheadAttention):
 * element (b,h,q,k) is kept iff a counter-based hash of (seed, ((b*heads+h)*Sq+q)*Sk+k) >= p, survivors scaled by
 * 1/(1-p); the backward regenerates the mask from the same (dropout_p, seed).  seed_dev (device, may be NULL): a
 * second seed word read by the kernel at run time -- the part of the seed that changes between replays of a captured
 * (hipGraph) training step, whose kernel arguments are frozen. */

/* Backward of the attention core: dQ, dK, dV from dO (fp32, exact-fp32 MFMA; masked keys get zero gradient).
 * delta [B][heads][Sqp] is a workspace (sum_d dO*O per query). */
int gridmm_attention_bwd(const float* Q, int64_t q_bs, int q_rs, const float* K, int64_t k_bs, int k_rs,
                         const float* V, int64_t v_bs, int v_rs, const uint8_t* kmask, int mask_bs, const float* O,
                         int64_t o_bs, int o_rs, const float* dO, int64_t do_bs, int do_rs, const float* lse,
                         float* delta, float* dQ, int64_t dq_bs, int dq_rs, float* dK, int64_t dk_bs, int dk_rs,
                         float* dV, int64_t dv_bs, int dv_rs, int B, int heads, int Sq, int Sk, int Sqp, float scale,
                         float dropout_p, unsigned long long seed, const unsigned long long* seed_dev,
                         gridmm_stream_t stream);

/* ---- the same attention core on the bf16 matrix pipe (round 5; csrc/attention_train.hip) ---------------------------
 * Forward + backward of softmax(Q K^T scale + key mask) (with dropout on the probabilities) V for operands that arrive as
 * bf16 hi/lo PLANES (what the QKV / KV GEMMs of the differentiable path emit beside their fp32 result): the 3-term bf16
 * split of gridmm_attention_rows (K / V -- or Q / dO -- staged once per workgroup in LDS by a loader wave) instead of exact
 * fp32 on the f32 matrix pipe.  Same reference lines as gridmm_attention_train / gridmm_attention_bwd
 * (map_nav_src/models/vilmodel.py:95-157, 317-368; transformer.py:176-177), same dropout mask for the same seed.
 *   lse2 [B][heads][Sqp]: log2 sum_k 2^(s_k scale log2 e) per query (+1e30 for a fully masked row), Sqp = roundup(Sq,16)
 *   O fp32 and / or its planes (forward); dQ / dK / dV fp32 with their own strides (backward; every row of the three
 *   column blocks is written, masked keys get zero)
 *   workspace >= gridmm_attention_rows_bwd_workspace(B, heads, Sq) bytes: delta = <dO, O> (and <dO, vbar>) per query + the
 *   planes of dO
 *   vbar (may be NULL) [B][vb_bs]: the K / V planes are SHIFTED per episode -- K - K[row 0], V - V[row 0], what
 *   gridmm_linear_planes_shift writes -- and vbar + b * vb_bs holds the row V[row 0] of episode b (all heads, 16-byte aligned,
 *   vb_bs % 4 == 0).  Same function values and gradients (softmax is invariant to a shift of K; P V = P (V - v0) + (sum P) v0),
 *   without the common component of the rows in the bf16 products: the q / k weight gradients then sit at the error level of
 *   the exact-fp32 kernels (they were 10-15x above it without the shift, csrc/attention_train.hip).
 * Strides of the plane operands in elements, % 8 == 0; Sk <= 2048. */
int gridmm_attention_rows_train(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi, const void* K_lo,
                                int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo, int64_t v_bs, int v_rs,
                                const uint8_t* kmask, int mask_bs, float* O, int64_t o_bs, int o_rs, void* O_hi, void* O_lo,
                                int64_t p_bs, int p_rs, float* lse2, int Sqp, const float* vbar, int64_t vb_bs, int B, int heads,
                                int Sq, int Sk, float scale, float dropout_p, unsigned long long seed,
                                const unsigned long long* seed_dev, gridmm_stream_t stream);
size_t gridmm_attention_rows_bwd_workspace(int B, int heads, int Sq);
int gridmm_attention_rows_bwd(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi, const void* K_lo,
                              int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo, int64_t v_bs, int v_rs,
                              const uint8_t* kmask, int mask_bs, const float* O, int64_t o_bs, int o_rs, const float* dO,
                              int64_t do_bs, int do_rs, const float* lse2, const float* vbar, int64_t vb_bs, void* workspace,
                              size_t workspace_bytes, float* dQ,
                              int64_t dq_bs, int dq_rs, float* dK, int64_t dk_bs, int dk_rs, float* dV, int64_t dv_bs, int dv_rs,
                              int B, int heads, int Sq, int Sk, int Sqp, float scale, float dropout_p, unsigned long long seed,
                              const unsigned long long* seed_dev, gridmm_stream_t stream);
/* (_planes: dQ_hi / dQ_lo and dK_hi / dK_lo / dV_hi / dV_lo, each pair optional -- the bf16 planes of the gradients with the strides
 * of their fp32 tensors: the dY operand of the q / k / v projection's backward, whose split pass is then skipped) */
int gridmm_attention_rows_bwd_planes(const void* Q_hi, const void* Q_lo, int64_t q_bs, int q_rs, const void* K_hi, const void* K_lo,
                                     int64_t k_bs, int k_rs, const void* V_hi, const void* V_lo, int64_t v_bs, int v_rs,
                                     const uint8_t* kmask, int mask_bs, const float* O, int64_t o_bs, int o_rs, const float* dO,
                                     int64_t do_bs, int do_rs, const float* lse2, const float* vbar, int64_t vb_bs, void* workspace,
                                     size_t workspace_bytes, float* dQ, int64_t dq_bs, int dq_rs, float* dK, int64_t dk_bs,
                                     int dk_rs, float* dV, int64_t dv_bs, int dv_rs, void* dQ_hi, void* dQ_lo, void* dK_hi,
                                     void* dK_lo, void* dV_hi, void* dV_lo, int B, int heads, int Sq, int Sk, int Sqp, float scale,
                                     float dropout_p, unsigned long long seed, const unsigned long long* seed_dev,
                                     gridmm_stream_t stream);

/* Backward of gridmm_grid_aggregate w.r.t. text = text_proj(txt_embeds) (vilmodel.py:795-807; the gradient
 * reaches text_proj and the language encoder through the max / softmax weights):
 *   relevance [B][cap] as written by the forward (by sorted position), text [B][L][D] f32, dcells [B][196][D] f32 (gradient of the
 *   reduced cell vectors, i.e. after grid_proj's own backward) -> dtext [B][L][D] f32.
 *   da_ws [B][cap] f32 and amax_ws [B][cap] int32 are workspaces. */
int gridmm_grid_aggregate_bwd(const void* slab, const int32_t* perm, const int32_t* cell_start,
                              const float* relevance, const float* text, const float* dcells, float* dtext,
                              float* da_ws, int32_t* amax_ws, int B, int cap, int D, int L, gridmm_stream_t stream);

/* The same gradient from the forward's routing (gridmm_grid_aggregate_train): streaming passes, no search, no atomics,
 * deterministic.  relevance / amax [B][cap] by sorted position; da_ws, dw_ws [B][cap] f32 workspaces; part_ws (may be NULL:
 * the per-token form of round 3) >= gridmm_grid_aggregate_bwd_workspace(B, D, L) bytes: the point-balanced gather of round 5
 * (chunk tables summed in chunk order). */
size_t gridmm_grid_aggregate_bwd_workspace(int B, int D, int L);
int gridmm_grid_aggregate_bwd_routed(const void* slab, const int32_t* perm, const int32_t* cell_start,
                                     const float* relevance, const int32_t* amax, const float* dcells, float* dtext,
                                     float* da_ws, float* dw_ws, float* part_ws, int B, int cap, int D, int L,
                                     gridmm_stream_t stream);

/* Backward of gridmm_fuse_logits (vilmodel.py:859-899) -- SURVEY.md 8b's gridmm_fuse_logits_bwd: gradients of the four
 * logit sets (any of them NULL = zero) -> gradients of the three raw head outputs and of the pre-sigmoid fusion weight
 * (d_fuse_raw [B], NULL when fuse_raw is NULL).  Masked positions pass no gradient; deterministic (fixed-order sums). */
int gridmm_fuse_logits_bwd(const float* g_raw, const float* l_raw, const float* fuse_raw, const uint8_t* gmap_masks,
                           const uint8_t* gmap_visited, const uint8_t* vp_nav_masks, const int32_t* cand_of_node,
                           const uint8_t* cand_visited, const float* d_global, const float* d_local, const float* d_grid,
                           const float* d_fused, float* d_g_raw, float* d_l_raw, float* d_grid_raw, float* d_fuse_raw,
                           int B, int G, int V, gridmm_stream_t stream);

/* Backward of the cell compaction (gridmm_cells_compact / vilmodel.py:813-823): d_out rows [0,196) of a [B][.][H] buffer
 * (batch stride d_out_bs elements) -> d_cells [B][196][H] in cell order (zeros for empty cells). */
int gridmm_cells_compact_bwd(const float* d_out, int64_t d_out_bs, const uint8_t* occ, float* d_cells, int B, int H,
                             gridmm_stream_t stream);

/* Hidden-state dropout (vilmodel.py:86,166,205; transformer.py:179-181): y[i] = keep_i ? x[i] / (1 - p) : 0 with keep_i
 * from the counter-based hash of (seed [+ *seed_dev], i) that the attention dropout uses; the backward is the same call
 * on the gradient.  n % 4 == 0, n < 2^32; seed_dev (device, may be NULL): per-replay seed word of captured steps. */
int gridmm_dropout(const float* x, float* y, int64_t n, float p, unsigned long long seed,
                   const unsigned long long* seed_dev, gridmm_stream_t stream);

/* Optimizer step: gradient-norm clipping + AdamW without a host round trip.
 * gridmm_grad_sumsq adds sum(g^2) of one gradient tensor into *acc (zero it first; call once per tensor).
 * gridmm_adamw_step updates one parameter tensor in place; with sumsq != NULL the gradient is first scaled by
 * min(1, max_norm / (sqrt(*sumsq) + 1e-6)) (torch.nn.utils.clip_grad_norm_).  step_size = lr * sqrt(1-b2^t)/(1-b1^t)
 * is computed by the caller.  decay_first = 0: pretrain_src/optim/adamw.py:56-112 (decay after the update);
 * decay_first = 1: torch.optim.AdamW order (fine-tune, agent_base.py:131).  dtype 0 = fp32, 1 = fp16 (the
 * reference's fp16 grid_proj keeps fp16 optimizer state).  dyn (device float[3], may be NULL): lr, step_size, eps read
 * at run time instead of the arguments -- for captured (hipGraph) steps, whose kernel arguments are frozen. */
int gridmm_grad_sumsq(const void* g, int64_t n, int dtype, float* acc, gridmm_stream_t stream);
int gridmm_adamw_step(void* p, const void* g, void* m, void* v, int64_t n, int dtype, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float step_size, int decay_first,
                      const float* sumsq, float max_norm, const float* dyn, gridmm_stream_t stream);

/* C (fp32, M x N, contiguous) = A W^T like gridmm_linear_planes, with the contraction split over `splits` (2..64)
 * workgroups per output tile; partial tiles go to `workspace` (splits * M * N floats) and are summed in a fixed order
 * (deterministic; no bias / activation / planes).  For the weight-gradient GEMMs dW = dY^T X: small output,
 * contraction over all rows of the batch. */
int gridmm_linear_planes_splitk(const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo,
                                int Kp, float* C, float* workspace, int M, int N, int K, int splits,
                                gridmm_stream_t stream);

/* Multi-tensor forms of the two kernels above: ONE launch over all parameters.
 *   desc        device array of n_tensors records {void* p; const void* g; void* m; void* v; int64 n;
 *               float lr, step_size, eps, weight_decay; int32 dtype (0 = fp32, 1 = fp16: parameter, gradient and both
 *               moments alike); int32 pad;}  (64 bytes, natural C layout).  The scalars are read from the record at run
 *               time, so a captured (hipGraph) step advances lr / bias correction by rewriting the table.
 *   chunk_first device int32 [n_tensors + 1]: prefix sums of ceil(n / 16384); n_chunks = chunk_first[n_tensors]
 * gridmm_multi_grad_sumsq: partial = n_chunks-float workspace (one partial per chunk, summed in a fixed order: the norm
 * is bit-reproducible), out = the global sum of squares (device scalar). */
int gridmm_multi_grad_sumsq(const void* desc, const int* chunk_first, int n_tensors, int n_chunks, float* partial,
                            float* out, gridmm_stream_t stream);
/* planes (device, may be NULL): n_tensors records {void *hi, *lo, *thi, *tlo; int32 N, K, ldw, ldt} (48 bytes).  A tensor whose
 * record has hi != NULL is a contiguous fp32 [N][K] weight (N % 64 == 0, K % 64 == 0) whose bf16 hi / lo planes the update
 * writes itself: row planes [N][ldw] at hi / lo and the planes of W^T [K][ldt] at thi / tlo -- what the forward and dX GEMMs
 * of the differentiable path read (replaces one gridmm_transpose_split launch per weight and step).  Same update arithmetic. */
int gridmm_multi_adamw_step(const void* desc, const int* chunk_first, int n_tensors, int n_chunks, float beta1,
                            float beta2, int decay_first, const float* sumsq, float max_norm, const void* planes,
                            gridmm_stream_t stream);

/* Linear layers with K <= 16 input features on the differentiable path (position / angle embeddings: loc_fts K = 7,
 * gmap_pos_fts / vp_pos_fts K = 7 / 14; map_nav_src/models/vilmodel.py:454-470, 538-552, 640-655) in plain fp32 FMAs:
 *   gridmm_linear_skinny      Y [M][N] (row stride ldy) = X [M][K] (row stride ldx) W^T + bias; W fp32 [N][K] contiguous, N % 4 == 0
 *   gridmm_linear_skinny_bwd  dW [N][K] = dY^T X, db [N] = column sums of dY (either may be NULL); one partial per 64 rows in
 *                             `workspace` (gridmm_linear_skinny_bwd_workspace bytes), summed in order (deterministic). */
int gridmm_linear_skinny(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int K,
                         gridmm_stream_t stream);
size_t gridmm_linear_skinny_bwd_workspace(int M, int N, int K);
int gridmm_linear_skinny_bwd(const float* dY, int ldy, const float* X, int ldx, float* dW, float* db, float* workspace, int M,
                             int N, int K, gridmm_stream_t stream);

/* Linear layers with ONE output feature on the differentiable path (the last Linear of the heads' ClsPrediction, H -> 1;
 * map_nav_src/models/vilmodel.py:437-446): Y [M] = X [M][K] w + bias[0] (K % 4 == 0), and its backward dX [M][K] = dY[m] w (NULL:
 * skipped), dw [K] = sum_m dY[m] X[m][:], db [1] = sum_m dY[m] (one partial per 64 rows in `workspace`,
 * gridmm_rowdot_bwd_workspace bytes, summed in order). */
int gridmm_rowdot(const float* X, int ldx, const float* w, const float* bias, float* Y, int M, int K, gridmm_stream_t stream);
size_t gridmm_rowdot_bwd_workspace(int M, int K);
int gridmm_rowdot_bwd(const float* dY, const float* X, int ldx, const float* w, float* dX, int lddx, float* dw, float* db,
                      float* workspace, int M, int K, gridmm_stream_t stream);

/* Gradient accumulation of a multi-step backward as ONE launch (fine-tuning: one backward through the 7 .. 15 navigation
 * steps of a rollout, map_nav_src/r2r/agent_base.py:190-199 -- under torch autograd every step's gradient of every parameter
 * is added by its own launch).  desc: device array of n_tensors records {float* dst; const float* src[7]; int64 n;
 * int32 n_src, pad;} (80 bytes); dst[i] = ((dst[i] + src[0][i]) + src[1][i]) + ... over n_src <= 7 sources, fp32, in list
 * order (bit-identical to the sequential in-place adds).  chunk_first / n_chunks as above. */
int gridmm_multi_grad_accumulate(const void* desc, const int* chunk_first, int n_tensors, int n_chunks,
                                 gridmm_stream_t stream);

/* ---- one cross-modal layer of the DIFFERENTIABLE path: forward that keeps what the backward needs + the whole backward
 * of the layer as ONE call (gridmm_xattn_layer_bwd).  GraphLXRTXLayer.forward with graph_sprels = None
 * (map_nav_src/models/vilmodel.py:399-414, pretrain_src/model/vilmodel.py:404-415) under torch autograd in the reference
 * (agent_base.py:199 / train_r2r.py:262).  Attention on the bf16 matrix pipe (gridmm_attention_rows_train / _bwd: the self
 * attention always, the cross attention when the planes KV_hi / KV_lo of the context projections are given -- NULL: the exact-
 * fp32 kernels gridmm_attention_train / _bwd read KV; KV_shift != NULL: the planes are the shifted ones of
 * gridmm_linear_planes_shift and KV_shift + b * kv_shift_bs is row 0 of episode b's projection, same column layout as KV),
 * bf16x3 GEMMs; hidden-state
 * dropout inside the LayerNorm kernels (p_hidden), attention-probability dropout inside the attention kernels (p_attn);
 * seed[0..4] = cross-attention probabilities, cross LayerNorm, self-attention probabilities, self LayerNorm, FFN LayerNorm
 * (+ seed_dev, the per-replay word of a captured step).  The kernels, their order and their tile choices are those of the
 * op-by-op path (gridmm_amd/autograd.py): outputs and gradients are bit-identical to it.
 *   X   [B*Sq][H] fp32 tokens; KV [B][Sk][.] fp32 context projections (K at k_col, V at v_col; row stride kv_rs, episode
 *       stride kv_bs, in elements); masks as in gridmm_attention; Y [B*Sq][H] fp32 out.
 *   saved      caller-provided block (>= gridmm_xattn_layer_train_saved_bytes) the forward fills and the backward reads
 *   workspace  >= gridmm_xattn_layer_train_workspace bytes (both calls), 256-byte aligned
 *   backward:  dY [B*Sq][H] -> dX [B*Sq][H], dKV (K / V gradients written at k_col / v_col of a buffer with strides dkv_bs /
 *              dkv_rs; other columns untouched), and the parameter gradients of gridmm_xlayer_grads_t (dense [N][K] / [N] /
 *              [H] fp32, overwritten). */
typedef struct {
  const void *w_hi, *w_lo; int Kp;      /* planes of W  [N][Kp]  (forward GEMM) */
  const void *wt_hi, *wt_lo; int Np;    /* planes of W^T [K][Np] (dX GEMM; only the backward reads them) */
  const float* bias; int N, K;
} gridmm_linear_train_t;
typedef struct {
  gridmm_linear_train_t xq, xo, sqkv, so, ffn_i, ffn_o;
  gridmm_ln_t x_ln, s_ln, f_ln;
  float p_hidden, p_attn;
  unsigned long long seed[5];
  const unsigned long long* seed_dev;
  int attention_fp32;            /* != 0: both attentions on the exact-fp32 kernels (A / B against the bf16 matrix-pipe form) */
} gridmm_xlayer_train_t;
typedef struct {
  float *xq_w, *xq_b, *xo_w, *xo_b, *sqkv_w, *sqkv_b, *so_w, *so_b, *ffn_i_w, *ffn_i_b, *ffn_o_w, *ffn_o_b;
  float *x_ln_g, *x_ln_b, *s_ln_g, *s_ln_b, *f_ln_g, *f_ln_b;
} gridmm_xlayer_grads_t;
/* KV == NULL (forward and backward alike): a BertLayer -- self attention + feed forward only (vilmodel.py:214-231: the layers of
 * the text and panorama encoders); the xq / xo / x_ln members of L, the context's KV / plane / mask arguments and dKV are
 * not read / written, Sk is ignored. */
size_t gridmm_xattn_layer_train_saved_bytes(int B, int Sq, int H, int I);
size_t gridmm_xattn_layer_train_workspace(int B, int Sq, int H, int I);
int gridmm_xattn_layer_train_fwd(const gridmm_xlayer_train_t* L, const float* X, const float* KV, const void* KV_hi,
                                 const void* KV_lo, const float* KV_shift, int64_t kv_shift_bs, int64_t kv_bs, int kv_rs,
                                 int k_col, int v_col, const uint8_t* ctx_mask, int ctx_mask_bs, const uint8_t* self_mask,
                                 int self_mask_bs, float* Y, void* saved, size_t saved_bytes, void* workspace,
                                 size_t workspace_bytes, int B, int Sq, int Sk, int heads, gridmm_stream_t stream);
int gridmm_xattn_layer_bwd(const gridmm_xlayer_train_t* L, const float* X, const float* KV, const void* KV_hi, const void* KV_lo,
                           const float* KV_shift, int64_t kv_shift_bs, int64_t kv_bs, int kv_rs,
                           int k_col, int v_col, const uint8_t* ctx_mask, int ctx_mask_bs, const uint8_t* self_mask,
                           int self_mask_bs, const void* saved, size_t saved_bytes, const float* dY, float* dX, float* dKV,
                           int64_t dkv_bs, int dkv_rs, const gridmm_xlayer_grads_t* G, void* workspace,
                           size_t workspace_bytes, int B, int Sq, int Sk, int heads, gridmm_stream_t stream);

/* ---- one PRE-LayerNorm transformer layer of the differentiable path (the panorama encoder's and the grid encoder's layers:
 * TransformerEncoderLayer.forward_pre, map_nav_src/models/transformer.py:170-182 via models/ops.py:11-16), forward and whole
 * backward as one C call each:  x1 = x + drop(out_proj(attention(in_proj(LN1(x)))));
 *                               y  = x1 + drop(linear2(drop(gelu(linear1(LN2(x1)))))).
 * p: the layer's dropout probability (attention probabilities and the three hidden-state dropouts; 0 = none), seed[0..3] =
 * attention, after out_proj, after the activation, after linear2 (seed_dev as gridmm_dropout).  X, Y, dY, dX contiguous
 * (B, S, H); mask [B][mask_bs] (1 = key is valid; NULL = all).  saved / workspace: caller-provided blocks of at least
 * gridmm_preln_layer_saved_bytes / _workspace bytes (saved: written by the forward, read by the backward).  The same kernels in
 * the same order as the op-by-op autograd form (gridmm_amd.vilmodel_train.pre_ln_encoder): bit-identical results. */
typedef struct {
  gridmm_linear_train_t qkv, out, ffn1, ffn2;
  gridmm_ln_t ln1, ln2;
  float p;
  unsigned long long seed[4];
  const unsigned long long* seed_dev;
} gridmm_preln_layer_t;
typedef struct {
  float *qkv_w, *qkv_b, *out_w, *out_b, *ffn1_w, *ffn1_b, *ffn2_w, *ffn2_b, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
} gridmm_preln_grads_t;
size_t gridmm_preln_layer_saved_bytes(int B, int S, int H, int I);
size_t gridmm_preln_layer_workspace(int B, int S, int H, int I);
int gridmm_preln_layer_train_fwd(const gridmm_preln_layer_t* L, const float* X, const uint8_t* mask, int mask_bs, float* Y,
                                 void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, int B, int S,
                                 int heads, gridmm_stream_t stream);
int gridmm_preln_layer_bwd(const gridmm_preln_layer_t* L, const float* X, const uint8_t* mask, int mask_bs, const void* saved,
                           size_t saved_bytes, const float* dY, float* dX, const gridmm_preln_grads_t* G, void* workspace,
                           size_t workspace_bytes, int B, int S, int heads, gridmm_stream_t stream);
/* y = r + dropout(x) in one pass (p = 0: y = r + x; r = NULL: y = dropout(x); mask as gridmm_dropout); y_hi / y_lo (optional,
 * then y may be NULL): the bf16 planes of the result.  Two roundings (product, then sum): equal to gridmm_dropout followed
 * by an fp32 add bit for bit.  n % 4 == 0, n < 2^32. */
int gridmm_dropout_add(const float* x, const float* r, float* y, void* y_hi, void* y_lo, int64_t n, float p,
                       unsigned long long seed, const unsigned long long* seed_dev, gridmm_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Host-side helpers of the agent loop's collation (HOST pointers, no device work, no stream)
 * ---------------------------------------------------------------------------------------- */

/* Number of legs of the route from cur[b] to every target node of episode b's topological map: len(FloydGraph.path(x, y))
 * (map_nav_src/models/graph_utils.py:75-100: path(x, y) = [y] without a pivot, else path(x, k) + path(k, y)), evaluated on
 * the current pivot matrix like the reference's recursion; feeds the "hops" column of get_pos_fts
 * (graph_utils.py:139-151, agent.py:96-147).
 *   via [B][cap][cap] int32 pivots (-1: direct / unknown); cur [B] int64; tgt [B][T] int64; mask [B][T] uint8 or NULL
 *   (0: skipped, hops = 1); hops [B][T] float64 out (0 for tgt == cur). */
int gridmm_route_lengths(const int32_t* via, int B, int cap, const int64_t* cur, const int64_t* tgt, const uint8_t* mask,
                         int T, double* hops);

/* Navigation-input collation of a lock-step batch from the (B, cap, ...) arrays of its topological maps: what
 * _nav_gmap_variable / _nav_vp_variable (map_nav_src/r2r/agent.py:96-205: node order [visited | unvisited], get_pos_fts of the
 * graph nodes / candidates / start node, pair distances, step ids, visited and length masks) and the fused-logit loops of
 * models/vilmodel.py:881-899 compute per episode in Python.  *_plan orders the nodes and returns the longest sequence (the
 * caller picks the padded node axis G >= 1 + that); *_fill writes every host-built array of the step.  Array shapes: see
 * csrc/hostutil.hip.  *_fill returns 0, GRIDMM_EINVAL, or 1 + (b * G + row) of the first graph node without an embedding. */
int gridmm_collate_nav_plan(const uint8_t* seen, const int64_t* n, const int64_t* cur, int B, int cap, int enc_full_graph,
                            int act_visited_nodes, int64_t* order, int64_t* m, int64_t* n_vis, int64_t* n_unv,
                            uint8_t* seen_eff);
int gridmm_collate_nav_fill(const double* pos, const double* dist, const int32_t* via, const int64_t* step,
                            const int64_t* order, const int64_t* m, const int64_t* n_vis, const uint8_t* seen_eff,
                            const int64_t* cur, const int64_t* start, const int64_t* cid, const int64_t* nc,
                            const double* heading, const double* elevation, const int32_t* cnt, int B, int cap, int Cw,
                            int slots, int G, int V1, int afs, int enc_full_graph, float* gpos, float* vpos, float* pair,
                            int64_t* steps, uint8_t* visited, int64_t* slot, float* inv, uint8_t* gmask, int32_t* cand_of_node,
                            uint8_t* cand_visited);

#ifdef GRIDMM_DEBUG_HOOKS
/* ---- development build only (gridmm_amd/csrc: `make debug` -> libgridmm_hip_dbg.so) -------------------------------
 * Process-global tuning hooks for tools/sweep_gemm_cfg_step.py / sweep_gemm_cfg_train.py; the shipping library neither
 * exports them nor holds the state behind them (tests/test_capi_symbols.py asserts that no gridmm_debug_* symbol exists
 * in libgridmm_hip.so). */
/* Force tile configuration `cfg` (0 = back to the heuristic) for the problem shape (M, N, K) in this process. */
int gridmm_debug_gemm_cfg_override(int M, int N, int K, int cfg);
/* The problem shapes the tile heuristic was asked about since recording started: rows (M, N, K, chosen cfg, calls) into
 * out[max_rows][5]; returns the number of rows.  log = 1 starts (and clears) the recording, 0 stops it, -1 leaves it. */
int gridmm_debug_gemm_shapes(int* out, int max_rows, int log);
/* The same for gridmm_attention_rows: configuration of the calls with more than / at most four 16-query tiles. */
int gridmm_debug_attention_cfg_override(int cfg_big, int cfg_small);
/* gridmm_linear_planes_map with an explicit tile configuration out of the whole experiment table. */
int gridmm_debug_linear_planes_map_cfg(const void* A_hi, const void* A_lo, int lda, int a_rpb, int64_t a_bs, const void* W_hi,
                                       const void* W_lo, int Kp, int w_layout, const float* bias, const float* residual,
                                       int ldr, float* C, int ldc, void* C_hi, void* C_lo, int ldp, int M, int N, int K,
                                       int act, int cfg, gridmm_stream_t stream);
#endif

#ifdef __cplusplus
}
#endif
#endif /* GRIDMM_H */
