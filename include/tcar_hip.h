/*
 * tcar_hip.h — C-ABI of the MI355X (gfx950) TCAR hot path.
 *
 * Drop-in boundary.  The reference has ONE device boundary per training step: the
 * `sess.run([loss, global_step, train_op], feed_dict)` call at model_combine.py:231 (and
 * `sess.run([softmax_input, cross_loss])` at model_combine.py:283 for evaluation); everything behind it is
 * the TensorFlow op list built by model_combine.py:52-163 from modules.py:13-152 and util.py:59-100.
 * Each entry point below replaces the TF ops named in its comment.  The reference is Python, so the
 * binding a maintainer adds is a ctypes stub; the library itself has no Python or torch
 * dependency.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked "host"; no entry point allocates;
 *   - `stream` is a hipStream_t passed as void*; calls are stream-ordered and re-entrant; the library keeps no mutable
 *     state: per-device kernel attributes are idempotent (set once per device, thread-safe), the diagnostic TCAR_*
 *     environment switches are read once per process and never written (tcar_tuning_t), and everything a step carries
 *     from one call to the next lives in buffers of the caller's tcar_ctx_t;
 *   - return value: 0 = ok, negative = TCAR_E_* (never throws across the boundary);
 *   - device layout ("padded-concat space"): H is padded to ldh, Ht to ldt (multiples of 64, zero filled);
 *       ic = 2*ldh   item | content           (model_combine.py:111  seq_item_cont)
 *       pt = 5*ldt   month|day|week|hour|min  (model_combine.py:84   seq_publish_t)
 *       ct = 2*ldt   week | hour              (model_combine.py:94-97 click_t)
 *       ek = ic + pt                          (model_combine.py:132,136 attout / items_emb)
 *     The candidate matrix E [N, ek] (row n = item id n+1) holds, in place, the trainable item table
 *     (cols 0..H), the frozen content table (cols ldh..ldh+H) and the clipped candidate time vectors
 *     (cols ic..ek) — items_emb of model_combine.py:135-136 without the per-step concat;
 *   - item ids in `seq` are 1-based (sampler.py:68), labels and negatives 0-based (sampler.py:69,99).
 */
#ifndef TCAR_HIP_H
#define TCAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCAR_OK 0
#define TCAR_E_ARG (-1)     /* bad argument (alignment, range, unsupported size) */
#define TCAR_E_LAUNCH (-2)  /* hip launch error */

/* Diagnostic tuning switches.  The process-wide values are the shipped defaults overridden by the TCAR_* environment
 * variables of the same names, read ONCE per process and immutable afterwards (README.md).  A caller that wants other values
 * for ONE context (tcar_ctx_t.tune) or ONE call (the *_tuned entry points) passes its own copy: the library keeps no
 * mutable switch state.  Field <-> variable: lower-case name without the TCAR_ prefix.  Twelve switches (round 6; the eleven others of
 * round 5 — rest-pass grid, softmax variant, gather workgroups, small-GEMM ring, projection / output-transform splits, fork delay,
 * in-kernel waits, fused query backward, fused column sums, 16x16x32 logits — had settled A/Bs and are constants of the build now). */
typedef struct {
  int32_t bf16_tile;        /* TCAR_BF16_TILE      force a workgroup tile of the bf16 GEMMs (0 = heuristic; tests pin every tile shape with it): 128 / 192 / 193 /
                                                   256 / 384 / 512 = tiles of tcar_gemm_bf16 and the one-hot dX; 1922 / 1923 / 1283 / 2562 = codes of the dE (q, z)
                                                   launcher only (192-row double buffer / 192-row three-stage ring / 128-row ring / 256-row double buffer) */
  int32_t bf16_ks;          /* TCAR_BF16_KS        LDS stages of the hi-only bf16 GEMM: 1 = 32-deep double buffer, 2 = 64-deep for the dX / logits layouts (default),
                                                   3 = 64-deep everywhere, 4 = three-stage ring of 32-deep stages for the 256 x 128 dX tile (round 6) */
  int32_t wgrad_ks;         /* TCAR_WGRAD_KS       K chunk of the weight-gradient split */
  int32_t gather_big_rows;  /* TCAR_GATHER_BIG_ROWS  session rows from which the forward gather runs its throughput form */
  int32_t mha_mfma;         /* TCAR_MHA_MFMA       0: multihead_attention core always in its scalar form */
  int32_t sort_scatter;     /* TCAR_SORT_SCATTER   0: item-row scatter with float atomics instead of the sorted segmented sum */
  int32_t det_small;        /* TCAR_DET_SMALL      0: position / time / dwell table gradients through LDS + float atomics (sorted mode) */
  int32_t fused_ce;         /* TCAR_FUSED_CE       0: training steps materialise the fp32 logits and run the row-resident softmax kernel;
                                                   1: softmax epilogue with group maxima + the rescale pass over the plane (tcar_ce_finish);
                                                   2 (default): ANCHORED epilogue where the step's form allows it (tcar_ce_anchor_fold: no
                                                   pass over the plane), else as 1 */
  int32_t onehot_time;      /* TCAR_ONEHOT_TIME    0: the scoring GEMMs of a training step contract the 5 ldt clipped candidate time columns instead of the 160-column one-hot form */
  int32_t flag_fork;        /* TCAR_FLAG_FORK      mask over the fork slots: 0 = every fork of the main stream records an event (6-7 us of
                                                   bubble on it) instead of letting the producing kernel publish a device flag a polling
                                                   kernel of the side stream waits for */
  int32_t ce_fold;          /* TCAR_CE_FOLD        w > 0 (default 1024): tcar_ce_finish as ONE launch of about w workgroups — each folds the (max, sum)
                                                   pairs of its own 16 session rows, then rescales its slice of the plane — instead of a combine
                                                   launch + a rescale launch (0); the same bits either way */
  int32_t proj_split_rows;  /* TCAR_PROJ_SPLIT_ROWS  the session-side projections take their split-K slab form only for batches of at most this many
                                                   rows B * T: the form trades one global round trip per workgroup for 12 slabs of [B*T, ldh] fp32 that
                                                   the pool kernel folds — right for the latency-bound short buckets, 5 .. 17 % slower per step from
                                                   T = 5 up (profiles/r05_ab_experiments.txt) */
} tcar_tuning_t;
/* *out = the process-wide values (shipped defaults + TCAR_* environment) */
int tcar_tuning_defaults(tcar_tuning_t* out /*host*/);
/* sets the switch called `name` ("TCAR_BF16_TILE", ...) in the CALLER's copy; returns the previous value, INT_MIN for an unknown name */
int tcar_tuning_set(tcar_tuning_t* t /*host*/, const char* name /*host*/, int value);

#define TCAR_POS_VOCAB 40   /* model_combine.py:57 */
#define TCAR_DUR_VOCAB 11   /* model_combine.py:106 */
#define TCAR_NSLOT 32       /* squared-norm slots (one per trainable variable, 23 used) */

/* vocabularies of the five time tables: model_combine.py:73-81 */
static const int32_t TCAR_TIME_VOCAB[5] = {13, 32, 8, 25, 61};

typedef struct {
  int32_t n_items;  /* N  = len(item_dict)                         main.py:25 */
  int32_t H;        /* --hidden_size                               main.py:109 */
  int32_t Ht;       /* --time_hidden_size                          main.py:110 */
  int32_t ldh, ldt; /* padded H / Ht (multiples of 64)             */
} tcar_dims_t;

/* Parameter tables read by the lookups (model_combine.py:54-107).  Small tables are [vocab, ld*]. */
typedef struct {
  const float* E;        /* [N, ek]   item | content | cand-time (see header comment) */
  const float* pos;      /* [40, ldh] dec_pos                       model_combine.py:57 */
  const float* time[5];  /* month, day, week, hour, minute          model_combine.py:73-81 */
  const float* dur;      /* [11, ldt] duration_embedding            model_combine.py:106 */
} tcar_tables_t;

/* Gradient accumulators that mirror tcar_tables_t (fp32, caller zeroes g_pos/g_time/g_dur/sqn per step;
 * g_time[0..4] and g_dur MUST be one contiguous block in this order). */
typedef struct {
  float* g_item;         /* [N, ldh]  dense item-table gradient (row n = item id n+1) */
  float* g_pos;          /* [40, ldh] */
  float* g_time[5];
  float* g_dur;
  float* sqn;            /* [TCAR_NSLOT] IndexedSlices value-norm^2 pieces, slots below */
  int32_t slot_item, slot_pos, slot_time[5], slot_dur;
  float* rows_out;       /* NULL, or [B*T, ldh]: tcar_gather_clip_bwd WRITES each item-row gradient here instead
                            of adding it into g_item (data-parallel path: rows are all-gathered, then applied on
                            every rank with tcar_scatter_add_rows) */
  float* norms_out;      /* NULL, or [B*T]: with it the squared norm of each item-row gradient is WRITTEN here instead of
                            being added (atomically) into sqn[slot_item] — summed later in a fixed order (segsum) */
  int64_t rows_ld;       /* row stride of rows_out in floats (0 = ldh): a packed exchange buffer keeps the row's id beside it */
  int32_t skip_small;    /* != 0: tcar_gather_clip_bwd leaves the position / time / dwell tables and the click rows to
                            tcar_small_tables_bwd_det (order-fixed) and handles the item rows only */
} tcar_grads_t;

/* One mini-batch = the feed_dict of model_combine.py:214-227 (int32, row-major). */
typedef struct {
  int32_t B, T, K;
  const int32_t* seq;     /* [B,T] 1-based item ids            input_seq */
  const int32_t* pub[5];  /* [B,T] month, day, week, hour+1, minute+1   input_publish_* */
  const int32_t* cw;      /* [B]   click isoweekday-1           input_click_week */
  const int32_t* ch;      /* [B]   click hour                   input_click_hour */
  const int32_t* gap;     /* [B,T] dwell bucket 0..11           active_interval */
  const int32_t* label;   /* [B]   0-based                      label */
  const int32_t* neg;     /* [B,K] 0-based or NULL              label_neg */
} tcar_batch_t;

/* ---- embedding lookups --------------------------------------------------------------------------------
 * tcar_gather_clip_fwd: the 8 session-side + 2 click-side `embedding_lookup(..., max_norm=1)` calls of
 * modules.py:36 / model_combine.py:54-82,94-97,106 fused with the concats of :65,84,94,111.
 *   x_icp [B*T, ic] = clip(item)+clip(pos) | clip(content);  x_pt [B*T, pt];  x_act [B*T, ldt];
 *   click_t [B, ct].  Dwell id >= 11 yields a zero row. */
int tcar_gather_clip_fwd(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt,
                         float* x_icp, float* x_pt, float* x_act, float* click_t, void* stream);
/* ..._tuned: with the caller's switch values (gather_big_rows / gather_wg_per_cu choose between the latency form and the
 * throughput form; both give the same bits) */
int tcar_gather_clip_fwd_tuned(const tcar_tuning_t* tune, const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt,
                               float* x_icp, float* x_pt, float* x_act, float* click_t, void* stream);
/* Both layers of the click-query MLP in one launch (modules.py:138-139): q1 [B, ldh] = relu(click_t Wq1 + b1), q [B, 2 ldh] =
 * tanh(q1 Wq2 + b2), fp32 FMAs in a fixed order.  Only for ldh == 256 and ldt == 64 (the reference's hidden sizes, padded);
 * TCAR_E_ARG otherwise — run the layers through tcar_gemm_*_grouped then. */
int tcar_query_mlp(const tcar_dims_t* d, int B, const float* click_t, const float* q1_w, const float* q1_b, const float* q2_w,
                   const float* q2_b, float* q1, float* q, void* stream);

/* The input-gradient half of the click-query MLP's backward pass in one launch (modules.py:138-139): dq1 [B, ldh] = (dq Wq2^T) *
 * relu'(q1) and dclick [B, 2 ldt] = dq1 Wq1^T, fp32 FMAs in a fixed order; dq [B, 2 ldh] arrives through tanh' already (the pool
 * backward applies it).  dq == NULL: dq1 is an INPUT and only dclick is computed (q1, q2_w unused) — the form the fused step
 * runs on its aux stream in front of the small tables' pass.  Same restriction as tcar_query_mlp (ldh == 256, ldt == 64). */
int tcar_query_mlp_bwd(const tcar_dims_t* d, int B, const float* dq, const float* q1, const float* q1_w, const float* q2_w, float* dq1,
                       float* dclick, void* stream);

/* tcar_gather_clip_bwd: gradient of the above w.r.t. the tables (through the norm clip), i.e. the
 * IndexedSlices that tf.gradients builds for model_combine.py:156.  Adds into `g` (atomics) and adds
 * sum ||row-gradient||^2 into g->sqn[slot] for every gathered row. */
int tcar_gather_clip_bwd(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt,
                         const float* dx_icp, const float* dx_pt, const float* dx_act, const float* dclick,
                         const tcar_grads_t* g, void* stream);

/* Order-fixed session-side backward of the SMALL tables (position, month..minute, dwell; the click rows of the week / hour
 * tables): one workgroup per destination row sums its sources in source order (the clip Jacobian is applied once per row,
 * from S = sum gy, Q = sum ||gy||^2, D2 = sum (x.gy)^2), one add per gradient element and per norm slot.  Same values as
 * tcar_gather_clip_bwd up to rounding, bit-for-bit repeatable.  ws: tcar_small_det_ws_floats() floats (any contents: the row pieces +
 * the chunk partials of long buckets, B * T >= 2,048, where every table row's sources are cut into ~1,024-row chunks with a
 * workgroup each and one wave per row adds the chunks in order).  d->ldt must be 64 (TCAR_E_ARG otherwise: the time / dwell /
 * click rows are summed one 64-column group per row). */
int tcar_small_det_ws_floats(void);
int tcar_small_tables_bwd_det(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, const float* dx_icp,
                              const float* dx_pt, const float* dx_act, const float* dclick, const tcar_grads_t* g, float* ws,
                              void* stream);

/* tcar_gather_clip_bwd + the block partials of tcar_sqnorm_det (sum sq_g^2 into the last 4096 bytes of the segsum workspace) in
 * ONE launch: extra workgroups beside the row gradients */
int tcar_gather_clip_bwd_sqnorm(const tcar_dims_t* d, const tcar_tables_t* tab, const tcar_batch_t* bt, const float* dx_icp,
                                const float* dx_pt, const float* dx_act, const float* dclick, const tcar_grads_t* g,
                                const float* sq_g, int64_t sq_len, void* ws, int64_t ws_bytes, void* stream);

/* tcar_scatter_add_rows: g_item[ids[r]-1, :] += rows[r, :] for r < R (ids 1-based like `seq`; id 0 = padding
 * row, skipped).  The "bucketed sparse-embedding exchange" applies the all-gathered (id, row) pairs with it. */
int tcar_scatter_add_rows(const tcar_dims_t* d, const int32_t* ids, const float* rows, int64_t R, float* g_item,
                          void* stream);
/* ..._packed: the exchange buffer keeps each row's id behind it — packed[r, 0:ldh] = row, packed[r, ldh] = the 1-based id as
 * int32 bits, row stride ld floats (ld >= ldh + 1, ld % 4 == 0).  g_item[id - id0 - 1] += row for id0 < id <= id0 + n_items
 * (a catalog shard keeps the rows it owns; everything else, incl. id 0 padding, falls out). */
int tcar_scatter_add_rows_packed(const tcar_dims_t* d, const float* packed, int64_t ld, int64_t R, int32_t id0, float* g_item,
                                 void* stream);

/* tcar_cand_time_fwd: candidate_publish_t of model_combine.py:86-92 written into E[:, ic:ek].
 * mwdhm [N,5] int32 = publish_time_MWDHM (model_combine.py:37). */
int tcar_cand_time_fwd(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* mwdhm,
                       float* E, void* stream);

/* ..._bf16: additionally refreshes the same columns of the bf16 hi / lo planes of E (operands of tcar_gemm_bf16);
 * E may then be NULL (planes only: the split-bf16 scoring modes never read the fp32 time block). */
int tcar_cand_time_fwd_bf16(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* mwdhm, float* E,
                            void* e16_hi, void* e16_lo, void* stream);

/* tcar_cand_time_bwd: gradient of the above; d_et [N, pt] (ld = pt) is the time-column block of dE. */
int tcar_cand_time_bwd(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* mwdhm,
                       const float* d_et, const tcar_grads_t* g, void* stream);

/* Same result through a static inverted index (deterministic, no atomics): inv_n [5N] lists, per table row
 * r = rowoff(k)+v (month 0..12, day 13..44, week 45..52, hour 53..77, minute 78..138), the candidates n with
 * mwdhm[n,k] == v; inv_off [140] are the list offsets; ws holds tcar_cand_time_ws_floats(d) floats.
 * permuted != 0: d_et is stored IN LIST ORDER (segment i of ldt floats belongs to list entry i — the layout
 * tcar_gemm_bf16_perm writes with c2_perm = the inverse of the index), so every list is one contiguous stream. */
int tcar_cand_time_bwd_indexed(const tcar_dims_t* d, const float* const time_tab[5], const int32_t* inv_n,
                               const int32_t* inv_off, const float* d_et, int permuted, float* ws, const tcar_grads_t* g,
                               void* stream);
int tcar_cand_time_ws_floats(const tcar_dims_t* d);

/* ---- dense contractions --------------------------------------------------------------------------------
 * tcar_gemm_f32: fp32-in / fp32-accumulate MFMA GEMM (v_mfma_f32_32x32x2_f32), C = act(A*B + bias) (+C).
 * Replaces tf.matmul / BatchMatMul of modules.py:52,67 (linear_2d / linear_3d), model_combine.py:138
 * (full-catalog logits) and their gradients.
 *   layout 0 "NN": A[M,K] (lda), B[K,N] (ldb)      y = x W
 *   layout 1 "NT": A[M,K] (lda), B[N,K] (ldb)      logits = attout E^T ;  dx = dy W^T
 *   layout 2 "TN": A[K,M] (lda), B[K,N] (ldb)      dW = x^T dy ; dE = dlogits^T attout
 * act: 0 none, 1 relu, 2 tanh (util.py:60-90).  beta: 0 overwrite, 1 accumulate into C.
 * splitk > 1: C is [splitk, M, ldc] partial slabs (combine with tcar_splitk_reduce); bias/act/beta must be 0.
 * Requirements: pointers 16-byte aligned, lda/ldb % 4 == 0; K % 4 == 0 for k-contiguous operands. */
int tcar_gemm_f32(int layout, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                  float* C, int64_t ldc, const float* bias, int act, int beta, int splitk, void* stream);
int tcar_splitk_reduce(const float* slabs, int splitk, int M, int N, int64_t ld, float* out, void* stream);

/* Grouped form: up to 10 INDEPENDENT problems of one layout in a single launch (the step's ~25 small GEMMs are
 * bound by launch latency and 16-workgroup grids, not FLOPs).  A problem may K-concatenate up to 3 operand pairs
 * into one accumulator: C = act(sum_s A_s B_s + bias) (+C) — e.g. pre1 = X_ic W_in + X_c W_c + X_act W_int
 * (modules.py:126-131).  splitk > 1 (single segment, no bias/act/beta): atomic = 0 writes slabs
 * [splitk_eff, M, ldc]; atomic = 1 adds with fp32 atomics into a C the caller has zeroed (weight gradients,
 * whose K is the batch). */
typedef struct {
  int32_t nseg;
  const float* A[3];
  const float* B[3];
  int64_t lda[3], ldb[3];
  int32_t K[3];
  float* C;
  int64_t ldc;
  const float* bias;
  int32_t M, N, act, beta, splitk, atomic;
  /* Optional fused epilogues of tcar_gemm_x3_grouped (ignored by tcar_gemm_f32_grouped; all NULL / 0 = plain):
   *  - activation BACKWARD: dact = 1 (relu) / 2 (tanh): C = acc * act'(dact_y[m, n]) and, with colsum != NULL,
   *    colsum[n] += sum_m C[m, n] (fp32 atomics into a caller-zeroed vector): the bias + activation backward of linear_2d
   *    (modules.py:52-54) in the epilogue of the GEMM that produces the incoming gradient;
   *  - bf16 planes: the result (after bias / act) additionally goes to hi / lo KB32 planes with inner dimension
   *    plane_inner (>= N, % 32 == 0) at column offset plane_col0, and columns outside [pack_c0, pack_c1) to a second,
   *    PACKED plane pair (column index rebased past the gap, inner dimension pack_inner) — the operands the scoring GEMMs
   *    read (tcar_split_bf16 without its own launch). */
  int32_t dact;
  const float* dact_y;
  int64_t ld_dact_y;
  float* colsum;
  void* plane_hi; void* plane_lo; int32_t plane_inner, plane_col0;
  void* pack_hi; void* pack_lo; int32_t pack_inner, pack_c0, pack_c1;
} tcar_gemm_desc_t;
int tcar_gemm_f32_grouped(int layout, int nprob, const tcar_gemm_desc_t* descs /*host*/, void* stream);
/* Same contract on the bf16 matrix cores: each staged fp32 tile is split on the fly into bf16 hi / lo planes and every
 * product is three bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate, ~1e-5 relative).  For the latency-bound small
 * contractions of the step when the scoring precision is a bf16 mode. */
int tcar_gemm_x3_grouped(int layout, int nprob, const tcar_gemm_desc_t* descs /*host*/, void* stream);
/* number of slabs tcar_gemm_f32 actually writes for a requested split (K is cut in multiples of 32) */
int tcar_gemm_splitk_effective(int K, int splitk);

/* tcar_gemm_bf16: the same three layouts on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for
 * the full-catalog scoring GEMMs.  Operands are bf16 PLANES of fp32 data, x = hi + lo, in the KB32 blocked layout
 * (csrc/tcar_bf16_layout.h: 128-row x 32-inner blocks of 8 KB, rows padded to 128 with zeros) produced by
 * tcar_split_bf16 / tcar_clip_adam_2d_bf16 / tcar_cand_time_fwd_bf16 / tcar_softmax_ce_bf16.
 * nsplit = 3 computes a_hi b_hi + a_hi b_lo + a_lo b_hi (fp32-class accuracy, ~1e-5), nsplit = 1 uses hi only.
 * *_inner = inner (contiguous) dimension of the plane, % 32 == 0; *_rows = its row count.  K % 32 == 0 (zero padded).
 *   layout 0: A plane [M rows, inner >= K],  B plane [K rows, inner >= N]
 *   layout 1: A plane [M rows, inner >= K],  B plane [N rows, inner >= K]
 *   layout 2: A plane [K rows, inner >= M],  B plane [K rows, inner >= N]
 * C2 != NULL: output columns >= csplit are written to C2[:, col - csplit] (dE: item block | time block).
 * splitk > 1: C is [splitk_eff, M, ldc] slabs (tcar_gemm_splitk_effective / tcar_splitk_reduce). */
int tcar_gemm_bf16(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                   const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, float* C, int64_t ldc, float* C2,
                   int64_t ldc2, int csplit, int nsplit, int splitk, void* stream);

/* ..._tuned: with the caller's switch values (bf16_tile pins the workgroup tile, bf16_ks the stage depth) */
int tcar_gemm_bf16_tuned(const tcar_tuning_t* tune, int layout, int M, int N, int K, const void* A_hi, const void* A_lo,
                         int64_t a_inner, int64_t a_rows, const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows,
                         float* C, int64_t ldc, float* C2, int64_t ldc2, int csplit, int nsplit, int splitk, void* stream);

/* ..._perm: as tcar_gemm_bf16, with a grouped row permutation of the SECOND destination: element (m, cc) of the C2 block
 * (cc = column - csplit) is stored at C2[c2_perm[(cc / c2_group) * M + m] * c2_group + cc % c2_group] (ldc2 unused).
 * The dE GEMM writes the candidate-time block this way, in the order of the static inverted index of
 * publish_time_MWDHM, so that tcar_cand_time_bwd_indexed streams it (permuted = 1).  c2_perm == NULL: plain layout. */
int tcar_gemm_bf16_perm(int layout, int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner,
                        int64_t a_rows, const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, float* C,
                        int64_t ldc, float* C2, int64_t ldc2, int csplit, const int32_t* c2_perm, int c2_group, int nsplit,
                        int splitk, void* stream);
/* Logits GEMM with the softmax EPILOGUE (model_combine.py:138 + :145 without materialising [M, N] fp32 logits): layout 1 of
 * tcar_gemm_bf16 (C = A B^T, both operands k-contiguous).  Instead of C the kernel writes
 *   p_hi   bf16 KB32 plane [ceil128(M) rows, p_inner]: exp(x[m, n] - gmax[m, group(n)]), 0 for n >= N;
 *   stats  [M, *ngroups, 2] floats: (group maximum, sum of the group's exponentials) — a group = *group_width (64 or 96)
 *          consecutive columns, the slice one wave of the chosen workgroup tile owns; stats_floats >= M * (ceil(N/64) + 8) * 2;
 *   lab_logit [M]: x[m, label[m]].
 * Optional SECOND K segment (B2_hi != NULL; nsplit 3): the contraction runs over K1 columns of (A, B) and then K - K1 columns of
 * (A2 hi / lo, B2 hi) — planes of inner dimension inner2 with the same row counts — where B2 is exact in bf16 (ONE plane: two
 * MFMAs per product).  The step driver uses it for the candidate-side time vectors (model_combine.py:86-92,135): A2 =
 * tcar_time_scores, B2 = tcar_time_onehot, K1 = 2 ldh, K - K1 = 160 instead of 5 ldt = 320 two-plane columns.
 * tcar_ce_finish combines the groups of every row (lse, ce = lse - x_label), then rescales the plane IN PLACE to
 * softmax - onehot (the dlogits operand of the two gradient GEMMs, hi plane only) and zeroes rows [B, ceil128(B)).  rowstat [B, 2]. */
int tcar_gemm_bf16_ce(int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows, const void* B_hi,
                      const void* B_lo, int64_t b_inner, int64_t b_rows, int K1, const void* A2_hi, const void* A2_lo,
                      const void* B2_hi, int64_t inner2, void* p_hi, int64_t p_inner, int64_t p_rows, float* stats,
                      int64_t stats_floats, const int32_t* label, float* lab_logit, int nsplit, int32_t* group_width /*host*/,
                      int32_t* ngroups /*host*/, void* stream);
/* OH[n, r] = 1 for r = rowoff_k + publish_time_MWDHM[n, k] (k = 0..4; rows of the month | day | week | hour | minute tables numbered
 * 0..138), else 0: bf16 KB32 plane [ceil128(N), inner >= 160].  Static per catalog. */
int tcar_time_onehot(const tcar_dims_t* d, const int32_t* mwdhm, void* oh_hi, int64_t inner, void* stream);
/* P[b, r] = attout[b, 2 ldh + k(r) ldt ...] . clip(time table row r) as bf16 hi / lo KB32 planes [ceil128(B), inner >= 160]:
 * sum_k attout_tk[b] . candidate_publish_t_k[n] = (P OH^T)[b, n] */
int tcar_time_scores(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* attout, int64_t ld_att, void* p_hi,
                     void* p_lo, int64_t inner, void* stream);
/* Output transforms of the session side finished in one launch: attout [B, 2 ldh + 5 ldt] = tanh(sum of split-K slabs + bias)
 * (model_combine.py:119,127,132) from the slabs tcar_gemm_x3_grouped leaves with splitk = nd_ic / nd_pt (slab k of the item|content
 * problem at slabs + k stride, of the time problem at slabs + 2 ldh + k stride; row stride 2 ldh + 5 ldt), folded in slab order;
 * optional hi / lo planes of attout (a_*), packed [item | time] planes (ap_*: the dE operand), time scores P (p_*) and clipped
 * table rows (tclip) as tcar_time_scores_clip computes them.  ldt == 64, ldh % 64 == 0. */
int tcar_attout_finish_scores(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* slabs, int nd_ic, int nd_pt,
                              int64_t stride, const float* bias_o, const float* bias_ot, float* attout, int64_t ld_out, void* a_hi,
                              void* a_lo, int64_t a_inner, void* ap_hi, void* ap_lo, int64_t ap_inner, void* p_hi, void* p_lo,
                              int64_t p_inner, float* tclip, void* stream);
/* ..._clip: additionally writes the clipped table rows the scores were taken against — tclip [160 ldt + 320] floats: row r of the
 * month | day | week | hour | minute tables after max_norm = 1 (rows 139..159 untouched), then scale[160] = 1 / max(||row||, 1) and
 * clipped[160] = 1.0 where ||row|| > 1 — for the one-hot form of the scoring GRADIENTS (tcar_gemm_bf16_de_qz, tcar_reduce_dact_onehot,
 * tcar_cand_time_bwd_onehot) */
int tcar_time_scores_clip(const tcar_dims_t* d, const float* const time_tab[5], int B, const float* attout, int64_t ld_att,
                          void* p_hi, void* p_lo, int64_t inner, float* tclip, void* stream);
int tcar_ce_finish(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit, const int32_t* label,
                   float* rowstat, float* ce, void* dl_hi, int64_t inner, void* stream);
/* ANCHORED form of the softmax epilogue + finish (round 6).  When every logit of row b comes out of the GEMM as x - a[b] for a per-row
 * reference a[b] — the fused step: the label's score up to rounding, subtracted INSIDE the contraction: the forward pass's finishing
 * launch leaves minus its partial sums in eight spare columns (139 .. 146) of the time-score planes, the one-hot plane of
 * tcar_time_onehot has ones there — tcar_gemm_bf16_ce_anchor writes the plane exp(accumulator) without group maxima, statistics
 * (group sum of the plane's bf16-ROUNDED entries, group sum of the exponentials) and lab_logit = the label's accumulator: all groups of a row share ONE scale, and the plane never needs the rescale
 * pass.  tcar_ce_anchor_fold (one wave per row; any B: the scaled attout rows [B, ceil32(B)) are zeroed) folds the group sums in a fixed order:
 * S_b of the exponentials (ce = log S_b - lab_logit) and S_r of the plane's rounded entries (the gradient's scale: plane / S_r sums to one
 * exactly — S_b below is S_r wherever a gradient is scaled); writes
 * rowstat[b] = (0, 1 / S_b) (may be NULL); puts the label's -1 into the plane as v = bf16(e_l - S_b); writes
 * scale2[b] = (1 / S_b, ((e_l - S_b) - v) / S_b) for the consumer that is linear in the plane's rows (the slab reduce of dX,
 * tcar_reduce_dact_onehot_scaled: softmax part scaled exactly, the one-hot's rounding residual added back in fp32); and writes
 * aps = bf16((ap_hi + ap_lo)[b, :] / S') with S' = e_l - v, the per-row scaled copy of the packed attout planes [B, ap_cols]
 * (inner ap_inner) that the dE GEMM contracts the UNSCALED plane with — v / S' = e_l / S' - 1 exactly: the one-hot part of dE is
 * exact, the rounding of v becomes a common factor 1 +- 2^-8 on the row's softmax part.  S_b >= ~1 by construction of the anchor; the
 * GEMM's exponent is clamped at 2^100: a logit more than 69 nats above its row's reference (a per-session loss > 69) saturates instead
 * of overflowing — every quantity stays finite. */
int tcar_ce_anchor_fold(int B, int N, int group_width, int ngroups, const float* stats, const float* lab_logit, const int32_t* label,
                        float* rowstat, float* ce, float* scale2, void* dl_hi, int64_t inner, const void* ap_hi, const void* ap_lo,
                        void* aps_hi, int ap_cols, int64_t ap_inner, void* stream);
/* tcar_gemm_bf16_ce with the anchored epilogue: the caller has put the row references into the contraction (see above); plane =
 * exp(accumulator), statistics (group sum of the plane's rounded entries, group sum of the exponentials), lab_logit = the label's accumulator */
int tcar_gemm_bf16_ce_anchor(int M, int N, int K, const void* A_hi, const void* A_lo, int64_t a_inner, int64_t a_rows,
                             const void* B_hi, const void* B_lo, int64_t b_inner, int64_t b_rows, int K1, const void* A2_hi,
                             const void* A2_lo, const void* B2_hi, int64_t inner2, void* p_hi, int64_t p_inner, int64_t p_rows,
                             float* stats, int64_t stats_floats, const int32_t* label, float* lab_logit, int nsplit,
                             int32_t* group_width, int32_t* ngroups, void* stream);
/* tcar_reduce_dact_onehot for the anchored form: the slab sums (and dP) of row m times scale2[m].x, plus scale2[m].y times
 * [E[label[m], 0 .. ic) | onehot(mwdhm[label[m]])] (E: fp32 candidate rows [n_items, ldE]), in front of the addend; no bias column sums */
int tcar_reduce_dact_onehot_scaled(const float* slabs, int splitk, int M, int ic, int64_t ld, const float* addend, int64_t ld_add,
                                   const float* y, int64_t ldy, const float* tclip, float* out, int64_t ldo, float* dP,
                                   const float* scale2, const int32_t* label, const float* E, int64_t ldE, const int32_t* mwdhm,
                                   int n_items, void* stream);
/* The two halves of tcar_ce_finish for the catalog-sharded step (sharded.py), where the row statistics cross the ranks in between:
 *   tcar_ce_shard_stats  out3[b] = (max, sum exp(x - max), label score) of THIS shard's columns from the epilogue's per-group pairs;
 *                        the label score is 0 unless label[b] lies in [n0, n0 + n_loc)  (= tcar_softmax_stats without the logits)
 *   tcar_ce_rescale      plane[b, n] = e[b, n] exp(m_g - rowstat[b].x) rowstat[b].y - [n == label[b] - lab_off]; lab_window != 0: a
 *                        label outside [lab_off, lab_off + N) belongs to another shard, nothing is subtracted (0: clamped, as
 *                        tcar_ce_finish).  rowstat = (lse, 1) from tcar_softmax_combine_rowstat. */
int tcar_ce_shard_stats(int B, int ngroups, const float* stats, const float* lab_logit, const int32_t* label, int n0, int n_loc,
                        float* out3, void* stream);
int tcar_ce_rescale(int B, int N, int group_width, int ngroups, const float* stats, const float* rowstat, const int32_t* label,
                    int lab_off, int lab_window, void* dl_hi, int64_t inner, void* stream);
/* ---- one-hot form of the two scoring GRADIENT GEMMs (training steps, hi planes only) -------------------------------------------
 * The candidate-side time columns of items_emb (model_combine.py:86-92,135-136) are five clipped table rows per item, selected by
 * publish_time_MWDHM: E_time = OH T_clip with the static 0/1 matrix OH [N, 160] (tcar_time_onehot) and the 139 clipped rows
 * T_clip (tcar_time_scores_clip).  Gradient of model_combine.py:138 through that factorisation:
 *   tcar_gemm_bf16_dx_onehot   slabs[s] = dlogits [E_item | E_content | OH]  (layout 0 of tcar_gemm_bf16; N1 = 2 ldh columns from the
 *                              planes B, then 160 from the one-hot plane B2): split-K slabs [splitk_eff, M, ldc], ldc >= N1 + 160
 *   tcar_reduce_dact_onehot    tcar_splitk_reduce_dact for those slabs: d attout [M, ic + 320] (the time columns expanded as
 *                              (dlogits OH) T_clip, everything through tanh') and dP = dlogits OH [M, 160]
 *   tcar_gemm_bf16_de_qz       dE = dlogits^T [attout_item | attout_time] (layout 2): the item block [M = N, ldh] goes to C; the time
 *                              block is NOT stored — per (n, k): qz[perm[k N + n]] = (||gy||^2, x . gy), gy = its 64-column gradient
 *                              block, x = the clipped row item n looks up in table k.  perm = position of (k, n) in the inverted index
 *   tcar_cand_time_bwd_onehot  the IndexedSlices gradient of the five time tables' candidate-side lookups and its norm pieces
 *                               from qz, dP and attout — what tcar_cand_time_bwd_indexed computes from a stored
 *                              [N, 5 ldt] block.  ldt must be 64.
 * Against the materialised form this moves 118 MB less per step at the Globo catalog (no [N, 320] fp32 block written and re-read)
 * and the dX GEMM contracts 160 instead of 320 time columns. */
int tcar_gemm_bf16_dx_onehot(int M, int N1, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi,
                             int64_t b_inner, int64_t b_rows, const void* B2_hi, int64_t inner2, float* C, int64_t ldc, int splitk,
                             void* stream);
/* the same with the caller's copy of the switches (TCAR_BF16_TILE = 384: the 256 x 384 tile of long contractions, whose second
 * column tile straddles the boundary between the B and B2 planes; TCAR_BF16_KS) */
int tcar_gemm_bf16_dx_onehot_tuned(const tcar_tuning_t* tune, int M, int N1, int K, const void* A_hi, int64_t a_inner, int64_t a_rows,
                                   const void* B_hi, int64_t b_inner, int64_t b_rows, const void* B2_hi, int64_t inner2, float* C,
                                   int64_t ldc, int splitk, void* stream);
int tcar_reduce_dact_onehot(const float* slabs, int splitk, int M, int ic, int64_t ld, const float* addend, int64_t ld_add,
                            const float* y, int64_t ldy, const float* tclip, float* out, int64_t ldo, float* dP, float* bias_grad0,
                            float* bias_grad1, void* stream);
int tcar_gemm_bf16_de_qz(int M, int K, const void* A_hi, int64_t a_inner, int64_t a_rows, const void* B_hi, int64_t b_inner,
                         int64_t b_rows, int ldh, float* C, int64_t ldc, const int32_t* mwdhm, const int32_t* perm, const float* tclip,
                         float* qz, int tile /* 0: 192 x 192 workgroup tile (9 waves), 256: 256 x 192 (12), 128: 128 x 192 (6), 64: 64 x 192 (3) */, void* stream);
int tcar_cand_time_bwd_onehot(const tcar_dims_t* d, int B, const int32_t* inv_off, const float* qz, const float* dP,
                              const float* attout, int64_t ld_att, const float* tclip, float* ws, const tcar_grads_t* g, void* stream);
/* Names the kernel instantiation (template arguments, workgroup tile, grid) that tcar_gemm_bf16 would launch for this
 * problem, without launching it (profiling tools match rocprofv3 kernel names with it).  buf: host, buflen >= 96. */
int tcar_gemm_bf16_variant(int layout, int M, int N, int K, int nsplit, int splitk, char* buf /*host*/, int buflen);
/* fp32 [rows, cols] (ld) -> bf16 hi / lo KB32 planes with inner dimension `inner` (>= cols, % 32 == 0); padding rows
 * up to ceil128(rows) and columns >= cols are zero filled (lo may be NULL).  packed_* != NULL additionally writes
 * columns [0,c0) U [c1,cols) as a second plane pair with inner dimension packed_inner. */
int tcar_split_bf16(const float* x, int64_t ld, int rows, int cols, void* hi, void* lo, int64_t inner, void* packed_hi,
                    void* packed_lo, int64_t packed_inner, int c0, int c1, void* stream);

/* ---- attention pools (modules.py:72-152, util.py:92-100) --------------------------------------------------
 * pre1 [B*T, ldh] = X_ic W_in + X_c W_c + X_act W_int (no activation), pre2 likewise for the time pool,
 * q [B, ic] = tanh(relu(click_t Wq1 + b) Wq2 + b).  Computes alpha1 = expnorm(sigmoid(pre1) . w_res1),
 * alpha2 = expnorm(X_ic . q), alpha_t = expnorm(sigmoid(pre2) . w_res2) and
 * pooled [B, ek] = [ (alpha1+alpha2)^T X_ic | alpha_t^T X_pt ];  alpha [3, B*T] is saved for backward. */
int tcar_attn_pool_fwd(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                       const float* pre1, const float* pre2, const float* q, const float* w_res1,
                       const float* w_res2, float* pooled, float* alpha, void* stream);
/* Backward: writes dx_icp, dx_pt (pool + query-dot parts), dq [B, ic], dpre1, dpre2 [B*T, ldh];
 * adds into g_wres1, g_wres2 [ldh] (atomics). */
int tcar_attn_pool_bwd(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                       const float* pre1, const float* pre2, const float* q, const float* w_res1,
                       const float* w_res2, const float* alpha, const float* dpooled, float* dx_icp,
                       float* dx_pt, float* dq, float* dpre1, float* dpre2, float* g_wres1, float* g_wres2,
                       void* stream);

/* ..._q: with g_qbias != NULL (a caller-zeroed [ic] vector) dq leaves already multiplied by tanh'(q) = 1 - q^2 and its column
 * sums are added into g_qbias — the activation + bias backward of query_trans2 (modules.py:139) without its own launch. */
int tcar_attn_pool_bwd_q(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt,
                         const float* pre1, const float* pre2, const float* q, const float* w_res1,
                         const float* w_res2, const float* alpha, const float* dpooled, float* dx_icp,
                         float* dx_pt, float* dq, float* dpre1, float* dpre2, float* g_wres1, float* g_wres2,
                         float* g_qbias, void* stream);
/* tcar_attn_pool_fwd with pre1 / pre2 arriving as n1 / n2 split-K partial products (slab s at pre?_slabs + s * slab_stride,
 * [B*T, ldh] each): folded in slab order while they are read; the sums are also written to pre1 / pre2 [B*T, ldh], which the
 * backward pass re-reads (modules.py:94-96,126-131 -> :97-100,132-135) */
int tcar_attn_pool_fwd_slabs(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt, const float* pre1_slabs,
                             int n1, const float* pre2_slabs, int n2, int64_t slab_stride, float* pre1, float* pre2,
                             const float* q, const float* w_res1, const float* w_res2, float* pooled, float* alpha, void* stream);
/* tcar_attn_pool_bwd_det with dpooled arriving as split-K partial products: nd_ic slabs for the item | content columns, nd_pt
 * for the time columns, slab s at dpooled + s * dp_stride ([B, ek] each), folded in slab order (1, 1: plain dpooled) */
int tcar_attn_pool_bwd_slabs(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt, const float* pre1,
                             const float* pre2, const float* q, const float* w_res1, const float* w_res2, const float* alpha,
                             const float* dpooled, int nd_ic, int nd_pt, int64_t dp_stride, float* dx_icp, float* dx_pt, float* dq,
                             float* dpre1, float* dpre2, float* gw_rows, void* stream);
/* order-fixed form of the same: dq leaves through tanh'(q), the per-session rows d w_res1 | d w_res2 are written to gw_rows
 * [B, 2*ldh] and nothing is summed atomically (tcar_colsum_det folds gw_rows and dq into the three gradients) */
int tcar_attn_pool_bwd_det(const tcar_dims_t* d, int B, int T, const float* x_icp, const float* x_pt, const float* pre1,
                           const float* pre2, const float* q, const float* w_res1, const float* w_res2, const float* alpha,
                           const float* dpooled, float* dx_icp, float* dx_pt, float* dq, float* dpre1, float* dpre2,
                           float* gw_rows, void* stream);

/* ---- scoring loss -----------------------------------------------------------------------------------------
 * tcar_softmax_ce: tf.nn.sparse_softmax_cross_entropy_with_logits (model_combine.py:145) and its gradient.
 * logits [B, ld] (first N columns valid) is overwritten by dlogits = softmax - onehot (pad columns 0). */
int tcar_softmax_ce(int B, int N, float* logits, int64_t ld, const int32_t* label, float* ce, void* stream);

/* ..._bf16: the gradient goes to bf16 hi / lo KB32 planes [ceil128(B), ld] (ld % 32 == 0; padding rows zeroed);
 * the logits stay intact.  dl_lo may be NULL (hi-only backward: the lo plane is neither read nor written). */
int tcar_softmax_ce_bf16(int B, int N, float* logits, int64_t ld, const int32_t* label, float* ce, void* dl_hi,
                         void* dl_lo, void* stream);

/* tcar_neg_term: neg_logits / neg_feedback of model_combine.py:142-143 and their gradients.
 * neg_fb[b] = -log(1 - sigmoid(x_b) + 1e-24), x_b = sum_k E[neg[b,k], 0:ic] . attout[b, 0:ic].
 * Adds weight * d neg_fb into dattout[b, 0:ic] (plain adds: one workgroup owns a row) and into g_item (atomics);
 * neg_fb, dattout and g_item may each be NULL to skip that output.  With loss != NULL it also writes the per-session
 * training loss of model_combine.py:147, loss[b] = ce[b] + weight * neg_fb[b] (ce: the softmax cross entropy). */
int tcar_neg_term(const tcar_dims_t* d, int B, int K, const float* E, const int32_t* neg,
                  const float* attout, float weight, float* neg_fb, float* dattout, float* g_item,
                  const float* ce, float* loss, void* stream);

/* The same term split along the step's dependency graph (model_combine.py:142-143,147): tcar_neg_fwd needs only
 * forward quantities and runs beside the logits GEMM — neg_fb[b], coef[b] = weight * d neg_fb / d x_b and
 * negpart[b, 0:ic] = coef[b] * sum_k E[neg[b,k], 0:ic] (the term's gradient w.r.t. attout[b, 0:ic]);
 * tcar_neg_scatter adds coef[b] * attout[b, 0:ldh] into g_item[neg[b,k]] (atomics) once dE is in place and, with
 * loss != NULL, writes loss[b] = ce[b] + weight * neg_fb[b]. */
int tcar_neg_fwd(const tcar_dims_t* d, int B, int K, const float* E, const int32_t* neg, const float* attout,
                 float weight, float* neg_fb, float* coef, float* negpart, void* stream);
int tcar_neg_scatter(const tcar_dims_t* d, int B, int K, const int32_t* neg, const float* attout, const float* coef,
                     float* g_item, const float* neg_fb, const float* ce, float weight, float* loss, void* stream);

/* tcar_splitk_reduce_dact: out[m,n] = (sum_s slabs[s,m,n] + (n < n_add ? addend[m,n] : 0)) * act'(y[m,n]) and the bias
 * gradients bias_grad0[n] (n < split_col) / bias_grad1[n - split_col] += sum_m out[m,n] (atomics into zeroed buffers).
 * Fuses tcar_splitk_reduce, the dattout part of tcar_neg_term and tcar_dact_colsum for the output transforms
 * (model_combine.py:119,127,132 backward).  addend / bias_grad* may be NULL; act as in tcar_gemm_f32. */
int tcar_splitk_reduce_dact(const float* slabs, int splitk, int M, int N, int64_t ld, const float* addend, int64_t ld_add,
                            int n_add, const float* y, int64_t ldy, int act, float* out, float* bias_grad0, int split_col,
                            float* bias_grad1, void* stream);

/* tcar_dact_colsum: dz = dy * act'(y) in place over dy ([M, ncol], ld) and bias_grad[c] += sum_m dz[m,c]
 * (gradient of linear_2d's bias + activation, modules.py:52-54; fp32 atomics into a caller-zeroed bias_grad).
 * act: 1 relu, 2 tanh. */
int tcar_dact_colsum(int M, int ncol, int64_t ld, const float* y, float* dy, float* bias_grad, int act,
                     void* stream);

/* ---- evaluation (util.py:8-18, model_combine.py:301) --------------------------------------------------------
 * rank[b] = 1 + #{n : logits[b,n] > logits[b,label[b]]};  topk[b, 0:k] = indices of the k largest scores,
 * descending, ties broken towards the higher index (np.argsort(...)[::-1]). */
int tcar_rank_topk(int B, int N, const float* logits, int64_t ld, const int32_t* label, int k,
                   int32_t* rank, int32_t* topk, void* stream);

/* tcar_eval_rows: tcar_rank_topk plus (ce != NULL) the sparse softmax cross entropy of the same rows, from ONE read of the
 * score matrix (row-resident kernel, ld <= 49,152 and 16-byte aligned rows; otherwise ce must be NULL and the streaming
 * kernel runs).  The logits are left untouched.  This is the whole of `sess.run([softmax_input, cross_loss])` +
 * cau_metrics + argsort (model_combine.py:283,293-301) after the logits GEMM. */
int tcar_eval_rows(int B, int N, const float* logits, int64_t ld, const int32_t* label, int k, int32_t* rank, int32_t* topk,
                   float* ce, void* stream);

/* tcar_eval_diversity: the diversity metrics of the evaluation loop on the device, from the top-k lists tcar_eval_rows wrote.
 *   getILD    (model_combine.py:174-182)  ild_cnt[b]   = #{(i, j), i != j : cat[topk[b,i]] != cat[topk[b,j]]}
 *   getUnexp  (model_combine.py:184-194)  unexp_cnt[b] = #{(i, t) : cat[topk[b,i]] != cat[seq[b,t] - 1]}      (seq is 1-based)
 *   resultItemDict (:305-306,313)         seen[n] = 1 for every recommended item n (byte map [n_items], caller-zeroed per
 *                                         evaluation; coverage = its sum; ranks union it with a MAX all-reduce); may be NULL
 *   n_rec[b] = entries of the list (topk entries < 0 — a catalog shorter than k — are not in it); may be NULL.
 * cat [n_items] int32 = category code of item n (category_id[reverse_item[n]], any injective coding: only != is used).
 * The counts are exact integers; the reference's score / (n (n - 1)) and score / (n len(inSeq)) are left to the host
 * (int / int in double precision, as Python does).  k <= 64. */
int tcar_eval_diversity(int B, int T, int k, int n_items, const int32_t* topk, const int32_t* seq, const int32_t* cat,
                        int32_t* ild_cnt, int32_t* unexp_cnt, int32_t* n_rec, uint8_t* seen, void* stream);

/* ---- optimizer (model_combine.py:155-163) ---------------------------------------------------------------------
 * Segments of one flat fp32 arena (identical offsets in w, g, m, v). */
typedef struct {
  int32_t nseg;
  int64_t off[TCAR_NSLOT];   /* start of segment (floats, multiple of 4) */
  int64_t len[TCAR_NSLOT];   /* floats (multiple of 4) */
  int32_t slot[TCAR_NSLOT];  /* squared-norm slot of the variable the segment belongs to */
} tcar_segments_t;

/* sqn_dense[slot] += sum g^2 over each segment. */
/* dst[c] += sum over rows of x[r, c] in a FIXED order (64 columns per workgroup, 16 row phases folded in order): the bias
 * gradients of linear_2d (column sums of dy * act', modules.py:52-54 backward) and the residual-weight gradients of the
 * attention layers without float atomics.  Up to 8 matrices per launch. */
typedef struct {
  const float* x; int64_t ld; int32_t rows, cols;
  float* dst;
} tcar_colsum_t;
int tcar_colsum_det(int nseg, const tcar_colsum_t* segs /*host*/, void* stream);
/* dst[e] += sum over k < ks of slabs[k * stride + e], k in order: the ordered fold of split-K slabs (weight gradients of
 * batches longer than the un-split limit).  n, stride multiples of 4; up to 9 segments per launch. */
typedef struct {
  float* dst; const float* slabs; int64_t n; int32_t ks; int64_t stride;
} tcar_fold_t;
int tcar_fold_slabs(int nseg, const tcar_fold_t* segs /*host*/, void* stream);
/* (segments of <= 262,144 floats: one workgroup each, fixed summation order — identical inputs give identical bits;
 *  longer segments: chunked with one float atomic per chunk) */
int tcar_sqnorm(const float* g, const tcar_segments_t* segs /*host*/, float* sqn_dense, void* stream);
/* Per-variable tf.clip_by_norm(g, clip) then TF-1 Adam.  norm^2 of a variable =
 * use_dense[slot]*sqn_dense[slot] + sqn_pieces[slot]; scale = clip / max(norm, clip) (clip <= 0: no clipping).
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t m / (sqrt(v) + eps). */
int tcar_clip_adam(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs /*host*/,
                   const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense /*device [NSLOT]*/,
                   float clip, float lr_t, float b1, float b2, float eps, void* stream);
/* Same for a 2-D strided parameter block (the item table inside E): w [rows, cols] with leading dim ldw;
 * g, m, v are compact [rows, cols]. */
int tcar_clip_adam_2d(float* w, int64_t ldw, const float* g, float* m, float* v, int64_t rows, int32_t cols,
                      int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense,
                      float clip, float lr_t, float b1, float b2, float eps, void* stream);

/* ..._bf16: also rewrites the updated block into the bf16 hi / lo planes [rows, ld16] of the candidate matrix. */
int tcar_clip_adam_2d_bf16(float* w, int64_t ldw, const float* g, float* m, float* v, int64_t rows, int32_t cols,
                           int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense,
                           float clip, float lr_t, float b1, float b2, float eps, void* e16_hi, void* e16_lo,
                           int64_t ld16, void* stream);

/* Library self-description (for loaders). */
/* tcar_clip_adam_all: tcar_clip_adam over the arena segments AND tcar_clip_adam_2d_bf16 over the item table in ONE launch
 * (the arena update is a 14-us latency-bound launch on its own; here its blocks ride behind the item table's). */
int tcar_clip_adam_all(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs, float* w2d, int64_t ldw,
                       const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols, int32_t slot,
                       const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip, float lr_t,
                       float b1, float b2, float eps, void* e16_hi, void* e16_lo, int64_t ld16, void* stream);

/* ---- optional op, NOT on TCAR's executed graph: the attention core of multihead_attention (modules.py:220-304) ------
 * Q [N,Tq,C], K, V [N,Tk,C] are the dense projections (tcar_gemm_f32), key_mask [N,Tk] = sign(|sum_c keys|),
 * query_mask [N,Tq] = sign(|sum_c queries|) (modules.py:263,283).  Per head h (C % heads == 0, Tq, Tk <= 64):
 * S = Q_h K_h^T / sqrt(C/heads), S = -2^32+1 where key_mask == 0 or (causal and tk > tq), P = softmax(S) * query_mask,
 * O_h = P V_h.  P [N*heads, Tq, Tk] is saved for the backward pass, which returns dQ, dK, dV. */
int tcar_mha_core_fwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* Q, const float* K, const float* V,
                      const float* key_mask, const float* query_mask, float* O, float* P, void* stream);
int tcar_mha_core_bwd(int N, int Tq, int Tk, int C, int heads, int causal, const float* Q, const float* K, const float* V,
                      const float* P, const float* key_mask, const float* query_mask, const float* dO, float* dQ, float* dK,
                      float* dV, void* stream);

/* ..._tuned: mha_mfma = 0 forces the scalar form */
int tcar_mha_core_fwd_tuned(const tcar_tuning_t* tune, int N, int Tq, int Tk, int C, int heads, int causal, const float* Q,
                            const float* K, const float* V, const float* key_mask, const float* query_mask, float* O, float* P,
                            void* stream);
int tcar_mha_core_bwd_tuned(const tcar_tuning_t* tune, int N, int Tq, int Tk, int C, int heads, int causal, const float* Q,
                            const float* K, const float* V, const float* P, const float* key_mask, const float* query_mask,
                            const float* dO, float* dQ, float* dK, float* dV, void* stream);

/* ---- optional op, NOT on TCAR's executed graph: `normalize` of modules.py:194-218 (layer normalisation over the last axis):
 * y = gamma * (x - mean) / sqrt(var + eps) + beta with the biased variance of tf.nn.moments; stats [M, 2] = (mean, 1/std) is
 * saved for the backward pass, which returns dx and ADDS the column sums into caller-zeroed dgamma / dbeta (atomics). */
int tcar_layernorm_fwd(int64_t M, int C, const float* x, const float* gamma, const float* beta, float eps, float* y,
                       float* stats, void* stream);
int tcar_layernorm_bwd(int64_t M, int C, const float* x, const float* gamma, const float* stats, const float* dy, float* dx,
                       float* dgamma, float* dbeta, void* stream);

/* Split update.  tcar_clip_adam_early: tcar_clip_adam over the arena segments plus the item rows listed in `ids` (1-based
 * item ids, repeats allowed: a bit per row in `bitmap` — zero on entry — makes every row update exactly once);
 * tcar_clip_adam_rest: every item row whose bit is clear, then clears the bitmap — `bitmap` holds ceil(rows / 32) words rounded UP to
 * a multiple of 16 (whole 64-byte units: the clear is one fill).  Together they equal
 * tcar_clip_adam_all; the step driver runs the second on the aux stream beside the next forward pass. */
int tcar_clip_adam_early(float* w, const float* g, float* m, float* v, const tcar_segments_t* segs, float* w2d, int64_t ldw,
                         const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols, int32_t slot,
                         const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip, float lr_t,
                         float b1, float b2, float eps, void* e16_hi, void* e16_lo, int64_t ld16, const int32_t* ids,
                         int64_t n_ids, uint32_t* bitmap, void* stream);
int tcar_clip_adam_rest(float* w2d, int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols,
                        int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip,
                        float lr_t, float b1, float b2, float eps, void* e16_hi, void* e16_lo, int64_t ld16, uint32_t* bitmap,
                        void* stream);
/* the same without clearing the bitmap (the step driver records its join event first and clears behind it) */
int tcar_clip_adam_rest_keep(float* w2d, int64_t ldw, const float* g2d, float* m2d, float* v2d, int64_t rows, int32_t cols,
                        int32_t slot, const float* sqn_dense, const float* sqn_pieces, const int32_t* use_dense, float clip,
                        float lr_t, float b1, float b2, float eps, void* e16_hi, void* e16_lo, int64_t ld16, uint32_t* bitmap,
                        void* stream);

/* ---- deterministic sparse backward of the item table: sort by row + segmented wavefront reduction -----------------------
 * The IndexedSlices gradient of the item lookups (model_combine.py:54,142,156): B*T session rows + B*K negative rows.
 *   tcar_segsum_ws_bytes    workspace for up to max_sources = B*(T+K) sources
 *   tcar_segsum_index       sorts the sources of each list of `bt` (session clicks, negatives) by destination row, stably:
 *                           lists of up to 16384 sources by one workgroup each in LDS, longer ones by the multi-workgroup form of the same sort; depends
 *                           on the feed only
 *   tcar_segsum_rows_buffer [B*T, ldh] buffer inside ws for the session sources' gradient rows (tcar_grads_t.rows_out)
 *   tcar_segsum_norms_buffer [B*T] buffer inside ws for their squared norms (tcar_grads_t.norms_out)
 *   tcar_segsum_apply       mode 0: g_item[row] += sum of those rows per destination; *sqn_slot += sum of the norms buffer
 *                           (S5); *dense_slot += sum of tcar_sqnorm_det's partials;
 *                           mode 1: g_item[row] += sum over the negatives (b,k) of the row of coef[b] * attout[b, 0:ldh],
 *                           and with `loss`: loss[b] = ce[b] + weight * neg_fb[b]                    model_combine.py:147
 *                           ONE writer per destination row, fixed summation order: bit-for-bit repeatable
 *   tcar_sqnorm_det         block partials of sum g^2 into the last 4096 bytes of ws (folded by mode 0 above) */
int64_t tcar_segsum_ws_bytes(const tcar_dims_t* d, int64_t max_sources);
int tcar_segsum_index(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws, int64_t ws_bytes, void* stream);
float* tcar_segsum_rows_buffer(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws);
float* tcar_segsum_norms_buffer(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws);
int tcar_segsum_apply(const tcar_dims_t* d, const tcar_batch_t* bt, void* ws, int64_t ws_bytes, int mode, const float* rows,
                      const float* coef, const float* attout, int64_t ld_att, float* g_item, float* sqn_slot, float* dense_slot,
                      const float* ce, const float* neg_fb, float weight, float* loss, void* stream);
int tcar_sqnorm_det(const float* g, int64_t len, void* ws, int64_t ws_bytes, void* stream);

/* ---- device-side batch formation and negative sampling (sampler.py:52-113,118-140) ---------------------------------
 * The tensorised session store (CSR over clicks, per-click uint8 features; host/data.py SessionStore) and the negative
 * source (CSR of neighbor_dict / the impression lists) stay resident in HBM; one call writes the packed int32 feed
 *   seq[B,T] | month | day | week | hour+1 | minute+1 [B,T] | dwell bucket [B,T] | click week [B] | click hour [B] |
 *   label [B] | negatives [B,K]
 * of the batch whose example indices are idx[B] (all of input length T).  Negatives follow the reference's rules with a
 * counter-based generator keyed by (seed, counter, example index): same key, same negatives. */
typedef struct {
  const int64_t* off;          /* [n_examples + 1] CSR offsets into the per-click arrays */
  const int32_t* items;        /* [clicks] 1-based item ids; the last click of an example is its label */
  const uint8_t* pub;          /* [clicks, 5] publish month, day, isoweekday, hour+1, minute+1        sampler.py:81-85 */
  const uint8_t* clk;          /* [clicks, 5] click month-1, day-1, isoweekday-1, hour, minute          sampler.py:105-109 */
  const uint8_t* gap_active;   /* [clicks] bucketized(active_t)                                         sampler.py:87 */
  const uint8_t* gap_delta;    /* [clicks] bucketized(seconds to the next click)                        sampler.py:91-94 */
  int64_t n_examples;
} tcar_store_t;
typedef struct {
  int32_t mode;                /* 0 uniform (sampler.py:98-99), 1 neighbour (:133-140), 2 impression (:118-131) */
  const int64_t* off;          /* [n_lists + 1]: neighbour mode: list of 0-based item i; impression mode: list of slot s */
  const int32_t* flat;         /* candidates: 0-based item ids; impression mode: -1 = article outside the catalog */
  const int32_t* slot_of_example;   /* impression mode: [n_examples] example -> list slot */
  int64_t n_lists;
} tcar_negsrc_t;
/* gap_mode: 0 = active_t buckets (sampler.py:87), 1 = click-delta buckets (sampler.py:91-94).  src == NULL: uniform. */
int tcar_form_batch(const tcar_dims_t* d, const tcar_store_t* st, const tcar_negsrc_t* src, const int32_t* idx, int B, int T,
                    int K, int gap_mode, uint64_t seed, uint64_t counter, int32_t* feed, void* stream);

/* ---- catalog-sharded data-parallel step (no reference equivalent: SURVEY.md 8(e)) ------------------------------------
 * Rank r scores the catalog rows [n0, n0 + N) against the sessions of every rank; the softmax over the whole catalog
 * (model_combine.py:145) is split into per-shard statistics, an exchange, and the gradient:
 *   tcar_softmax_stats    stats[b] = (max, sum exp(x - max), logits[b, label[b] - n0] if the label lives here else 0)
 *   tcar_softmax_combine  stats_all [W, B, 3] (all-gathered) -> lse [B], ce [B] = lse - label logit (ce may be NULL);
 *                         label != NULL: sessions with label < 0 (padding of an uneven shard) get lse = +inf, i.e. a zero
 *                         gradient row
 *   tcar_softmax_grad     dlogits = exp(x - lse) - onehot as bf16 hi / lo KB32 planes [ceil128(B), ld] (ld % 32 == 0;
 *                         dl_lo may be NULL: hi-only backward)
 *   tcar_neg_scatter_range  g_item[neg[b,k] - n0, 0:ldh] += coef[b] * attout[b, 0:ldh] for the negatives inside the shard */
int tcar_softmax_stats(int B, int N, const float* logits, int64_t ld, const int32_t* label, int n0, float* stats, void* stream);
int tcar_softmax_combine(int W, int B, const float* stats_all, const int32_t* label, float* lse, float* ce, void* stream);
/* ... and rowstat [B, 2] = (lse, 1) — +inf for a padding session — for tcar_ce_rescale (softmax-epilogue form of the shard's scoring) */
int tcar_softmax_combine_rowstat(int W, int B, const float* stats_all, const int32_t* label, float* lse, float* ce, float* rowstat,
                                 void* stream);
int tcar_softmax_grad(int B, int N, const float* logits, int64_t ld, const float* lse, const int32_t* label, int n0, void* dl_hi,
                      void* dl_lo, void* stream);
int tcar_neg_scatter_range(const tcar_dims_t* d, int64_t B, int K, int n0, int n_loc, const int32_t* neg, const float* attout,
                           int64_t ld_att, const float* coef, float* g_item, void* stream);
/* Packed exchange rows of the catalog-sharded step (one all-gather instead of four; ints travel as bits):
 *   tcar_shard_pack_head    head[b] = [attout[b, 0:ek] | label | negative-term coefficient | Kc negatives | pad], row stride ld
 *                           (ld >= ek + 2 + Kc, ld % 4 == 0); rows B <= b < cap are padding sessions (zero, label -1, negatives -1)
 *   tcar_shard_unpack_head  the all-gathered rows [Bq, ld] back into contiguous label [Bq], coef [Bq], neg [Bq, K]
 *   tcar_shard_pack_ids     rows[r, ldh] = seq[r] (r < n_live) or 0 (padding) behind the packed item-row gradients
 *                           [n_total, ld]; with `loss`: loss[b] = ce[b] + weight * neg_fb[b]          model_combine.py:147 */
int tcar_shard_pack_head(int B, int cap, int ek, int K, int Kc, const float* attout, const int32_t* label, const float* coef,
                         const int32_t* neg, float* head, int64_t ld, void* stream);
int tcar_shard_unpack_head(int Bq, int ek, int K, const float* head, int64_t ld, int32_t* label, float* coef, int32_t* neg,
                           void* stream);
int tcar_shard_pack_ids(int64_t n_live, int64_t n_total, int ldh, const int32_t* seq, float* rows, int64_t ld, int B,
                        const float* ce, const float* neg_fb, float weight, float* loss, void* stream);

/* bumped whenever a struct layout or a signature in this header changes; the loader refuses a mismatch */
#define TCAR_ABI_VERSION 29
int tcar_abi_version(void);
/* hex digest of the sources this binary was compiled from (every .hip and .h under csrc, and this header): loaders compare it with the
 * digest of the sources they sit next to, so a stale binary is detected ("unknown" when built without the in-tree builder) */
const char* tcar_build_id(void);

/* ---- step-level entry points ------------------------------------------------------------------------------------
 * tcar_train_step IS `sess.run([self.loss, self.global_step, self.train_op], feed_dict)` (model_combine.py:231) and
 * tcar_eval_step IS `sess.run([self.softmax_input, self.cross_loss], feed_dict)` + cau_metrics / top-k
 * (model_combine.py:283,296,301): they sequence the op-level entry points above on `stream` with no host
 * synchronisation and no allocation.  All buffers are caller-owned and described by tcar_ctx_t.
 * Arena segment order (identical offsets in W, G, M, V), index into ctx->off[]: */
enum {
  TCAR_V_POS = 0, TCAR_V_MONTH, TCAR_V_DAY, TCAR_V_WEEK, TCAR_V_HOUR, TCAR_V_MINUTE, TCAR_V_DUR,
  TCAR_V_M_WRES, TCAR_V_S_WRES,                      /* [0..8] accumulate with atomics */
  TCAR_V_M_WIN, TCAR_V_M_WC, TCAR_V_M_WINT, TCAR_V_Q1_W, TCAR_V_Q1_B, TCAR_V_Q2_W, TCAR_V_Q2_B,
  TCAR_V_O_W, TCAR_V_O_B, TCAR_V_S_WIN, TCAR_V_S_WC, TCAR_V_OT_W, TCAR_V_OT_B, TCAR_NVAR
};

typedef struct {
  tcar_dims_t d;
  int32_t splitk;                 /* slabs of the catalog-contraction GEMM (d attout) */
  int32_t slot_of[TCAR_NVAR];     /* norm slot of each arena segment */
  int32_t slot_item;
  float b1, b2, eps, clip, neg_weight;
  /* parameters / optimizer state */
  float* E;                       /* [Npad, ek] */
  float* W; float* Gx; float* M; float* V;   /* arenas; Gx = gradients followed by TCAR_NSLOT norm pieces */
  int64_t arena_n;
  int64_t off[TCAR_NVAR];
  float* big;                     /* [N*ldh | N*pt]: dense item gradient, then the time block of dE */
  float* Mi; float* Vi;           /* [N, ldh] */
  float* sqn_dense;               /* [TCAR_NSLOT] */
  const int32_t* use_dense;       /* [TCAR_NSLOT] */
  const int32_t* mwdhm;           /* [N,5] publish_time_MWDHM */
  const int32_t* inv_n; const int32_t* inv_off; float* ct_ws;   /* inverted index of mwdhm + its workspace */
  tcar_segments_t segs_all, segs_dense;
  /* workspace (sized by the caller for the largest B and B*T it will submit) */
  float *x_icp, *x_pt, *x_act, *click_t, *pre1, *pre2, *q1, *q, *alpha, *pooled, *attout, *logits, *ce, *neg_fb, *loss, *neg_coef, *negpart;
  float *dattout, *dpooled, *dq, *dq1, *dclick, *slabs, *dx_icp, *dx_pt, *dx_act, *dpre1, *dpre2;
  int32_t* rank; int32_t* topk;
  /* scoring precision: 0 = fp32 MFMA, 3 = split-bf16 (hi/lo planes, 3 MFMAs per product, fp32-class accuracy),
   * 1 = plain bf16.  The bf16 modes need the planes below: e16* [Npad, ek] (kept in step with E by
   * tcar_clip_adam_2d_bf16 / tcar_cand_time_fwd_bf16), a16* [B, ek], ap16* [B, ldh+pt], dl16* [B, Npad]. */
  int32_t scoring;
  int32_t scoring_bwd;   /* 0 = same as `scoring`; 1 = the two scoring GRADIENT GEMMs use the hi planes only (plain bf16) */
  void *e16h, *e16l, *a16h, *a16l, *ap16h, *ap16l, *dl16h, *dl16l;
  /* optional auxiliary stream + 4 events (hipStream_t / hipEvent_t, caller-created): the candidate-side time refresh
   * (forward) and the dE chain (backward) run on it concurrently with the session-side chain; NULL = one stream */
  void* stream2; void* ev[6];
  uint32_t* adam_bitmap;    /* [ceil16(ceil(N/32))] zeroed words (whole 64-byte units): rows already updated by the early pass of a split update */
  const int32_t* et_perm;   /* [5, N] position of (k, n) in the inverted index (bf16 scoring modes: dE writes d_et in that order) */
  /* optional device timing of the three full-catalog GEMMs, of the largest session-side small GEMM and of the gather: ev_start /
   * ev_stop hold 5 * ev_n hipEvent_t each ([kind][slot]: kind 0 = logits, 1 = dX = dlogits E, 2 = dE = dlogits^T attout, 3 = the
   * grouped projection launch of both attention layers, modules.py:94-96,126-131, 4 = the step's embedding gather,
   * model_combine.py:54-107), recorded on the stream the kernel is
   * launched on; slots are used round-robin through the host counter ev_cursor[0] (ev_cursor[1] = slot of the step in
   * flight, written by the forward pass); ev_n = 0 disables it */
  void* const* ev_start; void* const* ev_stop; int32_t ev_n; int32_t* ev_cursor;
  /* optional third stream + one event (fused single-rank step only): the weight-gradient GEMM and the dense-weight norms
   * run on it, so that they start the moment their inputs exist instead of queueing behind the aux stream's
   * candidate-time backward; NULL = they follow on the aux stream */
  void* stream3; void* ev3;
  /* optional workspace of the sorted segmented sum (tcar_segsum_*): with it the fused step adds the item-row gradients of
   * the gathers and of the negatives in a fixed order (bit-for-bit repeatable) instead of with float atomics */
  void* segsum_ws; int64_t segsum_bytes;
  /* optional [B, 2*ldh] workspace: with it (split-bf16 modes) the bias gradients and the residual-weight gradients are column
   * sums in a fixed order (tcar_attn_pool_bwd_det + tcar_colsum_det) instead of float atomics */
  float* gw_rows;
  /* optional slab workspace of the weight gradients (K splits of batches over 1,536 rows folded in split order instead of
   * float atomics); 16 * (sum of the nine M*N) floats covers every batch */
  float* wgrad_slabs; int64_t wgrad_slab_floats;
  /* optional slab workspace of the session-side small GEMMs (split-bf16 modes): with (ceil(2ldh/128) + 2 ceil(ldh/128) +
   * ceil(ldt/128) + ceil(5ldt/128)) * B*T * ldh floats the projections of both attention layers (modules.py:94-96,126-131) run as
   * one 128-deep K chunk per workgroup, each chunk into its own slab, folded in slab order by tcar_attn_pool_fwd_slabs; the
   * backward pass reuses it for the input gradient of the output transforms (tcar_attn_pool_bwd_slabs) */
  float* proj_slabs; int64_t proj_slab_floats;
  /* optional workspace of the softmax epilogue (training steps with scoring_bwd == 1): B * (ceil(N / 64) + 8) * 2 + 4 B floats,
   * 16-byte aligned, and two HOST ints that carry the group geometry from the forward to the backward half of a step.  With it
   * the logits GEMM of a training step does not write [B, N] fp32 logits (tcar_gemm_bf16_ce + tcar_ce_finish). */
  float* ce_ws; int64_t ce_ws_floats; int32_t* ce_geo /*host*/;
  /* optional planes of the one-hot form of the candidate-side time scores (training steps with the softmax epilogue): oh16
   * [ceil128(N), 160] from tcar_time_onehot (static), p16h / p16l [ceil128(B), 160] written by tcar_time_scores every step */
  void* oh16; void* p16h; void* p16l;
  /* optional buffers of the one-hot form of the scoring GRADIENTS (with oh16 / p16*; ldt = 64): tclip [160 ldt + 320] clipped time
   * rows + scales (tcar_time_scores_clip), dP [B, 160] = dlogits OH, qz [5 N, 2] the per-candidate (||gy||^2, x . gy) pairs.  With
   * them a training step writes no [N, 5 ldt] block of dE, its dX contracts 2 ldh + 160 columns, and the candidate-side time planes
   * of E are NOT refreshed by training steps (nothing of a training step reads them; evaluation refreshes them) */
  float* tclip; float* dP; float* qz;
  /* optional words of the flag forks (step.hip fork_arm / fork_go): sig_dev = 33 zeroed device words (16 workgroup counters,
   * 16 flags, 1 error count), fork_host = tcar_fork_state_bytes() zeroed HOST bytes owned by this context (the driver's fork
   * slots and epoch counter live there — nothing is kept per thread or per process, so contexts may be stepped from different
   * host threads).  With them the main stream records no event where a side stream is forked: the producing kernel publishes a
   * flag, a one-wave kernel of the side stream polls it.  sig_err_host: optional DEVICE-ACCESSIBLE HOST word (pinned, mapped),
   * zeroed: a poll that gives up (the streams do not run concurrently) also counts there with a system-scope atomic, so the
   * host can test it after every step without synchronising; the engine raises when it or sig_dev[32] is non-zero. */
  uint32_t* sig_dev; void* fork_host /*host*/; uint32_t* sig_err_host;
  /* optional tuning copy of THIS context (NULL = the process-wide values, tcar_tuning_defaults) */
  const tcar_tuning_t* tune /*host*/;
  /* optional: zeroed device words for order-fixed last-arrival folds (word 0 = arrival counter, back to zero when a launch ends;
   * then one float per 32,768-float chunk of the dense weights: 36 for the reference's shapes).  With it the dense-weight norms of
   * the fused step run several workgroups per variable (tcar_sqnorm: one) — same bits on every rank and every run either way. */
  uint32_t* fold_scratch; int32_t fold_scratch_words;
  /* optional: tcar_small_det_ws_floats() floats for the order-fixed small-table backward of the fused step — with it a long bucket
   * (B * T >= 2,048) splits every table row's sources into chunks with a workgroup each (tcar_small_tables_bwd_det); without it the
   * step keeps the row pieces in the tail of segsum_ws and one workgroup per table row */
  float* small_det_ws; int64_t small_det_ws_floats;
  /* optional buffers of the ANCHORED softmax form (fused training steps in the one-hot form;
   * TCAR_FUSED_CE = 2; tcar_ce_anchor_fold): ce_rowscale [B, 2] = (1 / S_b, one-hot residual), aps16h = the packed attout plane
   * scaled per row, [ceil128(B), ldh + 5 ldt] bf16 (KB32), ce_form = ONE host int that carries the form the forward half of a step
   * chose to its backward half.  With them a step's plane of exponentials is never rescaled: 94 MB of traffic and one [B, N] pass
   * less per step at the Globo shape. */
  float* ce_rowscale; void* aps16h; int32_t* ce_form /*host*/;
} tcar_ctx_t;

/* Probe before setting tcar_ctx_t.sig_dev: does a polling kernel on `side_stream` run beside a kernel enqueued BEHIND it on
 * `main_stream`?  *concurrent = 0 (a tool serialises kernels, or the two streams share a hardware queue): leave sig_dev NULL.
 * Synchronises both streams; costs one 20-ms time-out when the answer is no. */
int tcar_flag_fork_selftest(uint32_t* sig_dev, void* main_stream, void* side_stream, int32_t* concurrent);
/* bytes of tcar_ctx_t.fork_host */
int64_t tcar_fork_state_bytes(void);
/* sizeof(tcar_ctx_t) of this build: a binding that mirrors the struct checks its own size against it when it loads the library */
int64_t tcar_ctx_bytes(void);
/* diagnostic: one polling kernel on `stream` that gives up after ~10 us (it waits for an epoch nobody publishes): sig_dev[32] and
 * *err_host (may be NULL) each count one time-out — what a step whose streams do not overlap leaves behind */
int tcar_flag_poll_expire(uint32_t* sig_dev, uint32_t* err_host /* device-accessible host word */, void* stream);

/* forward through the full-catalog logits (model_combine.py:52-138); refresh_time != 0 rebuilds E[:, ic:ek] first */
int tcar_step_forward(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, void* stream);
/* loss + every gradient that needs no other rank (everything except the row scatter / candidate-side clip backward) */
int tcar_step_backward_local(const tcar_ctx_t* c, const tcar_batch_t* bt, void* stream);
/* single-GPU completion: dense item norm, embedding backward (scatter), candidate-side backward, dense norms */
int tcar_step_finish(const tcar_ctx_t* c, const tcar_batch_t* bt, void* stream);
/* per-variable clip + Adam with the bias-corrected rate lr_t */
int tcar_step_update(const tcar_ctx_t* c, float lr_t, void* stream);
int tcar_train_step(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, float lr_t, void* stream);

/* tcar_train_step_deferred: forward + backward of `bt` WITHOUT its optimizer update; if `pending` != 0 the update of the
 * PREVIOUS step (bias-corrected rate lr_pending) is applied first, split: arena + the item rows of bt's sessions on the
 * main stream, all other item rows on the aux stream beside this forward pass (they are only needed by the logits GEMM).
 * The caller owes one tcar_step_update (or a further deferred step) for `bt`.  Same results as tcar_train_step. */
int tcar_train_step_deferred(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, int pending, float lr_pending,
                             void* stream);
/* The form a fused training step of `bt` takes on this context — the driver's own predicates, nothing is launched: form[0] softmax
 * epilogue in the logits GEMM (no fp32 logits), form[1] one-hot time segment in the logits GEMM, form[2] one-hot form of the two
 * scoring-gradient GEMMs, form[3] order-fixed (sorted) item-row sum, form[4] anchored softmax form (tcar_ce_anchor_fold: no rescale
 * pass).  For tools that label measurements (bench.py). */
int tcar_step_form(const tcar_ctx_t* c, const tcar_batch_t* bt, int32_t* form /*host, 5 ints*/);
/* rank [B], topk [B,k], ce [B] (the logits buffer is consumed) */
int tcar_eval_step(const tcar_ctx_t* c, const tcar_batch_t* bt, int refresh_time, int k, void* stream);

/* Step-level pieces of the catalog-sharded step, sequenced from C++ like tcar_train_step (sharded.py only adds the
 * collectives between them).  `c` is the rank's context: its SESSION side (tables, workspace) spans the whole catalog
 * (c->d.n_items = N, c->E = the whole candidate matrix), its CANDIDATE side (e16*, big, Mi, Vi, mwdhm, inv_*, et_perm) covers
 * the shard s->n0 .. s->n0 + s->n_loc only. */
typedef struct {
  int32_t world, cap, n0, n_loc;     /* ranks; sessions every rank contributes; first catalog row and row count of the shard */
  const float* att_all;              /* [world*cap, ek]  all-gathered attout (zero rows = padding sessions), row stride ld_att */
  int64_t ld_att;                    /* floats between session rows of att_all (0 = ek): the packed exchange buffer carries label,
                                        negatives and coefficient behind each row */
  const int32_t* lab_all;            /* [world*cap]      labels, -1 = padding session */
  float* logits;                     /* [world*cap, ceil128(n_loc)] */
  float* stats;                      /* [world*cap, 3]   this shard's (max, sum exp, label logit) */
  float* lse; float* ce;             /* [world*cap] */
  void *a16h, *a16l, *ap16h, *ap16l, *dl16h, *dl16l;   /* bf16 planes for ceil128(world*cap) session rows */
  float* slabs;                      /* [c->splitk, world*cap, ek] */
  float* dx;                         /* [world*cap, ek]  this shard's contribution to d attout of EVERY session */
  /* packed exchange (ld_att > ek): tcar_shard_score first unpacks label / negatives / coefficient from the rows of att_all into
   * lab_all (written here), neg_all [world*cap, head_K] and coef_all [world*cap] */
  int32_t head_K;
  int32_t* neg_all;
  float* coef_all;
  /* optional buffers of the ANCHORED softmax form on the shard (TCAR_FUSED_CE = 2): aps16h = the packed
   * attout plane scaled per row [ceil128(world*cap), ldh + 5 ldt] bf16 (KB32), scale2 [world*cap, 2] = (1 / S_b, one-hot residual).
   * With them the shard's plane of exponentials is never rescaled: every shard subtracts the same per-session anchor inside its logits
   * GEMM (tcar_shard_score), the statistics exchange adds plain sums, tcar_shard_backward scales per row. */
  void* aps16h; float* scale2;
  int32_t n_total;                   /* rows of the WHOLE catalog (the anchor reads E[label] of every session's label); 0: no anchored form */
} tcar_shard_t;
/* The form the shard pieces take for (c, s) — their own predicates, nothing is launched: form[0] the one-hot schedule with the softmax
 * epilogue (no fp32 logits of the shard), form[1] the anchored softmax form (no rescale pass over the shard's plane). */
int tcar_shard_form(const tcar_ctx_t* c, const tcar_shard_t* s, int32_t* form /*host, 2 ints*/);
/* forward of the local sessions up to attout (+ the negative term's forward part when bt->K > 0) */
int tcar_step_session_forward(const tcar_ctx_t* c, const tcar_batch_t* bt, void* stream);
/* attout planes, logits = att_all E_shard^T, per-shard softmax statistics; refresh_time != 0 rebuilds the shard's time planes.
 * With the workspaces of the benchmarked single-GPU schedule in the context — ce_ws sized for world*cap rows, the shard's one-hot
 * plane oh16, p16h / p16l for world*cap rows, tclip, dP [world*cap, 160], qz [5 n_loc] — and scoring 3 / scoring_bwd 1 / ldt 64,
 * tcar_shard_score / _backward / _finish run THAT schedule on the shard: the logits GEMM with the softmax epilogue and the one-hot
 * time segment (no fp32 logits; s->logits may then be NULL), tcar_ce_rescale, the one-hot forms of dX and dE and
 * tcar_cand_time_bwd_onehot; the time planes of the shard are never read (pass refresh_time = 0). */
int tcar_shard_score(const tcar_ctx_t* c, const tcar_shard_t* s, int refresh_time, void* stream);
/* stats_all [world, world*cap, 3] -> lse / ce; dlogits planes; dE of the shard (item | time block, local for good);
 * dX partial of every session -> s->dx */
int tcar_shard_backward(const tcar_ctx_t* c, const tcar_shard_t* s, const float* stats_all, void* stream);
/* negative-term rows of ALL sessions inside the shard, the shard's dense item norm (added into the item slot of the norm
 * pieces, which the arena all-reduce sums over the shards: DESIGN.md S5), candidate-side time backward of the shard */
int tcar_shard_finish(const tcar_ctx_t* c, const tcar_shard_t* s, int K, const int32_t* neg_all, const float* coef_all,
                      void* stream);
/* backward of the local sessions from their summed d attout rows dx_rows [B, ek]; the item-row gradients of the gathers go to
 * rows_out [B*T, ldh] (all-gathered by the caller), everything else into the arena gradients */
int tcar_step_session_backward(const tcar_ctx_t* c, const tcar_batch_t* bt, const float* dx_rows, float* rows_out,
                               int64_t rows_ld /* floats between rows of rows_out; 0 = ldh */,
                               int64_t rows_total /* packed form (rows_ld > ldh): ids are written behind the rows, 0 for the
                                                     padding rows B*T <= r < rows_total */,
                               const float* ce_rows /* NULL, or [B]: also loss = ce_rows + neg_weight * neg_fb */, void* stream);
/* first piece of the catalog-sharded step: zero the gradient arena, session forward (+ negative-term forward) of the local
 * sessions — bt may be NULL on a rank whose shard of the batch is empty — and the packed exchange rows (tcar_shard_pack_head).
 * lr_pending >= 0 (ONE rank only: n_loc = n_items, an aux stream, the update marks): the previous step's optimizer update is still
 * owed and is applied here as the single-GPU step's split update (tcar_train_step_deferred) — arena + the item rows this batch
 * gathers first, every other row and the arena zero on the aux stream beside the session forward; < 0: nothing pending.
 * CROSS-CALL INVARIANT of the split update: the update marks (tcar_ctx_t.adam_bitmap) are cleared on the AUX stream behind the event
 * this call's main stream waits for, so no event of THIS call orders the clear before a later reader.  The next reader of the marks
 * — the next tcar_shard_begin's early pass, or tcar_step_update — runs on the main stream, which must have been joined with the aux
 * stream in between: tcar_shard_join does that, and every step of the sharded schedule calls it (sharded.py: Pieces.scatter, also on a
 * rank whose batch shard is empty).  A caller that sequences these pieces itself must keep that join between two begins. */
int tcar_shard_begin(const tcar_ctx_t* c, const tcar_batch_t* bt, int cap, int Kc, float* head, int64_t ld_head,
                     int refresh_time /* as for tcar_shard_score: with an aux stream the shard's time planes are rebuilt there */,
                     int n_loc, float lr_pending, void* stream);
/* squared norms of the dense variables' gradients (arena) in a fixed summation order — the sharded step calls it behind the arena
 * exchange; tcar_sqnorm over the context's dense segments, several workgroups per variable when the context has the fold scratch */
int tcar_step_dense_norms(const tcar_ctx_t* c, void* stream);
/* orders `stream` behind the aux-stream work of the step (tcar_shard_finish, weight gradients): call before
 * tcar_scatter_add_rows_packed and the arena exchange */
int tcar_shard_join(const tcar_ctx_t* c, void* stream);

/* The catalog-sharded step of a rank that exchanges nothing (ONE rank, no live collectives), sequenced in C++: tcar_shard_begin ->
 * _score -> _backward (stats_all = s->stats) -> _finish -> tcar_step_session_backward (dx_rows = s->dx, ce_rows = s->ce) ->
 * tcar_shard_join -> tcar_scatter_add_rows_packed(d_cand, rows, ldr, rows_total, s->n0, c->big) -> tcar_step_dense_norms
 * [-> tcar_step_update(sc, lr_update) when lr_update >= 0].  c: the engine's context, sc: the shard's; s->world must be 1 and
 * s->att_all / ld_att / head_K describe `head`.  Mirrors sharded.ShardExchange.step (session-based-news-recommendation_amd/sharded.py), which
 * stays the sequencer wherever a collective sits between the pieces. */
int tcar_shard_step_local(const tcar_ctx_t* c, const tcar_ctx_t* sc, const tcar_shard_t* s, const tcar_batch_t* bt, int Kc, float* head,
                          int64_t ld_head, int refresh_time, float lr_pending, float* rows, int64_t ldr, int64_t rows_total,
                          const tcar_dims_t* d_cand, float lr_update, void* stream);

/* diagnostic: capture one fused step into a hipGraph and time its replay (tools/graph_probe.py); not a training path */
int tcar_graph_probe(const tcar_ctx_t* c, const tcar_batch_t* bt, float lr_t, int iters, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TCAR_HIP_H */
