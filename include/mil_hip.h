/*
 * mil_hip.h - C ABI of the MI355X (gfx950) hot-path library `libmil_hip.so`.
 *
 * The reference (KyleKWKim/LLM-guided-Multimodal-MIL) has no native layer, plugin registry
 * or FFI: its boundary is the Python class `aggregator(args).forward(x_list, x_CI)`
 * (model/aggregator.py:9-209) whose device work is implicit ATen launches.  This header
 * defines the native boundary the build adds underneath that class.  Each entry point
 * names the reference code whose arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (fp32 unless the name says otherwise); sizes are
 *    host scalars; `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *    nothing synchronises, allocates or frees: capture-safe.
 *  - outputs and workspaces are caller-allocated; ownership never moves.
 *  - return value: 0 on success, a negative MIL_E* code on a rejected argument, or the
 *    positive hipError_t of a failed launch.  No exceptions cross the ABI; no global state.
 *  - bags are concatenated along the row axis: x is [R, L] row-major and bag b owns rows
 *    [bag_off[b], bag_off[b+1]).  One softmax per bag (the reference runs one bag per
 *    forward: model/dim1/ABMIL.py:48,57; test_ddp.py:73).
 *
 * What a binding needs (the CORE - 12 entries), and what the rest is
 *  - the image-only model (BASELINE configs 1 / 2 / 4 / 5): `mil_image_only_step_run` (one call per training step: the struct
 *    at the end of this file) + `mil_gate_bwd_workspace_floats[_bf16]`, `mil_pool_tail_workspace_floats`; for bf16 storage
 *    also `mil_cast_bf16`.
 *  - the input side: `mil_patch_drop_select` (once per epoch), `mil_cohort_feed` (once per step), `mil_set_i32`.
 *  - the fusion model through autograd: one forward + one backward entry per layer family - `mil_gemm` /
 *    `mil_linear_bwd_params` (every nn.Linear), `mil_layernorm_fwd` / `_bwd_res`, `mil_absorbed_pool_value_fwd` /
 *    `mil_absorbed_pool_bwd` (token -> image attention with one text token per bag), `mil_adam_step_dev_segs`.
 *  Everything else is a VARIANT of one of these for a shape or a regime, named by suffix:
 *    `_rows`      capacity-bucket form: the true row count is a device int32 (one ragged bag per step, length on the device)
 *    `_h`, `_head`, `_adam`, `_ws`   the neighbouring stage fused into the launch (head projections, head gradients, Adam, a
 *                 workspace for long bags)
 *    `_bf16`, `_x16`   bf16 storage of x / gates
 *    `_small_`, `_mid_`, `_grouped`, `_nt2` / `_tn2`, `_aux`, `_split`   the same product for <= 64 rows, 65 - 1024 rows, per-bag
 *                 groups, whole rounds of 256 x 256 tiles, an extra epilogue operand, split-bf16 operands
 *    `_pad`       operands / outputs in the grouped products' padded layout
 *    `_counted`, `_dev`, `_segs`   Adam with the step number / learning rate in device memory (hipGraph replay), several ranges
 *    `_time`, `_profile[_rot]`     measurement helpers of the one-call step
 */
#ifndef MIL_HIP_H
#define MIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIL_OK 0
#define MIL_EINVAL (-22)   /* unsupported shape / null pointer */
#define MIL_ENOSPC (-28)   /* workspace too small */

#define MIL_GATE_D 192     /* gate width D of ABMIL (model/dim1/ABMIL.py:7), fixed by the reference */
#define MIL_POOL_TILE 32   /* rows per attention-pool tile (tile map granularity) */
#define MIL_LOSS_BCE 0            /* BCELoss on the sigmoid outputs, mean over B x C  (train_ddp.py:98: num_classes <= 2) */
#define MIL_LOSS_CE_ON_SIGMOID 1  /* CrossEntropyLoss applied to the sigmoid outputs with the one-hot float labels as
                                     class probabilities, mean over B (train_ddp.py:95-96: num_classes > 2) */
#define MIL_SMALL_ROWS 64  /* most rows the token-side mil_linear_small_* entry points accept */

/* Library/ABI version, for the host mirror's load-time check. */
int mil_abi_version(void);   /* 6 */

/* ---- dropout keep bits (train mode) -------------------------------------------------------
 * model.train() upstream drops the bag rows with p = 0.5 BEFORE the gate and pools the dropped rows
 * (model/dim1/ABMIL.py:26,49,59) and drops the bag embedding with p = 0.25 in front of the head
 * (model/aggregator.py:128-131).  Here a keep mask is a packed bit tensor, uint32 [rows][cols/32], bit (col & 31) of
 * word (col >> 5) set = element kept; every entry point below that consumes x (or M) takes such a tensor (NULL = eval
 * mode, no dropout) plus the survivors' scale 1/(1-p).  The backward reads the same bits, so it sees the forward's mask
 * by construction, and a caller (or a parity test) may supply any mask.
 * mil_dropout_keep_bits fills a tensor from Philox4x32-10: key = seed, counter = (128-bit block index, stream offset);
 * offset_dev (nullable) is a device int32 added to `offset`, so a hipGraph replay draws a fresh mask when the counter
 * moves (the trainer points it at its device step counter).  p_drop = 0.5 uses one random bit per element, 0.25 two,
 * anything else a 32-bit threshold.  cols % 32 == 0. */
int mil_dropout_keep_bits(uint32_t* bits, int rows, int cols, float p_drop, uint64_t seed, uint64_t offset,
                          const int32_t* offset_dev, void* stream);
/* Round 4 - the fusion model's route: patch keep bits [R, L/32] (p = 0.5, key seed) and the head's keep words [B, L/32]
 * (p = 0.25, key mseed; mdelta >= 0 stream positions further) at stream position offset + offset_dev[0] in one launch whose last workgroup advances
 * advance[0] by one (advance / done both or neither; done = a zero int32 between launches) - the counter launch and the
 * second generator launch of a captured train-mode step are gone.  R == 0 or B == 0: the other tensor alone. */
int mil_dropout_keep_bits_pair(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed,
                               uint64_t offset, const int32_t* offset_dev, int32_t* advance, int32_t* done, int mdelta,
                               void* stream);
/* counter[0] += v on the stream (a device-side pass counter that a replayed hipGraph advances: dropout offsets). */
int mil_counter_add(int32_t* counter, int v, void* stream);
/* dst[0..n) = values_host[0..n), n <= 8, the values travelling as kernel arguments of one tiny launch (no staging copy): the
 * per-step bag lengths of a capacity bucket (bags.DeviceBagLayout.set_lengths, segments.FusionBucket.set_lengths). */
int mil_set_i32(int32_t* dst, const int32_t* values_host, int n, void* stream);
/* t[row][col] = keep ? t * scale : 0, in place (dropout backward for a consumer that cannot fold the mask in). */
int mil_dropout_apply_bits(float* t, const uint32_t* bits, int rows, int cols, float scale, void* stream);

/* ---- HBM-resident cohort: per-epoch patch drop + per-step feed (csrc/cohort.hip) -----------------
 * Replaces the reference's host-side input pipeline for the pathology bags: dataset.py:366-393 (np.load of one
 * `<patient>.npy` [n, 768] per item, `sorted(random.sample(range(n), int(n * keep)))` with keep = 0.9 / 0.8, zero-pad) and
 * train_ddp.py:193,274-293 (DataLoader workers, pinned memory, `.cuda(non_blocking=True)`).  The cohort lives in ONE flat
 * device buffer `cohort [total_rows, L]`, bag j owning rows [row_off[j], row_off[j+1]).
 *
 * mil_patch_drop_select: one launch per epoch for ALL nbags bags.  sel[out_off[j] + i], i < keep[j], = the cohort rows
 * (absolute row numbers) of a uniformly random keep[j]-subset of bag j's rows in ASCENDING order - the rows holding the
 * keep[j] smallest 32-bit keys, ties by row number, key of row i of bag j = word (i & 3) of
 * Philox4x32-10(counter = (i >> 2, j, epoch_lo, epoch_hi), key = seed ^ 0x70617463685F6472).  keep[j] == n_j copies the
 * identity (no drop).  row_off [nbags + 1], keep [nbags], out_off [nbags + 1] are device int32 arrays.  bag0: the tables
 * describe bags bag0 .. bag0 + nbags - 1 of the cohort (j in the counter = bag0 + table slot), so a bag streamed from the host
 * alone (cohort.HostFeed) draws the subset it would draw as part of the resident cohort. */
int mil_patch_drop_select(const int32_t* row_off, const int32_t* keep, const int32_t* out_off, int nbags, int bag0,
                          uint64_t seed, uint64_t epoch, int32_t* sel, void* stream);

#define MIL_FEED_MAX_BAGS 8
#define MIL_FEED_MAX_AUX 4
/* One step's bags, by value (a HOST struct: everything in it is known on the host without a sync - keep counts follow
 * from the bag lengths).  The nb bags land back to back in dst from row dst_row0; bag b contributes rows[b] rows:
 * cohort rows sel[sel_off[b] + i] when `sel` is given, else src_row0[b] + i (no drop).  The same launch writes
 * len_dev[dst_bag0 + b] = rows[b] and, for each of the naux side tables (labels [nbags, C] fp32, token ids [nbags, P, 77]
 * int64, cached text embeddings [nbags, P, 512] fp32, ...), copies the aux_words[a] 4-byte words of table row bag_id[b] to
 * row dst_bag0 + b of aux_dst[a]. */
typedef struct mil_cohort_feed_desc {
    uint32_t struct_bytes;          /* sizeof(mil_cohort_feed_desc) */
    int32_t nb;                     /* 1 .. MIL_FEED_MAX_BAGS */
    int32_t L;                      /* row width in floats, L % 4 == 0 */
    int32_t dst_row0, dst_bag0;     /* first output row / first output bag slot */
    int32_t naux;                   /* 0 .. MIL_FEED_MAX_AUX */
    int32_t sel_off[MIL_FEED_MAX_BAGS];
    int32_t src_row0[MIL_FEED_MAX_BAGS];
    int32_t rows[MIL_FEED_MAX_BAGS];
    int32_t bag_id[MIL_FEED_MAX_BAGS];
    int32_t aux_words[MIL_FEED_MAX_AUX];
    const void* aux_table[MIL_FEED_MAX_AUX];
    void* aux_dst[MIL_FEED_MAX_AUX];
} mil_cohort_feed_desc;
/* dst [>= dst_row0 + sum rows, L] - typically the static input buffer of a capacity bucket; rows behind the bags are left
 * as they are (every consumer masks them by the device-side lengths).  len_dev nullable. */
int mil_cohort_feed(const float* cohort, const int32_t* sel, const mil_cohort_feed_desc* d, float* dst, int32_t* len_dev,
                    void* stream);

/* ---- tile map ---------------------------------------------------------------------------
 * The attention-pool kernels split every bag into tiles of MIL_POOL_TILE rows.
 * tile_map is int32 [T][4] = {bag, row0, nrows, 0}; bag_tile_off is int32 [B+1] (first tile
 * of each bag).  The host mirror builds both from the bag lengths. */

/* ---- K1a: gate scores -------------------------------------------------------------------
 * s[i] = w . (tanh(Wv x_i + bv) * sigmoid(Wu x_i + bu)) + b      for all R rows.
 * Replaces ABMIL.forward lines 52-54 (model/dim1/ABMIL.py).  fp32 MFMA (v_mfma_f32_32x32x2),
 * exact f32 products/accumulation.  Wv, Wu: [192, L]; bv, bu: [192]; w: [192]; b: [1].
 * gates (nullable): [R, 384] = {V | U} post-activation, saved for the backward.
 * Requires L % 32 == 0, D == 192. */
int mil_gate_scores_fwd(const float* x, const float* Wv, const float* bv, const float* Wu,
                        const float* bu, const float* w, const float* b, float* scores,
                        float* gates, int R, int L, int D, const uint32_t* xbits, float xscale, void* stream);

/* Train-mode forward that DRAWS the dropout keep bits instead of reading them: the x words (Dropout(0.5), ABMIL.py:49) are
 * written to xbits_out [R, L/32] for the later consumers (pool, weight gradient) and - when mbits_out is given - the
 * head's words (Dropout(0.25), aggregator.py:129; [B, L/32]) as well, exactly the words
 * mil_dropout_keep_bits(xbits_out, R, L, 0.5, seed, offset, offset_dev) and
 * mil_dropout_keep_bits(mbits_out, B, L, 0.25, mseed, offset, offset_dev) produce.  Where the shape allows (L <= 1024,
 * L % 128 == 0, every row on the 128-row kernel) the forward kernel draws them itself, each workgroup its own rows'
 * words: no generator launch; otherwise the generator runs first.  L % 64 == 0. */
int mil_gate_scores_fwd_draw(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu,
                             const float* w, const float* b, float* scores, float* gates, int R, int L, int D,
                             uint32_t* xbits_out, float xscale, uint32_t* mbits_out, int B, uint64_t seed, uint64_t mseed,
                             uint64_t offset, const int32_t* offset_dev, void* stream);


/* ---- K1b: attention pool ----------------------------------------------------------------
 * A = softmax over the rows of each bag of s; M[b] = sum_i A_i x_i; lse[b] = logsumexp(s).
 * Replaces ABMIL.forward lines 56-59.  Split-N: one workgroup per tile writes an online-
 * softmax partial (max, sum, weighted row sum), a per-bag workgroup merges them.
 * partials workspace: [T, L + 2] floats.  M: [B, L]; lse: [B]. */
int mil_attn_pool_fwd(const float* x, const float* scores, const int32_t* tile_map,
                      const int32_t* bag_tile_off, int T, int B, int L, float* partials,
                      float* M, float* lse, const uint32_t* xbits, float xscale, void* stream);

/* The two halves of mil_attn_pool_fwd as separate entry points, so a training step can replace
 * the plain merge by the fused tail below.  mil_attn_pool_partial writes the tile partials only. */
int mil_attn_pool_partial(const float* x, const float* scores, const int32_t* tile_map, int T, int L,
                          float* partials, const uint32_t* xbits, float xscale, void* stream);

/* Tile map built on the device from device-resident bag lengths (see mil_image_only_step.bag_len_dev): tile_map
 * [T_cap][4], bag_tile_off [B + 1], rows_out [1] = sum of the lengths.  Tiles past the last real one are {0,0,0,0}:
 * the pool kernels emit a neutral partial for them.  B <= 1024, T_cap >= sum ceil(len / MIL_POOL_TILE). */
int mil_build_tile_map(const int32_t* bag_len, int B, int32_t* tile_map, int32_t* bag_tile_off, int32_t* rows_out,
                       int T_cap, void* stream);

/* Pool partial pass that also emits the head's projection of every patch, hrow[row][c] = x_row . Wf[c] (C <= 4):
 * when ABMIL feeds the linear head directly (model/aggregator.py:199-200), dM = dz Wf, so the backward's
 * x_i . dM equals sum_c dz[bag][c] hrow[i][c] and mil_attn_pool_bwd_from_h yields ds WITHOUT re-reading x
 * (replaces mil_attn_pool_bwd: ds_i = A_i (sum_c dz[bag][c] hrow[i][c] - cdot[bag])). */
/* Train mode: xbits / xscale as above; mbits [B][L/32] / mscale = the head's dropout on the bag embedding, folded into
 * the head rows (hrow[row][c] = x_row^dropped . (Wf[c] * keepM[bag] * mscale)) so the identity above still holds. */
int mil_attn_pool_partial_h(const float* x, const float* scores, const int32_t* tile_map, int T, int L,
                            float* partials, const float* Wf, int C, float* hrow, const uint32_t* xbits, float xscale,
                            const uint32_t* mbits, float mscale, void* stream);
int mil_attn_pool_bwd_from_h(const float* scores, const float* lse, const float* hrow, const float* dz,
                             const float* cdot, const int32_t* tile_map, int T, int C, float* ds, void* stream);

/* Fused per-bag tail (one workgroup per bag): merge the partials -> M, lse; head z, p; and, when
 * labels y are given, the bag's BCE loss loss_bag[b] = scale * sum_c BCE(p_bc, y_bc) (log clamped at -100;
 * summed in fixed order by mil_head_bwd_params), dz, dM = dz Wf and cdot = M . dM, i.e. everything
 * between the pool's partial pass and the pool's backward.  With ds (and tile_map, scores, hrow = the head
 * projections of mil_attn_pool_partial_h) it also writes the score gradient of every row of the bag
 * (= mil_attn_pool_bwd_from_h, folded into this launch); ds requires y.
 * ABMIL.py:57-59 + aggregator.py:128-131,200 + train_ddp.py:99,323-324.  L in {256, 512, 768, 1024}. */
int mil_pool_merge_head(const float* partials, const int32_t* bag_tile_off, int T, int B, int L,
                        const float* Wf, const float* bf, int C, const float* y, float scale, float* M,
                        float* lse, float* z, float* p, float* loss_bag, float* dz, float* dM, float* cdot,
                        const int32_t* tile_map, const float* scores, const float* hrow, float* ds,
                        const uint32_t* mbits, float mscale, float* Mdrop, int loss_kind, void* stream);
/* The same with a workspace (mil_pool_tail_workspace_floats(B) floats, contents irrelevant): a batch of at most 8 LONG bags (on
 * average >= 64 tiles = 2048 rows per bag - the authors' one-bag-per-GPU regime, run_train.sh:81) then spreads every bag's
 * tail over up to 16 + L / 64 workgroups in two short launches instead of walking it in one or two workgroups (ragged
 * image-only step 0.157 -> 0.150 ms).  tail_ws NULL = mil_pool_merge_head. */
size_t mil_pool_tail_workspace_floats(int B);
int mil_pool_merge_head_ws(const float* partials, const int32_t* bag_tile_off, int T, int B, int L, const float* Wf,
                           const float* bf, int C, const float* y, float scale, float* M, float* lse, float* z, float* p,
                           float* loss_sum, float* dz, float* dM, float* cdot, const int32_t* tile_map,
                           const float* scores, const float* hrow, float* ds, const uint32_t* mbits, float mscale,
                           float* Mdrop, int loss_kind, float* tail_ws, void* stream);
/* mbits / mscale (nullable / 1): keep bits of the head's Dropout(.25); M stays the un-dropped pool output, the head
 * reads M * keep * mscale (also written to Mdrop [B, L] when given: it is the M of dWf = dz^T M), dM carries the mask. */

/* ---- K3b: per-bag head ------------------------------------------------------------------
 * z = M Wf^T + bf (logits), p = sigmoid(z).  model/aggregator.py:128-131,200 (eval: the
 * Dropout(0.25) is the identity).  Wf: [C, L]; z, p: [B, C]; C <= 32. */
int mil_head_fwd(const float* M, const float* Wf, const float* bf, float* z, float* p, int B,
                 int L, int C, void* stream);

/* BCELoss(mean) on p vs one-hot float labels y, and its gradient through the sigmoid:
 * loss_sum[0] += sum_bc -[y log p + (1-y) log(1-p)] * scale   (log clamped at -100 as torch),
 * dz = (p - y) * scale.  train_ddp.py:99,323-324.  scale = 1 / (C * global_bags).
 * loss_sum must be zeroed by the caller (it is accumulated so ranks/steps can share it). */
int mil_bce_fwd_bwd(const float* p, const float* y, float* loss_sum, float* dz, int B, int C,
                    float scale, void* stream);

/* Head backward: given dz (or dp when p != NULL: dz = dp * p * (1 - p)),
 * dM = dz Wf, dWf = dz^T M, dbf = sum_b dz, cdot[b] = M[b] . dM[b]. */
int mil_head_bwd(const float* dz_or_dp, const float* p, const float* M, const float* Wf, float* dM,
                 float* dWf, float* dbf, float* cdot, int B, int L, int C, void* stream);

/* Parameter half of mil_head_bwd alone (dWf = dz^T M, dbf = sum_b dz), for use after the fused tail; if
 * loss_bag [B] is given it also writes loss_out[0] = sum_b loss_bag[b] (overwrite: no memset, no atomics). */
int mil_head_bwd_params(const float* dz, const float* M, float* dWf, float* dbf, int B, int L, int C,
                        const float* loss_bag, float* loss_out, void* stream);
/* The same with accumulate != 0 adding to dWf, dbf and loss_out (gradient accumulation over micro-batches). */
int mil_head_bwd_params_acc(const float* dz, const float* M, float* dWf, float* dbf, int B, int L, int C,
                            const float* loss_bag, float* loss_out, int accumulate, void* stream);

/* CLIP-as-loss of the image-only variant (reference utils.py:247-284 CLIPloss_v1, forward :261-284):
 * out [b, E] bag embeddings, feat [b, F, E] frozen CLIP text features of the F per-feature prompts of each sample;
 * logits_f = out feat_f^T [b, b], identity targets, cross entropy over the bag axis, mean over F*b.
 * loss [1] and d_out [b, E] are overwritten.  b <= 64.  workspace: 4*ceil(F/4) + F*b*E floats. */
int mil_clip_contrastive_loss(const float* out, const float* feat, int b, int F, int E, float* loss, float* d_out,
                              float* workspace, void* stream);

/* out[b] = a[b] . c[b] for two [B, L] matrices (cdot = M . dM when dM comes from autograd). */
int mil_rowdot(const float* a, const float* c, float* out, int B, int L, void* stream);

/* ---- K1 backward ------------------------------------------------------------------------
 * ds_i = A_i (x_i . dM[bag] - cdot[bag]),  A_i = exp(s_i - lse[bag]).
 * If dx != NULL also writes the pool term dx_i = A_i dM[bag] (overwrites dx). */
int mil_attn_pool_bwd(const float* x, const float* scores, const float* lse, const float* dM,
                      const float* cdot, const int32_t* tile_map, int T, int L, float* ds,
                      float* dx, const uint32_t* xbits, float xscale, void* stream);

/* Gate parameter gradients from the saved gates and ds:
 * dWv = dPreV^T x, dWu = dPreU^T x, dbv, dbu, dw = sum_i ds_i V_i U_i, db = sum_i ds_i,
 * with dPreV = ds w U (1 - V^2), dPreU = ds w V U (1 - U).  Split-K fp32 MFMA GEMM over the
 * rows + a reduce kernel.  workspace: mil_gate_bwd_workspace_floats(R, L) floats.
 * If accumulate != 0 the results are added to the outputs, else they overwrite them. */
size_t mil_gate_bwd_workspace_floats(int R, int L);
int mil_gate_bwd_params(const float* x, const float* gates, const float* ds, const float* w,
                        int R, int L, int D, float* workspace, size_t workspace_floats,
                        float* dWv, float* dbv, float* dWu, float* dbu, float* dw, float* db,
                        int accumulate, const uint32_t* xbits, float xscale, void* stream);

/* The two launches of mil_gate_bwd_params on their own: the split-K MFMA kernel (fills the workspace) and the
 * reduce (workspace -> outputs). */
int mil_gate_bwd_partials(const float* x, const float* gates, const float* ds, const float* w, int R, int L,
                          int D, float* workspace, size_t workspace_floats, const uint32_t* xbits, void* stream);
/* mil_gate_bwd_partials sized for R rows (grid, split-K plan, workspace) of which only rows_dev[0] <= R are real. */
int mil_gate_bwd_partials_rows(const float* x, const float* gates, const float* ds, const float* w, int R, int L,
                               int D, float* workspace, size_t workspace_floats, const uint32_t* xbits,
                               const int32_t* rows_dev, void* stream);
int mil_gate_bwd_reduce(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                        float* dw, float* db, int accumulate, float xscale, void* stream);

/* mil_gate_bwd_params with the head's parameter gradients (mil_head_bwd_params: dWf [C, L], dbf [C], loss_out = sum of
 * loss_bag) computed by workgroups appended to the reduce launch - one launch less per training step; the bag
 * embeddings M [B, L] have the gate's input width L (the image-only model, aggregator_clip.py:79-118). */
int mil_gate_bwd_params_head(const float* x, const float* gates, const float* ds, const float* w, int R, int L, int D,
                             float* workspace, size_t workspace_floats, float* dWv, float* dbv, float* dWu, float* dbu,
                             float* dw, float* db, int accumulate, const float* dz, const float* M, float* dWf,
                             float* dbf, int B, int C, const float* loss_bag, float* loss_out, const uint32_t* xbits,
                             float xscale, void* stream);

/* The reduce launch of mil_gate_bwd_params_head on its own (after mil_gate_bwd_partials). */
int mil_gate_bwd_reduce_head(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                             float* dw, float* db, int accumulate, float xscale, const float* dz, const float* M,
                             float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out, void* stream);

/* mil_gate_bwd_reduce_head with torch.optim.Adam (train_ddp.py:115-118) applied by the very threads that produce the final
 * gradients - valid when nothing sits between gradient and update (world size 1).  dWv .. dbf must lie inside
 * grad_flat [n_param]; param_flat / exp_avg / exp_avg_sq are indexed alike; step >= 1.  One launch less per step. */
int mil_gate_bwd_reduce_head_adam(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                                  float* dw, float* db, int accumulate, float xscale, const float* dz, const float* M,
                                  float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out,
                                  float* param_flat, const float* grad_flat, size_t n_param, float* exp_avg,
                                  float* exp_avg_sq, int step, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, float grad_scale, void* stream);


/* Input gradient through the gate (needed when the bag is itself a computed tensor, i.e.
 * the fused text+image path): dx += dPreV Wv + dPreU Wu. */
/* With xbits this launch, the last writer of dx, also applies the backward of the patch dropout: dx = keep ? dx * xscale : 0. */
int mil_gate_bwd_input(const float* gates, const float* ds, const float* w, const float* Wv,
                       const float* Wu, int R, int L, int D, float* dx, const uint32_t* xbits, float xscale,
                       void* stream);
/* The same with the pool's own input gradient formed in the epilogue instead of read from dx:
 * dx[row] = a_row dM[row_bag[row]] + dPreV Wv + dPreU Wu, a_row = exp(scores[row] - lse[bag]) (ABMIL.py:57-59: the
 * d/dx of M = sum_i a_i x_i at fixed weights; the part through the scores is the gate term).  dx is WRITTEN, never read:
 * no pool-backward pass over [R, L].  row_bag int32 [R] = the bag of every row (< 0: padding row of a capacity bucket -
 * no pool term, see mil_build_fusion_segs); dM [B, L]. */
int mil_gate_bwd_input_pool(const float* gates, const float* ds, const float* w, const float* Wv, const float* Wu, int R,
                            int L, int D, float* dx, const uint32_t* xbits, float xscale, const float* scores,
                            const float* lse, const int32_t* row_bag, const float* dM, void* stream);

/* ---- K1, bf16-storage variant (BASELINE config 5: N=4096, D=1024) -------------------------------
 * x and the gate weights are stored as bf16 (uint16_t bit patterns); all accumulation is fp32.  Same
 * arithmetic as the fp32 entry points on rounded inputs (model/dim1/ABMIL.py:47-59); the deviation from
 * the fp32 oracle is reported, the 1e-3 bar applies to the fp32 path.  L % 64 == 0 (gate), L in {512,1024} (pool). */
int mil_cast_bf16(const float* src, uint16_t* dst, size_t n, void* stream);
/* gates (fp32 [R, 384], nullable) and/or gates16 (bf16 [R, 384], nullable): the saved {V | U}.  The bf16 form is what
 * mil_gate_bwd_params_bf16 reads: half the bytes of the largest tensor of the step, written with 16-byte stores. */
int mil_gate_scores_fwd_bf16(const uint16_t* x, const uint16_t* Wv, const float* bv, const uint16_t* Wu,
                             const float* bu, const float* w, const float* b, float* scores, float* gates,
                             int R, int L, int D, uint16_t* gates16, const uint32_t* xbits, float xscale, void* stream);
/* Train mode (xbits / xscale, mbits / mscale: the keep-bit tensors of mil_dropout_keep_bits, nullable / 1): as in the fp32
 * entry points the kernels read x through the mask - no dropped copy of the bf16 bag either.  With xbits the forward runs
 * on the 128-row kernel for every R. */
int mil_attn_pool_partial_bf16(const uint16_t* x, const float* scores, const int32_t* tile_map, int T, int L,
                               float* partials, const uint32_t* xbits, float xscale, void* stream);
/* mil_attn_pool_partial_bf16 with the head-projection by-product of mil_attn_pool_partial_h (hrow [R, C] = x Wf^T on the
 * stored bf16 values, C <= 4): the backward then takes mil_attn_pool_bwd_from_h and never re-reads x. */
int mil_attn_pool_partial_h_bf16(const uint16_t* x, const float* scores, const int32_t* tile_map, int T, int L,
                                 float* partials, const float* Wf, int C, float* hrow, const uint32_t* xbits, float xscale,
                                 const uint32_t* mbits, float mscale, void* stream);
int mil_attn_pool_bwd_bf16(const uint16_t* x, const float* scores, const float* lse, const float* dM,
                           const float* cdot, const int32_t* tile_map, int T, int L, float* ds, const uint32_t* xbits,
                           float xscale, void* stream);
/* Gate parameter gradients on the bf16 MFMA: x and dPre rounded to bf16, fp32 accumulation, fp32 partials and
 * outputs (same contract as mil_gate_bwd_params; L % 256 == 0; workspace mil_gate_bwd_workspace_floats_bf16). */
size_t mil_gate_bwd_workspace_floats_bf16(int R, int L);
int mil_gate_bwd_params_bf16(const uint16_t* x, const uint16_t* gates16, const float* ds, const float* w, int R, int L,
                             int D, float* workspace, size_t workspace_floats, float* dWv, float* dbv, float* dWu,
                             float* dbu, float* dw, float* db, int accumulate, const uint32_t* xbits, float xscale,
                             void* stream);
/* mil_gate_bwd_params with x stored as bf16 (widened while staged; fp32 MFMA product: exact on the rounded x). */
int mil_gate_bwd_params_x16(const uint16_t* x, const float* gates, const float* ds, const float* w, int R, int L,
                            int D, float* workspace, size_t workspace_floats, float* dWv, float* dbv, float* dWu,
                            float* dbu, float* dw, float* db, int accumulate, const uint32_t* xbits, float xscale,
                            void* stream);

/* ---- K3a: generic fp32-MFMA GEMM with fused epilogue -----------------------------------------
 * C[M,N] (+)= act(A_op[M,K] . B_op[K,N] + bias) + residual
 *   a_mode 0: A_op[i][k] = A[i*lda + k];   a_mode 1: A_op[i][k] = A[k*lda + i]
 *   b_mode 0: B_op[k][j] = B[j*ldb + k] (nn.Linear weight [N,K]);   b_mode 1: B_op[k][j] = B[k*ldb + j]
 * Forms used: y = x W^T (0,0)  [fc_pathology aggregator.py:47, Attention projections sam/transformer.py:413-416,
 * MLPBlock sam/common.py:21-26, CLIP blocks clip/model.py:171-178];  dx = dy W (0,1);  dW = dy^T x (1,1).
 * act: 0 none, 1 tanh, 2 relu, 3 QuickGELU x*sigmoid(1.702x) (clip/model.py:162-164).  bias [N] / residual
 * [M, ldr] may be NULL.  k-contiguous operands need K % 32 == 0; all leading dimensions % 4 == 0.
 * With a workspace of mil_gemm_workspace_floats() floats (0 = not worth it) K is split across workgroups
 * and a second launch sums the partials and applies the epilogue: always for the (1,1) weight-gradient form,
 * for the others only when the output is a few tiles (token-side projections with a few dozen rows). */
size_t mil_gemm_workspace_floats(int M, int N, int K, int a_mode);
int mil_gemm(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
             int M, int N, int K, const float* bias, int act, const float* residual, int ldr,
             int accumulate, float* workspace, size_t workspace_floats, void* stream);
/* mil_gemm for a CAPACITY BUCKET (a_mode 0): rows_dev = device int32 with the true number of rows of A / C (<= M, the capacity the
 * launch is sized for; the bag lengths of fusion_step.RaggedFusionStepper live on the device).  Output tiles wholly behind
 * it are written as zeros without being computed, so the launch's time follows the true row count.  NULL = mil_gemm. */
int mil_gemm_rows(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc, int M, int N,
                  int K, const float* bias, int act, const float* residual, int ldr, int accumulate, float* workspace,
                  size_t workspace_floats, const int32_t* rows_dev, void* stream);

/* Tall NT products C[M, N] = act(A[M, K] W[N, K]^T + bias), act in {0 none, 1 tanh, 2 ReLU}, on the low-VALU LDS-DMA
 * pipeline of the gate forward kernel (256 x 256 tiles, csrc/linear_nt2.hip): what mil_gemm dispatches nn.Linear layers on
 * tall activations to when the tiles fill whole rounds of the chip (fc_pathology, model/aggregator.py:47,141-149).
 * N % 256 == 0, K % 32 == 0, K >= 64, lda / ldw multiples of 4, A and W 16-byte aligned (mil_gemm_nt2_ok). */
/* The matching weight-gradient product (TN, split over the rows): partial [S][N][K] = sum over a row chunk of
 * (dY (.) act'(Y))^T X and cs_partial [S][N] = the column sums of the same operand, S = mil_gemm_tn2_splits(rows, N, K);
 * what mil_linear_bwd_params dispatches to for tall activations with whole 128 x 128 output tiles (mil_gemm_tn2_ok). */
int mil_gemm_tn2_ok(int lddy, int ldy, int ldx, int rows, int N, int K);
int mil_gemm_tn2_splits(int rows, int N, int K);
int mil_gemm_tn2(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx, int rows, int N, int K,
                 float* partial, float* cs_partial, void* stream);
int mil_gemm_nt2_ok(int lda, int ldw, int M, int N, int K);
int mil_gemm_nt2(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                 const float* bias, int act, void* stream);

/* nn.Linear on 65 .. ~1000 rows (T text tokens x bags on the token side of sam/transformer.py:413-416 / common.py:21-26,
 * the few-hundred-row text tower of a one-bag learnable-prompt step, clip/model.py:171-178): one launch per product, the
 * workgroup of a 32 x 32 / 64 x 64 output tile contracts all of K (csrc/mid_linear.hip), no workspace.
 *   fwd: y = act(x W^T + b) + residual       x [M, K], W [N, K]; K % 8 == 0; act 0 none, 1 tanh, 2 relu, 3 QuickGELU
 *   bwd: dx = dpre W (needs N % 8 == 0), dW = dpre^T x, db = column sums of dpre, dpre = dy (.) act'(yv) (act 0..2; yv =
 *        the saved OUTPUT, NULL for act 0).  dx / dW / db may be NULL (db needs dW). */
int mil_linear_mid_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias, int act, const float* residual,
                       int ldr, float* y, int ldy, int M, int N, int K, void* stream);
int mil_linear_mid_bwd(const float* dy, int lddy, const float* yv, int ldyv, int act, const float* x, int ldx, const float* W,
                       int ldw, float* dx, int lddx, float* dW, int lddw, float* db, int M, int N, int K, void* stream);

/* Parameter half of a Linear layer's backward (every nn.Linear (+Tanh / ReLU) of aggregator.py:44-68 and
 * sam/transformer.py:413-416, sam/common.py:21-26 above MIL_SMALL_ROWS rows) in one product launch + split-K fold:
 *   dW[n_out, k_in] (+)= (dY (.) act'(Y))^T . X        db[n_out] (+)= column sums of dY (.) act'(Y)
 * dY [rows, n_out] upstream gradient, Y the layer's saved output (needed for act 1 tanh / 2 relu; NULL for act 0),
 * X [rows, k_in] the layer input; db may be NULL.  The activation derivative is applied while dY is staged and the bias
 * gradient is accumulated from the staged operand, so no activation-backward or column-sum pass is launched.
 * workspace: mil_linear_bwd_params_workspace_floats(rows, n_out, k_in) floats. */
size_t mil_linear_bwd_params_workspace_floats(int rows, int n_out, int k_in);
int mil_linear_bwd_params(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx, int rows,
                          int n_out, int k_in, float* dW, int lddw, float* db, int accumulate, float* workspace,
                          size_t workspace_floats, void* stream);
/* The same for a capacity bucket: only the first rows_dev[0] rows (device int32, <= rows) carry gradients; the tall-activation
 * kernel spreads those over its row chunks instead of walking the padding.  NULL = mil_linear_bwd_params. */
int mil_linear_bwd_params_rows(const float* dY, int lddy, const float* Y, int ldy, int act, const float* X, int ldx, int rows,
                               int n_out, int k_in, float* dW, int lddw, float* db, int accumulate, float* workspace,
                               size_t workspace_floats, const int32_t* rows_dev, void* stream);

/* mil_gemm (a_mode 0) with one auxiliary [M, N] tensor touched in the epilogue:
 *   aux_mode 1: aux = the pre-activation (product + bias, before act) is stored too - QuickGELU's backward needs it;
 *   aux_mode 2: the result is multiplied by QuickGELU'(aux) - the activation backward of clip/model.py:162-164 fused
 *               into the product that forms the gradient w.r.t. the activation output. */
int mil_gemm_aux(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc, int M, int N,
                 int K, const float* bias, int act, const float* residual, int ldr, int accumulate, float* workspace,
                 size_t workspace_floats, float* aux, int ldaux, int aux_mode, void* stream);
/* Split-bf16 NT product for FROZEN weights (opt-in: the CLIP text tower under learnable prompts):
 *   C[M, N] = act(A[M, K] . B[N, K]^T + bias) + residual, with the aux modes of mil_gemm_aux.
 * A is fp32 and split into `pieces` bf16 summands while it is staged; B_pieces uint16 [pieces][N][ldb] holds the
 * weights' summands (mil_split_bf16, formed once).  pieces = 2 keeps 3 cross terms (relative error ~2^-16), pieces = 3
 * keeps 6 (~2^-23, the rounding of an fp32 product); accumulation is fp32 (v_mfma_f32_32x32x16_bf16).
 * K % 32 == 0, lda % 4 == 0, ldb % 8 == 0, 16-byte aligned operands. */
int mil_split_bf16(const float* src, uint16_t* dst, size_t n, int pieces, void* stream);
int mil_gemm_split(const float* A, int lda, const uint16_t* B_pieces, int pieces, int ldb, float* C, int ldc, int M, int N,
                   int K, const float* bias, int act, const float* residual, int ldr, float* aux, int ldaux, int aux_mode,
                   void* stream);
/* Grouped form (one group = one bag, rows [grp_off[g], grp_off[g+1])), used by the absorbed multi-token attention
 * where every bag multiplies its rows with its own small matrix (T x H absorbed query / key / value vectors):
 *   a_mode 0: C[rows_g, :N] = A[rows_g, :K] . B_g + bias_g + residual[rows_g]   B_g = B + g * strideB
 *             (b_mode 0: B_g [N, K], b_mode 1: B_g [K, N]);  bias_g = bias + g * strideBias (nullable);  K % 32 == 0
 *   a_mode 1: C_g[M, N] = A[rows_g, :M]^T . B[rows_g, :N],  C_g = C + g * strideC    (M % 4 == 0, N % 4 == 0)
 * max_group_rows bounds the launch grid.  With a workspace of mil_gemm_grouped_workspace_floats(...) floats the a_mode 1
 * form (few output tiles per group) splits each group's rows over several workgroups and folds the partial tiles in
 * a fixed order; results are deterministic either way. */
size_t mil_gemm_grouped_workspace_floats(int a_mode, int G, int max_group_rows, int M, int N);
int mil_gemm_grouped(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
                     const int32_t* grp_off, int G, int max_group_rows, int M, int N, int K, long strideB, long strideC,
                     const float* bias, long strideBias, const float* residual, int ldr, float* workspace,
                     size_t workspace_floats, void* stream);
/* a_mode 0 with pad_rows > 0: C has pad_rows rows, and those outside every group - the padding rows of a capacity bucket
 * (segments.FusionBucket: bag lengths on the device) - are left holding ZERO.  One group and T H <= 96: the product's own
 * launch writes those zeros (no fill in front of it); otherwise C is cleared first. */
int mil_gemm_grouped_pad(const float* A, int lda, int a_mode, const float* B, int ldb, int b_mode, float* C, int ldc,
                         const int32_t* grp_off, int G, int max_group_rows, int M, int N, int K, long strideB, long strideC,
                         const float* bias, long strideBias, const float* residual, int ldr, float* workspace,
                         size_t workspace_floats, int pad_rows, void* stream);
/* out[j] (+)= sum_i Y[i][j]  (bias gradients).  With a workspace of mil_colsum_workspace_floats(M, N) floats
 * a tall matrix is summed in 256-row chunks by many workgroups and folded in a second launch (fixed order). */
size_t mil_colsum_workspace_floats(int M, int N);
int mil_colsum(const float* Y, int ldy, int M, int N, float* out, int accumulate, float* workspace,
               void* stream);
/* dpre = dy * act'(y) from the post-activation output y (act 0 none, 1 tanh, 2 relu). */
int mil_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act, void* stream);

/* nn.Linear on the text-token stream (M <= MIL_SMALL_ROWS rows: one token per bag; model/aggregator.py:44-68,
 * model/sam/transformer.py:413-416, sam/common.py:21-26).  Same arithmetic as mil_gemm's NT form, one launch:
 *   y[M, N] = act(x[M, K] . W[N, K]^T + bias) + residual          K % 16 == 0, ldx % 4 == 0, ldw % 4 == 0.
 * act: 0 none, 1 tanh, 2 relu, 3 QuickGELU, 4 sigmoid. */
int mil_linear_small_fwd(const float* x, int ldx, const float* W, int ldw, const float* bias, int act,
                         const float* residual, int ldr, float* y, int ldy, int M, int N, int K, void* stream);
/* Whole backward of that layer in ONE launch (autograd of the Linear + activation above):
 *   dpre = dy * act'(.)   from y_or_pre = the activation OUTPUT (act 1, 2, 4) or the PRE-activation (act 3); NULL if act 0
 *   dW[N, K] = dpre^T x,  db[N] = column sums of dpre,  dx[M, K] = dpre W.      N % 16 == 0.
 * Any of dx / dW / db may be NULL (not wanted). */
int mil_linear_small_bwd(const float* dy, int lddy, const float* y_or_pre, int ldyv, int act, const float* x, int ldx,
                         const float* W, int ldw, float* dx, int lddx, float* dW, int lddw, float* db, int M, int N,
                         int K, void* stream);
/* Round 4 - the token-side layers take the SUMS autograd would form with elementwise launches as operands:
 *   mil_linear_small_fwd_add   y = act((x + x2) W^T + b) (+ residual); xin [M, K] contiguous receives x + x2 for the weight
 *                              gradient (x2 / xin: both or neither) - `queries + query_pe` in front of a q_proj
 *                              (sam/transformer.py:114-118) without the add launch;
 *   mil_linear_small_bwd_sum   mil_linear_small_bwd on dy + dy2 + dy3 + dy4 (the gradients several consumers sent to the
 *                              layer's output; extras [M, N] contiguous, nullable); dysum [M, N] (nullable, needs dx)
 *                              receives the sum for a weight gradient formed later (mil_linear_small_dw_grouped);
 *   mil_linear_small_ln_bwd3   mil_linear_small_ln_bwd with a third addend g3. */
int mil_linear_small_fwd_add(const float* x, int ldx, const float* x2, int ldx2, float* xin, const float* W, int ldw,
                             const float* bias, int act, const float* residual, int ldr, float* y, int ldy, int M, int N,
                             int K, void* stream);
int mil_linear_small_bwd_sum(const float* dy, int lddy, const float* dy2, const float* dy3, const float* dy4, float* dysum,
                             const float* y_or_pre, int ldyv, int act, const float* x, int ldx, const float* W, int ldw,
                             float* dx, int lddx, float* dW, int lddw, float* db, int M, int N, int K, void* stream);
/* out = a + b (+ c) (+ d) over n floats (n % 4 == 0, 16-byte aligned; c, d nullable, d needs c): one launch for a gradient
 * sum that has no backward kernel to ride on. */
int mil_sum4(const float* a, const float* b, const float* c, const float* d, float* out, int n, void* stream);
/*   mil_linear_small_bwd_split  the dx of mil_linear_small_bwd as nsplit (2 | 4) partial sums over n, N / nsplit a multiple
 *                               of 512: dx_parts [nsplit][M][K] - a 2048-wide contraction (mlp.lin1) spread over 4 x the
 *                               workgroups; the next backward kernel sums the parts while it stages them:
 *   mil_linear_small_ln_bwd5    mil_linear_small_ln_bwd with up to five addends (g4, g5 [M, 512] contiguous, nullable). */
int mil_linear_small_bwd_split(const float* dy, int lddy, const float* y_or_pre, int ldyv, int act, const float* W, int ldw,
                               float* dx_parts, int M, int N, int K, int nsplit, void* stream);
int mil_linear_small_ln_bwd5(const float* g1, int ldg1, const float* g2, int ldg2, const float* g3, int ldg3, const float* g4,
                             const float* g5, const float* u, int ldu, const float* stats, const float* gamma, const float* W,
                             int ldw, float* dx, int lddx, float* du, float* dgamma, float* dbeta, int M, int K, void* stream);
int mil_linear_small_ln_bwd3(const float* g1, int ldg1, const float* g2, int ldg2, const float* g3, int ldg3, const float* u,
                             int ldu, const float* stats, const float* gamma, const float* W, int ldw, float* dx, int lddx,
                             float* du, float* dgamma, float* dbeta, int M, int K, void* stream);

/* LayerNorm folded into its neighbours on the token stream (P -> LayerNorm -> C, sam/transformer.py:287-300; E = 512):
 *   mil_linear_small_ln_fwd   C's forward with the norm applied while its operand is staged:
 *       xn = LN(u) gamma + beta;  xin = xn + x2 (x2 nullable: queries + query_pe);  y = act(xin W^T + b) (+ residual);
 *       writes y [M, N], xn [M, 512], xin [M, 512] (with x2 only) and stats [M, 2] = (mean, rstd).
 *   mil_linear_small_ln_bwd   input gradient of the layer P that feeds the norm, with the norm's backward applied while its
 *       dy operand is staged: the gradient at the norm's output is g1 + g2 (g2 nullable - what the output's other consumer
 *       sent, no add launch), du = LN backward of it, dx = du W_P [M, K]; du [M, 512] is written out as well (residual
 *       branch, P's weight gradient), dgamma / dbeta [512] by one extra workgroup.  dx NULL: du and the sums only.
 * M <= MIL_SMALL_ROWS rows. */
int mil_linear_small_ln_fwd(const float* u, int ldu, const float* gamma, const float* beta, float eps, const float* x2,
                            int ldx2, const float* W, int ldw, const float* bias, int act, const float* residual, int ldr,
                            float* y, int ldy, float* xn, float* xin, float* stats, int M, int N, void* stream);
int mil_linear_small_ln_bwd(const float* g1, int ldg1, const float* g2, int ldg2, const float* u, int ldu,
                            const float* stats, const float* gamma, const float* W, int ldw, float* dx, int lddx, float* du,
                            float* dgamma, float* dbeta, int M, int K, void* stream);

/* The weight / bias gradients of up to MIL_SMALL_DW_MAX few-rows layers in ONE launch: dW_l = (dy_l (.) act'(y_l))^T x_l,
 * db_l = its column sums (the dW half of mil_linear_small_bwd, for every queued layer at once).  The backward chain of the
 * token side then only carries the dx launches; the host queues one descriptor per layer and calls this at the end of the
 * backward pass.  Descriptors are passed by value (kernel arguments). */
#define MIL_SMALL_DW_MAX 32
typedef struct mil_small_dw_desc {
    const float* dy;      /* [M, N] incoming gradient */
    const float* yv;      /* [M, N] saved output (act 1 tanh, 2 ReLU, 4 sigmoid) or pre-activation (3 QuickGELU); NULL for act 0 */
    const float* x;       /* [M, K] layer input */
    float* dW;            /* [N, K] or NULL */
    float* db;            /* [N] or NULL */
    int32_t lddy, ldyv, ldx, lddw, act, M, N, K;
} mil_small_dw_desc;
int mil_linear_small_dw_grouped(const mil_small_dw_desc* descs, int n, void* stream);


/* ---- K2: attention cores, LayerNorm, positional encoding ---------------------------------------
 * softmax(q k^T / sqrt(C)) v per head and per bag, AFTER the q/k/v projections and BEFORE out_proj of
 * model/sam/transformer.py:428-450 (and clip/model.py:171-184 with causal = 1).  q [Tq, H*C], k, v [Tk, H*C]
 * row-major; query rows [q_off[b], q_off[b+1]) attend to key rows [k_off[b], k_off[b+1]).  C in {32, 64}.
 *
 * "rows" form: one thread per (query row, head), keys looped (image->token attention, token self-attention,
 * CLIP).  q_bag [Tq] = bag of each query row.  lse [Tq, H] (nullable) is saved for the backward.
 * The backward needs <= 16 keys per bag; blk_map int32 [nblk][3] = {bag, row0, nrows <= 32} tiles the query
 * rows, bag_blk_off [B+1]; workspace nblk * 2 * 16 * H*C floats. */
int mil_attn_rows_fwd(const float* q, const float* k, const float* v, const int32_t* q_off,
                      const int32_t* k_off, const int32_t* q_bag, int Tq, int H, int C, int causal, float* o,
                      float* lse, void* stream);
int mil_attn_rows_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                      const float* lse, const int32_t* k_off, const int32_t* blk_map,
                      const int32_t* bag_blk_off, int nblk, int B, int H, int C, float* dq, float* dk, float* dv,
                      float* workspace, void* stream);
/* The rows form's backward for ANY number of queries and keys per bag (plain per-(row, head) loops; workspace Tq * H floats):
 * what `--alignment_base CT` needs (sam/transformer.py:78-86: 160 CT tokens as queries) - no shipped run does, so no fast
 * kernel exists for it.  q_bag [Tq] / k_bag [Tk]: row -> bag.  Not causal. */
int mil_attn_rows_bwd_general(const float* q, const float* k, const float* v, const float* o, const float* dout,
                              const float* lse, const int32_t* q_off, const int32_t* k_off, const int32_t* q_bag,
                              const int32_t* k_bag, int Tq, int Tk, int H, int C, float* dq, float* dk, float* dv,
                              float* workspace, void* stream);
/* "pool" form: <= 16 queries per bag over many keys (token->image attention: an H-head attention pool over
 * the patches).  tile_map int32 [ntiles][3] = {bag, key0, nkeys <= 64}, bag_tile_off [B+1].
 * forward workspace: ntiles * 16 * (H*C + 2*H) floats; backward workspace: ntiles * 16 * H*C floats. */
int mil_attn_pool_fwd_mh(const float* q, const float* k, const float* v, const int32_t* q_off,
                         const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles, int B, int Tmax, int H,
                         int C, float* o, float* lse, float* workspace, void* stream);
int mil_attn_pool_bwd_mh(const float* q, const float* k, const float* v, const float* o, const float* dout,
                         const float* lse, const int32_t* q_off, const int32_t* tile_map,
                         const int32_t* bag_tile_off, int ntiles, int B, int Tmax, int H, int C, float* dq, float* dk,
                         float* dv, float* workspace, void* stream);
/* "seq" form: self-attention over whole sequences of <= 96 tokens (q, k, v share q_off), with the causal mask of
 * clip/model.py:324-330 if causal != 0 - the CLIP text blocks (clip/model.py:171-184), forward and, for learnable
 * prompts (model/dim1/CLIP.py:29-62), backward.  One workgroup per (sequence, head): the head's q / k / v rows are
 * staged in LDS and every product runs on fp32 MFMA.  lse [Tq, H] is written by the forward (nullable) and read by
 * the backward.  q, k, v rows are ld floats apart (ld = 3 H C when they are the column blocks of one packed
 * in_proj output, clip/model.py:171-178), dq, dk, dv rows ldd floats; o, dout, lse are dense. */
int mil_attn_seq_fwd(const float* q, const float* k, const float* v, int ld, const int32_t* q_off, int B, int Tmax, int H,
                     int C, int causal, float* o, float* lse, void* stream);
int mil_attn_seq_bwd(const float* q, const float* k, const float* v, int ld, const float* o, const float* dout,
                     const float* lse, const int32_t* q_off, int B, int Tmax, int H, int C, int causal, float* dq,
                     float* dk, float* dv, int ldd, void* stream);
/* QuickGELU x*sigmoid(1.702x) (clip/model.py:162-164): out = act(x) if dy == NULL, else out = dy * act'(x). */
int mil_quickgelu(const float* x, const float* dy, float* out, size_t n, void* stream);
/* ---- one-text-token form of the token->image attention (sam/transformer.py:291-295,113-118 with T = 1) -------
 * The K and V projections are absorbed into H query vectors and H pooled key vectors (csrc/absorbed_attn.hip):
 *   Qp[b][h] = Wk_h^T qp[b][h]              mil_absorb_query      (qp = q_proj output [B, H*C], Wk [H*C, E])
 *   pooled[b][h] = sum_n softmax_n(Qp[b][h] . (keys_n + pe_n) / sqrt(C)) keys_n     mil_absorbed_pool_fwd
 *   o[b][hC + c] = Wv[hC + c] . pooled[b][h] + bv[hC + c]                           mil_value_proj
 * i.e. the [N, E] x [E, H*C] projections of k and v are replaced by one HBM-bound pass over the keys; k_proj.bias
 * drops out of the softmax.  tile_map int32 [ntiles][3] = {bag, key0, nkeys <= 64}; H == 8, E == 512.
 * forward workspace: ntiles * H * (E + 2) floats; backward workspace: ntiles * H * E + 16 * n_keys floats
 * (n_keys = total key rows).  dkeys_acc (nullable): a gradient the keys already received elsewhere, added into dkeys.
 * mil_absorb_query_bwd: dqp [B, H*C] and/or dWk [H*C, E] (either may be NULL) from dQp [B, H, E]; the same two
 * entry points give the value projection's backward (dpooled = absorb_query(do, Wv); dWv = absorb_query_bwd(do, pooled)). */
int mil_absorb_query(const float* qp, const float* Wk, int B, int H, int C, int E, float* Qp, void* stream);
/* The same for T text tokens per bag (B = bags x T rows of qp): the output goes straight into the padded operand layout of
 * the grouped products, Qp [B / T, THp, E] with row t H + h of group g and rows T H .. THp - 1 written as zeros, multiplied
 * by `scale` (the 1 / sqrt(c) of the scores) - no pad / scale launches around it; _bwd_pad reads dQp in that layout.
 * bias [H C] and cb [B / T, THp] (both or neither): cb[g][t H + h] = scale * bias_h . qp[b][h], the constant the OTHER
 * projection's bias adds to a score column (sam/transformer.py:113-118 with q_proj.bias), zeros in the padding.  In the
 * backward dcb (needs bias) adds scale * dcb * bias to dqp and dbias (needs dcb and dWk) receives the bias gradient. */
int mil_absorb_query_pad(const float* qp, const float* Wk, int B, int H, int C, int E, int T, int THp, float scale,
                         const float* bias, float* Qp, float* cb, void* stream);
int mil_absorb_query_bwd_pad(const float* qp, const float* Wk, const float* dQp, int B, int H, int C, int E, int T, int THp,
                             float scale, const float* bias, const float* dcb, float* dqp, float* dWk, float* dbias,
                             void* stream);
int mil_absorb_query_bwd(const float* qp, const float* Wk, const float* dQp, int B, int H, int C, int E, float* dqp,
                         float* dWk, void* stream);
int mil_absorbed_pool_fwd(const float* keys, const float* pe, const float* Qp, const int32_t* k_off,
                          const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles, int B, int H, int C, int E,
                          float* pooled, float* lse, float* workspace, void* stream);
/* pooled [B, H, E]: the forward's result; the softmax backward's row constant cdot_h = dpooled_h . pooled_h is formed inside
 * (round 2 took it as an input computed by a mil_rowdot launch). */
int mil_absorbed_pool_bwd(const float* keys, const float* pe, const float* Qp, const float* lse, const float* dpooled,
                          const float* pooled, const int32_t* k_off, const int32_t* tile_map,
                          const int32_t* bag_tile_off, int ntiles, int n_keys, int B, int H, int C, int E,
                          const float* dkeys_acc, float* dkeys, float* dQp, float* workspace, void* stream);
/* Round 4 - LayerNorm(x + o[bag]) (the block's image->token attention with one text token per bag:
 * model/sam/transformer.py:303-309, every patch receives the same row o[bag]) fused with the NEXT attention site's absorbed
 * pool (:291-295 of the next block, or :113-118): the pool's forward kernel makes the keys it reads (y [rows, E] + stats
 * [rows, 2] = mean, rstd), and in the backward the pool's rank-16 update of dkeys is added by the LayerNorm backward to the
 * gradient it loads - [N, 512] passes saved per pair: 1 forward, 4 backward.  tile_map: the 64-key tile map of the keys
 * (padding tiles of a capacity bucket -> zero rows of y / dx).  x, y, dx, dy_acc [rows, E]; o, d_o [B, E]; dy_acc (nullable)
 * = the gradient y receives from its other consumer.  E = 512, H = 8, C in {32, 64}.
 * workspaces (floats): forward ntiles H (E + 2); backward ntiles H E + 16 n_keys + 3 ntiles E. */
int mil_lnbr_absorbed_pool_value_fwd(const float* x, const float* o, const float* gamma, const float* beta, float eps,
                                     const float* pe, const float* Qp, const int32_t* k_off, const int32_t* tile_map,
                                     const int32_t* bag_tile_off, int ntiles, int B, int H, int C, int E, const float* Wv,
                                     const float* bv, float* y, float* stats, float* pooled, float* lse, float* o_attn,
                                     float* workspace, void* stream);
int mil_lnbr_absorbed_pool_bwd(const float* x, const float* o, const float* gamma, const float* beta, const float* stats, const float* y,
                               const float* pe, const float* Qp, const float* lse, const float* dpooled, const float* pooled,
                               const int32_t* k_off, const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles,
                               int n_keys, int B, int H, int C, int E, const float* dy_acc, float* dx, float* d_o,
                               float* dgamma, float* dbeta, float* dQp, float* workspace, void* stream);
int mil_value_proj(const float* pooled, const float* Wv, const float* bv, int B, int H, int C, int E, float* o,
                   void* stream);
/* pooled in the grouped layout [B / T, THp, E] (mil_absorb_query_pad's; the multi-token pool's result as it stands). */
int mil_value_proj_pad(const float* pooled, const float* Wv, const float* bv, int B, int H, int C, int E, int T, int THp,
                       float* o, void* stream);
/* Fewer launches on the token-side chain (each dependent launch costs ~4 us whatever it computes):
 *   mil_absorbed_pool_value_fwd   mil_absorbed_pool_fwd whose merge launch also forms o = Wv pooled + bv [B, H C]
 *   mil_value_proj_bwd            dpooled [B, H, E], dWv [H C, E], dbv [H C] (nullable) from do [B, H C] in one launch
 * (mil_absorb_query_bwd likewise runs both of its halves in one launch when both are requested). */
int mil_absorbed_pool_value_fwd(const float* keys, const float* pe, const float* Qp, const int32_t* k_off,
                                const int32_t* tile_map, const int32_t* bag_tile_off, int ntiles, int B, int H, int C, int E,
                                const float* Wv, const float* bv, float* pooled, float* lse, float* o, float* workspace,
                                void* stream);
int mil_value_proj_bwd(const float* dO, const float* Wv, const float* pooled, int B, int H, int C, int E, float* dpooled,
                       float* dWv, float* dbv, void* stream);
/* Softmax stages of the multi-token absorbed attention (T text tokens per bag: T x H absorbed vectors, score matrix
 * [rows, ld] with column c = t H + h, columns >= T H are padding and come out as zeros):
 *   mil_grp_col_softmax      in place, over the ROWS of each group (bag) per column     (token -> image)
 *   mil_row_softmax_t        in place, over the T tokens of each (row, head)           (image -> token)
 * and their backwards dS = A (dA - sum A dA) over the same axis.  max_group_rows = rows of the longest group (0 = not
 * known): selects the workgroup shape that holds a whole group in registers (up to 32 768 rows forward, 16 384 backward;
 * longer groups re-read their columns). */
int mil_grp_col_softmax(float* S, int ld, const int32_t* grp_off, int G, int max_group_rows, int TH, void* stream);
int mil_grp_col_softmax_bwd(const float* A, const float* dA, int ld, const int32_t* grp_off, int G, int max_group_rows,
                            int TH, float* dS, void* stream);
/* The same with a workspace of mil_grp_col_softmax_workspace_floats(G, max_group_rows, ld) floats (0: not applicable): at most 8
 * groups of more than 2048 rows (one ragged bag per step) take a row-parallel two-launch form - partial column statistics per
 * 256-row chunk, then every chunk folds them and rewrites its rows - instead of one workgroup per two columns. */
size_t mil_grp_col_softmax_workspace_floats(int G, int max_group_rows, int ld);
int mil_grp_col_softmax_ws(float* S, int ld, const int32_t* grp_off, int G, int max_group_rows, int TH, float* ws, void* stream);
int mil_grp_col_softmax_bwd_ws(const float* A, const float* dA, int ld, const int32_t* grp_off, int G, int max_group_rows,
                               int TH, float* dS, float* ws, void* stream);
int mil_row_softmax_t(float* S, int ld, int R, int T, int H, void* stream);
int mil_row_softmax_t_bwd(const float* A, const float* dA, int ld, int R, int T, int H, float* dS, void* stream);

/* nn.LayerNorm over the last dim E (multiple of 64, <= 512), eps inside the sqrt.  stats [rows, 2] =
 * (mean, rstd), saved for the backward.  backward workspace: mil_layernorm_bwd_blocks(rows) * 2 * E floats. */
int mil_layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int E, float eps, float* y,
                      float* stats, void* stream);
int mil_layernorm_bwd_blocks(int rows);
int mil_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* stats, int rows, int E,
                      float* dx, float* dgamma, float* dbeta, float* workspace, void* stream);
/* The same backward with (a) dres [rows, E] or NULL: the gradient that reached x along the residual branch around the
 * norm (x + f(LN(x)), clip/model.py:183-199, sam/transformer.py:283-309), added into dx here; (b) dgamma = dbeta = NULL
 * for frozen parameters (the CLIP tower under learnable prompts): no parameter sums, workspace may be NULL. */
int mil_layernorm_bwd_res(const float* x, const float* gamma, const float* dy, const float* stats, const float* dres,
                          int rows, int E, float* dx, float* dgamma, float* dbeta, float* workspace, void* stream);

/* LayerNorm(x + o[row_bag[row]]) with the per-bag row o [B, E] added on the way in (one text token per bag: the
 * image->token attention of sam/transformer.py:303-309 degenerates to this add followed by norm4) - the sum is never
 * stored.  The backward recomputes xhat from x + o and returns dx, d_o [B, E] (= the per-bag column sums of dx, folded in
 * fixed order), dgamma, dbeta in two launches.  Precondition (checked by the caller, who knows the bag lengths): every bag
 * has at least mil_layernorm_bagrow_rows_per_block(rows) rows, so that a workgroup's row range meets at most two bags.
 * workspace: mil_layernorm_bwd_blocks(rows) * 4 * E floats.  row_off int32 [B + 1], row_bag int32 [rows]. */
int mil_layernorm_bagrow_fwd(const float* x, const float* o, const int32_t* row_bag, const float* gamma, const float* beta,
                             int rows, int E, float eps, float* y, float* stats, void* stream);
int mil_layernorm_bagrow_bwd(const float* x, const float* o, const int32_t* row_bag, const int32_t* row_off, int B,
                             const float* gamma, const float* dy, const float* stats, int rows, int E, float* dx, float* d_o,
                             float* dgamma, float* dbeta, float* workspace, void* stream);
int mil_layernorm_bagrow_rows_per_block(int rows);

/* out[row] = x[row] + pe[row - row_off[row_bag[row]]]: keys + key_pe (sam/transformer.py:292,304) with the
 * table rows indexed by the position inside the bag (aggregator.py:190 passes pe[:, :N]). */
int mil_add_pe(const float* x, const float* pe, const int32_t* row_bag, const int32_t* row_off, int rows, int E,
               float* out, void* stream);
/* One-text-token fast path of the image->token attention (sam/transformer.py:303-307): with a single key the
 * softmax is 1, so every patch of bag b receives the same row o[b]:  out[row] = x[row] + o[row_bag[row]].
 * mil_segment_colsum is its backward w.r.t. o: out[b] = sum of Y over the rows [row_off[b], row_off[b+1]);
 * workspace (nullable) ceil(max_rows / 256) * B * E floats lets long bags be summed by many workgroups. */
int mil_add_bag_row(const float* x, const float* o, const int32_t* row_bag, int rows, int E, float* out,
                    void* stream);
int mil_segment_colsum(const float* Y, const int32_t* row_off, int B, int max_rows, int E, float* out,
                       float* workspace, void* stream);
/* CT feature map -> tokens (sam/transformer.py:86-98; the map is the CT encoder's output [B, C = 512, D = 160, h, w],
 * model/aggregator.py:139-140, HW = h * w): reduce != 0 (resnetMC3_18): out [B * D, C] = mean over (h, w), permuted;
 * reduce == 0 (medicalNet): out [B * D * HW, C] = flatten(2).permute(0, 2, 1).  The encoders themselves are out of scope:
 * the module accepts the precomputed map. */
int mil_ct_map_tokens(const float* ct, int B, int C, int D, int HW, int reduce, float* out, void* stream);
/* The sinusoidal table of model/aggregator.py:99-106 built on the device: pe [n, E]. */
int mil_sinusoid_pe(float* pe, int n, int E, void* stream);

/* ---- K4: CLIP text tower front/back ends (clip/model.py:339-352; the blocks in between are
 * mil_layernorm_fwd + mil_gemm + mil_attn_rows_fwd(causal)) -------------------------------------
 * x[s][p] = token_embedding[ids[s][p]] + positional_embedding[p];  ids int64 [nseq, ctx]. */
int mil_embed_tokens(const int64_t* ids, const float* table, const float* pos, int nseq, int ctx, int W,
                     float* out, void* stream);
/* out[s] = x[s][argmax_p ids[s][p]]: the row at the EOT token (largest id), x [nseq, ctx, W]. */
int mil_gather_eot(const int64_t* ids, const float* x, int nseq, int ctx, int W, float* out, void* stream);

/* ---- optimizer --------------------------------------------------------------------------
 * torch.optim.Adam step with L2 weight decay folded into the gradient (train_ddp.py:115-118)
 * over a flat fp32 buffer; grad is multiplied by grad_scale first (1/world after all-reduce). */
int mil_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                  int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float grad_scale, void* stream);
/* The same step with the step number kept on the device: step_counter (int32, device) holds the number of steps
 * already taken, the bias corrections use step_counter + 1 and the counter is incremented afterwards - so the
 * launch sequence is identical every step and can be replayed from a hipGraph.  All four buffers 16-byte aligned
 * (also required by mil_adam_step). */
int mil_adam_step_counted(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                          int32_t* step_counter, float lr, float beta1, float beta2, float eps, float weight_decay,
                          float grad_scale, void* stream);
/* The same launch without the increment, for a buffer updated in several segments (parameters without a gradient are
 * skipped, as torch.optim.Adam skips them); advance the counter once afterwards with mil_counter_add. */
int mil_adam_step_counted_noinc(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                const int32_t* step_counter, float lr, float beta1, float beta2, float eps,
                                float weight_decay, float grad_scale, void* stream);

/* mil_adam_step_counted with the learning rate in DEVICE memory too (lr_dev [1] float): the host writes the scheduled rate
 * (utils.py:232-241 adjust_learning_rate) into it before a launch or a graph replay, so one captured step serves every
 * epoch of a schedule.  inc != 0 advances the counter after the update. */
int mil_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int32_t* step_counter,
                      const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                      int inc, void* stream);
#define MIL_ADAM_MAX_SEGS 8
/* mil_adam_step_dev over nseg <= MIL_ADAM_MAX_SEGS contiguous ranges [seg_begin[i], seg_end[i]) (HOST arrays of element
 * offsets into the flat buffers, begins 16-byte aligned) in ONE launch, the step counter advanced by that same launch
 * (inc != 0): optim.FlatAdam's update of the live parameter ranges - torch.optim.Adam skips parameters without a gradient
 * (train_ddp.py:115-118) - was one launch per range plus a one-thread increment.  done_counter: device int32 [1], zero before
 * the first launch; the launch leaves it zero. */
int mil_adam_step_dev_segs(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const size_t* seg_begin,
                           const size_t* seg_end, int nseg, int32_t* step_counter, const float* lr_dev,
                           int32_t* done_counter, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                           int inc, void* stream);

/* torch.optim.SGD step without momentum (train_ddp.py:103-108: the optimizer of the learnable-prompt runs) over a
 * flat fp32 buffer: g = grad_scale * grad + weight_decay * param;  param -= lr * g.  Both buffers 16-byte aligned. */
int mil_sgd_step(float* param, const float* grad, size_t n, float lr, float weight_decay, float grad_scale,
                 void* stream);

/* ---- the image-only training step as ONE call -------------------------------------------------------------
 * Replaces, for the image-only branch (BASELINE configs 2/4/5), the reference's per-step sequence
 *     out = generator(x); loss = criterion(out, y); optimizer.zero_grad(); loss.backward(); optimizer.step()
 * (train_ddp.py:295-348) with one host call that enqueues every launch of the step on `stream`:
 *   MIL_STAGE_DROPBITS   keep bits for the patches (p = .5, ABMIL.py:49) and the bag embeddings (p = .25, aggregator.py:129)
 *   MIL_STAGE_GATE_FWD   gate scores (+ saved gates)                                  ABMIL.py:52-54
 *   MIL_STAGE_POOL       attention-pool tile partials (+ head projections)            ABMIL.py:56-59
 *   MIL_STAGE_TAIL       per-bag merge, head, sigmoid, BCE, dz, dM, ds                aggregator.py:128-131,200; train_ddp.py:323
 *   MIL_STAGE_GATE_BWD   split-K weight gradient of the gate                          autograd of ABMIL.py:52-54
 *   MIL_STAGE_REDUCE     split-K fold + head parameter gradients + loss sum
 *   MIL_STAGE_ADAM       Adam over the flat parameter buffer                          train_ddp.py:115-118,348
 * The host mirror fills the struct once per (model, batch layout) and re-submits it every step: the per-step host cost
 * is one foreign call.  With world size > 1 the gradient all-reduce sits between REDUCE and ADAM: run the step with
 * stages = ALL & ~ADAM, all-reduce `grad_flat`, then run stages = ADAM.
 * All pointers are device pointers owned by the caller.  y == NULL: inference (stages up to TAIL, no gradients).
 * x_bf16 != 0: x, Wv16, Wu16 are bf16 bit patterns (BASELINE config 5); eval-mode only (train != 0 is rejected). */
#define MIL_STAGE_DROPBITS 0x01u
#define MIL_STAGE_GATE_FWD 0x02u
#define MIL_STAGE_POOL     0x04u
#define MIL_STAGE_TAIL     0x08u
#define MIL_STAGE_GATE_BWD 0x10u
#define MIL_STAGE_REDUCE   0x20u
#define MIL_STAGE_ADAM     0x40u
#define MIL_STAGE_TILEMAP  0x80u     /* build tile_map / bag_tile_off / rows_dev on the device from bag_len_dev */
#define MIL_STAGE_ALL      0xffu
/* Hint, not a stage (outside MIL_STAGE_ALL): the caller asserts that tile_map is the plain single-segment map of bags whose
 * lengths are all multiples of 32 - every tile full, tile t = rows 32 t .. 32 t + 31.  With GATE_FWD and POOL in the same
 * call (fp32 x, C == 2, hrow given) the pool partial pass then runs in the epilogue of the gate-forward launch instead of
 * as a launch of its own (same partials / hrow; ABMIL.py:52-59 in one kernel).  Never set it for multi-segment maps. */
#define MIL_STAGE_POOL_FUSED 0x100u
typedef struct mil_image_only_step {
    uint32_t struct_bytes;          /* sizeof(mil_image_only_step): ABI check */
    uint32_t stages;                /* MIL_STAGE_* to run */
    /* batch */
    const void* x;                  /* [R, L] fp32 (or bf16 bits) */
    const float* y;                 /* [B, C] one-hot labels, or NULL */
    const int32_t* tile_map;        /* [T, 4] */
    const int32_t* bag_tile_off;    /* [B + 1] */
    int32_t R, L, B, C, T;          /* with bag_len_dev: R and T are the CAPACITY (bucket) the launches are sized for */
    /* Ragged batches whose lengths change every step (the authors' regime: one bag of 1 000 - 15 592 patches per GPU,
     * 10-20 % of the patches dropped at random per epoch, dataset.py:374-381): bag_len_dev [B] int32 holds the true
     * lengths ON THE DEVICE; MIL_STAGE_TILEMAP rebuilds tile_map [T][4], bag_tile_off [B + 1] (both written, so they must be
     * writable buffers) and rows_dev [1] from it, and the backward masks rows >= rows_dev[0].  Shapes, grids and
     * workspaces depend on the bucket (R, T) only: one struct / one captured graph per bucket.  fp32 x only. */
    const int32_t* bag_len_dev;     /* or NULL: tile_map / bag_tile_off are host-built for exactly R rows */
    int32_t* rows_dev;
    int32_t x_bf16;
    float loss_scale;               /* 1 / (C * global bags) for MIL_LOSS_BCE, 1 / global bags for MIL_LOSS_CE_ON_SIGMOID */
    int32_t loss_kind;              /* MIL_LOSS_* */
    int32_t accumulate;             /* != 0: add this batch's gradients to the buffers (gradient accumulation) */
    /* parameters (fp32 masters; views of one flat buffer in the host mirror) */
    const float *Wv, *bv, *Wu, *bu, *w, *b, *Wf, *bf;
    const uint16_t *Wv16, *Wu16;    /* bf16 shadows of Wv, Wu (x_bf16 only) */
    /* gradients */
    float *dWv, *dbv, *dWu, *dbu, *dw, *db, *dWf, *dbf;
    float* loss_out;                /* [1]: sum over the bags of loss_scale * BCE */
    /* per-step state kept for the backward, caller-allocated */
    float* scores;                  /* [R] */
    float* gates;                   /* [R, 384] fp32 (fp32 x; bf16 x with the fp32-MFMA weight gradient) */
    uint16_t* gates16;              /* [R, 384] bf16 (bf16 x with the bf16-MFMA weight gradient) */
    float* partials;                /* [T, L + 2] */
    float* hrow;                    /* [R, C] (C <= 4) or NULL: the backward then re-reads x */
    float* ds;                      /* [R] */
    float* dw_ws;                   /* split-K workspace of the weight gradient */
    uint64_t dw_ws_floats;
    float *M, *Mdrop, *lse, *logits, *prob, *loss_bag, *dz, *dM, *cdot;   /* [B, L] x2, [B], [B, C] x2, [B], [B, C], [B, L], [B] */
    /* dropout (train != 0) */
    int32_t train;
    int32_t bf16_grad_mfma;         /* x_bf16: weight gradient on the bf16 MFMA (else fp32 MFMA on the widened x) */
    uint32_t* xbits;                /* [R, L/32] */
    uint32_t* mbits;                /* [B, L/32] */
    uint64_t seed, offset;          /* Philox key / stream position of this pass */
    const int32_t* offset_dev;      /* nullable device counter added to offset (hipGraph replay) */
    /* optimizer (MIL_STAGE_ADAM) */
    float *param_flat, *grad_flat, *exp_avg, *exp_avg_sq;
    uint64_t n_param;
    int32_t adam_step;              /* 1-based step number (host bias corrections) ... */
    int32_t* adam_step_dev;         /* ... or a device counter of steps already taken (incremented by the launch) */
    float lr, beta1, beta2, eps, weight_decay, grad_scale;
    const float* lr_dev;            /* nullable, needs adam_step_dev: [1] learning rate read on the device instead of `lr`, so a
                                     * captured step follows the schedule of utils.py:232-241 without re-capture */
    float* tail_ws;                 /* nullable: mil_pool_tail_workspace_floats(B) floats for MIL_STAGE_TAIL (long bags) */
    int32_t* done_dev;              /* nullable, with adam_step_dev: [1] sign-off word, zero before the first step and left zero by
                                     * every step - the fold launch that applies Adam then advances adam_step_dev itself (its last
                                     * workgroup to read the step number does it) instead of a one-thread launch behind it */
} mil_image_only_step;

int mil_image_only_step_run(const mil_image_only_step* a, void* stream);
/* Measurement helper: runs the given stage subset `iters` times back to back on `stream` between two HIP events (after
 * `warm` untimed runs) and returns the average duration of one run in *ms_out.  Synchronises the stream.  The state a
 * stage reads must exist (run the whole step once first).  bench.py's per-kernel figures come from here: the launches
 * are issued from C, so a 10 us kernel is not timed behind 30 us of interpreter. */
int mil_image_only_step_time(const mil_image_only_step* a, uint32_t stages, int warm, int iters, float* ms_out,
                             void* stream);

/* torch.nn.CosineEmbeddingLoss()(x1, x2, target = +1), mean over the B rows, forward + backward in one launch: the optional
 * 'textCosSim' term between the text-aligned tokens x_CT2CI and x_Pth2CI (reference train_ddp.py:102,266,325-329).
 * x1, x2 [B, E]; loss [1] = scale * sum_b (1 - cos_b) (scale = weight / B); dx1, dx2 [B, E] = d loss / d x (both or neither).
 * ATen's arithmetic: cos = x1.x2 / sqrt((|x1|^2 + 1e-12)(|x2|^2 + 1e-12)).  B <= 1024. */
int mil_cosine_embedding_loss(const float* x1, const float* x2, int B, int E, float scale, float* loss, float* dx1,
                              float* dx2, void* stream);

/* Device-side segments of the one-note fusion step for a CAPACITY bucket (cap patch rows, B bags, P text tokens per bag):
 * every map model/aggregator.py:186-192 + sam/transformer.py need (patch offsets, row -> bag, 64-key tiles of the absorbed
 * attention pool, 32-row tiles of the multi-modal bag [patch rows | token rows at cap + b P], row -> bag with -1 on padding
 * rows) rebuilt from len_dev [B] inside the step, so one captured hipGraph per bucket serves every bag length (reference
 * regime: one ragged bag per GPU, length changing every step, dataset.py:366-393, run_train.sh:81).
 * T64 >= cap / 64 + B + 2 (the 64-key tile map ends in PADDING tiles {0, row0, -count} over the rows beyond the bags: empty
 * for the pool kernels, zero-gradient rows for mil_absorbed_pool_bwd), T32 >= cap / 32 + B (1 + ceil(P / 32)); tile32
 * 16-byte aligned; ds_zero [cap + B P] float: the bucket's score-gradient buffer, padding rows zeroed here. */
int mil_build_fusion_segs(const int32_t* len_dev, int B, int P, int cap, int32_t* k_off, int32_t* k_bag, int32_t* tile64,
                          int32_t* bag_tile64_off, int T64, int32_t* tile32, int32_t* bag_tile32_off, int T32,
                          int32_t* row_bag, int32_t* rows_out, float* ds_zero, void* stream);

/* mil_build_fusion_segs for a multi-modal bag with several static segments behind the patch rows (aggregator.py:173, CT +
 * pathology: P text-from-CT tokens, D CT tokens, P text-from-pathology tokens per bag): seg_rows is a HOST array of nseg <= 4
 * per-bag row counts; segment s of bag b sits at cap + B (seg_rows[0] + .. + seg_rows[s-1]) + b seg_rows[s].
 * T32 >= cap / 32 + B (1 + sum ceil(seg_rows[s] / 32)); row_bag / ds_zero cover cap + B sum(seg_rows) rows. */
int mil_build_fusion_segs_tail(const int32_t* len_dev, int B, int nseg, const int32_t* seg_rows, int cap, int32_t* k_off,
                               int32_t* k_bag, int32_t* tile64, int32_t* bag_tile64_off, int T64, int32_t* tile32,
                               int32_t* bag_tile32_off, int T32, int32_t* row_bag, int32_t* rows_out, float* ds_zero,
                               void* stream);

/* In-step timing (bench.py's `roofline` / `kernels_ms`): runs the whole step `iters` times as `ngroups` consecutive
 * mil_image_only_step_run calls (groups[i] = stage mask of group i) with a HIP event recorded on `stream` between the
 * groups; ms_out[i] = average duration of group i where it runs inside the step, ms_out[ngroups] = average first-to-last
 * event.  Synchronises the stream.  The step really executes (parameters move on when MIL_STAGE_ADAM is in a group). */
int mil_image_only_step_profile(const mil_image_only_step* a, const uint32_t* groups, int ngroups, int warm, int iters,
                                float* ms_out, void* stream);
/* The same with the input batch ROTATING: iteration i runs on (xs[i % nrot], ys[i % nrot]) (host arrays of nrot <= 64 device
 * pointers, every batch of the descriptor's shape) - the cache regime of a loop that cycles through several resident
 * batches (bench.py --batches); one batch alone stays in the Infinity Cache from step to step.  nrot == 0: as above. */
int mil_image_only_step_profile_rot(const mil_image_only_step* a, const void* const* xs, const float* const* ys, int nrot,
                                    const uint32_t* groups, int ngroups, int warm, int iters, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIL_HIP_H */
