// The image-only training step as one host call (include/mil_hip.h: mil_image_only_step).
//
// Reference: train_ddp.py:295-348 runs, per step, generator(x) -> criterion -> zero_grad -> backward -> optimizer.step
// through the Python interpreter and ATen's dispatcher (dozens of implicit launches).  Here the step is a fixed
// sequence of eight launches issued from C on the caller's stream; nothing is allocated, nothing synchronises, so the
// call is capture-safe (the host mirror replays it from a hipGraph for tiny one-bag steps) and costs one foreign call.
#include <stdlib.h>

#include "mil_common.h"

// gated_pool.hip: gate forward + (when the batch allows) the pool partial pass in the same launch
int gate_fwd_with_pool(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu, const float* w,
                       const float* b, float* scores, float* gates, int R, int L, int draw, uint32_t* xbits, float xscale,
                       uint32_t* mbits, float mscale, int B, uint64_t seed, uint64_t mseed, uint64_t offset,
                       const int32_t* offset_dev, const int32_t* tile_map, int T, float* partials, const float* Wf, float* hrow,
                       int* fused, void* stream);

// gated_pool.hip: gate forward of a bucketed batch - tiles beyond the true row count (rows_dev) are skipped
int gate_fwd_rows_dev(const float* x, const float* Wv, const float* bv, const float* Wu, const float* bu, const float* w,
                      const float* b, float* scores, float* gates, int R, int L, int draw, uint32_t* xbits, float xscale,
                      uint32_t* mbits, int B, uint64_t seed, uint64_t mseed, uint64_t offset, const int32_t* offset_dev,
                      const int32_t* rows_dev, void* stream, const TileMapJob* tmap);

// gated_pool.hip: split-K fold + head gradients + Adam in one launch, the step number on the host or in a device counter
int gate_bwd_reduce_head_adam_impl(const float* workspace, int R, int L, float* dWv, float* dbv, float* dWu, float* dbu,
                                   float* dw, float* db, int accumulate, float xscale, const float* dz, const float* M,
                                   float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out,
                                   float* param_flat, const float* grad_flat, size_t n_param, float* exp_avg,
                                   float* exp_avg_sq, int step, const int* step_dev, float lr, const float* lr_dev, float beta1,
                                   float beta2, float eps, float weight_decay, float grad_scale, void* stream, int* done);

// gated_pool_bf16.hip: bf16-MFMA weight gradient whose fold launch carries the head gradients, the loss and (param_flat != NULL)
// Adam + the refresh of the bf16 weight shadows
int gate_bwd_params_bf16_tail(const uint16_t* x, const uint16_t* gates, const float* ds, const float* w, int R, int L,
                              float* workspace, size_t workspace_floats, float* dWv, float* dbv, float* dWu, float* dbu,
                              float* dw, float* db, int accumulate, const uint32_t* xbits, float xscale, const float* dz,
                              const float* M, float* dWf, float* dbf, int B, int C, const float* loss_bag, float* loss_out,
                              float* param_flat, const float* grad_flat, size_t n_param, float* exp_avg, float* exp_avg_sq,
                              int step, const int* step_dev, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, uint16_t* Wv16, uint16_t* Wu16, void* stream, int* done);

// gated_pool_bf16.hip: bf16 gate forward with the pool partial pass in its epilogue when the batch allows
int gate_fwd_bf16_with_pool(const uint16_t* x, const uint16_t* Wv, const float* bv, const uint16_t* Wu, const float* bu,
                            const float* w, const float* b, float* scores, uint16_t* gates16, int R, int L, const uint32_t* xbits,
                            float xscale, const int32_t* tile_map, int T, float* partials, const float* Wf, float* hrow,
                            const uint32_t* mbits, float mscale, int* fused, void* stream);

// dropout.hip: both keep-bit tensors of a step in one launch
int dropout_keep_bits_pair(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed, uint64_t offset,
                           const int32_t* offset_dev, void* stream);
int dropout_keep_bits_pair_tilemap(uint32_t* xbits, int R, uint32_t* mbits, int B, int L, uint64_t seed, uint64_t mseed,
                                   uint64_t offset, const int32_t* offset_dev, const TileMapJob& tm, void* stream);

static int step_check(const mil_image_only_step* a) {
    if (!a || a->struct_bytes != sizeof(mil_image_only_step)) return MIL_EINVAL;
    if (!a->x || !a->tile_map || !a->bag_tile_off) return MIL_EINVAL;
    if (a->R < 0 || a->B < 0 || a->T < 0 || a->L <= 0 || a->C <= 0 || a->C > 32) return MIL_EINVAL;
    if (!a->Wv || !a->bv || !a->Wu || !a->bu || !a->w || !a->b || !a->Wf || !a->bf) return MIL_EINVAL;
    if (!a->scores || !a->partials || !a->M || !a->lse || !a->logits || !a->prob) return MIL_EINVAL;
    if (a->x_bf16 && (!a->Wv16 || !a->Wu16)) return MIL_EINVAL;
    if (a->bag_len_dev && (a->x_bf16 || !a->rows_dev || a->B > 1024)) return MIL_EINVAL;
    if (a->train && (!a->xbits || !a->mbits || !a->Mdrop)) return MIL_EINVAL;
    if (a->lr_dev && !a->adam_step_dev) return MIL_EINVAL;       // a device learning rate belongs to the replayable (counted) step
    if (a->y) {
        if ((!a->gates && !a->gates16) || !a->ds || !a->loss_bag || !a->dz || !a->dM || !a->cdot) return MIL_EINVAL;
    }
    return MIL_OK;
}

// MIL_FUSE_POOL=0: keep the pool partial pass a launch of its own (A/B measurements, bit-equality tests)
static bool fuse_pool_enabled() {
    const char* e = getenv("MIL_FUSE_POOL");
    return e == nullptr || e[0] != '0';
}

// bf16 step: the pool pass in the deep forward's epilogue (round 4) is OPT-IN (MIL_FUSE_POOL16=1).  Measured at config 5
// on one box (tools/cfg5_quick.py): eval mode 0.3464 ms fused vs 0.3456 stand-alone - a tie; train mode 0.504 vs 0.415 ms - the
// keep-word handling of eight rows per unit pushes the 256-register forward into scratch.  The re-read is served by the
// Infinity Cache, but FETCH_SIZE counts those requests like HBM ones (profiles/r04_hbm_traffic_pmc.csv: 605 + 118 MiB for the
// fused launch against 327 + 98 and 257 + 17 for the two separate ones), so no traffic saving can be shown either.
static bool fuse_pool16_enabled(bool train) {
    const char* e = getenv("MIL_FUSE_POOL16");
    (void)train;
    return e != nullptr && e[0] != '0';
}

extern "C" int mil_image_only_step_run(const mil_image_only_step* a, void* stream) {
    int rc = step_check(a);
    if (rc != MIL_OK) return rc;
    const uint32_t st = a->stages;
    const bool train = a->train != 0;
    const bool grads = a->y != nullptr;
    const float xscale = train ? 2.0f : 1.0f;                   // 1 / (1 - 0.5)   ABMIL.py:26
    const float mscale = train ? 1.0f / 0.75f : 1.0f;           // 1 / (1 - 0.25)  aggregator.py:129
    const uint32_t* xbits = train ? a->xbits : nullptr;
    const uint32_t* mbits = train ? a->mbits : nullptr;
    const bool use_h = a->hrow != nullptr && grads && a->C <= 4;
    bool adam_in_reduce = false;
    int pool_fused = 0;

    // keep bits drawn by the forward kernel itself (no generator launches) when both stages run in this call
    const bool draw_in_fwd = (st & MIL_STAGE_DROPBITS) && (st & MIL_STAGE_GATE_FWD) && train && !a->x_bf16;
    const bool gen_launch = (st & MIL_STAGE_DROPBITS) && train && !draw_in_fwd;
    const bool want_tmap = (st & MIL_STAGE_TILEMAP) && a->bag_len_dev;
    const TileMapJob tmj{a->bag_len_dev, a->B, const_cast<int32_t*>(a->tile_map), const_cast<int32_t*>(a->bag_tile_off), a->rows_dev,
                         a->T};
    // the fp32 bucketed forward (gate_fwd_rows_dev) takes the tile map along: on its generator launch when it has one
    const bool tmap_with_fwd = want_tmap && (st & MIL_STAGE_GATE_FWD) && !a->x_bf16 && a->rows_dev != nullptr && !gen_launch;
    if (want_tmap && gen_launch) {
        // one launch: the generator's workgroups + one more that builds the tile map (round 4: a bucketed one-bag step began
        // with two launches of 5 us each)
        rc = dropout_keep_bits_pair_tilemap(a->xbits, a->R, a->mbits, a->B, a->L, a->seed, a->seed ^ 0x9E3779B97F4A7C15ull,
                                            a->offset, a->offset_dev, tmj, stream);
        if (rc != MIL_OK) return rc;
    } else if (want_tmap && !tmap_with_fwd) {
        rc = mil_build_tile_map(a->bag_len_dev, a->B, const_cast<int32_t*>(a->tile_map), const_cast<int32_t*>(a->bag_tile_off),
                                a->rows_dev, a->T, stream);
        if (rc != MIL_OK) return rc;
    }
    if (gen_launch && !((st & MIL_STAGE_TILEMAP) && a->bag_len_dev)) {
        // patch bits and (a second key, same stream position) the head's mask, one launch
        rc = dropout_keep_bits_pair(a->xbits, a->R, a->mbits, a->B, a->L, a->seed, a->seed ^ 0x9E3779B97F4A7C15ull, a->offset,
                                    a->offset_dev, stream);
        if (rc != MIL_OK) return rc;
    }
    if (st & MIL_STAGE_GATE_FWD) {
        float* gates = grads ? a->gates : nullptr;
        const bool g16 = a->x_bf16 && a->bf16_grad_mfma && (a->L % 256) == 0 && a->gates16 != nullptr;
        if (a->x_bf16 && g16 && grads && (st & MIL_STAGE_POOL) && (st & MIL_STAGE_POOL_FUSED) && use_h && a->C == 2 &&
            fuse_pool_enabled() && fuse_pool16_enabled(train))
            // forward + pool partial pass in one launch (falls back inside when the shape does not allow it)
            rc = gate_fwd_bf16_with_pool((const uint16_t*)a->x, a->Wv16, a->bv, a->Wu16, a->bu, a->w, a->b, a->scores, a->gates16,
                                         a->R, a->L, xbits, xscale, a->tile_map, a->T, a->partials, a->Wf, a->hrow, mbits, mscale,
                                         &pool_fused, stream);
        else if (a->x_bf16)
            rc = mil_gate_scores_fwd_bf16((const uint16_t*)a->x, a->Wv16, a->bv, a->Wu16, a->bu, a->w, a->b, a->scores,
                                          g16 ? nullptr : gates, a->R, a->L, MIL_GATE_D, (grads && g16) ? a->gates16 : nullptr,
                                          xbits, xscale, stream);
        else if ((st & MIL_STAGE_POOL) && (st & MIL_STAGE_POOL_FUSED) && use_h && a->C == 2 && !a->bag_len_dev &&
                 (!train || draw_in_fwd) && fuse_pool_enabled())
            // forward + pool partial pass in one launch (falls back inside when the tile map is not all full tiles)
            rc = gate_fwd_with_pool((const float*)a->x, a->Wv, a->bv, a->Wu, a->bu, a->w, a->b, a->scores, gates, a->R, a->L,
                                    draw_in_fwd ? 1 : 0, a->xbits, xscale, a->mbits, mscale, a->B, a->seed,
                                    a->seed ^ 0x9E3779B97F4A7C15ull, a->offset, a->offset_dev, a->tile_map, a->T, a->partials,
                                    a->Wf, a->hrow, &pool_fused, stream);
        else if (a->bag_len_dev && a->rows_dev)
            rc = gate_fwd_rows_dev((const float*)a->x, a->Wv, a->bv, a->Wu, a->bu, a->w, a->b, a->scores, gates, a->R, a->L,
                                   draw_in_fwd ? 1 : 0, a->xbits, xscale, a->mbits, a->B, a->seed, a->seed ^ 0x9E3779B97F4A7C15ull,
                                   a->offset, a->offset_dev, a->rows_dev, stream, tmap_with_fwd ? &tmj : nullptr);
        else if (draw_in_fwd)
            rc = mil_gate_scores_fwd_draw((const float*)a->x, a->Wv, a->bv, a->Wu, a->bu, a->w, a->b, a->scores, gates, a->R,
                                          a->L, MIL_GATE_D, a->xbits, xscale, a->mbits, a->B, a->seed,
                                          a->seed ^ 0x9E3779B97F4A7C15ull, a->offset, a->offset_dev, stream);
        else
            rc = mil_gate_scores_fwd((const float*)a->x, a->Wv, a->bv, a->Wu, a->bu, a->w, a->b, a->scores, gates, a->R, a->L,
                                     MIL_GATE_D, xbits, xscale, stream);
        if (rc != MIL_OK) return rc;
    }
    if ((st & MIL_STAGE_POOL) && !pool_fused) {
        if (a->x_bf16) {
            rc = use_h ? mil_attn_pool_partial_h_bf16((const uint16_t*)a->x, a->scores, a->tile_map, a->T, a->L, a->partials,
                                                      a->Wf, a->C, a->hrow, xbits, xscale, mbits, mscale, stream)
                       : mil_attn_pool_partial_bf16((const uint16_t*)a->x, a->scores, a->tile_map, a->T, a->L, a->partials,
                                                    xbits, xscale, stream);
        } else {
            rc = use_h ? mil_attn_pool_partial_h((const float*)a->x, a->scores, a->tile_map, a->T, a->L, a->partials, a->Wf,
                                                 a->C, a->hrow, xbits, xscale, mbits, mscale, stream)
                       : mil_attn_pool_partial((const float*)a->x, a->scores, a->tile_map, a->T, a->L, a->partials, xbits,
                                               xscale, stream);
        }
        if (rc != MIL_OK) return rc;
    }
    if (st & MIL_STAGE_TAIL) {
        rc = mil_pool_merge_head_ws(a->partials, a->bag_tile_off, a->T, a->B, a->L, a->Wf, a->bf, a->C, a->y, a->loss_scale, a->M,
                                    a->lse, a->logits, a->prob, grads ? a->loss_bag : nullptr, grads ? a->dz : nullptr,
                                    grads ? a->dM : nullptr, grads ? a->cdot : nullptr, use_h ? a->tile_map : nullptr,
                                    use_h ? a->scores : nullptr, use_h ? a->hrow : nullptr, use_h ? a->ds : nullptr, mbits, mscale,
                                    train ? a->Mdrop : nullptr, a->loss_kind, a->tail_ws, stream);
        if (rc != MIL_OK) return rc;
        if (grads && !use_h) {          // no head projections: the score gradient takes a second pass over x
            rc = a->x_bf16 ? mil_attn_pool_bwd_bf16((const uint16_t*)a->x, a->scores, a->lse, a->dM, a->cdot, a->tile_map, a->T,
                                                    a->L, a->ds, xbits, xscale, stream)
                           : mil_attn_pool_bwd((const float*)a->x, a->scores, a->lse, a->dM, a->cdot, a->tile_map, a->T, a->L,
                                               a->ds, nullptr, xbits, xscale, stream);
            if (rc != MIL_OK) return rc;
        }
    }
    if (!grads) return MIL_OK;
    if (!a->dWv || !a->dbv || !a->dWu || !a->dbu || !a->dw || !a->db || !a->dWf || !a->dbf || !a->loss_out || !a->dw_ws)
        return (st & (MIL_STAGE_GATE_BWD | MIL_STAGE_REDUCE)) ? MIL_EINVAL : MIL_OK;
    const float* Mhead = train ? a->Mdrop : a->M;               // the M the head saw (dWf = dz^T M)
    if (a->x_bf16) {
        // bf16 storage: one entry point per weight-gradient flavour (its launch pair), then the head's parameter gradients
        const bool tail16 = a->bf16_grad_mfma && (a->L % 256) == 0 && a->gates16 && (st & MIL_STAGE_GATE_BWD) &&
                            (st & MIL_STAGE_REDUCE);
        if (tail16) {
            // weight gradient + ONE fold launch that also forms the head gradients / loss and, when the optimizer stage runs
            // in this call, applies Adam and rewrites the bf16 shadows (no k_head_bwd_params / k_adam / k_cast_bf16 launches)
            adam_in_reduce = (st & MIL_STAGE_ADAM) && a->param_flat && a->grad_flat && a->exp_avg && a->exp_avg_sq &&
                             (a->adam_step_dev || a->adam_step >= 1);
            rc = gate_bwd_params_bf16_tail((const uint16_t*)a->x, a->gates16, a->ds, a->w, a->R, a->L, a->dw_ws,
                                           (size_t)a->dw_ws_floats, a->dWv, a->dbv, a->dWu, a->dbu, a->dw, a->db, a->accumulate,
                                           xbits, xscale, a->dz, Mhead, a->dWf, a->dbf, a->B, a->C, a->loss_bag, a->loss_out,
                                           adam_in_reduce ? a->param_flat : nullptr, a->grad_flat, (size_t)a->n_param, a->exp_avg,
                                           a->exp_avg_sq, a->adam_step, a->adam_step_dev, a->lr, a->lr_dev, a->beta1, a->beta2,
                                           a->eps, a->weight_decay, a->grad_scale, const_cast<uint16_t*>(a->Wv16),
                                           const_cast<uint16_t*>(a->Wu16), stream, a->done_dev);
            // (with done_dev the fold launch advances the counter itself: its last workgroup to read the step number does it)
            if (rc == MIL_OK && adam_in_reduce && a->adam_step_dev && !a->done_dev) rc = mil_counter_add(a->adam_step_dev, 1, stream);
            if (rc != MIL_OK) return rc;
        } else if (st & (MIL_STAGE_GATE_BWD | MIL_STAGE_REDUCE)) {
            if (a->bf16_grad_mfma && (a->L % 256) == 0 && a->gates16)
                rc = mil_gate_bwd_params_bf16((const uint16_t*)a->x, a->gates16, a->ds, a->w, a->R, a->L, MIL_GATE_D, a->dw_ws,
                                              (size_t)a->dw_ws_floats, a->dWv, a->dbv, a->dWu, a->dbu, a->dw, a->db,
                                              a->accumulate, xbits, xscale, stream);
            else
                rc = mil_gate_bwd_params_x16((const uint16_t*)a->x, a->gates, a->ds, a->w, a->R, a->L, MIL_GATE_D, a->dw_ws,
                                             (size_t)a->dw_ws_floats, a->dWv, a->dbv, a->dWu, a->dbu, a->dw, a->db,
                                             a->accumulate, xbits, xscale, stream);
            if (rc != MIL_OK) return rc;
        }
        if ((st & MIL_STAGE_REDUCE) && !tail16) {
            rc = mil_head_bwd_params_acc(a->dz, Mhead, a->dWf, a->dbf, a->B, a->L, a->C, a->loss_bag, a->loss_out, a->accumulate,
                                         stream);
            if (rc != MIL_OK) return rc;
        }
    } else {
        if (st & MIL_STAGE_GATE_BWD) {
            rc = mil_gate_bwd_partials_rows((const float*)a->x, a->gates, a->ds, a->w, a->R, a->L, MIL_GATE_D, a->dw_ws,
                                            (size_t)a->dw_ws_floats, xbits, a->bag_len_dev ? a->rows_dev : nullptr, stream);
            if (rc != MIL_OK) return rc;
        }
        if (st & MIL_STAGE_REDUCE) {
            // reduce and Adam in ONE launch when both stages run in this call with a host-side step number (world size 1,
            // no all-reduce in between): the threads that store the final gradients update their parameters on the spot
            adam_in_reduce = (st & MIL_STAGE_ADAM) && a->param_flat && a->grad_flat && a->exp_avg && a->exp_avg_sq &&
                             (a->adam_step_dev || a->adam_step >= 1);
            if (adam_in_reduce) {
                rc = gate_bwd_reduce_head_adam_impl(a->dw_ws, a->R, a->L, a->dWv, a->dbv, a->dWu, a->dbu, a->dw, a->db,
                                                    a->accumulate, xscale, a->dz, Mhead, a->dWf, a->dbf, a->B, a->C, a->loss_bag,
                                                    a->loss_out, a->param_flat, a->grad_flat, (size_t)a->n_param, a->exp_avg,
                                                    a->exp_avg_sq, a->adam_step, a->adam_step_dev, a->lr, a->lr_dev, a->beta1,
                                                    a->beta2, a->eps, a->weight_decay, a->grad_scale, stream, a->done_dev);
                // the device counter moves on after the update, as mil_adam_step_counted does - by the fold launch itself when
                // the caller gave it a sign-off word (done_dev), by a one-thread launch otherwise
                if (rc == MIL_OK && a->adam_step_dev && !a->done_dev) rc = mil_counter_add(a->adam_step_dev, 1, stream);
            }
            else
                rc = mil_gate_bwd_reduce_head(a->dw_ws, a->R, a->L, a->dWv, a->dbv, a->dWu, a->dbu, a->dw, a->db, a->accumulate,
                                              xscale, a->dz, Mhead, a->dWf, a->dbf, a->B, a->C, a->loss_bag, a->loss_out, stream);
            if (rc != MIL_OK) return rc;
        }
    }
    if ((st & MIL_STAGE_ADAM) && !adam_in_reduce) {
        if (!a->param_flat || !a->grad_flat || !a->exp_avg || !a->exp_avg_sq) return MIL_EINVAL;
        if (a->adam_step_dev && a->lr_dev)
            rc = mil_adam_step_dev(a->param_flat, a->grad_flat, a->exp_avg, a->exp_avg_sq, (size_t)a->n_param,
                                   a->adam_step_dev, a->lr_dev, a->beta1, a->beta2, a->eps, a->weight_decay, a->grad_scale, 1,
                                   stream);
        else if (a->adam_step_dev)
            rc = mil_adam_step_counted(a->param_flat, a->grad_flat, a->exp_avg, a->exp_avg_sq, (size_t)a->n_param,
                                       a->adam_step_dev, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, a->grad_scale,
                                       stream);
        else
            rc = mil_adam_step(a->param_flat, a->grad_flat, a->exp_avg, a->exp_avg_sq, (size_t)a->n_param, a->adam_step, a->lr,
                               a->beta1, a->beta2, a->eps, a->weight_decay, a->grad_scale, stream);
        if (rc != MIL_OK) return rc;
        if (a->x_bf16) {
            // the bf16 shadows of the gate weights follow their fp32 masters here, once per update, instead of being
            // re-cast in front of every forward (two launches per pass in round 1)
            rc = mil_cast_bf16(a->Wv, const_cast<uint16_t*>(a->Wv16), (size_t)MIL_GATE_D * a->L, stream);
            if (rc != MIL_OK) return rc;
            rc = mil_cast_bf16(a->Wu, const_cast<uint16_t*>(a->Wu16), (size_t)MIL_GATE_D * a->L, stream);
            if (rc != MIL_OK) return rc;
        }
    }
    return MIL_OK;
}

extern "C" int mil_image_only_step_time(const mil_image_only_step* a, uint32_t stages, int warm, int iters, float* ms_out,
                                        void* stream) {
    if (!a || !ms_out || iters <= 0 || warm < 0) return MIL_EINVAL;
    mil_image_only_step s = *a;
    s.stages = stages;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return MIL_EINVAL;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return MIL_EINVAL; }
    int rc = MIL_OK;
    for (int i = 0; i < warm && rc == MIL_OK; ++i) rc = mil_image_only_step_run(&s, stream);
    if (rc == MIL_OK) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters && rc == MIL_OK; ++i) rc = mil_image_only_step_run(&s, stream);
        (void)hipEventRecord(e1, st);
        const hipError_t e = hipEventSynchronize(e1);
        if (rc == MIL_OK && e != hipSuccess) rc = (int)e;
        float ms = 0.f;
        if (rc == MIL_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *ms_out = ms / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// In-step timing: the WHOLE step `iters` times, a HIP event recorded between its launch groups, so that every group's
// duration is measured where it runs - after its predecessor, with the caches in the state the step leaves them in -
// and not as a stand-alone repetition of one kernel.  groups[i] is the stage mask of group i (run in order, each as
// one mil_image_only_step_run call; MIL_STAGE_POOL_FUSED and friends ride along as given); ms_out[i] = average
// duration of group i, ms_out[ngroups] = average of first-event -> last-event (the step with its event gaps).
// nrot > 0: iteration i runs on batch (xs[i % nrot], ys[i % nrot]) - the same rotation as the caller's timed loop, so the
// groups are timed in the cache regime the headline step runs in (one resident batch stays in the Infinity Cache).
extern "C" int mil_image_only_step_profile_rot(const mil_image_only_step* a, const void* const* xs, const float* const* ys,
                                               int nrot, const uint32_t* groups, int ngroups, int warm, int iters,
                                               float* ms_out, void* stream) {
    if (!a || !groups || !ms_out || ngroups <= 0 || ngroups > 16 || iters <= 0 || iters > 256 || warm < 0) return MIL_EINVAL;
    if (nrot < 0 || nrot > 64 || (nrot > 0 && (!xs || !ys))) return MIL_EINVAL;
    for (int i = 0; i < nrot; ++i)
        if (!xs[i] || !ys[i]) return MIL_EINVAL;
    mil_image_only_step s = *a;
    hipStream_t st = (hipStream_t)stream;
    const int per = ngroups + 1, nev = per * iters;
    hipEvent_t* ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * (size_t)nev);
    if (!ev) return MIL_EINVAL;
    int made = 0, rc = MIL_OK, turn = 0;
    for (; made < nev; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) { rc = MIL_EINVAL; break; }
    for (int i = 0; i < warm && rc == MIL_OK; ++i, ++turn) {
        if (nrot > 0) { s.x = xs[turn % nrot]; s.y = ys[turn % nrot]; }
        for (int g = 0; g < ngroups && rc == MIL_OK; ++g) { s.stages = groups[g]; rc = mil_image_only_step_run(&s, stream); }
    }
    for (int i = 0; i < iters && rc == MIL_OK; ++i, ++turn) {
        if (nrot > 0) { s.x = xs[turn % nrot]; s.y = ys[turn % nrot]; }
        (void)hipEventRecord(ev[i * per], st);
        for (int g = 0; g < ngroups && rc == MIL_OK; ++g) {
            s.stages = groups[g];
            rc = mil_image_only_step_run(&s, stream);
            (void)hipEventRecord(ev[i * per + g + 1], st);
        }
    }
    if (rc == MIL_OK && hipStreamSynchronize(st) != hipSuccess) rc = MIL_EINVAL;
    if (rc == MIL_OK) {
        for (int g = 0; g <= ngroups; ++g) ms_out[g] = 0.f;
        for (int i = 0; i < iters; ++i) {
            float ms = 0.f;
            for (int g = 0; g < ngroups; ++g)
                if (hipEventElapsedTime(&ms, ev[i * per + g], ev[i * per + g + 1]) == hipSuccess) ms_out[g] += ms;
            if (hipEventElapsedTime(&ms, ev[i * per], ev[i * per + ngroups]) == hipSuccess) ms_out[ngroups] += ms;
        }
        for (int g = 0; g <= ngroups; ++g) ms_out[g] /= (float)iters;
    }
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    free(ev);
    return rc;
}

extern "C" int mil_image_only_step_profile(const mil_image_only_step* a, const uint32_t* groups, int ngroups, int warm,
                                           int iters, float* ms_out, void* stream) {
    return mil_image_only_step_profile_rot(a, nullptr, nullptr, 0, groups, ngroups, warm, iters, ms_out, stream);
}
