// accel_score.hip — the hot path behind the C ABI: scoring of a result window (dense logistic / dense SVR / list route), the print-exact re-score,
// replay + condense, collapse, record text, survivor re-scores of mixed designs, downloads and device views.
#include "accel_internal.h"

extern "C" {

int mipgen_pb_check(mipgen_accel* h)
{
    if (!h->pb_over || *h->pb_over == 0u) return MIPGEN_OK;
    const unsigned int v = *h->pb_over;
    *h->pb_over = 0u;
    return fail(MIPGEN_E_STATE, "print-exact re-score: %u scores on a 6-digit rounding boundary did not fit the re-score list "
                                "(mipgen_accel_set_print_exact(h, 0) waives the guarantee)", v);
}

// SVR scores within the dense / list kernels' error of a midpoint between two 6-significant-digit numbers (what the front end prints,
// mipgen.cpp:774) are re-scored by k_candidates in the reference's own operation order and overwritten, so that the printed digit is the
// reference's (its error against the reference's double is ~1e-16 relative: the libm exponential).  No host round trip: the list length
// stays on the device, the re-scoring grid is the list's capacity.  list = nullptr: the dense results of regions [r0, r1).
// Logistic scores (method = MIPGEN_SCORE_LOGISTIC; round 5) take the same route: the dense kernel regroups the 69 terms by window (error ~1e-14 on the
// exponent), the re-score evaluates them in the reference's order with every operation rounded on its own (logistic_exponent_exact).
static int fix_print_boundaries(mipgen_accel* h, int r0, int r1, const mipgen_candidate* list, double* scores, const uint64_t* records, int64_t n, int method = MIPGEN_SCORE_SVR)
{
    if (!h->print_exact || n <= 0 || (method == MIPGEN_SCORE_SVR && h->n_sv <= 0)) return MIPGEN_OK;
    // (the re-scoring grid is the list's capacity: logistic scores sit within 1e-11 of a midpoint with probability <= 2e-5 - a short list keeps the
    // launch cheap beside the 0.1-ms kernels of small logistic batches)
    const unsigned int cap = method == MIPGEN_SCORE_SVR ? (unsigned int)std::min<int64_t>(n / 1024 + 4096, (int64_t)1 << 24)
                                                        : (unsigned int)std::min<int64_t>(n / 8192 + 256, (int64_t)1 << 22);
    if (h->pb_cands.reserve(cap) || h->pb_idx.reserve(cap) || h->pb_scores.reserve(cap) || h->pb_count.reserve(1)) return MIPGEN_E_NOMEM;
    const double tol_rel = method == MIPGEN_SCORE_SVR ? 1e-10 : 1e-11, tol_abs = method == MIPGEN_SCORE_SVR ? 1e-13 * std::max(1.0, h->sum_abs_coef) : 1e-300;
    HIP_TRY(hipMemsetAsync(h->pb_count.p, 0, sizeof(unsigned int), h->stream));
    if (list) HIP_TRY(mipgen_launch_print_boundary_scan_list(h->stream, list, scores, records, (int)n, tol_rel, tol_abs, h->pb_cands.p, h->pb_idx.p, h->pb_count.p, cap));
    else HIP_TRY(mipgen_launch_print_boundary_scan(h->stream, h->dp, h->regions.p, r0, r1, scores, records, n, tol_rel, tol_abs, h->pb_cands.p, h->pb_idx.p, h->pb_count.p, cap, h->n_cu));
    HIP_TRY(mipgen_launch_candidates(h->stream, (int)cap, h->dp, h->regions.p, h->pb_cands.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->model.p, h->n_sv,
                                     h->gamma, h->rho, method, h->pb_scores.p, nullptr, nullptr, nullptr, 1, h->pb_count.p));
    HIP_TRY(mipgen_launch_scatter_scores(h->stream, h->pb_scores.p, h->pb_idx.p, (int)cap, h->pb_count.p, scores, h->pb_over));
    return MIPGEN_OK;
}

// the silent path (mipgen_accel_score_condense_all): only the condensed survivors of the window are ever printed, so only they are tested
// and re-scored (2 per scan position instead of the whole dense grid); the value goes into the survivor's score field
static int fix_print_boundaries_survivors(mipgen_accel* h, int w, int method = MIPGEN_SCORE_SVR)
{
    const Window& W = h->windows[(size_t)w];
    const int64_t n = 2 * W.n_pos;
    if (!h->print_exact || n <= 0 || (method == MIPGEN_SCORE_SVR && h->n_sv <= 0)) return MIPGEN_OK;
    const unsigned int cap = method == MIPGEN_SCORE_SVR ? (unsigned int)std::min<int64_t>(n / 256 + 1024, (int64_t)1 << 22)
                                                        : (unsigned int)std::min<int64_t>(n / 4096 + 64, (int64_t)1 << 20);
    if (h->pb_cands.reserve(cap) || h->pb_idx.reserve(cap) || h->pb_scores.reserve(cap) || h->pb_count.reserve(1)) return MIPGEN_E_NOMEM;
    const double tol_rel = method == MIPGEN_SCORE_SVR ? 1e-10 : 1e-11, tol_abs = method == MIPGEN_SCORE_SVR ? 1e-13 * std::max(1.0, h->sum_abs_coef) : 1e-300;
    mipgen_survivor* surv = h->survivors.p + 2 * W.pos0;
    HIP_TRY(hipMemsetAsync(h->pb_count.p, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(mipgen_launch_print_boundary_scan_surv(h->stream, h->dp, h->regions.p, W.r0, W.r1, surv, n, W.cand0, tol_rel, tol_abs, h->pb_cands.p, h->pb_idx.p, h->pb_count.p, cap));
    HIP_TRY(mipgen_launch_candidates(h->stream, (int)cap, h->dp, h->regions.p, h->pb_cands.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->model.p, h->n_sv,
                                     h->gamma, h->rho, method, h->pb_scores.p, nullptr, nullptr, nullptr, 1, h->pb_count.p));
    HIP_TRY(mipgen_launch_scatter_surv_scores(h->stream, h->pb_scores.p, h->pb_idx.p, (int)cap, h->pb_count.p, surv, h->pb_over));
    return MIPGEN_OK;
}

// The dense grid of a window through the LIST scorer: the route of SVR parameter sets outside the tiled kernel's limits (scan size < 3,
// > 240 arm pairs, a tile beyond 160 KiB of LDS; h->svr_geometry_error / svr_batch_error say which) - the reference completes every
// -arm_lengths / -capture_increment / range it is given (mipgen.cpp:222-261, 427-444), so does this library, at the list scorer's rate
// (k_features_batch builds all 192 features per candidate, k_svr_gemm contracts them with the model on the FP64 matrix cores).
static int svr_window_via_list(mipgen_accel* h, const Window& W)
{
    const int64_t CH = (int64_t)1 << 19;                                  // candidates per chunk: 0.8 GB of features
    const size_t n_max = (size_t)std::min<int64_t>(CH, std::max<int64_t>(W.n_cand, 1));
    if (h->cand_in.reserve(n_max) || h->cand_records.reserve(n_max) || h->cand_feats.reserve(n_max * MIPGEN_N_FEATURES)) return MIPGEN_E_NOMEM;
    for (int64_t c0 = 0; c0 < W.n_cand; c0 += CH) {
        const int n = (int)std::min<int64_t>(CH, W.n_cand - c0);
        HIP_TRY(mipgen_launch_dense_candidates(h->stream, h->dp, h->regions.p, W.r0, W.r1, c0, n, h->cand_in.p));
        HIP_TRY(mipgen_launch_features_batch(h->stream, n, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->cand_records.p,
                                             h->cand_feats.p));
        HIP_TRY(mipgen_launch_svr_gemm(h->stream, n, h->cand_feats.p, h->cand_records.p, h->model_t.p, h->sv_norm.p, h->sv_coef.p, h->sv_center.p, h->n_sv_pad,
                                       h->gamma, h->rho, h->scores.p + c0));
        HIP_TRY(mipgen_launch_dense_list_fix(h->stream, n, h->records.p + c0, h->rho, h->s_guard, h->scores.p + c0));
    }
    return MIPGEN_OK;
}

static int score_window_impl(mipgen_accel* h, int w, int32_t method, bool fix_dense = true)
{
    if (int rc = mipgen_ensure_tiles(h, method)) return rc;
    const Window& W = h->windows[(size_t)w];
    const bool svr_via_list = method == MIPGEN_SCORE_SVR && (!h->svr_geometry_error.empty() || !h->svr_batch_error.empty());
    if (svr_via_list && h->model_t.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    hipEvent_t* ev = nullptr;
    if (h->timing) { if (mipgen_ensure_events(h)) return MIPGEN_E_HIP; ev = &h->ev[4 * (size_t)w]; }
    if (ev) HIP_TRY(hipEventRecord(ev[0], h->stream));
    unsigned int sat_cap = 0;
    if (method == MIPGEN_SCORE_LOGISTIC) {
        // the logistic kernels list the candidates whose score turns on the last bit of the reference's pow (b^x in [2^53, 2^54): GC-rich
        // microsatellites get there, nothing else)
        sat_cap = (unsigned int)std::min<int64_t>(W.n_cand / 4096 + 4096, (int64_t)1 << 20);      // (0.014 % of a (CCG)n region's candidates sit in the tie binade; the re-score grid is the capacity)
        if (h->sat_idx.reserve(sat_cap) || h->sat_count.reserve(1) || h->sat_cands.reserve(sat_cap) || h->sat_scores.reserve(sat_cap)) return MIPGEN_E_NOMEM;
        HIP_TRY(hipMemsetAsync(h->sat_count.p, 0, sizeof(unsigned int), h->stream));
    }
    if (method == MIPGEN_SCORE_LOGISTIC && h->ld_lds > 0)
        // records + logistic scores from per-window tables (kernels_logistic_dense.hip)
        HIP_TRY(mipgen_launch_logistic_dense(h->stream, W.n_ld_tiles, h->ld_lds, h->dp, h->regions.p, h->ld_tiles.p + W.ld_tile0, h->bases.p, h->copy.p,
                                             h->unmap.p, h->dconsts, h->scores.p, h->records.p, h->sat_idx.p, h->sat_count.p, sat_cap));
    else
        HIP_TRY(mipgen_launch_records_logistic(h->stream, method == MIPGEN_SCORE_LOGISTIC, W.n_log_tiles, h->log_span_max, h->dp, h->regions.p,
                                               h->log_tiles.p + W.log_tile0, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->scores.p, h->records.p,
                                               sat_cap ? h->sat_idx.p : nullptr, sat_cap ? h->sat_count.p : nullptr, sat_cap));
    if (ev) HIP_TRY(hipEventRecord(ev[1], h->stream));
    if (sat_cap) {
        // the listed candidates in the reference's own term order with the correctly rounded power (k_candidates: logistic_exponent_exact, pow_base_cr) - before
        // the replay, the condense fold and the collapse compare anything.  (A list that overflows leaves its surplus with the dense kernel's value: there is no
        // guarantee to break - the reference's own double is a coin there - and no region short of a megabase of (CCG)n fills it.)
        HIP_TRY(mipgen_launch_index_candidates(h->stream, h->dp, h->regions.p, W.r0, W.r1, h->sat_idx.p, h->sat_count.p, sat_cap, h->sat_cands.p));
        HIP_TRY(mipgen_launch_candidates(h->stream, (int)sat_cap, h->dp, h->regions.p, h->sat_cands.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->model.p, h->n_sv,
                                         h->gamma, h->rho, MIPGEN_SCORE_LOGISTIC, h->sat_scores.p, nullptr, nullptr, nullptr, 1, h->sat_count.p));
        HIP_TRY(mipgen_launch_scatter_scores(h->stream, h->sat_scores.p, h->sat_idx.p, (int)sat_cap, h->sat_count.p, h->scores.p, nullptr));
    }
    // logistic scores that sit on a rounding midpoint of the six printed digits: re-scored in the reference's term order (dense windows of non-silent
    // designs; the silent path tests its survivors only)
    if (method == MIPGEN_SCORE_LOGISTIC && fix_dense) { if (int rc = fix_print_boundaries(h, W.r0, W.r1, nullptr, h->scores.p, h->records.p, W.n_cand, MIPGEN_SCORE_LOGISTIC)) return rc; }
    if (svr_via_list) {
        if (int rc = svr_window_via_list(h, W)) return rc;
        if (fix_dense) { if (int rc = fix_print_boundaries(h, W.r0, W.r1, nullptr, h->scores.p, h->records.p, W.n_cand)) return rc; }
    } else if (method == MIPGEN_SCORE_SVR) {
        const double gamma_l2e = h->gamma * 1.4426950408889634074;
        int split = h->sv_split > 0 ? h->sv_split : mipgen_pick_sv_split(W.n_svr_tiles, h->n_sv, h->n_cu);
        split = std::max(1, std::min(split, (h->n_sv + SVR_GROUP - 1) / SVR_GROUP));
        if (split > 1 && h->partials.reserve((size_t)(split - 1) * (size_t)std::max<int64_t>(W.n_cand, 1))) return MIPGEN_E_NOMEM;
        const int levels = (int)W.lvl_tile0.size() - 1;
        if (h->dyn_skip && h->svr_levels > 1 && levels > 1) {
            // The runs of capture sizes in order, largest first; between two runs the enumeration state of every scan position (mipgen.cpp:426-497)
            // is advanced over the run just scored, and the tiles of the next run whose positions have all stopped (:430) are left out
            // (kernels_skip.hip).  One launch per run (no split along the SV list: these are large batches), one small synchronisation per run
            // for the number of tiles that are left.
            int max_run = 0;
            for (int l = 0; l < levels; l++) max_run = std::max(max_run, W.lvl_tile0[(size_t)l + 1] - W.lvl_tile0[(size_t)l]);
            if (h->run_pbs.reserve((size_t)W.n_pos) || h->run_state.reserve((size_t)W.n_pos) || h->run_keep.reserve((size_t)max_run + 1) ||
                h->run_offs.reserve((size_t)max_run + 1) || h->svr_tiles_kept.reserve((size_t)max_run) || h->skip_count.reserve(1))
                return MIPGEN_E_NOMEM;
            HIP_TRY(hipMemsetAsync(h->run_pbs.p, 0, (size_t)W.n_pos * sizeof(double), h->stream));
            HIP_TRY(hipMemsetAsync(h->run_state.p, 0, (size_t)W.n_pos, h->stream));
            if (!h->skip_count_valid) { HIP_TRY(hipMemsetAsync(h->skip_count.p, 0, sizeof(unsigned long long), h->stream)); h->skip_count_valid = true; }
            // a score this close to the limit could land on the other side after the print-exact re-score: such a position is never marked as stopped
            const double margin = 1e-6 * std::max(1.0, fabs(h->hp.upper)) + 1e-12 * std::max(1.0, h->sum_abs_coef);
            for (int l = 0; l < levels; l++) {
                const SvrTile* tl = h->svr_tiles_lvl.p + W.lvl_tile0[(size_t)l];
                int nt = W.lvl_tile0[(size_t)l + 1] - W.lvl_tile0[(size_t)l];
                if (l > 0 && nt > 0) {
                    HIP_TRY(mipgen_launch_svr_run_state(h->stream, W.n_pos, h->dp, h->regions.p, h->pos_region.p + W.pos0, h->pos_local.p + W.pos0, h->run_bounds.p,
                                                        h->svr_levels, l - 1, margin, h->scores.p, h->records.p, h->run_pbs.p, h->run_state.p));
                    HIP_TRY(mipgen_launch_svr_tile_keep(h->stream, nt, tl, h->region_pos0.p, W.pos0, h->run_state.p, h->run_keep.p));
                    size_t temp_bytes = 0;
                    HIP_TRY(mipgen_scan_i64(h->stream, nullptr, &temp_bytes, h->run_keep.p, h->run_offs.p, (int64_t)nt + 1));
                    if (h->fmt_temp.reserve(temp_bytes + 16)) return MIPGEN_E_NOMEM;
                    HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->run_keep.p, h->run_offs.p, (int64_t)nt + 1));
                    HIP_TRY(mipgen_launch_svr_tile_compact(h->stream, nt, tl, h->run_keep.p, h->run_offs.p, h->svr_tiles_kept.p, h->dp, h->regions.p, h->scores.p,
                                                           h->skip_count.p));
                    int64_t kept = 0;
                    HIP_TRY(hipMemcpyAsync(&kept, h->run_offs.p + nt, sizeof kept, hipMemcpyDeviceToHost, h->stream));
                    HIP_TRY(hipStreamSynchronize(h->stream));
                    tl = h->svr_tiles_kept.p; nt = (int)kept;
                }
                const int nt_few = l == 0 ? W.lvl0_few : 0;
                HIP_TRY(mipgen_launch_svr_dense(h->stream, nt - nt_few, nt_few, h->svr_lds, h->dp, &h->geom, &h->geom_few, h->regions.p, tl, h->bases.p, h->copy.p,
                                                (const double*)h->dconsts, h->model.p, h->n_sv, gamma_l2e, h->rho, h->s_guard, h->records.p, h->scores.p, W.n_cand, 1,
                                                nullptr));
            }
        } else
        HIP_TRY(mipgen_launch_svr_dense(h->stream, W.n_svr_tiles - W.n_svr_few, W.n_svr_few, h->svr_lds, h->dp, &h->geom, &h->geom_few, h->regions.p,
                                        h->svr_tiles.p + W.svr_tile0, h->bases.p, h->copy.p, (const double*)h->dconsts /* log10_tab is the first member */,
                                        h->model.p, h->n_sv, gamma_l2e, h->rho, h->s_guard, h->records.p, h->scores.p, W.n_cand, split, h->partials.p));
        if (fix_dense) { if (int rc = fix_print_boundaries(h, W.r0, W.r1, nullptr, h->scores.p, h->records.p, W.n_cand)) return rc; }
    }
    if (ev) { HIP_TRY(hipEventRecord(ev[2], h->stream)); h->ev_used[(size_t)w] |= 1; }
    h->cur_window = w; h->scored = true; h->replayed = false;
    if ((size_t)w < h->win_state.size()) h->win_state[(size_t)w] = 0;          // its survivors / collapse results are of older scores now
    return MIPGEN_OK;
}

static int replay_window_impl(mipgen_accel* h, bool want_mask)
{
    const int w = h->cur_window;
    const Window& W = h->windows[(size_t)w];
    // asynchronous on the handle's stream; buffers and the position map were laid out at upload
    if (W.r1 > W.r0) HIP_TRY(hipMemsetAsync(h->emitted_per_region.p + W.r0, 0, (size_t)(W.r1 - W.r0) * sizeof(unsigned long long), h->stream));
    // the per-candidate emitted flags are only materialised for callers that fetch them (the all_mips file); the fold itself keeps
    // them as lane masks in LDS
    if (want_mask && W.n_cand) HIP_TRY(hipMemsetAsync(h->emitted.p, 0, (size_t)W.n_cand, h->stream));
    HIP_TRY(mipgen_launch_replay_condense(h->stream, h->n_regions, (int)W.n_pos, h->dp, h->hp.n_pairs, h->hp.n_sizes_all, h->regions.p, h->pos_region.p + W.pos0,
                                          h->pos_local.p + W.pos0, h->scores.p, h->records.p, h->copy.p, W.cand0, want_mask ? h->emitted.p : nullptr,
                                          h->survivors.p + 2 * W.pos0, h->emitted_per_region.p));
    h->mask_valid = want_mask;
    if (h->timing && h->ev.size() >= 4 * ((size_t)w + 1)) { HIP_TRY(hipEventRecord(h->ev[4 * (size_t)w + 3], h->stream)); h->ev_used[(size_t)w] |= 2; }
    h->replayed = true;
    if ((size_t)w < h->win_state.size()) h->win_state[(size_t)w] = 1;
    return MIPGEN_OK;
}

static int collapse_window_impl(mipgen_accel* h)
{
    const Window& W = h->windows[(size_t)h->cur_window];
    HIP_TRY(mipgen_launch_collapse(h->stream, W.n_col_tiles, h->col_tiles.p + W.col_tile0, h->dp, h->regions.p, h->region_pos0.p, h->region_base0.p,
                                   h->survivors.p, h->copy.p, W.cand0, h->collapsed.p, h->hp.max_capture - h->hp.min_sum));
    if ((size_t)h->cur_window < h->win_state.size()) h->win_state[(size_t)h->cur_window] |= 2;
    return MIPGEN_OK;
}

static int check_scoring_args(mipgen_accel* h, int32_t method)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (method != MIPGEN_SCORE_LOGISTIC && method != MIPGEN_SCORE_SVR) return fail(MIPGEN_E_INVALID, "method must be logistic or svr");
    if (h->windows.empty()) return fail(MIPGEN_E_STATE, "no resident region batch");
    if (method == MIPGEN_SCORE_SVR && h->n_sv <= 0 && h->model.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    return MIPGEN_OK;
}

int mipgen_accel_score_window(mipgen_accel* h, int32_t w, int32_t method)
{
    if (int rc = check_scoring_args(h, method)) return rc;
    if (w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window %d out of range (%zu windows)", w, h->windows.size());
    HIP_TRY(hipSetDevice(h->device));
    std::fill(h->ev_used.begin(), h->ev_used.end(), 0);
    return score_window_impl(h, w, method);
}

int mipgen_accel_score_resident(mipgen_accel* h, int32_t method)
{
    if (int rc = check_scoring_args(h, method)) return rc;
    if (h->windows.size() != 1)
        return fail(MIPGEN_E_STATE, "the resident batch spans %zu result windows: use mipgen_accel_score_window / mipgen_accel_score_condense_all", h->windows.size());
    return mipgen_accel_score_window(h, 0, method);
}

int mipgen_accel_score_condense_window(mipgen_accel* h, int32_t w, int32_t method)
{
    if (int rc = check_scoring_args(h, method)) return rc;
    if (w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window %d out of range (%zu windows)", w, h->windows.size());
    HIP_TRY(hipSetDevice(h->device));
    std::fill(h->ev_used.begin(), h->ev_used.end(), 0);
    if (int rc = score_window_impl(h, w, method, false)) return rc;
    if (int rc = replay_window_impl(h, false)) return rc;
    return fix_print_boundaries_survivors(h, w, method);
}

int mipgen_accel_score_condense_all(mipgen_accel* h, int32_t method)
{
    if (int rc = check_scoring_args(h, method)) return rc;
    HIP_TRY(hipSetDevice(h->device));
    std::fill(h->ev_used.begin(), h->ev_used.end(), 0);
    for (int w = 0; w < (int)h->windows.size(); w++) {
        if (int rc = score_window_impl(h, w, method, false)) return rc;
        if (int rc = replay_window_impl(h, false)) return rc;
        if (int rc = fix_print_boundaries_survivors(h, w, method)) return rc;
        if (int rc = collapse_window_impl(h)) return rc;
    }
    return MIPGEN_OK;
}

int mipgen_accel_set_dynamic_skip(mipgen_accel* h, int32_t on)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->dyn_skip = on != 0;
    return MIPGEN_OK;
}

int mipgen_accel_skip_state(mipgen_accel* h, uint8_t* state, double* previous_best, int64_t capacity)
{
    if (!h || capacity < 0) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (h->cur_window < 0 || !h->run_state.p) return fail(MIPGEN_E_STATE, "no window scored with the dynamic skip");
    const Window& W = h->windows[(size_t)h->cur_window];
    if (capacity < W.n_pos) return fail(MIPGEN_E_INVALID, "capacity %lld < %lld scan positions", (long long)capacity, (long long)W.n_pos);
    HIP_TRY(hipSetDevice(h->device));
    if (state) HIP_TRY(hipMemcpyAsync(state, h->run_state.p, (size_t)W.n_pos, hipMemcpyDeviceToHost, h->stream));
    if (previous_best) HIP_TRY(hipMemcpyAsync(previous_best, h->run_pbs.p, (size_t)W.n_pos * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_skipped_candidates(mipgen_accel* h, int64_t* n)
{
    if (!h || !n) return fail(MIPGEN_E_INVALID, "bad arguments");
    *n = 0;
    if (!h->skip_count_valid) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, h->skip_count.p, sizeof v, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemsetAsync(h->skip_count.p, 0, sizeof v, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *n = (int64_t)v;
    return MIPGEN_OK;
}

int mipgen_accel_result_device_ptrs(const mipgen_accel* h, void** scores_dev, void** records_dev)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (scores_dev) *scores_dev = h->scores.p;
    if (records_dev) *records_dev = h->records.p;
    return MIPGEN_OK;
}

int mipgen_accel_download_results(mipgen_accel* h, double* scores, uint64_t* records, int64_t first, int64_t count)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->scored || h->cur_window < 0) return fail(MIPGEN_E_STATE, "nothing scored yet");
    const Window& W = h->windows[(size_t)h->cur_window];
    if (first < W.cand0 || count < 0 || first + count > W.cand0 + W.n_cand)
        return fail(MIPGEN_E_INVALID, "range [%lld,+%lld) outside the scored window [%lld,+%lld)", (long long)first, (long long)count, (long long)W.cand0, (long long)W.n_cand);
    HIP_TRY(hipSetDevice(h->device));
    if (count == 0) { HIP_TRY(hipStreamSynchronize(h->stream)); return MIPGEN_OK; }
    const int64_t off = first - W.cand0;
    if (scores) HIP_TRY(hipMemcpyAsync(scores, h->scores.p + off, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (records) HIP_TRY(hipMemcpyAsync(records, h->records.p + off, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = mipgen_pb_check(h)) return rc_pb;
    return MIPGEN_OK;
}

int mipgen_accel_score_regions(mipgen_accel* h, const mipgen_region* regions, int32_t n, int32_t method, mipgen_grid* grids_out,
                               double* scores, uint64_t* records, int64_t capacity)
{
    int rc = mipgen_accel_upload_regions(h, regions, n, grids_out);
    if (rc) return rc;
    if (h->n_cand > capacity) return fail(MIPGEN_E_INVALID, "result capacity %lld < %lld candidates", (long long)capacity, (long long)h->n_cand);
    for (int w = 0; w < (int)h->windows.size(); w++) {
        const Window& W = h->windows[(size_t)w];
        rc = mipgen_accel_score_window(h, w, method);
        if (rc) return rc;
        rc = mipgen_accel_download_results(h, scores ? scores + W.cand0 : nullptr, records ? records + W.cand0 : nullptr, W.cand0, W.n_cand);
        if (rc) return rc;
    }
    return MIPGEN_OK;
}

int mipgen_accel_score_candidates(mipgen_accel* h, const mipgen_candidate* cands, int32_t n, int32_t method, double* scores,
                                  uint64_t* records, double* features, mipgen_candidate_ints* ints)
{
    if (!h || n < 0 || (n > 0 && !cands)) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (method != MIPGEN_SCORE_LOGISTIC && method != MIPGEN_SCORE_SVR) return fail(MIPGEN_E_INVALID, "method must be logistic or svr");
    if (h->regions.p == nullptr) return fail(MIPGEN_E_STATE, "no resident region batch");
    if (method == MIPGEN_SCORE_SVR && h->model.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    if (n == 0) return MIPGEN_OK;
    for (int i = 0; i < n; i++) if (cands[i].region < 0 || cands[i].region >= h->n_regions) return fail(MIPGEN_E_INVALID, "candidate %d: region %d not resident", i, cands[i].region);
    HIP_TRY(hipSetDevice(h->device));
    if (h->cand_in.reserve((size_t)n) || h->cand_scores.reserve((size_t)n) || h->cand_records.reserve((size_t)n) ||
        (features && h->cand_feats.reserve((size_t)n * MIPGEN_N_FEATURES)) || (ints && h->cand_ints.reserve((size_t)n)))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->cand_in.p, cands, (size_t)n * sizeof(mipgen_candidate), hipMemcpyHostToDevice, h->stream));
    // long SVR lists (a mixed design re-scores every condensed survivor): features + records by k_features_batch (a wavefront per candidate),
    // then all candidate x support-vector distances through the FP64 matrix cores (k_svr_gemm) instead of one model walk per candidate
    const bool batched = method == MIPGEN_SCORE_SVR && n >= 256 && scores;
    if (batched && h->cand_feats.reserve((size_t)n * MIPGEN_N_FEATURES)) return MIPGEN_E_NOMEM;
    hipEvent_t le[3] = {nullptr, nullptr, nullptr};
    const bool time_list = batched && !ints && h->timing;
    if (time_list) { for (hipEvent_t& e : le) HIP_TRY(hipEventCreate(&e)); HIP_TRY(hipEventRecord(le[0], h->stream)); }
    if (batched && !ints)
        // lists: one wavefront per candidate for the features + records, then the matrix-core scorer
        HIP_TRY(mipgen_launch_features_batch(h->stream, n, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->cand_records.p,
                                             h->cand_feats.p));
    else
        HIP_TRY(mipgen_launch_candidates(h->stream, n, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts,
                                         h->model.p, h->n_sv, h->gamma, h->rho, method, batched ? nullptr : h->cand_scores.p, h->cand_records.p,
                                         (features || batched) ? h->cand_feats.p : nullptr, ints ? h->cand_ints.p : nullptr, 0, nullptr));
    if (time_list) HIP_TRY(hipEventRecord(le[1], h->stream));
    if (batched) HIP_TRY(mipgen_launch_svr_gemm(h->stream, n, h->cand_feats.p, h->cand_records.p, h->model_t.p, h->sv_norm.p, h->sv_coef.p, h->sv_center.p, h->n_sv_pad,
                                            h->gamma, h->rho, h->cand_scores.p));
    if (time_list) HIP_TRY(hipEventRecord(le[2], h->stream));
    if (batched) { if (int rc = fix_print_boundaries(h, 0, 0, h->cand_in.p, h->cand_scores.p, h->cand_records.p, n)) return rc; }
    if (scores) HIP_TRY(hipMemcpyAsync(scores, h->cand_scores.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (records) HIP_TRY(hipMemcpyAsync(records, h->cand_records.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    if (features) HIP_TRY(hipMemcpyAsync(features, h->cand_feats.p, (size_t)n * MIPGEN_N_FEATURES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (ints) HIP_TRY(hipMemcpyAsync(ints, h->cand_ints.p, (size_t)n * sizeof(mipgen_candidate_ints), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = mipgen_pb_check(h)) return rc_pb;
    if (time_list) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, le[0], le[1]) == hipSuccess && hipEventElapsedTime(&b, le[1], le[2]) == hipSuccess) { h->list_feat_ms = a; h->list_svr_ms = b; }
        for (hipEvent_t e : le) (void)hipEventDestroy(e);
    }
    return MIPGEN_OK;
}

static void lrc_mers(LrcMers& M)
{
    static const char* mers[MIPGEN_N_LRC] = MIPGEN_FEATURE_MERS;
    auto code = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; };
    for (int i = 0; i < MIPGEN_N_LRC; i++) {
        const int k = (int)strlen(mers[i]);
        int cd = 0, rc = 0;
        for (int j = 0; j < k; j++) cd = cd * 4 + code(mers[i][j]);
        for (int j = k - 1; j >= 0; j--) rc = rc * 4 + (3 - code(mers[i][j]));
        M.k[i] = (int8_t)k; M.code[i] = (int8_t)cd; M.rc[i] = (int8_t)(rc == cd ? -1 : rc);
    }
}

int mipgen_accel_long_range_content_batch(mipgen_accel* h, int32_t n, const char* const* extended_seqs, const int32_t* lens,
                                          const int32_t* chrom_seq_starts, const int32_t* chrom_seq_stops, double* out)
{
    if (!h || n < 0 || (n > 0 && (!extended_seqs || !lens || !chrom_seq_starts || !chrom_seq_stops || !out))) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (n == 0) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    LrcMers M;
    lrc_mers(M);
    std::vector<int64_t> offs((size_t)n);
    std::vector<int32_t> denoms((size_t)n);
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        if (lens[i] < 0 || (lens[i] > 0 && !extended_seqs[i])) return fail(MIPGEN_E_INVALID, "long-range sequence %d is missing", i);
        offs[(size_t)i] = total; total += lens[i];
        denoms[(size_t)i] = chrom_seq_stops[i] - chrom_seq_starts[i] + 2001;          // Featurev5.cpp:49,53
    }
    std::vector<char> packed((size_t)std::max<int64_t>(total, 1));
    for (int i = 0; i < n; i++) if (lens[i]) memcpy(&packed[(size_t)offs[(size_t)i]], extended_seqs[i], (size_t)lens[i]);
    if (h->lrc_seq.reserve(packed.size()) || h->lrc_out.reserve((size_t)n * MIPGEN_N_LRC) || h->lrc_offs.reserve((size_t)n) ||
        h->lrc_lens.reserve((size_t)n) || h->lrc_denoms.reserve((size_t)n))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->lrc_seq.p, packed.data(), packed.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->lrc_offs.p, offs.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->lrc_lens.p, lens, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->lrc_denoms.p, denoms.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(mipgen_launch_long_range(h->stream, n, h->lrc_seq.p, h->lrc_offs.p, h->lrc_lens.p, h->lrc_denoms.p, &M, h->lrc_out.p));
    HIP_TRY(hipMemcpyAsync(out, h->lrc_out.p, (size_t)n * MIPGEN_N_LRC * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_long_range_content(mipgen_accel* h, const char* extended_seq, int32_t len, int32_t chrom_seq_start,
                                    int32_t chrom_seq_stop, double* out44)
{
    if (!h || !extended_seq || len < 0 || !out44) return fail(MIPGEN_E_INVALID, "bad arguments");
    return mipgen_accel_long_range_content_batch(h, 1, &extended_seq, &len, &chrom_seq_start, &chrom_seq_stop, out44);
}

int mipgen_accel_replay_condense(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->scored || h->cur_window < 0) return fail(MIPGEN_E_STATE, "replay requested before scoring");
    HIP_TRY(hipSetDevice(h->device));
    return replay_window_impl(h, true);
}

int mipgen_accel_download_replay(mipgen_accel* h, int64_t* emitted_per_region, mipgen_survivor* survivors, int64_t survivor_capacity,
                                 uint8_t* emitted_mask, int64_t mask_capacity)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->replayed || h->cur_window < 0) return fail(MIPGEN_E_STATE, "mipgen_accel_replay_condense has not run on these scores");
    const Window& W = h->windows[(size_t)h->cur_window];
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = mipgen_pb_check(h)) return rc_pb;
    if (emitted_per_region && W.r1 > W.r0)
        HIP_TRY(hipMemcpy(emitted_per_region, h->emitted_per_region.p + W.r0, (size_t)(W.r1 - W.r0) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (survivors) {
        if (survivor_capacity < 2 * W.n_pos) return fail(MIPGEN_E_INVALID, "survivor capacity too small");
        if (W.n_pos) HIP_TRY(hipMemcpy(survivors, h->survivors.p + 2 * W.pos0, (size_t)(2 * W.n_pos) * sizeof(mipgen_survivor), hipMemcpyDeviceToHost));
    }
    if (emitted_mask) {
        if (!h->mask_valid) return fail(MIPGEN_E_STATE, "the emitted mask is only kept by mipgen_accel_replay_condense, not by mipgen_accel_score_condense_all");
        if (mask_capacity < W.n_cand) return fail(MIPGEN_E_INVALID, "mask capacity too small");
        if (W.n_cand) HIP_TRY(hipMemcpy(emitted_mask, h->emitted.p, (size_t)W.n_cand, hipMemcpyDeviceToHost));
    }
    return MIPGEN_OK;
}

int mipgen_accel_collapse(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->replayed || h->cur_window < 0) return fail(MIPGEN_E_STATE, "collapse requested before replay + condense");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = collapse_window_impl(h)) return rc;
    return MIPGEN_OK;
}

int mipgen_accel_region_bases(const mipgen_accel* h, int32_t region, int64_t* first_entry, int32_t* n_bases)
{
    if (!h || region < 0 || region >= h->n_regions) return fail(MIPGEN_E_INVALID, "region %d out of range", region);
    if (first_entry) *first_entry = h->h_region_base0[(size_t)region];
    if (n_bases) *n_bases = (int32_t)((h->h_region_base0[(size_t)region + 1] - h->h_region_base0[(size_t)region]) / 2);
    return MIPGEN_OK;
}

int mipgen_accel_download_collapsed(mipgen_accel* h, int32_t window, int32_t* best_scan_index, int64_t capacity)
{
    if (!h || !best_scan_index) return fail(MIPGEN_E_INVALID, "bad arguments");
    // validity is per result window: every window asked for must have been collapsed since it was last scored
    for (size_t w = 0; w < h->win_state.size(); w++)
        if ((window < 0 || (size_t)window == w) && !(h->win_state[w] & 2))
            return fail(MIPGEN_E_STATE, "mipgen_accel_collapse / mipgen_accel_score_condense_all has not run on the current scores of window %zu", w);
    if (h->win_state.empty()) return fail(MIPGEN_E_STATE, "no resident region batch");
    int64_t first = 0, count = h->h_region_base0.empty() ? 0 : h->h_region_base0.back();
    if (window >= 0) {
        if (window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window %d out of range", window);
        first = h->windows[(size_t)window].base0; count = h->windows[(size_t)window].n_base_entries;
    }
    if (capacity < count) return fail(MIPGEN_E_INVALID, "collapsed capacity %lld < %lld entries", (long long)capacity, (long long)count);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (count) HIP_TRY(hipMemcpy(best_scan_index, h->collapsed.p + first, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MIPGEN_OK;
}

// ---- section 8f-4: all_mips records formatted on the device -----------------------------------------------------------------------
int mipgen_accel_format_all_mips(mipgen_accel* h, const mipgen_record_names* names, const char* middle, int64_t first_index, int64_t* n_records, int64_t* n_bytes)
{
    if (!h || !middle || !n_records || !n_bytes) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (!h->replayed || !h->mask_valid || h->cur_window < 0) return fail(MIPGEN_E_STATE, "format_all_mips needs mipgen_accel_replay_condense on the scored window first");
    const Window& W = h->windows[(size_t)h->cur_window];
    const int nr = W.r1 - W.r0;
    if (nr > 0 && !names) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (strlen(middle) >= sizeof(((FmtConst*)0)->middle)) return fail(MIPGEN_E_INVALID, "middle sequence too long");
    HIP_TRY(hipSetDevice(h->device));
    FmtConst FC;
    memset(&FC, 0, sizeof FC);
    FC.middle_len = (int32_t)strlen(middle); memcpy(FC.middle, middle, (size_t)FC.middle_len);
    FC.n_regions = nr; FC.first_index = first_index;
    std::vector<FmtRegion> fr((size_t)std::max(nr, 1));
    std::string pool;
    int64_t n_rb = 0;
    for (int i = 0; i < nr; i++) {
        const DevRegion& d = h->hregions[(size_t)(W.r0 + i)];
        FmtRegion& f = fr[(size_t)i];
        const char* chr = names[i].chr ? names[i].chr : ""; const char* label = names[i].label ? names[i].label : "";
        f.chr_off = (int32_t)pool.size(); f.chr_len = (int32_t)strlen(chr); pool += chr;
        f.label_off = (int32_t)pool.size(); f.label_len = (int32_t)strlen(label); pool += label;
        f.feature_start = names[i].feature_start; f.feature_stop = names[i].feature_stop;
        f.rb0 = n_rb;
        n_rb += (int64_t)d.n_pos * d.n_sizes;
    }
    *n_records = 0; *n_bytes = 0; h->fmt_bytes = 0;
    if (n_rb == 0) return MIPGEN_OK;
    if (n_rb + 1 > INT32_MAX) return fail(MIPGEN_E_INVALID, "window too large for record formatting (%lld row blocks)", (long long)n_rb);
    if (h->fmt_regions.reserve(fr.size()) || h->fmt_pool.reserve(std::max<size_t>(pool.size(), 1)) || h->fmt_a.reserve((size_t)n_rb + 1) || h->fmt_b.reserve((size_t)n_rb + 1) ||
        h->fmt_c.reserve((size_t)n_rb + 1) || h->fmt_d.reserve((size_t)n_rb + 1))
        return MIPGEN_E_NOMEM;
    size_t temp_bytes = 0;
    HIP_TRY(mipgen_scan_i64(h->stream, nullptr, &temp_bytes, h->fmt_a.p, h->fmt_b.p, n_rb + 1));
    if (h->fmt_temp.reserve(temp_bytes + 16)) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->fmt_regions.p, fr.data(), fr.size() * sizeof(FmtRegion), hipMemcpyHostToDevice, h->stream));
    if (!pool.empty()) HIP_TRY(hipMemcpyAsync(h->fmt_pool.p, pool.data(), pool.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->fmt_a.p + n_rb, 0, sizeof(int64_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->fmt_c.p + n_rb, 0, sizeof(int64_t), h->stream));
    // records per row block -> ranks; bytes per row block -> offsets; then the bytes
    HIP_TRY(mipgen_launch_fmt_count(h->stream, n_rb, W.r0, &FC, h->fmt_regions.p, h->dp, h->regions.p, h->emitted.p, h->fmt_a.p));
    HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->fmt_a.p, h->fmt_b.p, n_rb + 1));
    HIP_TRY(mipgen_launch_fmt_records(h->stream, 0, n_rb, W.r0, &FC, h->fmt_regions.p, h->fmt_pool.p, h->dp, h->regions.p, h->letters.p, h->copy.p, h->scores.p,
                                      h->records.p, h->emitted.p, h->fmt_b.p, nullptr, h->fmt_c.p, nullptr));
    HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->fmt_c.p, h->fmt_d.p, n_rb + 1));
    int64_t totals[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&totals[0], h->fmt_b.p + n_rb, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&totals[1], h->fmt_d.p + n_rb, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = mipgen_pb_check(h)) return rc_pb;
    if (h->fmt_text.reserve((size_t)std::max<int64_t>(totals[1], 1))) return MIPGEN_E_NOMEM;
    HIP_TRY(mipgen_launch_fmt_records(h->stream, 1, n_rb, W.r0, &FC, h->fmt_regions.p, h->fmt_pool.p, h->dp, h->regions.p, h->letters.p, h->copy.p, h->scores.p,
                                      h->records.p, h->emitted.p, h->fmt_b.p, h->fmt_d.p, nullptr, h->fmt_text.p));
    *n_records = totals[0]; *n_bytes = totals[1]; h->fmt_bytes = totals[1];
    return MIPGEN_OK;
}

int mipgen_accel_download_text(mipgen_accel* h, char* dst, int64_t capacity)
{
    if (!h || (!dst && capacity > 0)) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (h->fmt_bytes < 0) return fail(MIPGEN_E_STATE, "mipgen_accel_format_all_mips has not run on this window");
    if (capacity < h->fmt_bytes) return fail(MIPGEN_E_INVALID, "text capacity %lld < %lld bytes", (long long)capacity, (long long)h->fmt_bytes);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->fmt_bytes) HIP_TRY(hipMemcpy(dst, h->fmt_text.p, (size_t)h->fmt_bytes, hipMemcpyDeviceToHost));
    return MIPGEN_OK;
}

int mipgen_accel_download_survivors(mipgen_accel* h, int64_t* emitted_per_region, mipgen_survivor* survivors, int64_t survivor_capacity)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    // batch-wide survivors: every result window must have been replayed + condensed since it was last scored
    if (h->win_state.empty()) return fail(MIPGEN_E_STATE, "mipgen_accel_score_condense_all has not run on this batch");
    for (size_t w = 0; w < h->win_state.size(); w++)
        if (!(h->win_state[w] & 1)) return fail(MIPGEN_E_STATE, "mipgen_accel_score_condense_all has not run on this batch (window %zu holds no current survivors)", w);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = mipgen_pb_check(h)) return rc_pb;
    if (emitted_per_region && h->n_regions)
        HIP_TRY(hipMemcpy(emitted_per_region, h->emitted_per_region.p, (size_t)h->n_regions * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (survivors) {
        if (survivor_capacity < 2 * h->total_pos) return fail(MIPGEN_E_INVALID, "survivor capacity too small");
        if (h->total_pos) HIP_TRY(hipMemcpy(survivors, h->survivors.p, (size_t)(2 * h->total_pos) * sizeof(mipgen_survivor), hipMemcpyDeviceToHost));
    }
    return MIPGEN_OK;
}

int mipgen_accel_survivors_device_ptr(const mipgen_accel* h, void** survivors_dev, int64_t* n_survivors)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (survivors_dev) *survivors_dev = h->survivors.p;
    if (n_survivors) *n_survivors = 2 * h->total_pos;
    return MIPGEN_OK;
}

// ---- mixed designs: every condensed survivor of the window scored last through the SVR, on the device ------------------------------------------
// The reference re-scores the MIPs its pick stage tests, one at a time (mipgen.cpp:1523-1527, 1533-1537, 1546-1550, 1873-1877); here all survivors
// of the window go through the list scorer in ONE call - candidate list built on the device from the survivor array (slot order), features +
// matrix-core SVR (short lists: the literal per-candidate kernel), print-exact re-score - and the values stay in HBM beside the survivors
// (slot for slot; NaN where a slot holds no survivor).  Same kernels, same values as mipgen_accel_score_candidates on the same list.
int mipgen_accel_rescore_survivors(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->replayed || h->cur_window < 0) return fail(MIPGEN_E_STATE, "mipgen_accel_replay_condense has not run on these scores");
    if (h->model.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    const int w = h->cur_window;
    const Window& W = h->windows[(size_t)w];
    const int64_t n = 2 * W.n_pos;
    HIP_TRY(hipSetDevice(h->device));
    if (h->surv_svr.reserve((size_t)std::max<int64_t>(2 * h->total_pos, 1))) return MIPGEN_E_NOMEM;
    if (n > 0) {
        if (h->rs_keep.reserve((size_t)n + 1) || h->rs_offs.reserve((size_t)n + 1) || h->rs_idx.reserve((size_t)n) || h->cand_in.reserve((size_t)n)) return MIPGEN_E_NOMEM;
        const mipgen_survivor* surv = h->survivors.p + 2 * W.pos0;
        double* svr = h->surv_svr.p + 2 * W.pos0;
        HIP_TRY(mipgen_launch_surv_keep(h->stream, surv, n, h->rs_keep.p, svr));
        size_t temp_bytes = 0;
        HIP_TRY(mipgen_scan_i64(h->stream, nullptr, &temp_bytes, h->rs_keep.p, h->rs_offs.p, n + 1));
        if (h->fmt_temp.reserve(temp_bytes + 16)) return MIPGEN_E_NOMEM;
        HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->rs_keep.p, h->rs_offs.p, n + 1));
        HIP_TRY(mipgen_launch_surv_candidates(h->stream, h->dp, h->regions.p, W.r0, W.r1, surv, n, W.cand0, h->rs_offs.p, h->cand_in.p, h->rs_idx.p));
        int64_t m = 0;
        HIP_TRY(hipMemcpyAsync(&m, h->rs_offs.p + n, sizeof m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (m > 0) {
            if (m > INT32_MAX) return fail(MIPGEN_E_INVALID, "too many survivors in one window for the list scorer");
            if (h->cand_scores.reserve((size_t)m) || h->cand_records.reserve((size_t)m)) return MIPGEN_E_NOMEM;
            const bool batched = m >= 256;                                   // as mipgen_accel_score_candidates decides
            if (batched) {
                if (h->cand_feats.reserve((size_t)m * MIPGEN_N_FEATURES)) return MIPGEN_E_NOMEM;
                HIP_TRY(mipgen_launch_features_batch(h->stream, (int)m, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->cand_records.p,
                                                     h->cand_feats.p));
                HIP_TRY(mipgen_launch_svr_gemm(h->stream, (int)m, h->cand_feats.p, h->cand_records.p, h->model_t.p, h->sv_norm.p, h->sv_coef.p, h->sv_center.p, h->n_sv_pad,
                                               h->gamma, h->rho, h->cand_scores.p));
                if (int rc = fix_print_boundaries(h, 0, 0, h->cand_in.p, h->cand_scores.p, h->cand_records.p, m)) return rc;
            } else {
                HIP_TRY(mipgen_launch_candidates(h->stream, (int)m, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts,
                                                 h->model.p, h->n_sv, h->gamma, h->rho, MIPGEN_SCORE_SVR, h->cand_scores.p, h->cand_records.p, nullptr, nullptr, 0, nullptr));
            }
            HIP_TRY(mipgen_launch_scatter_f64(h->stream, h->cand_scores.p, h->rs_idx.p, m, svr));
        }
    }
    if ((size_t)w < h->win_state.size()) h->win_state[(size_t)w] |= 4;
    return MIPGEN_OK;
}

int mipgen_accel_download_survivor_scores(mipgen_accel* h, int32_t window, double* svr, int64_t capacity)
{
    if (!h || !svr || window < 0 || window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "bad arguments");
    if ((size_t)window >= h->win_state.size() || !(h->win_state[(size_t)window] & 4)) return fail(MIPGEN_E_STATE, "mipgen_accel_rescore_survivors has not run on the current survivors of window %d", window);
    const Window& W = h->windows[(size_t)window];
    if (capacity < 2 * W.n_pos) return fail(MIPGEN_E_INVALID, "capacity too small");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = mipgen_pb_check(h)) return rc_pb;
    if (W.n_pos) HIP_TRY(hipMemcpy(svr, h->surv_svr.p + 2 * W.pos0, (size_t)(2 * W.n_pos) * sizeof(double), hipMemcpyDeviceToHost));
    return MIPGEN_OK;
}

// ---- device-side views of a result window: for a caller that moves results between devices itself (the RCCL gather of the multi-GPU front end) ----
int mipgen_accel_window_views(mipgen_accel* h, int32_t window, mipgen_window_views* out)
{
    if (!h || !out || window < 0 || window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "bad arguments");
    const uint8_t st = (size_t)window < h->win_state.size() ? h->win_state[(size_t)window] : 0;
    if (!(st & 1)) return fail(MIPGEN_E_STATE, "window %d holds no current survivors (replay + condense first)", window);
    const Window& W = h->windows[(size_t)window];
    memset(out, 0, sizeof *out);
    out->emitted = h->emitted_per_region.p + W.r0; out->n_emitted = W.r1 - W.r0;
    out->survivors = h->survivors.p + 2 * W.pos0; out->n_survivors = 2 * W.n_pos;
    if (st & 2) { out->collapsed = h->collapsed.p + W.base0; out->n_collapsed = W.n_base_entries; }
    if (st & 4) out->survivor_svr = h->surv_svr.p + 2 * W.pos0;
    if (window == h->cur_window && h->fmt_bytes > 0) { out->text = h->fmt_text.p; out->n_text_bytes = h->fmt_bytes; }
    out->first_candidate = W.cand0;
    return MIPGEN_OK;
}

int mipgen_accel_synchronize(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return mipgen_pb_check(h);
}

}  // extern "C"
