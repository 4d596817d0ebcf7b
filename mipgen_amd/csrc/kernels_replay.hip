// kernels_replay.hip — replay of the reference's score-dependent enumeration control flow over the scored dense
// grid, fused with the condense_mips fold, on the device.
//
//   mipgen::tile_regions early exits   /root/reference/mipgen.cpp:426-437,440,494-497
//   mipgen::condense_mips              /root/reference/mipgen.cpp:1670-1746
//
// Both are sequential state machines, but only *within* one scan-start position: every position is independent
// (previous_best_score is reset at :426; the condense state is declared per position at :1677-1680).  One WAVEFRONT replays one
// position; its lanes are the arm pairs of one (capture size, strand) row, so every load is a coalesced row segment:
//
//   replay    per capture size the row's plus / minus scores and valid flags are loaded once; the arm-sum lists of the row are then
//             walked with wave-level bit operations (ballot / count-leading-zeros on 64-bit lane masks): the only sequential state
//             is previous_best_score between lists, `previous_*_score` / skip_ahead inside a list - a list costs a handful of scalar
//             instructions, not one step per pair.  The emitted flags of the position stay in LDS as lane masks.
//   condense  the fold over the newest-first candidate list changes its state only when a candidate is TAKEN, which is rare (a
//             record-breaking event).  All lanes evaluate the take rules against the current state at once; the first taker in fold
//             order is applied (v_readlane), the lanes after it are re-evaluated against the new state.  A row costs 1 + (number of
//             takes) wave iterations instead of one step per candidate, and the fold ends early exactly where the reference's does
//             (skip_ahead, :1736).
//
// This removes the device-to-host copy of the dense grid (16 B/candidate) for silent designs: 2 survivors per position remain.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <algorithm>
#include "common.h"
#include "device_utils.h"

#define REPLAY_WAVES 4

namespace {

// (int) of a double as x86-64 cvttsd2si gives it (the reference stores double scores into int locals, :496-497)
__device__ __forceinline__ int to_int_x86(double v)
{
    if (!(v == v) || v >= 2147483648.0 || v <= -2147483649.0) return INT_MIN;
    return (int)v;
}

__device__ __forceinline__ double readlane_d(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t x, int l)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(x >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
}
__device__ __forceinline__ int top_bit(uint64_t m) { return 63 - __builtin_clzll(m); }      // m != 0

}  // namespace

__global__ __launch_bounds__(REPLAY_WAVES * 64) void k_replay_condense(
    int total_pos, int n_chunks, int lds_pitch, const DevParams* __restrict__ P, const DevRegion* __restrict__ regions,
    const int32_t* __restrict__ pos_region, const int32_t* __restrict__ pos_local, const double* __restrict__ scores,
    const uint64_t* __restrict__ records, const int32_t* __restrict__ copy, int64_t cand_base, uint8_t* __restrict__ emitted,
    mipgen_survivor* __restrict__ survivors, unsigned long long* __restrict__ emitted_per_region)
{
    extern __shared__ uint64_t s_masks[];                                  // [wave][size index][chunk of 64 pairs]: emitted lanes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gp = blockIdx.x * REPLAY_WAVES + wave;
    if (gp >= total_pos) return;
    uint64_t* emask = s_masks + (size_t)wave * lds_pitch;
    const int ri = pos_region[gp], pi = pos_local[gp];
    const DevRegion& R = regions[ri];
    const int A = P->n_pairs, nK = R.n_sizes;
    const int64_t per_pos = (int64_t)nK * A * 2;
    const int64_t base = R.out_off + (int64_t)pi * per_pos;
    const double upper = P->upper, lower = P->lower;
    const bool heuristic = P->score_method == MIPGEN_SCORE_LOGISTIC && P->logistic_heuristic;
    const int min_sum = P->key_min_sum;
    const uint64_t lane_bit = 1ull << lane, below_me = lane_bit - 1;

    // ---- replay, mipgen.cpp:426-497 ---------------------------------------------------------------------
    unsigned long long n_emitted = 0;
    double pbs = 0.0;                                                        // previous_best_score (:426)
    for (int ki = 0; ki < nK; ki++) {
        if (lane < n_chunks) emask[ki * n_chunks + lane] = 0;
        if (pbs > upper) continue;                                           // :430
        int prev_plus = 0, prev_minus = 0;                                   // previous_plus_score / previous_minus_score (:435-436)
        bool skip_ahead = false, list_skipped = false;
        int list_end = 0;                                                    // end of the arm-sum list in progress
        for (int c = 0; c < n_chunks; c++) {
            const int a = c * 64 + lane;
            const bool in = a < A;
            const int64_t idx = base + ((int64_t)ki * 2) * A + a;
            const double plus = in ? scores[idx] : 0.0, minus = in ? scores[idx + A] : 0.0;
            const bool valid = in && (MIPGEN_REC_FLAGS(records[idx]) & MIPGEN_FLAG_VALID);       // :443-444
            const uint64_t vmask_all = __ballot(valid);
            uint64_t emit_chunk = 0;
            const int c_end = min(A, c * 64 + 64);
            int a0 = c * 64;
            while (a0 < c_end) {
                if (a0 == list_end) {                                        // a new arm-sum list starts here (:431-436)
                    list_end = P->group_end[a0];
                    list_skipped = pbs > upper && (P->arm_ext[a0] + P->arm_lig[a0]) != min_sum;   // :434
                    prev_plus = 0; prev_minus = 0; skip_ahead = false;
                }
                const int a1 = min(list_end, c_end);
                if (!list_skipped && !skip_ahead) {
                    const int lo = a0 - c * 64, hi = a1 - c * 64;            // lanes [lo, hi)
                    const uint64_t seg = (hi >= 64 ? ~0ull : ((1ull << hi) - 1)) & ~((1ull << lo) - 1);
                    const uint64_t vmask = vmask_all & seg;
                    if (vmask) {
                        uint64_t emit = vmask;
                        if (heuristic) {
                            // the pair before me in this list that was constructed: its truncated scores are what :494 compares against
                            const uint64_t below = vmask & below_me;
                            const int src = below ? top_bit(below) : lane;
                            const double pp = __shfl(plus, src, 64), pm = __shfl(minus, src, 64);
                            const int ip = below ? to_int_x86(pp) : prev_plus, im = below ? to_int_x86(pm) : prev_minus;
                            const bool cond = (vmask & lane_bit) && plus < (double)ip && minus < (double)im;      // :494
                            const uint64_t cmask = __ballot(cond);
                            if (cmask) { emit = vmask & ((2ull << __builtin_ctzll(cmask)) - 1); skip_ahead = true; }   // that pair is still constructed
                        }
                        const int last = top_bit(emit);
                        const double lp = readlane_d(plus, last), lm = readlane_d(minus, last);
                        pbs = (lm > lp) ? lm : lp;                           // :495
                        prev_minus = to_int_x86(lm); prev_plus = to_int_x86(lp);                                  // :496-497
                        emit_chunk |= emit;
                        n_emitted += 2ull * (unsigned)__builtin_popcountll(emit);
                    }
                }
                a0 = a1;
            }
            if (emit_chunk) {
                if (lane == 0) emask[ki * n_chunks + c] = emit_chunk;
                if (emitted && (emit_chunk & lane_bit)) { emitted[idx] = 1; emitted[idx + A] = 1; }
            }
        }
    }
    if (n_emitted && lane == 0) atomicAdd(&emitted_per_region[ri], n_emitted);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");              // the lane masks written above are read by every lane below
    __builtin_amdgcn_wave_barrier();

    // ---- condense, mipgen.cpp:1670-1746 -------------------------------------------------------------------
    const int target_copy = P->target_arm_copy;
    const int64_t max_product = P->max_arm_copy_product;
    const double thr = P->masked_arm_threshold;
    int chosen_copy = 0;
    double chosen_masked = 0.0;                                                      // per position, not per strand (:1677-1680)
    for (int s = 0; s < 2; s++) {
        int64_t best_idx = -1;
        double best_score = 0.0;
        uint64_t best_rec = 0;
        int best_snp = 0;
        bool stop = false;
        for (int ki = nK - 1; ki >= 0 && !stop; ki--) {                              // newest first (push_front, :475,489)
            for (int c = n_chunks - 1; c >= 0 && !stop; c--) {
                const uint64_t em = emask[ki * n_chunks + c];
                if (!em) continue;
                const int a = c * 64 + lane;
                const bool mine = (em & lane_bit) != 0;
                const int64_t idx = base + ((int64_t)ki * 2 + s) * A + a;
                const uint64_t r = mine ? records[idx] : 0;
                const double sc = mine ? scores[idx] : 0.0;
                int ext_copy = (int)MIPGEN_REC_EXT_COPY(r), lig_copy = (int)MIPGEN_REC_LIG_COPY(r);
                const int e = mine ? P->arm_ext[a] : 1, l = mine ? P->arm_lig[a] : 1;
                if (mine && (ext_copy == 65535 || lig_copy == 65535) && R.copy_off >= 0) {
                    // the record's 16-bit fields saturate; the reference compares bwa's unbounded X0 counts (mipgen.cpp:586-587,1692,1709):
                    // fetch the true values from the copy table
                    const int C = P->max_capture - (R.k0 + ki) * P->inc, p = R.first_pos + pi, ss = C - e - l;
                    const int ext_start = s ? p + ss : p - e, lig_start = s ? p - l : p + ss;
                    const int se = P->len_slot[e], sl = P->len_slot[l];
                    const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
                    ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
                    lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
                }
                const bool ok = mine && !((int64_t)ext_copy * lig_copy > max_product) && !(MIPGEN_REC_FLAGS(r) & MIPGEN_FLAG_MAPPING);   // :1689-1690
                const int cur_copy = ext_copy > lig_copy ? ext_copy : lig_copy;
                const double cur_masked = (double)MIPGEN_REC_MASKED_N(r) / (double)(l + e);
                const int snp = (int)MIPGEN_REC_SNP_COUNT(r);
                uint64_t pending = __ballot(ok);
                while (pending) {
                    // every pending lane evaluates the take rules against the current state; the first taker in fold order (highest pair
                    // index first) is applied, the lanes after it are re-evaluated
                    bool take = false, update_chosen = true, stops = false;
                    if (best_idx < 0) take = true;                                                                    // :1695
                    else if (cur_masked > thr && cur_masked < chosen_masked) take = true;                             // :1701
                    else if (cur_copy > target_copy && cur_copy < chosen_copy) take = true;                           // :1709
                    else if (cur_copy <= target_copy) {
                        if (sc < lower && sc > best_score) take = true;                                               // :1717
                        else if (sc > lower) {
                            if (snp < best_snp) take = true;                                                          // :1725
                            else if (snp == best_snp && sc > best_score) { take = true; update_chosen = false; stops = sc > upper; }   // :1731-1737
                        }
                    }
                    const uint64_t tmask = __ballot(take) & pending;
                    if (!tmask) break;
                    const int f = top_bit(tmask);
                    best_idx = base + ((int64_t)ki * 2 + s) * A + (c * 64 + f);
                    best_score = readlane_d(sc, f);
                    best_rec = readlane_u64(r, f);
                    best_snp = (int)MIPGEN_REC_SNP_COUNT(best_rec);
                    if (__builtin_amdgcn_readlane((int)update_chosen, f)) {
                        chosen_masked = readlane_d(cur_masked, f);
                        chosen_copy = __builtin_amdgcn_readlane(cur_copy, f);
                    }
                    if (__builtin_amdgcn_readlane((int)stops, f)) { stop = true; break; }
                    pending &= (1ull << f) - 1;
                }
            }
        }
        if (lane == 0) {
            mipgen_survivor out;
            out.cand_index = best_idx < 0 ? -1 : best_idx + cand_base; out.score = best_score; out.record = best_rec;
            survivors[2 * (int64_t)gp + s] = out;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same for designs with at most 64 arm pairs (the reference's defaults give 25): one row fits one wavefront, and the walk over the
// arm-sum lists of a row turns into lane-parallel work.  Inside a list nothing depends on the lists before it (previous_plus_score /
// previous_minus_score / skip_ahead are reset at :435-436), so every lane finds the constructed pair before it in ITS list, tests the
// :494 heuristic and masks itself off behind the first hit - all lists of the row at once, one wave shuffle per row.  What does chain
// through the lists is previous_best_score (:434,495); it can only switch a list off once some list's last pair scored above the upper
// threshold, so the scalar chain over the lists runs for those rows alone.  Rows are loaded REPLAY_ROWS at a time: a position costs
// ceil(n_sizes / REPLAY_ROWS) dependent round trips to memory in the replay and as many per strand in the condense fold, not one per row.
// ---------------------------------------------------------------------------------------------------------
#define REPLAY_ROWS 8

template <bool WIDE>
__global__ __launch_bounds__(REPLAY_WAVES * 64) void k_replay_condense_narrow(
    int total_pos, int lds_pitch, const DevParams* __restrict__ P, const DevRegion* __restrict__ regions,
    const int32_t* __restrict__ pos_region, const int32_t* __restrict__ pos_local, const double* __restrict__ scores,
    const uint64_t* __restrict__ records, const int32_t* __restrict__ copy, int64_t cand_base, uint8_t* __restrict__ emitted,
    mipgen_survivor* __restrict__ survivors, unsigned long long* __restrict__ emitted_per_region)
{
    extern __shared__ uint64_t s_masks[];                                  // [wave][size index]: emitted lanes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gp = blockIdx.x * REPLAY_WAVES + wave;
    if (gp >= total_pos) return;
    uint64_t* emask = s_masks + (size_t)wave * lds_pitch;
    const int ri = pos_region[gp], pi = pos_local[gp];
    const DevRegion& R = regions[ri];
    const int A = P->n_pairs, nK = R.n_sizes;
    const int64_t per_pos = (int64_t)nK * A * 2;
    const int64_t base = R.out_off + (int64_t)pi * per_pos;
    const double upper = P->upper, lower = P->lower;
    const bool heuristic = P->score_method == MIPGEN_SCORE_LOGISTIC && P->logistic_heuristic;
    const uint64_t lane_bit = 1ull << lane, below_me = lane_bit - 1;

    // my pair, and the lanes of my arm-sum list
    const bool in = lane < A;
    const int e = in ? P->arm_ext[lane] : 1, l = in ? P->arm_lig[lane] : 1, gend = in ? P->group_end[lane] : 0;
    const uint64_t list_ends = __ballot(in && lane + 1 == gend);           // the last pair of every list
    const uint64_t min_lists = __ballot(in && e + l == P->key_min_sum);        // pairs of the list :434 never switches off
    const uint64_t ends_below = list_ends & below_me;
    const int lo = ends_below ? top_bit(ends_below) + 1 : 0;
    const uint64_t seg = in ? ((gend >= 64 ? ~0ull : (1ull << gend) - 1) & ~((1ull << lo) - 1)) : 0;
    const uint64_t seg_below = seg & below_me, seg_above = seg & ~below_me & ~lane_bit;
    // :443-444, the bounds skips, from the geometry alone: a pair is constructed at scan start p and capture size C iff p > max(e, l),
    // p + C - 1 - min(e, l) <= seq_stop and the scan target is not empty (C > e + l) - exactly the VALID bit of the candidate's record
    // (kernels_logistic_dense.hip, kernels_logistic.hip: `ss > 0`; mipgen_accel_create refuses parameter sets with min_capture <= max arm sum, the
    // test keeps this kernel in step with the other two replay kernels should that ever change), which the replay therefore does not fetch: 16 instead of
    // 24 bytes per pair and row.  One unsigned compare per row: e + l < C <= c_lim  <=>  (unsigned)(C - e - l - 1) <= (unsigned)(c_lim - e - l - 1).
    const int p_scan = R.first_pos + pi;
    const int c_lim = R.seq_stop - p_scan + 1 + min(e, l);                   // largest capture size that still fits
    const int c_span = c_lim - (e + l) - 1;
    const bool ok_lo = in && p_scan > max(e, l) && c_span >= 0;
    const int c_top = P->max_capture - R.k0 * P->inc - (e + l) - 1, c_inc = P->inc;   // (capture size of row ki) - e - l - 1 = c_top - ki * c_inc

    // ---- replay, mipgen.cpp:426-497 ---------------------------------------------------------------------
    unsigned long long n_emitted = 0;
    double pbs = 0.0;                                                        // previous_best_score (:426)
    for (int k0 = 0; k0 < nK; k0 += REPLAY_ROWS) {
        double bp[REPLAY_ROWS], bm[REPLAY_ROWS];
        if (pbs > upper) {                                                   // :430 - nothing more is constructed at this position
            if (lane < REPLAY_ROWS && k0 + lane < nK) emask[k0 + lane] = 0;
            continue;
        }
        if constexpr (WIDE) {
        // every load of the batch is issued before the first value is looked at (selects afterwards: with the loads predicated per lane and the flag
        // tests next to them the compiler waits for each row's record before it issues the next row - eight dependent round trips; designs of few
        // capture sizes run faster that way, hence the two forms)
        {
            const double* srow = scores + base;
            const int la = min(lane, A - 1);
#pragma unroll
            for (int q = 0; q < REPLAY_ROWS; q++) {
                bp[q] = 0.0; bm[q] = 0.0;
                if (k0 + q < nK) {                                              // (wave-uniform: the rows behind the last capture size are not fetched)
                    const int off = ((k0 + q) * 2) * A + la;
                    bp[q] = srow[off]; bm[q] = srow[off + A];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < REPLAY_ROWS; q++) {
                bp[q] = in ? bp[q] : 0.0; bm[q] = in ? bm[q] : 0.0;
            }
        }
        } else {
#pragma unroll
        for (int q = 0; q < REPLAY_ROWS; q++) {
            const bool on = in && k0 + q < nK;
            const int64_t idx = base + ((int64_t)(k0 + q) * 2) * A + lane;
            bp[q] = on ? scores[idx] : 0.0; bm[q] = on ? scores[idx + A] : 0.0;
        }
        }
#pragma unroll
        for (int q = 0; q < REPLAY_ROWS; q++) {
            const int ki = k0 + q;
            if (ki >= nK) break;
            uint64_t emit_all = 0;
            if (!(pbs > upper)) {                                            // :430
                const double plus = bp[q], minus = bm[q];
                const bool valid = ok_lo && (unsigned)(c_top - ki * c_inc) <= (unsigned)c_span;  // :443-444 (false outside the row)
                const uint64_t vmask = __ballot(valid);
                bool mine = valid;
                // (rows whose constructed pairs all score in [0, 1) cannot trigger :494 - every truncated score is 0: see k_replay_condense_carry)
                if (heuristic && vmask && __ballot(valid && !(plus >= 0.0 && plus < 1.0 && minus >= 0.0 && minus < 1.0))) {
                    // the pair before me in my list that was constructed: its truncated scores are what :494 compares against
                    const uint64_t below = vmask & seg_below;
                    const int src = below ? top_bit(below) : lane;
                    const double pp = __shfl(plus, src, 64), pm = __shfl(minus, src, 64);
                    const int ip = below ? to_int_x86(pp) : 0, im = below ? to_int_x86(pm) : 0;
                    const bool cond = valid && plus < (double)ip && minus < (double)im;           // :494
                    const uint64_t cmask = __ballot(cond);
                    mine = valid && !(cmask & seg_below);                    // the first hit of a list is still constructed, nothing behind it
                }
                emit_all = __ballot(mine);
                if (emit_all) {
                    const double pb = (minus > plus) ? minus : plus;         // :495 if I am the last pair constructed in my list
                    const bool last_of_list = mine && !(emit_all & seg_above);
                    if (!__ballot(last_of_list && pb > upper)) {
                        pbs = readlane_d(pb, top_bit(emit_all));             // no list of this row can switch a later one off
                    } else {
                        uint64_t ends = list_ends;
                        int a0 = 0;
                        while (ends) {                                       // the chain through previous_best_score, list by list
                            const int a1 = __builtin_ctzll(ends) + 1;
                            ends &= ends - 1;
                            const uint64_t sg = (a1 >= 64 ? ~0ull : (1ull << a1) - 1) & ~((1ull << a0) - 1);
                            if (pbs > upper && !((min_lists >> a0) & 1)) emit_all &= ~sg;          // :434
                            else if (emit_all & sg) pbs = readlane_d(pb, top_bit(emit_all & sg));
                            a0 = a1;
                        }
                    }
                    n_emitted += 2ull * (unsigned)__builtin_popcountll(emit_all);
                    if (emitted && (emit_all & lane_bit)) {
                        const int64_t idx = base + ((int64_t)ki * 2) * A + lane;
                        emitted[idx] = 1; emitted[idx + A] = 1;
                    }
                }
            }
            if (lane == 0) emask[ki] = emit_all;
        }
    }
    if (n_emitted && lane == 0) atomicAdd(&emitted_per_region[ri], n_emitted);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");              // the lane masks written above are read by every lane below
    __builtin_amdgcn_wave_barrier();

    // ---- condense, mipgen.cpp:1670-1746 -------------------------------------------------------------------
    const int target_copy = P->target_arm_copy;
    const int64_t max_product = P->max_arm_copy_product;
    const double thr = P->masked_arm_threshold;
    const bool masked_any_counts = thr < 0.0;                                        // then even an arm without masked bases is "above the threshold" (:1701)
    const double arm_sum = (double)(l + e);
    int chosen_copy = 0;
    double chosen_masked = 0.0;                                                      // per position, not per strand (:1677-1680)
    const int k_top = ((nK - 1) / REPLAY_ROWS) * REPLAY_ROWS;
    for (int s = 0; s < 2; s++) {
        int64_t best_idx = -1;
        double best_score = 0.0;
        uint64_t best_rec = 0;
        int best_snp = 0;
        bool stop = false;
        for (int k0 = k_top; k0 >= 0 && !stop; k0 -= REPLAY_ROWS) {                  // newest first (push_front, :475,489)
            uint64_t em[REPLAY_ROWS], br[REPLAY_ROWS];
            double bs[REPLAY_ROWS];
            if constexpr (WIDE) {
            {
                const double* srow = scores + base;
                const uint64_t* rrow = records + base;
                const int la = min(lane, A - 1);
#pragma unroll
                for (int q = 0; q < REPLAY_ROWS; q++) em[q] = k0 + q < nK ? emask[k0 + q] : 0;
#pragma unroll
                for (int q = 0; q < REPLAY_ROWS; q++) {
                    br[q] = 0; bs[q] = 0.0;
                    if (em[q]) {                                                 // (wave-uniform: a row nothing was emitted from is not fetched)
                        const int off = ((k0 + q) * 2 + s) * A + la;
                        br[q] = rrow[off]; bs[q] = srow[off];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < REPLAY_ROWS; q++) {
                    const bool mine = (em[q] & lane_bit) != 0;
                    br[q] = mine ? br[q] : 0; bs[q] = mine ? bs[q] : 0.0;
                }
            }
            } else {
#pragma unroll
            for (int q = 0; q < REPLAY_ROWS; q++) {
                em[q] = k0 + q < nK ? emask[k0 + q] : 0;
                const bool mine = (em[q] & lane_bit) != 0;
                const int64_t idx = base + ((int64_t)(k0 + q) * 2 + s) * A + lane;
                br[q] = mine ? records[idx] : 0; bs[q] = mine ? scores[idx] : 0.0;
            }
            }
#pragma unroll
            for (int q = REPLAY_ROWS - 1; q >= 0; q--) {
                if (stop || !em[q]) continue;
                const int ki = k0 + q;
                const bool mine = (em[q] & lane_bit) != 0;
                const uint64_t r = br[q];
                const double sc = bs[q];
                int ext_copy = (int)MIPGEN_REC_EXT_COPY(r), lig_copy = (int)MIPGEN_REC_LIG_COPY(r);
                // Quick reject, exact: once a candidate is held (:1695 no longer applies) every other rule needs a higher score (:1717,1731),
                // fewer SNPs (:1725), a copy number above the target (:1709) or masked bases (:1701) - most rows have none of these
                if (best_idx >= 0 && !(em[q] & __ballot(sc > best_score || ext_copy > target_copy || lig_copy > target_copy || masked_any_counts ||
                                                        MIPGEN_REC_MASKED_N(r) != 0 || (int)MIPGEN_REC_SNP_COUNT(r) < best_snp))) continue;
                const bool saturated = mine && (ext_copy == 65535 || lig_copy == 65535) && R.copy_off >= 0;
                if (__ballot(saturated) && saturated) {                    // (the uniform test first: no exec-mask bookkeeping in the common case)
                    // the record's 16-bit fields saturate; the reference compares bwa's unbounded X0 counts (mipgen.cpp:586-587,1692,1709):
                    // fetch the true values from the copy table
                    const int C = P->max_capture - (R.k0 + ki) * P->inc, p = R.first_pos + pi, ss = C - e - l;
                    const int ext_start = s ? p + ss : p - e, lig_start = s ? p - l : p + ss;
                    const int se = P->len_slot[e], sl = P->len_slot[l];
                    const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
                    ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
                    lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
                }
                const bool ok = mine && !((int64_t)ext_copy * lig_copy > max_product) && !(MIPGEN_REC_FLAGS(r) & MIPGEN_FLAG_MAPPING);   // :1689-1690
                const int cur_copy = ext_copy > lig_copy ? ext_copy : lig_copy;
                double cur_masked = 0.0;                                   // (the division only where an emitted candidate has masked bases: 0 / x is exactly 0)
                if (em[q] & __ballot(MIPGEN_REC_MASKED_N(r) != 0)) cur_masked = (double)MIPGEN_REC_MASKED_N(r) / arm_sum;
                const int snp = (int)MIPGEN_REC_SNP_COUNT(r);
                uint64_t pending = __ballot(ok);
                while (pending) {
                    // every pending lane evaluates the take rules against the current state; the first taker in fold order (highest pair
                    // index first) is applied, the lanes after it are re-evaluated
                    // the else-if chain of :1695-1737 as predicates (bitwise: no divergent branches); only the last rule leaves chosen_* alone
                    const bool rA = best_idx < 0;                                                                     // :1695
                    const bool rB = (cur_masked > thr) & (cur_masked < chosen_masked);                                // :1701
                    const bool rC = (cur_copy > target_copy) & (cur_copy < chosen_copy);                              // :1709
                    const bool rD = cur_copy <= target_copy;
                    const bool above = sc > best_score;
                    const bool rE = (sc < lower) & above;                                                             // :1717
                    const bool hi_sc = sc > lower;
                    const bool rF = snp < best_snp;                                                                   // :1725
                    const bool rG = (snp == best_snp) & above;                                                        // :1731-1737
                    const bool early = rA | rB | rC;
                    const bool last_rule = !early & rD & !rE & hi_sc & !rF & rG;
                    const bool take = early | (rD & (rE | (hi_sc & (rF | rG))));
                    const bool update_chosen = !last_rule, stops = last_rule & (sc > upper);
                    const uint64_t tmask = __ballot(take) & pending;
                    if (!tmask) break;
                    const int f = top_bit(tmask);
                    best_idx = base + ((int64_t)ki * 2 + s) * A + f;
                    best_score = readlane_d(sc, f);
                    best_rec = readlane_u64(r, f);
                    best_snp = (int)MIPGEN_REC_SNP_COUNT(best_rec);
                    if (__builtin_amdgcn_readlane((int)update_chosen, f)) {
                        chosen_masked = readlane_d(cur_masked, f);
                        chosen_copy = __builtin_amdgcn_readlane(cur_copy, f);
                    }
                    if (__builtin_amdgcn_readlane((int)stops, f)) { stop = true; break; }
                    pending &= (1ull << f) - 1;
                }
            }
        }
        if (lane == 0) {
            mipgen_survivor out;
            out.cand_index = best_idx < 0 ? -1 : best_idx + cand_base; out.score = best_score; out.record = best_rec;
            survivors[2 * (int64_t)gp + s] = out;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// The same fold for designs with at most CR capture sizes (CR <= 8: exome-style ranges such as 150-170 step 5).  A position's rows fit one
// batch, so the replay keeps them in registers - both strands' scores AND records - and the condense fold reads nothing again: three
// dependent round trips to memory per position (position map, region, rows) instead of five.  The kernel is latency bound (a position is
// 2 CR short rows), which is why this matters: 8,192 exons 10.4 -> see DESIGN.md.  The emitted lanes of a row are wave-uniform masks and
// stay in scalar registers (no LDS).
// ---------------------------------------------------------------------------------------------------------
template <int CR>
__global__ __launch_bounds__(REPLAY_WAVES * 64) void k_replay_condense_carry(
    int total_pos, const DevParams* __restrict__ P, const DevRegion* __restrict__ regions,
    const int32_t* __restrict__ pos_region, const int32_t* __restrict__ pos_local, const double* __restrict__ scores,
    const uint64_t* __restrict__ records, const int32_t* __restrict__ copy, int64_t cand_base, uint8_t* __restrict__ emitted,
    mipgen_survivor* __restrict__ survivors, unsigned long long* __restrict__ emitted_per_region)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gp = blockIdx.x * REPLAY_WAVES + wave;
    if (gp >= total_pos) return;
    const int ri = pos_region[gp], pi = pos_local[gp];
    const DevRegion& R = regions[ri];
    const int A = P->n_pairs, nK = R.n_sizes;                               // nK <= CR (the launcher checks the parameter set)
    const int64_t per_pos = (int64_t)nK * A * 2;
    const int64_t base = R.out_off + (int64_t)pi * per_pos;
    const double upper = P->upper, lower = P->lower;
    const bool heuristic = P->score_method == MIPGEN_SCORE_LOGISTIC && P->logistic_heuristic;
    const uint64_t lane_bit = 1ull << lane, below_me = lane_bit - 1;

    const bool in = lane < A;
    const int e = in ? P->arm_ext[lane] : 1, l = in ? P->arm_lig[lane] : 1, gend = in ? P->group_end[lane] : 0;
    const uint64_t list_ends = __ballot(in && lane + 1 == gend);
    const uint64_t min_lists = __ballot(in && e + l == P->key_min_sum);
    const uint64_t ends_below = list_ends & below_me;
    const int lo = ends_below ? top_bit(ends_below) + 1 : 0;
    const uint64_t seg = in ? ((gend >= 64 ? ~0ull : (1ull << gend) - 1) & ~((1ull << lo) - 1)) : 0;
    const uint64_t seg_below = seg & below_me, seg_above = seg & ~below_me & ~lane_bit;
    const int target_copy = P->target_arm_copy;
    const int64_t max_product = P->max_arm_copy_product;
    const double thr = P->masked_arm_threshold;
    const bool masked_any_counts = thr < 0.0;                                        // then even an arm without masked bases is "above the threshold" (:1701)
    const double arm_sum = (double)(l + e);

    // ---- the position's rows: scores and records of both strands, one round trip ----
    double bp[CR], bm[CR];
    uint64_t rp[CR], rm[CR], em[CR];
#pragma unroll
    for (int q = 0; q < CR; q++) {
        const bool on = in && q < nK;
        const int64_t idx = base + ((int64_t)q * 2) * A + lane;
        bp[q] = on ? scores[idx] : 0.0; bm[q] = on ? scores[idx + A] : 0.0;
        rp[q] = on ? records[idx] : 0; rm[q] = on ? records[idx + A] : 0;
        em[q] = 0;
    }
    // ---- replay, mipgen.cpp:426-497 (as k_replay_condense_narrow) ----
    unsigned long long n_emitted = 0;
    double pbs = 0.0;
#pragma unroll
    for (int q = 0; q < CR; q++) {
        if (q >= nK) break;
        uint64_t emit_all = 0;
        if (!(pbs > upper)) {                                                // :430
            const double plus = bp[q], minus = bm[q];
            const bool valid = in && (MIPGEN_REC_FLAGS(rp[q]) & MIPGEN_FLAG_VALID) != 0;         // :443-444
            const uint64_t vmask = __ballot(valid);
            bool mine = valid;
            // :494 compares a score with the TRUNCATED score of the pair before it: while every constructed pair of the row scores in [0, 1) - any
            // logistic row without a guard value or a NaN - all those integers are 0 and the test (plus < 0 && minus < 0) fails on every lane: one
            // ballot instead of the shuffles (exact: any other row takes the general path)
            if (heuristic && vmask && __ballot(valid && !(plus >= 0.0 && plus < 1.0 && minus >= 0.0 && minus < 1.0))) {
                const uint64_t below = vmask & seg_below;
                const int src = below ? top_bit(below) : lane;
                const double pp = __shfl(plus, src, 64), pm = __shfl(minus, src, 64);
                const int ip = below ? to_int_x86(pp) : 0, im = below ? to_int_x86(pm) : 0;
                const bool cond = valid && plus < (double)ip && minus < (double)im;               // :494
                const uint64_t cmask = __ballot(cond);
                mine = valid && !(cmask & seg_below);
            }
            emit_all = __ballot(mine);
            if (emit_all) {
                const double pb = (minus > plus) ? minus : plus;             // :495
                const bool last_of_list = mine && !(emit_all & seg_above);
                if (!__ballot(last_of_list && pb > upper)) {
                    pbs = readlane_d(pb, top_bit(emit_all));
                } else {
                    uint64_t ends = list_ends;
                    int a0 = 0;
                    while (ends) {
                        const int a1 = __builtin_ctzll(ends) + 1;
                        ends &= ends - 1;
                        const uint64_t sg = (a1 >= 64 ? ~0ull : (1ull << a1) - 1) & ~((1ull << a0) - 1);
                        if (pbs > upper && !((min_lists >> a0) & 1)) emit_all &= ~sg;              // :434
                        else if (emit_all & sg) pbs = readlane_d(pb, top_bit(emit_all & sg));
                        a0 = a1;
                    }
                }
                n_emitted += 2ull * (unsigned)__builtin_popcountll(emit_all);
                if (emitted && __builtin_amdgcn_inverse_ballot_w64(emit_all)) {
                    const int64_t idx = base + ((int64_t)q * 2) * A + lane;
                    emitted[idx] = 1; emitted[idx + A] = 1;
                }
            }
        }
        em[q] = emit_all;
    }
    if (n_emitted && lane == 0) atomicAdd(&emitted_per_region[ri], n_emitted);

    // ---- condense, mipgen.cpp:1670-1746, from the registers ----
    int chosen_copy = 0;
    double chosen_masked = 0.0;
    for (int s = 0; s < 2; s++) {
        int64_t best_idx = -1;
        double best_score = 0.0;
        uint64_t best_rec = 0;
        int best_snp = 0;
        bool stop = false;
#pragma unroll
        for (int q = CR - 1; q >= 0; q--) {                                  // newest first (push_front, :475,489)
            if (stop || !em[q]) continue;
            const int ki = q;
            const uint64_t r = s ? rm[q] : rp[q];
            const double sc = s ? bm[q] : bp[q];
            int ext_copy = (int)MIPGEN_REC_EXT_COPY(r), lig_copy = (int)MIPGEN_REC_LIG_COPY(r);
            // quick reject, exact (see k_replay_condense_narrow)
            // (the emitted lanes gate every ballot as a uniform mask: no per-lane copy of it)
            if (best_idx >= 0 && !(em[q] & __ballot(sc > best_score || ext_copy > target_copy || lig_copy > target_copy || masked_any_counts ||
                                                    MIPGEN_REC_MASKED_N(r) != 0 || (int)MIPGEN_REC_SNP_COUNT(r) < best_snp))) continue;
            const uint64_t sat_mask = R.copy_off >= 0 ? em[q] & __ballot(ext_copy == 65535 || lig_copy == 65535) : 0;
            if (sat_mask && __builtin_amdgcn_inverse_ballot_w64(sat_mask)) {
                const int C = P->max_capture - (R.k0 + ki) * P->inc, p = R.first_pos + pi, ss = C - e - l;
                const int ext_start = s ? p + ss : p - e, lig_start = s ? p - l : p + ss;
                const int se = P->len_slot[e], sl = P->len_slot[l];
                const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
                ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
                lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
            }
            const bool ok = !((int64_t)ext_copy * lig_copy > max_product) && !(MIPGEN_REC_FLAGS(r) & MIPGEN_FLAG_MAPPING);   // :1689-1690 (of an emitted lane: the mask below)
            const int cur_copy = ext_copy > lig_copy ? ext_copy : lig_copy;
            // (the division only where an emitted candidate has masked bases at all: 0 / x is exactly 0)
            double cur_masked = 0.0;
            if (em[q] & __ballot(MIPGEN_REC_MASKED_N(r) != 0)) cur_masked = (double)MIPGEN_REC_MASKED_N(r) / arm_sum;
            const int snp = (int)MIPGEN_REC_SNP_COUNT(r);
            uint64_t pending = em[q] & __ballot(ok);
            while (pending) {
                const bool rA = best_idx < 0;                                                                     // :1695
                const bool rB = (cur_masked > thr) & (cur_masked < chosen_masked);                                // :1701
                const bool rC = (cur_copy > target_copy) & (cur_copy < chosen_copy);                              // :1709
                const bool rD = cur_copy <= target_copy;
                const bool above = sc > best_score;
                const bool rE = (sc < lower) & above;                                                             // :1717
                const bool hi_sc = sc > lower;
                const bool rF = snp < best_snp;                                                                   // :1725
                const bool rG = (snp == best_snp) & above;                                                        // :1731-1737
                const bool early = rA | rB | rC;
                const bool last_rule = !early & rD & !rE & hi_sc & !rF & rG;
                const bool take = early | (rD & (rE | (hi_sc & (rF | rG))));
                const bool update_chosen = !last_rule, stops = last_rule & (sc > upper);
                const uint64_t tmask = __ballot(take) & pending;
                if (!tmask) break;
                const int f = top_bit(tmask);
                best_idx = base + ((int64_t)ki * 2 + s) * A + f;
                best_score = readlane_d(sc, f);
                best_rec = readlane_u64(r, f);
                best_snp = (int)MIPGEN_REC_SNP_COUNT(best_rec);
                if (__builtin_amdgcn_readlane((int)update_chosen, f)) {
                    chosen_masked = readlane_d(cur_masked, f);
                    chosen_copy = __builtin_amdgcn_readlane(cur_copy, f);
                }
                if (__builtin_amdgcn_readlane((int)stops, f)) { stop = true; break; }
                pending &= (1ull << f) - 1;
            }
        }
        if (lane == 0) {
            mipgen_survivor out;
            out.cand_index = best_idx < 0 ? -1 : best_idx + cand_base; out.score = best_score; out.record = best_rec;
            survivors[2 * (int64_t)gp + s] = out;
        }
    }
}

extern "C" hipError_t mipgen_launch_replay_condense(
    hipStream_t stream, int n_regions, int total_pos, const DevParams* P, int n_pairs, int n_sizes_max, const DevRegion* regions,
    const int32_t* pos_region, const int32_t* pos_local, const double* scores, const uint64_t* records, const int32_t* copy, int64_t cand_base,
    uint8_t* emitted, mipgen_survivor* survivors, unsigned long long* emitted_per_region)
{
    (void)n_regions;
    if (total_pos <= 0) return hipSuccess;
    const int n_chunks = (n_pairs + 63) / 64;
    const int pitch = n_sizes_max * n_chunks;
    const size_t lds = (size_t)REPLAY_WAVES * pitch * sizeof(uint64_t);
    const dim3 grid((total_pos + REPLAY_WAVES - 1) / REPLAY_WAVES), block(REPLAY_WAVES * 64);
    if (n_chunks == 1 && n_sizes_max <= 3)
        hipLaunchKernelGGL(k_replay_condense_carry<3>, grid, block, 0, stream, total_pos, P, regions, pos_region, pos_local, scores, records, copy, cand_base, emitted,
                           survivors, emitted_per_region);
    else if (n_chunks == 1 && n_sizes_max <= 5)
        hipLaunchKernelGGL(k_replay_condense_carry<5>, grid, block, 0, stream, total_pos, P, regions, pos_region, pos_local, scores, records, copy, cand_base, emitted,
                           survivors, emitted_per_region);
    else if (n_chunks == 1 && n_sizes_max <= 8)
        hipLaunchKernelGGL(k_replay_condense_carry<8>, grid, block, 0, stream, total_pos, P, regions, pos_region, pos_local, scores, records, copy, cand_base, emitted,
                           survivors, emitted_per_region);
    else if (n_chunks == 1)
        if (n_sizes_max > 2 * REPLAY_ROWS)
            hipLaunchKernelGGL(k_replay_condense_narrow<true>, grid, block, lds, stream, total_pos, pitch, P, regions, pos_region, pos_local, scores, records, copy,
                               cand_base, emitted, survivors, emitted_per_region);
        else
            hipLaunchKernelGGL(k_replay_condense_narrow<false>, grid, block, lds, stream, total_pos, pitch, P, regions, pos_region, pos_local, scores, records, copy,
                               cand_base, emitted, survivors, emitted_per_region);
    else
        hipLaunchKernelGGL(k_replay_condense, grid, block, lds, stream, total_pos, n_chunks, pitch, P, regions, pos_region, pos_local, scores, records,
                           copy, cand_base, emitted, survivors, emitted_per_region);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------
// collapse_mips (/root/reference/mipgen.cpp:1616-1649) on the device: per base and strand, the best of the condensed survivors whose
// scan target covers the base.  The reference folds the survivors in scan-start order (install; fewer SNPs replaces; equal SNPs and a
// strictly higher score replaces - NaN scores never win or lose a comparison), so one thread per (base, strand) walks the <= max scan
// size survivors that can reach its base in that order.  Adjacent threads read adjacent survivors (24 B each): the window stays in L1.
// Result: for each base the scan-start index of the winning survivor, or -1.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_collapse(int n_tiles, const CollapseTile* __restrict__ tiles, const DevParams* __restrict__ P,
                                                  const DevRegion* __restrict__ regions, const int64_t* __restrict__ region_pos0,
                                                  const int64_t* __restrict__ region_base0, const mipgen_survivor* __restrict__ survivors,
                                                  const int32_t* __restrict__ copy, int64_t cand_base, int32_t* __restrict__ collapsed)
{
    // Two passes per workgroup (128 bases x 2 strands).  What a survivor contributes does not depend on the base: its last covered base, whether
    // the copy / masking filters keep it (:1628-1629), its SNP count and score are worked out once per survivor - the candidate index is taken
    // apart there, two integer divisions - and parked in LDS; a base then folds its <= max scan size survivors from there in scan-start order.
    extern __shared__ __align__(16) unsigned char col_smem[];
    const CollapseTile t = tiles[blockIdx.x];
    const DevRegion& R = regions[t.region];
    const int A = P->n_pairs, nK = R.n_sizes;
    const int max_scan = P->max_capture - R.k0 * P->inc - P->min_sum;
    const int n_base = R.n_pos + max_scan - 1;
    if (nK <= 0) return;
    const int p_lo = max(0, t.j0 - max_scan + 1), p_hi = min(t.j0 + 127, R.n_pos - 1), n_sv = 2 * (p_hi - p_lo + 1);
    double* s_score = (double*)col_smem;                                   // [n_sv]
    int* s_end = (int*)(s_score + max(n_sv, 0));                           // last base the survivor's scan target covers; -1: filtered out / absent
    int* s_snp = s_end + max(n_sv, 0);
    const int64_t per_pos = (int64_t)nK * A * 2;
    const mipgen_survivor* sv = survivors + 2 * region_pos0[t.region];
    const int target = P->target_arm_copy;
    const int64_t max_product = P->max_arm_copy_product;
    const double thr = P->masked_arm_threshold;
    for (int q = threadIdx.x; q < n_sv; q += blockDim.x) {
        const int pi = p_lo + (q >> 1), s = q & 1;
        const mipgen_survivor m = sv[2 * (int64_t)pi + s];
        int end = -1, snp = 0;
        if (m.cand_index >= 0) {
            const int rel = (int)(m.cand_index - cand_base - R.out_off - (int64_t)pi * per_pos);     // < per_pos: the candidate inside its position
            const int a = rel % A, ki = rel / (2 * A);
            const int e = P->arm_ext[a], l = P->arm_lig[a];
            const int C = P->max_capture - (R.k0 + ki) * P->inc, ss = C - e - l;
            int ext_copy = (int)MIPGEN_REC_EXT_COPY(m.record), lig_copy = (int)MIPGEN_REC_LIG_COPY(m.record);
            if ((ext_copy == 65535 || lig_copy == 65535) && R.copy_off >= 0) {  // saturated record fields: the true counts (see k_replay_condense)
                const int p = R.first_pos + pi;
                const int ext_start = s ? p + ss : p - e, lig_start = s ? p - l : p + ss;
                const int se = P->len_slot[e], sl = P->len_slot[l];
                const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
                ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
                lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
            }
            const bool keep = !((int64_t)ext_copy * lig_copy > max_product || ext_copy > target || lig_copy > target)          // :1628
                              && !((double)MIPGEN_REC_MASKED_N(m.record) / (double)(l + e) > thr);                          // :1629
            if (keep) end = pi + ss - 1;
            snp = (int)MIPGEN_REC_SNP_COUNT(m.record);
        }
        s_end[q] = end; s_snp[q] = snp; s_score[q] = m.score;
    }
    __syncthreads();
    const int j = t.j0 + (threadIdx.x >> 1), s = threadIdx.x & 1;
    if (j >= n_base) return;
    int best = -1, best_snp = 0;
    double best_score = 0.0;
    for (int pi = max(0, j - max_scan + 1); pi <= min(j, R.n_pos - 1); pi++) {
        const int q = 2 * (pi - p_lo) + s;
        if (s_end[q] < j) continue;                                          // absent, filtered out, or its scan target ends before this base
        const int snp = s_snp[q];
        const double sc = s_score[q];
        if (best < 0 || snp < best_snp || (sc > best_score && snp == best_snp)) { best = pi; best_snp = snp; best_score = sc; }   // :1634-1645
    }
    collapsed[region_base0[t.region] + 2 * (int64_t)j + s] = best;
}

extern "C" hipError_t mipgen_launch_collapse(hipStream_t stream, int n_tiles, const CollapseTile* tiles, const DevParams* P, const DevRegion* regions,
                                             const int64_t* region_pos0, const int64_t* region_base0, const mipgen_survivor* survivors,
                                             const int32_t* copy, int64_t cand_base, int32_t* collapsed, int max_scan_all)
{
    if (n_tiles <= 0) return hipSuccess;
    // LDS: the survivors of the 128 bases' scan positions and of the max_scan - 1 positions before them, both strands, 16 bytes each
    const size_t lds = (size_t)2 * (size_t)(128 + (max_scan_all > 0 ? max_scan_all : 0)) * 16;
    if (lds > 160 * 1024) return hipErrorInvalidValue;                       // (scan targets beyond ~5,000 bases)
    if (lds > 48 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)k_collapse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k_collapse, dim3(n_tiles), dim3(256), lds, stream, n_tiles, tiles, P, regions, region_pos0, region_base0, survivors, copy, cand_base, collapsed);
    return hipGetLastError();
}

// the scan-position map of a batch (position -> region, position inside the region) from the regions' first-position offsets:
// region_pos0[r] = positions before region r, region_pos0[n_regions] = total
__global__ __launch_bounds__(256) void k_fill_pos_map(const int64_t* __restrict__ region_pos0, int n_regions, int64_t total, int32_t* __restrict__ pos_region,
                                                      int32_t* __restrict__ pos_local)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = n_regions - 1;
    while (lo < hi) {                                                  // the last region that starts at or before i (empty regions share a start: the last wins)
        const int mid = (lo + hi + 1) >> 1;
        if (region_pos0[mid] <= i) lo = mid; else hi = mid - 1;
    }
    pos_region[i] = lo;
    pos_local[i] = (int32_t)(i - region_pos0[lo]);
}

extern "C" hipError_t mipgen_launch_fill_pos_map(hipStream_t stream, const int64_t* region_pos0, int n_regions, int64_t total, int32_t* pos_region, int32_t* pos_local)
{
    if (total <= 0 || n_regions <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_pos_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, region_pos0, n_regions, total, pos_region, pos_local);
    return hipGetLastError();
}
