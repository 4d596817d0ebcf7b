// kernels_replay.hip — replay of the reference's score-dependent enumeration control flow over the scored dense
// grid, fused with the condense_mips fold, on the device.
//
//   mipgen::tile_regions early exits   /root/reference/mipgen.cpp:426-437,440,494-497
//   mipgen::condense_mips              /root/reference/mipgen.cpp:1670-1746
//
// Both are sequential state machines, but only *within* one scan-start position: every position is independent
// (previous_best_score is reset at :426; the condense state is declared per position at :1677-1680).  One lane
// therefore replays one position end to end: it walks the position's (capture size, arm pair) rows in
// generation order, marks the candidates the reference would have constructed, then folds the two strands'
// newest-first lists into the (position, strand) survivors that collapse/pick consume.  This removes the
// device-to-host copy of the dense grid (16 B/candidate) for silent designs: 2 survivors per position remain.
#include <hip/hip_runtime.h>
#include <limits.h>
#include "common.h"
#include "device_utils.h"

namespace {

// (int) of a double as x86-64 cvttsd2si gives it (the reference stores double scores into int locals, :496-497)
__device__ __forceinline__ int to_int_x86(double v)
{
    if (!(v == v) || v >= 2147483648.0 || v <= -2147483649.0) return INT_MIN;
    return (int)v;
}

}  // namespace

__global__ __launch_bounds__(64) void k_replay_condense(
    int total_pos, const DevParams* __restrict__ P, const DevRegion* __restrict__ regions,
    const int32_t* __restrict__ pos_region, const int32_t* __restrict__ pos_local, const double* __restrict__ scores,
    const uint64_t* __restrict__ records, const int32_t* __restrict__ copy, int64_t cand_base, uint8_t* __restrict__ emitted,
    mipgen_survivor* __restrict__ survivors, unsigned long long* __restrict__ emitted_per_region)
{
    const int gp = blockIdx.x * blockDim.x + threadIdx.x;
    if (gp >= total_pos) return;
    const int ri = pos_region[gp], pi = pos_local[gp];
    const DevRegion& R = regions[ri];
    const int A = P->n_pairs, nK = R.n_sizes;
    const int64_t per_pos = (int64_t)nK * A * 2;
    const int64_t base = R.out_off + (int64_t)pi * per_pos;
    const double upper = P->upper, lower = P->lower;
    const bool heuristic = P->score_method == MIPGEN_SCORE_LOGISTIC && P->logistic_heuristic;

    // ---- replay, mipgen.cpp:426-497 ---------------------------------------------------------------------
    unsigned long long n_emitted = 0;
    double previous_best_score = 0.0;
    for (int ki = 0; ki < nK; ki++) {
        if (previous_best_score > upper) continue;                                   // :430
        int a = 0;
        while (a < A) {
            const int a_end = P->group_end[a];
            const int sum = P->arm_ext[a] + P->arm_lig[a];
            if (previous_best_score > upper && sum != P->min_sum) { a = a_end; continue; }      // :434
            int previous_minus_score = 0, previous_plus_score = 0;
            bool skip_ahead = false;
            for (; a < a_end; a++) {
                if (skip_ahead) continue;                                            // :440
                const int64_t idx = base + ((int64_t)ki * 2) * A + a, idm = idx + A;     // strand-major rows
                if (!(MIPGEN_REC_FLAGS(records[idx]) & MIPGEN_FLAG_VALID)) continue; // :443-444
                emitted[idx] = 1; emitted[idm] = 1;
                n_emitted += 2;
                const double plus = scores[idx], minus = scores[idm];
                if (heuristic && plus < (double)previous_plus_score && minus < (double)previous_minus_score) skip_ahead = true;   // :494
                previous_best_score = (minus > plus) ? minus : plus;                 // :495
                previous_minus_score = to_int_x86(minus);                            // :496
                previous_plus_score = to_int_x86(plus);                              // :497
            }
        }
    }
    if (n_emitted) atomicAdd(&emitted_per_region[ri], n_emitted);

    // ---- condense, mipgen.cpp:1670-1746 -------------------------------------------------------------------
    int chosen_copy = 0;
    double chosen_masked = 0.0;                                                      // per position, not per strand (:1677-1680)
    for (int s = 0; s < 2; s++) {
        int64_t best_idx = -1;
        double best_score = 0.0;
        uint64_t best_rec = 0;
        bool skip_ahead = false;
        for (int64_t j = per_pos / 2 - 1; j >= 0 && !skip_ahead; j--) {              // newest first (push_front, :475,489)
            const int64_t idx = base + ((j / A) * 2 + s) * A + (j % A);              // j = (size, pair) in generation order
            if (!emitted[idx]) continue;
            const uint64_t r = records[idx];
            int ext_copy = (int)MIPGEN_REC_EXT_COPY(r), lig_copy = (int)MIPGEN_REC_LIG_COPY(r);
            const int a = (int)(j % A);
            if (ext_copy == 65535 || lig_copy == 65535) {
                // the record's 16-bit fields saturate; the reference compares bwa's unbounded X0 counts (mipgen.cpp:586-587,1692,1709):
                // fetch the true values from the copy table
                const int e = P->arm_ext[a], l = P->arm_lig[a];
                const int C = P->max_capture - (R.k0 + (int)(j / A)) * P->inc, p = R.first_pos + pi, ss = C - e - l;
                const int ext_start = s ? p + ss : p - e, lig_start = s ? p - l : p + ss;
                const int se = P->len_slot[e], sl = P->len_slot[l];
                const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
                if (R.copy_off >= 0) {
                    ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
                    lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
                }
            }
            if ((int64_t)ext_copy * lig_copy > P->max_arm_copy_product) continue;     // :1689
            if (MIPGEN_REC_FLAGS(r) & MIPGEN_FLAG_MAPPING) continue;                  // :1690
            const int cur_copy = ext_copy > lig_copy ? ext_copy : lig_copy;
            const double cur_masked = (double)MIPGEN_REC_MASKED_N(r) / (double)(P->arm_lig[a] + P->arm_ext[a]);
            const int snp = (int)MIPGEN_REC_SNP_COUNT(r);
            const double sc = scores[idx];
            bool take = false, update_chosen = true;
            if (best_idx < 0) take = true;                                                                    // :1695
            else if (cur_masked > P->masked_arm_threshold && cur_masked < chosen_masked) take = true;         // :1701
            else if (cur_copy > P->target_arm_copy && cur_copy < chosen_copy) take = true;                    // :1709
            else if (cur_copy <= P->target_arm_copy) {
                if (sc < lower && sc > best_score) take = true;                                               // :1717
                else if (sc > lower) {
                    const int bsnp = (int)MIPGEN_REC_SNP_COUNT(best_rec);
                    if (snp < bsnp) take = true;                                                              // :1725
                    else if (snp == bsnp && sc > best_score) {                                                // :1731-1737
                        take = true; update_chosen = false;
                        if (sc > upper) skip_ahead = true;
                    }
                }
            }
            if (take) {
                best_idx = idx; best_score = sc; best_rec = r;
                if (update_chosen) { chosen_masked = cur_masked; chosen_copy = cur_copy; }
            }
        }
        mipgen_survivor out;
        out.cand_index = best_idx < 0 ? -1 : best_idx + cand_base; out.score = best_score; out.record = best_rec;
        survivors[2 * (int64_t)gp + s] = out;
    }
}

extern "C" hipError_t mipgen_launch_replay_condense(
    hipStream_t stream, int n_regions, int total_pos, const DevParams* P, const DevRegion* regions, const int32_t* pos_region,
    const int32_t* pos_local, const double* scores, const uint64_t* records, const int32_t* copy, int64_t cand_base, uint8_t* emitted,
    mipgen_survivor* survivors, unsigned long long* emitted_per_region)
{
    (void)n_regions;
    if (total_pos <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_replay_condense, dim3((total_pos + 63) / 64), dim3(64), 0, stream, total_pos, P, regions, pos_region,
                       pos_local, scores, records, copy, cand_base, emitted, survivors, emitted_per_region);
    return hipGetLastError();
}
