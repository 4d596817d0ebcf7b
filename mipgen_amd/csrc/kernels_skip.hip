// kernels_skip.hip — the reference's dynamic skip between capture sizes (/root/reference/mipgen.cpp:430) applied between the capture-size RUNS of
// the dense SVR scorer, exactly.
//
// The enumeration of a scan position walks the capture sizes from the largest down and stops constructing candidates for good once a size starts
// with previous_best_score above the upper score limit (:430: `continue` for this and - nothing resets the score any more - every later size).  A
// region with more than nine capture sizes is scored in runs of <= 9 sizes (accel_tiles.hip: build_svr_tiles).  When the runs are launched in order,
// largest sizes first, the state of every position after run r is known before run r + 1 starts: a tile of run r + 1 whose positions have ALL
// stopped constructs nothing the reference would have constructed - its 14,000 candidates x n_sv kernel values need not be computed.  The replay
// of the early exits over the finished dense grid (kernels_replay.hip) stays the authority: it never consults rows behind the exit of a position,
// so what those rows hold (NaN, written here) does not matter, and a position is only marked as stopped when the decision cannot depend on the last
// digits of a score (a margin around the limit keeps the position alive otherwise - computing more is always safe).
//
//   k_svr_run_state   per scan position: previous_best_score carried through the sizes and arm-sum lists of run `level` (the control flow of
//                     mipgen.cpp:426-497 for the SVR: no :494 heuristic), "stopped" flag for the runs behind it
//   k_svr_tile_keep   per tile of the next run: 1 unless all of its positions have stopped
//   k_svr_tile_compact  kept tiles -> a dense tile list in their old (longest first) order; skipped tiles fill their score rows with NaN
#include <hip/hip_runtime.h>
#include "common.h"
#include "device_utils.h"

// run_bounds[region * max_levels + level] = first size index of the run | sizes of the run << 16 (0 = the region has no such run)
__global__ __launch_bounds__(256) void k_svr_run_state(int64_t n_pos, const DevParams* __restrict__ P, const DevRegion* __restrict__ regions,
                                                       const int32_t* __restrict__ pos_region, const int32_t* __restrict__ pos_local,
                                                       const uint32_t* __restrict__ run_bounds, int max_levels, int level, double margin,
                                                       const double* __restrict__ scores, const uint64_t* __restrict__ records,
                                                       double* __restrict__ pbs_io, uint8_t* __restrict__ state)
{
    const int64_t gp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gp >= n_pos) return;
    if (state[gp]) return;                                             // 1 = stopped, 2 = undecidable (kept alive for good)
    const int ri = pos_region[gp], pi = pos_local[gp];
    const uint32_t rb = run_bounds[(int64_t)ri * max_levels + level];
    const int ki0 = (int)(rb & 0xFFFFu), kc = (int)(rb >> 16);
    if (kc == 0) return;
    const DevRegion& R = regions[ri];
    const int A = P->n_pairs, nK = R.n_sizes, min_sum = P->key_min_sum;
    const double upper = P->upper;
    const int64_t base = R.out_off + (int64_t)pi * nK * A * 2;
    double pb = pbs_io[gp];                                            // previous_best_score (:426: 0 at the first size)
    uint8_t st = 0;
    auto above = [&](double v) -> int {                                // 1: v > upper, 0: not, 2: too close to call (another rounding of the score could flip it)
        if (v > upper + margin) return 1;
        if (!(v > upper - margin)) return 0;                           // (NaN: `NaN > upper` is false in the reference as well)
        return 2;
    };
    for (int ki = ki0; ki < ki0 + kc && !st; ki++) {
        const int ab = above(pb);
        if (ab == 2) { st = 2; break; }
        if (ab == 1) { st = 1; break; }                                // :430 - and nothing changes the score any more
        for (int a0 = 0; a0 < A;) {
            const int list_end = P->group_end[a0];
            const int ab2 = above(pb);
            if (ab2 == 2) { st = 2; break; }
            const bool skipped = ab2 == 1 && (P->arm_ext[a0] + P->arm_lig[a0]) != min_sum;       // :434
            if (!skipped) {
                // the list is constructed in full (SVR designs have no :494 heuristic); previous_best_score = its LAST constructed pair, i.e. the
                // last one the bounds tests of :443-444 let through (the record's valid flag; the same for both strands)
                for (int a = list_end - 1; a >= a0; a--) {
                    const int64_t idx = base + ((int64_t)ki * 2) * A + a;
                    if (MIPGEN_REC_FLAGS(records[idx]) & MIPGEN_FLAG_VALID) {
                        const double lp = scores[idx], lm = scores[idx + A];
                        pb = (lm > lp) ? lm : lp;                      // :495
                        break;
                    }
                }
            }
            a0 = list_end;
        }
    }
    if (!st) { const int ab = above(pb); if (ab == 1) st = 1; else if (ab == 2) st = 2; }     // the next size - in the next run - starts with this test
    pbs_io[gp] = pb;
    state[gp] = st;
}

__global__ __launch_bounds__(256) void k_svr_tile_keep(int n_tiles, const SvrTile* __restrict__ tiles, const int64_t* __restrict__ region_pos0, int64_t win_pos0,
                                                       const uint8_t* __restrict__ state, int64_t* __restrict__ keep)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) { keep[t] = 0; return; }                         // (the scan's extra element)
    const SvrTile tile = tiles[t];
    const int64_t pb = region_pos0[tile.region] - win_pos0 + tile.p0;
    int k = 0;
    for (int i = 0; i < tile.np; i++) if (state[pb + i] != 1) { k = 1; break; }
    keep[t] = k;
}

// one workgroup per tile of the run: a kept tile goes to slot offs[t] of the compacted list (the order - longest first - is kept); a skipped one
// fills its score rows with NaN (never read by the replay: every position of the tile stopped before these capture sizes) and is counted
__global__ __launch_bounds__(256) void k_svr_tile_compact(int n_tiles, const SvrTile* __restrict__ tiles, const int64_t* __restrict__ keep, const int64_t* __restrict__ offs,
                                                          SvrTile* __restrict__ out, const DevParams* __restrict__ P, const DevRegion* __restrict__ regions,
                                                          double* __restrict__ scores, unsigned long long* __restrict__ skipped)
{
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const SvrTile tile = tiles[t];
    if (keep[t]) { if (threadIdx.x == 0) out[offs[t]] = tile; return; }
    const DevRegion& R = regions[tile.region];
    const int A = P->n_pairs, nK = R.n_sizes;
    const int64_t n = (int64_t)tile.np * tile.kc * A;
    const double nan = __longlong_as_double(0x7FF8000000000000LL);
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t row = e / A;
        const int a = (int)(e - row * A);
        const int pl = (int)(row / tile.kc), kcI = (int)(row - (int64_t)pl * tile.kc);
        scores[R.out_off + ((((int64_t)(tile.p0 + pl) * nK + (tile.ki0 + kcI)) * 2 + tile.strand) * A) + a] = nan;
    }
    if (threadIdx.x == 0) atomicAdd(skipped, (unsigned long long)n);
}

extern "C" hipError_t mipgen_launch_svr_run_state(hipStream_t s, int64_t n_pos, const DevParams* P, const DevRegion* regions, const int32_t* pos_region,
                                                  const int32_t* pos_local, const uint32_t* run_bounds, int max_levels, int level, double margin,
                                                  const double* scores, const uint64_t* records, double* pbs, uint8_t* state)
{
    if (n_pos <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_svr_run_state, dim3((unsigned)((n_pos + 255) / 256)), dim3(256), 0, s, n_pos, P, regions, pos_region, pos_local, run_bounds, max_levels, level,
                       margin, scores, records, pbs, state);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_svr_tile_keep(hipStream_t s, int n_tiles, const SvrTile* tiles, const int64_t* region_pos0, int64_t win_pos0, const uint8_t* state,
                                                  int64_t* keep)
{
    if (n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_svr_tile_keep, dim3((unsigned)((n_tiles + 1 + 255) / 256)), dim3(256), 0, s, n_tiles, tiles, region_pos0, win_pos0, state, keep);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_svr_tile_compact(hipStream_t s, int n_tiles, const SvrTile* tiles, const int64_t* keep, const int64_t* offs, SvrTile* out,
                                                     const DevParams* P, const DevRegion* regions, double* scores, unsigned long long* skipped)
{
    if (n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_svr_tile_compact, dim3((unsigned)n_tiles), dim3(256), 0, s, n_tiles, tiles, keep, offs, out, P, regions, scores, skipped);
    return hipGetLastError();
}
