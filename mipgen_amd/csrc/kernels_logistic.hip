// kernels_logistic.hip — dense-grid candidate construction (integer record of design_mip) and the logistic
// scorer, hand-written for gfx950.
//
// Replaces, per candidate (p, C, (e,l), strand) of the dense grid:
//   Plus/MinusSVMipv4 ctor geometry      /root/reference/PlusSVMipv4.cpp:7-14, MinusSVMipv4.cpp:30-37
//   design_mip ints                      /root/reference/mipgen.cpp:599-762 (copies, masked N, mapping/masking/SNP flags)
//   SVMipv4::get_score                   /root/reference/SVMipv4.cpp:114-248
//
// Mapping to the hardware: one workgroup per run of scan-start positions of one region.  The region bytes the
// tile can touch are staged in LDS once and turned into packed 16-bit prefix counts (three u64 words per base),
// so every window statistic of a candidate (A/C/G counts, N guard, masked N, SNP counts, GC/AT class switches)
// is two ds_read_b64 and a subtract - no per-base loop, no std::string.  One candidate per lane; consecutive
// lanes own consecutive dense-grid indices, so the 8-byte score and 8-byte record stores of a wave are two
// contiguous 512-byte segments (coalesced, the only compulsory HBM traffic of this kernel: 16 B/candidate).
#include <hip/hip_runtime.h>
#include "common.h"
#include "device_utils.h"

#include "logistic_device.h"

// grid = number of tiles; block = LOG_THREADS; dynamic LDS = lds_bytes(span_max)
#define LOG_RINV 512
template <bool SCORE>
__global__ __launch_bounds__(LOG_THREADS) void k_records_logistic(
    const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, const LogTile* __restrict__ tiles,
    int n_tiles, const uint8_t* __restrict__ bases, const int32_t* __restrict__ copy,
    const uint8_t* __restrict__ unmap, const HostConsts* __restrict__ HC, double* __restrict__ scores,
    uint64_t* __restrict__ records, int64_t* __restrict__ sat_idx, unsigned int* __restrict__ sat_count, unsigned int sat_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LogTile tile = tiles[xcd_remap(blockIdx.x, n_tiles)];
    const DevRegion& R = regions[tile.region];
    const int tid = threadIdx.x;

    const int A = P->n_pairs;
    const int nK = R.n_sizes;
    const int Cmax = P->max_capture - R.k0 * P->inc;
    const int Lmax = max(P->e_max, P->l_max);
    const int p_first = R.first_pos + tile.p0;                 // chromosome coordinate of the tile's first p
    const int lo = p_first - Lmax;                             // chromosome coordinate of tile-local base 0
    const int span = tile.np + Cmax + Lmax;                    // bases [lo, lo+span)

    uint64_t* W0 = (uint64_t*)smem;                            // span+1 each
    uint64_t* W1 = W0 + (span + 1);
    uint64_t* W2 = W1 + (span + 1);
    uint64_t* scratch = W2 + (span + 1);                       // 8
    double* rinv = (double*)(scratch + 8);                     // LOG_RINV entries: 1.0 / i by true division
    uint8_t* sb = (uint8_t*)(rinv + LOG_RINV);                 // span

    for (int i = tid; i < LOG_RINV; i += LOG_THREADS) rinv[i] = i > 0 ? 1.0 / (double)i : 0.0;
    // ---- stage bases, per-base indicator words ---------------------------------------------------
    for (int i = tid; i < span; i += LOG_THREADS) {
        int pos = lo + i;
        int ri = pos - R.seq_start;
        uint8_t b = (ri >= 0 && ri < R.seq_len) ? bases[R.seq_off + ri] : (uint8_t)BASE_OTHER;
        sb[i] = b;
    }
    __syncthreads();
    for (int i = tid; i < span; i += LOG_THREADS) {
        uint8_t b = sb[i];
        int c = b & BASE_CODE_MASK;
        int snp = (b >> BASE_SNP_SHIFT) & 3;
        uint64_t w0 = (uint64_t)(c == BASE_A) | ((uint64_t)(c == BASE_C) << 16) | ((uint64_t)(c == BASE_G) << 32) |
                      ((uint64_t)(c == BASE_N || c == BASE_DASH) << 48);
        uint64_t w1 = (uint64_t)((b & BASE_MASKED_BIT) != 0) | ((uint64_t)(snp != 0) << 16) |
                      ((uint64_t)(snp == 2) << 32) | ((uint64_t)(snp == 1) << 48);
        int sw = 0;
        if (i > 0) {
            int cp = sb[i - 1] & BASE_CODE_MASK;
            bool gc = (c == BASE_G || c == BASE_C), gcp = (cp == BASE_G || cp == BASE_C);
            sw = gc != gcp;
        }
        uint64_t w2 = (uint64_t)sw | ((uint64_t)(c >= 4) << 16);
        W0[i] = w0; W1[i] = w1; W2[i] = w2;
    }
    __syncthreads();
    block_exclusive_scan_u64(W0, span, scratch);
    block_exclusive_scan_u64(W1, span, scratch);
    block_exclusive_scan_u64(W2, span, scratch);

    // ---- one candidate per lane --------------------------------------------------------------------
    const uint32_t row = (uint32_t)(2 * A);                    // candidates per (p, C)
    const uint32_t per_pos = row * (uint32_t)nK;
    const uint32_t n_cand = per_pos * (uint32_t)tile.np;
    const uint32_t M_row = 0xFFFFFFFFu / row + 1, M_k = 0xFFFFFFFFu / (uint32_t)nK + 1;
    const int64_t out_base = R.out_off + (int64_t)tile.p0 * per_pos;
    const double thr = P->masked_arm_threshold;

    for (uint32_t idx = tid; idx < n_cand; idx += LOG_THREADS) {
        uint32_t rc = fastdiv(idx, M_row);                     // (pl, ki)
        uint32_t in_row = idx - rc * row;
        uint32_t pl = nK == 1 ? rc : fastdiv(rc, M_k);            // the magic constant overflows for divisor 1
        uint32_t ki = rc - pl * (uint32_t)nK;
        const bool minus = in_row >= (uint32_t)A;              // strand-major inside a (position, size) row
        const int a = (int)(minus ? in_row - (uint32_t)A : in_row);
        const int e = P->arm_ext[a], l = P->arm_lig[a];
        const int C = Cmax - (int)ki * P->inc;
        const int ss = C - e - l;
        const int p = p_first + (int)pl;

        // bounds skips, mipgen.cpp:443-444
        bool valid = !(p - e <= 0 || p - l <= 0) && !(p + C - e - 1 > R.seq_stop || p + C - l - 1 > R.seq_stop) && ss > 0;
        uint64_t rec = 0;
        double score = 0.0;
        if (valid) {
            // geometry: PlusSVMipv4.cpp:9-12 / MinusSVMipv4.cpp:32-35
            const int ext_start = minus ? p + ss : p - e;
            const int lig_start = minus ? p - l : p + ss;
            const int be = ext_start - lo, bl = lig_start - lo, bi = p - lo;    // tile-local
            const uint64_t e0 = W0[be + e] - W0[be], l0 = W0[bl + l] - W0[bl];
            const uint64_t e1 = W1[be + e] - W1[be], l1 = W1[bl + l] - W1[bl];
            const bool guard = (f16(e0, 3) + f16(l0, 3)) != 0;                  // SVMipv4.cpp:63,116

            // copies, mipgen.cpp:612-613 (absent key -> 0)
            int ext_copy = 1, lig_copy = 1;
            if (R.copy_off >= 0) {
                const int se = P->len_slot[e], sl = P->len_slot[l];
                const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
                ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
                lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
            }
            // mapping flag, mipgen.cpp:615-625
            bool mapping = false;
            if (R.unmap_off >= 0 && P->check_copy_number) {
                const int ms = (minus ? lig_start : ext_start) - R.seq_start;
                if (ms >= 0 && ms < R.seq_len) mapping = unmap[R.unmap_off + (int64_t)(R.k0 + (int)ki) * R.seq_len + ms] != 0;
            }
            const uint32_t masked_n = f16(e1, 0) + f16(l1, 0);
            uint32_t flags = MIPGEN_FLAG_VALID | (guard ? MIPGEN_FLAG_GUARD : 0u);
            uint32_t snp_count = 0;
            if (mapping) {
                flags |= MIPGEN_FLAG_MAPPING;                   // early return: masking/SNP fields stay at their defaults
            } else {
                if ((double)masked_n / (double)(l + e) > thr) flags |= MIPGEN_FLAG_MASKING;     // :610,626
                snp_count = f16(e1, 1) + f16(l1, 1);
                const uint32_t snp_bad = f16(e1, 2) + f16(l1, 2), snp_ok = f16(e1, 3) + f16(l1, 3);
                if (snp_bad != 0 || snp_count > 1) flags |= MIPGEN_FLAG_SNP;                    // :690-693,759-760
                if (snp_ok != 0) flags |= MIPGEN_FLAG_HAS_SNP_MIP;
            }
            // ligation junction = first two bases of the oriented ligation arm
            int j0, j1;
            if (!minus) { j0 = sb[bl] & BASE_CODE_MASK; j1 = sb[bl + 1] & BASE_CODE_MASK; }
            else { j0 = comp_code(sb[bl + l - 1] & BASE_CODE_MASK); j1 = comp_code(sb[bl + l - 2] & BASE_CODE_MASK); }
            const uint32_t jc = (j0 < 4 && j1 < 4) ? (uint32_t)(4 * j0 + j1) : 255u;
            const uint32_t ec = (uint32_t)min(max(ext_copy, 0), 65535), lc = (uint32_t)min(max(lig_copy, 0), 65535);
            rec = (uint64_t)ec | ((uint64_t)lc << 16) | ((uint64_t)min(masked_n, 255u) << 32) |
                  ((uint64_t)min(snp_count, 255u) << 40) | ((uint64_t)flags << 48) | ((uint64_t)jc << 56);

            if (SCORE) {
                if (guard) score = -1000.0;
                else {
                    const uint64_t t0 = W0[bi + ss] - W0[bi];
                    const uint64_t t2 = W2[bi + ss] - W2[bi];
                    const uint32_t t_other = f16(t2, 1);
                    int run;
                    if (t_other == 0) run = 1 + (int)(f16(W2[bi + ss], 0) - f16(W2[bi + 1], 0));
                    else run = run_count_slow(sb, bi, ss, minus);
                    // counts on the oriented strand: revcomp swaps G<->C and A<->T
                    const uint32_t eA = f16(e0, 0), eC = f16(e0, 1), eG = f16(e0, 2);
                    const uint32_t lA = f16(l0, 0), lC = f16(l0, 1), lG = f16(l0, 2);
                    const uint32_t tA = f16(t0, 0), tC = f16(t0, 1), tG = f16(t0, 2);
                    // T = length - A - C - G - (non-ACGT characters, e.g. IUPAC codes, which count as nothing)
                    const uint32_t e_other = f16(W2[be + e] - W2[be], 1), l_other = f16(W2[bl + l] - W2[bl], 1);
                    const uint32_t eT = (uint32_t)e - eA - eC - eG - e_other;
                    const uint32_t lT = (uint32_t)l - lA - lC - lG - l_other;
                    const uint32_t tT = (uint32_t)ss - tA - tC - tG - t_other;
                    Vars x;
                    const double dl = (double)e, ll = (double)l, dn = (double)ss;
                    const double e_g = minus ? (double)eC : (double)eG, l_g = minus ? (double)lC : (double)lG,
                                 t_g = minus ? (double)tC : (double)tG;
                    const double e_a = minus ? (double)eT : (double)eA, l_a = minus ? (double)lT : (double)lA,
                                 t_a = minus ? (double)tT : (double)tA;
                    const double e_gc = (double)(eC + eG), l_gc = (double)(lC + lG), t_gc = (double)(tC + tG);
                    // contents = count * (1/len) with 1/len from the table (the reference divides; the quotients agree to 1 ulp)
                    const double re = rinv[e], rl = rinv[l], rn = ss < LOG_RINV ? rinv[ss] : 1.0 / dn;
                    x.v[MLV_BPS] = run < LOG_RINV ? dn * rinv[run] : dn / (double)run;
                    x.v[MLV_TLEN] = ss > 250 ? 250.0 : dn;
                    x.v[MLV_ELEN] = dl; x.v[MLV_LLEN] = ll;
                    x.v[MLV_EGC] = e_gc * re; x.v[MLV_LGC] = l_gc * rl; x.v[MLV_TGC] = t_gc * rn;
                    x.v[MLV_EG] = e_g * re; x.v[MLV_LG] = l_g * rl; x.v[MLV_TG] = t_g * rn;
                    x.v[MLV_EA] = e_a * re; x.v[MLV_LA] = l_a * rl; x.v[MLV_TA] = t_a * rn;
                    x.v[MLV_JS] = jc < 16 ? c_junction_scores[jc] : 0.0;
                    x.v[MLV_LEC] = log_copy_dev(HC, ext_copy);
                    x.v[MLV_LLC] = log_copy_dev(HC, lig_copy);
                    const double ex = logistic_exponent(x);
                    score = logistic_from_exponent_fast(HC, ex);
                    // b^x in [2^53, 2^54): the reference's score turns on the last bit of its pow there (kernels_logistic_dense.hip): listed for the
                    // re-score in the reference's term order with the correctly rounded power (accel_score.hip)
                    const double tb = ex * (HC->ln_base * 1.4426950408889634074);
                    if (SCORE && sat_count && tb >= 52.99 && tb <= 54.01) { const unsigned int at = atomicAdd(sat_count, 1u); if (at < sat_cap) sat_idx[at] = out_base + idx; }
                }
            }
        }
        records[out_base + idx] = rec;
        if (SCORE) scores[out_base + idx] = score;
    }
}

extern "C" size_t mipgen_logistic_lds_bytes(int span)
{
    return (size_t)(3 * (span + 1) + 8 + LOG_RINV) * sizeof(uint64_t) + (size_t)span + 16;
}

extern "C" hipError_t mipgen_launch_records_logistic(
    hipStream_t stream, int score, int n_tiles, int span_max, const DevParams* P, const DevRegion* regions,
    const LogTile* tiles, const uint8_t* bases, const int32_t* copy, const uint8_t* unmap, const HostConsts* HC, double* scores,
    uint64_t* records, int64_t* sat_idx, unsigned int* sat_count, unsigned int sat_cap)
{
    if (n_tiles <= 0) return hipSuccess;
    size_t lds = mipgen_logistic_lds_bytes(span_max);
    if (score)
        hipLaunchKernelGGL(k_records_logistic<true>, dim3(n_tiles), dim3(LOG_THREADS), lds, stream, P, regions, tiles, n_tiles,
                           bases, copy, unmap, HC, scores, records, sat_idx, sat_count, sat_cap);
    else
        hipLaunchKernelGGL(k_records_logistic<false>, dim3(n_tiles), dim3(LOG_THREADS), lds, stream, P, regions, tiles, n_tiles,
                           bases, copy, unmap, HC, scores, records, (int64_t*)nullptr, (unsigned int*)nullptr, 0u);
    return hipGetLastError();
}
