// kernels_misc.hip — sparse-candidate scorer (mixed-mode re-scores, inspection) and Featurev5::get_long_range_content.
//
// k_candidates: one workgroup per candidate.  It is the literal form of the reference arithmetic:
//   * the strand-oriented ext / lig / insert sequences are materialised in LDS (reverse_comp, MinusSVMipv4.cpp:6-29),
//   * one lane per mer counts its overlapping occurrences (SVMipv4.cpp:31-57),
//   * the 192 features are formed with the reference's own expressions (SVMipv4.cpp:72-112),
//   * lanes then own support vectors: each walks the 192 dimensions in index order (svm.cpp:329-368), takes
//     exp(-gamma*d2) and the partial sums coef*k are reduced with wavefront shuffles (svm.cpp:2511-2515).
// It doubles as an on-device cross-check of the window-separable dense kernel (kernels_svr.hip).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "common.h"
#include "device_utils.h"
#include "logistic_device.h"
#include "pow_base_cr.h"

#define CAND_THREADS 256
#define MAX_INSERT 1024

namespace {

// mer idx -> (k, codes) in the reference's lexicographic lists
__device__ __forceinline__ void arm_mer(int idx, int& k, int& x, int& y)
{
    x = idx / 5; int r = idx - 5 * x;
    if (r == 0) { k = 1; y = 0; } else { k = 2; y = r - 1; }
}
__device__ __forceinline__ void ins_mer(int idx, int& k, int& x, int& y, int& z)
{
    x = idx / 21; int r = idx - 21 * x;
    y = 0; z = 0;
    if (r == 0) { k = 1; return; }
    r -= 1; y = r / 5; int r2 = r - 5 * y;
    if (r2 == 0) { k = 2; return; }
    k = 3; z = r2 - 1;
}
// windows that START in s[0, n) and end inside s[0, nx)
__device__ int count_mer(const uint8_t* s, int n, int nx, int k, int x, int y, int z)
{
    int c = 0;
    for (int i = 0; i < n && i + k <= nx; i++) {
        bool m = s[i] == x;
        if (k >= 2) m = m && s[i + 1] == y;
        if (k >= 3) m = m && s[i + 2] == z;
        c += m;
    }
    return c;
}

// one piece of the class-switch walk of run_count_slow (logistic_device.h; SVMipv4.cpp:118-142) over oriented codes: `last` < 0 before the first base
__device__ void run_count_piece(const uint8_t* s, int n, int& last, int& run)
{
    for (int i = 0; i < n; i++) {
        const int b = s[i];
        const int c = (b == BASE_G || b == BASE_C) ? 0 : ((b == BASE_A || b == BASE_T) ? 1 : 2);
        if (last < 0) last = c;
        else if (c == 0) { if (last != 0) { run++; last = 0; } }
        else if (last != 1) { run++; last = c; }
    }
}

}  // namespace

__global__ __launch_bounds__(CAND_THREADS) void k_candidates(
    const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, const mipgen_candidate* __restrict__ cands,
    const uint8_t* __restrict__ bases, const int32_t* __restrict__ copy, const uint8_t* __restrict__ unmap,
    const HostConsts* __restrict__ HC, const double* __restrict__ model, int n_sv, double gamma, double rho, int method,
    double* __restrict__ scores, uint64_t* __restrict__ records, double* __restrict__ features,
    mipgen_candidate_ints* __restrict__ ints_out, int literal, const unsigned int* __restrict__ n_dev)
{
    if (n_dev && blockIdx.x >= *n_dev) return;          // a device-side list length (the grid is the list's capacity): no host round trip
    __shared__ double s_term[CAND_THREADS];
    __shared__ uint8_t s_ext[MIPGEN_MAX_OLIGO + 2], s_lig[MIPGEN_MAX_OLIGO + 2], s_ins[MAX_INSERT + 2];
    __shared__ uint8_t s_raw[2][MIPGEN_MAX_OLIGO + 2];     // the arms' bytes as stored (code | masked bit | SNP class), in genome order
    __shared__ int s_cnt[128];          // 0..83 insert mers, 84..103 ext mers, 104..123 lig mers
    __shared__ double s_x[MIPGEN_N_FEATURES];
    __shared__ double s_red[CAND_THREADS / WAVE];
    __shared__ int s_info[16];

    const int tid = threadIdx.x;
    const mipgen_candidate c = cands[blockIdx.x];
    const DevRegion& R = regions[c.region];
    const int p = c.scan_start, C = c.capture_size, e = c.ext_len, l = c.lig_len;
    const bool minus = c.strand != 0;
    const int ss = C - e - l;
    const bool valid = !(p - e <= 0 || p - l <= 0) && !(p + C - e - 1 > R.seq_stop || p + C - l - 1 > R.seq_stop) &&
                       ss > 0 && e <= MIPGEN_MAX_OLIGO && l <= MIPGEN_MAX_OLIGO && e >= 2 && l >= 2;
    if (!valid) {
        if (tid == 0) {
            if (scores) scores[blockIdx.x] = 0.0;
            if (records) records[blockIdx.x] = 0;
        }
        if (features) for (int j = tid; j < MIPGEN_N_FEATURES; j += CAND_THREADS) features[(int64_t)blockIdx.x * MIPGEN_N_FEATURES + j] = 0.0;
        if (ints_out && tid == 0) { mipgen_candidate_ints z = {}; ints_out[blockIdx.x] = z; }
        return;
    }
    const int ext_start = minus ? p + ss : p - e;
    const int lig_start = minus ? p - l : p + ss;
    auto base_at = [&](int pos) -> uint8_t {
        int ri = pos - R.seq_start;
        return (ri >= 0 && ri < R.seq_len) ? bases[R.seq_off + ri] : (uint8_t)BASE_OTHER;
    };
    // oriented sequences (codes); complement + reverse on '-'
    for (int i = tid; i < e; i += CAND_THREADS) {
        const int off = minus ? e - 1 - i : i;
        const uint8_t raw = base_at(ext_start + off);
        s_raw[0][off] = raw;
        const int b = raw & BASE_CODE_MASK;
        s_ext[i] = (uint8_t)(minus ? comp_code(b) : b);
    }
    for (int i = tid; i < l; i += CAND_THREADS) {
        const int off = minus ? l - 1 - i : i;
        const uint8_t raw = base_at(lig_start + off);
        s_raw[1][off] = raw;
        const int b = raw & BASE_CODE_MASK;
        s_lig[i] = (uint8_t)(minus ? comp_code(b) : b);
    }
    __syncthreads();
    // integer record fields (design_mip, mipgen.cpp:606-760)
    if (tid == 0) {
        int masked_n = 0, snp_any = 0, snp_bad = 0, snp_ok = 0, bad = 0;
        for (int arm = 0; arm < 2; arm++) {
            const int n = arm == 0 ? e : l;
            for (int i = 0; i < n; i++) {
                const uint8_t b = s_raw[arm][i];
                int code = b & BASE_CODE_MASK, snp = (b >> BASE_SNP_SHIFT) & 3;
                masked_n += (b & BASE_MASKED_BIT) != 0;
                snp_any += snp != 0; snp_bad += snp == 2; snp_ok += snp == 1;
                bad += (code == BASE_N || code == BASE_DASH);
            }
        }
        int ext_copy = 1, lig_copy = 1;
        if (R.copy_off >= 0) {
            const int se = P->len_slot[e], sl = P->len_slot[l];
            const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
            ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
            lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
        }
        bool mapping = false;
        const int k = (P->max_capture - C) / P->inc;
        if (R.unmap_off >= 0 && P->check_copy_number && k >= 0 && k < P->n_sizes_all) {
            const int ms = (minus ? lig_start : ext_start) - R.seq_start;
            if (ms >= 0 && ms < R.seq_len) mapping = unmap[R.unmap_off + (int64_t)k * R.seq_len + ms] != 0;
        }
        uint32_t flags = MIPGEN_FLAG_VALID | (bad ? MIPGEN_FLAG_GUARD : 0u);
        int snp_count = 0;
        if (mapping) flags |= MIPGEN_FLAG_MAPPING;
        else {
            if ((double)masked_n / (double)(l + e) > P->masked_arm_threshold) flags |= MIPGEN_FLAG_MASKING;
            snp_count = snp_any;
            if (snp_bad != 0 || snp_count > 1) flags |= MIPGEN_FLAG_SNP;
            if (snp_ok != 0) flags |= MIPGEN_FLAG_HAS_SNP_MIP;
        }
        s_info[0] = ext_copy; s_info[1] = lig_copy; s_info[2] = masked_n; s_info[3] = snp_count; s_info[4] = (int)flags;
        s_info[5] = bad;
    }
    __syncthreads();
    const int ext_copy = s_info[0], lig_copy = s_info[1];
    const bool guard = s_info[5] != 0;
    const int j0 = s_lig[0], j1 = s_lig[1];
    const int jc = (j0 < 4 && j1 < 4) ? 4 * j0 + j1 : 255;

    // mer counts: one lane per mer.  The oriented insert passes through LDS in pieces of MAX_INSERT bases (+ two of look-ahead for the 2- / 3-mers
    // that start in a piece): one piece for every realistic capture size, no limit on the others; the class-switch walk (SVMipv4.cpp:118-142) of the
    // logistic score carries its state across the pieces
    const bool want_run = ints_out || method == MIPGEN_SCORE_LOGISTIC;
    int ins_count = 0, run_last = -1, run_n = 0;
    for (int c0 = 0; c0 < ss; c0 += MAX_INSERT) {
        const int len = min(MAX_INSERT, ss - c0), lenx = min(len + 2, ss - c0);
        if (c0) __syncthreads();
        for (int i = tid; i < lenx; i += CAND_THREADS) {
            const int b = base_at(minus ? p + ss - 1 - (c0 + i) : p + c0 + i) & BASE_CODE_MASK;
            s_ins[i] = (uint8_t)(minus ? comp_code(b) : b);
        }
        __syncthreads();
        if (tid < 84) { int k, x, y, z; ins_mer(tid, k, x, y, z); ins_count += count_mer(s_ins, len, lenx, k, x, y, z); }
        if (tid == 0 && want_run) run_count_piece(s_ins, len, run_last, run_n);   // (a serial walk: only where it is used)
    }
    if (tid < 84) s_cnt[tid] = ins_count;
    else if (tid < 104) { int k, x, y; arm_mer(tid - 84, k, x, y); s_cnt[tid] = count_mer(s_ext, e, e, k, x, y, 0); }
    else if (tid < 124) { int k, x, y; arm_mer(tid - 104, k, x, y); s_cnt[tid] = count_mer(s_lig, l, l, k, x, y, 0); }
    __syncthreads();

    // 192 features, SVMipv4.cpp:72-112
    if (tid < MIPGEN_N_FEATURES) {
        double v;
        const int f = tid;
        if (guard) v = 0.0;
        else if (f < F_LRC) {                         // ext block
            if (f == F_EXT_LEN) v = (double)e;
            else if (f == F_EXT_GC) v = ((double)s_cnt[84 + 10] + (double)s_cnt[84 + 5]) / (double)(uint64_t)((uint64_t)e - 1 + 1);
            else { int idx = f < F_EXT_GC ? f : f - 1; int k = (idx % 5) ? 2 : 1; v = (double)s_cnt[84 + idx] / ((double)(uint64_t)((uint64_t)e - k) + 1.); }
        } else if (f < F_INS) v = R.lrc[f - F_LRC];
        else if (f < F_LIG) {                         // insert block
            const int g = f - F_INS;
            if (f == F_INS_LEN) v = (double)ss;
            else if (f == F_INS_GC) v = ((double)s_cnt[42] + (double)s_cnt[21]) / ((double)(uint64_t)((uint64_t)ss - 1) + 1.);
            else {
                int idx = g < 63 ? g : g - 1;
                int r = idx % 21; int k = r == 0 ? 1 : (((r - 1) % 5) == 0 ? 2 : 3);
                v = (double)s_cnt[idx] / ((double)(uint64_t)((uint64_t)ss - k) + 1.);
            }
        } else if (f < F_JUNC) {                      // lig block
            const int g = f - F_LIG;
            if (f == F_LIG_LEN) v = (double)l;
            else if (f == F_LIG_GC) v = ((double)s_cnt[104 + 10] + (double)s_cnt[104 + 5]) / ((double)(uint64_t)((uint64_t)l - 1) + 1.);
            else { int idx = g < 15 ? g : g - 1; int k = (idx % 5) ? 2 : 1; v = (double)s_cnt[104 + idx] / ((double)(uint64_t)((uint64_t)l - k) + 1.); }
        } else if (f < F_LEC) v = (jc == f - F_JUNC) ? 1.0 : 0.0;
        else v = log_copy_dev(HC, f == F_LEC ? ext_copy : lig_copy);
        s_x[f] = v;
        if (features) features[(int64_t)blockIdx.x * MIPGEN_N_FEATURES + f] = v;
    }
    __syncthreads();

    // integer features + logistic score
    double logistic = 0.0;
    if (tid == 0) {
        const int eA = s_cnt[84 + 0], eC = s_cnt[84 + 5], eG = s_cnt[84 + 10], eT = s_cnt[84 + 15];
        const int lA = s_cnt[104 + 0], lC = s_cnt[104 + 5], lG = s_cnt[104 + 10], lT = s_cnt[104 + 15];
        const int tA = s_cnt[0], tC = s_cnt[21], tG = s_cnt[42], tT = s_cnt[63];
        const int run = want_run ? run_n + 1 : 1;
        if (ints_out) {
            mipgen_candidate_ints o;
            o.ext_a = eA; o.ext_c = eC; o.ext_g = eG; o.ext_t = eT;
            o.lig_a = lA; o.lig_c = lC; o.lig_g = lG; o.lig_t = lT;
            o.ins_a = tA; o.ins_c = tC; o.ins_g = tG; o.ins_t = tT;
            o.run_count = run; o.junction = jc; o.ext_copy = ext_copy; o.lig_copy = lig_copy;
            o.masked_n = s_info[2]; o.snp_count = s_info[3]; o.flags = s_info[4]; o.scan_size = ss;
            ints_out[blockIdx.x] = o;
        }
        if (records) {
            const uint32_t ec = (uint32_t)min(max(ext_copy, 0), 65535), lc = (uint32_t)min(max(lig_copy, 0), 65535);
            records[blockIdx.x] = (uint64_t)ec | ((uint64_t)lc << 16) | ((uint64_t)min(s_info[2], 255) << 32) |
                                  ((uint64_t)min(s_info[3], 255) << 40) | ((uint64_t)(uint32_t)s_info[4] << 48) | ((uint64_t)(uint32_t)jc << 56);
        }
        if (method == MIPGEN_SCORE_LOGISTIC) {
            if (guard) logistic = -1000.0;
            else {
                Vars x;
                const double dl = (double)e, ll = (double)l, dn = (double)ss;
                x.v[MLV_BPS] = dn / (double)run;
                x.v[MLV_TLEN] = ss > 250 ? 250.0 : dn;
                x.v[MLV_ELEN] = dl; x.v[MLV_LLEN] = ll;
                x.v[MLV_EGC] = ((double)eC + (double)eG) / dl; x.v[MLV_LGC] = ((double)lC + (double)lG) / ll; x.v[MLV_TGC] = ((double)tC + (double)tG) / dn;
                x.v[MLV_EG] = (double)eG / dl; x.v[MLV_LG] = (double)lG / ll; x.v[MLV_TG] = (double)tG / dn;
                x.v[MLV_EA] = (double)eA / dl; x.v[MLV_LA] = (double)lA / ll; x.v[MLV_TA] = (double)tA / dn;
                x.v[MLV_JS] = jc < 16 ? c_junction_scores[jc] : 0.0;
                x.v[MLV_LEC] = log_copy_dev(HC, ext_copy); x.v[MLV_LLC] = log_copy_dev(HC, lig_copy);
                const double ex = logistic_exponent_exact(x);                     // every operation rounded on its own, as the reference's
                // SVMipv4.cpp:247: pow(2.71828, ex), correctly rounded (pow_base_cr.h: glibc's own double in 99.9 % of cases; the device math library's pow in 91 %)
                const double y = fabs(ex) < 700.0 ? pow_base_cr(ex) : pow(MIPGEN_LOGISTIC_BASE, ex);
                logistic = y / (1.0 + y);
            }
            if (scores) scores[blockIdx.x] = logistic;
        }
    }
    if (method != MIPGEN_SCORE_SVR || !scores) return;

    if (literal) {
        // the reference's own operation order (svm.cpp:329-368 k_function: d = x - y, sum += d * d in index order; :2511-2515 svm_predict_values:
        // sum += coef * k in support-vector order, then - rho), every operation rounded on its own (g++ on x86-64 does not contract to FMA):
        // what mipgen_accel prints for the scores that sit on a rounding boundary of the 6 printed digits (accel_score.hip: fix_print_boundaries).
        // Plain operators under `fp contract(off)`: HIP's __dmul_rn / __dadd_rn are plain operators compiled under contract(fast) - inlined, hipcc
        // fuses them into FMAs whatever the caller says.
#pragma clang fp contract(off)
        double total = 0.0;
        for (int base = 0; base < n_sv; base += CAND_THREADS) {
            const int i = base + tid;
            double term = 0.0;
            if (i < n_sv) {
                const double* sv = model + (int64_t)i * SV_ROW;
                double sum = 0.0;
                for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
                    const double d = s_x[j] - sv[j];
                    const double dd = d * d;
                    sum = sum + dd;
                }
                sum = sum + sv[SVR_N_EXTRA];
                const double arg = -gamma * sum;
                const double k = exp(arg);
                term = sv[SVR_COEF] * k;
            }
            s_term[tid] = term;
            __syncthreads();
            if (tid == 0) { const int m = min(CAND_THREADS, n_sv - base); for (int q = 0; q < m; q++) total = total + s_term[q]; }
            __syncthreads();
        }
        if (tid == 0) scores[blockIdx.x] = total - rho;
        return;
    }
    // SVR: lanes own support vectors; 192-dimension walk in index order, then shuffle reduction
    double part = 0.0;
    for (int i = tid; i < n_sv; i += CAND_THREADS) {
        const double* sv = model + (int64_t)i * SV_ROW;
        double sum = 0.0;
        for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
            const double d = s_x[j] - sv[j];
            sum += d * d;
        }
        sum += sv[SVR_N_EXTRA];
        part += sv[SVR_COEF] * exp(-gamma * sum);
    }
    part = wave_sum_f64(part);
    if ((tid & (WAVE - 1)) == 0) s_red[tid / WAVE] = part;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < CAND_THREADS / WAVE; w++) s += s_red[w];
        scores[blockIdx.x] = s - rho;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_features_batch: the 192 features + the integer record of every candidate of a LIST, one WAVEFRONT per candidate (the list scorer's
// front end: kernels_svr_gemm.hip).  Same values as k_candidates (SVMipv4.cpp:60-113, mipgen.cpp:606-760) - the mer counts come from an LDS
// histogram filled by one pass over the oriented sequences (a lane per start position, three atomic increments) instead of one lane per
// mer walking the whole sequence, the record's counts from wave ballots over the arms' bytes.
// ---------------------------------------------------------------------------------------------------------
#define FB_WAVES 4

__global__ __launch_bounds__(FB_WAVES * 64) void k_features_batch(
    int n, const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, const mipgen_candidate* __restrict__ cands,
    const uint8_t* __restrict__ bases, const int32_t* __restrict__ copy, const uint8_t* __restrict__ unmap, const HostConsts* __restrict__ HC,
    uint64_t* __restrict__ records, double* __restrict__ features)
{
    __shared__ uint8_t s_ext_a[FB_WAVES][MIPGEN_MAX_OLIGO + 2], s_lig_a[FB_WAVES][MIPGEN_MAX_OLIGO + 2], s_ins_a[FB_WAVES][MAX_INSERT + 2];
    __shared__ int s_cnt_a[FB_WAVES][128];          // 0..83 insert mers, 84..103 ext mers, 104..123 lig mers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ci = blockIdx.x * FB_WAVES + wave;
    if (ci >= n) return;
    uint8_t* s_ext = s_ext_a[wave]; uint8_t* s_lig = s_lig_a[wave]; uint8_t* s_ins = s_ins_a[wave];
    int* s_cnt = s_cnt_a[wave];
    const mipgen_candidate c = cands[ci];
    const DevRegion& R = regions[c.region];
    const int p = c.scan_start, C = c.capture_size, e = c.ext_len, l = c.lig_len;
    const bool minus = c.strand != 0;
    const int ss = C - e - l;
    const bool valid = !(p - e <= 0 || p - l <= 0) && !(p + C - e - 1 > R.seq_stop || p + C - l - 1 > R.seq_stop) &&
                       ss > 0 && e <= MIPGEN_MAX_OLIGO && l <= MIPGEN_MAX_OLIGO && e >= 2 && l >= 2;
    double* fo = features + (int64_t)ci * MIPGEN_N_FEATURES;
    if (!valid) {
        if (lane == 0) records[ci] = 0;
        for (int j = lane; j < MIPGEN_N_FEATURES; j += 64) fo[j] = 0.0;
        return;
    }
    const int ext_start = minus ? p + ss : p - e;
    const int lig_start = minus ? p - l : p + ss;
    auto base_at = [&](int pos) -> uint8_t {
        const int ri = pos - R.seq_start;
        return (ri >= 0 && ri < R.seq_len) ? bases[R.seq_off + ri] : (uint8_t)BASE_OTHER;
    };
    s_cnt[lane] = 0; s_cnt[lane + 64] = 0;
    // arms: lane i holds byte i of the arm in genome order; oriented codes go to LDS
    const uint8_t raw_e = lane < e ? base_at(ext_start + lane) : (uint8_t)0, raw_l = lane < l ? base_at(lig_start + lane) : (uint8_t)0;
    if (lane < e) { const int b = raw_e & BASE_CODE_MASK; s_ext[minus ? e - 1 - lane : lane] = (uint8_t)(minus ? comp_code(b) : b); }
    if (lane < l) { const int b = raw_l & BASE_CODE_MASK; s_lig[minus ? l - 1 - lane : lane] = (uint8_t)(minus ? comp_code(b) : b); }
    // integer record fields (design_mip, mipgen.cpp:606-760): counts over the arms' bytes by wave ballots
    const bool in_e = lane < e, in_l = lane < l;
    auto cnt2 = [&](bool pe, bool pl) -> int { return __builtin_popcountll(__ballot(in_e && pe)) + __builtin_popcountll(__ballot(in_l && pl)); };
    const int snp_e = (raw_e >> BASE_SNP_SHIFT) & 3, snp_l = (raw_l >> BASE_SNP_SHIFT) & 3;
    const int code_e = raw_e & BASE_CODE_MASK, code_l = raw_l & BASE_CODE_MASK;
    const int masked_n = cnt2((raw_e & BASE_MASKED_BIT) != 0, (raw_l & BASE_MASKED_BIT) != 0);
    const int snp_any = cnt2(snp_e != 0, snp_l != 0), snp_bad = cnt2(snp_e == 2, snp_l == 2), snp_ok = cnt2(snp_e == 1, snp_l == 1);
    const int bad = cnt2(code_e == BASE_N || code_e == BASE_DASH, code_l == BASE_N || code_l == BASE_DASH);
    int ext_copy = 1, lig_copy = 1;
    if (R.copy_off >= 0) {
        const int se = P->len_slot[e], sl = P->len_slot[l];
        const int ie = ext_start - R.seq_start, il = lig_start - R.seq_start;
        ext_copy = (se >= 0 && ie >= 0 && ie < R.seq_len) ? copy[R.copy_off + (int64_t)se * R.seq_len + ie] : 0;
        lig_copy = (sl >= 0 && il >= 0 && il < R.seq_len) ? copy[R.copy_off + (int64_t)sl * R.seq_len + il] : 0;
    }
    bool mapping = false;
    {
        const int k = (P->max_capture - C) / P->inc;
        if (R.unmap_off >= 0 && P->check_copy_number && k >= 0 && k < P->n_sizes_all) {
            const int ms = (minus ? lig_start : ext_start) - R.seq_start;
            if (ms >= 0 && ms < R.seq_len) mapping = unmap[R.unmap_off + (int64_t)k * R.seq_len + ms] != 0;
        }
    }
    uint32_t flags = MIPGEN_FLAG_VALID | (bad ? MIPGEN_FLAG_GUARD : 0u);
    int snp_count = 0;
    if (mapping) flags |= MIPGEN_FLAG_MAPPING;
    else {
        if ((double)masked_n / (double)(l + e) > P->masked_arm_threshold) flags |= MIPGEN_FLAG_MASKING;
        snp_count = snp_any;
        if (snp_bad != 0 || snp_count > 1) flags |= MIPGEN_FLAG_SNP;
        if (snp_ok != 0) flags |= MIPGEN_FLAG_HAS_SNP_MIP;
    }
    const bool guard = bad != 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int j0 = s_lig[0], j1 = s_lig[1];
    const int jc = (j0 < 4 && j1 < 4) ? 4 * j0 + j1 : 255;
    if (lane == 0) {
        const uint32_t ec = (uint32_t)min(max(ext_copy, 0), 65535), lc = (uint32_t)min(max(lig_copy, 0), 65535);
        records[ci] = (uint64_t)ec | ((uint64_t)lc << 16) | ((uint64_t)min(masked_n, 255) << 32) | ((uint64_t)min(snp_count, 255) << 40) |
                      ((uint64_t)flags << 48) | ((uint64_t)(uint32_t)jc << 56);
    }
    // mer histogram: a lane per window start; windows touching a non-ACGT code count nowhere (count_mer compares codes 0..3).  The oriented insert
    // passes through LDS in pieces of MAX_INSERT bases + two of look-ahead (one piece for every realistic capture size)
    for (int c0 = 0; c0 < ss; c0 += MAX_INSERT) {
        const int len = min(MAX_INSERT, ss - c0), lenx = min(len + 2, ss - c0);
        if (c0) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        for (int i = lane; i < lenx; i += 64) {
            const int b = base_at(minus ? p + ss - 1 - (c0 + i) : p + c0 + i) & BASE_CODE_MASK;
            s_ins[i] = (uint8_t)(minus ? comp_code(b) : b);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < len; i += 64) {
            const int x = s_ins[i];
            if (x < 4) {
                atomicAdd(&s_cnt[21 * x], 1);
                if (i + 1 < lenx) {
                    const int y = s_ins[i + 1];
                    if (y < 4) {
                        atomicAdd(&s_cnt[21 * x + 1 + 5 * y], 1);
                        if (i + 2 < lenx) { const int z = s_ins[i + 2]; if (z < 4) atomicAdd(&s_cnt[21 * x + 1 + 5 * y + 1 + z], 1); }
                    }
                }
            }
        }
    }
    if (lane < e) {
        const int x = s_ext[lane];
        if (x < 4) { atomicAdd(&s_cnt[84 + 5 * x], 1); if (lane + 1 < e) { const int y = s_ext[lane + 1]; if (y < 4) atomicAdd(&s_cnt[84 + 5 * x + 1 + y], 1); } }
    }
    if (lane < l) {
        const int x = s_lig[lane];
        if (x < 4) { atomicAdd(&s_cnt[104 + 5 * x], 1); if (lane + 1 < l) { const int y = s_lig[lane + 1]; if (y < 4) atomicAdd(&s_cnt[104 + 5 * x + 1 + y], 1); } }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 192 features, SVMipv4.cpp:72-112 (the arithmetic of k_candidates, three features per lane)
    for (int f = lane; f < MIPGEN_N_FEATURES; f += 64) {
        double v;
        if (guard) v = 0.0;
        else if (f < F_LRC) {                         // ext block
            if (f == F_EXT_LEN) v = (double)e;
            else if (f == F_EXT_GC) v = ((double)s_cnt[84 + 10] + (double)s_cnt[84 + 5]) / (double)(uint64_t)((uint64_t)e - 1 + 1);
            else { int idx = f < F_EXT_GC ? f : f - 1; int k = (idx % 5) ? 2 : 1; v = (double)s_cnt[84 + idx] / ((double)(uint64_t)((uint64_t)e - k) + 1.); }
        } else if (f < F_INS) v = R.lrc[f - F_LRC];
        else if (f < F_LIG) {                         // insert block
            const int g = f - F_INS;
            if (f == F_INS_LEN) v = (double)ss;
            else if (f == F_INS_GC) v = ((double)s_cnt[42] + (double)s_cnt[21]) / ((double)(uint64_t)((uint64_t)ss - 1) + 1.);
            else {
                int idx = g < 63 ? g : g - 1;
                int r = idx % 21; int k = r == 0 ? 1 : (((r - 1) % 5) == 0 ? 2 : 3);
                v = (double)s_cnt[idx] / ((double)(uint64_t)((uint64_t)ss - k) + 1.);
            }
        } else if (f < F_JUNC) {                      // lig block
            const int g = f - F_LIG;
            if (f == F_LIG_LEN) v = (double)l;
            else if (f == F_LIG_GC) v = ((double)s_cnt[104 + 10] + (double)s_cnt[104 + 5]) / ((double)(uint64_t)((uint64_t)l - 1) + 1.);
            else { int idx = g < 15 ? g : g - 1; int k = (idx % 5) ? 2 : 1; v = (double)s_cnt[104 + idx] / ((double)(uint64_t)((uint64_t)l - k) + 1.); }
        } else if (f < F_LEC) v = (jc == f - F_JUNC) ? 1.0 : 0.0;
        else v = log_copy_dev(HC, f == F_LEC ? ext_copy : lig_copy);
        fo[f] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The slow SVR route of the dense grid (parameter sets the tiled kernel of kernels_svr.hip cannot take: scan sizes below 3, more than
// 240 arm pairs, a tile beyond 160 KiB of LDS - the reference accepts any -arm_lengths / -capture_increment / range, mipgen.cpp:222-261,
// 427-444): the candidates of a chunk of the window's dense index range are written out as a LIST (k_dense_candidates) and go through the
// list scorer (k_features_batch + k_svr_gemm); k_dense_list_fix then applies the constants of the dense kernel's own fix-ups from the
// integer records, so that both routes hand out identical values for the flagged candidates.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_candidates(const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, int r0, int r1, int64_t c0, int n,
                                                          mipgen_candidate* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t idx = c0 + i;                                    // window-relative dense index (DevRegion::out_off is window-relative too)
    int lo = r0, hi = r1;                                          // the last region with out_off <= idx: empty regions share the offset of the
    while (hi - lo > 1) {                                          // next non-empty one and come before it
        const int mid = (lo + hi) >> 1;
        if (regions[mid].out_off <= idx) lo = mid; else hi = mid;
    }
    const DevRegion& R = regions[lo];
    const int64_t rel = idx - R.out_off;
    const int A = P->n_pairs;
    const int a = (int)(rel % A);
    const int64_t row = rel / A;
    const int strand = (int)(row & 1);
    const int64_t rest = row >> 1;
    const int k = (int)(rest % R.n_sizes), pi = (int)(rest / R.n_sizes);
    mipgen_candidate c;
    c.region = lo; c.scan_start = R.first_pos + pi; c.capture_size = P->max_capture - (R.k0 + k) * P->inc;
    c.ext_len = P->arm_ext[a]; c.lig_len = P->arm_lig[a]; c.strand = strand;
    out[i] = c;
}
__global__ __launch_bounds__(256) void k_dense_list_fix(int n, const uint64_t* __restrict__ records, double rho, double s_guard, double* __restrict__ scores)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t rec = records[i];
    const uint32_t flags = MIPGEN_REC_FLAGS(rec);
    if (!(flags & MIPGEN_FLAG_VALID)) scores[i] = 0.0;
    else if (flags & MIPGEN_FLAG_GUARD) scores[i] = s_guard;           // all-zero feature vector (SVMipv4.cpp:63-68)
    else if (MIPGEN_REC_EXT_COPY(rec) == 0 || MIPGEN_REC_LIG_COPY(rec) == 0) scores[i] = -rho;   // log10(0) = -inf: every kernel value is 0
}
extern "C" hipError_t mipgen_launch_dense_candidates(hipStream_t stream, const DevParams* P, const DevRegion* regions, int r0, int r1, int64_t c0, int n,
                                                     mipgen_candidate* out)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dense_candidates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, regions, r0, r1, c0, n, out);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_dense_list_fix(hipStream_t stream, int n, const uint64_t* records, double rho, double s_guard, double* scores)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dense_list_fix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, records, rho, s_guard, scores);
    return hipGetLastError();
}

extern "C" hipError_t mipgen_launch_features_batch(hipStream_t stream, int n, const DevParams* P, const DevRegion* regions, const mipgen_candidate* cands,
                                                   const uint8_t* bases, const int32_t* copy, const uint8_t* unmap, const HostConsts* HC, uint64_t* records,
                                                   double* features)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_features_batch, dim3((n + FB_WAVES - 1) / FB_WAVES), dim3(FB_WAVES * 64), 0, stream, n, P, regions, cands, bases, copy, unmap, HC, records,
                       features);
    return hipGetLastError();
}

extern "C" hipError_t mipgen_launch_candidates(
    hipStream_t stream, int n, const DevParams* P, const DevRegion* regions, const mipgen_candidate* cands, const uint8_t* bases,
    const int32_t* copy, const uint8_t* unmap, const HostConsts* HC, const double* model, int n_sv, double gamma, double rho,
    int method, double* scores, uint64_t* records, double* features, mipgen_candidate_ints* ints, int literal, const unsigned int* n_dev)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_candidates, dim3(n), dim3(CAND_THREADS), 0, stream, P, regions, cands, bases, copy, unmap, HC, model,
                       n_sv, gamma, rho, method, scores, records, features, ints, literal, n_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Scores on a rounding boundary of the printed digits.  The front end prints scores with 6 significant digits (mipgen.cpp:774: default
// ostream precision); a dense SVR score differs from the reference's double by ~1e-12 (another summation order, table factors), so a
// score within that distance of a midpoint between two 6-digit numbers could print another last digit.  k_print_boundary_scan lists the
// dense-grid candidates of a window whose score lies within tol of such a midpoint (guard / zero-copy candidates carry exact constants and
// are skipped); they are re-scored by k_candidates in the reference's own operation order and written back (k_scatter_scores).
// ---------------------------------------------------------------------------------------------------------

// distance of |s| to the nearest midpoint between two 6-significant-digit decimal numbers (what "%g" / the default ostream precision prints,
// mipgen.cpp:774), relative test against tol: decade from the binary exponent + a table for 1e-13 .. 1e13, log10 / pow outside
__device__ __forceinline__ bool near_print_midpoint(double s, double tol_rel, double tol_abs)
{
    const double a = fabs(s);
    if (!(a > 1e-300) || !(a < 1e300)) return false;                        // 0, NaN, inf
    double unit;
    if (a >= 1e-8 && a < 1e8) {
        constexpr double P10[32] = {1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e0,
                                    1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16};
        const int e2 = (int)((__double2hiint(a) >> 20) & 0x7FF) - 1023;
        int e10 = (e2 * 1233) >> 12;                                        // ~ floor(e2 * log10(2)), off by at most one
        if (a < P10[e10 + 15]) e10--; else if (a >= P10[e10 + 16]) e10++;
        unit = P10[e10 + 10];                                               // 10^(e10 - 5)
    } else {
        int e10 = (int)floor(log10(a));
        unit = pow(10.0, (double)(e10 - 5));
        const double t0 = a / unit;
        if (t0 >= 1e6) unit *= 10.0; else if (t0 < 1e5) unit *= 0.1;        // log10 rounding at powers of ten
    }
    const double t = a / unit;
    const double frac = t - floor(t);
    return fabs(frac - 0.5) * unit <= tol_rel * a + tol_abs;
}

__global__ __launch_bounds__(256) void k_print_boundary_scan(const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, int r0, int r1,
                                                             const double* __restrict__ scores, const uint64_t* __restrict__ records, int64_t n,
                                                             double tol_rel, double tol_abs, mipgen_candidate* __restrict__ out, int64_t* __restrict__ out_idx,
                                                             unsigned int* __restrict__ count, unsigned int cap)
{
    const int A = P->n_pairs;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        if (!near_print_midpoint(scores[idx], tol_rel, tol_abs)) continue;
        const uint64_t rec = records[idx];
        const uint32_t flags = MIPGEN_REC_FLAGS(rec);
        if (!(flags & MIPGEN_FLAG_VALID) || (flags & MIPGEN_FLAG_GUARD) || MIPGEN_REC_EXT_COPY(rec) == 0 || MIPGEN_REC_LIG_COPY(rec) == 0) continue;
        int lo = r0, hi = r1 - 1;                                           // the region of the candidate: the last one that starts at or before idx
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (regions[mid].out_off <= idx) lo = mid; else hi = mid - 1; }
        const DevRegion& R = regions[lo];
        const int64_t local = idx - R.out_off;
        const int a_i = (int)(local % A);
        const int64_t row = local / A;
        const int strand = (int)(row & 1);
        const int64_t rest = row >> 1;
        const int ki = (int)(rest % R.n_sizes);
        const int pi = (int)(rest / R.n_sizes);
        if (pi >= R.n_pos) continue;                                        // (empty regions share an offset with their successor)
        const unsigned int at = atomicAdd(count, 1u);
        if (at < cap) {
            mipgen_candidate c;
            c.region = lo; c.scan_start = R.first_pos + pi; c.capture_size = P->max_capture - (R.k0 + ki) * P->inc;
            c.ext_len = P->arm_ext[a_i]; c.lig_len = P->arm_lig[a_i]; c.strand = strand;
            out[at] = c; out_idx[at] = idx;
        }
    }
}

// dense indices of a window -> candidates (the decode of k_print_boundary_scan for a ready-made index list: the saturated logistic candidates the dense kernel listed)
__global__ __launch_bounds__(256) void k_index_candidates(const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, int r0, int r1, const int64_t* __restrict__ idx_list,
                                                          const unsigned int* __restrict__ count, unsigned int cap, mipgen_candidate* __restrict__ out)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || i >= *count) return;
    const int64_t idx = idx_list[i];
    const int A = P->n_pairs;
    int lo = r0, hi = r1 - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (regions[mid].out_off <= idx) lo = mid; else hi = mid - 1; }
    const DevRegion& R = regions[lo];
    const int64_t local = idx - R.out_off;
    const int a_i = (int)(local % A);
    const int64_t row = local / A;
    const int64_t rest = row >> 1;
    mipgen_candidate c;
    c.region = lo; c.scan_start = R.first_pos + (int)(rest / R.n_sizes); c.capture_size = P->max_capture - (R.k0 + (int)(rest % R.n_sizes)) * P->inc;
    c.ext_len = P->arm_ext[a_i]; c.lig_len = P->arm_lig[a_i]; c.strand = (int)(row & 1);
    out[i] = c;
}
extern "C" hipError_t mipgen_launch_index_candidates(hipStream_t stream, const DevParams* P, const DevRegion* regions, int r0, int r1, const int64_t* idx, const unsigned int* count,
                                                     unsigned int cap, mipgen_candidate* out)
{
    if (cap == 0) return hipSuccess;
    hipLaunchKernelGGL(k_index_candidates, dim3((cap + 255) / 256), dim3(256), 0, stream, P, regions, r0, r1, idx, count, cap, out);
    return hipGetLastError();
}

// `over` (host-mapped word, may be null): set when the scan listed more entries than the list holds - the surplus was not re-scored, the caller reports it
__global__ __launch_bounds__(256) void k_scatter_scores(const double* __restrict__ src, const int64_t* __restrict__ idx, int cap, const unsigned int* __restrict__ n_dev,
                                                        double* __restrict__ scores, unsigned int* __restrict__ over)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap && (unsigned int)i < *n_dev) scores[idx[i]] = src[i];
    if (i == 0 && over && *n_dev > (unsigned int)cap) *over = *n_dev - (unsigned int)cap;
}

// the same test over a candidate LIST (mixed designs: the survivors re-scored by the matrix-core scorer): entry i of the list
__global__ __launch_bounds__(256) void k_print_boundary_scan_list(const mipgen_candidate* __restrict__ cands, const double* __restrict__ scores,
                                                                  const uint64_t* __restrict__ records, int n, double tol_rel, double tol_abs,
                                                                  mipgen_candidate* __restrict__ out, int64_t* __restrict__ out_idx,
                                                                  unsigned int* __restrict__ count, unsigned int cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!near_print_midpoint(scores[i], tol_rel, tol_abs)) return;
    const uint64_t rec = records[i];
    const uint32_t flags = MIPGEN_REC_FLAGS(rec);
    if (!(flags & MIPGEN_FLAG_VALID) || (flags & MIPGEN_FLAG_GUARD) || MIPGEN_REC_EXT_COPY(rec) == 0 || MIPGEN_REC_LIG_COPY(rec) == 0) return;
    const unsigned int at = atomicAdd(count, 1u);
    if (at < cap) { out[at] = cands[i]; out_idx[at] = i; }
}

extern "C" hipError_t mipgen_launch_print_boundary_scan(hipStream_t stream, const DevParams* P, const DevRegion* regions, int r0, int r1, const double* scores,
                                                        const uint64_t* records, int64_t n, double tol_rel, double tol_abs, mipgen_candidate* out,
                                                        int64_t* out_idx, unsigned int* count, unsigned int cap, int n_cu)
{
    if (n <= 0 || r1 <= r0) return hipSuccess;
    const int64_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)std::min<int64_t>(want, (int64_t)std::max(n_cu, 1) * 16);
    hipLaunchKernelGGL(k_print_boundary_scan, dim3(grid), dim3(256), 0, stream, P, regions, r0, r1, scores, records, n, tol_rel, tol_abs, out, out_idx, count, cap);
    return hipGetLastError();
}

// the same test over the condensed survivors of a window (silent designs print nothing else): entry i of the survivor array; the re-scored
// value goes back into the survivor's score field
__global__ __launch_bounds__(256) void k_print_boundary_scan_surv(const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, int r0, int r1,
                                                                  const mipgen_survivor* __restrict__ surv, int64_t n, int64_t cand0, double tol_rel, double tol_abs,
                                                                  mipgen_candidate* __restrict__ out, int64_t* __restrict__ out_idx,
                                                                  unsigned int* __restrict__ count, unsigned int cap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mipgen_survivor sv = surv[i];
    if (sv.cand_index < 0 || !near_print_midpoint(sv.score, tol_rel, tol_abs)) return;
    const uint32_t flags = MIPGEN_REC_FLAGS(sv.record);
    if (!(flags & MIPGEN_FLAG_VALID) || (flags & MIPGEN_FLAG_GUARD) || MIPGEN_REC_EXT_COPY(sv.record) == 0 || MIPGEN_REC_LIG_COPY(sv.record) == 0) return;
    const int64_t idx = sv.cand_index - cand0;                             // window-relative, like DevRegion::out_off
    const int A = P->n_pairs;
    int lo = r0, hi = r1 - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (regions[mid].out_off <= idx) lo = mid; else hi = mid - 1; }
    const DevRegion& R = regions[lo];
    const int64_t local = idx - R.out_off;
    const int a_i = (int)(local % A);
    const int64_t row = local / A;
    const int64_t rest = row >> 1;
    const int ki = (int)(rest % R.n_sizes), pi = (int)(rest / R.n_sizes);
    if (pi >= R.n_pos) return;
    const unsigned int at = atomicAdd(count, 1u);
    if (at < cap) {
        mipgen_candidate c;
        c.region = lo; c.scan_start = R.first_pos + pi; c.capture_size = P->max_capture - (R.k0 + ki) * P->inc;
        c.ext_len = P->arm_ext[a_i]; c.lig_len = P->arm_lig[a_i]; c.strand = (int)(row & 1);
        out[at] = c; out_idx[at] = i;
    }
}

__global__ __launch_bounds__(256) void k_scatter_surv_scores(const double* __restrict__ src, const int64_t* __restrict__ idx, int cap, const unsigned int* __restrict__ n_dev,
                                                             mipgen_survivor* __restrict__ surv, unsigned int* __restrict__ over)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap && (unsigned int)i < *n_dev) surv[idx[i]].score = src[i];
    if (i == 0 && over && *n_dev > (unsigned int)cap) *over = *n_dev - (unsigned int)cap;
}

// ---- every condensed survivor of a window as a candidate list, in slot order (mixed designs: the pick stage re-scores survivors with the SVR,
// mipgen.cpp:1523-1527,1873-1877 - all of them are scored here in one list call and the selection stage looks the values up) ----
__global__ __launch_bounds__(256) void k_surv_keep(const mipgen_survivor* __restrict__ surv, int64_t n, int64_t* __restrict__ keep, double* __restrict__ svr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    keep[i] = (i < n && surv[i].cand_index >= 0) ? 1 : 0;
    if (i < n) svr[i] = __longlong_as_double(0x7ff8000000000000ll);         // NaN: no survivor in this slot
}
__global__ __launch_bounds__(256) void k_surv_candidates(const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, int r0, int r1,
                                                         const mipgen_survivor* __restrict__ surv, int64_t n, int64_t cand0, const int64_t* __restrict__ offs,
                                                         mipgen_candidate* __restrict__ out, int64_t* __restrict__ out_idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mipgen_survivor sv = surv[i];
    if (sv.cand_index < 0) return;
    const int64_t idx = sv.cand_index - cand0;                             // window-relative, like DevRegion::out_off
    const int A = P->n_pairs;
    int lo = r0, hi = r1 - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (regions[mid].out_off <= idx) lo = mid; else hi = mid - 1; }
    const DevRegion& R = regions[lo];
    const int64_t local = idx - R.out_off;
    const int a_i = (int)(local % A);
    const int64_t row = local / A;
    const int64_t rest = row >> 1;
    const int ki = (int)(rest % R.n_sizes), pi = (int)(rest / R.n_sizes);
    mipgen_candidate c;
    c.region = lo; c.scan_start = R.first_pos + pi; c.capture_size = P->max_capture - (R.k0 + ki) * P->inc;
    c.ext_len = P->arm_ext[a_i]; c.lig_len = P->arm_lig[a_i]; c.strand = (int)(row & 1);
    const int64_t at = offs[i];
    out[at] = c; out_idx[at] = i;
}
__global__ __launch_bounds__(256) void k_scatter_f64(const double* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, double* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}
extern "C" hipError_t mipgen_launch_surv_keep(hipStream_t stream, const mipgen_survivor* surv, int64_t n, int64_t* keep, double* svr)
{
    hipLaunchKernelGGL(k_surv_keep, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream, surv, n, keep, svr);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_surv_candidates(hipStream_t stream, const DevParams* P, const DevRegion* regions, int r0, int r1, const mipgen_survivor* surv,
                                                    int64_t n, int64_t cand0, const int64_t* offs, mipgen_candidate* out, int64_t* out_idx)
{
    if (n <= 0 || r1 <= r0) return hipSuccess;
    hipLaunchKernelGGL(k_surv_candidates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, regions, r0, r1, surv, n, cand0, offs, out, out_idx);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_scatter_f64(hipStream_t stream, const double* src, const int64_t* idx, int64_t n, double* dst)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, idx, n, dst);
    return hipGetLastError();
}

extern "C" hipError_t mipgen_launch_print_boundary_scan_surv(hipStream_t stream, const DevParams* P, const DevRegion* regions, int r0, int r1, const mipgen_survivor* surv,
                                                             int64_t n, int64_t cand0, double tol_rel, double tol_abs, mipgen_candidate* out, int64_t* out_idx,
                                                             unsigned int* count, unsigned int cap)
{
    if (n <= 0 || r1 <= r0) return hipSuccess;
    hipLaunchKernelGGL(k_print_boundary_scan_surv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, regions, r0, r1, surv, n, cand0, tol_rel, tol_abs, out, out_idx, count, cap);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_scatter_surv_scores(hipStream_t stream, const double* src, const int64_t* idx, int cap, const unsigned int* n_dev, mipgen_survivor* surv, unsigned int* over)
{
    if (cap <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_surv_scores, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, src, idx, cap, n_dev, surv, over);
    return hipGetLastError();
}

extern "C" hipError_t mipgen_launch_print_boundary_scan_list(hipStream_t stream, const mipgen_candidate* cands, const double* scores, const uint64_t* records, int n,
                                                             double tol_rel, double tol_abs, mipgen_candidate* out, int64_t* out_idx, unsigned int* count, unsigned int cap)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_print_boundary_scan_list, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cands, scores, records, n, tol_rel, tol_abs, out, out_idx, count, cap);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_scatter_scores(hipStream_t stream, const double* src, const int64_t* idx, int cap, const unsigned int* n_dev, double* scores, unsigned int* over)
{
    if (cap <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_scores, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, src, idx, cap, n_dev, scores, over);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Featurev5::get_long_range_content (/root/reference/Featurev5.cpp:18-56; mers mipgen.cpp:32)
// ---------------------------------------------------------------------------------------------------------
struct LrcMers {
    int8_t k[MIPGEN_N_LRC];        // mer length
    int8_t code[MIPGEN_N_LRC];     // base-4 code of the mer
    int8_t rc[MIPGEN_N_LRC];       // base-4 code of its reverse complement, -1 if palindromic
};

// one workgroup per region: LDS histogram of the 1/2/3-mers of its extended sequence, then the 44 frequencies
__global__ __launch_bounds__(256) void k_long_range(const char* __restrict__ seqs, const int64_t* __restrict__ offs, const int32_t* __restrict__ lens,
                                                    const int32_t* __restrict__ denoms, LrcMers M, double* __restrict__ out_all)
{
    __shared__ int c1[4], c2[16], c3[64];
    const int tid = threadIdx.x;
    const char* seq = seqs + offs[blockIdx.x];
    const int len = lens[blockIdx.x], denom = denoms[blockIdx.x];
    double* out = out_all + (int64_t)blockIdx.x * MIPGEN_N_LRC;
    if (tid < 4) c1[tid] = 0;
    if (tid < 16) c2[tid] = 0;
    if (tid < 64) c3[tid] = 0;
    __syncthreads();
    auto code = [](char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; };
    for (int i = tid; i < len; i += 256) {
        const int b0 = code(seq[i]);
        if (b0 > 3) continue;
        atomicAdd(&c1[b0], 1);
        if (i + 1 >= len) continue;
        const int b1 = code(seq[i + 1]);
        if (b1 > 3) continue;
        atomicAdd(&c2[4 * b0 + b1], 1);
        if (i + 2 >= len) continue;
        const int b2 = code(seq[i + 2]);
        if (b2 > 3) continue;
        atomicAdd(&c3[16 * b0 + 4 * b1 + b2], 1);
    }
    __syncthreads();
    if (tid < MIPGEN_N_LRC) {
        const int k = M.k[tid];
        auto cnt = [&](int cd) { return k == 1 ? c1[cd] : (k == 2 ? c2[cd] : c3[cd]); };
        double fwd = (double)cnt(M.code[tid]);
        if (M.rc[tid] >= 0) out[tid] = (fwd + (double)cnt(M.rc[tid])) / denom;      // Featurev5.cpp:49
        else out[tid] = fwd / denom;                                                // :53
    }
}

extern "C" hipError_t mipgen_launch_long_range(hipStream_t stream, int n, const char* seqs, const int64_t* offs, const int32_t* lens,
                                               const int32_t* denoms, const LrcMers* M, double* out_dev)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_long_range, dim3(n), dim3(256), 0, stream, seqs, offs, lens, denoms, *M, out_dev);
    return hipGetLastError();
}

