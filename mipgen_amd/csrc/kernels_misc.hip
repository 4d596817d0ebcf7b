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
#include "common.h"
#include "device_utils.h"
#include "logistic_device.h"

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
__device__ int count_mer(const uint8_t* s, int n, int k, int x, int y, int z)
{
    int c = 0;
    for (int i = 0; i + k <= n; i++) {
        bool m = s[i] == x;
        if (k >= 2) m = m && s[i + 1] == y;
        if (k >= 3) m = m && s[i + 2] == z;
        c += m;
    }
    return c;
}

}  // namespace

__global__ __launch_bounds__(CAND_THREADS) void k_candidates(
    const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, const mipgen_candidate* __restrict__ cands,
    const uint8_t* __restrict__ bases, const int32_t* __restrict__ copy, const uint8_t* __restrict__ unmap,
    const HostConsts* __restrict__ HC, const double* __restrict__ model, int n_sv, double gamma, double rho, int method,
    double* __restrict__ scores, uint64_t* __restrict__ records, double* __restrict__ features,
    mipgen_candidate_ints* __restrict__ ints_out)
{
    __shared__ uint8_t s_ext[MIPGEN_MAX_OLIGO + 2], s_lig[MIPGEN_MAX_OLIGO + 2], s_ins[MAX_INSERT + 2];
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
                       ss > 0 && ss <= MAX_INSERT && e <= MIPGEN_MAX_OLIGO && l <= MIPGEN_MAX_OLIGO && e >= 2 && l >= 2;
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
        int b = base_at(minus ? ext_start + e - 1 - i : ext_start + i) & BASE_CODE_MASK;
        s_ext[i] = (uint8_t)(minus ? comp_code(b) : b);
    }
    for (int i = tid; i < l; i += CAND_THREADS) {
        int b = base_at(minus ? lig_start + l - 1 - i : lig_start + i) & BASE_CODE_MASK;
        s_lig[i] = (uint8_t)(minus ? comp_code(b) : b);
    }
    for (int i = tid; i < ss; i += CAND_THREADS) {
        int b = base_at(minus ? p + ss - 1 - i : p + i) & BASE_CODE_MASK;
        s_ins[i] = (uint8_t)(minus ? comp_code(b) : b);
    }
    // integer record fields (design_mip, mipgen.cpp:606-760)
    if (tid == 0) {
        int masked_n = 0, snp_any = 0, snp_bad = 0, snp_ok = 0, bad = 0;
        for (int arm = 0; arm < 2; arm++) {
            int s0 = arm == 0 ? ext_start : lig_start, n = arm == 0 ? e : l;
            for (int i = 0; i < n; i++) {
                uint8_t b = base_at(s0 + i);
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

    // mer counts: one lane per mer
    if (tid < 84) { int k, x, y, z; ins_mer(tid, k, x, y, z); s_cnt[tid] = count_mer(s_ins, ss, k, x, y, z); }
    else if (tid < 104) { int k, x, y; arm_mer(tid - 84, k, x, y); s_cnt[tid] = count_mer(s_ext, e, k, x, y, 0); }
    else if (tid < 124) { int k, x, y; arm_mer(tid - 104, k, x, y); s_cnt[tid] = count_mer(s_lig, l, k, x, y, 0); }
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
        const int run = run_count_slow(s_ins, 0, ss, false);
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
                const double ex = logistic_exponent(x);
                const double y = pow(MIPGEN_LOGISTIC_BASE, ex);                 // SVMipv4.cpp:247
                logistic = y / (1.0 + y);
            }
            if (scores) scores[blockIdx.x] = logistic;
        }
    }
    if (method != MIPGEN_SCORE_SVR || !scores) return;

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

extern "C" hipError_t mipgen_launch_candidates(
    hipStream_t stream, int n, const DevParams* P, const DevRegion* regions, const mipgen_candidate* cands, const uint8_t* bases,
    const int32_t* copy, const uint8_t* unmap, const HostConsts* HC, const double* model, int n_sv, double gamma, double rho,
    int method, double* scores, uint64_t* records, double* features, mipgen_candidate_ints* ints)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_candidates, dim3(n), dim3(CAND_THREADS), 0, stream, P, regions, cands, bases, copy, unmap, HC, model,
                       n_sv, gamma, rho, method, scores, records, features, ints);
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

// ---------------------------------------------------------------------------------------------------------
// k_svr_batch: the RBF-SVR of a LIST of candidates from their 192 features (mixed designs re-score every condensed survivor).
// k_candidates walks the whole model per candidate (n_sv x 192 doubles from L2 per candidate: ~0.4 us each, bound by model traffic).
// Here a workgroup takes 32 candidates and streams the model through LDS in chunks of 64 support vectors, so a model row is read
// once per 32 candidates; a thread owns 4 candidates x 2 support vectors (6 LDS reads feed 8 subtract + multiply-add pairs per
// dimension: FP64-issue bound, not LDS bound).  Same arithmetic as svm.cpp:329-368, 2511-2515: sum_j (x_j - sv_j)^2 in index order per
// (candidate, SV), exp(-gamma d2), coefficient-weighted sum (the order of that last sum differs from the reference's: ~1e-13).
// ---------------------------------------------------------------------------------------------------------
#define SB_CANDS 32
#define SB_SVS 64
#define SB_PITCH 193                 // doubles per staged row (192 features + 1: odd pitch spreads the banks)

__global__ __launch_bounds__(256) void k_svr_batch(int n, const double* __restrict__ feats, const uint64_t* __restrict__ records,
                                                   const double* __restrict__ model, int n_sv, double gamma, double rho, double* __restrict__ scores)
{
    extern __shared__ __align__(16) double sm[];
    double* X = sm;                                   // [SB_CANDS][SB_PITCH]
    double* S = X + SB_CANDS * SB_PITCH;              // [SB_SVS][SB_PITCH]
    double* aux = S + SB_SVS * SB_PITCH;              // [SB_SVS][2]: coef, extra |sv|^2 of libsvm indices > 192
    double* red = aux + SB_SVS * 2;                   // [32 sgrps][SB_CANDS]
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * SB_CANDS;
    for (int i = tid; i < SB_CANDS * MIPGEN_N_FEATURES; i += 256) {
        const int c = i / MIPGEN_N_FEATURES, j = i - c * MIPGEN_N_FEATURES;
        X[c * SB_PITCH + j] = (c0 + c < n) ? feats[(int64_t)(c0 + c) * MIPGEN_N_FEATURES + j] : 0.0;
    }
    const int cg = tid & 7, sg = tid >> 3;            // 8 candidate groups of 4, 32 SV lanes of 2
    const double* x0 = X + (cg * 4 + 0) * SB_PITCH; const double* x1 = x0 + SB_PITCH; const double* x2 = x1 + SB_PITCH; const double* x3 = x2 + SB_PITCH;
    double sum0 = 0.0, sum1 = 0.0, sum2 = 0.0, sum3 = 0.0;
    for (int s0 = 0; s0 < n_sv; s0 += SB_SVS) {
        __syncthreads();
        for (int i = tid; i < SB_SVS * MIPGEN_N_FEATURES; i += 256) {
            const int s = i / MIPGEN_N_FEATURES, j = i - s * MIPGEN_N_FEATURES;
            S[s * SB_PITCH + j] = (s0 + s < n_sv) ? model[(int64_t)(s0 + s) * SV_ROW + j] : 0.0;
        }
        if (tid < SB_SVS) {
            aux[2 * tid] = (s0 + tid < n_sv) ? model[(int64_t)(s0 + tid) * SV_ROW + SVR_COEF] : 0.0;       // padding rows: coefficient 0
            aux[2 * tid + 1] = (s0 + tid < n_sv) ? model[(int64_t)(s0 + tid) * SV_ROW + SVR_N_EXTRA] : 0.0;
        }
        __syncthreads();
        const double* va = S + (sg * 2) * SB_PITCH; const double* vb = va + SB_PITCH;
        double d0a = 0, d1a = 0, d2a = 0, d3a = 0, d0b = 0, d1b = 0, d2b = 0, d3b = 0;
#pragma unroll 4
        for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
            const double a = va[j], b = vb[j], p0 = x0[j], p1 = x1[j], p2 = x2[j], p3 = x3[j];
            double t;
            t = p0 - a; d0a = fma(t, t, d0a); t = p1 - a; d1a = fma(t, t, d1a); t = p2 - a; d2a = fma(t, t, d2a); t = p3 - a; d3a = fma(t, t, d3a);
            t = p0 - b; d0b = fma(t, t, d0b); t = p1 - b; d1b = fma(t, t, d1b); t = p2 - b; d2b = fma(t, t, d2b); t = p3 - b; d3b = fma(t, t, d3b);
        }
        const double ca = aux[4 * sg], ea = aux[4 * sg + 1], cb = aux[4 * sg + 2], eb = aux[4 * sg + 3];
        sum0 += ca * exp(-gamma * (d0a + ea)) + cb * exp(-gamma * (d0b + eb));
        sum1 += ca * exp(-gamma * (d1a + ea)) + cb * exp(-gamma * (d1b + eb));
        sum2 += ca * exp(-gamma * (d2a + ea)) + cb * exp(-gamma * (d2b + eb));
        sum3 += ca * exp(-gamma * (d3a + ea)) + cb * exp(-gamma * (d3b + eb));
    }
    __syncthreads();
    red[sg * SB_CANDS + cg * 4 + 0] = sum0; red[sg * SB_CANDS + cg * 4 + 1] = sum1; red[sg * SB_CANDS + cg * 4 + 2] = sum2; red[sg * SB_CANDS + cg * 4 + 3] = sum3;
    __syncthreads();
    if (tid < SB_CANDS && c0 + tid < n) {
        double s = 0.0;
        for (int k = 0; k < 32; k++) s += red[k * SB_CANDS + tid];
        scores[c0 + tid] = (MIPGEN_REC_FLAGS(records[c0 + tid]) & MIPGEN_FLAG_VALID) ? s - rho : 0.0;     // a candidate the bounds skips remove scores 0, as in k_candidates
    }
}

extern "C" hipError_t mipgen_launch_svr_batch(hipStream_t stream, int n, const double* feats, const uint64_t* records, const double* model, int n_sv, double gamma,
                                              double rho, double* scores)
{
    if (n <= 0) return hipSuccess;
    const size_t lds = (size_t)(SB_CANDS * SB_PITCH + SB_SVS * SB_PITCH + SB_SVS * 2 + 32 * SB_CANDS) * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*)k_svr_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_svr_batch, dim3((n + SB_CANDS - 1) / SB_CANDS), dim3(256), lds, stream, n, feats, records, model, n_sv, gamma, rho, scores);
    return hipGetLastError();
}
