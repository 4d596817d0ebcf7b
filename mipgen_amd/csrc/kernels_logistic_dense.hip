// kernels_logistic_dense.hip — dense-grid integer records + logistic scores with per-window tables (gfx950).
//
// Same outputs as k_records_logistic<true> (kernels_logistic.hip): for every dense-grid candidate the 8-byte record of design_mip
// (/root/reference/mipgen.cpp:599-762) and SVMipv4::get_score (/root/reference/SVMipv4.cpp:114-248).  That kernel evaluates all 69
// terms of the exponent and walks the copy / mappability tables per candidate (430 VALU instructions per candidate-wave: 12 % of the
// HBM roofline of its 16 B per candidate).  Here the exponent is regrouped by sequence window (logistic_groups.h, generated from the
// model header):
//     exponent = (C0 - C1) + fE + fL + fT + LG*gLG + LLC*gLLC + LLEN*gLLEN + JS*gJS + LA*gLA + BPS*(hE + hL) + TGC*(hE' + hL') + TA*hL''
// where fE / g* / hE* depend only on the extension-arm window, fL / hL* on the ligation-arm window and fT on the insert.  A workgroup
// (region, run of positions, run of <= 9 capture sizes; both strands, one after the other; 512 threads, <= 80 KB of LDS so that two
// workgroups share a compute unit and one builds tables while the other scores) builds one table entry per distinct
// arm window (8 doubles + a packed integer word: copy number, masked-N / SNP counts, guard, junction; the global copy / mappability
// gathers happen HERE, once per window, not once per candidate) and per insert window (4 doubles), then a wavefront takes one
// (position, capture size) row at a time with its lanes on the arm pairs: a candidate is ~20 LDS reads, a dozen FMAs, one exp2, one
// division, and the two 8-byte stores of a row are contiguous.
#include <hip/hip_runtime.h>
#include "common.h"
#include "device_utils.h"
#include "logistic_device.h"
#include "logistic_groups.h"
#include "exp2_tab.h"

#define LD_THREADS 512
#define LD_ARM_STRIDE 9              // doubles per arm-window entry: 8 values + one packed 64-bit word
#define LD_INS_STRIDE 4              // fT, BPS, TGC, TA
#define LD_MOVE 10                   // words per thread and piece when the downstream-arm table slides (45 starts x 12 lengths x 9 words = 4,860)

// shared with the host (tile sizing)
extern "C" size_t mipgen_logistic_dense_lds_bytes(int np_all, int np, int ssr, int ssmax, int Lmax, int n_up, int n_dn)
{
    // np_all: positions of the tile (bases and prefix words are staged for all of them); np: positions of one sub-run (its tables);
    // ssr: scan sizes of the widest size run of the tile; ssmax: the largest scan size of the whole tile (its first size run)
    const int nq = np + ssr - 1;
    const int span = np_all + ssmax + 2 * Lmax + 2;
    size_t b = 0;
    b += (size_t)3 * (span + 1) * 8 + 48 * 8;                   // W0..W2 prefix words + scan scratch
    b += 102 * 8;                                                // log10 of the copy numbers 0..100 (101 = "more than 100")
    b += 256 * 8;                                                // 2^(j/256) table of the exponential
    b += (size_t)span + 16 + 16;                                 // bases (+ alignment of what follows)
    const int n_max = n_up > n_dn ? n_up : n_dn;                 // both strands reuse the tables: sized for either role assignment
    b += (size_t)np * n_max * LD_ARM_STRIDE * 8;                 // upstream arm windows
    b += (size_t)nq * n_max * LD_ARM_STRIDE * 8;                 // downstream arm windows
    b += (size_t)np * ssr * LD_INS_STRIDE * 8;                   // insert windows
    b += (size_t)np * n_max * 2 + 16;                            // mappability masks of the upstream windows (bit per capture size of the run)
    return (b + 15) & ~(size_t)15;
}

namespace {

// Integer word of an arm window, laid out so that a candidate's record is its two windows' words combined by one ADD (the counts), one OR
// (everything else) and one bit-field insert:
//   low dword   copy number saturated to 16 bits, in the record's field of the window's ROLE (extension: bits 0..15, ligation: 16..31)
//   high dword  bits 0..7 masked-sequence N count | 8..15 SNP count (<= 30 each: the sum of two never carries out of its byte)
//               | bits 16..23 the record's flag byte as far as this window decides it: MIPGEN_FLAG_GUARD (N or '-' in the window), MIPGEN_FLAG_SNP
//                 (an SNP without an alternate-allele arm), MIPGEN_FLAG_HAS_SNP_MIP (one with), LD_COPY0 (copy number <= 0: the score is NaN; not a
//                 record flag, cleared before the record is written)
//               | bits 24..31 the record's junction byte (ligation role: 4*b0+b1 of the oriented arm, 255 = not ACGT; extension role: 0)
#define LD_COPY0 0x40u
__device__ __forceinline__ uint64_t pack_word(bool lig_role, int copy, uint32_t masked, uint32_t snp_any, uint32_t snp_bad, bool snp_ok, bool guard, uint32_t jc)
{
    const uint32_t c16 = (uint32_t)min(max(copy, 0), 65535);
    const uint32_t fl = (guard ? MIPGEN_FLAG_GUARD : 0u) | (snp_ok ? MIPGEN_FLAG_HAS_SNP_MIP : 0u) | (snp_bad != 0 ? MIPGEN_FLAG_SNP : 0u) | (copy <= 0 ? LD_COPY0 : 0u);
    const uint32_t lo = lig_role ? c16 << 16 : c16;
    const uint32_t hi = min(masked, 255u) | (min(snp_any, 255u) << 8) | (fl << 16) | (lig_role ? (jc < 16 ? jc : 255u) << 24 : 0u);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__device__ const double EXP2_TAB[256] = EXP2_TAB_VALUES;

}  // namespace

__global__ __launch_bounds__(LD_THREADS, 2) void k_logistic_dense(
    const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, const SvrTile* __restrict__ tiles, int n_tiles,
    const uint8_t* __restrict__ bases, const int32_t* __restrict__ copy, const uint8_t* __restrict__ unmap,
    const HostConsts* __restrict__ HC, double* __restrict__ scores, uint64_t* __restrict__ records,
    int64_t* __restrict__ sat_idx, unsigned int* __restrict__ sat_count, unsigned int sat_cap
#ifdef MIPGEN_DIAG
    , unsigned long long* __restrict__ prof
#endif
    )
{
#ifdef MIPGEN_DIAG
    unsigned long long pc[5] = {0, 0, 0, 0, 0}, tq = __builtin_readcyclecounter();
#define LD_STAMP(k) { const unsigned long long tn = __builtin_readcyclecounter(); pc[k] += tn - tq; tq = tn; }
#else
#define LD_STAMP(k)
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    const SvrTile tile = tiles[xcd_remap(blockIdx.x, n_tiles)];
    const DevRegion& R = regions[tile.region];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), NW = LD_THREADS / WAVE;
    const int wid = __builtin_amdgcn_readfirstlane(tid / WAVE);       // a scalar: the row arithmetic of the candidate loop stays on the scalar unit
    const int A = P->n_pairs, nK = R.n_sizes, inc = P->inc;
    const int Lmax = max(P->e_max, P->l_max);
    // The tile holds ALL capture sizes of its positions; they are worked off in runs of <= 9 (the tables of a run must fit the LDS budget),
    // largest sizes first.  The bases and their prefix words are staged once for the whole tile - the first run reaches furthest.
    const int n_runs = (tile.kc + 8) / 9;
    // ... and the positions in sub-runs of np_sub (tile.strand carries it: these tiles hold both strands), which slide the downstream-arm table
    // along the region: consecutive sub-runs share all but np_sub of its np_sub + ssr - 1 window starts
    const int np_all = tile.np, np_sub = tile.strand;
    const int n_e = P->e_max - P->e_min + 1, n_l = P->l_max - P->l_min + 1, n_max = max(n_e, n_l);
    const int p_first = R.first_pos + tile.p0;
    const int lo = p_first - Lmax;                                     // chromosome coordinate of local base 0
    const int span = np_all + (P->max_capture - (R.k0 + tile.ki0) * inc - P->min_sum) + 2 * Lmax + 2;

    uint64_t* W0 = (uint64_t*)smem;
    uint64_t* W1 = W0 + (span + 1);
    uint64_t* W2 = W1 + (span + 1);
    uint64_t* scratch = W2 + (span + 1);                               // 48
    double* LG = (double*)(scratch + 48);                              // log10(copy), as log_copy_dev gives it
    double* XT = LG + 102;                                             // 2^(j/256)
    uint8_t* sb = (uint8_t*)(XT + 256);
    double* const tables = (double*)(smem + ((((size_t)(sb - smem) + span + 16 + 15) & ~(size_t)15)));   // the run's tables, carved per run
    for (int i = tid; i < 256; i += LD_THREADS) XT[i] = EXP2_TAB[i];

    // ---- stage bases and the packed prefix words (as k_records_logistic) ---------------------------------------------------------
    for (int i = tid; i < span; i += LD_THREADS) {
        const int ri = lo + i - R.seq_start;
        sb[i] = (ri >= 0 && ri < R.seq_len) ? bases[R.seq_off + ri] : (uint8_t)BASE_OTHER;
    }
    for (int i = tid; i < 102; i += LD_THREADS) LG[i] = i <= 100 ? HC->log10_tab[i] : 2.0;      // SVMipv4.cpp:173-174: copy > 100 ? 2 : log10(copy)
    __syncthreads();
    for (int i = tid; i < span; i += LD_THREADS) {
        const uint8_t b = sb[i];
        const int c = b & BASE_CODE_MASK, snp = (b >> BASE_SNP_SHIFT) & 3;
        int sw = 0;
        if (i > 0) {
            const int cp = sb[i - 1] & BASE_CODE_MASK;
            sw = ((c == BASE_G || c == BASE_C) != (cp == BASE_G || cp == BASE_C));
        }
        W0[i] = (uint64_t)(c == BASE_A) | ((uint64_t)(c == BASE_C) << 16) | ((uint64_t)(c == BASE_G) << 32) | ((uint64_t)(c == BASE_N || c == BASE_DASH) << 48);
        W1[i] = (uint64_t)((b & BASE_MASKED_BIT) != 0) | ((uint64_t)(snp != 0) << 16) | ((uint64_t)(snp == 2) << 32) | ((uint64_t)(snp == 1) << 48);
        W2[i] = (uint64_t)sw | ((uint64_t)(c >= 4) << 16);
    }
    __syncthreads();
    block_exclusive_scan3_u64(W0, W1, W2, span, scratch);
    LD_STAMP(0)

    for (int run = 0; run < n_runs; run++) {
    const int rk0 = tile.ki0 + (int)((int64_t)tile.kc * run / n_runs), rkc = tile.ki0 + (int)((int64_t)tile.kc * (run + 1) / n_runs) - rk0;
    const int Cmax_t = P->max_capture - (R.k0 + rk0) * inc, Cmin_t = Cmax_t - (rkc - 1) * inc;
    const int ssmax = Cmax_t - P->min_sum, ssmin = Cmin_t - P->max_sum, ssr = ssmax - ssmin + 1;
    const int nq_cap = np_sub + ssr - 1;
    double* TU = tables;
    double* TD = TU + (size_t)np_sub * n_max * LD_ARM_STRIDE;
    double* TT = TD + (size_t)nq_cap * n_max * LD_ARM_STRIDE;
    uint16_t* UM = (uint16_t*)(TT + (size_t)np_sub * ssr * LD_INS_STRIDE);

    // Both strands of the tile, one after the other: the '+' and '-' rows of a (position, capture size) are neighbours in the result
    // arrays (912 bytes together), so writing them from the same compute unit a few microseconds apart lets L2 merge them into full lines
    // (a per-strand tile can only ever write half of every line).  The prefix words are shared; the tables are rebuilt per strand.
    for (int strand = 0; strand < 2; strand++) {
    const bool minus = strand != 0;
    // roles: the upstream arm of p is the extension arm on '+', the ligation arm on '-'; the downstream arm the other one
    const int up_min = minus ? +P->l_min : +P->e_min, dn_min = minus ? +P->e_min : +P->l_min;
    const int n_up = minus ? n_l : n_e, n_dn = minus ? n_e : n_l;
    for (int pb = 0; pb < np_all; pb += np_sub) {                     // sub-runs of positions pb .. pb + np - 1
    const int np = min(np_sub, np_all - pb), nq = np + ssr - 1;
    __syncthreads();                                                  // every wavefront is done with the previous tables
    // ---- the downstream-arm table slides: the window starts the previous sub-run shares with this one move to the front ----
    int wl_new = 0;                                                   // first downstream window start that has to be built
    if (pb) {
        wl_new = ssr - 1;                                             // the previous sub-run was a full one: its starts np_sub .. np_sub + ssr - 2 are this one's 0 .. ssr - 2
        const int n_words = (ssr - 1) * n_dn * LD_ARM_STRIDE;         // 8-byte words to move (an entry is 9 of them)
        const uint64_t* src = (const uint64_t*)TD + (size_t)np_sub * n_dn * LD_ARM_STRIDE;
        uint64_t* dst = (uint64_t*)TD;
        // dst < src.  Pieces of LD_MOVE * LD_THREADS words in ascending order: every piece is read into registers by all threads, one barrier,
        // then written - a piece's writes land below everything the later pieces still have to read
        for (int i0 = 0; i0 < n_words; i0 += LD_MOVE * LD_THREADS) {
            uint64_t v[LD_MOVE];
#pragma unroll
            for (int j = 0; j < LD_MOVE; j++) { const int i = i0 + j * LD_THREADS + tid; v[j] = i < n_words ? src[i] : 0; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LD_MOVE; j++) { const int i = i0 + j * LD_THREADS + tid; if (i < n_words) dst[i] = v[j]; }
        }
    }
    // ---- arm-window entries ----------------------------------------------------------------------------------------------------------
    // entry w < nU: upstream window ending at position pl (start = Lmax + pl - len); else downstream window starting at q (= Lmax + ssmin + q)
    const int nU = np * n_up, nD = (nq - wl_new) * n_dn;
    for (int w = tid; w < nU + nD; w += LD_THREADS) {
        const bool up = w < nU;
        int wl, li, len, s;
        if (up) { wl = w / n_up; li = w - wl * n_up; len = up_min + li; s = Lmax + pb + wl - len; }
        else { const int w2 = w - nU; wl = w2 / n_dn; li = w2 - wl * n_dn; wl += wl_new; len = dn_min + li; s = Lmax + ssmin + pb + wl; }
        const bool lig_role = up ? minus : !minus;
        const uint64_t d0 = W0[s + len] - W0[s], d1 = W1[s + len] - W1[s], d2 = W2[s + len] - W2[s];
        const uint32_t nA = f16(d0, 0), nC = f16(d0, 1), nG = f16(d0, 2), nBad = f16(d0, 3), nOther = f16(d2, 1);
        const uint32_t nT = (uint32_t)len - nA - nC - nG - nOther;
        // copy number of the oligo, mipgen.cpp:612-613 (absent key -> 0)
        int cp = 1;
        const int start_chr = lo + s;                                  // chromosome coordinate of the window start
        if (R.copy_off >= 0) {
            const int slot = P->len_slot[len], ri = start_chr - R.seq_start;
            cp = (slot >= 0 && ri >= 0 && ri < R.seq_len) ? copy[R.copy_off + (int64_t)slot * R.seq_len + ri] : 0;
        }
        uint32_t jc = 16;
        if (lig_role) {
            int j0, j1;
            if (!minus) { j0 = sb[s] & BASE_CODE_MASK; j1 = sb[s + 1] & BASE_CODE_MASK; }
            else { j0 = comp_code(sb[s + len - 1] & BASE_CODE_MASK); j1 = comp_code(sb[s + len - 2] & BASE_CODE_MASK); }
            if (j0 < 4 && j1 < 4) jc = (uint32_t)(4 * j0 + j1);
        }
        double* e = (up ? TU : TD) + (size_t)(up ? wl * n_up + li : wl * n_dn + li) * LD_ARM_STRIDE;
        // oriented base contents: the reverse complement swaps A<->T and C<->G
        const double rl = 1.0 / (double)len, dl = (double)len;
        const double a_c = (double)(minus ? nT : nA) * rl, g_c = (double)(minus ? nC : nG) * rl, gc_c = (double)(nC + nG) * rl;
        const double lcopy = cp > 100 ? 2.0 : (cp >= 0 ? LG[cp] : __longlong_as_double(0x7FF8000000000000LL));   // as log_copy_dev
        if (!lig_role) {
            e[0] = (MIPGEN_LOGISTIC_C0 - MIPGEN_LOGISTIC_C1) + MLG_FE(a_c, g_c, gc_c, dl, lcopy);     // the constant term rides on the extension-arm entry
            e[1] = MLG_G_LG(a_c, g_c, gc_c, dl, lcopy); e[2] = MLG_G_LLC(a_c, g_c, gc_c, dl, lcopy); e[3] = MLG_G_LLEN(a_c, g_c, gc_c, dl, lcopy);
            e[4] = MLG_G_JS(a_c, g_c, gc_c, dl, lcopy); e[5] = MLG_G_LA(a_c, g_c, gc_c, dl, lcopy);
            e[6] = MLG_HE_BPS(a_c, g_c, gc_c, dl, lcopy); e[7] = MLG_HE_TGC(a_c, g_c, gc_c, dl, lcopy);
        } else {
            const double js = jc < 16 ? c_junction_scores[jc] : 0.0;
            e[0] = MLG_FL(gc_c, g_c, js, a_c, dl, lcopy);
            e[1] = g_c; e[2] = lcopy; e[3] = js; e[4] = a_c;
            e[5] = MLG_HL_BPS(gc_c, g_c, js, a_c, dl, lcopy); e[6] = MLG_HL_TA(gc_c, g_c, js, a_c, dl, lcopy); e[7] = MLG_HL_TGC(gc_c, g_c, js, a_c, dl, lcopy);
        }
        ((uint64_t*)e)[8] = pack_word(lig_role, cp, f16(d1, 0), f16(d1, 1), f16(d1, 2), f16(d1, 3) != 0, nBad != 0, jc);
        if (up) {
            // mapping flag, mipgen.cpp:615-625: the MIP starts at its upstream arm on either strand; one bit per capture size of the tile
            uint32_t mask = 0;
            if (R.unmap_off >= 0 && P->check_copy_number) {
                const int ms = start_chr - R.seq_start;
                if (ms >= 0 && ms < R.seq_len)
                    for (int k = 0; k < rkc; k++) mask |= (uint32_t)(unmap[R.unmap_off + (int64_t)(R.k0 + rk0 + k) * R.seq_len + ms] != 0) << k;
            }
            UM[wl * n_up + li] = (uint16_t)mask;
        }
    }
    LD_STAMP(1)
    // ---- insert-window entries (position pl, scan size ssmin + ssi) ------------------------------------------------------------------------
    for (int w = tid; w < np * ssr; w += LD_THREADS) {
        const int pl = w / ssr, ssi = w - pl * ssr, ss = ssmin + ssi, bi = Lmax + pb + pl;
        double* t = TT + (size_t)w * LD_INS_STRIDE;
        if (ss <= 0) { t[0] = 0.0; t[1] = 0.0; t[2] = 0.0; t[3] = 0.0; continue; }
        const uint64_t t0 = W0[bi + ss] - W0[bi], t2 = W2[bi + ss] - W2[bi];
        const uint32_t tA = f16(t0, 0), tC = f16(t0, 1), tG = f16(t0, 2), t_other = f16(t2, 1);
        const uint32_t tT = (uint32_t)ss - tA - tC - tG - t_other;
        int run;
        if (t_other == 0) run = 1 + (int)(f16(W2[bi + ss], 0) - f16(W2[bi + 1], 0));        // GC/AT class switches inside the window, SVMipv4.cpp:118-142
        else run = run_count_slow(sb, bi, ss, minus);
        const double dn = (double)ss, rn = 1.0 / dn;
        const double bps = dn / (double)run, tlen = ss > 250 ? 250.0 : dn;
        const double ta = (double)(minus ? tT : tA) * rn, tg = (double)(minus ? tC : tG) * rn, tgc = (double)(tC + tG) * rn;
        t[0] = MLG_FT(bps, tlen, ta, tgc, tg); t[1] = bps; t[2] = tgc; t[3] = ta;
    }
    LD_STAMP(2)
    __syncthreads();
    LD_STAMP(3)

    // ---- candidates: one (position, capture size) row per wavefront pass, lanes on the arm pairs ----------------------------------------
    // The loop is instruction-issue bound (about 100 vector instructions per row before this layout, most of them VOP3 = 4 cycles): whatever is
    // the same for the whole row is computed on the scalar unit - the row's table addresses, its output address, the bounds tests of
    // interior rows -, a lane adds its constant part.
    const double thr = P->masked_arm_threshold;
    constexpr double xc[4] = EXP2_TAB_POLY;
    const double EXP_MAGIC = 6755399441055744.0;                        // 1.5 * 2^52: rounds to an integer, which lands in the low mantissa bits
    const double k256 = HC->ln_base * (1.4426950408889634074 * 256.0);   // exponent -> units of 1/256 of a binary order
    const int n_rows = np * rkc;
    const int arm_hi = max(P->e_max, P->l_max), arm_lo = min(P->e_min, P->l_min);
    for (int a0 = 0; a0 < A; a0 += WAVE) {
        const int a = a0 + lane;
        const bool have = a < A;
        const int e = have ? P->arm_ext[a] : P->arm_ext[0], l = have ? P->arm_lig[a] : P->arm_lig[0], S = e + l;   // (idle lanes compute pair 0, store nothing)
        const int ulen = minus ? l : e, dlen = minus ? e : l;
        const int u_col = ulen - up_min, d_col = dlen - dn_min;
        const int arm_mx = max(e, l), arm_mn = min(e, l);
        const double dl_lig = (double)l;
        // masking_failed is "masked N count / (l + e) > threshold" (mipgen.cpp:610,626): the smallest count that satisfies it, found with the
        // reference's own double division once per pair - the candidates then compare integers
        int mthr;
        {
            const double den = (double)(l + e);
            const double est = thr * den;
            mthr = est >= 1.0 ? (est < den + 2.0 ? (int)est - 1 : l + e + 1) : 0;      // a safe start just below the boundary, then <= 3 exact tests
            while (mthr <= l + e && !((double)mthr / den > thr)) mthr++;
        }
        // volatile: plain ds_read_b64 (2 LDS cycles each).  Left to the compiler, neighbouring doubles of an entry are fetched with ds_read2_b64, which
        // occupies the LDS array for 8 cycles per pair (MI355X_MICROARCH.md, LDS table): 5.8 array cycles per LDS instruction measured in this loop
        typedef volatile __attribute__((address_space(3))) const double lds_cd;
        typedef volatile __attribute__((address_space(3))) const uint64_t lds_cq;
        typedef __attribute__((address_space(3))) const uint16_t lds_ch;
        const uint32_t u_pitch = (uint32_t)n_up * (LD_ARM_STRIDE * 8), d_pitch = (uint32_t)n_dn * (LD_ARM_STRIDE * 8), t_pitch = (uint32_t)ssr * (LD_INS_STRIDE * 8);
        const uint32_t tu_a = (uint32_t)(__UINTPTR_TYPE__)(lds_cd*)TU, td_a = (uint32_t)(__UINTPTR_TYPE__)(lds_cd*)TD, tt_a = (uint32_t)(__UINTPTR_TYPE__)(lds_cd*)TT;
        const uint32_t um_a = (uint32_t)(__UINTPTR_TYPE__)(lds_ch*)UM;
        // lane parts of the table addresses.  Upstream entry of (pl, pair): TU + pl * u_pitch + u_col * 72; downstream entry: TD + (pl + ss - ssmin) * d_pitch
        // + d_col * 72 with ss = C - S: the scalar part holds pl and C, the lane part the pair
        const uint32_t c_up = (uint32_t)u_col * (LD_ARM_STRIDE * 8);
        const uint32_t c_dn = (uint32_t)d_col * (LD_ARM_STRIDE * 8) - (uint32_t)S * d_pitch;
        const uint32_t c_ext = minus ? c_dn : c_up, c_lig = minus ? c_up : c_dn;     // extension-role / ligation-role entry of this strand
        const uint32_t c_ins = (uint32_t)0 - (uint32_t)S * (LD_INS_STRIDE * 8);
        const uint32_t c_um = (uint32_t)u_col * 2u;
        const uint32_t a8 = (uint32_t)a * 8u;
        int pl = wid / rkc, kci = wid - pl * rkc;                                  // scalars; row + NW -> kci + NW, carried into pl (no division per row)
        for (int row = wid; row < n_rows; row += NW) {
            const int C = Cmax_t - kci * inc, p = p_first + pb + pl;
            // scalar parts of the addresses
            const uint32_t s_up = tu_a + (uint32_t)pl * u_pitch, s_dn = td_a + (uint32_t)(pl + C - ssmin) * d_pitch;
            const uint32_t s_ext = minus ? s_dn : s_up, s_lig = minus ? s_up : s_dn;
            const uint32_t s_ins = tt_a + (uint32_t)pl * t_pitch + (uint32_t)(C - ssmin) * (LD_INS_STRIDE * 8);
            const uint32_t s_um = um_a + (uint32_t)(pl * n_up) * 2u;
            const int64_t out_row = R.out_off + ((((int64_t)(tile.p0 + pb + pl) * nK + (rk0 + kci)) * 2 + (minus ? 1 : 0)) * A);
            char* const rrow = (char*)(records + out_row);
            char* const srow = (char*)(scores + out_row);
            // bounds skips, mipgen.cpp:443-444: no lane can fail them in a row this far inside the region (scalar test); otherwise three compares per lane
            const int over = p + C - 1 - R.seq_stop;                              // the arms must be longer than this
            bool valid = have;
            if (!(p > arm_hi && over <= arm_lo && C > P->max_sum)) valid = have && p > arm_mx && over <= arm_mn && C > S;
            lds_cd* T = (lds_cd*)(unsigned long)(s_ins + c_ins);
            lds_cd* Ee = (lds_cd*)(unsigned long)(s_ext + c_ext);      // extension-role entry
            lds_cd* Le = (lds_cd*)(unsigned long)(s_lig + c_lig);      // ligation-role entry
            const uint64_t we = ((lds_cq*)Ee)[8], wl_ = ((lds_cq*)Le)[8];
            const uint32_t um = *(lds_ch*)(unsigned long)(s_um + c_um);
            // ---- exponent (no dependence on the integer words); the constant term is part of Ee[0] ----
            double x = Ee[0] + Le[0];
            x += T[0];
            x = fma(Le[1], Ee[1], x);                                  // LG  * gLG
            x = fma(Le[2], Ee[2], x);                                  // LLC * gLLC
            x = fma(dl_lig, Ee[3], x);                                 // LLEN * gLLEN
            x = fma(Le[3], Ee[4], x);                                  // JS  * gJS
            x = fma(Le[4], Ee[5], x);                                  // LA  * gLA
            x = fma(T[1], Ee[6] + Le[5], x);                           // BPS * (hE + hL)
            x = fma(T[2], Ee[7] + Le[7], x);                           // TGC * (hE + hL)
            x = fma(T[3], Le[6], x);                                   // TA  * hL
            // ---- record: counts from the sum of the two windows' words, flags and junction from their OR, then the flag logic of design_mip ----
            const uint32_t lo32 = (uint32_t)we | (uint32_t)wl_;
            const uint32_t hs = (uint32_t)(we >> 32) + (uint32_t)(wl_ >> 32), ho = (uint32_t)(we >> 32) | (uint32_t)(wl_ >> 32);
            const uint32_t hw = (hs & 0x0000FFFFu) | (ho & 0xFFFF0000u);
            const uint32_t f_mask = (int)(hs & 0xFFu) >= mthr ? (MIPGEN_FLAG_MASKING << 16) : 0u;                        // :610,626
            const uint32_t f_snp = ((hs >> 8) & 0xFFu) > 1u ? (MIPGEN_FLAG_SNP << 16) : 0u;                             // :690-693,759-760 (or an SNP without an arm: in hw)
            const uint32_t h_full = (hw & ~(LD_COPY0 << 16)) | f_mask | f_snp | (MIPGEN_FLAG_VALID << 16);
            // early return of design_mip when the MIP does not map uniquely: masking / SNP fields stay at their defaults
            const uint32_t h_map = (hw & (0xFF0000FFu | (MIPGEN_FLAG_GUARD << 16))) | ((MIPGEN_FLAG_VALID | MIPGEN_FLAG_MAPPING) << 16);
            const uint32_t hi32 = ((um >> kci) & 1u) ? h_map : h_full;
            uint64_t rec = (uint64_t)lo32 | ((uint64_t)hi32 << 32);
            // ---- score = b^x / (1 + b^x), b = 2.71828 (SVMipv4.cpp:247): 2^(t/256) = 2^n * T[j] * (1 + r p(r)) with a 256-entry table ----
            double t = x * k256;
            asm volatile("" : "+v"(t));                                // rounded on its own: not contracted into the two uses below
            const double tm = t + EXP_MAGIC;
            const int ti = __double2loint(tm);
            const double Tj = ((lds_cd*)XT)[ti & 255];
            const double r = t - (tm - EXP_MAGIC);
            double pp = fma(xc[3], r, xc[2]);
            pp = fma(pp, r, xc[1]);
            pp = fma(pp, r, xc[0]);
            const double y0 = fma(pp * r, Tj, Tj);
            const double y = __hiloint2double(__double2hiint(y0) + ((ti & ~255) << 12), __double2loint(y0));
            const double d = 1.0 + y;
            double rc = __builtin_amdgcn_rcp(d);
            rc = fma(fma(-d, rc, 1.0), rc, rc);
            const double q = y * rc;
            double score = fma(fma(-d, q, y), rc, q);                  // one correction of the quotient: within an ulp of the division
            // the rare cases, skipped by the whole wavefront when no lane has one: an exponent beyond the table route (never seen: the libm route keeps
            // the reference's inf / NaN), a copy number <= 0 (log10(0) = -inf or NaN meets terms of both signs: NaN), the guard (SVMipv4.cpp:116)
            const bool odd = !(fabs(t) < 256000.0);
            const uint32_t special = ho & ((LD_COPY0 | MIPGEN_FLAG_GUARD) << 16);
            if (__builtin_expect(__ballot(odd || special != 0) != 0, 0)) {
                if (odd) { const double yy = exp(x * HC->ln_base); score = yy / (1.0 + yy); }
                if (special & (LD_COPY0 << 16)) score = __longlong_as_double(0xFFF8000000000000LL);
                if (special & (MIPGEN_FLAG_GUARD << 16)) score = -1000.0;
            }
            // Saturated scores (b^x >= 2^20: score within 1e-6 of 1 - GC-rich microsatellites get there): the candidates of a position pile up on the
            // last few hundred doubles below 1.0, where the reference's comparisons (collapse / condense: strictly greater; :494: (int)score is 1 only for
            // exactly 1.0) turn on the LAST bit.  There 1 + y is exact and y / (1 + y) moves by delta^2 * dy - nothing - with the 1e-13 this kernel's y
            // is off the reference's, so the correctly rounded IEEE quotient IS the reference's double (tests/golden: design_hard_saturated_*; found by
            // the probe on the hard genome: 16 of 3,540 such scores were an ulp off with the reciprocal route, 6 of 3,313 exact ones not exactly 1.0).
#ifndef LD_NO_BAND                                                      // (A/B timing of the branch: tools/exp/scratch builds only)
            if (__builtin_expect(__ballot(ti >= (20 << 8)) != 0, 0)) {
                if (ti >= (20 << 8) && !odd && special == 0) score = y / d;
                // ... except where b^x reaches [2^53, 2^54): there 1 + y is a tie the reference rounds to even on the LAST bit of its pow, and its score -
                // 1.0, or one / two ulps below - is a coin this kernel's y (1e-13 off) cannot call.  Those candidates are listed and re-scored in the
                // reference's term order with the correctly rounded power (accel_score.hip: rescore_saturated; pow_base_cr.h) before anything is replayed.
                const bool coin = valid && special == 0 && !odd && ti >= 53 * 256 - 1 && ti <= 54 * 256;
                if (__ballot(coin) && coin && sat_count) { const unsigned int at = atomicAdd(sat_count, 1u); if (at < sat_cap) sat_idx[at] = out_row + a; }
            }
#endif
            if (__builtin_expect(__ballot(have && !valid) != 0, 0)) { if (!valid) { rec = 0; score = 0.0; } }
            if (have) { *(uint64_t*)(rrow + a8) = rec; *(double*)(srow + a8) = score; }
            kci += NW;
            while (kci >= rkc) { kci -= rkc; pl++; }
        }
    }
    LD_STAMP(4)
    }   // sub-run of positions
    }   // strand
    }   // run
#ifdef MIPGEN_DIAG
    if (prof && lane == 0 && blockIdx.x < 512) for (int k = 0; k < 5; k++) prof[(blockIdx.x * 8 + wid) * 5 + k] = pc[k];
#endif
}

#ifdef MIPGEN_DIAG
static unsigned long long* g_ld_prof = nullptr;   // diagnostic builds only: cycles per stage, 512 blocks x 8 waves x 5
extern "C" int mipgen_logistic_debug_dump(unsigned long long* host, int n)
{
    if (!g_ld_prof || n > 512 * 8 * 5) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpy(host, g_ld_prof, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

extern "C" hipError_t mipgen_launch_logistic_dense(hipStream_t stream, int n_tiles, size_t lds_bytes, const DevParams* P, const DevRegion* regions,
                                                   const SvrTile* tiles, const uint8_t* bases, const int32_t* copy, const uint8_t* unmap,
                                                   const HostConsts* HC, double* scores, uint64_t* records, int64_t* sat_idx, unsigned int* sat_count, unsigned int sat_cap)
{
    if (n_tiles <= 0) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)k_logistic_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
#ifdef MIPGEN_DIAG
    if (!g_ld_prof && hipMalloc((void**)&g_ld_prof, 512 * 8 * 5 * sizeof(unsigned long long)) != hipSuccess) g_ld_prof = nullptr;
    hipLaunchKernelGGL(k_logistic_dense, dim3(n_tiles), dim3(LD_THREADS), lds_bytes, stream, P, regions, tiles, n_tiles, bases, copy, unmap, HC, scores, records, sat_idx, sat_count, sat_cap, g_ld_prof);
#else
    hipLaunchKernelGGL(k_logistic_dense, dim3(n_tiles), dim3(LD_THREADS), lds_bytes, stream, P, regions, tiles, n_tiles, bases, copy, unmap, HC, scores, records, sat_idx, sat_count, sat_cap);
#endif
    return hipGetLastError();
}
