// common.h — device/host shared definitions for libmipgen_accel (gfx950 only).
//
// Data layout in HBM for one resident region batch (see DESIGN.md "Data layout"):
//   bases   : u8  per base, all regions concatenated:  bits 0-2 base code (A0 C1 G2 T3 N4 '-'5 other6),
//                 bit 3 = masked-sequence 'N', bits 4-5 = SNP class (0 none / 1 alt ok / 2 alt fails)
//   copy    : i32 [n_len_slots][seq_len] per region (copy_chr_start_stop slices), or absent (= all 1)
//   unmap   : u8  [n_sizes_all][seq_len] per region (unmappable_positions slices), or absent
//   regions : DevRegion[n]
//   scores  : f64 [batch candidates]   records : u64 [batch candidates]     (dense-grid order, mipgen_accel.h)
//   model   : f64 [n_sv][SV_ROW]  row = 192 SV values, coef, partial |sv|^2 per feature block, 1-mer values with the GC slot folded in
#pragma once
#include <stdint.h>
#include "../../include/mipgen_accel.h"

#define BASE_A 0
#define BASE_C 1
#define BASE_G 2
#define BASE_T 3
#define BASE_N 4
#define BASE_DASH 5
#define BASE_OTHER 6
#define BASE_CODE_MASK 7
#define BASE_MASKED_BIT 8
#define BASE_SNP_SHIFT 4

// feature-vector layout (SVMipv4.cpp:69-112)
#define F_EXT 0          // 0..20 mers (+GC at 15), 21 = ext length
#define F_EXT_GC 15
#define F_EXT_LEN 21
#define F_LRC 22         // 22..65
#define F_INS 66         // 66..150 mers (+GC at 129), 151 = scan size
#define F_INS_GC 129
#define F_INS_LEN 151
#define F_LIG 152        // 152..172 mers (+GC at 167), 173 = lig length
#define F_LIG_GC 167
#define F_LIG_LEN 173
#define F_JUNC 174       // 174..189 one-hot
#define F_LEC 190
#define F_LLC 191

// model row
#define SV_ROW 212
#define SVR_COEF 192
#define SVR_N_EXT 193    // sum sv_j^2, j in 0..20
#define SVR_N_INS 194    // j in 66..150
#define SVR_N_LIG 195    // j in 152..172
#define SVR_N_JUNC 196   // j in 174..189
#define SVR_N_TOTAL 197  // all 192 + any libsvm index > 192
#define SVR_N_EXTRA 198
#define SVR_ZERO 199       // always 0.0: target of "no contribution" gathers  // libsvm indices > 192 (x has none: they contribute sv^2, svm.cpp:359-363)
// 200..211: [feature block: ext, insert, lig][base A, C, G, T] = the SV's 1-mer value of that base + (C or G) its GC-fraction value:
// what ONE base adds to the block's 1-mer window sum (the dense kernel's prefix scans gather one slot per base)
#define SVR_C1 200

struct DevRegion {
    int64_t out_off;       // first candidate in scores/records
    int64_t seq_off;       // into bases
    int64_t copy_off;      // into copy (in int32 units), -1 = all copies 1
    int64_t unmap_off;     // into unmap, -1 = none
    int32_t seq_len, seq_start, seq_stop;
    int32_t start_fl, stop_fl;
    int32_t first_pos, n_pos;
    int32_t k0, n_sizes;   // surviving capture sizes k0 .. k0+n_sizes-1
    int32_t pad;
    double lrc[MIPGEN_N_LRC];
};

struct DevParams {
    int32_t min_capture, max_capture, inc, n_sizes_all;
    int32_t n_pairs;
    int32_t check_copy_number;
    int32_t logistic_heuristic, score_method;
    int32_t min_sum, max_sum;
    int32_t e_min, e_max, l_min, l_max;        // arm-length ranges over the pair list
    int32_t n_len_slots;                        // distinct oligo lengths (copy table slots)
    int32_t max_arm_copy_product, target_arm_copy;
    int32_t key_min_sum;                        // the smallest KEY of the reference's arm-sum map (mipgen.cpp:434; = min_sum unless a smaller sum holds no pair)
    double masked_arm_threshold, upper, lower;
    uint8_t arm_ext[MIPGEN_MAX_ARM_PAIRS];
    uint8_t arm_lig[MIPGEN_MAX_ARM_PAIRS];
    int8_t len_slot[MIPGEN_MAX_OLIGO + 1];      // oligo length -> slot in the copy table, -1 unused
    uint16_t group_end[MIPGEN_MAX_ARM_PAIRS];    // for pair a: index one past the last pair of a's arm-sum list
    // dense SVR kernel, per strand and pair (the tail repeats the last pair: chunks are padded): the pair's part of the factor-table
    // byte offsets, upstream | downstream << 16; a lane adds its own (position, capture size) part with ONE 32-bit add
    uint32_t pair_prog[2][MIPGEN_MAX_ARM_PAIRS + 16];
};

// one workgroup of the logistic / records kernel: a run of scan-start positions of one region
struct LogTile {
    int32_t region;
    int32_t p0;            // index of first position (0-based within region)
    int32_t np;
    int32_t pad;
};

// one workgroup of the collapse kernel: 128 consecutive bases (both strands) of one region
struct CollapseTile {
    int32_t region;
    int32_t j0;            // first base (0-based from the region's first scan position)
};

// one workgroup of the dense SVR kernel
struct SvrTile {
    int32_t region;
    int32_t strand;
    int32_t p0, np;        // positions (0-based within region)
    int32_t ki0, kc;       // capture sizes ki0 .. ki0+kc-1 (0-based within the region's surviving list)
    int32_t level, pad;    // dense SVR tiles: which of the region's capture-size runs this is (0 = the largest sizes; kernels_skip.hip); pad = the tile's
                           // scores fit the LDS staging area of the kernel's epilogue (whole-row stores), as the host has checked
};

// static thread geometry of the dense SVR kernel for a parameter set (computed on the host once)
struct SvrGeom {
    int32_t nchunk;        // per-thread chunks of the arm-pair list
    int32_t chunk_len;     // pairs per chunk (<= SVR_MAX_CHUNK)
    int32_t n_e, n_l;      // e_max-e_min+1, l_max-l_min+1
    int32_t group;         // support vectors staged per iteration
    int32_t wpc;           // waves per arm-pair chunk (block = nchunk * wpc waves)
    int32_t pad[2];
};

#define SVR_MAX_CHUNK 15
#define SVR_GROUP 3            // support vectors staged (and interleaved in every stage) per iteration
#define SVR_MAX_THREADS 1024  // nchunk * wpc * 64 <= 1024 (16 waves = 4 per SIMD at <= 128 VGPRs)
#define LOG_THREADS 256

// LDS layout of one dense-SVR tile (offsets in doubles unless noted); shared by host sizing and the kernel.
// The RBF value of a candidate factorises over the feature blocks, K = 2^U(pl,e) * 2^D(pl+ss,l) * coef 2^I(pl,ss), so a group of
// SVR_GROUP support vectors goes through two phases (one barrier after each):
//     X  tables(k):      reads PF[k % 2] and rows[k % 2]; writes the factor tables TB (upstream | downstream arm windows) and IT (inserts)
//        scan(k+1):      reads rows[(k+1) % 2], writes PF[(k+1) % 2] (the seven prefix arrays + per-length constants + insert
//                        constant of each SV of the next group) - latency bound, it hides under the VALU-bound table entries
//     Y  accumulate(k):  reads TB, IT;   rows of group k+2 -> rows[k % 2]
// PF, TB, IT and rows are SV-interleaved ([slot][SVR_GROUP]): one address register serves the three staged SVs through the
// instruction's immediate offset.
#define SVR_N_ARR 7          // prefix arrays per support vector: insert 1/2/3-mers, upstream arm 1/2-mers, downstream arm 1/2-mers
struct SvrLayout {
    int rinv, lg10, rows, pf, tb, it;          // SV-independent: reciprocals, log10(0..101); then the buffers
    // the seven prefix arrays of a PF block (len+1 slots each) are described by svr_arr_len / svr_arr_off / svr_arr_chunk
    // below: plain scalars only, so that the device copy of this struct stays in SGPRs (an indexed member would put it in scratch)
    int ku, kd, ci, pf_stride;                 // PF block: per-length constants, insert constant; slots per SV
    int tj, ti;                                // PF block: junction terms [18], insert scan-size terms [ssr] (per SV, like the rest of the block)
    int tu, td, tb_stride;                     // TB block: first upstream / downstream slot; slots per SV
    int ssr_p;                                 // IT row pitch in slots (scan-size range rounded up to odd: bank spread)
    int bytes_kpar;                            // 16 ints: the scalars of the window-norm pass after the SV loop
    int bytes_desc, bytes_ent, bytes_idx, bytes_sb, bytes_psum;   // byte offsets: scan descriptors (int), table-entry
                                               // descriptors (2 x u32), per-slot SV-row indices of the scans (2 x u16), bases (u8),
                                               // arm-pair sums (u16)
    int n_ent;                                 // arm-window table entries per SV
    int n_ent_lds;                             // descriptors kept in LDS (all of them: registers are at their budget)
    int total_bytes;
    int nq, ins_len, up_cnt, dn_cnt, span_b, rinv_len;
    int ins_sl, up_sl, dn_sl;                  // slots of one insert / upstream / downstream prefix array: len + 1 rounded up to 16 * (2k + 1)
                                               // (a scan lane owns 2k + 1 consecutive slots; the padding gathers the always-zero SV slot)
};

// elements of prefix array k (without the trailing total slot): insert 1/2/3-mers, upstream 1/2-mers, downstream 1/2-mers
#if defined(__HIPCC__)
__host__ __device__
#endif
__attribute__((always_inline)) static inline int svr_arr_len(const SvrLayout& L, int k) { return k < 3 ? +L.ins_len : (k < 5 ? +L.up_cnt : +L.dn_cnt); }   // unary +: select values, not addresses
// slots of array k (padded), and its first slot inside a PF block; also its first entry in the scan index table
#if defined(__HIPCC__)
__host__ __device__
#endif
__attribute__((always_inline)) static inline int svr_pad_slots(int len) { return (((len + 1 + 15) / 16) | 1) * 16; }   // 16 lanes x an ODD number of slots:
                                                                    // the lanes of a scan row then write 16 distinct LDS bank pairs (lane stride = odd x 24 bytes)
#if defined(__HIPCC__)
__host__ __device__
#endif
__attribute__((always_inline)) static inline int svr_arr_off(const SvrLayout& L, int k)
{
    const int ki = k < 3 ? k : 3, ku = k < 3 ? 0 : (k < 5 ? k - 3 : 2), kd = k < 5 ? 0 : k - 5;
    return ki * L.ins_sl + ku * L.up_sl + kd * L.dn_sl;
}
// prefix-array slots per lane in that array's scan unit: one 16-lane DPP row per (array, SV); odd
#if defined(__HIPCC__)
__host__ __device__
#endif
__attribute__((always_inline)) static inline int svr_arr_chunk(const SvrLayout& L, int k) { return (k < 3 ? +L.ins_sl : (k < 5 ? +L.up_sl : +L.dn_sl)) / 16; }
#if defined(__HIPCC__)
__host__ __device__
#endif
__attribute__((always_inline)) static inline SvrLayout svr_layout(int np, int ssmin, int ssmax, int Lmax, int n_arm, int group, int n_up, int n_dn, int n_threads)
{
    SvrLayout L;
    const int ssr = ssmax - ssmin + 1;
    L.nq = np + ssr - 1;
    L.ins_len = np + ssmax;                 // insert prefix arrays cover positions [p_first, p_first + ins_len)
    L.up_cnt = Lmax + np;
    L.dn_cnt = L.nq + Lmax;
    L.ins_sl = svr_pad_slots(L.ins_len); L.up_sl = svr_pad_slots(L.up_cnt); L.dn_sl = svr_pad_slots(L.dn_cnt);
    L.span_b = np + ssmax + 2 * Lmax + 1;
    L.rinv_len = (ssmax > Lmax ? ssmax : Lmax) + 2;
    int o = 0;
    L.rinv = o; o += 3 * L.rinv_len;                    // records of {1/i, 1/(i-1), 1/(i-2)}: 24-byte pitch like every other table the table stage
                                                        // reads with consecutive lengths in consecutive lanes (32 bytes was a 4-way bank conflict)
    L.lg10 = o; o += 102 + 44;                          // + the region's 44 long-range frequencies (read per SV by the constants unit)
    L.n_ent = np * n_up + L.nq * n_dn;
    L.rows = o; o += 2 * group * SV_ROW;
    int g = 3 * L.ins_sl + 2 * L.up_sl + 2 * L.dn_sl;                         // the seven prefix arrays
    L.ku = g; g += n_arm; L.kd = g; g += n_arm;
    L.ci = g; g += 1;
    L.tj = g; g += 18;                                  // junction term of a ligation-arm window: code 0 / 17 -> 0, 1..16 -> the 16 junctions
    L.ti = g; g += ssr;                                 // (scan size - sv_len)^2 term + the insert-side constants, per scan size
    L.pf_stride = g;
    // the scratch of the window-norm pass after the SV loop (np * 80 u32 k-mer counters + np * ssr * 4 ints) aliases the PF area
    int pf_doubles = 2 * group * L.pf_stride;                 // double buffered: scan(k+1) runs beside tables(k)
    const int scratch_doubles = (np * (80 * 4 + ssr * 16) + 7) / 8;
    if (pf_doubles < scratch_doubles) pf_doubles = scratch_doubles;
    L.pf = o; o += pf_doubles;
    L.tu = 0; L.td = np * n_arm; L.tb_stride = np * n_arm + L.nq * n_arm;
    L.tb = o; o += group * L.tb_stride;
    L.ssr_p = ssr | 1;
    L.it = o; o += group * np * L.ssr_p;
    int bytes = o * 8;
    L.bytes_desc = bytes; bytes += SVR_N_ARR * 16 * 4;
    L.n_ent_lds = L.n_ent; (void)n_threads;
    L.bytes_ent = bytes; bytes += 2 * L.n_ent_lds * 4; // per entry: packed slots, packed fields
    L.bytes_psum = bytes; bytes += 2 * SVR_MAX_CHUNK * (SVR_MAX_THREADS / 64) + 16;   // + the work counter of the table stage
    bytes = (bytes + 15) & ~15;
    L.bytes_kpar = bytes; bytes += 64;
    L.bytes_idx = bytes;                               // u32 per slot: byte offset of the SV-row slot it gathers
    bytes += 4 * (3 * L.ins_sl + 2 * L.up_sl + 2 * L.dn_sl);
    L.bytes_sb = bytes; bytes += L.span_b + 8;
    L.total_bytes = (bytes + 15) & ~15;
    return L;
}


