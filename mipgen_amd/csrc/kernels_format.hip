// kernels_format.hip — the all_mips records of a scored + replayed result window, formatted on the device.
//
//   mipgen::print_details   /root/reference/mipgen.cpp:765-794   20 tab-separated columns per constructed candidate
//
// The front end used to copy the dense results (17 B per candidate) to the host and build every record with string streams; here the
// 331-byte records are assembled in HBM and leave the device as text.  Candidates are numbered in the reference's generation order
// (position, capture size, arm pair, plus then minus: mipgen.cpp:421-491), which within one (position, capture size) row block is a
// fixed interleaving of the two strand-major rows of the dense layout.  Three passes, one wavefront per row block, lanes on the 2 * pairs
// generation slots:
//   k_fmt_count   emitted candidates per row block            -> exclusive scan = rank of the block's first record
//   k_fmt_length  record lengths (they depend on the rank: mip_name carries the running index) per block -> exclusive scan = byte offset
//   k_fmt_write   the bytes
// Numbers are printed as the C library prints them: fmt_g6.h is printf("%g") to the last digit (checked against glibc on 1.5e7 values).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "common.h"
#include "device_utils.h"
#include "fmt_g6.h"

struct FmtRegion {                    // per region of the window: what print_details reads from Featurev5
    int32_t chr_off, chr_len;         // into the string pool
    int32_t label_off, label_len;
    int32_t feature_start, feature_stop;   // start_position - 1, stop_position (mipgen.cpp:788-789)
    int64_t rb0;                      // first row block of the region in the window
};
struct FmtConst {
    char middle[96];                  // universal_middle_mip_seq (mipgen.cpp:199-200)
    int32_t middle_len;
    int32_t n_regions;
    int64_t first_index;              // all_mip_counter before this window
};

__device__ const Pow10DD d_pow10[] = POW10_DD_TABLE;

namespace {

struct Cand {                         // one generation slot of a row block
    bool emitted;
    int a, s, e, l, C, p, ss;
    int64_t idx;
};

__device__ __forceinline__ int find_region(const FmtRegion* __restrict__ fr, int n, int64_t rb)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (fr[mid].rb0 <= rb) lo = mid; else hi = mid - 1; }
    return lo;
}

__device__ __forceinline__ int ndigits(uint64_t v) { int n = 1; while (v >= 10) { v /= 10; n++; } return n; }
__device__ __forceinline__ int nint(int64_t v) { return v < 0 ? 1 + ndigits((uint64_t)(-v)) : ndigits((uint64_t)v); }

__device__ __forceinline__ char comp_letter(char c)
{
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; default: return c; }   // MinusSVMipv4.cpp:6-29
}

// copy number as the reference prints it: the table's value, not the record's saturating field
__device__ __forceinline__ int true_copy(const DevParams* P, const DevRegion& R, const int32_t* copy, int start, int len, uint32_t rec_field)
{
    if (rec_field != 65535u || R.copy_off < 0) return (int)rec_field;
    const int slot = P->len_slot[len], ri = start - R.seq_start;
    return (slot >= 0 && ri >= 0 && ri < R.seq_len) ? copy[R.copy_off + (int64_t)slot * R.seq_len + ri] : 0;
}

struct Geometry { int ext_start, ext_stop, lig_start, lig_stop, scan_stop; };
__device__ __forceinline__ Geometry geometry(const Cand& c)
{
    Geometry g;
    g.scan_stop = c.p + c.ss - 1;
    if (c.s == 0) { g.ext_start = c.p - c.e; g.ext_stop = c.p - 1; g.lig_start = g.scan_stop + 1; g.lig_stop = g.scan_stop + c.l; }      // PlusSVMipv4.cpp:9-12
    else { g.ext_start = g.scan_stop + 1; g.ext_stop = g.scan_stop + c.e; g.lig_start = c.p - c.l; g.lig_stop = c.p - 1; }                // MinusSVMipv4.cpp:32-35
    return g;
}

// writes seq[start, start+len) oriented for the strand; bytes outside the region string are skipped (std::string::substr clips)
__device__ __forceinline__ int put_seq(char* out, const char* __restrict__ letters, const DevRegion& R, int start, int len, bool minus)
{
    int a = start - R.seq_start, b = a + len;
    if (a < 0 || a > R.seq_len) return 0;
    if (b > R.seq_len) b = R.seq_len;
    const char* s = letters + R.seq_off;
    int n = 0;
    if (!minus) for (int i = a; i < b; i++) out[n++] = s[i];
    else for (int i = b - 1; i >= a; i--) out[n++] = comp_letter(s[i]);
    return n;
}
__device__ __forceinline__ int seq_len_clipped(const DevRegion& R, int start, int len)
{
    int a = start - R.seq_start, b = a + len;
    if (a < 0 || a > R.seq_len) return 0;
    if (b > R.seq_len) b = R.seq_len;
    return b - a;
}

}  // namespace

// row block rb of the window -> (region, position, size index); slot g of the block -> candidate
#define FMT_PROLOGUE                                                                                              \
    const int lane = threadIdx.x & 63;                                                                            \
    const int64_t rb = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                              \
    if (rb >= n_rb) return;                                                                                       \
    const int ri = find_region(fr, FC.n_regions, rb);                                                             \
    const DevRegion& R = regions[r0 + ri];                                                                        \
    const int A = P->n_pairs, nK = R.n_sizes;                                                                     \
    const int64_t rl = rb - fr[ri].rb0;                                                                           \
    const int pi = (int)(rl / nK), ki = (int)(rl - (int64_t)pi * nK);                                             \
    const int64_t row0 = R.out_off + ((int64_t)pi * nK + ki) * 2 * A;

#define FMT_SLOT(g, c)                                                                                            \
    Cand c;                                                                                                       \
    c.a = (g) >> 1; c.s = (g) & 1;                                                                                \
    c.emitted = (g) < 2 * A;                                                                                      \
    c.idx = row0 + (int64_t)c.s * A + c.a;                                                                        \
    if (c.emitted) c.emitted = emitted[c.idx] != 0;                                                               \
    c.e = c.emitted ? P->arm_ext[c.a] : 2; c.l = c.emitted ? P->arm_lig[c.a] : 2;                                 \
    c.C = P->max_capture - (R.k0 + ki) * P->inc; c.p = R.first_pos + pi; c.ss = c.C - c.e - c.l;

__global__ __launch_bounds__(256) void k_fmt_count(int64_t n_rb, int r0, FmtConst FC, const FmtRegion* __restrict__ fr, const DevParams* __restrict__ P,
                                                   const DevRegion* __restrict__ regions, const uint8_t* __restrict__ emitted, int64_t* __restrict__ cnt)
{
    FMT_PROLOGUE
    int n = 0;
    for (int g0 = 0; g0 < 2 * A; g0 += 64) {
        const int g = g0 + lane;
        FMT_SLOT(g, c)
        n += __builtin_popcountll(__ballot(c.emitted));
    }
    if (lane == 0) cnt[rb] = n;
}

// length of one record (must agree byte for byte with write_record)
__device__ __forceinline__ int record_length(const Cand& c, const DevParams* P, const DevRegion& R, const FmtRegion& F, const FmtConst& FC,
                                             const int32_t* copy, double score, uint64_t rec, int64_t index)
{
    const Geometry G = geometry(c);
    char tmp[24];
    const int ls = fmt_g6(score, tmp, d_pow10);
    const int ec = true_copy(P, R, copy, G.ext_start, c.e, MIPGEN_REC_EXT_COPY(rec)), lc = true_copy(P, R, copy, G.lig_start, c.l, MIPGEN_REC_LIG_COPY(rec));
    const int le = seq_len_clipped(R, G.ext_start, c.e), ll = seq_len_clipped(R, G.lig_start, c.l), li = seq_len_clipped(R, c.p, c.ss);
    int n = 0;
    n += F.chr_len + 1 + nint(c.s == 0 ? G.ext_start : G.lig_start) + 1 + nint(c.s == 0 ? G.lig_stop : G.ext_stop) + 1 + nint(c.e) + 1 + nint(c.l) + 1 + 1;   // key
    n += 1 + ls + 1 + F.chr_len + 1 + nint(G.ext_start) + 1 + nint(G.ext_stop) + 1 + nint(ec) + 1 + le + 1 + nint(G.lig_start) + 1 + nint(G.lig_stop) + 1 + nint(lc) + 1 + ll;
    n += 1 + nint(c.p) + 1 + nint(G.scan_stop) + 1 + li + 1 + (ll + FC.middle_len + le) + 1 + nint(F.feature_start) + 1 + nint(F.feature_stop) + 1 + 1 + 1 + 3 + 1;
    n += F.label_len + 1 + (index < 1000 ? 4 : ndigits((uint64_t)index)) + (MIPGEN_REC_SNP_COUNT(rec) == 1 ? 6 : 0) + 1;
    return n;
}

__device__ __forceinline__ int write_record(char* out, const Cand& c, const DevParams* P, const DevRegion& R, const FmtRegion& F, const FmtConst& FC,
                                            const char* __restrict__ pool, const char* __restrict__ letters, const int32_t* copy, double score, uint64_t rec, int64_t index)
{
    const Geometry G = geometry(c);
    const bool minus = c.s != 0;
    const int ec = true_copy(P, R, copy, G.ext_start, c.e, MIPGEN_REC_EXT_COPY(rec)), lc = true_copy(P, R, copy, G.lig_start, c.l, MIPGEN_REC_LIG_COPY(rec));
    const uint32_t f = MIPGEN_REC_FLAGS(rec);
    int n = 0;
    auto chr = [&]() { for (int i = 0; i < F.chr_len; i++) out[n++] = pool[F.chr_off + i]; };
    auto num = [&](int64_t v) { n += fmt_int(v, out + n); };
    auto tab = [&]() { out[n++] = '\t'; };
    chr(); out[n++] = ':'; num(minus ? G.lig_start : G.ext_start); out[n++] = '-'; num(minus ? G.ext_stop : G.lig_stop); out[n++] = '/'; num(c.e); out[n++] = ','; num(c.l);
    out[n++] = '/'; out[n++] = minus ? '-' : '+'; tab();
    n += fmt_g6(score, out + n, d_pow10); tab();
    chr(); tab(); num(G.ext_start); tab(); num(G.ext_stop); tab(); num(ec); tab();
    const int ext_at = n; n += put_seq(out + n, letters, R, G.ext_start, c.e, minus); const int ext_n = n - ext_at; tab();
    num(G.lig_start); tab(); num(G.lig_stop); tab(); num(lc); tab();
    const int lig_at = n; n += put_seq(out + n, letters, R, G.lig_start, c.l, minus); const int lig_n = n - lig_at; tab();
    num(c.p); tab(); num(G.scan_stop); tab();
    n += put_seq(out + n, letters, R, c.p, c.ss, minus); tab();
    for (int i = 0; i < lig_n; i++) out[n++] = out[lig_at + i];                       // mip_sequence = lig + middle + ext (mipgen.cpp:605)
    for (int i = 0; i < FC.middle_len; i++) out[n++] = FC.middle[i];
    for (int i = 0; i < ext_n; i++) out[n++] = out[ext_at + i];
    tab(); num(F.feature_start); tab(); num(F.feature_stop); tab(); out[n++] = minus ? '-' : '+'; tab();
    out[n++] = (f & MIPGEN_FLAG_MAPPING) ? '1' : '0'; out[n++] = (f & MIPGEN_FLAG_SNP) ? '1' : '0'; out[n++] = (f & MIPGEN_FLAG_MASKING) ? '1' : '0'; tab();
    for (int i = 0; i < F.label_len; i++) out[n++] = pool[F.label_off + i];
    out[n++] = '_';
    { char d[24]; const int k = fmt_uint((uint64_t)index, d); for (int i = k; i < 4; i++) out[n++] = '0'; for (int i = 0; i < k; i++) out[n++] = d[i]; }   // %04d
    if (MIPGEN_REC_SNP_COUNT(rec) == 1) { out[n++] = '_'; out[n++] = 'S'; out[n++] = 'N'; out[n++] = 'P'; out[n++] = '_'; out[n++] = 'a'; }               // :792
    out[n++] = '\n';
    return n;
}

// PASS 1: fill `len_out[rb]` with the bytes of the block; PASS 2: write them at off[rb]
template <bool WRITE>
__global__ __launch_bounds__(256) void k_fmt_records(int64_t n_rb, int r0, FmtConst FC, const FmtRegion* __restrict__ fr, const char* __restrict__ pool,
                                                     const DevParams* __restrict__ P, const DevRegion* __restrict__ regions, const char* __restrict__ letters,
                                                     const int32_t* __restrict__ copy, const double* __restrict__ scores, const uint64_t* __restrict__ records,
                                                     const uint8_t* __restrict__ emitted, const int64_t* __restrict__ rank0, const int64_t* __restrict__ off,
                                                     int64_t* __restrict__ len_out, char* __restrict__ text)
{
    FMT_PROLOGUE
    const FmtRegion F = fr[ri];
    int64_t rank = rank0[rb];                                   // records before this block in the window
    int64_t at = WRITE ? off[rb] : 0;
    int64_t total = 0;
    for (int g0 = 0; g0 < 2 * A; g0 += 64) {
        const int g = g0 + lane;
        FMT_SLOT(g, c)
        const uint64_t em = __ballot(c.emitted);
        const int before = __builtin_popcountll(em & ((1ull << lane) - 1));
        int mylen = 0;
        double score = 0.0; uint64_t rec = 0;
        const int64_t index = FC.first_index + rank + before + 1;
        if (c.emitted) { score = scores[c.idx]; rec = records[c.idx]; mylen = record_length(c, P, R, F, FC, copy, score, rec, index); }
        // exclusive prefix of the lengths over the lanes (wave scan)
        int incl = mylen;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const int chunk_total = __shfl(incl, 63, 64);
        if (WRITE && c.emitted) write_record(text + at + (incl - mylen), c, P, R, F, FC, pool, letters, copy, score, rec, index);
        at += chunk_total; total += chunk_total;
        rank += __builtin_popcountll(em);
    }
    if (!WRITE && lane == 0) len_out[rb] = total;
}

extern "C" hipError_t mipgen_launch_fmt_count(hipStream_t s, int64_t n_rb, int r0, const FmtConst* FC, const FmtRegion* fr, const DevParams* P, const DevRegion* regions,
                                              const uint8_t* emitted, int64_t* cnt)
{
    if (n_rb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fmt_count, dim3((unsigned)((n_rb + 3) / 4)), dim3(256), 0, s, n_rb, r0, *FC, fr, P, regions, emitted, cnt);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_fmt_records(hipStream_t s, int write, int64_t n_rb, int r0, const FmtConst* FC, const FmtRegion* fr, const char* pool, const DevParams* P,
                                                const DevRegion* regions, const char* letters, const int32_t* copy, const double* scores, const uint64_t* records,
                                                const uint8_t* emitted, const int64_t* rank0, const int64_t* off, int64_t* len_out, char* text)
{
    if (n_rb <= 0) return hipSuccess;
    if (write) hipLaunchKernelGGL(k_fmt_records<true>, dim3((unsigned)((n_rb + 3) / 4)), dim3(256), 0, s, n_rb, r0, *FC, fr, pool, P, regions, letters, copy, scores, records, emitted, rank0, off, len_out, text);
    else hipLaunchKernelGGL(k_fmt_records<false>, dim3((unsigned)((n_rb + 3) / 4)), dim3(256), 0, s, n_rb, r0, *FC, fr, pool, P, regions, letters, copy, scores, records, emitted, rank0, off, len_out, text);
    return hipGetLastError();
}
// exclusive prefix sums of n + 1 int64 (the last input is ignored: out[n] = total); temp storage managed by the caller
extern "C" hipError_t mipgen_scan_i64(hipStream_t s, void* temp, size_t* temp_bytes, const int64_t* in, int64_t* out, int64_t n)
{
    return hipcub::DeviceScan::ExclusiveSum(temp, *temp_bytes, in, out, (int)n, s);
}
