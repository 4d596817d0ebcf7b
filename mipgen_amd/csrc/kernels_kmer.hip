// kernels_kmer.hip — exact arm-oligo copy numbers without the BWA round trip (opt-in).
//
// The reference writes every arm oligo of every region (each oligo length x every position) to a FASTQ file, aligns it with
// `bwa aln` / `bwa samse` and reads the number of best hits back from the X0:i tag (/root/reference/mipgen.cpp:796-873 writes,
// :558-596 parses; the counts feed features 190/191 and the copy filters of condense / collapse / pick).  An oligo always matches
// itself, so its best hits are its EXACT occurrences on either strand - which is a counting problem, not an alignment problem:
//
//   1. k_kmer_insert   every (oligo length k, position) of the region sequences -> canonical 2-bit key (the smaller of the k-mer and
//                      its reverse complement; k <= 31 is exact in 62 bits) into an open-addressing table in HBM, one partition per k
//   2. k_kmer_count    ONE streaming pass over the genome (1 byte per genome base).  Almost no genome position can match: a genome
//                      window equals a region oligo (or its reverse complement) only if its first kmin bases are a kmin-mer of some
//                      region (kmin = the shortest requested length), so a position is first tested against a BLOOM FILTER of the
//                      regions' canonical kmin-mers - a 32 KB fold of it in LDS (one rolling 2-bit update, one multiplicative hash and
//                      one LDS bit test per base), then the full bitmap in HBM/L2 for the few that pass - and only the survivors,
//                      compacted through an LDS queue, walk all requested lengths and probe the table.
//   3. k_kmer_lookup   the counts back into mipgen_region.copy layout: int32 [length][position]
//
// Semantics: occurrences are counted per genome position and strand-agnostic (a locus that reads the same on both strands counts
// once); oligos containing a non-ACGT byte get 100, the value the reference assigns to a read without an X0 tag (:589-592); positions
// whose oligo would run past the end of the region string get 0, the absent-key value (:612-613,829).  BWA's mismatch-tolerant X1 /
// uniqueness test of whole capture windows (:841-868) is NOT reproduced here: parity at that boundary is unpinned.
#include "kmer_common.h"

__global__ __launch_bounds__(256) void k_kmer_insert(const char* __restrict__ seq, int64_t len, KmerParams P, uint64_t* __restrict__ keys,
                                                     uint32_t* __restrict__ filter)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    walk(seq, i, len, P, [&](int slot, uint64_t key) {
        if (slot == 0) { const uint32_t b = P.k[0] <= 16 ? filter_index32((uint32_t)key, P.filter_bits) : filter_index(key, P.filter_bits); atomicOr(&filter[b >> 5], 1u << (b & 31)); }   // canonical kmin-mer of this position
        uint64_t* part = keys + (uint64_t)slot * (P.cap_mask + 1);
        uint64_t h = mix64(key) & P.cap_mask;
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long*)&part[h], (unsigned long long)KMER_EMPTY, (unsigned long long)key);
            if (prev == KMER_EMPTY || prev == key) break;
            h = (h + 1) & P.cap_mask;
        }
    });
}

// bit B of the fold = OR of the 2^(filter_bits - KMER_LDS_BITS) filter bits whose index starts with B (the LDS level tests the top
// KMER_LDS_BITS of the same hash): one thread per fold bit, the wave's ballot assembles two words
__global__ __launch_bounds__(256) void k_kmer_fold(const uint32_t* __restrict__ filter, int filter_bits, uint32_t* __restrict__ folded)
{
    const uint32_t B = blockIdx.x * 256 + threadIdx.x;                     // < 2^KMER_LDS_BITS
    const int extra = filter_bits - KMER_LDS_BITS;
    uint32_t any = 0;
    if (extra >= 5) { const uint32_t n = 1u << (extra - 5); for (uint32_t q = 0; q < n; q++) any |= filter[(size_t)B * n + q]; }
    else { const uint32_t bit0 = B << extra; any = (filter[bit0 >> 5] >> (bit0 & 31)) & ((1u << (1 << extra)) - 1u); }
    const uint64_t m = __ballot(any != 0);
    if ((threadIdx.x & 63) == 0) { folded[B >> 5] = (uint32_t)m; folded[(B >> 5) + 1] = (uint32_t)(m >> 32); }
}

// bit pairs of x in reverse order (pair 0 <-> pair 31): a 2-bit packed k-mer read backwards
__device__ __forceinline__ uint64_t pair_reverse64(uint64_t x)
{
    x = __brevll(x);
    return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}

// The genome pass.  Persistent workgroups of KMER_THREADS threads walk chunks of KMER_CHUNK window starts.  A chunk is staged as 2 bits per base
// (16 bases per word, base i at bits 2i) + one "not ACGT" bit per base, double buffered: the 16-byte global loads of the NEXT chunk are in flight
// while this one is scanned.  A thread owns 16 consecutive window starts: its first kmin-1 bases are placed directly (the packed word IS the
// reverse-complement key up to a complement; the forward key is its pair reversal), then 16 rolling updates with compile-time register indices -
// on 32-bit words when the shortest requested length is <= 16 (K32: the default arm lengths start at 16), else on 64-bit ones.
template <bool K32>
__global__ __launch_bounds__(KMER_THREADS, 6) void k_kmer_count(const char* __restrict__ genome, int64_t len, KmerParams P, const uint64_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ filter, const uint32_t* __restrict__ folded,
                                                             unsigned int* __restrict__ counts)
{
    constexpr int NV = (KMER_CHUNK + 48) / 16;                            // 16-base words of a chunk: its window starts + the 48 bases behind them
    __shared__ uint32_t s_bits[1 << (KMER_LDS_BITS - 5)];                 // fold of the filter: bit b = OR of the filter bits with hash prefix b
    __shared__ uint32_t s_pk[2][NV + 1];                                  // 2-bit base codes
    __shared__ uint16_t s_bad[2][NV + 3];                                 // bit i: base i of the word is not ACGT
    constexpr uint32_t QCAP = KMER_CHUNK * 3 / 4;                         // (three workgroups per compute unit: 51 KB each)
    __shared__ uint16_t s_queue[QCAP];                                    // chunk-relative positions that passed the filter; the overflow is walked in place
    __shared__ uint32_t s_nq[2];
    const int tid = threadIdx.x;
    const int kmin = P.k[0];
    const uint64_t kmask = kmin >= 32 ? ~0ull : ((1ull << (2 * kmin)) - 1);
    // the LDS fold pays while it rejects most positions: up to ~2^16 region positions (then <= 22 % pass it)
    const bool use_fold = P.filter_bits <= KMER_LDS_BITS + 3;
    if (use_fold) for (int w = tid; w < (1 << (KMER_LDS_BITS - 5)); w += KMER_THREADS) s_bits[w] = folded[w];      // the 32 KB fold of the filter (k_kmer_fold)
    if (tid < 2) s_nq[tid] = 0;
    const int64_t n_chunks = (len + KMER_CHUNK - 1) / KMER_CHUNK;

    // 16 genome bytes of word v of chunk c (the chunk starts on a multiple of KMER_CHUNK of a 256-byte aligned buffer; bytes past the end read as 0 = "other")
    auto fetch = [&](int64_t c, int v) -> uint4 {
        const int64_t g = c * KMER_CHUNK + (int64_t)v * 16;
        if (g + 16 <= len) return *(const uint4*)(genome + g);
        uint32_t w[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; b++) if (g + b < len) w[b >> 2] |= (uint32_t)(uint8_t)genome[g + b] << (8 * (b & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    auto pack = [&](int buf, int v, const uint4& q) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        uint32_t pk = 0, bad = 0;
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const int c = code_of((char)(w[b >> 2] >> (8 * (b & 3))));
            pk |= (uint32_t)(c & 3) << (2 * b);
            bad |= (uint32_t)(c > 3) << b;
        }
        s_pk[buf][v] = pk; s_bad[buf][v] = (uint16_t)bad;
    };
    // a window start that passed the filter: its k-mers of every requested length against the table, shortest first
    auto survivor = [&](int buf, int rel) {
        uint64_t kf = 0, kr = 0;
        int slot = 0;
        for (int j = 0; j < P.kmax && slot < P.n_k; j++) {
            const int at = rel + j;
            if ((s_bad[buf][at >> 4] >> (at & 15)) & 1) break;
            const uint64_t b = (s_pk[buf][at >> 4] >> (2 * (at & 15))) & 3u;
            kf = (kf << 2) | b; kr |= (3 - b) << (2 * j);
            if (j + 1 != P.k[slot]) continue;
            const uint64_t key = kf < kr ? kf : kr;
            const uint64_t base = (uint64_t)slot * (P.cap_mask + 1);
            uint64_t h = mix64(key) & P.cap_mask;
            bool found = false;
            for (;;) {
                const uint64_t cur = keys[base + h];
                if (cur == key) { atomicAdd(&counts[base + h], 1u); found = true; break; }
                if (cur == KMER_EMPTY) break;
                h = (h + 1) & P.cap_mask;
            }
            // Absent at this length = absent at every longer one: a longer genome window that equals a region oligo (or its reverse complement)
            // starts with this window, which then equals the oligo's own prefix (or the reverse complement of its suffix) - a region oligo
            // of this length that k_kmer_insert has stored.  The filter's false positives end here after ONE dependent probe instead of n_k.
            if (!found) break;
            slot++;
        }
    };
    // a thread stages word tid; the first threads also the NV - KMER_THREADS words behind the chunk
    constexpr int NX = NV - KMER_THREADS;
    int64_t c = blockIdx.x;
    if (c < n_chunks) {
        pack(0, tid, fetch(c, tid));
        if (tid < NX) pack(0, KMER_THREADS + tid, fetch(c, KMER_THREADS + tid));
    }
    for (int it = 0; c < n_chunks; c += gridDim.x, it++) {
        const int buf = it & 1;
        __syncthreads();                                                   // this chunk is staged (and the fold, the first time); the previous chunk's survivors are done
        const int64_t cn = c + gridDim.x;
        uint4 raw0 = make_uint4(0, 0, 0, 0), raw1 = make_uint4(0, 0, 0, 0);
        if (cn < n_chunks) { raw0 = fetch(cn, tid); if (tid < NX) raw1 = fetch(cn, KMER_THREADS + tid); }   // in flight across the scan
        // ---- one rolling kmin-mer per thread over 16 window starts ----
        {
            const uint32_t w0 = s_pk[buf][tid], w1 = s_pk[buf][tid + 1], w2 = s_pk[buf][tid + 2];
            const uint64_t v01 = (uint64_t)w0 | ((uint64_t)w1 << 32);      // bases 0..31 of this thread, base j at bits 2j
            const int pre = kmin - 1;                                      // bases before the first window end
            const int rc_shift = 2 * (kmin - 1);
            // bit q of `inval`: window q holds a byte that is not ACGT (a window that starts at or behind the end of the genome does: the bytes
            // there are staged as "other") - the OR of the kmin bits from q on, by doubling
            uint64_t inval = (uint64_t)s_bad[buf][tid] | ((uint64_t)s_bad[buf][tid + 1] << 16) | ((uint64_t)s_bad[buf][tid + 2] << 32);
            for (int have = 1; have < kmin;) { const int sh = min(have, kmin - have); inval |= inval >> sh; have += sh; }
            const uint32_t ok16 = ~(uint32_t)inval;
            const int fold_shift = P.filter_bits - KMER_LDS_BITS;
            uint32_t fidx[16];
            if constexpr (K32) {
                // kmin <= 16: the keys are 32-bit words (pre <= 15: the first 16 bases hold the prefix, bases pre .. pre + 15 lie inside bases 0..31)
                const uint32_t kmask32 = kmin >= 16 ? ~0u : ((1u << (2 * kmin)) - 1u), pmask32 = (1u << (2 * pre)) - 1u;
                uint32_t fwd = pre ? __brev(w0 & pmask32) : 0u;
                fwd = (((fwd & 0x55555555u) << 1) | ((fwd >> 1) & 0x55555555u)) >> (pre ? 32 - 2 * pre : 0);
                uint32_t rc = ((~w0) & pmask32) << 2;
                const uint32_t tl = (uint32_t)(v01 >> (2 * pre));
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t b = (tl >> (2 * q)) & 3u;
                    fwd = ((fwd << 2) | b) & kmask32; rc = (rc >> 2) | ((3u - b) << rc_shift);
                    // straight-line: the hash and the fold test are evaluated for every lane (any fold index is in range), one select at the end
                    const uint32_t bi = filter_index32(min(fwd, rc), P.filter_bits);
                    bool pass = (ok16 >> q) & 1;
                    if (use_fold) { const uint32_t lb = bi >> fold_shift; pass = pass && ((s_bits[lb >> 5] >> (lb & 31)) & 1); }
                    fidx[q] = pass ? bi : 0xFFFFFFFFu;
                }
            } else {
                const uint64_t v12 = (uint64_t)w1 | ((uint64_t)w2 << 32);      // bases 16..47
                const uint64_t pmask = (1ull << (2 * pre)) - 1;
                // after `pre` rolling steps: fwd = the bases in reading order in the low 2*pre bits, rc = their complements, base j at bits 2(j+1)
                uint64_t fwd = pair_reverse64(v01 & pmask) >> (64 - 2 * pre);  // pre >= 16 here
                uint64_t rc = ((~v01) & pmask) << 2;
                const uint32_t tl = (uint32_t)(v12 >> (2 * (pre - 16)));       // the 16 bases that end the 16 windows: bases pre .. pre + 15 <= 45
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t b = (tl >> (2 * q)) & 3u;
                    fwd = ((fwd << 2) | (uint64_t)b) & kmask; rc = (rc >> 2) | ((uint64_t)(3u - b) << rc_shift);
                    const uint32_t bi = filter_index(fwd < rc ? fwd : rc, P.filter_bits);
                    bool pass = (ok16 >> q) & 1;
                    if (use_fold) { const uint32_t lb = bi >> fold_shift; pass = pass && ((s_bits[lb >> 5] >> (lb & 31)) & 1); }
                    fidx[q] = pass ? bi : 0xFFFFFFFFu;
                }
            }
            // the filter words of the positions the fold let through: all loads of a thread in flight
            uint32_t fw[16];
#pragma unroll
            for (int q = 0; q < 16; q++) fw[q] = fidx[q] != 0xFFFFFFFFu ? filter[fidx[q] >> 5] : 0u;
#pragma unroll
            for (int q = 0; q < 16; q++)
                if ((fw[q] >> (fidx[q] & 31)) & 1) {
                    const uint32_t at = atomicAdd(&s_nq[buf], 1u);
                    if (at < QCAP) s_queue[at] = (uint16_t)(tid * 16 + q); else survivor(buf, tid * 16 + q);     // (a genome of repeats of the design)
                }
        }
        __syncthreads();
        // ---- survivors: every requested length against the table, shortest first ----
        const uint32_t nq = min(s_nq[buf], QCAP);
        if (tid == 0) s_nq[buf ^ 1] = 0;                                   // the next chunk's counter (last read before this chunk's first barrier)
        for (uint32_t q = tid; q < nq; q += KMER_THREADS) survivor(buf, (int)s_queue[q]);
        // ---- the next chunk's bases into the other buffer (last read by the previous chunk's survivors) ----
        if (cn < n_chunks) { pack(buf ^ 1, tid, raw0); if (tid < NX) pack(buf ^ 1, KMER_THREADS + tid, raw1); }
    }
}

// out: [n_k][len] int32 over the concatenated region sequences (regions are separated by an 'N', which no k-mer crosses); the host
// zeroes the positions whose oligo would run past its region string (the reference never writes those reads, mipgen.cpp:829)
__global__ __launch_bounds__(256) void k_kmer_lookup(const char* __restrict__ seq, int64_t len, KmerParams P, const uint64_t* __restrict__ keys,
                                                     const unsigned int* __restrict__ counts, int32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const int got = walk(seq, i, len, P, [&](int slot, uint64_t key) {
        const uint64_t base = (uint64_t)slot * (P.cap_mask + 1);
        uint64_t h = mix64(key) & P.cap_mask;
        int32_t c = 0;
        for (;;) {
            const uint64_t cur = keys[base + h];
            if (cur == key) { c = (int32_t)counts[base + h]; break; }
            if (cur == KMER_EMPTY) break;
            h = (h + 1) & P.cap_mask;
        }
        out[(int64_t)slot * len + i] = c;
    });
    for (int slot = got; slot < P.n_k; slot++) out[(int64_t)slot * len + i] = 100;     // a non-ACGT byte inside the window (the host zeroes region tails)
}

// The resident layout of the copy tables (common.h: per region int32 [n_len_slots][seq_len], regions back to back) from the [n_k][len] image
// of k_kmer_lookup, so that the counts never leave the device between the counter and the scoring kernels.  roff[r] = start of region r in
// the concatenated sequence (one separator after every region; roff[n_regions] = len).  Oligos that would run past their region string are
// 0 (mipgen.cpp:829); counts the 16-bit record fields cannot carry (>= 65535) are listed for the host as well.
struct KmerBig { int32_t region, length, start, copies; };

__global__ __launch_bounds__(256) void k_kmer_place(const int32_t* __restrict__ src, int64_t len, KmerParams P, const int64_t* __restrict__ roff, int n_regions,
                                                    int32_t* __restrict__ dst, KmerBig* __restrict__ big, unsigned int* __restrict__ n_big, unsigned int big_cap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    int lo = 0, hi = n_regions - 1;
    while (lo < hi) {                                                  // the last region that starts at or before i
        const int mid = (lo + hi + 1) >> 1;
        if (roff[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const int64_t r0 = roff[lo];
    const int rl = (int)(roff[lo + 1] - r0 - 1), rel = (int)(i - r0);
    if (rel >= rl) return;                                             // the separator
    int32_t* block = dst + (r0 - lo) * P.n_k;                          // sum of the earlier regions' lengths x slots
    for (int s = 0; s < P.n_k; s++) {
        const int32_t v = rel < rl - P.k[s] ? src[(int64_t)s * len + i] : 0;
        block[(int64_t)s * rl + rel] = v;
        if (v >= 65535) {
            const unsigned int at = atomicAdd(n_big, 1u);
            if (at < big_cap) big[at] = KmerBig{lo, P.k[s], rel, v};
        }
    }
}

extern "C" hipError_t mipgen_launch_kmer_insert(hipStream_t s, const char* seq, int64_t len, const KmerParams* P, uint64_t* keys, uint32_t* filter)
{
    if (len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_kmer_insert, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, seq, len, *P, keys, filter);
    return hipGetLastError();
}
// once after k_kmer_insert: folded = 2^(KMER_LDS_BITS - 5) words
extern "C" hipError_t mipgen_launch_kmer_fold(hipStream_t s, const uint32_t* filter, int filter_bits, uint32_t* folded)
{
    hipLaunchKernelGGL(k_kmer_fold, dim3((1u << KMER_LDS_BITS) / 256), dim3(256), 0, s, filter, filter_bits, folded);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_kmer_count(hipStream_t s, const char* genome, int64_t len, const KmerParams* P, const uint64_t* keys, const uint32_t* filter,
                                               const uint32_t* folded, unsigned int* counts, int n_cu)
{
    if (len <= 0) return hipSuccess;
    // persistent workgroups (the LDS fold of the filter is built once per workgroup): three per compute unit, each walking chunks of KMER_CHUNK positions
    const int64_t n_chunks = (len + KMER_CHUNK - 1) / KMER_CHUNK;
    const unsigned grid = (unsigned)std::min<int64_t>(n_chunks, (int64_t)std::max(n_cu, 1) * 3);
    if (P->k[0] <= 16) hipLaunchKernelGGL(k_kmer_count<true>, dim3(grid), dim3(KMER_THREADS), 0, s, genome, len, *P, keys, filter, folded, counts);
    else hipLaunchKernelGGL(k_kmer_count<false>, dim3(grid), dim3(KMER_THREADS), 0, s, genome, len, *P, keys, filter, folded, counts);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_kmer_lookup(hipStream_t s, const char* seq, int64_t len, const KmerParams* P, const uint64_t* keys,
                                                const unsigned int* counts, int32_t* out)
{
    if (len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_kmer_lookup, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, seq, len, *P, keys, counts, out);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_kmer_place(hipStream_t s, const int32_t* src, int64_t len, const KmerParams* P, const int64_t* roff, int n_regions,
                                               int32_t* dst, void* big, unsigned int* n_big, unsigned int big_cap)
{
    if (len <= 0 || n_regions <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_kmer_place, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, src, len, *P, roff, n_regions, dst, (KmerBig*)big, n_big, big_cap);
    return hipGetLastError();
}
