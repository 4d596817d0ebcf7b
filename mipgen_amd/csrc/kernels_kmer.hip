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
        if (slot == 0) { const uint32_t b = filter_index(key, P.filter_bits); atomicOr(&filter[b >> 5], 1u << (b & 31)); }   // canonical kmin-mer of this position
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

__global__ __launch_bounds__(256) void k_kmer_count(const char* __restrict__ genome, int64_t len, KmerParams P, const uint64_t* __restrict__ keys,
                                                    const uint32_t* __restrict__ filter, const uint32_t* __restrict__ folded,
                                                    unsigned int* __restrict__ counts)
{
    __shared__ uint32_t s_bits[1 << (KMER_LDS_BITS - 5)];                 // fold of the filter: bit b = OR of the filter bits with hash prefix b
    __shared__ uint8_t s_code[KMER_CHUNK + 64];                           // base codes of the chunk (+ the bases the last windows reach into)
    __shared__ uint32_t s_queue[KMER_CHUNK];                              // chunk-relative positions that passed the LDS filter
    __shared__ uint32_t s_nq;
    const int tid = threadIdx.x;
    const int kmin = P.k[0];
    const uint64_t kmask = kmin >= 32 ? ~0ull : ((1ull << (2 * kmin)) - 1);
    // the LDS fold pays while it rejects most positions: up to ~2^16 region positions (then <= 22 % pass it)
    const bool use_fold = P.filter_bits <= KMER_LDS_BITS + 3;
    if (use_fold) for (int w = tid; w < (1 << (KMER_LDS_BITS - 5)); w += 256) s_bits[w] = folded[w];      // the 32 KB fold of the filter (k_kmer_fold)
    const int64_t n_chunks = (len + KMER_CHUNK - 1) / KMER_CHUNK;
    for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const int64_t g0 = c * KMER_CHUNK;
        __syncthreads();                                                   // the fold is complete / the previous chunk's queue is drained
        if (tid == 0) s_nq = 0;
        {
            // 16 bytes per thread (the chunk starts on a multiple of 4096 of a 256-byte aligned buffer), the 48 bytes behind it by three more loads
            for (int v = tid; v < (KMER_CHUNK + 48) / 16; v += 256) {
                const int64_t g = g0 + (int64_t)v * 16;
                uint32_t w[4] = {0, 0, 0, 0};
                if (g + 16 <= len) { const uint4 q = *(const uint4*)(genome + g); w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; }
                else for (int b = 0; b < 16; b++) if (g + b < len) w[b >> 2] |= (uint32_t)(uint8_t)genome[g + b] << (8 * (b & 3));   // (a zero byte is "other")
                uint32_t o[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t c4 = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) c4 |= (uint32_t)code_of((char)(w[q] >> (8 * b))) << (8 * b);
                    o[q] = c4;
                }
                *(uint4*)(s_code + v * 16) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
        // ---- one rolling kmin-mer per thread over KMER_PER_THREAD window starts ----
        {
            const int p0 = tid * KMER_PER_THREAD;
            // the 48 codes this thread's windows can touch, in registers (three aligned 16-byte LDS reads)
            uint32_t cw[12];
#pragma unroll
            for (int q = 0; q < 3; q++) { const uint4 v = *(const uint4*)(s_code + p0 + 16 * q); cw[4 * q] = v.x; cw[4 * q + 1] = v.y; cw[4 * q + 2] = v.z; cw[4 * q + 3] = v.w; }
            uint64_t fwd = 0, rc = 0;
            int run = 0;                                                   // ACGT bases since the last other byte
            const int rc_shift = 2 * (kmin - 1);
            // pass A: the filter bit index of every window start (0xFFFFFFFF = no window); with a selective LDS fold the positions it
            // rejects never touch memory, otherwise (large designs) the filter words are fetched directly - all loads of a thread in flight
            uint32_t fidx[KMER_PER_THREAD];
#pragma unroll
            for (int q = 0; q < KMER_PER_THREAD; q++) fidx[q] = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < KMER_PER_THREAD + KMER_MAX_K - 1; j++) {   // byte j ends the window that starts at j - (kmin - 1)
                if (j >= kmin - 1 + KMER_PER_THREAD) continue;          // (no break: the loop must unroll for the register arrays)
                const int b = (int)((cw[j >> 2] >> (8 * (j & 3))) & 0xFF);
                fwd = ((fwd << 2) | (uint64_t)(b & 3)) & kmask; rc = (rc >> 2) | ((uint64_t)(3 - (b & 3)) << rc_shift);
                run = b > 3 ? 0 : run + 1;
                const int st = j - (kmin - 1);
                if (st >= 0 && run >= kmin && g0 + p0 + st < len) {
                    const uint32_t bi = filter_index(fwd < rc ? fwd : rc, P.filter_bits);
                    bool pass = true;
                    if (use_fold) { const uint32_t lb = bi >> (P.filter_bits - KMER_LDS_BITS); pass = (s_bits[lb >> 5] >> (lb & 31)) & 1; }
                    // static register index: st = j - (kmin - 1) with kmin uniform
#pragma unroll
                    for (int q = 0; q < KMER_PER_THREAD; q++) if (q == st && pass) fidx[q] = bi;
                }
            }
            uint32_t fw[KMER_PER_THREAD];
#pragma unroll
            for (int q = 0; q < KMER_PER_THREAD; q++) fw[q] = fidx[q] != 0xFFFFFFFFu ? filter[fidx[q] >> 5] : 0u;
#pragma unroll
            for (int q = 0; q < KMER_PER_THREAD; q++)
                if ((fw[q] >> (fidx[q] & 31)) & 1) s_queue[atomicAdd(&s_nq, 1u)] = (uint32_t)(p0 + q);
        }
        __syncthreads();
        // ---- survivors: the full filter, then every requested length against the table ----
        const uint32_t nq = s_nq;
        for (uint32_t q = tid; q < nq; q += 256) {
            const int rel = (int)s_queue[q];
            walk_codes(s_code + rel, (int)std::min<int64_t>(len - (g0 + rel), KMER_CHUNK + 48 - rel), P, [&](int slot, uint64_t key) -> bool {
                const uint64_t base = (uint64_t)slot * (P.cap_mask + 1);
                uint64_t h = mix64(key) & P.cap_mask;
                for (;;) {
                    const uint64_t cur = keys[base + h];
                    if (cur == key) { atomicAdd(&counts[base + h], 1u); break; }
                    if (cur == KMER_EMPTY) break;
                    h = (h + 1) & P.cap_mask;
                }
                return true;
            });
        }
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
    hipLaunchKernelGGL(k_kmer_count, dim3(grid), dim3(256), 0, s, genome, len, *P, keys, filter, folded, counts);
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
