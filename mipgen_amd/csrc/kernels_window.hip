// kernels_window.hip — uniqueness of whole capture windows without the BWA round trip (opt-in).
//
// The reference writes every capture window (each capture size x every start in [start_flanked - C, stop_flanked)) to a FASTQ file, aligns it
// with `bwa aln` / `bwa samse` and marks the start "unmappable" unless the SAM line contains "X0:i:1" and "X1:i:0"
// (/root/reference/mipgen.cpp:806-823 writes, :841-868 parses); design_mip then sets mapping_failed for every candidate whose footprint
// starts there (:615-625).  A window taken from the genome matches itself with 0 differences, so X0 = its exact occurrences (both strands)
// and X1 = its occurrences with exactly ONE difference.  Modelled here: differences = substitutions (Hamming distance) - bwa aln's gapped
// one-difference hits (-o 1, not within -i 5 bases of the ends) are NOT searched; its -n 0.04 bound never binds at one difference.
//
// Pigeonhole: a window of C >= 2k bases within Hamming distance 1 of a locus contains its first k-mer [0,k) or its second one [k,2k) exactly.
// k = the seed length (the longest arm oligo, <= 31: exact 2-bit keys).  So
//   1. the region k-mers go into the open-addressing table of kernels_kmer.hip (k_kmer_insert) and the genome is streamed past it once
//      (k_kmer_count): the number of genome loci of every seed;
//   2. k_seed_index   the region positions of every seed (lists per table slot; a seed may occur at several region positions);
//   3. k_window_verify  second genome pass: a genome position whose k-mer is a seed with >= 2 loci is extended along its diagonal, for every
//      region position of the seed, as the window's first seed (role 0) and as its second one (role 1, counted only when the first seed does
//      NOT match there - so every (window, locus, strand) is counted once): the offsets of the first two mismatches decide the distance of
//      every capture size at once;  X0 / X1 counters per (capture size, window start) by atomics (one word: a 31-bit X0 count + a sticky "some X1" bit);
//   4. k_window_flags   flag = window contains a non-ACGT byte, or not (X0 prints with a leading '1' and X1 == 0) - the reference's substring
//      tests (:852) accept X0 = 1, 10-19, 100-199, ...; seeds with ONE locus (the common case) are never extended: their windows have
//      X0 = 1 through the first seed alone.
#include "kmer_common.h"

#define WIN_INF 0x7FFFFFFF

namespace {

// canonical key + orientation of the k-mer at seq[i .. i+k): returns false if it holds a non-ACGT byte or runs past len
__device__ __forceinline__ bool seed_at(const char* __restrict__ seq, int64_t i, int64_t len, int k, uint64_t& key, int& flip)
{
    if (i < 0 || i + k > len) return false;
    uint64_t fwd = 0, rc = 0;
    for (int j = 0; j < k; j++) {
        const int b = code_of(seq[i + j]);
        if (b > 3) return false;
        fwd = (fwd << 2) | (uint64_t)b;
        rc |= (uint64_t)(3 - b) << (2 * j);
    }
    flip = fwd <= rc ? 0 : 1;                                    // 0: the forward reading is the canonical one
    key = flip ? rc : fwd;
    return true;
}

__device__ __forceinline__ int64_t probe(const uint64_t* __restrict__ keys, uint64_t cap_mask, uint64_t key)
{
    uint64_t h = mix64(key) & cap_mask;
    for (;;) {
        const uint64_t cur = keys[h];
        if (cur == key) return (int64_t)h;
        if (cur == KMER_EMPTY) return -1;
        h = (h + 1) & cap_mask;
    }
}

}  // namespace

// per region (one thread each, the strings are short): for every position the distance to the next non-ACGT byte at or after it (the
// separator behind the region included), to the end of its region, and back to the start of its region; all capped at 65535
__global__ __launch_bounds__(256) void k_window_spans(const char* __restrict__ q, const int64_t* __restrict__ roff, int n_regions,
                                                      uint16_t* __restrict__ dist_bad, uint16_t* __restrict__ dist_end, uint16_t* __restrict__ dist_start)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    const int64_t a = roff[r], b = roff[r + 1] - 1;                // [a, b) = the region, q[b] = separator
    int64_t nb = 0;
    dist_bad[b] = 0; dist_end[b] = 0; dist_start[b] = 0;
    for (int64_t i = b - 1; i >= a; i--) {
        nb = code_of(q[i]) > 3 ? 0 : nb + 1;
        dist_bad[i] = (uint16_t)(nb > 65535 ? 65535 : nb);
        dist_end[i] = (uint16_t)(b - i > 65535 ? 65535 : b - i);
        dist_start[i] = (uint16_t)(i - a > 65535 ? 65535 : i - a);
    }
}

// FILL = false: region multiplicity of every seed (rmult[slot]++);  FILL = true: the region positions into the slot's list
template <bool FILL>
__global__ __launch_bounds__(256) void k_seed_index(const char* __restrict__ q, int64_t total, int k, const uint64_t* __restrict__ keys, uint64_t cap_mask,
                                                    unsigned int* __restrict__ rmult, const unsigned int* __restrict__ rstart, unsigned int* __restrict__ rfill,
                                                    uint32_t* __restrict__ rlist)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint64_t key; int flip;
    if (!seed_at(q, i, total, k, key, flip)) return;
    const int64_t h = probe(keys, cap_mask, key);
    if (h < 0) return;                                                 // (cannot happen: k_kmer_insert put it there)
    if (!FILL) atomicAdd(&rmult[h], 1u);
    else rlist[rstart[h] + atomicAdd(&rfill[h], 1u)] = (uint32_t)i | ((uint32_t)flip << 31);
}

__global__ __launch_bounds__(256) void k_seed_alloc(uint64_t cap, const unsigned int* __restrict__ rmult, unsigned int* __restrict__ rstart, unsigned int* __restrict__ alloc)
{
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= cap) return;
    const unsigned int m = rmult[h];
    if (m) rstart[h] = atomicAdd(alloc, m);
}

// offsets of the first two mismatches between the window q[s ..] and the genome read from gs in direction dir (-1: the complementary
// strand, read backwards); bytes outside the chromosome and non-ACGT bytes on either side are mismatches
__device__ __forceinline__ void first_two_mismatches(const char* __restrict__ q, int64_t s, const char* __restrict__ G, int64_t glen, int64_t gs, int dir,
                                                     int maxC, int& m1, int& m2)
{
    m1 = WIN_INF; m2 = WIN_INF;
    for (int i = 0; i < maxC; i++) {
        const int a = code_of(q[s + i]);
        const int64_t gi = gs + (int64_t)dir * i;
        int b = (gi < 0 || gi >= glen) ? 5 : code_of(G[gi]);
        if (dir < 0 && b < 4) b = 3 - b;
        if (a > 3 || b > 3 || a != b) {
            if (m1 == WIN_INF) m1 = i; else { m2 = i; return; }
        }
    }
}

struct WinParams {
    int32_t n_sizes;
    int32_t k;                         // seed length
    int32_t max_size;
    int32_t sizes[64];                 // capture sizes (any order)
};

// one genome position per lane-slot: WIN_PER_THREAD consecutive positions per thread, one rolling k-mer
#define WIN_PER_THREAD 16
__global__ __launch_bounds__(256) void k_window_verify(const char* __restrict__ G, int64_t glen, const char* __restrict__ q, int64_t total, WinParams W,
                                                       const uint64_t* __restrict__ keys, uint64_t cap_mask, const unsigned int* __restrict__ counts,
                                                       const uint32_t* __restrict__ filter, int filter_bits,
                                                       const unsigned int* __restrict__ rmult, const unsigned int* __restrict__ rstart, const uint32_t* __restrict__ rlist,
                                                       const uint16_t* __restrict__ dist_start, unsigned int* __restrict__ ctr)
{
    const int k = W.k;
    const int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * WIN_PER_THREAD;
    if (p0 >= glen) return;
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rc_shift = 2 * (k - 1);
    uint64_t fwd = 0, rc = 0;
    int run = 0;
    for (int j = 0; j < WIN_PER_THREAD + k - 1; j++) {
        const int64_t gi = p0 + j;
        if (gi >= glen) break;
        const int b = code_of(G[gi]);
        fwd = ((fwd << 2) | (uint64_t)(b & 3)) & kmask; rc = (rc >> 2) | ((uint64_t)(3 - (b & 3)) << rc_shift);
        run = b > 3 ? 0 : run + 1;
        const int st = j - (k - 1);
        if (st < 0 || run < k) continue;
        const int gflip = fwd <= rc ? 0 : 1;
        const uint64_t key = gflip ? rc : fwd;
        // the same hash k_kmer_insert set the bit with (kernels_kmer.hip: 32-bit keys hash through filter_index32)
        const uint32_t bi = k <= 16 ? filter_index32((uint32_t)key, filter_bits) : filter_index(key, filter_bits);
        if (!((filter[bi >> 5] >> (bi & 31)) & 1)) continue;
        const int64_t h = probe(keys, cap_mask, key);
        if (h < 0 || counts[h] < 2) continue;                          // not a seed, or a seed whose only locus this is
        const bool palin = fwd == rc;
        const int64_t g = p0 + st;                                     // genome start of the k-mer
        const unsigned int n = rmult[h], r0 = rstart[h];
        for (unsigned int e = 0; e < n; e++) {
            const uint32_t ent = rlist[r0 + e];
            const int64_t r = (int64_t)(ent & 0x7FFFFFFFu);
            const int rflip = (int)(ent >> 31);
            for (int o = 0; o < (palin ? 2 : 1); o++) {
                const bool reverse = palin ? o == 1 : (gflip != rflip);
                for (int role = 0; role < 2; role++) {
                    const int64_t s = r - (int64_t)role * k;           // window start in the region concatenation
                    if (role == 1 && (int)dist_start[r] < k) continue;   // the window would start before its region
                    // window[i] <-> forward: G[g - role*k + i];  reverse: comp(G[g + (role + 1) * k - 1 - i])
                    const int64_t gs = reverse ? g + (int64_t)(role + 1) * k - 1 : g - (int64_t)role * k;
                    int m1, m2;
                    first_two_mismatches(q, s, G, glen, gs, reverse ? -1 : 1, W.max_size, m1, m2);
                    if (role == 1 && m1 >= k) continue;                // the first seed matches here as well: counted as role 0 of that seed
                    for (int c = 0; c < W.n_sizes; c++) {
                        const int C = W.sizes[c];
                        const int d = (m1 < C) + (m2 < C);
                        if (d == 0) atomicAdd(&ctr[(int64_t)c * total + s], 1u);
                        else if (d == 1) atomicOr(&ctr[(int64_t)c * total + s], 1u << 31);   // X1 only matters as zero / non-zero (:852-868): a sticky bit cannot wrap
                    }
                }
            }
        }
    }
}

// unmap[c][s] for every capture size c and window start s of the region concatenation
__global__ __launch_bounds__(256) void k_window_flags(const char* __restrict__ q, int64_t total, WinParams W, const uint64_t* __restrict__ keys, uint64_t cap_mask,
                                                      const unsigned int* __restrict__ counts, const uint16_t* __restrict__ dist_bad,
                                                      const uint16_t* __restrict__ dist_end, const unsigned int* __restrict__ ctr, uint8_t* __restrict__ unmap,
                                                      const int64_t* __restrict__ roff, int n_regions, const int4* __restrict__ bounds, uint8_t* __restrict__ any)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total) return;
    // bounds (optional): per region {start_flanked, stop_flanked, seq_start, seq_stop} - only the window starts the reference looks up keep their
    // flag (mipgen.cpp:808-813: current_mip_start in [start_flanked - C, stop_flanked), > 0, the window inside the region string), and
    // any[r] says whether region r has a flagged start at all
    int reg = -1;
    long pos = 0;
    int4 B = make_int4(0, 0, 0, 0);
    if (bounds) {
        int lo = 0, hi = n_regions;                                    // the region whose [roff[r], roff[r + 1]) holds s
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (roff[mid] <= s) lo = mid; else hi = mid; }
        reg = lo; B = bounds[reg];
        pos = (long)B.z + (long)(s - roff[reg]);
    }
    uint64_t key; int flip;
    unsigned int seed_loci = 0;
    if (seed_at(q, s, total, W.k, key, flip)) { const int64_t h = probe(keys, cap_mask, key); if (h >= 0) seed_loci = counts[h]; }
    const int nb = dist_bad[s], ne = dist_end[s];
    for (int c = 0; c < W.n_sizes; c++) {
        const int C = W.sizes[c];
        uint8_t f;
        if (C > ne) f = 0;                                             // the window does not fit into the region string: never written (:813)
        else if (C > nb) f = 1;                                        // a non-ACGT byte: no exact self match
        else {
            const unsigned int w = ctr[(int64_t)c * total + s];
            unsigned int x0 = (w & 0x7FFFFFFFu) + (seed_loci == 1 ? 1u : 0u), x1 = w >> 31;   // X0: 31-bit count (cannot carry into the X1 bit: a genome has < 2^31 loci per strand pair here)
            unsigned int lead = x0;
            while (lead >= 10) lead /= 10;                             // "X0:i:1" is a substring test (:852): any count printed with a leading 1
            f = (lead == 1 && x1 == 0) ? 0 : 1;
        }
        if (f && bounds) {
            if (!(pos >= (long)B.x - C && pos < (long)B.y && pos > 0 && pos + C - 1 <= (long)B.w)) f = 0;
            else any[reg] = 1;
        }
        unmap[(int64_t)c * total + s] = f;
    }
}

extern "C" hipError_t mipgen_launch_window_spans(hipStream_t st, const char* q, const int64_t* roff, int n_regions, uint16_t* dist_bad, uint16_t* dist_end, uint16_t* dist_start)
{
    if (n_regions <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_window_spans, dim3((unsigned)((n_regions + 255) / 256)), dim3(256), 0, st, q, roff, n_regions, dist_bad, dist_end, dist_start);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_seed_index(hipStream_t st, const char* q, int64_t total, int k, const uint64_t* keys, uint64_t cap_mask, unsigned int* rmult,
                                               unsigned int* rstart, unsigned int* rfill, uint32_t* rlist, unsigned int* alloc, int phase)
{
    if (total <= 0) return hipSuccess;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (phase == 0) hipLaunchKernelGGL(k_seed_index<false>, grid, dim3(256), 0, st, q, total, k, keys, cap_mask, rmult, rstart, rfill, rlist);
    else if (phase == 1) hipLaunchKernelGGL(k_seed_alloc, dim3((unsigned)((cap_mask + 1 + 255) / 256)), dim3(256), 0, st, cap_mask + 1, rmult, rstart, alloc);
    else hipLaunchKernelGGL(k_seed_index<true>, grid, dim3(256), 0, st, q, total, k, keys, cap_mask, rmult, rstart, rfill, rlist);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_window_verify(hipStream_t st, const char* G, int64_t glen, const char* q, int64_t total, const int32_t* sizes, int n_sizes, int k,
                                                  const uint64_t* keys, uint64_t cap_mask, const unsigned int* counts, const uint32_t* filter, int filter_bits,
                                                  const unsigned int* rmult, const unsigned int* rstart, const uint32_t* rlist, const uint16_t* dist_start, unsigned int* ctr)
{
    if (glen <= 0 || total <= 0) return hipSuccess;
    WinParams W;
    W.n_sizes = n_sizes; W.k = k; W.max_size = 0;
    for (int i = 0; i < n_sizes; i++) { W.sizes[i] = sizes[i]; W.max_size = std::max(W.max_size, sizes[i]); }
    const int64_t threads = (glen + WIN_PER_THREAD - 1) / WIN_PER_THREAD;
    hipLaunchKernelGGL(k_window_verify, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, G, glen, q, total, W, keys, cap_mask, counts, filter, filter_bits,
                       rmult, rstart, rlist, dist_start, ctr);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_window_flags(hipStream_t st, const char* q, int64_t total, const int32_t* sizes, int n_sizes, int k, const uint64_t* keys,
                                                 uint64_t cap_mask, const unsigned int* counts, const uint16_t* dist_bad, const uint16_t* dist_end, const unsigned int* ctr,
                                                 uint8_t* unmap, const int64_t* roff, int n_regions, const int32_t* bounds, uint8_t* any)
{
    if (total <= 0) return hipSuccess;
    WinParams W;
    W.n_sizes = n_sizes; W.k = k; W.max_size = 0;
    for (int i = 0; i < n_sizes; i++) { W.sizes[i] = sizes[i]; W.max_size = std::max(W.max_size, sizes[i]); }
    hipLaunchKernelGGL(k_window_flags, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q, total, W, keys, cap_mask, counts, dist_bad, dist_end, ctr, unmap,
                       roff, n_regions, (const int4*)bounds, any);
    return hipGetLastError();
}
