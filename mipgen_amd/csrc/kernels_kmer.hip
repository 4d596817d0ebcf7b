// kernels_kmer.hip — exact arm-oligo copy numbers without the BWA round trip (opt-in).
//
// The reference writes every arm oligo of every region (each oligo length x every position) to a FASTQ file, aligns it with
// `bwa aln` / `bwa samse` and reads the number of best hits back from the X0:i tag (/root/reference/mipgen.cpp:796-873 writes,
// :558-596 parses; the counts feed features 190/191 and the copy filters of condense / collapse / pick).  An oligo always matches
// itself, so its best hits are its EXACT occurrences on either strand - which is a counting problem, not an alignment problem:
//
//   1. k_kmer_insert   every (oligo length k, position) of the region sequences -> canonical 2-bit key (the smaller of the k-mer and
//                      its reverse complement; k <= 31 is exact in 62 bits) into an open-addressing table in HBM, one partition per k
//   2. k_kmer_count    ONE streaming pass over the genome: a thread per genome position extends its forward / reverse-complement keys
//                      base by base and probes the table at every requested k (the table is small and L2-resident; the genome is
//                      read once: this is the HBM-bound kernel of the path, 1 byte per genome base)
//   3. k_kmer_lookup   the counts back into mipgen_region.copy layout: int32 [length][position]
//
// Semantics: occurrences are counted per genome position and strand-agnostic (a locus that reads the same on both strands counts
// once); oligos containing a non-ACGT byte get 100, the value the reference assigns to a read without an X0 tag (:589-592); positions
// whose oligo would run past the end of the region string get 0, the absent-key value (:612-613,829).  BWA's mismatch-tolerant X1 /
// uniqueness test of whole capture windows (:841-868) is NOT reproduced here: parity at that boundary is unpinned.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

#define KMER_EMPTY 0xFFFFFFFFFFFFFFFFull
#define KMER_MAX_K 31

struct KmerParams {
    int32_t n_k;                       // requested oligo lengths, ascending
    int32_t k[MIPGEN_MAX_OLIGO];
    int32_t kmax;
    int32_t pad;
    uint64_t cap_mask;                 // partition capacity - 1 (power of two)
};

namespace {

__device__ __forceinline__ int code_of(char c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// walks the k-mers starting at seq[i] for every requested k; f(kslot, canonical key) is called for each k whose window is all ACGT
template <typename F>
__device__ __forceinline__ int walk(const char* __restrict__ seq, int64_t i, int64_t len, const KmerParams& P, F f)
{
    uint64_t fwd = 0, rc = 0;
    int slot = 0, j = 0;
    for (; j < P.kmax && i + j < len && slot < P.n_k; j++) {
        const int b = code_of(seq[i + j]);
        if (b > 3) break;
        fwd = (fwd << 2) | (uint64_t)b;
        rc |= (uint64_t)(3 - b) << (2 * j);
        if (j + 1 == P.k[slot]) { f(slot, fwd < rc ? fwd : rc); slot++; }
    }
    return slot;                        // number of lengths that produced a key
}

}  // namespace

__global__ __launch_bounds__(256) void k_kmer_insert(const char* __restrict__ seq, int64_t len, KmerParams P, uint64_t* __restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    walk(seq, i, len, P, [&](int slot, uint64_t key) {
        uint64_t* part = keys + (uint64_t)slot * (P.cap_mask + 1);
        uint64_t h = mix64(key) & P.cap_mask;
        for (;;) {
            const unsigned long long prev = atomicCAS((unsigned long long*)&part[h], (unsigned long long)KMER_EMPTY, (unsigned long long)key);
            if (prev == KMER_EMPTY || prev == key) break;
            h = (h + 1) & P.cap_mask;
        }
    });
}

__global__ __launch_bounds__(256) void k_kmer_count(const char* __restrict__ genome, int64_t len, KmerParams P, const uint64_t* __restrict__ keys,
                                                    unsigned int* __restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    walk(genome, i, len, P, [&](int slot, uint64_t key) {
        const uint64_t base = (uint64_t)slot * (P.cap_mask + 1);
        uint64_t h = mix64(key) & P.cap_mask;
        for (;;) {
            const uint64_t cur = keys[base + h];
            if (cur == key) { atomicAdd(&counts[base + h], 1u); break; }
            if (cur == KMER_EMPTY) break;
            h = (h + 1) & P.cap_mask;
        }
    });
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

extern "C" hipError_t mipgen_launch_kmer_insert(hipStream_t s, const char* seq, int64_t len, const KmerParams* P, uint64_t* keys)
{
    if (len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_kmer_insert, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, seq, len, *P, keys);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_kmer_count(hipStream_t s, const char* genome, int64_t len, const KmerParams* P, const uint64_t* keys, unsigned int* counts)
{
    if (len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_kmer_count, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, genome, len, *P, keys, counts);
    return hipGetLastError();
}
extern "C" hipError_t mipgen_launch_kmer_lookup(hipStream_t s, const char* seq, int64_t len, const KmerParams* P, const uint64_t* keys,
                                                const unsigned int* counts, int32_t* out)
{
    if (len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_kmer_lookup, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, seq, len, *P, keys, counts, out);
    return hipGetLastError();
}
