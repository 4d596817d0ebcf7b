// kmer_common.h — shared pieces of the k-mer kernels (kernels_kmer.hip: arm-oligo copy numbers; kernels_window.hip: capture-window uniqueness)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <algorithm>
#include "common.h"

#define KMER_EMPTY 0xFFFFFFFFFFFFFFFFull
#define KMER_MAX_K 31

struct KmerParams {
    int32_t n_k;                       // requested oligo lengths, ascending
    int32_t k[MIPGEN_MAX_OLIGO];
    int32_t kmax;
    int32_t filter_bits;               // log2 of the Bloom bitmap size in bits (>= KMER_LDS_BITS)
    uint64_t cap_mask;                 // partition capacity - 1 (power of two)
};

#define KMER_LDS_BITS 18               // the LDS fold of the filter: 2^18 bits = 32 KB
#define KMER_CHUNK 4096                // genome positions per workgroup pass
#define KMER_PER_THREAD 16             // consecutive positions per thread (256 threads)

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

// bit index of a canonical kmin-mer in the 2^bits-bit filter (multiplicative hash; the LDS fold uses the top KMER_LDS_BITS of it)
__device__ __forceinline__ uint32_t filter_index(uint64_t key, int bits)
{
    return (uint32_t)(((key ^ (key >> 29)) * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}

// f returns void, or bool = "go on to the longer lengths"
template <typename F>
__device__ __forceinline__ auto call_step(F& f, int slot, uint64_t key) -> decltype(f(slot, key), bool())
{
    if constexpr (std::is_same<decltype(f(slot, key)), bool>::value) return f(slot, key);
    else { f(slot, key); return true; }
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
        if (j + 1 == P.k[slot]) { if (!call_step(f, slot, fwd < rc ? fwd : rc)) break; slot++; }
    }
    return slot;                        // number of lengths that produced a key
}

// the same over staged base codes (0..3 = ACGT, else other)
template <typename F>
__device__ __forceinline__ int walk_codes(const uint8_t* code, int avail, const KmerParams& P, F f)
{
    uint64_t fwd = 0, rc = 0;
    int slot = 0, j = 0;
    for (; j < P.kmax && j < avail && slot < P.n_k; j++) {
        const int b = code[j];
        if (b > 3) break;
        fwd = (fwd << 2) | (uint64_t)b;
        rc |= (uint64_t)(3 - b) << (2 * j);
        if (j + 1 == P.k[slot]) { if (!call_step(f, slot, fwd < rc ? fwd : rc)) break; slot++; }
    }
    return slot;
}

}  // namespace

