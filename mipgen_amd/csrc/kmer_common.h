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
#define KMER_THREADS 512                // threads of a genome-pass workgroup, 16 consecutive window starts each
#define KMER_CHUNK (KMER_THREADS * 16) // genome positions per workgroup pass

namespace {

// 0..3 = A C G T (either case), 4 = anything else.  Branch-free: bits 1-2 of the four letters are distinct (A 00, C 01, G 11, T 10), the byte is
// then compared with the letter that code stands for.  (A switch compiles to divergent branches: the genome pass spent two thirds of its time in it.)
__device__ __forceinline__ int code_of(char c)
{
    const uint32_t x = (uint8_t)c;
    const uint32_t code = ((x >> 1) ^ (x >> 2)) & 3u;
    return (x & 0xDFu) == ((0x54474341u >> (8 * code)) & 0xFFu) ? (int)code : 4;
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

// the same for designs whose shortest oligo has <= 16 bases (32-bit keys): one multiply (k_kmer_insert and k_kmer_count agree on the choice)
__device__ __forceinline__ uint32_t filter_index32(uint32_t key, int bits)
{
    return ((key ^ (key >> 15)) * 0x9E3779B1u) >> (32 - bits);
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

