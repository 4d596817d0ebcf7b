// device_utils.h — small device helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

#define WAVE 64

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int delta)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_up(lo, delta, WAVE);
    hi = __shfl_up(hi, delta, WAVE);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ double shfl_up_f64(double v, int delta)
{
    return __longlong_as_double((long long)shfl_up_u64((uint64_t)__double_as_longlong(v), delta));
}

__device__ __forceinline__ double shfl_f64(double v, int src)
{
    uint64_t u = (uint64_t)__double_as_longlong(v);
    uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
    lo = __shfl(lo, src, WAVE);
    hi = __shfl(hi, src, WAVE);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    uint64_t u = (uint64_t)__double_as_longlong(v);
    uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
    lo = __shfl_xor(lo, mask, WAVE);
    hi = __shfl_xor(hi, mask, WAVE);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

// XCD-aware block remap (8 XCDs, round-robin dispatch): consecutive logical tiles, which share a region's
// bytes, land on the same XCD's L2.  Bijective for any grid size (cdna_hip_programming.md section 5.5 T1).
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    int q = n >> 3, r = n & 7;
    int xcd = b & 7, k = b >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// exact n / d for n < 2^32 / d using a precomputed M = floor(2^32 / d) + 1
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t M) { return __umulhi(n, M); }

// Block-wide exclusive prefix sum, in place, of data[0..n-1] (LDS); data[n] receives the total.
// Packed 16-bit fields in a u64 scan independently as long as no field reaches 65536.
// scratch: at least 8 u64 in LDS.  All threads of the block must call.
__device__ inline void block_exclusive_scan_u64(uint64_t* data, int n, uint64_t* scratch)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int chunk = (n + nt - 1) / nt;
    const int lo = tid * chunk;
    const int hi = min(lo + chunk, n);
    uint64_t total = 0;
    for (int i = lo; i < hi; i++) total += data[i];
    // inclusive scan of per-thread totals within the wave
    uint64_t inc = total;
    const int lane = tid & (WAVE - 1), wid = tid / WAVE;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint64_t t = shfl_up_u64(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == WAVE - 1) scratch[wid] = inc;
    __syncthreads();
    uint64_t woff = 0;
    for (int w = 0; w < wid; w++) woff += scratch[w];
    uint64_t run = woff + inc - total;
    for (int i = lo; i < hi; i++) {
        uint64_t v = data[i];
        data[i] = run;
        run += v;
    }
    __syncthreads();
    if (tid == 0) {
        uint64_t t = 0;
        const int nw = (nt + WAVE - 1) / WAVE;
        for (int w = 0; w < nw; w++) t += scratch[w];
        data[n] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ int comp_code(int c) { return c < 4 ? 3 - c : c; }
