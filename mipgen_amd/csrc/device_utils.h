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

// The same for three arrays of equal length at once: one pass, three barriers instead of nine, the three wave scans interleaved.
// scratch: at least 3 * 16 u64 in LDS.
__device__ inline void block_exclusive_scan3_u64(uint64_t* d0, uint64_t* d1, uint64_t* d2, int n, uint64_t* scratch)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int chunk = (n + nt - 1) / nt;
    const int lo = tid * chunk;
    const int hi = min(lo + chunk, n);
    uint64_t t0 = 0, t1 = 0, t2 = 0;
    for (int i = lo; i < hi; i++) { t0 += d0[i]; t1 += d1[i]; t2 += d2[i]; }
    uint64_t i0 = t0, i1 = t1, i2 = t2;
    const int lane = tid & (WAVE - 1), wid = tid / WAVE, nw = (nt + WAVE - 1) / WAVE;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t u0 = shfl_up_u64(i0, d), u1 = shfl_up_u64(i1, d), u2 = shfl_up_u64(i2, d);
        if (lane >= d) { i0 += u0; i1 += u1; i2 += u2; }
    }
    if (lane == WAVE - 1) { scratch[wid] = i0; scratch[16 + wid] = i1; scratch[32 + wid] = i2; }
    __syncthreads();
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    for (int w = 0; w < wid; w++) { w0 += scratch[w]; w1 += scratch[16 + w]; w2 += scratch[32 + w]; }
    uint64_t r0 = w0 + i0 - t0, r1 = w1 + i1 - t1, r2 = w2 + i2 - t2;
    for (int i = lo; i < hi; i++) {
        const uint64_t v0 = d0[i], v1 = d1[i], v2 = d2[i];
        d0[i] = r0; d1[i] = r1; d2[i] = r2;
        r0 += v0; r1 += v1; r2 += v2;
    }
    if (tid == nt - 1) { d0[n] = r0 + 0; d1[n] = r1; d2[n] = r2; }      // the last thread's running sums are the totals (its chunk ends the array or is empty)
    __syncthreads();
    (void)nw;
}

__device__ __forceinline__ int comp_code(int c) { return c < 4 ? 3 - c : c; }
