// logistic_device.h — device-side pieces of SVMipv4::get_score (/root/reference/SVMipv4.cpp:114-248) shared by the
// dense-grid kernel (kernels_logistic.hip) and the sparse-candidate kernel (kernels_misc.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/mipgen_logistic_model.h"
#include "exp2_coef.h"

// constants computed by the HOST libm once per process, so that table-driven values are bit-identical to what the
// reference's own log10()/log() calls return on the host
struct HostConsts {
    double log10_tab[101];     // log10(0..100); log10(0) = -inf
    double ln_base;            // log(2.71828)
    double pad[2];
};

namespace {

constexpr mipgen_logistic_term k_terms[MIPGEN_LOGISTIC_NTERMS] = MIPGEN_LOGISTIC_TERMS;
__constant__ double c_junction_scores[16] = MIPGEN_JUNCTION_SCORES;

// packed prefix words
//  W0: A | C<<16 | G<<32 | bad(N or '-')<<48
//  W1: maskedN | snp_any<<16 | snp_bad<<32 | snp_ok<<48
//  W2: class switch | nonACGT<<16
__device__ __forceinline__ uint32_t f16(uint64_t w, int k) { return (uint32_t)(w >> (16 * k)) & 0xFFFFu; }

__device__ __forceinline__ double log_copy_dev(const HostConsts* HC, int copy)
{
    // SVMipv4.cpp:173-174: copy > 100 ? 2 : log10(copy)
    if (copy > 100) return 2.0;
    if (copy >= 0) return HC->log10_tab[copy];
    return __longlong_as_double(0x7FF8000000000000LL);   // log10 of a negative int: NaN
}

// SVMipv4.cpp:118-142 on the strand-oriented insert, for windows containing non-ACGT characters.
// sb: base bytes; window [b0, b0+n) in forward coordinates; minus = walk it reversed (classes are
// complement-invariant: G<->C, A<->T).
__device__ inline int run_count_slow(const uint8_t* sb, int b0, int n, bool minus)
{
    // state: 0 = last in {G,C}; 1 = last in {A,T}; 2 = last is anything else
    auto cls = [&](int i) {
        int c = sb[minus ? (b0 + n - 1 - i) : (b0 + i)] & BASE_CODE_MASK;
        return (c == BASE_G || c == BASE_C) ? 0 : ((c == BASE_A || c == BASE_T) ? 1 : 2);
    };
    int last = cls(0), run = 0;
    for (int i = 1; i < n; i++) {
        int c = cls(i);
        if (c == 0) { if (last != 0) { run++; last = 0; } }
        else { if (last != 1) { run++; last = c; } }   // current is A/T/other: a switch unless last is A/T
    }
    return run + 1;
}

struct Vars { double v[MLV_COUNT]; };

// exponent of the logistic, same term order as the reference (SVMipv4.cpp:176-246): (C0 - C1) + t1 + t2 + ...
__device__ __forceinline__ double logistic_exponent(const Vars& x)
{
    double ex = MIPGEN_LOGISTIC_C0 - MIPGEN_LOGISTIC_C1;
#pragma unroll
    for (int i = 0; i < MIPGEN_LOGISTIC_NTERMS; i++) {
        double t;
        if (k_terms[i].kind == MLT_LIN) t = k_terms[i].coef * x.v[k_terms[i].v1];
        else if (k_terms[i].kind == MLT_BIL) t = k_terms[i].coef * x.v[k_terms[i].v1] * x.v[k_terms[i].v2];
        else t = k_terms[i].coef * (x.v[k_terms[i].v1] * x.v[k_terms[i].v1]);
        ex = ex + t;
    }
    return ex;
}

// The same with every multiplication and addition rounded on its own, as g++ compiles the reference's expression on x86-64 (no FMA contraction):
// the exponent of the print-exact re-score (k_candidates) equals the reference's double bit for bit.
__device__ __forceinline__ double logistic_exponent_exact(const Vars& x)
{
    // Plain operators under `fp contract(off)`: HIP's __dmul_rn / __dadd_rn are plain operators compiled under the default contract(fast) - inlined, their
    // instructions carry the contract flag and hipcc fuses them into FMAs whatever the caller says (measured: up to 98 ulp from the reference's score).
#pragma clang fp contract(off)
    double ex = MIPGEN_LOGISTIC_C0 - MIPGEN_LOGISTIC_C1;
#pragma unroll
    for (int i = 0; i < MIPGEN_LOGISTIC_NTERMS; i++) {
        double t;
        if (k_terms[i].kind == MLT_LIN) t = k_terms[i].coef * x.v[k_terms[i].v1];
        else if (k_terms[i].kind == MLT_BIL) { t = k_terms[i].coef * x.v[k_terms[i].v1]; t = t * x.v[k_terms[i].v2]; }
        else { t = x.v[k_terms[i].v1] * x.v[k_terms[i].v1]; t = k_terms[i].coef * t; }
        ex = ex + t;
    }
    return ex;
}

// pow(2.71828, ex) / (1 + pow(2.71828, ex))   (SVMipv4.cpp:247), with pow(b, x) = exp(x * ln b)
__device__ __forceinline__ double logistic_from_vars(const HostConsts* HC, const Vars& x)
{
    const double y = exp(logistic_exponent(x) * HC->ln_base);
    return y / (1.0 + y);
}

// Same value through exp2: y = 2^(ex * log2(2.71828)) with the degree-10 near-minimax polynomial (6.7e-16), no libm call.
// Non-finite exponents (copy number 0 -> log10 = -inf -> inf - inf) take the libm route so NaN / 0 / NaN come out as in the reference.
__device__ __forceinline__ double logistic_from_exponent_fast(const HostConsts* HC, double ex)
{
    constexpr double c[11] = EXP2_COEF_10;
    const double t0 = ex * (HC->ln_base * 1.4426950408889634074);
    if (!(fabs(t0) < 1000.0)) {                                 // also catches NaN
        const double y = exp(ex * HC->ln_base);
        return y / (1.0 + y);
    }
    const double MAGIC = 6755399441055744.0;
    const double tm = t0 + MAGIC;
    const double f = t0 - (tm - MAGIC);
    double p = c[10];
#pragma unroll
    for (int k = 9; k >= 0; k--) p = fma(p, f, c[k]);
    const double y = __hiloint2double(__double2hiint(p) + (__double2loint(tm) << 20), __double2loint(p));
    return y / (1.0 + y);
}

// The same with y / (1 + y) as y * rcp(1 + y): v_rcp_f64 refined by one Newton step (relative error ~1e-16) instead of the
// IEEE division sequence.
__device__ __forceinline__ double logistic_from_exponent_rcp(const HostConsts* HC, double ex)
{
    constexpr double c[11] = EXP2_COEF_10;
    const double t0 = ex * (HC->ln_base * 1.4426950408889634074);
    if (!(fabs(t0) < 1000.0)) {                                 // also catches NaN
        const double y = exp(ex * HC->ln_base);
        return y / (1.0 + y);
    }
    const double MAGIC = 6755399441055744.0;
    const double tm = t0 + MAGIC;
    const double f = t0 - (tm - MAGIC);
    double p = c[10];
#pragma unroll
    for (int k = 9; k >= 0; k--) p = fma(p, f, c[k]);
    const double y = __hiloint2double(__double2hiint(p) + (__double2loint(tm) << 20), __double2loint(p));
    const double d = 1.0 + y;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    const double q = y * r;
    return fma(fma(-d, q, y), r, q);                            // one correction of the quotient: within an ulp of the division
}

}  // namespace
