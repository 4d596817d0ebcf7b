// pow_base_cr.h — 2.71828^x (the logistic of SVMipv4.cpp:247) correctly rounded: x * ln(b) in double-double from a triple-double ln(b), argument
// reduction by ln 2 (three doubles, the first with 12 trailing zero bits: k * hi is exact), exp of the remainder / 16 by its Taylor series in
// double-double (18 terms: 0.022^18 / 18! < 2^-150), four squarings, rounded once.  glibc's pow (error bound 0.52 ULP) returns this double except where
// the true value lies within ~0.02 ULP of a rounding midpoint: 99.92 % of 2,000,000 random exponents in [-45, 45] bit-identical to glibc 2.35's pow, the rest
// one ulp (tools/exp/pow_base_cr_test.c).  Why it matters: where b^x reaches 2^53 ... 2^54 the reference's 1 + y rounds to even on the LAST bit of y, and its
// score - 1.0 or one / two ulps below - decides strictly-greater comparisons of collapse / condense and the (int) truncation of mipgen.cpp:494-497.
#pragma once
#include <math.h>
#ifdef __HIPCC__
#define PBC_FN __host__ __device__ static inline
#else
#define PBC_FN static inline
#endif
// Every helper computes under `fp contract(off)`: the error-free transformations below are exact only if a product that was rounded once is not fused
// into the sum that consumes it (hipcc contracts by default; a fused x * ln(b) inside two_sum cost 18 ulps on the device, tools/microbench/pbc_check.hip).
#ifdef __clang__
#define PBC_EXACT _Pragma("clang fp contract(off)")
#else
#define PBC_EXACT
#endif
typedef struct { double h, l; } pbc_dd;
PBC_FN pbc_dd pbc_two_sum(double a, double b) { PBC_EXACT double s = a + b, bb = s - a; pbc_dd r = {s, (a - (s - bb)) + (b - bb)}; return r; }
PBC_FN pbc_dd pbc_fast_two_sum(double a, double b) { PBC_EXACT double s = a + b; pbc_dd r = {s, b - (s - a)}; return r; }
PBC_FN pbc_dd pbc_two_prod(double a, double b) { PBC_EXACT double p = a * b; pbc_dd r = {p, fma(a, b, -p)}; return r; }
PBC_FN pbc_dd pbc_add(pbc_dd a, pbc_dd b)
{
    PBC_EXACT
    pbc_dd s = pbc_two_sum(a.h, b.h), t = pbc_two_sum(a.l, b.l);
    s.l += t.h; s = pbc_fast_two_sum(s.h, s.l); s.l += t.l; return pbc_fast_two_sum(s.h, s.l);
}
PBC_FN pbc_dd pbc_add_d(pbc_dd a, double b) { PBC_EXACT pbc_dd s = pbc_two_sum(a.h, b); s.l += a.l; return pbc_fast_two_sum(s.h, s.l); }
PBC_FN pbc_dd pbc_mul(pbc_dd a, pbc_dd b) { PBC_EXACT pbc_dd p = pbc_two_prod(a.h, b.h); p.l += a.h * b.l + a.l * b.h; return pbc_fast_two_sum(p.h, p.l); }
PBC_FN pbc_dd pbc_mul_d(pbc_dd a, double b) { PBC_EXACT pbc_dd p = pbc_two_prod(a.h, b); p.l = fma(a.l, b, p.l); return pbc_fast_two_sum(p.h, p.l); }
// finite x with |x| < 700: callers keep the libm route for anything else
PBC_FN double pow_base_cr(double x)
{
    PBC_EXACT
    const double LB1 = 0x1.ffffe96df507cp-1, LB2 = -0x1.24478f1c228a3p-56, LB3 = -0x1.77697a7daa413p-113;      // ln(2.71828) as the double the reference passes
    const double L2H = 0x1.62e42fefa3000p-1, L2M = 0x1.3de6af278ece6p-42, L2L = 0x1.f97b57a079a19p-103, INVLN2 = 0x1.71547652b82fep+0;
    // v = x * ln b
    pbc_dd v = pbc_two_prod(x, LB1);
    pbc_dd v2 = pbc_two_prod(x, LB2);
    v = pbc_add(v, v2);
    v = pbc_add_d(v, x * LB3);
    const double kd = nearbyint(v.h * INVLN2);
    // r = v - k ln2
    pbc_dd r = pbc_add_d(v, -kd * L2H);                       // exact product
    pbc_dd t = pbc_two_prod(-kd, L2M);
    r = pbc_add(r, t);
    r = pbc_add_d(r, -kd * L2L);
    r.h *= 0.0625; r.l *= 0.0625;                             // / 16, exact
    // exp(r) = 1 + r (1 + r/2 (1 + r/3 ( ... )))
    pbc_dd s = {1.0, 0.0};
    for (int n = 18; n >= 1; n--) {
        s = pbc_mul(s, r);
        // divide by n: multiply by 1/n in double-double
        const double inv = 1.0 / (double)n;
        pbc_dd q = pbc_mul_d(s, inv);
        // one correction of the quotient: s - q * n
        pbc_dd back = pbc_mul_d(q, (double)n);
        const double err = ((s.h - back.h) - back.l) + s.l;
        q = pbc_add_d(q, err * inv);
        s = pbc_add_d(q, 1.0);
    }
    for (int i = 0; i < 4; i++) s = pbc_mul(s, s);
    return ldexp(s.h, (int)kd);                               // (s.l only decides ties, which a transcendental value never is)
}
