// fmt_g6.h — printf("%g") (6 significant digits, the default std::ostream formatting of a double: /root/reference/mipgen.cpp:774) and
// decimal integers, usable on the device.  Correctly rounded like glibc's printf: the value is scaled by a power of ten held as a
// double-double (pow10_dd.h, exact to ~106 bits) with an error-free product, and rounded half-to-even on the exact tie only.
#pragma once
#include <stdint.h>
#include <math.h>
#include "pow10_dd.h"

#if defined(__HIPCC__)
#define FMT_HD __host__ __device__ __forceinline__
#else
#define FMT_HD static inline
#endif

struct Pow10DD { double hi, lo; };

// v * 10^k rounded to the nearest integer (ties to even); v > 0 finite, result < 2^53
FMT_HD double fmt_scale_round(double v, int k, const Pow10DD* tab)
{
    double s = v, slo = 0.0;
    int left = k;
    while (left != 0) {                                   // 10^k in one or two table factors (|k| can exceed the double range for denormals)
        int step = left > 300 ? 300 : (left < -300 ? -300 : left);
        const Pow10DD t = tab[step - POW10_DD_KMIN];
        // (s + slo) * (t.hi + t.lo), error-free on the leading product
        const double p = s * t.hi;
        double e = fma(s, t.hi, -p);
        e = fma(s, t.lo, e);
        e = fma(slo, t.hi, e);
        const double n = p + e;
        slo = e - (n - p);
        s = n;
        left -= step;
    }
    double r = rint(s);
    if (k < 0 && k >= -22) {
        // 10^k is not a binary fraction, so the product above cannot see an exact tie; 10^|k| is exact: compare v with (r +- 1/2) * 10^|k|
        // through error-free products (r < 2^20, so r +- 1/2 is exact)
        const double P = tab[-k - POW10_DD_KMIN].hi;
        const double hi = (r + 0.5) * P, hie = fma(r + 0.5, P, -hi);
        if (v > hi || (v == hi && hie < 0.0)) r += 1.0;
        else if (v == hi && hie == 0.0) { if (fmod(r, 2.0) != 0.0) r += 1.0; }
        else {
            const double lo = (r - 0.5) * P, loe = fma(r - 0.5, P, -lo);
            if (v < lo || (v == lo && loe > 0.0)) r -= 1.0;
            else if (v == lo && loe == 0.0) { if (fmod(r, 2.0) != 0.0) r -= 1.0; }
        }
        return r;
    }
    const double t = s - r;                               // exact
    if (t == 0.5) { if (slo > 0.0 || (slo == 0.0 && fmod(r, 2.0) != 0.0)) r += 1.0; }
    else if (t == -0.5) { if (slo < 0.0 || (slo == 0.0 && fmod(r, 2.0) != 0.0)) r -= 1.0; }
    return r;
}

FMT_HD int fmt_uint(uint64_t v, char* out)               // decimal digits, no sign; returns the length
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + (int)(v % 10)); v /= 10; } while (v);
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    return n;
}
FMT_HD int fmt_int(int64_t v, char* out)
{
    if (v < 0) { out[0] = '-'; return 1 + fmt_uint((uint64_t)(-v), out + 1); }
    return fmt_uint((uint64_t)v, out);
}

// "%g" of v into out (at most 16 bytes: "-1.23456e-308"); NaN prints "-nan" (the reference's NaNs come from inf - inf and inf / inf on
// x86, which set the sign bit; the front end prints every NaN this way)
FMT_HD int fmt_g6(double v, char* out, const Pow10DD* tab)
{
    int n = 0;
    if (v != v) { out[0] = '-'; out[1] = 'n'; out[2] = 'a'; out[3] = 'n'; return 4; }
    if (signbit(v)) { out[n++] = '-'; v = -v; }
    if (v == 0.0) { out[n++] = '0'; return n; }
    if (isinf(v)) { out[n++] = 'i'; out[n++] = 'n'; out[n++] = 'f'; return n; }
    int e2;
    (void)frexp(v, &e2);                                   // v = m * 2^e2, 0.5 <= m < 1
    int e10 = (int)floor((double)(e2 - 1) * 0.30102999566398120);   // floor(log10 v) or one less
    double N = fmt_scale_round(v, 5 - e10, tab);
    if (N >= 1000000.0) { e10 += 1; N = fmt_scale_round(v, 5 - e10, tab); }
    if (N >= 1000000.0) { e10 += 1; N = 100000.0; }       // 999999.5.. rounded up to 10^6: the digits are 100000 one decade higher
    if (N < 100000.0) { e10 -= 1; N = fmt_scale_round(v, 5 - e10, tab); if (N >= 1000000.0) { e10 += 1; N = 100000.0; } }
    uint32_t d = (uint32_t)N;
    char dig[6];
    for (int i = 5; i >= 0; i--) { dig[i] = (char)('0' + (int)(d % 10)); d /= 10; }
    int nd = 6;
    while (nd > 1 && dig[nd - 1] == '0') nd--;            // %g strips trailing zeros
    if (e10 < -4 || e10 >= 6) {
        out[n++] = dig[0];
        if (nd > 1) { out[n++] = '.'; for (int i = 1; i < nd; i++) out[n++] = dig[i]; }
        out[n++] = 'e';
        int x = e10;
        if (x < 0) { out[n++] = '-'; x = -x; } else out[n++] = '+';
        if (x >= 100) { out[n++] = (char)('0' + x / 100); x %= 100; out[n++] = (char)('0' + x / 10); out[n++] = (char)('0' + x % 10); }
        else { out[n++] = (char)('0' + x / 10); out[n++] = (char)('0' + x % 10); }
    } else if (e10 >= 0) {
        for (int i = 0; i <= e10; i++) out[n++] = i < nd ? dig[i] : '0';
        if (nd > e10 + 1) { out[n++] = '.'; for (int i = e10 + 1; i < nd; i++) out[n++] = dig[i]; }
    } else {
        out[n++] = '0'; out[n++] = '.';
        for (int i = 0; i < -e10 - 1; i++) out[n++] = '0';
        for (int i = 0; i < nd; i++) out[n++] = dig[i];
    }
    return n;
}
