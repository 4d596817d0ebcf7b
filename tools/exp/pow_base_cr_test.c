#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../mipgen_amd/csrc/pow_base_cr.h"   /* gcc -O2 -ffp-contract=off -o /tmp/pbc tools/exp/pow_base_cr_test.c -lm && /tmp/pbc */
int main() {
    srand48(12345);
    long n = 2000000, same = 0, off1 = 0, more = 0;
    for (long i = 0; i < n; i++) {
        double x = (drand48() - 0.5) * 90.0;          // [-45, 45]
        if (i % 4 == 0) x = 36.0 + drand48() * 2.0;   // the saturation edge
        double a = pow(2.71828, x), b = pow_base_cr(x);
        int64_t ia, ib; memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
        long d = labs(ia - ib);
        if (d == 0) same++; else if (d == 1) off1++; else { more++; if (more < 5) printf("x=%.17g glibc %a cr %a\n", x, a, b); }
    }
    printf("n %ld identical %ld (%.4f %%) 1 ulp %ld more %ld\n", n, same, 100.0 * same / n, off1, more);
    return 0;
}
