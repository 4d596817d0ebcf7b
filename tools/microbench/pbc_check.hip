// pbc_check.hip — pow_base_cr (mipgen_amd/csrc/pow_base_cr.h) on the DEVICE against the host's glibc pow(2.71828, x): the same doubles?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pbc_check tools/microbench/pbc_check.hip && /tmp/pbc_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../mipgen_amd/csrc/pow_base_cr.h"
__global__ void k(const double* x, double* y, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = pow_base_cr(x[i]); }
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), y(n);
    srand48(777);
    for (int i = 0; i < n; i++) x[i] = i % 2 ? 36.5 + drand48() * 1.2 : (drand48() - 0.5) * 90.0;
    double *dx, *dy;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dy, n * 8) != hipSuccess) return 1;
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    (void)hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
    long same = 0, one = 0, more = 0, host_same = 0;
    for (int i = 0; i < n; i++) {
        const double g = pow(2.71828, x[i]), h = pow_base_cr(x[i]);
        long long a, b, c; memcpy(&a, &g, 8); memcpy(&b, &y[i], 8); memcpy(&c, &h, 8);
        const long long d = llabs(a - b);
        if (d == 0) same++; else if (d == 1) one++; else { more++; if (more < 4) printf("x %.17g glibc %a device %a host-cr %a\n", x[i], g, y[i], h); }
        host_same += b == c;
    }
    printf("n %d: device == glibc %ld (%.4f %%), 1 ulp %ld, more %ld; device == host pow_base_cr %ld\n", n, same, 100.0 * same / n, one, more, host_same);
    return 0;
}
