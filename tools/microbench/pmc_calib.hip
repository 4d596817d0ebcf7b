// pmc_calib.hip — what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for the access widths of this repository's kernels.
// MI355X_MICROARCH.md (HBM section) calibrates the counters for 16-byte-per-lane streaming accesses only (FETCH_SIZE reads half the bytes,
// WRITE_SIZE reads them exactly) and asks for a calibration on a known byte count for any other width: k_svr_dense reads 8-byte records and
// writes 8-byte scores in 456-byte row pieces, consecutive lanes on consecutive doubles.
//   hipcc --offload-arch=gfx950 -O2 -o pmc_calib pmc_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -o p -- ./pmc_calib ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -o p -- ./pmc_calib
// Every kernel moves exactly N * 8 bytes (N = 2^26 doubles = 512 MiB: twice the Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void w8(double* o, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = (double)i; }
__global__ void w16(double2* o, size_t n2) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) o[i] = make_double2((double)i, 1.0); }
// rows of 57 doubles (456 bytes) written as whole rows by consecutive lanes, every other row skipped (the other strand's row): k_svr_dense's store shape
__global__ void w8_rows(double* o, size_t n_rows)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_rows * 57; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / 57, a = e - r * 57;
        o[(2 * r) * 57 + a] = (double)e;
    }
}
// k_svr_dense's epilogue exactly: a workgroup of 1024 threads owns 27 positions x 9 capture sizes of ONE strand (243 rows of 57 doubles inside a
// 27 x 8,208-byte block whose other rows belong to the other strand's tile), element e = (size * 27 + position) * 57 + pair, thread t takes
// e = t, t + 1024, ... four at a time: consecutive lanes on consecutive doubles of a row, consecutive rows of a wave 8,208 bytes apart
__global__ void w8_tile(double* o, int n_tiles, int strand)
{
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t base = (size_t)tile * 27 * 9 * 2 * 57;
        for (int e = threadIdx.x; e < 243 * 57; e += 1024) {
            const int it = e / 57, a = e - it * 57, kc = it / 27, pl = it - kc * 27;
            o[base + ((size_t)(pl * 9 + kc) * 2 + strand) * 57 + a] = (double)e;
        }
    }
}
__global__ void r8(const double* in, size_t n, double* sink)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 12345.678) *sink = s;
}
__global__ void r16(const double2* in, size_t n2, double* sink)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { const double2 v = in[i]; s += v.x + v.y; }
    if (s == 12345.678) *sink = s;
}
int main()
{
    const size_t N = (size_t)1 << 26;
    double *a, *b, *sink;
    if (hipMalloc(&a, N * 8 * 2) != hipSuccess || hipMalloc(&b, N * 8) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
    hipMemset(a, 0, N * 16); hipMemset(b, 0, N * 8);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(w8, dim3(4096), dim3(256), 0, 0, b, N);
        hipLaunchKernelGGL(w16, dim3(4096), dim3(256), 0, 0, (double2*)b, N / 2);
        hipLaunchKernelGGL(w8_rows, dim3(4096), dim3(256), 0, 0, a, N / 57);
        hipLaunchKernelGGL(w8_tile, dim3(1254), dim3(1024), 0, 0, a, (int)(N / (27 * 9 * 57)), 0);      // '+' tiles: N * 8 bytes again (half of the rows of a 2 N block)
        hipLaunchKernelGGL(w8_tile, dim3(1254), dim3(1024), 0, 0, a, (int)(N / (27 * 9 * 57)), 1);      // '-' tiles
        hipLaunchKernelGGL(r8, dim3(4096), dim3(256), 0, 0, b, N, sink);
        hipLaunchKernelGGL(r16, dim3(4096), dim3(256), 0, 0, (const double2*)b, N / 2, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("bytes per kernel: %zu (w8_rows: %zu)\n", N * 8, (N / 57) * 57 * 8);
    return 0;
}
