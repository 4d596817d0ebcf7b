// LDS read throughput microbenchmark (scratch): CU cycles per wave64 ds_read instruction on gfx950, 8 loads in flight per wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 16
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(unsigned long long* out, int n, int stride_bytes, unsigned mis)
{
    extern __shared__ double sm[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(lane * stride_bytes + wave * 64) & 0xfff8;
    if (MODE == 3) addr = wave * 64;                       // broadcast: all lanes one address
    const unsigned addr2 = (addr & 0xfff0) + mis * 8 * (lane & 1);
    double a0 = 0, a1 = 0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (MODE == 2) {
                double2 y0, y1, y2, y3, y4, y5, y6, y7;
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:2048\n ds_read_b128 %2, %8 offset:4096\n ds_read_b128 %3, %8 offset:6144\n"
                             "ds_read_b128 %4, %8 offset:8192\n ds_read_b128 %5, %8 offset:10240\n ds_read_b128 %6, %8 offset:12288\n ds_read_b128 %7, %8 offset:14336\n s_waitcnt lgkmcnt(0)"
                             : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3), "=v"(y4), "=v"(y5), "=v"(y6), "=v"(y7) : "v"(addr2));
                a0 += y0.x + y2.x + y4.y + y6.x; a1 += y1.y + y3.x + y5.x + y7.y;
            } else {
                double x0, x1, x2, x3, x4, x5, x6, x7;
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:2048\n ds_read_b64 %2, %8 offset:4096\n ds_read_b64 %3, %8 offset:6144\n"
                             "ds_read_b64 %4, %8 offset:8192\n ds_read_b64 %5, %8 offset:10240\n ds_read_b64 %6, %8 offset:12288\n ds_read_b64 %7, %8 offset:14336\n s_waitcnt lgkmcnt(0)"
                             : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(addr));
                a0 += x0 + x2 + x4 + x6; a1 += x1 + x3 + x5 + x7;
            }
        }
    }
    if (a0 + a1 == 1.2345) out[0] = 1;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(MODE, stride, misal, label) { hipFuncSetAttribute((const void*)k_lds<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    k_lds<MODE><<<256, 1024, 160 * 1024>>>(d, 20, stride, misal); hipDeviceSynchronize(); hipEventRecord(e0); k_lds<MODE><<<256, 1024, 160 * 1024>>>(d, 1000, stride, misal); hipEventRecord(e1); hipDeviceSynchronize(); \
    float ms; hipEventElapsedTime(&ms, e0, e1); double instr = 16.0 * 1000.0 * REP * 8; \
    printf("%-40s %.3f ms, %.2f CU-cycles (2.4 GHz) per wave64 LDS instruction, %.0f B/clk/CU\n", label, ms, ms * 1e-3 * 2.4e9 / instr, (MODE == 2 ? 1024.0 : 512.0) / (ms * 1e-3 * 2.4e9 / instr)); }
    RUN(1, 8, 0, "ds_read_b64 stride 8 B")
    RUN(1, 24, 0, "ds_read_b64 stride 24 B")
    RUN(1, 120, 0, "ds_read_b64 stride 120 B")
    RUN(1, 16, 0, "ds_read_b64 stride 16 B (2-way)")
    RUN(3, 0, 0, "ds_read_b64 broadcast")
    RUN(2, 16, 0, "ds_read_b128 stride 16 B")
    RUN(2, 48, 0, "ds_read_b128 stride 48 B")
    RUN(2, 32, 0, "ds_read_b128 stride 32 B")
    RUN(2, 16, 1, "ds_read_b128 stride 16 B, odd lanes +8")
    RUN(2, 48, 1, "ds_read_b128 stride 48 B, odd lanes +8")
    return 0;
}
