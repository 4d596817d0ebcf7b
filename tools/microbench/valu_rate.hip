// instruction issue-rate microbenchmark (scratch): cycles per wave64 instruction for a few VALU opcodes on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
#define DEF(name, body) __global__ void name(unsigned long long* out, int n) { \
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11; double x = threadIdx.x, y = 1.0001, z = 0.5; \
    unsigned long long q = threadIdx.x; (void)q; \
    unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int i = 0; i < n; i++) { _Pragma("unroll") for (int r = 0; r < REP; r++) { body } } \
    unsigned long long t1 = __builtin_readcyclecounter(); \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0; if (a + b + c + d == 12345 && x + y + z == 1.5 && q == 77) out[1000] = a; }
DEF(k_add, asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEF(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %0\n v_mad_u32_u24 %2, %2, %3, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEF(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEF(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %1, %0" : "+v"(q), "+v"(a), "+v"(b) : : "vcc");)
DEF(k_fma64, asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %2, %2, %1, %0" : "+v"(x), "+v"(y), "+v"(z));)
DEF(k_add64, asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %2, %2, %1" : "+v"(x), "+v"(y), "+v"(z));)
DEF(k_max64, asm volatile("v_max_f64 %0, %0, %1\n v_max_f64 %2, %2, %1" : "+v"(x), "+v"(y), "+v"(z));)
DEF(k_rndne, asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1" : "+v"(x), "+v"(z));)
DEF(k_cvt, asm volatile("v_cvt_i32_f64 %0, %1\n v_cvt_i32_f64 %2, %3" : "+v"(a), "+v"(x), "+v"(c), "+v"(z));)
DEF(k_lshladd, asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %2, %2, 3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEF(k_and, asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
DEF(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5" : : "v"(a), "v"(b) : "s20", "s21");)
DEF(k_mov_dpp, asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
int main() {
    unsigned long long* d; hipMalloc(&d, 8 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(k, waves) { k<<<1024, 64 * waves>>>(d, 200); hipDeviceSynchronize(); hipEventRecord(e0); k<<<1024, 64 * waves>>>(d, 4000); hipEventRecord(e1); hipDeviceSynchronize(); \
    float ms; hipEventElapsedTime(&ms, e0, e1); double instr = 1024.0 * waves * 4000.0 * REP * 2; \
    printf("%-12s %2d waves/block: %.3f ms, %.2f SIMD-cycles (2.4 GHz) per wave-instruction\n", #k, waves, ms, ms * 1e-3 * 2.4e9 * 1024.0 / instr); }
    RUN(k_add, 4) RUN(k_add, 16) RUN(k_mad24, 16) RUN(k_mullo, 16) RUN(k_mad64, 16) RUN(k_fma64, 4) RUN(k_fma64, 16) RUN(k_add64, 16) RUN(k_max64, 16) RUN(k_rndne, 16) RUN(k_cvt, 16)
    RUN(k_lshladd, 16) RUN(k_and, 16) RUN(k_readlane, 16) RUN(k_mov_dpp, 16)
    return 0;
}
