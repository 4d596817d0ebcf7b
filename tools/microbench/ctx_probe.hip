// ctx_probe.hip — what do FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for a kernel that touches NO global memory?
// k_svr_dense carries ~0.16 GB per direction and launch under --pmc that does not scale with the batch, with its score stores
// compiled out as well (tools/exp/wr_probe2.sh).  Hypothesis: it is the counter collection itself - the profiler samples the counters by
// pre-empting the running waves, and a wave-context save / restore of a chip fully occupied by 1,024-thread workgroups at 128 VGPRs and
// ~146 KB of LDS is 256 CUs x (16 waves x 128 VGPRs x 256 B + 146 KB) = 0.17 GB in each direction.  This program launches "hold" kernels of
// the same footprint that only spin (no loads, no stores: one never-taken store keeps the result alive), in four shapes:
//     A  1,254 workgroups x 1,024 threads, 146 KB LDS, ~8 ms      (k_svr_dense's footprint and duration on the bench batch)
//     B  the same, ~0.3 ms                                         (does the constant need a long kernel?)
//     C  1,254 workgroups x 1,024 threads, no LDS, ~8 ms           (registers only)
//     D  ONE workgroup, 146 KB LDS, ~8 ms                          (one CU occupied)
//   hipcc --offload-arch=gfx950 -O2 -o ctx_probe ctx_probe.hip ; tools/exp/ctx_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>

#define NACC 28                                              // ~60 live VGPRs: no scratch (checked: private_segment_fixed_size 0)
template <bool LDS>
__global__ __launch_bounds__(1024) void hold(int iters, double seed, double* sink)
{
    extern __shared__ double lds[];
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = seed + i + threadIdx.x;
    if (LDS) lds[threadIdx.x] = seed;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        const double x = LDS ? lds[(threadIdx.x + it) & 1023] : 0.5;
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = fma(acc[i], 1.0000001, x);
        if (LDS && (it & 63) == 0) { __syncthreads(); lds[threadIdx.x] = acc[0]; __syncthreads(); }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    if (s == 12345.678) *sink = s;                             // never true: no store is executed
}

int main()
{
    double* sink;
    if (hipMalloc(&sink, 8) != hipSuccess) return 1;
    const size_t lds = 146 * 1024;
    (void)hipFuncSetAttribute((const void*)hold<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char* name; int wgs; bool lds; int iters; } shapes[] = {{"A_1254wg_lds_long", 1254, true, 6500}, {"B_1254wg_lds_short", 1254, true, 250},
                                                                             {"C_1254wg_nolds_long", 1254, false, 6500}, {"D_1wg_lds_long", 1, true, 26000}};
    for (int rep = 0; rep < 3; rep++)
        for (auto& s : shapes) {
            (void)hipEventRecord(e0, 0);
            if (s.lds) hipLaunchKernelGGL(hold<true>, dim3(s.wgs), dim3(1024), lds, 0, s.iters, 1.0, sink);
            else hipLaunchKernelGGL(hold<false>, dim3(s.wgs), dim3(1024), 0, 0, s.iters, 1.0, sink);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s: %.3f ms\n", s.name, ms);
        }
    return 0;
}
