// hipMalloc / hipFree cost of large device buffers on a cold and a warm device (seconds per call):
//   hipcc --offload-arch=gfx950 -O2 -o alloc_cost alloc_cost.hip && ./alloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const double gb[] = {25.8, 12.9, 6.7, 0.14};
    for (int round = 0; round < 3; round++) {
        std::vector<void*> p;
        for (double g : gb) {
            void* q = nullptr;
            const size_t bytes = (size_t)(g * 1e9);
            double t0 = now();
            hipError_t e = hipMalloc(&q, bytes);
            double t1 = now();
            hipMemsetAsync(q, 0xFF, bytes, 0);
            hipDeviceSynchronize();
            double t2 = now();
            printf("round %d: hipMalloc(%.1f GB) %.3f s (%s), first memset %.3f s\n", round, g, t1 - t0, hipGetErrorString(e), t2 - t1);
            p.push_back(q);
        }
        for (size_t i = 0; i < p.size(); i++) {
            double t0 = now();
            hipFree(p[i]);
            printf("round %d: hipFree(%.1f GB) %.3f s\n", round, gb[i], now() - t0);
        }
        const double gb2[] = {4.5, 8.6, 8.6, 1.07};
        std::vector<void*> r;
        for (double g : gb2) {
            void* q = nullptr;
            double t0 = now();
            hipMalloc(&q, (size_t)(g * 1e9));
            printf("round %d: after the frees, hipMalloc(%.1f GB) %.3f s\n", round, g, now() - t0);
            r.push_back(q);
        }
        for (void* q : r) hipFree(q);
    }
    return 0;
}
