// tools/micro/kernarg_gap.hip -- does a 3.5 KB by-value kernel argument lengthen the gap between dependent launches?
// hipcc --offload-arch=gfx950 -O2 tools/micro/kernarg_gap.hip -o /tmp/kernarg_gap && /tmp/kernarg_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int v[880]; };      // 3520 bytes, like OrbGeom
struct Small { int v[8]; };
__global__ void k_big(Big b, int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += b.v[3]; }
__global__ void k_small(Small b, int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += b.v[3]; }
int main()
{
    int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    Big B{}; Small S{}; B.v[3] = 1; S.v[3] = 1;
    hipStream_t st; hipStreamCreate(&st);
    const int N = 2000;
    for (int rep = 0; rep < 3; rep++) {
        for (int which = 0; which < 2; which++) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (which) hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, st, B, d);
                else hipLaunchKernelGGL(k_small, dim3(2048), dim3(256), 0, st, S, d);
            }
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / N;
            std::printf("%s argument: %.2f us per dependent launch of 2048 workgroups\n", which ? "3520-byte" : "32-byte", us);
        }
    }
    return 0;
}
