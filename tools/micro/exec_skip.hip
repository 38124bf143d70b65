// Does a wave64 VALU instruction get cheaper when whole 16- / 32-lane groups are masked off by EXEC?
// (float64 fma chain; lanes < ACTIVE run the loop, the rest skip it)   hipcc --offload-arch=gfx950 -O3 exec_skip.hip -o exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ACTIVE>
__global__ __launch_bounds__(64) void k(double* out, int iters) {
    const int lane = threadIdx.x;
    double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const double m = 1.0000001, c = 1e-9;
    if (lane < ACTIVE) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
                a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int ACTIVE>
void run(double* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<ACTIVE>, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ACTIVE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * 64;      // fma per wave
    printf("active lanes %2d, %d waves: %.3f ms, %.2f ns per wave-instruction\n", ACTIVE, blocks, ms, ms * 1e6 / inst);
}

int main() {
    double* d;
    hipMalloc(&d, 4096 * 64 * sizeof(double));
    for (int blocks : {1024, 2048}) {      // 1 / 2 wavefronts per SIMD
        run<64>(d, blocks); run<48>(d, blocks); run<33>(d, blocks); run<32>(d, blocks); run<16>(d, blocks);
    }
    return 0;
}
