// fp64 rates on MI355X: v_fma_f64 against v_mfma_f64_16x16x4_f64 (predictor as a banded GEMM).
// Per wavefront and iteration: 32 independent v_fma_f64 (4096 flop) or 8 MFMA (8 x 2048 flop), 4 accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_rate.hip -o tools/micro/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(double* out, int iters) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001 + i;
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const double a = v[3], b = v[5];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k & 15] = __builtin_fma(v[k & 15], 1.0001, 0.5);
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m & 3], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(int waves_per_simd, int iters, double* out) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, out, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * 4096);
    const int iters = 100000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    for (int w : {1, 2, 4}) {
        const double tv = run<0>(w, iters, out), tm = run<1>(w, iters, out);
        const double waves = (double)prop.multiProcessorCount * 4 * w;
        printf("%d wave(s)/SIMD: v_fma_f64 %.1f TFLOP/s   v_mfma_f64_16x16x4 %.1f TFLOP/s\n", w,
               waves * iters * 32 * 128 / tv * 1e-12, waves * iters * 8 * 2048 / tm * 1e-12);
    }
    return 0;
}
