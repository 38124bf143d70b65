// Does fp32 MFMA work co-execute with fp32 VALU work on MI355X, and at what price?  (VERDICT r1 item 8:
// 16-/8-point DFTs as v_mfma_f32_16x16x4_f32 beside the VALU staging / peak work of the small-tile kernels.)
// Three loops per wavefront, timed with hipEvents at 1, 2 and 4 wavefronts per SIMD:
//   valu   64 independent v_fma_f32 per iteration
//   mfma   8 v_mfma_f32_16x16x4_f32 per iteration (4 independent accumulators)
//   both   the two bodies interleaved (8 fma after every MFMA)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_coexec.hip -o tools/micro/mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = v[3], b = v[5];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MODE != 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
            if (MODE != 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[(m * 8 + k) & 15] = __builtin_fmaf(v[(m * 8 + k) & 15], 1.0001f, 0.5f);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(int waves_per_simd, int iters, float* out) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * waves_per_simd;      // 256 threads = 4 wavefronts = one per SIMD
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
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 4096);
    const int iters = 200000;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("nominal clock %.0f MHz; per iteration: 64 v_fma_f32 and/or 8 v_mfma_f32_16x16x4_f32 per wavefront\n", clk_khz / 1e3);
    for (int w : {1, 2, 4}) {
        const double tv = run<0>(w, iters, out), tm = run<1>(w, iters, out), tb = run<2>(w, iters, out);
        const double cyc = clk_khz * 1e3 / iters / w;      // cycles per iteration per wavefront slot of a SIMD
        printf("%d wave(s)/SIMD: valu %.1f  mfma %.1f  both %.1f cycles per (iteration x wavefront)   both / (valu + mfma) = %.2f,  "
               "both / max = %.2f\n", w, tv * cyc, tm * cyc, tb * cyc, tb / (tv + tm), tb / (tv > tm ? tv : tm));
    }
    return 0;
}
