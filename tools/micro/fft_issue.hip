// Microbenchmark: how fast can a SIMD issue the in-register FFT instruction mix?
// Each lane keeps N complex values in registers and runs forward+inverse codelets K times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../torchpiv_amd/csrc/fft_inreg.hpp"
using namespace tpiv;

template <int N, int OCC>
__global__ __launch_bounds__(64, OCC) void k(const cf* in, cf* out, int iters) {
    cf x[N];
    const cf* p = in + ((size_t)blockIdx.x * 64 + threadIdx.x) * N;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = p[i];
    for (int it = 0; it < iters; ++it) {
        fft_inreg<N, 1>(x);
        cf t[N];
        static_for<0, N>([&](auto kc) TPIV_LAMBDA_INLINE { constexpr int q = decltype(kc)::value; t[q] = x[FFT_POS<q, N>]; });
        fft_inreg<N, -1>(t);
        static_for<0, N>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int q = decltype(kc)::value;
            x[q].x = t[FFT_POS<q, N>].x * (1.0f / N);
            x[q].y = t[FFT_POS<q, N>].y * (1.0f / N);
        });
    }
    cf* q = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * N;
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = x[i];
}

template <int N, int OCC>
void run(int wg_per_cu, int iters, double valu_per_iter) {
    const int blocks = 256 * wg_per_cu;
    size_t n = (size_t)blocks * 64 * N;
    cf *in, *out;
    hipMalloc(&in, n * sizeof(cf)); hipMalloc(&out, n * sizeof(cf));
    std::vector<cf> h(n);
    for (size_t i = 0; i < n; ++i) { h[i].x = (float)rand() / RAND_MAX - 0.5f; h[i].y = (float)rand() / RAND_MAX - 0.5f; }
    hipMemcpy(in, h.data(), n * sizeof(cf), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, OCC>), dim3(blocks), dim3(64), 0, 0, in, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, OCC>), dim3(blocks), dim3(64), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * iters * valu_per_iter;          // wave-level VALU instructions
    double rate = instr / (ms * 1e-3);
    printf("N=%d occ=%d wg/CU=%d: %.3f ms, %.1f G wave-instr/s, %.2f cycles/instr/SIMD at 2.4 GHz (%.0f%% of 1-per-2-cycles)\n",
           N, OCC, wg_per_cu, ms, rate * 1e-9, 1024 * 2.4e9 / rate, 100.0 * rate / (1024 * 1.2e9));
    hipFree(in); hipFree(out);
}

int main() {
    // VALU instructions per loop iteration, counted from the ISA (tools/micro/count.sh): fwd + inv + scale
    run<32, 2>(4, 2000, 431 + 431 + 64);
    run<32, 2>(8, 2000, 431 + 431 + 64);
    run<32, 4>(16, 2000, 431 + 431 + 64);
    run<64, 2>(4, 1000, 1057 + 1057 + 128);
    run<64, 2>(8, 1000, 1057 + 1057 + 128);
    return 0;
}
