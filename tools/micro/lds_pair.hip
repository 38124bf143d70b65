// LDS: ds_read2 / ds_write2 against two single accesses (32- and 64-bit), for the two access patterns of the tile kernels'
// transposes through a padded plane: lane = row reading along its row, lane = column reading down its column.
//   hipcc --offload-arch=gfx950 -O3 -w lds_pair.hip -o lds_pair
#include <hip/hip_runtime.h>
#include <cstdio>

// P = pitch in dwords.  COLS: lane = column (element e of the lane at base + e * P dwords), else lane = row (base + e).
template <int MODE, int P, bool COLS>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    __shared__ float plane[64 * 67];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 67; i += 64) plane[i] = (float)i;
    __syncthreads();
    constexpr int ST = COLS ? P : 1;                     // dwords between consecutive elements of a lane
    const unsigned base = (unsigned)(uintptr_t)(COLS ? plane + (lane & 31) : plane + lane * P);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float a[32];
        if constexpr (MODE == 0) {            // 32 x ds_read_b32
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[j]) : "v"(base), "n"(j * ST * 4));
        } else if constexpr (MODE == 1) {     // 16 x ds_read2_b32 (groups of 8 elements share a base: 8-bit dword offsets)
            constexpr int G = ST * 7 <= 255 ? 8 : 2;          // elements per base register (8-bit dword offsets)
#pragma unroll
            for (int g = 0; g < 32 / G; ++g) {
                const unsigned bg = base + g * G * ST * 4;
#pragma unroll
                for (int r = 0; r < G; r += 2)
                    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(*(double*)&a[G * g + r]) : "v"(bg), "n"(r * ST), "n"((r + 1) * ST));
            }
        } else if constexpr (MODE == 2) {     // 32 x ds_write_b32
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base), "v"(acc), "n"(j * ST * 4) : "memory");
        } else if constexpr (MODE == 3) {     // 16 x ds_write2_b32
            constexpr int G = ST * 7 <= 255 ? 8 : 2;
#pragma unroll
            for (int g = 0; g < 32 / G; ++g) {
                const unsigned bg = base + g * G * ST * 4;
#pragma unroll
                for (int r = 0; r < G; r += 2)
                    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(bg), "v"(acc), "v"(acc), "n"(r * ST), "n"((r + 1) * ST) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (MODE < 2) {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += a[j];
        } else {
            acc += 1.f;
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE, int P, bool COLS>
void run(float* d, const char* name) {
    const int iters = 4000, blocks = 2048;       // 2 wavefronts per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, P, COLS>), dim3(blocks), dim3(64), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, P, COLS>), dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * 64 * 32 * 4;
    printf("pitch %2d %-14s %-18s %.3f ms  %5.0f B/clk/CU at 2.4 GHz\n", P, COLS ? "lane = column" : "lane = row", name, ms,
           bytes / (ms * 1e-3) / 256 / 2.4e9);
}

template <int P, bool COLS>
void all(float* d) {
    run<0, P, COLS>(d, "32 ds_read_b32"); run<1, P, COLS>(d, "16 ds_read2_b32");
    run<2, P, COLS>(d, "32 ds_write_b32"); run<3, P, COLS>(d, "16 ds_write2_b32");
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 64 * sizeof(float));
    all<33, false>(d); all<33, true>(d); all<65, false>(d); all<65, true>(d);
    return 0;
}
