// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, found empirically: which (row, col) of D does
// (lane, register) hold when lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* a, const double* b, double* d) {
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = acc[i];
}
int main() {
    double ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + 0.37 * l + 0.011 * l * l; hb[l] = 2.0 + 0.53 * l - 0.007 * l * l; }
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 2048);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 2048, hipMemcpyDeviceToHost);
    double ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += ha[kk * 16 + i] * hb[kk * 16 + j]; ref[i][j] = s; }
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int fi = -1, fj = -1, n = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(ref[i][j] - hd[l * 4 + r]) < 1e-9 * fabs(ref[i][j])) { fi = i; fj = j; ++n; }
        if (n != 1) { ++bad; continue; }
        if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> D[%2d][%2d]\n", l, r, fi, fj);
        if (fi != 4 * (l / 16) + r || fj != l % 16) ++bad;
    }
    printf("layout D[4*(l/16)+r][l%%16]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return 0;
}
