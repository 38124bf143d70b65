// Generic-size PIV tile kernel: any window size 2..256 (the reference accepts any integer,
// ControlsWidgets.py:88-100, and multipass scales such as 1.5 give sizes like 42 or 28).
// Not on the hot path (the power-of-two sizes run xcorr_tile.hpp): one 256-thread workgroup per
// window, the complex tile lives in a global (L2-resident) scratch, the transforms are plain DFTs
// with a twiddle table in LDS.  Staging, peak search and hand-off to finalize_kernel follow the
// same reference semantics as xcorr_tile.hpp (PIVbackend.py:147-216, 249-257, 346-422, 459-520).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_inreg.hpp"
#include "piv_kernels.h"

namespace tpiv {

namespace {

constexpr int GT = 256;      // threads per workgroup

__device__ __forceinline__ float fetch_clamped_g(const uint8_t* __restrict__ f, long long q, int HW) {
    q = q < 0 ? 0 : (q > (long long)(HW - 1) ? (long long)(HW - 1) : q);      // B:177-180, B:214
    return (float)f[q];
}

__device__ __forceinline__ int f2i_sat_g(float v) {
    v = fminf(fmaxf(v, -1073741824.f), 1073741824.f);
    return (int)v;
}

// One bilinear sample exactly as PIVbackend.py:162-193 evaluates it (float32, no contraction).
__device__ __forceinline__ float cws_sample_g(const uint8_t* __restrict__ f, int HW, int W, int gx, int gy,
                                              float vx, float vy) {
#pragma clang fp contract(off)
    const float nx = (float)gx + vx, ny = (float)gy + vy;
    const float ux_f = ceilf(nx), dx_f = floorf(nx), uy_f = ceilf(ny), dy_f = floorf(ny);
    const int ux = f2i_sat_g(ux_f), dx = f2i_sat_g(dx_f), uy = f2i_sat_g(uy_f), dy = f2i_sat_g(dy_f);
    const float f11 = fetch_clamped_g(f, (long long)dy * W + dx, HW);
    const float f21 = fetch_clamped_g(f, (long long)dy * W + ux, HW);
    const float f12 = fetch_clamped_g(f, (long long)uy * W + dx, HW);
    const float f22 = fetch_clamped_g(f, (long long)uy * W + ux, HW);
    const float wxu = ux_f - nx, wxd = nx - dx_f, wyu = uy_f - ny, wyd = ny - dy_f;
    float r = (f11 * wxu) * wyu;
    r = r + (f21 * wxd) * wyu;
    r = r + (f12 * wxu) * wyd;
    r = r + (f22 * wxd) * wyd;
    const bool degenerate = ((long long)(ux - dx) * (long long)(uy - dy)) == 0;
    return degenerate ? f11 : r;
}

struct AM {
    float v;
    int idx;
};
__device__ __forceinline__ AM better_g(AM a, AM b) {
    return ((b.v > a.v) || (b.v == a.v && b.idx < a.idx)) ? b : a;
}

// block reductions through LDS (all threads get the result)
__device__ float block_sum(float v, float* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = GT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}
__device__ float block_min(float v, float* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = GT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    return red[0];
}
__device__ AM block_argmax(AM a, float* red, int* redi) {
    __syncthreads();
    red[threadIdx.x] = a.v;
    redi[threadIdx.x] = a.idx;
    __syncthreads();
    for (int s = GT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            AM x{red[threadIdx.x], redi[threadIdx.x]}, y{red[threadIdx.x + s], redi[threadIdx.x + s]};
            x = better_g(x, y);
            red[threadIdx.x] = x.v;
            redi[threadIdx.x] = x.idx;
        }
        __syncthreads();
    }
    return AM{red[0], redi[0]};
}

template <int MODE>
__global__ __launch_bounds__(GT) void xcorr_generic_kernel(PassParams p, cf* scratch) {
    __shared__ cf tw[256];            // exp(-2 pi i k / n)
    __shared__ float red[GT];
    __shared__ int redi[GT];
    const int n = p.ws, nn = n * n;
    const int tid = threadIdx.x;
    const int N = p.n_rows * p.n_cols;
    const long long items = (long long)p.batch * N;
    const int HW = p.H * p.W;
    const int st = p.ws - p.ov;
    cf* T0 = scratch + (size_t)blockIdx.x * 2 * nn;
    cf* T1 = T0 + nn;
    for (int k = tid; k < n; k += GT) {
        double s, c;
        sincospi(2.0 * (double)k / (double)n, &s, &c);
        tw[k] = cf{(float)c, (float)(-s)};
    }
    __syncthreads();

    for (long long item = blockIdx.x; item < items; item += gridDim.x) {
        const int pair = (int)(item / N), win = (int)(item % N);
        const int y0 = (win / p.n_cols) * st, x0 = (win % p.n_cols) * st;
        const uint8_t* __restrict__ fa = p.A + (size_t)pair * HW;
        const uint8_t* __restrict__ fb = p.B + (size_t)pair * HW;
        const size_t fidx = (size_t)item;
        float vx = 0.f, vy = 0.f;
        long long sh = 0;
        if constexpr (MODE == MODE_DWS) sh = (long long)p.v2[fidx] * p.W + (long long)p.u2[fidx];
        if constexpr (MODE == MODE_CWS) {
            vx = (float)p.u2[fidx];
            vy = (float)p.v2[fidx];
        }
        // ---- staging
        float sa = 0.f, sb = 0.f;
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, x = i % n;
            float a, b;
            if constexpr (MODE == MODE_PASS1) {
                const size_t q = (size_t)(y0 + y) * p.W + x0 + x;
                a = (float)fa[q];
                b = (float)fb[q];
            } else if constexpr (MODE == MODE_DWS) {
                const long long q = (long long)(y0 + y) * p.W + x0 + x;
                a = fetch_clamped_g(fa, q - sh, HW);
                b = fetch_clamped_g(fb, q + sh, HW);
            } else {
                a = cws_sample_g(fa, HW, p.W, x0 + x, y0 + y, -vx, -vy);
                b = cws_sample_g(fb, HW, p.W, x0 + x, y0 + y, vx, vy);
            }
            T0[i] = cf{a, b};
            sa += a;
            sb += b;
            if (p.dbg_win != nullptr) {
                p.dbg_win[fidx * 2 * nn + i] = a;
                p.dbg_win[fidx * 2 * nn + nn + i] = b;
            }
        }
        sa = block_sum(sa, red);
        sb = block_sum(sb, red);
        const float ma = sa / (float)nn, mb = sb / (float)nn;
        bool dead = false;
        float ka = 1.f, kb = 1.f;
        if constexpr (MODE == MODE_PASS1) {
            dead = (sa == 0.f) || (sb == 0.f);
            ka = dead ? 0.f : 1.0f / ma;
            kb = dead ? 0.f : 1.0f / mb;
        }
        __syncthreads();
        // ---- forward DFT over x (rows), with the mean removal folded in: T1[y][kx]
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, kx = i % n;
            float re = 0.f, im = 0.f;
            int idx = 0;
            for (int x = 0; x < n; ++x) {
                const cf z = T0[y * n + x];
                const float zr = (z.x - ma) * ka, zi = (z.y - mb) * kb;
                const cf w = tw[idx];
                re += zr * w.x - zi * w.y;
                im += zr * w.y + zi * w.x;
                idx += kx;
                if (idx >= n) idx -= n;
            }
            T1[i] = cf{re, im};
        }
        __syncthreads();
        // ---- forward DFT over y (columns): T0[ky][kx]
        for (int i = tid; i < nn; i += GT) {
            const int ky = i / n, kx = i % n;
            float re = 0.f, im = 0.f;
            int idx = 0;
            for (int y = 0; y < n; ++y) {
                const cf z = T1[y * n + kx];
                const cf w = tw[idx];
                re += z.x * w.x - z.y * w.y;
                im += z.x * w.y + z.y * w.x;
                idx += ky;
                if (idx >= n) idx -= n;
            }
            T0[i] = cf{re, im};
        }
        __syncthreads();
        // ---- cross-spectrum: T1[k] = conj(A) * B / n^2 with A, B split out of Z = FFT2(a + i b)
        const float scale = 0.25f / ((float)nn);
        for (int i = tid; i < nn; i += GT) {
            const int ky = i / n, kx = i % n;
            const cf zk = T0[i];
            const cf zm = T0[((n - ky) % n) * n + (n - kx) % n];
            cf pr;
            pr.x = (zk.x * zm.y + zk.y * zm.x) * (2.0f * scale);
            pr.y = ((zm.x * zm.x - zk.x * zk.x) + (zm.y * zm.y - zk.y * zk.y)) * scale;
            T1[i] = pr;
        }
        __syncthreads();
        // ---- inverse DFT over ky: T0[y][kx]
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, kx = i % n;
            float re = 0.f, im = 0.f;
            int idx = 0;
            for (int ky = 0; ky < n; ++ky) {
                const cf z = T1[ky * n + kx];
                const cf w = tw[idx];                       // conj(w) = (w.x, -w.y)
                re += z.x * w.x + z.y * w.y;
                im += z.y * w.x - z.x * w.y;
                idx += y;
                if (idx >= n) idx -= n;
            }
            T0[i] = cf{re, im};
        }
        __syncthreads();
        // ---- inverse DFT over kx, real part only, stored in fftshift coordinates into the map
        float* map = reinterpret_cast<float*>(T1);
        const int hshift = n / 2;
        float cmin = 3.4e38f;
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, x = i % n;
            float re = 0.f;
            int idx = 0;
            for (int kx = 0; kx < n; ++kx) {
                const cf z = T0[y * n + kx];
                const cf w = tw[idx];
                re += z.x * w.x + z.y * w.y;
                idx += x;
                if (idx >= n) idx -= n;
            }
            const int ys = (y + hshift) % n, xs = (x + hshift) % n;
            map[ys * n + xs] = re;
            cmin = fminf(cmin, re);
        }
        cmin = block_min(cmin, red);
        // ---- corr - min + eps (B:518, B:381), first peak
        AM best{-1.f, 0};
        for (int i = tid; i < nn; i += GT) {
            const float v = __fadd_rn(__fsub_rn(map[i], cmin), 1e-7f);
            map[i] = v;
            if (p.dbg_corr != nullptr) p.dbg_corr[fidx * nn + i] = v;
            if (v > best.v) {
                best.v = v;
                best.idx = i;
            }
        }
        best = block_argmax(best, red, redi);
        const int m = best.idx;
        // ---- second peak outside the flat-index neighbourhood (B:346-358), brute-force membership
        const int wv = p.val_win;
        AM second{-1.f, nn};
        for (int i = tid; i < nn; i += GT) {
            bool excl = false;
            for (int j = -wv; j <= wv; ++j) {
                const int t = i - m - n * j;
                if (t >= -wv && t <= wv) excl = true;
            }
            if (i == 0 && (m - wv - wv * n) <= 0) excl = true;              // clamp to 0
            if (i == nn - 1 && (m + wv + wv * n) >= nn - 1) excl = true;    // clamp to n*n-1
            const float v = map[i];
            if (!excl && v > second.v) {
                second.v = v;
                second.idx = i;
            }
        }
        second = block_argmax(second, red, redi);
        __syncthreads();
        if (tid < 8) {
            int left = m + 1, right = m - 1, top = m + n, bot = m - n;      // B:385-392
            if (left >= nn - 1) left = m;
            if (right <= 0) right = m;
            if (top >= nn - 1) top = m;
            if (bot <= 0) bot = m;
            int q = m;
            q = (tid == 1) ? left : q;
            q = (tid == 2) ? right : q;
            q = (tid == 3) ? top : q;
            q = (tid == 4) ? bot : q;
            q = (tid == 5) ? (second.idx < nn ? second.idx : 0) : q;
            float outv = map[q];
            // Every cell inside the exclusion zone (maps smaller than 7x7): the reference's second
            // arg-max then runs over an all-zero map and returns index 0 (B:357).  In pass 1 the
            // float64 `cor` aliases the zeroed map (B:382), so c[m2] = 0 and the ratio is +inf;
            // in passes >= 2 `cor` is a float64 copy made before the zeroing, so c[m2] = c[0].
            if (tid == 5 && second.idx >= nn && MODE == MODE_PASS1) outv = 0.0f;
            outv = (tid == 6) ? __int_as_float(m) : outv;
            outv = (tid == 7) ? __int_as_float(dead ? 1 : 0) : outv;
            p.peak_raw[fidx * 8 + tid] = outv;
        }
        __syncthreads();
    }
}

}  // namespace

int generic_blocks(int ws, long long items, int n_cu) {
    // scratch = 2 * ws^2 complex per workgroup; keep it below 256 MiB
    const long long per = 16LL * ws * ws;
    long long b = (256LL << 20) / per;
    if (b > (long long)n_cu * 4) b = (long long)n_cu * 4;
    if (b > items) b = items;
    if (b < 1) b = 1;
    return (int)b;
}

hipError_t launch_xcorr_generic(const PassParams& p, int mode, int n_cu, cf* scratch, hipStream_t stream) {
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    const int blocks = generic_blocks(p.ws, items, n_cu);
    switch (mode) {
        case MODE_PASS1:
            hipLaunchKernelGGL((xcorr_generic_kernel<MODE_PASS1>), dim3(blocks), dim3(GT), 0, stream, p, scratch);
            break;
        case MODE_DWS:
            hipLaunchKernelGGL((xcorr_generic_kernel<MODE_DWS>), dim3(blocks), dim3(GT), 0, stream, p, scratch);
            break;
        case MODE_CWS:
            hipLaunchKernelGGL((xcorr_generic_kernel<MODE_CWS>), dim3(blocks), dim3(GT), 0, stream, p, scratch);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace tpiv
