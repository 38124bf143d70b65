// Generic-size PIV tile kernel: any window size 2..256 (the reference accepts any integer,
// ControlsWidgets.py:88-100, and multipass scales such as 1.5 give sizes like 42 or 28).
// Not on the hot path (the power-of-two sizes run xcorr_tile.hpp): one 256-thread workgroup per
// window, the complex tile lives in a global (L2-resident) scratch, the transforms are plain DFTs
// with a twiddle table in LDS.  Staging, peak search and hand-off to finalize_kernel follow the
// same reference semantics as xcorr_tile.hpp (PIVbackend.py:147-216, 249-257, 346-422, 459-520).
// Templated on the arithmetic type: float (all modes) and double (first pass at reference precision,
// TPIV_PREC_REFERENCE: the reference promotes pass 1 to float64, B:513-514 -- windows divided by their
// mean, no mean removal, float64 transforms, map and peak analysis).
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

// ---- piv_iteration_CWS_Fast (B:644-653): torch's affine_grid + grid_sample(mode="bicubic",
// padding_mode="border", align_corners=False) of a window INSIDE ITSELF, in float32.
// cubic convolution with A = -0.75 (ATen UpSample.h: cubic_convolution1/2, get_cubic_upsample_coefficients)
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    constexpr float A = -0.75f;
    auto cc1 = [](float x) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; };
    auto cc2 = [](float x) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; };
    c[0] = cc2(t + 1.0f);
    c[1] = cc1(t);
    c[2] = cc1(1.0f - t);
    c[3] = cc2(2.0f - t);
}
// base grid coordinate of pixel j of an n-pixel axis: linspace(-1, 1, n)[j] * (n - 1) / n (affine_grid,
// align_corners=False), plus the translation
__device__ __forceinline__ float base_coord(int j, int n) {
    const float step = 2.0f / (float)(n - 1);
    const float l = j < n / 2 ? -1.0f + step * (float)j : 1.0f - step * (float)(n - 1 - j);
    return l * (float)(n - 1) / (float)n;
}
__device__ __forceinline__ float bicubic_local(const uint8_t* __restrict__ f, int W, int y0, int x0, int n, int x, int y,
                                               float tx, float ty) {
    // unnormalise (align_corners=False): ((coord + 1) * size - 1) / 2; the taps are clamped one by one (border)
    const float ix = ((base_coord(x, n) + tx + 1.0f) * (float)n - 1.0f) * 0.5f;
    const float iy = ((base_coord(y, n) + ty + 1.0f) * (float)n - 1.0f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    float cx[4], cy[4];
    cubic_coeffs(ix - fx, cx);
    cubic_coeffs(iy - fy, cy);
    const int bx = f2i_sat_g(fx) - 1, by = f2i_sat_g(fy) - 1;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int yy = by + i;
        yy = yy < 0 ? 0 : (yy > n - 1 ? n - 1 : yy);
        float row = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int xx = bx + j;
            xx = xx < 0 ? 0 : (xx > n - 1 ? n - 1 : xx);
            row += (float)f[(size_t)(y0 + yy) * W + x0 + xx] * cx[j];
        }
        acc += row * cy[i];
    }
    return acc;
}

template <typename R>
struct cplx {
    R x, y;
};
template <typename R>
struct AM {
    R v;
    int idx;
};
template <typename R>
__device__ __forceinline__ AM<R> better_g(AM<R> a, AM<R> b) {
    return ((b.v > a.v) || (b.v == a.v && b.idx < a.idx)) ? b : a;
}
template <typename R>
__device__ __forceinline__ R rmin(R a, R b) { return a < b ? a : (b < a ? b : (a == a ? a : b)); }     // fmin semantics
__device__ __forceinline__ float add_eps(float c, float cmin) { return __fadd_rn(__fsub_rn(c, cmin), 1e-7f); }
__device__ __forceinline__ double add_eps(double c, double cmin) { return __dadd_rn(__dsub_rn(c, cmin), 1e-7); }
// record slots 6 (peak index) and 7 (dead flag): bit patterns in float records, plain values in double records
__device__ __forceinline__ float rec_int(float, int v) { return __int_as_float(v); }
__device__ __forceinline__ double rec_int(double, int v) { return (double)v; }

// block reductions through LDS (all threads get the result)
template <typename R>
__device__ R block_sum(R v, R* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = GT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}
template <typename R>
__device__ R block_min(R v, R* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = GT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = rmin(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    return red[0];
}
template <typename R>
__device__ AM<R> block_argmax(AM<R> a, R* red, int* redi) {
    __syncthreads();
    red[threadIdx.x] = a.v;
    redi[threadIdx.x] = a.idx;
    __syncthreads();
    for (int s = GT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            AM<R> x{red[threadIdx.x], redi[threadIdx.x]}, y{red[threadIdx.x + s], redi[threadIdx.x + s]};
            x = better_g(x, y);
            red[threadIdx.x] = x.v;
            redi[threadIdx.x] = x.idx;
        }
        __syncthreads();
    }
    return AM<R>{red[0], redi[0]};
}

template <int MODE, typename R>
__global__ __launch_bounds__(GT) void xcorr_generic_kernel(PassParams p, cplx<R>* scratch) {
    using cf = cplx<R>;
    constexpr bool F64 = sizeof(R) == 8;
    __shared__ cf tw[256];            // exp(-2 pi i k / n)
    __shared__ cf tw2[256];           // odd n: exp(-2 pi i k / (n - 1)) for the last inverse transform
    __shared__ R red[GT];
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
        tw[k] = cf{(R)c, (R)(-s)};
        if (n > 1) {
            sincospi(2.0 * (double)k / (double)(n - 1), &s, &c);
            tw2[k] = cf{(R)c, (R)(-s)};
        }
    }
    __syncthreads();

    for (long long item = blockIdx.x; item < items; item += gridDim.x) {
        const int pair = (int)(item / N), win = (int)(item % N);
        const int y0 = (win / p.n_cols) * st, x0 = (win % p.n_cols) * st;
        const uint8_t* __restrict__ fa = p.A + (size_t)pair * HW;
        const uint8_t* __restrict__ fb = p.B + (size_t)pair * HW;
        const size_t fidx = (size_t)item;
        float vx = 0.f, vy = 0.f;     // (shifted passes exist in float only)
        long long sh = 0;
        if constexpr (MODE == MODE_DWS) {
            double sx, sy;
            pred_half_shift<MODE_DWS>(p, fidx, sx, sy);
            sh = (long long)sy * p.W + (long long)sx;
        }
        if constexpr (MODE == MODE_CWS) pred_half_shift_cws_f32(p, fidx, vx, vy);
        if constexpr (MODE == MODE_CWSF) {      // B:644-645: theta[:, 0, 2] = -u0 / wind_size (float64, stored as float32)
            vx = (float)(p.u0[fidx] / (double)n);
            vy = (float)(p.v0[fidx] / (double)n);
        }
        // ---- staging
        R sa = 0, sb = 0;
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, x = i % n;
            R a, b;
            if constexpr (MODE == MODE_PASS1) {
                const size_t q = (size_t)(y0 + y) * p.W + x0 + x;
                a = (R)fa[q];
                b = (R)fb[q];
            } else if constexpr (MODE == MODE_DWS) {
                const long long q = (long long)(y0 + y) * p.W + x0 + x;
                a = fetch_clamped_g(fa, q - sh, HW);
                b = fetch_clamped_g(fb, q + sh, HW);
            } else if constexpr (MODE == MODE_CWS) {
                a = cws_sample_g(fa, HW, p.W, x0 + x, y0 + y, -vx, -vy);
                b = cws_sample_g(fb, HW, p.W, x0 + x, y0 + y, vx, vy);
            } else {
                a = bicubic_local(fa, p.W, y0, x0, n, x, y, -vx, -vy);
                b = bicubic_local(fb, p.W, y0, x0, n, x, y, vx, vy);
            }
            T0[i] = cf{a, b};
            sa += a;
            sb += b;
            if (p.dbg_win != nullptr) {
                p.dbg_win[fidx * 2 * nn + i] = (float)a;
                p.dbg_win[fidx * 2 * nn + nn + i] = (float)b;
            }
        }
        sa = block_sum(sa, red);
        sb = block_sum(sb, red);
        R ma = sa / (R)nn, mb = sb / (R)nn;        // (sums of integers: exact in either type)
        bool dead = false;
        R ka = 1, kb = 1;
        if constexpr (MODE == MODE_PASS1 || MODE == MODE_CWSF) {      // a / mean(a): B:513-514, B:656-657
            dead = (sa == 0) || (sb == 0);
            ka = dead ? (R)0 : (R)1 / ma;
            kb = dead ? (R)0 : (R)1 / mb;
        }
        __syncthreads();
        if constexpr (F64) {
            // reference arithmetic (B:513-514): a / mean(a) by division, the DC pedestal stays in
            for (int i = tid; i < nn; i += GT) {
                const cf z = T0[i];
                T0[i] = dead ? cf{0, 0} : cf{z.x / ma, z.y / mb};
            }
            ma = 0;
            mb = 0;
            ka = 1;
            kb = 1;
            __syncthreads();
        }
        // ---- forward DFT over x (rows), with the mean removal folded in: T1[y][kx]
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, kx = i % n;
            R re = 0, im = 0;
            int idx = 0;
            for (int x = 0; x < n; ++x) {
                const cf z = T0[y * n + x];
                const R zr = (z.x - ma) * ka, zi = (z.y - mb) * kb;
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
            R re = 0, im = 0;
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
        const R scale = (R)0.25 / ((R)nn);
        for (int i = tid; i < nn; i += GT) {
            const int ky = i / n, kx = i % n;
            const cf zk = T0[i];
            const cf zm = T0[((n - ky) % n) * n + (n - kx) % n];
            cf pr;
            pr.x = (zk.x * zm.y + zk.y * zm.x) * ((R)2 * scale);
            pr.y = ((zm.x * zm.x - zk.x * zk.x) + (zm.y * zm.y - zk.y * zk.y)) * scale;
            T1[i] = pr;
        }
        __syncthreads();
        // ---- inverse DFT over ky: T0[y][kx]
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, kx = i % n;
            R re = 0, im = 0;
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
        // ---- inverse DFT over kx, real part only, stored in fftshift coordinates into the map.
        // ODD window sizes: the reference calls irfft2 WITHOUT `s` (B:255), so the (n+1)/2 spectrum columns
        // of its rfft2 are read as the half spectrum of an EVEN length mc = n - 1: the map is n x (n-1), its
        // last column bin is treated as a Nyquist bin, the normalisation is 1/(n (n-1)), and every
        // flat-index rule below runs on that n x (n-1) map (the reference's formulas use k = n-1 columns
        // and d = n rows, B:404-407, 415-417) -- reproduced as it is.
        R* map = reinterpret_cast<R*>(T1);
        const bool odd = (n & 1) != 0;
        const int mc = odd ? n - 1 : n;             // map columns (k in the reference)
        const int nm = n * mc;                      // map cells (k * d)
        R cmin = (R)3.4e38;
        if (!odd) {
            const int hshift = n / 2;
            for (int i = tid; i < nn; i += GT) {
                const int y = i / n, x = i % n;
                R re = 0;
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
                cmin = rmin(cmin, re);
            }
        } else {
            const int hb = mc / 2;                  // index of the bin that plays the Nyquist role
            const R renorm = (R)n / (R)mc;          // 1/(n mc) instead of the 1/n^2 folded into the spectrum
            for (int i = tid; i < nm; i += GT) {
                const int y = i / mc, x = i % mc;
                R re = T0[y * n].x + ((x & 1) ? -T0[y * n + hb].x : T0[y * n + hb].x);
                int idx = 0;
                for (int kx = 1; kx < hb; ++kx) {
                    idx += x;
                    if (idx >= mc) idx -= mc;
                    const cf z = T0[y * n + kx];
                    const cf w = tw2[idx];
                    re += (R)2 * (z.x * w.x + z.y * w.y);
                }
                re *= renorm;
                const int ys = (y + n / 2) % n, xs = (x + mc / 2) % mc;
                map[ys * mc + xs] = re;
                cmin = rmin(cmin, re);
            }
        }
        cmin = block_min(cmin, red);
        // ---- corr - min + eps (B:518, B:381), first peak
        AM<R> best{(R)-1, 0};
        for (int i = tid; i < nm; i += GT) {
            const R v = add_eps(map[i], cmin);
            map[i] = v;
            if (p.dbg_corr != nullptr && !odd) p.dbg_corr[fidx * nn + i] = (float)v;
            if (v > best.v) {
                best.v = v;
                best.idx = i;
            }
        }
        best = block_argmax(best, red, redi);
        const int m = best.idx;
        // ---- second peak outside the flat-index neighbourhood (B:346-358), brute-force membership
        const int wv = p.val_win;
        AM<R> second{(R)-1, nm};
        for (int i = tid; i < nm; i += GT) {
            bool excl = false;
            for (int j = -wv; j <= wv; ++j) {
                const int t = i - m - mc * j;
                if (t >= -wv && t <= wv) excl = true;
            }
            if (i == 0 && (m - wv - wv * mc) <= 0) excl = true;              // clamp to 0
            if (i == nm - 1 && (m + wv + wv * mc) >= nm - 1) excl = true;    // clamp to k*d-1
            const R v = map[i];
            if (!excl && v > second.v) {
                second.v = v;
                second.idx = i;
            }
        }
        second = block_argmax(second, red, redi);
        __syncthreads();
        if (tid < 8) {
            int left = m + 1, right = m - 1, top = m + mc, bot = m - mc;    // B:385-392
            if (left >= nm - 1) left = m;
            if (right <= 0) right = m;
            if (top >= nm - 1) top = m;
            if (bot <= 0) bot = m;
            int q = m;
            q = (tid == 1) ? left : q;
            q = (tid == 2) ? right : q;
            q = (tid == 3) ? top : q;
            q = (tid == 4) ? bot : q;
            q = (tid == 5) ? (second.idx < nm ? second.idx : 0) : q;
            R outv = map[q];
            // Every cell inside the exclusion zone (maps smaller than 7x7): the reference's second
            // arg-max then runs over an all-zero map and returns index 0 (B:357).  In pass 1 the
            // float64 `cor` aliases the zeroed map (B:382), so c[m2] = 0 and the ratio is +inf;
            // in passes >= 2 `cor` is a float64 copy made before the zeroing, so c[m2] = c[0].
            if (tid == 5 && second.idx >= nm && MODE == MODE_PASS1) outv = 0;
            outv = (tid == 6) ? rec_int(R(), m) : outv;
            outv = (tid == 7) ? rec_int(R(), dead ? 1 : 0) : outv;
            reinterpret_cast<R*>(p.peak_raw)[fidx * 8 + tid] = outv;
        }
        __syncthreads();
    }
}

}  // namespace

int generic_blocks(int ws, long long items, int n_cu, int elem_bytes) {
    // scratch = 2 * ws^2 complex per workgroup; keep it below 256 MiB
    const long long per = 4LL * elem_bytes * ws * ws;
    long long b = (256LL << 20) / per;
    if (b > (long long)n_cu * 4) b = (long long)n_cu * 4;
    if (b > items) b = items;
    if (b < 1) b = 1;
    return (int)b;
}

hipError_t launch_xcorr_generic(const PassParams& p, int mode, int n_cu, void* scratch, hipStream_t stream) {
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    if (p.precision != 0) {      // float64: first pass only
        if (mode != MODE_PASS1) return hipErrorInvalidValue;
        const int blocks = generic_blocks(p.ws, items, n_cu, 8);
        hipLaunchKernelGGL((xcorr_generic_kernel<MODE_PASS1, double>), dim3(blocks), dim3(GT), 0, stream, p,
                           static_cast<cplx<double>*>(scratch));
        return hipGetLastError();
    }
    const int blocks = generic_blocks(p.ws, items, n_cu, 4);
    cplx<float>* sc = static_cast<cplx<float>*>(scratch);
    switch (mode) {
        case MODE_PASS1:
            hipLaunchKernelGGL((xcorr_generic_kernel<MODE_PASS1, float>), dim3(blocks), dim3(GT), 0, stream, p, sc);
            break;
        case MODE_DWS:
            hipLaunchKernelGGL((xcorr_generic_kernel<MODE_DWS, float>), dim3(blocks), dim3(GT), 0, stream, p, sc);
            break;
        case MODE_CWS:
            hipLaunchKernelGGL((xcorr_generic_kernel<MODE_CWS, float>), dim3(blocks), dim3(GT), 0, stream, p, sc);
            break;
        case MODE_CWSF:
            hipLaunchKernelGGL((xcorr_generic_kernel<MODE_CWSF, float>), dim3(blocks), dim3(GT), 0, stream, p, sc);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace tpiv
