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
#include <stdlib.h>

#include "fft_inreg.hpp"
#include "fft_mixed.hpp"
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

// The same sample with the four corner pixels taken from a patch of the frame that the wavefront loaded into LDS
// (patch[py][px] = pixel with the flat index clamp((by + py) * W + (bx + px)), i.e. exactly what fetch_clamped_g returns for
// that corner); a corner outside the patch -- a wild predictor -- falls back to the global fetch.
__device__ __forceinline__ float cws_sample_patch(const uint8_t* __restrict__ patch, int PD, int pitch, int bx, int by,
                                                  const uint8_t* __restrict__ f, int HW, int W, int gx, int gy, float vx, float vy) {
#pragma clang fp contract(off)
    const float nx = (float)gx + vx, ny = (float)gy + vy;
    const float ux_f = ceilf(nx), dx_f = floorf(nx), uy_f = ceilf(ny), dy_f = floorf(ny);
    const int ux = f2i_sat_g(ux_f), dx = f2i_sat_g(dx_f), uy = f2i_sat_g(uy_f), dy = f2i_sat_g(dy_f);
    float f11, f21, f12, f22;
    // the four corners span at most 2 x 2 pixels: both extreme corners inside the patch = all four inside (the usual case
    // for every lane: one address and three small offsets instead of four bounds checks with a global-memory fallback each)
    const int py0 = dy - by, px0 = dx - bx, py1 = uy - by, px1 = ux - bx;
    const bool inside = ((unsigned)py0 < (unsigned)PD) & ((unsigned)px0 < (unsigned)PD) & ((unsigned)py1 < (unsigned)PD) &
                        ((unsigned)px1 < (unsigned)PD);
    if (__builtin_expect(__all(inside), 1)) {
        const uint8_t* q0 = patch + py0 * pitch + px0;
        const int ox = px1 - px0, oy = (py1 - py0) * pitch;
        f11 = (float)q0[0];
        f21 = (float)q0[ox];
        f12 = (float)q0[oy];
        f22 = (float)q0[oy + ox];
    } else {
        auto px = [&](int yy, int xx) {
            const int py = yy - by, pxx = xx - bx;
            return ((unsigned)py < (unsigned)PD && (unsigned)pxx < (unsigned)PD) ? (float)patch[py * pitch + pxx]
                                                                                   : fetch_clamped_g(f, (long long)yy * W + xx, HW);
        };
        f11 = px(dy, dx), f21 = px(dy, ux), f12 = px(uy, dx), f22 = px(uy, ux);
    }
    const float wxu = ux_f - nx, wxd = nx - dx_f, wyu = uy_f - ny, wyd = ny - dy_f;
    float r = (f11 * wxu) * wyu;
    r = r + (f21 * wxd) * wyu;
    r = r + (f12 * wxu) * wyd;
    r = r + (f22 * wxd) * wyd;
    const bool degenerate = (ux == dx) | (uy == dy);          // (ux - dx) * (uy - dy) == 0  (B:192-193)
    return degenerate ? f11 : r;
}

// ---- the same sample through SEPARABLE TABLES (round 5): the column part of B:162-172 (floor / ceil of the x coordinate,
// its two weights, the corner's offset inside the patch) depends on the column only, the row part on the row only -- n + n
// table entries per frame instead of n x n evaluations of both.  The sample itself is B:187-193 operation by operation with
// the same float32 numbers: bit-identical to cws_sample_patch (every staged-window test runs through it).
struct CwsAxis {
    int off;        // corner (floor) coordinate inside the patch: column index, or row index times the patch pitch
    int step;       // distance to the ceil corner: 0 (integral coordinate: B:170, B:193 -- the "nearest sample" quirk), 1 column or 1 row
    float w_up;     // ceil - coordinate: weight of the FLOOR corner
    float w_dn;     // coordinate - floor
};
// entry of grid coordinate g (pixel index along the axis) shifted by v; base = first pixel of the patch along the axis, PD = its
// extent, scale = 1 (columns) or the patch pitch (rows).  *inside: both corners lie inside the patch.
__device__ __forceinline__ CwsAxis cws_axis(int g, float v, int base, int PD, int scale, bool& inside) {
#pragma clang fp contract(off)
    const float n_ = (float)g + v;
    const float u_f = ceilf(n_), d_f = floorf(n_);
    const int u = f2i_sat_g(u_f), d = f2i_sat_g(d_f);
    inside = ((unsigned)(d - base) < (unsigned)PD) & ((unsigned)(u - base) < (unsigned)PD);
    CwsAxis e;
    e.off = (d - base) * scale;
    e.step = (u - d) * scale;
    e.w_up = u_f - n_;
    e.w_dn = n_ - d_f;
    return e;
}
__device__ __forceinline__ float cws_sample_tab(const uint8_t* __restrict__ patch, CwsAxis cx, CwsAxis cy) {
#pragma clang fp contract(off)
    const uint8_t* q0 = patch + cy.off + cx.off;
    const float f11 = (float)q0[0], f21 = (float)q0[cx.step], f12 = (float)q0[cy.step], f22 = (float)q0[cy.step + cx.step];
    float r = (f11 * cx.w_up) * cy.w_up;
    r = r + (f21 * cx.w_dn) * cy.w_up;
    r = r + (f12 * cx.w_up) * cy.w_dn;
    r = r + (f22 * cx.w_dn) * cy.w_dn;
    return (cx.step == 0) | (cy.step == 0) ? f11 : r;          // (ux - dx) * (uy - dy) == 0  (B:192-193)
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

// ---- precision "exact" (xcorr_exact.hip): the float32 map of a generic-size window only LOCATES the cells the refinement
// evaluates as exact integer sums.  PassParams::cand != nullptr (first pass only) switches the kernels of this file from
// the 8-float record to the candidate record of xcorr_tile.hpp's peak_candidates: the arg-max if it is the only cell within
// the band of the maximum, up to EXACT_MAX_SECOND cells outside the exclusion zone within the band of their maximum, up to
// EXACT_MAX_MIN cells within the band of the minimum (more: three and -2); undecided -1, dead -2.
// E+ = (|a'|^2 + |b'|^2) / 2, a' = a / mean(a) - 1, from the exact integer sums of the staged bytes (piv_kernels.h, "The band")
__device__ __forceinline__ float exact_e_plus(unsigned sa, unsigned sb, unsigned saa, unsigned sbb, int nn) {
    const double n_ = (double)nn, da = (double)sa, db = (double)sb;
    const float ea = (float)__fma_rn(-da, da, n_ * (double)saa), eb = (float)__fma_rn(-db, db, n_ * (double)sbb);
    const float ka = (float)nn / (float)sa, kb = (float)nn / (float)sb;
    return (0.5f / (float)nn) * (ea * (ka * ka) + eb * (kb * kb));
}
// map: n x n cells v = (c - min) + 1e-7 in fftshift layout (LDS or global), complete and visible to the workgroup; m / gmax:
// first peak; second_v: largest cell outside the exclusion zone (has_second: there is one); scr: 10 ints of LDS.
// Every thread of the workgroup calls it (two barriers).
__device__ __forceinline__ void map_candidates(const PassParams& p, const float* map, int n, int m, float gmax, float second_v,
                                               bool has_second, float band_abs, bool dead, bool store, size_t fidx, int* scr) {
    const int nn = n * n, wv = p.val_win, tid = (int)threadIdx.x, nthreads = (int)blockDim.x;
    if (tid < 3) scr[tid] = 0;
    __syncthreads();
    const float band = fmaxf(p.exact_band_range * (gmax - 1e-7f), band_abs);
    const float top_thr = gmax - band, sec_thr = second_v - band, min_thr = 1e-7f + band;
    for (int i = tid; i < nn; i += nthreads) {
        const float v = map[i];
        if (v >= top_thr) atomicAdd(&scr[0], 1);
        if (has_second && v >= sec_thr) {
            bool excl = false;                               // B:346-358: i == clamp(m + t + n j), |t|, |j| <= wv
            for (int j = -wv; j <= wv; ++j) {
                const int t = i - m - n * j;
                excl = excl || (t >= -wv && t <= wv);
            }
            excl = excl || (i == 0 && (m - wv - wv * n) <= 0) || (i == nn - 1 && (m + wv + wv * n) >= nn - 1);
            if (!excl) {
                const int k = atomicAdd(&scr[1], 1);
                if (k < EXACT_MAX_SECOND) scr[3 + k] = i;
            }
        }
        if (v <= min_thr) {
            const int k = atomicAdd(&scr[2], 1);
            if (k < EXACT_MAX_MIN) scr[3 + EXACT_MAX_SECOND + k] = i;
        }
    }
    __syncthreads();
    if (tid == 0 && store) {
        const int cnt = scr[0], ns = scr[1], nm_ = scr[2];
        const bool open = !(band > 0.0f) || !(gmax > 1e-7f) || cnt != 1 || ns > EXACT_MAX_SECOND;
        auto get = [&](int base, int cnt_, int k) { return k < cnt_ ? scr[base + k] : -1; };
        auto pack = [](int lo_, int hi_) { return ((unsigned)lo_ & 0xffffu) | ((unsigned)hi_ << 16); };
        const int m_out = dead ? -2 : (open ? -1 : m);
        const int n3 = nm_ > EXACT_MAX_MIN ? -2 : get(3 + EXACT_MAX_SECOND, nm_, 3);
        uint4 rec;
        rec.x = pack(m_out, get(3, ns, 0));
        rec.y = pack(get(3, ns, 1), get(3, ns, 2));
        rec.z = pack(get(3 + EXACT_MAX_SECOND, nm_, 0), get(3 + EXACT_MAX_SECOND, nm_, 1));
        rec.w = pack(get(3 + EXACT_MAX_SECOND, nm_, 2), n3);
        p.cand[fidx] = rec;
    }
    __syncthreads();
}

template <int MODE, typename R>
__global__ __launch_bounds__(GT) void xcorr_generic_kernel(PassParams p, cplx<R>* scratch) {
    using cf = cplx<R>;
    constexpr bool F64 = sizeof(R) == 8;
    __shared__ cf tw[256];            // exp(-2 pi i k / n)
    __shared__ cf tw2[256];           // odd n: exp(-2 pi i k / (n - 1)) for the last inverse transform
    __shared__ R red[GT];
    __shared__ int redi[GT];
    __shared__ int cscr[16];          // candidate lists of the exact scheme (map_candidates)
    // LIST (float64 pass 1 under precision "exact"): the windows of PassParams::fb_list, counted on the device
    const bool listed = F64 && MODE == MODE_PASS1 && p.fb_list != nullptr;
    constexpr bool CAN_CAND = !F64 && MODE == MODE_PASS1;
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

    const long long n_items = listed ? (long long)*p.fb_count : items;
    for (long long item_ = blockIdx.x; item_ < n_items; item_ += gridDim.x) {
        const long long item = listed ? (long long)p.fb_list[item_] : item_;
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
        unsigned iaa = 0u, ibb = 0u;      // CAND: sums of squares of the bytes (exact)
        for (int i = tid; i < nn; i += GT) {
            const int y = i / n, x = i % n;
            R a, b;
            if constexpr (MODE == MODE_PASS1) {
                const size_t q = (size_t)(y0 + y) * p.W + x0 + x;
                const unsigned ba = fa[q], bb = fb[q];
                a = (R)ba;
                b = (R)bb;
                if constexpr (CAN_CAND) {
                    iaa += ba * ba;
                    ibb += bb * bb;
                }
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
        float band_abs = 0.f;
        if constexpr (CAN_CAND) {
            if (p.cand != nullptr) {      // (window sums of squares < 2^31 for n <= 181; sa, sb < 2^24: exact in float)
                const unsigned saa = block_sum(iaa, reinterpret_cast<unsigned*>(red));
                const unsigned sbb = block_sum(ibb, reinterpret_cast<unsigned*>(red));
                band_abs = dead ? 0.f : p.exact_band * exact_e_plus((unsigned)sa, (unsigned)sb, saa, sbb, nn);
            }
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
        if constexpr (CAN_CAND) {
            if (p.cand != nullptr) {      // (even sizes only: launch_xcorr)
                map_candidates(p, reinterpret_cast<const float*>(map), n, m, (float)best.v, (float)second.v, second.idx < nm,
                               band_abs, dead, true, fidx, cscr);
                continue;
            }
        }
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


// =====================================================================================================
// Second generation for the sizes the reference's own schedule produces with multipass_scale != 2 (42, 28, 48, 24, 96,
// ...: PIVbackend.py:855-858, int(ws // scale)): the same staging / peak semantics, but
//   * both complex tiles live in LDS (2 n (n | 1) elements; n <= 96), not in an L2-resident scratch,
//   * every length-n transform is a two-factor Cooley-Tukey n = n1 n2 (2 <= n1 <= n2 <= 12: a radix-16 butterfly needs 248 registers and would set the kernel's occupancy): per line n2 radix-n1
//     butterflies, a twiddle w_n^(b k1), then n1 radix-n2 butterflies -- n (n1 + n2) complex multiply-adds per line
//     instead of n^2 (42 = 6 x 7: 13 instead of 42 per output), each butterfly a fully unrolled direct DFT on REGISTERS
//     (inputs, outputs and the radix's roots: one LDS read and one LDS write per element and step),
//   * no integer division per element (float reciprocal, exact for the index range).
// Float32 only (every mode); even sizes whose factors fit; everything else stays with xcorr_generic_kernel above.
// =====================================================================================================
using cff = cplx<float>;
constexpr int CT_T = 256;    // most threads per workgroup = per window.  Round 4 first ran ONE wavefront per window (free barriers,
                             // reductions as cross-lane moves): with ~29 KB of LDS per window that is five wavefronts per CU.
                             // Measured since (64 -> 42 -> 28 chain): four wavefronts per window take the 42 x 42 pass from
                             // 496 to 412 us per pair and the 28 x 28 pass from 431 to 596 (784 elements: three sweeps of
                             // 256 threads, barriers dominate) -- so windows of 40 pixels and more get four wavefronts, smaller
                             // ones one (ct_threads); the loops below take their stride from blockDim.x
constexpr int CT_WAVES = CT_T / 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = rmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ AM<float> wave_argmax(AM<float> a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = better_g(a, AM<float>{__shfl_xor(a.v, o, 64), __shfl_xor(a.idx, o, 64)});
    return a;
}

// workgroup-wide forms (red: CT_WAVES slots of the dynamic LDS; the trailing barrier frees them for the next call)
__device__ __forceinline__ float ct_sum(float v, float* red) {
    v = wave_sum(v);
    const int nw = (int)blockDim.x >> 6;
    if (nw == 1) return v;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < nw; ++w) r += red[w];
    __syncthreads();
    return r;
}
__device__ __forceinline__ unsigned ct_sum_u(unsigned v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    const int nw = (int)blockDim.x >> 6;
    if (nw == 1) return v;
    unsigned* redu = reinterpret_cast<unsigned*>(red);
    if ((threadIdx.x & 63) == 0) redu[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned r = redu[0];
    for (int w = 1; w < nw; ++w) r += redu[w];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float ct_min(float v, float* red) {
    v = wave_min(v);
    const int nw = (int)blockDim.x >> 6;
    if (nw == 1) return v;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < nw; ++w) r = rmin(r, red[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ AM<float> ct_argmax(AM<float> a, float* red) {
    a = wave_argmax(a);
    const int nw = (int)blockDim.x >> 6;
    if (nw == 1) return a;
    int* redi = reinterpret_cast<int*>(red + CT_WAVES);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = a.v;
        redi[threadIdx.x >> 6] = a.idx;
    }
    __syncthreads();
    AM<float> r{red[0], redi[0]};
    for (int w = 1; w < nw; ++w) r = better_g(r, AM<float>{red[w], redi[w]});
    __syncthreads();
    return r;
}

__device__ __forceinline__ int div_small(int i, float rcp) { return (int)(((float)i + 0.5f) * rcp); }     // i / n for i < 2^22

// One radix-R step over `count` butterflies.  Butterfly w (t = butterfly inside its line, line): inputs
// src[line * lsi + t * bsi + j * esi], j < R; outputs dst[line * lso + t * bso + k * eso] = sum_j in[j] w_R^(+-j k), times
// tw[(t k) % n] when TWIDDLE.  line_fast: consecutive work items walk the lines (unit stride of a column pass) instead of
// the butterflies of one line.
template <int R, bool TWIDDLE>
__device__ __noinline__ void radix_pass(const cff* __restrict__ src, cff* __restrict__ dst, int n, int nlines, int per_line,
                                           int lsi, int esi, int bsi, int lso, int eso, int bso, bool line_fast, bool inverse,
                                           const cff* __restrict__ tw) {
    cff w[R];
    const int step = n / R;                      // w_R^j = w_n^(j n / R)
#pragma unroll
    for (int j = 0; j < R; ++j) {
        w[j] = tw[j * step];
        if (inverse) w[j].y = -w[j].y;
    }
    const int count = nlines * per_line;
    const float rcp = 1.0f / (float)(line_fast ? nlines : per_line);
    for (int item = threadIdx.x; item < count; item += (int)blockDim.x) {
        const int hi = div_small(item, rcp);
        const int lo = item - hi * (line_fast ? nlines : per_line);
        const int line = line_fast ? lo : hi, t = line_fast ? hi : lo;
        cff in[R], out[R];
        const cff* s0 = src + line * lsi + t * bsi;
#pragma unroll
        for (int j = 0; j < R; ++j) in[j] = s0[j * esi];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            float re = in[0].x, im = in[0].y;
#pragma unroll
            for (int j = 1; j < R; ++j) {
                const cff ww = w[(j * k) % R];
                re += in[j].x * ww.x - in[j].y * ww.y;
                im += in[j].x * ww.y + in[j].y * ww.x;
            }
            out[k] = cff{re, im};
        }
        cff* d0 = dst + line * lso + t * bso;
        if constexpr (TWIDDLE) {
            int idx = 0;                          // (t k) % n, k ascending
#pragma unroll
            for (int k = 0; k < R; ++k) {
                cff ww = tw[idx];
                if (inverse) ww.y = -ww.y;
                d0[k * eso] = cff{out[k].x * ww.x - out[k].y * ww.y, out[k].x * ww.y + out[k].y * ww.x};
                idx += t;
                if (idx >= n) idx -= n;
            }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) d0[k * eso] = out[k];
        }
    }
}

// (the butterflies are separate functions, one per radix: a single out-of-line dispatcher would save and restore the
//  registers of its LARGEST case -- radix 16, 248 VGPRs -- around every call, whatever radix runs)
template <bool TWIDDLE>
__device__ __forceinline__ void radix_dispatch(int R, const cff* src, cff* dst, int n, int nlines, int per_line, int lsi, int esi,
                                            int bsi, int lso, int eso, int bso, bool line_fast, bool inverse, const cff* tw) {
#define TPIV_RADIX(r)                                                                                                   \
    case r: radix_pass<r, TWIDDLE>(src, dst, n, nlines, per_line, lsi, esi, bsi, lso, eso, bso, line_fast, inverse, tw); break;
    switch (R) {
        TPIV_RADIX(2) TPIV_RADIX(3) TPIV_RADIX(4) TPIV_RADIX(5) TPIV_RADIX(6) TPIV_RADIX(7) TPIV_RADIX(8)
        default: break;
    }
#undef TPIV_RADIX
}

// the length-n transform of every line of the tile `a` (n lines; element (line, pos) at line * ls + pos * es), in place from
// the caller's view: step 1 into `tmp` (lines of n, contiguous), step 2 back into `a`.
__device__ __forceinline__ void axis_transform(cff* a, cff* tmp, int n, int n1, int n2, int ls, int es, bool inverse,
                                               const cff* tw) {
    const bool line_fast = ls == 1;
    // step 1: n2 radix-n1 butterflies per line over a: in[a_] = a[line, n2 a_ + b]; tmp[line, k1 n2 + b] = (...) w_n^(b k1)
    radix_dispatch<true>(n1, a, tmp, n, n, n2, ls, n2 * es, es, n, n2, 1, line_fast, inverse, tw);
    __syncthreads();
    // step 2: n1 radix-n2 butterflies per line over b: in[b] = tmp[line, k1 n2 + b]; a[line, k1 + n1 k2] = out[k2]
    radix_dispatch<false>(n2, tmp, a, n, n, n1, n, 1, n2, ls, n1 * es, es, false, inverse, tw);
    __syncthreads();
}

// byte offset of the per-window data of the compile-time form inside the dynamic LDS (behind the tiles and the patches)
__host__ __device__ constexpr size_t ct_register_meta_offset(int n, int wpw) {
    return ((size_t)wpw * n * (n | 1) * 8 + 2 * (size_t)(n + 4) * ((n + 7) & ~3) + 15) / 16 * 16;
}
// windows per wavefront of the compile-time form: 64 / n lane groups of n lanes (two windows of 28, five of 12) -- but one for
// CWS: its staging (patch fetch + reference-order bilinear samples, all 64 lanes, window after window) is most of the pass,
// and the packed form's registers (161 -> 208 at 28) cost it the third wavefront per SIMD: 28x28 CWS 155 -> 180 us per pair
// packed, while DWS goes 124 -> 83
// (TPIV_CT_PACK_CWS=1: several windows per wavefront for CWS too -- measured again in round 5 with the sampling tables: 28 x 28 CWS
//  148 -> 156 us per pair, same registers (244); stays off)
#ifndef TPIV_CT_PACK_CWS
#define TPIV_CT_PACK_CWS 0
#endif
__host__ __device__ constexpr int ct_register_wpw(int n, int mode) { return (n > 0 && n <= 64 && (mode != MODE_CWS || TPIV_CT_PACK_CWS)) ? 64 / n : 1; }
// NC > 0: the window size as a compile-time constant -- ONE wavefront per window, lane = line, the four transforms as
// in-register mixed-radix codelets (fft_mixed.hpp) with LDS only for the two transpositions; staging and peak analysis are
// the loops of the run-time form with n known to the compiler.  NC = 0: the run-time form (any even n whose factors fit).
template <int MODE, int NC>
__global__ __launch_bounds__(NC > 0 ? 64 : CT_T) void xcorr_generic_ct_kernel(PassParams p, int n1, int n2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ct_smem[];
    const int n = NC > 0 ? NC : p.ws, nn = n * n, P = n | 1;      // tile pitch: odd (rows and columns both conflict-poor)
    // compile-time form: WPW windows per wavefront (lane groups of NC lanes), one tile each
    constexpr int WPW = ct_register_wpw(NC, MODE);
    cff* T0 = reinterpret_cast<cff*>(ct_smem);
    cff* T1 = T0 + WPW * n * P;                             // (NC > 0: no second tile -- the patches sit behind the WPW first ones)
    cff* tw = T1 + n * P;                                   // exp(-2 pi i k / n), k < n
    float* red = reinterpret_cast<float*>(tw + n);          // 2 x CT_WAVES reduction slots (run-time form)
    // compile-time form: per window of the wavefront {mean a, mean b, 1 / mean a, 1 / mean b, dead | stored << 1, decision band
    // of the exact scheme, record index}
    float* wmeta = reinterpret_cast<float*>(ct_smem + ct_register_meta_offset(n, WPW));
    // candidate lists of the exact scheme (map_candidates): 16 ints behind the reduction slots / the per-window data
    int* cscr = NC > 0 ? reinterpret_cast<int*>(wmeta + WPW * 8) : reinterpret_cast<int*>(red + 2 * CT_WAVES);
    // CWS: the separable sampling tables (cws_axis): columns / rows of frame a, columns / rows of frame b, n entries each, + a flag
    CwsAxis* const axes = reinterpret_cast<CwsAxis*>(cscr + 16);
    float band_abs = 0.f;
    const int PD = n + 4;                                   // CWS: source patch of a shifted window incl. the interpolation margin
    const int PDP = (PD + 3) & ~3;                          // its row pitch in LDS (rows start on dword boundaries)
    uint8_t* patch_a = reinterpret_cast<uint8_t*>(T1);      // (the patches live in the second tile's memory: it is idle until the
    uint8_t* patch_b = patch_a + PD * PDP;                  //  first transform -- more LDS would cost a resident wavefront per CU)
    const int tid = threadIdx.x;
    const int N = p.n_rows * p.n_cols;
    const long long items = (long long)p.batch * N;
    const int HW = p.H * p.W;
    const int st = n - p.ov;
    const float rcp_n = 1.0f / (float)n;
    if constexpr (NC == 0) {
        for (int k = tid; k < n; k += (int)blockDim.x) {
            double s, c;
            sincospi(2.0 * (double)k / (double)n, &s, &c);
            tw[k] = cff{(float)c, (float)(-s)};
        }
        __syncthreads();
    }

    const long long groups = (items + WPW - 1) / WPW;
    for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        // per window of the group: staging into its tile, then means, normalisation and record index into `wmeta` (a ROLLED
        // loop: unrolled, the compiler overlaps the windows' sampling code and the CWS instance goes from 161 to 247 registers)
        float ma = 0.f, mb = 0.f, ka = 1.f, kb = 1.f;
        bool dead = false;
        size_t fidx = 0;
#pragma unroll 1
        for (int wi = 0; wi < WPW; ++wi) {
        const bool act = grp * WPW + wi < items;              // (a slot past the last window re-does it; its record is not stored)
        const long long item = act ? grp * WPW + wi : items - 1;
        cff* const Tw = T0 + wi * n * P;
        const int pair = (int)(item / N), win = (int)(item % N);
        const int y0 = (win / p.n_cols) * st, x0 = (win % p.n_cols) * st;
        const uint8_t* __restrict__ fa = p.A + (size_t)pair * HW;
        const uint8_t* __restrict__ fb = p.B + (size_t)pair * HW;
        fidx = (size_t)item;
        float vx = 0.f, vy = 0.f;
        long long sh = 0;
        if constexpr (MODE == MODE_DWS) {
            double sx, sy;
            pred_half_shift<MODE_DWS>(p, fidx, sx, sy);
            sh = (long long)sy * p.W + (long long)sx;
        }
        if constexpr (MODE == MODE_CWS) pred_half_shift_cws_f32(p, fidx, vx, vy);
        if constexpr (MODE == MODE_CWSF) {      // B:644-645
            vx = (float)(p.u0[fidx] / (double)n);
            vy = (float)(p.v0[fidx] / (double)n);
        }
        // ---- staging (same sampling arithmetic as xcorr_generic_kernel).  CWS: the (n + 4)^2 source pixels of the shifted
        //      window go into LDS first -- independent byte loads, all in flight together -- and the bilinear samples read
        //      their corners there (per sample four dependent global byte loads before: 230 of 600 us per pair at 42 x 42)
        int bxa = 0, bya = 0, bxb = 0, byb = 0;
        if constexpr (MODE == MODE_CWS) {
            bxa = x0 + f2i_sat_g(floorf(-vx)) - 1;
            bya = y0 + f2i_sat_g(floorf(-vy)) - 1;
            bxb = x0 + f2i_sat_g(floorf(vx)) - 1;
            byb = y0 + f2i_sat_g(floorf(vy)) - 1;
            // a patch that lies inside the frame with its pitch-padded rows (every window but the border ones, every predictor
            // but a wild one) is fetched four pixels per load; its pixels are the same flat indices either way (B:177-180)
            const long long lo_a = (long long)bya * p.W + bxa, hi_a = (long long)(bya + PD - 1) * p.W + bxa + PDP - 1;
            const long long lo_b = (long long)byb * p.W + bxb, hi_b = (long long)(byb + PD - 1) * p.W + bxb + PDP - 1;
            if (lo_a >= 0 && lo_b >= 0 && hi_a <= (long long)HW - 1 && hi_b <= (long long)HW - 1) {      // (wave-uniform)
                const int ndw = PDP >> 2;
                const float rcp_dw = 1.0f / (float)ndw;
#pragma unroll 4
                for (int i = tid; i < PD * ndw; i += (int)blockDim.x) {
                    const int py = div_small(i, rcp_dw), k = i - py * ndw;
                    uint32_t da, db;
                    __builtin_memcpy(&da, fa + lo_a + (long long)py * p.W + 4 * k, 4);
                    __builtin_memcpy(&db, fb + lo_b + (long long)py * p.W + 4 * k, 4);
                    reinterpret_cast<uint32_t*>(patch_a)[py * ndw + k] = da;
                    reinterpret_cast<uint32_t*>(patch_b)[py * ndw + k] = db;
                }
            } else {
                const float rcp_pd = 1.0f / (float)PD;
#pragma unroll 4
                for (int i = tid; i < PD * PD; i += (int)blockDim.x) {
                    const int py = div_small(i, rcp_pd), px = i - py * PD;
                    patch_a[py * PDP + px] = (uint8_t)fetch_clamped_g(fa, (long long)(bya + py) * p.W + (bxa + px), HW);
                    patch_b[py * PDP + px] = (uint8_t)fetch_clamped_g(fb, (long long)(byb + py) * p.W + (bxb + px), HW);
                }
            }
            // the separable tables of this window: thread t < n makes the entries of column t and of row t (both frames)
            int* const tab_ok = reinterpret_cast<int*>(axes + 4 * n);
            if (tid == 0) *tab_ok = 1;
            __syncthreads();
            if (tid < n) {
                bool i0, i1, i2, i3;
                axes[tid] = cws_axis(x0 + tid, -vx, bxa, PD, 1, i0);
                axes[n + tid] = cws_axis(y0 + tid, -vy, bya, PD, PDP, i1);
                axes[2 * n + tid] = cws_axis(x0 + tid, vx, bxb, PD, 1, i2);
                axes[3 * n + tid] = cws_axis(y0 + tid, vy, byb, PD, PDP, i3);
                if (!(i0 & i1 & i2 & i3)) *tab_ok = 0;          // (a corner outside the patch -- a wild predictor: the per-sample form below)
            }
            __syncthreads();
        }
        bool tabs = false;
        if constexpr (MODE == MODE_CWS) tabs = *reinterpret_cast<int*>(axes + 4 * n) != 0;
        float sa = 0, sb = 0;
        unsigned iaa = 0u, ibb = 0u;      // first pass: sums of squares of the bytes (exact; the band of the exact scheme)
#pragma unroll 4
        for (int i = tid; i < nn; i += (int)blockDim.x) {
            const int y = div_small(i, rcp_n), x = i - y * n;
            float a, b;
            if constexpr (MODE == MODE_PASS1) {
                const size_t q = (size_t)(y0 + y) * p.W + x0 + x;
                const unsigned ba = fa[q], bb = fb[q];
                a = (float)ba;
                b = (float)bb;
                iaa += ba * ba;
                ibb += bb * bb;
            } else if constexpr (MODE == MODE_DWS) {
                const long long q = (long long)(y0 + y) * p.W + x0 + x;
                a = fetch_clamped_g(fa, q - sh, HW);
                b = fetch_clamped_g(fb, q + sh, HW);
            } else if constexpr (MODE == MODE_CWS) {
                if (tabs) {       // (wave-uniform)
                    a = cws_sample_tab(patch_a, axes[x], axes[n + y]);
                    b = cws_sample_tab(patch_b, axes[2 * n + x], axes[3 * n + y]);
                } else {
                    a = cws_sample_patch(patch_a, PD, PDP, bxa, bya, fa, HW, p.W, x0 + x, y0 + y, -vx, -vy);
                    b = cws_sample_patch(patch_b, PD, PDP, bxb, byb, fb, HW, p.W, x0 + x, y0 + y, vx, vy);
                }
            } else {
                a = bicubic_local(fa, p.W, y0, x0, n, x, y, -vx, -vy);
                b = bicubic_local(fb, p.W, y0, x0, n, x, y, vx, vy);
            }
            Tw[y * P + x] = cff{a, b};
            sa += a;
            sb += b;
            if (p.dbg_win != nullptr) {
                p.dbg_win[fidx * 2 * nn + i] = a;
                p.dbg_win[fidx * 2 * nn + nn + i] = b;
            }
        }
        sa = ct_sum(sa, red);
        sb = ct_sum(sb, red);
        ma = sa / (float)nn, mb = sb / (float)nn;
        dead = false;
        ka = 1, kb = 1;
        if constexpr (MODE == MODE_PASS1 || MODE == MODE_CWSF) {      // a / mean(a): B:513-514, B:656-657
            dead = (sa == 0) || (sb == 0);
            ka = dead ? 0.f : 1.f / ma;
            kb = dead ? 0.f : 1.f / mb;
        }
        if constexpr (MODE == MODE_PASS1) {
            if (p.cand != nullptr) {
                const unsigned saa = ct_sum_u(iaa, red), sbb = ct_sum_u(ibb, red);
                band_abs = dead ? 0.f : p.exact_band * exact_e_plus((unsigned)sa, (unsigned)sb, saa, sbb, nn);
            }
        }
        if constexpr (NC > 0) {
            if (tid == 0) {
                float* mt = wmeta + wi * 8;
                mt[0] = ma, mt[1] = mb, mt[2] = ka, mt[3] = kb;
                mt[4] = __int_as_float((dead ? 1 : 0) | (act ? 2 : 0)), mt[5] = band_abs;
                mt[6] = __int_as_float((int)(unsigned)fidx), mt[7] = __int_as_float((int)(unsigned)(fidx >> 32));
            }
        }
        __syncthreads();                                      // (tile complete; the patches are free for the next window)
        }
        float* map = reinterpret_cast<float*>(T0);
        float cmin = 3.4e38f;
        const int hshift = n / 2;
        if constexpr (NC > 0) {
            // ---- lane = line of window wq; every transform on registers, the tile only for the two transpositions
            using fmx::fft_mixed;
            const bool on = tid < WPW * NC;                   // (the lanes behind the last group idle along with lane 0's data; their stores are masked)
            const int wq = on ? tid / NC : 0;
            const int r = on ? tid - wq * NC : 0;
            const int base = wq * NC;                         // first lane of the window's group
            cff* const T0 = reinterpret_cast<cff*>(ct_smem) + wq * n * P;       // this lane's window: tile, then map
            float* const map = reinterpret_cast<float*>(T0);
            const float* mt = wmeta + wq * 8;                 // this lane's window
            const float ma = mt[0], mb = mt[1], ka = mt[2], kb = mt[3];
            const bool dead = (__float_as_int(mt[4]) & 1) != 0, act = (__float_as_int(mt[4]) & 2) != 0;
            const size_t fidx = (size_t)(unsigned)__float_as_int(mt[6]) | ((size_t)(unsigned)__float_as_int(mt[7]) << 32);
            const int mirror = base + (NC - r) % NC;          // the lane that holds column -kx
            // reductions over the lanes of the window (groups of NC lanes, any NC): halving steps towards the group's
            // first lane -- lane r takes lane r + o while that lane belongs to the group --, then a broadcast
            auto seg = [&](auto v, auto&& op) TPIV_LAMBDA_INLINE {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const auto t_ = __shfl_down(v, o, 64);
                    v = (o < NC && r + o < NC) ? op(v, t_) : v;
                }
                return __shfl(v, base, 64);
            };
            cf x[NC];
            static_for<0, NC>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                const cff z = T0[r * P + k];
                // mean removal (conditions the float32 transform; corr - min is unchanged by it) and the normalisation
                x[k] = cf{(z.x - ma) * ka, (z.y - mb) * kb};
            });
            fft_mixed<NC, 1>(x);                              // over x: bin kx at slot MIXED_POS<kx>
            if (on) {                                         // (each lane rewrites its own row: no other lane reads it)
                static_for<0, NC>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    T0[r * P + k] = cff{x[fmx::MIXED_POS<k, NC>].x, x[fmx::MIXED_POS<k, NC>].y};
                });
            }
            __syncthreads();
            static_for<0, NC>([&](auto yc) TPIV_LAMBDA_INLINE {       // lane = kx: column of the row transforms
                constexpr int y = decltype(yc)::value;
                const cff z = T0[y * P + r];
                x[y] = cf{z.x, z.y};
            });
            fft_mixed<NC, 1>(x);                              // over y: Z(ky, kx = lane) at slot MIXED_POS<ky>
            // cross-spectrum P = conj(A) B / n^2 of the packed transform; Z(-ky, -kx) sits in lane `mirror`, slot of -ky
            {
                const float scale = 0.25f / (float)(NC * NC);
                auto cross = [&](cf zk, cf zm) TPIV_LAMBDA_INLINE {
                    cf pr;
                    pr.x = (zk.x * zm.y + zk.y * zm.x) * (2.0f * scale);
                    pr.y = ((zm.x * zm.x - zk.x * zk.x) + (zm.y * zm.y - zk.y * zk.y)) * scale;
                    return pr;
                };
                static_for<0, NC / 2 + 1>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int ky = decltype(kc)::value;
                    constexpr int nky = (NC - ky) % NC;
                    constexpr int p1 = fmx::MIXED_POS<ky, NC>, p2 = fmx::MIXED_POS<nky, NC>;
                    const cf z1 = x[p1];
                    if constexpr (ky == nky) {
                        const cf m1{__shfl(z1.x, mirror, 64), __shfl(z1.y, mirror, 64)};
                        x[p1] = cross(z1, m1);
                    } else {
                        const cf z2 = x[p2];
                        const cf m1{__shfl(z2.x, mirror, 64), __shfl(z2.y, mirror, 64)};      // Z(-ky, -kx)
                        const cf m2{__shfl(z1.x, mirror, 64), __shfl(z1.y, mirror, 64)};      // Z(+ky, -kx): the mirror of bin -ky
                        x[p1] = cross(z1, m1);
                        x[p2] = cross(z2, m2);
                    }
                });
            }
            cf t[NC];
            static_for<0, NC>([&](auto kc) TPIV_LAMBDA_INLINE { t[decltype(kc)::value] = x[fmx::MIXED_POS<decltype(kc)::value, NC>]; });
            fft_mixed<NC, -1>(t);                             // inverse over ky: row y at slot MIXED_POS<y>
            __syncthreads();                                  // (every lane has read its column)
            if (on) {
                static_for<0, NC>([&](auto yc) TPIV_LAMBDA_INLINE {
                    constexpr int y = decltype(yc)::value;
                    T0[y * P + r] = cff{t[fmx::MIXED_POS<y, NC>].x, t[fmx::MIXED_POS<y, NC>].y};
                });
            }
            __syncthreads();
            static_for<0, NC>([&](auto kc) TPIV_LAMBDA_INLINE {       // lane = y: row of the spectrum over kx
                constexpr int k = decltype(kc)::value;
                const cff z = T0[r * P + k];
                x[k] = cf{z.x, z.y};
            });
            fft_mixed<NC, -1>(x);                             // inverse over kx: corr(y = lane, x) = real part at slot MIXED_POS<x>
            __syncthreads();                                  // (rows read: the tile becomes the map)
            // ---- peak analysis with the map row of the lane in registers: the decisions of the sweeps below (first flat
            //      index of the maximum, largest cell outside the flat-index neighbourhood), without their passes over LDS
            int ys = r + hshift;
            ys -= ys >= n ? n : 0;
            float rowv[NC];                                   // fftshift column order
            float rmn = 3.4e38f;
            static_for<0, NC>([&](auto xc) TPIV_LAMBDA_INLINE {
                constexpr int xs = decltype(xc)::value;
                rowv[xs] = x[fmx::MIXED_POS<(xs + NC / 2) % NC, NC>].x;
                rmn = fminf(rmn, rowv[xs]);
            });
            cmin = seg(on ? rmn : 3.4e38f, [](float a_, float b_) TPIV_LAMBDA_INLINE { return fminf(a_, b_); });
            float rmx = -1.f;
            static_for<0, NC>([&](auto xc) TPIV_LAMBDA_INLINE {
                constexpr int xs = decltype(xc)::value;
                rowv[xs] = add_eps(rowv[xs], cmin);           // B:518, B:381
                rmx = fmaxf(rmx, rowv[xs]);
            });
            if (on) {
                static_for<0, NC>([&](auto xc) TPIV_LAMBDA_INLINE { map[ys * NC + decltype(xc)::value] = rowv[decltype(xc)::value]; });
                if (p.dbg_corr != nullptr) {
                    static_for<0, NC>([&](auto xc) TPIV_LAMBDA_INLINE {
                        p.dbg_corr[fidx * nn + ys * NC + decltype(xc)::value] = rowv[decltype(xc)::value];
                    });
                }
            }
            const float gmax = seg(on ? rmx : -1.f, [](float a_, float b_) TPIV_LAMBDA_INLINE { return fmaxf(a_, b_); });
            // first flat index holding the maximum (B:383): smallest row, then smallest column of that row
            const int ywin = seg((on && rmx == gmax) ? ys : NC, [](int a_, int b_) TPIV_LAMBDA_INLINE { return a_ < b_ ? a_ : b_; });
            int xf = 0;
            static_for<0, NC>([&](auto xc) TPIV_LAMBDA_INLINE {
                constexpr int xs = NC - 1 - decltype(xc)::value;
                xf = rowv[xs] == gmax ? xs : xf;
            });
            int row_lane = ywin + NC - hshift;                // the lane that holds map row ywin
            row_lane -= row_lane >= NC ? NC : 0;
            const int xwin = __shfl(xf, base + (ywin < NC ? row_lane : 0), 64);
            const int m = ywin < NC ? ywin * NC + xwin : 0;   // (no cell compared equal: an all-NaN map keeps index 0)
            // largest cell outside the flat-index neighbourhood of m (B:346-358): q = clamp(m + i + n j), |i|, |j| <= wv -- in
            // row y' the columns mx+i (j = y'-my), mx+i+n (j = y'-my+1), mx+i-n (j = y'-my-1), plus the two clamps
            const int wv = p.val_win;
            int smax = 0;                                     // every cell is > 0: positive floats order like their bits
            {
                const int my_ = m / NC, mx_ = m - my_ * NC, dj = ys - my_;
                unsigned long long ex = 0ull;
                auto span = [&](int lo_, int hi_) TPIV_LAMBDA_INLINE {
                    lo_ = lo_ < 0 ? 0 : lo_;
                    hi_ = hi_ > NC - 1 ? NC - 1 : hi_;
                    if (lo_ > hi_) return 0ull;
                    const unsigned long long ones = (hi_ - lo_ + 1) >= 64 ? ~0ull : ((1ull << (hi_ - lo_ + 1)) - 1ull);
                    return ones << lo_;
                };
                if (dj >= -wv && dj <= wv) ex |= span(mx_ - wv, mx_ + wv);
                if (dj + 1 >= -wv && dj + 1 <= wv) ex |= span(mx_ - wv + NC, mx_ + wv + NC);
                if (dj - 1 >= -wv && dj - 1 <= wv) ex |= span(mx_ - wv - NC, mx_ + wv - NC);
                if (ys == 0 && (m - wv - wv * NC) <= 0) ex |= 1ull;
                if (ys == NC - 1 && (m + wv + wv * NC) >= nn - 1) ex |= 1ull << (NC - 1);
                const int exl = (int)(unsigned)ex, exh = (int)(unsigned)(ex >> 32);
                static_for<0, NC>([&](auto xc) TPIV_LAMBDA_INLINE {
                    constexpr int xs = decltype(xc)::value;
                    const int kill = __builtin_amdgcn_sbfe(xs < 32 ? exl : exh, xs & 31, 1);
                    const int cnd = __float_as_int(rowv[xs]) | kill;
                    smax = cnd > smax ? cnd : smax;
                });
            }
            smax = seg(on ? smax : 0, [](int a_, int b_) TPIV_LAMBDA_INLINE { return a_ > b_ ? a_ : b_; });
            __syncthreads();                                  // (the map is complete)
            if constexpr (MODE == MODE_PASS1) {
                if (p.cand != nullptr) {      // precision "exact": candidate cells instead of the record, window after window
#pragma unroll 1
                    for (int wi = 0; wi < WPW; ++wi) {
                        const int src = wi * NC;                                  // first lane of window wi
                        const float* mw = wmeta + wi * 8;
                        const int flags = __float_as_int(mw[4]);
                        const size_t fw = (size_t)(unsigned)__float_as_int(mw[6]) | ((size_t)(unsigned)__float_as_int(mw[7]) << 32);
                        const int sw = __shfl(smax, src, 64);
                        map_candidates(p, reinterpret_cast<const float*>(reinterpret_cast<cff*>(ct_smem) + wi * n * P), NC,
                                       __shfl(m, src, 64), __shfl(gmax, src, 64), __int_as_float(sw), sw > 0, mw[5],
                                       (flags & 1) != 0, (flags & 2) != 0, fw, cscr);
                    }
                    continue;
                }
            }
            if (on && r < 8 && act) {
                const int tid = r;                            // (record slot)
                int left = m + 1, right = m - 1, top = m + n, bot = m - n;    // B:385-392
                if (left >= nn - 1) left = m;
                if (right <= 0) right = m;
                if (top >= nn - 1) top = m;
                if (bot <= 0) bot = m;
                int q = m;
                q = (tid == 1) ? left : q;
                q = (tid == 2) ? right : q;
                q = (tid == 3) ? top : q;
                q = (tid == 4) ? bot : q;
                q = (tid == 5) ? 0 : q;
                float outv = map[q];
                // (nothing left outside the zone: the reference's second arg-max returns index 0 -- see xcorr_generic_kernel)
                if (tid == 5) outv = smax > 0 ? __int_as_float(smax) : (MODE == MODE_PASS1 ? 0.f : outv);
                outv = (tid == 6) ? __int_as_float(m) : outv;
                outv = (tid == 7) ? __int_as_float(dead ? 1 : 0) : outv;
                p.peak_raw[fidx * 8 + tid] = outv;
            }
            __syncthreads();
            continue;
        } else {
        // mean removal (conditions the float32 transform; corr - min is unchanged by it) and the normalisation
#pragma unroll 4
        for (int i = tid; i < nn; i += (int)blockDim.x) {
            const int y = div_small(i, rcp_n), x = i - y * n;
            const cff z = T0[y * P + x];
            T0[y * P + x] = cff{(z.x - ma) * ka, (z.y - mb) * kb};
        }
        __syncthreads();
        axis_transform(T0, T1, n, n1, n2, P, 1, false, tw);          // rows:    T0[y][kx]
        axis_transform(T0, T1, n, n1, n2, 1, P, false, tw);          // columns: T0[ky][kx]
        // ---- cross-spectrum: T1[k] = conj(A) * B / n^2 with A, B split out of Z = FFT2(a + i b)
        const float scale = 0.25f / (float)nn;
#pragma unroll 4
        for (int i = tid; i < nn; i += (int)blockDim.x) {
            const int ky = div_small(i, rcp_n), kx = i - ky * n;
            const cff zk = T0[ky * P + kx];
            const cff zm = T0[(ky ? n - ky : 0) * P + (kx ? n - kx : 0)];
            cff pr;
            pr.x = (zk.x * zm.y + zk.y * zm.x) * (2.0f * scale);
            pr.y = ((zm.x * zm.x - zk.x * zk.x) + (zm.y * zm.y - zk.y * zk.y)) * scale;
            T1[ky * P + kx] = pr;
        }
        __syncthreads();
        axis_transform(T1, T0, n, n1, n2, 1, P, true, tw);           // inverse over ky: T1[y][kx]
        axis_transform(T1, T0, n, n1, n2, P, 1, true, tw);           // inverse over kx: T1[y][x] (the real part is the map)
        // ---- map in fftshift coordinates (into T0's memory), minimum
#pragma unroll 4
        for (int i = tid; i < nn; i += (int)blockDim.x) {
            const int y = div_small(i, rcp_n), x = i - y * n;
            const float re = T1[y * P + x].x;
            int ys = y + hshift, xs = x + hshift;
            ys -= ys >= n ? n : 0;
            xs -= xs >= n ? n : 0;
            map[ys * n + xs] = re;
            cmin = rmin(cmin, re);
        }
        }       // NC == 0
        __syncthreads();
        cmin = ct_min(cmin, red);
        // ---- corr - min + eps (B:518, B:381), first peak
        AM<float> best{-1.f, 0};
#pragma unroll 4
        for (int i = tid; i < nn; i += (int)blockDim.x) {
            const float v = add_eps(map[i], cmin);
            map[i] = v;
            if (p.dbg_corr != nullptr) p.dbg_corr[fidx * nn + i] = v;
            if (v > best.v) {
                best.v = v;
                best.idx = i;
            }
        }
        best = ct_argmax(best, red);
        __syncthreads();
        const int m = best.idx;
        // ---- second peak outside the flat-index neighbourhood (B:346-358)
        const int wv = p.val_win;
        AM<float> second{-1.f, nn};
#pragma unroll 4
        for (int i = tid; i < nn; i += (int)blockDim.x) {
            // i = m + t + n j with |t|, |j| <= wv (2 wv < n: at most one such pair), or one of the two clamps
            const int d = i - m + wv;                         // = t' + n j with t' = t + wv in [0, 2 wv]
            const int j = (int)floorf(((float)d + 0.5f) * rcp_n);
            const int t = d - j * n;                          // 0 <= t < n
            bool excl = (t <= 2 * wv) & (j >= -wv) & (j <= wv);
            excl |= (i == 0) & ((m - wv - wv * n) <= 0);
            excl |= (i == nn - 1) & ((m + wv + wv * n) >= nn - 1);
            const float v = map[i];
            if (!excl && v > second.v) {
                second.v = v;
                second.idx = i;
            }
        }
        second = ct_argmax(second, red);
        __syncthreads();
        if constexpr (MODE == MODE_PASS1) {
            if (p.cand != nullptr) {          // precision "exact": candidate cells instead of the record
                map_candidates(p, map, n, m, best.v, second.v, second.idx < nn, band_abs, dead, true, fidx, cscr);
                continue;
            }
        }
        if (tid < 8) {
            int left = m + 1, right = m - 1, top = m + n, bot = m - n;    // B:385-392
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
            if (tid == 5 && second.idx >= nn && MODE == MODE_PASS1) outv = 0;       // (see xcorr_generic_kernel)
            outv = (tid == 6) ? __int_as_float(m) : outv;
            outv = (tid == 7) ? __int_as_float(dead ? 1 : 0) : outv;
            p.peak_raw[fidx * 8 + tid] = outv;
        }
        __syncthreads();
    }
}

static int ct_threads(int n) { return n >= 40 ? CT_T : 64; }
// window sizes with a compile-time instance of the kernel (in-register transforms): the sizes multipass scales of 1.5 and
// 1.33 make of 128 / 64 / 48 / 32 first passes, each in the three reachable modes; anything else takes the run-time form.
// (TPIV_GENERIC_REG=0: run-time form for every size -- A/B runs)
#ifndef TPIV_CT_REGISTER_SIZES
#define TPIV_CT_REGISTER_SIZES(X) X(12) X(14) X(18) X(20) X(24) X(28) X(30) X(36) X(40) X(42) X(48) X(56)
#endif
static bool ct_register_size(int n) {
    static const bool off = [] { const char* e_ = getenv("TPIV_GENERIC_REG"); return e_ && e_[0] == '0'; }();
    if (off) return false;
#define TPIV_CTR_IS(NCV) if (n == NCV) return true;
    TPIV_CT_REGISTER_SIZES(TPIV_CTR_IS)
#undef TPIV_CTR_IS
    return false;
}
// (one tile per window of a wavefront, one pair of CWS patches, and eight floats of per-window data)
static size_t ct_register_smem(int n, int mode) {
    return ct_register_meta_offset(n, ct_register_wpw(n, mode)) + (size_t)ct_register_wpw(n, mode) * 8 * sizeof(float) + 16 * sizeof(int) +
           (mode == MODE_CWS ? (size_t)(4 * n + 1) * 16 : 0);
}
constexpr int CT_MAX_RADIX = 8;
// n = n1 n2 with 2 <= n1 <= n2 <= CT_MAX_RADIX, n1 as large as possible; false if there is no such split
bool ct_factors(int n, int& n1, int& n2) {
    n1 = 0;
    for (int a = 2; a * a <= n; ++a)
        if (n % a == 0 && n / a <= CT_MAX_RADIX) n1 = a;
    if (n1 == 0) return false;
    n2 = n / n1;
    return n1 >= 2 && n2 <= CT_MAX_RADIX;
}
size_t ct_smem_bytes(int n) { return (size_t)(2 * n * (n | 1) + n) * sizeof(cff) + 2 * CT_WAVES * sizeof(float) + 16 * sizeof(int) + (size_t)(4 * n + 1) * 16; }     // (the two (n + 4) x pitch patches fit the second tile, n >= 4)
bool ct_usable(int n, int precision) {
    int a, b;
    return precision == 0 && (n & 1) == 0 && n >= 4 && n <= 96 && ct_factors(n, a, b) && ct_smem_bytes(n) <= 160 * 1024;
}

}  // namespace

// ... with a compile-time instance?  (the second template argument of the kernel's name: 0 = run-time form)
int generic_ct_register_size(int ws, int mode) { return (mode != MODE_CWSF && ct_usable(ws, 0) && ct_register_size(ws)) ? ws : 0; }

// does (ws, float32) run the second-generation kernel?  (bench / profile labels, piv_launch.hip)
bool generic_ct_usable(int ws) {
    static const bool off = [] { const char* e_ = getenv("TPIV_GENERIC_CT"); return e_ && e_[0] == '0'; }();
    return !off && ct_usable(ws, 0);
}

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
    static const bool ct_off = [] { const char* e_ = getenv("TPIV_GENERIC_CT"); return e_ && e_[0] == '0'; }();     // A/B runs
    if (!ct_off && ct_usable(p.ws, 0)) {
        int n1, n2;
        ct_factors(p.ws, n1, n2);
        const size_t smem = ct_smem_bytes(p.ws);
        const int per_cu = (int)((160 * 1024) / smem) < 12 ? (int)((160 * 1024) / smem) : 12;     // one-wave workgroups, LDS-limited
        long long blocks = (long long)n_cu * (per_cu < 1 ? 1 : per_cu);
        if (blocks > items) blocks = items;
        hipError_t e_ = hipSuccess;
        // sizes with a compile-time instance (in-register transforms, one wavefront per window): every mode but CWS_Fast
        if (mode != MODE_CWSF && ct_register_size(p.ws)) {
            const size_t smem_r = ct_register_smem(p.ws, mode);
            int per_cu_r = (int)((160 * 1024) / smem_r);
            per_cu_r = per_cu_r > 16 ? 16 : (per_cu_r < 1 ? 1 : per_cu_r);
            long long blocks_r = (long long)n_cu * per_cu_r;
            const int wpw_r = ct_register_wpw(p.ws, mode);
            const long long groups_r = (items + wpw_r - 1) / wpw_r;
            if (blocks_r > groups_r) blocks_r = groups_r;
#define TPIV_CTR_LAUNCH(M, NCV)                                                                                         \
    e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&xcorr_generic_ct_kernel<M, NCV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_r); \
    if (e_ == hipSuccess) hipLaunchKernelGGL((xcorr_generic_ct_kernel<M, NCV>), dim3((unsigned)blocks_r), dim3(64), smem_r, stream, p, n1, n2);
#define TPIV_CTR_MODES(NCV)                                                                                             \
    case NCV:                                                                                                           \
        if (mode == MODE_PASS1) { TPIV_CTR_LAUNCH(MODE_PASS1, NCV) }                                                    \
        else if (mode == MODE_DWS) { TPIV_CTR_LAUNCH(MODE_DWS, NCV) }                                                   \
        else { TPIV_CTR_LAUNCH(MODE_CWS, NCV) }                                                                         \
        break;
            switch (p.ws) {
                TPIV_CT_REGISTER_SIZES(TPIV_CTR_MODES)
                default: return hipErrorInvalidValue;
            }
#undef TPIV_CTR_MODES
#undef TPIV_CTR_LAUNCH
            return e_ != hipSuccess ? e_ : hipGetLastError();
        }
#define TPIV_CT_LAUNCH(M)                                                                                               \
    e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&xcorr_generic_ct_kernel<M, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (e_ == hipSuccess) hipLaunchKernelGGL((xcorr_generic_ct_kernel<M, 0>), dim3((unsigned)blocks), dim3(ct_threads(p.ws)), smem, stream, p, n1, n2);
        switch (mode) {
            case MODE_PASS1: TPIV_CT_LAUNCH(MODE_PASS1) break;
            case MODE_DWS: TPIV_CT_LAUNCH(MODE_DWS) break;
            case MODE_CWS: TPIV_CT_LAUNCH(MODE_CWS) break;
            case MODE_CWSF: TPIV_CT_LAUNCH(MODE_CWSF) break;
            default: return hipErrorInvalidValue;
        }
#undef TPIV_CT_LAUNCH
        return e_ != hipSuccess ? e_ : hipGetLastError();
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
