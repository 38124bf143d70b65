// Fused PIV tile kernels for gfx950 (MI355X).
//
// One kernel family does everything between "uint8 frames in HBM" and
// "(u, v, invalid) per interrogation window":
//   strip load (plain / integer-shifted DWS / bilinear CWS) -> mean removal ->
//   packed complex 2-D FFT (both frames in one transform) -> cross-spectrum ->
//   inverse 2-D FFT -> min-subtract -> first peak -> 3-point log-Gaussian fit ->
//   second-peak validation -> multipass combine.
// It replaces the ATen kernel sequence of the reference's
//   extended_search_area_piv        PIVbackend.py:459-520
//   piv_iteration_DWS.__call__      PIVbackend.py:757-812  (interpolation_DWS 197-216)
//   piv_iteration_CWS.__call__      PIVbackend.py:690-740  (biliniar_interpolation_CWS 147-194)
//   correalte_fft                   PIVbackend.py:249-257
//   correlation_to_displacement     PIVbackend.py:360-422
//   peak2peak_secondpeak            PIVbackend.py:346-358
//
// Layout: lane = one image column of one window; the lane holds the whole
// column (WS complex samples, a + i*b) in VGPRs.  Column FFTs run entirely in
// registers (fft_inreg.hpp), the LDS is used only to transpose the tile between
// the column and the row transform, for the k <-> -k exchange of the packed
// spectrum and for the final correlation map.  64/WS windows share a wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_inreg.hpp"
#include "piv_kernels.h"

namespace tpiv {

// ----------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------
template <int WS>
struct Geo {
    static constexpr int TPB = WS >= 64 ? WS : 64;   // threads per block
    static constexpr int WPB = TPB / WS;              // windows per block
    static constexpr int PITCH = WS + 1;              // complex elements per tile row (bank spread)
    static constexpr int TILE = WS * PITCH;           // complex elements per window tile
};

struct ArgMax {
    float v;
    int idx;
};

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
    // larger value wins; on equal values the smaller flat index wins (torch.argmax: first)
    bool takeb = (b.v > a.v) || (b.v == a.v && b.idx < a.idx);
    return takeb ? b : a;
}

// reductions over the WS lanes of one window (WS <= 64: inside a wavefront;
// WS == 128: two wavefronts, combined through a small LDS scratch)
template <int WS>
__device__ __forceinline__ float group_sum(float v, float* scratch) {
    constexpr int L = WS > 64 ? 64 : WS;
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if constexpr (WS > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
        __syncthreads();
        v = scratch[0] + scratch[1];
    }
    return v;
}

template <int WS>
__device__ __forceinline__ float group_min(float v, float* scratch) {
    constexpr int L = WS > 64 ? 64 : WS;
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    if constexpr (WS > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
        __syncthreads();
        v = fminf(scratch[0], scratch[1]);
    }
    return v;
}

template <int WS>
__device__ __forceinline__ ArgMax group_argmax(ArgMax a, float* scratch) {
    constexpr int L = WS > 64 ? 64 : WS;
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) {
        ArgMax o;
        o.v = __shfl_xor(a.v, off, 64);
        o.idx = __shfl_xor(a.idx, off, 64);
        a = better(a, o);
    }
    if constexpr (WS > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            scratch[(threadIdx.x >> 6) * 2] = a.v;
            scratch[(threadIdx.x >> 6) * 2 + 1] = __int_as_float(a.idx);
        }
        __syncthreads();
        ArgMax p{scratch[0], __float_as_int(scratch[1])};
        ArgMax q{scratch[2], __float_as_int(scratch[3])};
        a = better(p, q);
    }
    return a;
}

__device__ __forceinline__ double nan_to_num(double x) {
    // torch.nan_to_num_ defaults (PIVbackend.py:418-419)
    if (x != x) return 0.0;
    if (x > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (x < -1.7976931348623157e308) return -1.7976931348623157e308;
    return x;
}

// flat-index clamped uint8 fetch (PIVbackend.py:177-180, 214)
__device__ __forceinline__ float fetch_clamped(const uint8_t* __restrict__ f, long long q, int HW) {
    q = q < 0 ? 0 : (q > (long long)(HW - 1) ? (long long)(HW - 1) : q);
    return (float)f[q];
}

__device__ __forceinline__ int f2i_sat(float v) {
    // float -> int with saturation (garbage predictors must not index out of range)
    v = fminf(fmaxf(v, -1073741824.f), 1073741824.f);
    return (int)v;
}

// One bilinear sample exactly as PIVbackend.py:187-193 evaluates it in float32:
// ((f11*wxu)*wyu + (f21*wxd)*wyu) + (f12*wxu)*wyd + (f22*wxd)*wyd, left to right, every
// product and sum rounded separately (torch runs them as separate element-wise kernels),
// so FMA contraction must stay off here.
__device__ __forceinline__ float cws_sample(const uint8_t* __restrict__ f, int HW, int W, int dy, int uy,
                                            int dx, int ux, float wx_up, float wx_dn, float wy_up,
                                            float wy_dn) {
#pragma clang fp contract(off)
    const float f11 = fetch_clamped(f, (long long)dy * W + dx, HW);
    const float f21 = fetch_clamped(f, (long long)dy * W + ux, HW);
    const float f12 = fetch_clamped(f, (long long)uy * W + dx, HW);
    const float f22 = fetch_clamped(f, (long long)uy * W + ux, HW);
    float r = (f11 * wx_up) * wy_up;
    r = r + (f21 * wx_dn) * wy_up;
    r = r + (f12 * wx_up) * wy_dn;
    r = r + (f22 * wx_dn) * wy_dn;
    const bool degenerate = ((long long)(ux - dx) * (long long)(uy - dy)) == 0;    // B:170, B:193
    return degenerate ? f11 : r;
}

// ----------------------------------------------------------------------------
// the tile kernel
// ----------------------------------------------------------------------------
template <int WS, int MODE>
__global__ __launch_bounds__(Geo<WS>::TPB, (WS <= 32 ? 2 : 1)) void xcorr_kernel(PassParams p) {
    using G = Geo<WS>;
    constexpr int PITCH = G::PITCH;
    __shared__ cf tile[G::WPB * G::TILE];
    __shared__ float scratch[8];

    const int tid = threadIdx.x;
    const int w = tid / WS;       // window slot inside the block
    const int c = tid % WS;       // column (then row) handled by this lane
    cf* my_tile = tile + w * G::TILE;
    float* my_map = reinterpret_cast<float*>(my_tile);     // WS*WS floats, reused after the FFTs

    const int N = p.n_rows * p.n_cols;
    const int groups = (N + G::WPB - 1) / G::WPB;
    const long long items = (long long)p.batch * groups;
    const int HW = p.H * p.W;
    const int st = p.ws - p.ov;

    // XCD-aware item order: blocks b, b+8, ... share an XCD (and its L2); give each
    // XCD one contiguous run of windows so that overlapping windows hit the same L2.
    const int nb = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int per_xcd_blocks = nb >> 3;                      // host guarantees nb % 8 == 0
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = (lo + chunk < items) ? lo + chunk : items;

    for (long long item = lo + slot; item < hi; item += per_xcd_blocks) {
        const int pair = (int)(item / groups);
        const int g = (int)(item % groups);
        const int win_raw = g * G::WPB + w;
        const bool active = win_raw < N;
        const int win = active ? win_raw : N - 1;
        const int wr = win / p.n_cols, wc = win % p.n_cols;
        const int y0 = wr * st, x0 = wc * st;
        const uint8_t* __restrict__ fa = p.A + (size_t)pair * HW;
        const uint8_t* __restrict__ fb = p.B + (size_t)pair * HW;
        const size_t fidx = (size_t)pair * N + win;

        cf x[WS];

        // ---- stage 0: window samples -> registers (lane = column, coalesced along x)
        if constexpr (MODE == MODE_PASS1) {
            const uint8_t* pa = fa + (size_t)y0 * p.W + x0 + c;
            const uint8_t* pb = fb + (size_t)y0 * p.W + x0 + c;
#pragma unroll
            for (int y = 0; y < WS; ++y) {
                x[y].x = (float)pa[(size_t)y * p.W];
                x[y].y = (float)pb[(size_t)y * p.W];
            }
        } else if constexpr (MODE == MODE_DWS) {
            // integer shift on the flat index: a sampled at idx - (vy*W + vx), b at idx + ...
            const long long sx = (long long)p.u2[fidx], sy = (long long)p.v2[fidx];
            const long long off = sy * p.W + sx;
            const long long base = (long long)y0 * p.W + x0 + c;
#pragma unroll
            for (int y = 0; y < WS; ++y) {
                long long q = base + (long long)y * p.W;
                x[y].x = fetch_clamped(fa, q - off, HW);
                x[y].y = fetch_clamped(fb, q + off, HW);
            }
        } else {
            // bilinear shift in float32, exactly as PIVbackend.py:162-193 evaluates it
            const float vx = (float)p.u2[fidx], vy = (float)p.v2[fidx];
            const float gxf = (float)(x0 + c);
            // frame a: (-vx, -vy); frame b: (+vx, +vy).  x-direction terms are row-invariant.
            const float nxa = gxf - vx, nxb = gxf + vx;
            const float uxa_f = ceilf(nxa), dxa_f = floorf(nxa);
            const float uxb_f = ceilf(nxb), dxb_f = floorf(nxb);
            const int uxa = f2i_sat(uxa_f), dxa = f2i_sat(dxa_f);
            const int uxb = f2i_sat(uxb_f), dxb = f2i_sat(dxb_f);
            const float wxa_up = uxa_f - nxa, wxa_dn = nxa - dxa_f;
            const float wxb_up = uxb_f - nxb, wxb_dn = nxb - dxb_f;
#pragma unroll
            for (int y = 0; y < WS; ++y) {
                const float gyf = (float)(y0 + y);
                const float nya = gyf - vy, nyb = gyf + vy;
                const float uya_f = ceilf(nya), dya_f = floorf(nya);
                const float uyb_f = ceilf(nyb), dyb_f = floorf(nyb);
                x[y].x = cws_sample(fa, HW, p.W, f2i_sat(dya_f), f2i_sat(uya_f), dxa, uxa, wxa_up, wxa_dn,
                                    uya_f - nya, nya - dya_f);
                x[y].y = cws_sample(fb, HW, p.W, f2i_sat(dyb_f), f2i_sat(uyb_f), dxb, uxb, wxb_up, wxb_dn,
                                    uyb_f - nyb, nyb - dyb_f);
            }
        }

        if (p.dbg_win != nullptr && active) {     // test hook: the staged (shifted) windows
            float* d = p.dbg_win + fidx * 2 * WS * WS;
#pragma unroll
            for (int y = 0; y < WS; ++y) {
                d[y * WS + c] = x[y].x;
                d[WS * WS + y * WS + c] = x[y].y;
            }
        }

        // ---- mean removal (any constant offset leaves corr - min(corr) unchanged; it only
        //      conditions the float32 transform).  Pass 1 also divides by the mean (B:513-514).
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int y = 0; y < WS; ++y) {
            sa += x[y].x;
            sb += x[y].y;
        }
        sa = group_sum<WS>(sa, scratch);
        sb = group_sum<WS>(sb, scratch);
        const float ma = sa * (1.0f / (WS * WS)), mb = sb * (1.0f / (WS * WS));
        bool dead = false;        // pass 1 only: zero-mean window -> 0/0 = NaN map in the reference
        float ka = 1.f, kb = 1.f;
        if constexpr (MODE == MODE_PASS1) {
            dead = (sa == 0.f) || (sb == 0.f);
            ka = dead ? 0.f : 1.0f / ma;
            kb = dead ? 0.f : 1.0f / mb;
        }
#pragma unroll
        for (int y = 0; y < WS; ++y) {
            x[y].x = (x[y].x - ma) * ka;
            x[y].y = (x[y].y - mb) * kb;
        }

        // ---- forward: column FFT (over y) in registers
        fft_inreg<WS, 1>(x);
        // transpose through LDS: tile[ky][kx-as-column c]
        __syncthreads();
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int ky = decltype(kc)::value;
            my_tile[ky * PITCH + c] = x[FFT_POS<ky, WS>];
        });
        __syncthreads();
        // lane now owns row ky = c
#pragma unroll
        for (int kx = 0; kx < WS; ++kx) x[kx] = my_tile[c * PITCH + kx];
        fft_inreg<WS, 1>(x);      // Z(ky = c, kx) at x[fft_pos(kx)]

        // ---- cross-spectrum.  With Z = FFT2(a + i b):  A = (Z(k) + conj Z(-k))/2,
        //      B = (Z(k) - conj Z(-k))/(2i),  P = conj(A) * B.  Z(-k) lives in lane (-c mod WS).
        __syncthreads();
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int kx = decltype(kc)::value;
            my_tile[c * PITCH + kx] = x[FFT_POS<kx, WS>];
        });
        __syncthreads();
        {
            const int nr = (WS - c) % WS;
            constexpr float scale = 0.25f / (float)(WS * WS);     // 1/4 of the split, 1/n^2 of irfft2
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int kx = decltype(kc)::value;
                constexpr int nkx = (WS - kx) % WS;
                const cf zk = x[FFT_POS<kx, WS>];
                const cf zm = my_tile[nr * PITCH + nkx];
                const float a_ = zk.x, b_ = zk.y, c_ = zm.x, d_ = zm.y;
                cf pr;
                pr.x = ((a_ + c_) * (b_ + d_) + (b_ - d_) * (c_ - a_)) * scale;
                pr.y = ((c_ * c_ - a_ * a_) + (d_ * d_ - b_ * b_)) * scale;
                x[FFT_POS<kx, WS>] = pr;
            });
        }
        // ---- inverse: row IFFT needs natural order input -> permute in registers
        {
            cf t[WS];
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int kx = decltype(kc)::value;
                t[kx] = x[FFT_POS<kx, WS>];
            });
            fft_inreg<WS, -1>(t);         // over kx -> spatial x at t[fft_pos(xs)]
            __syncthreads();
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int xs = decltype(kc)::value;
                my_tile[c * PITCH + xs] = t[FFT_POS<xs, WS>];     // tile[ky = c][x]
            });
        }
        __syncthreads();
        // lane owns spatial column x = c: read over ky, inverse FFT over ky
#pragma unroll
        for (int ky = 0; ky < WS; ++ky) x[ky] = my_tile[ky * PITCH + c];
        fft_inreg<WS, -1>(x);             // corr(y, x = c) real part at x[fft_pos(y)].x
        __syncthreads();                  // everyone is done reading the tile; it becomes the map

        // ---- correlation map in fftshift coordinates: y' = (y + WS/2) % WS, x' = (c + WS/2) % WS
        const int xs = (c + WS / 2) % WS;
        float cmin = 3.4e38f;
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int y = decltype(kc)::value;
            cmin = fminf(cmin, x[FFT_POS<y, WS>].x);
        });
        cmin = group_min<WS>(cmin, scratch);
        ArgMax best{-1.f, 0};
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int ysft = decltype(kc)::value;           // shifted row, ascending
            constexpr int y = (ysft + WS / 2) % WS;
            // B:518 corr - min ; B:381 corr += eps (float32 in passes >= 2)
            float v = __fadd_rn(__fsub_rn(x[FFT_POS<y, WS>].x, cmin), 1e-7f);
            x[FFT_POS<y, WS>].x = v;
            my_map[ysft * WS + xs] = v;
            if (v > best.v) {
                best.v = v;
                best.idx = ysft * WS + xs;
            }
        });
        best = group_argmax<WS>(best, scratch);
        __syncthreads();                  // map complete

        if (p.dbg_corr != nullptr && active) {
            float* d = p.dbg_corr + fidx * WS * WS;
#pragma unroll
            for (int y = 0; y < WS; ++y) d[y * WS + c] = my_map[y * WS + c];
        }

        // ---- second peak: arg-max outside the 7x7 flat-index neighbourhood (B:346-358)
        const int m = best.idx;
        const int KD = WS * WS;
        const int wv = p.val_win;
        const int my_ = m / WS, mx_ = m % WS;
        ArgMax second{-1.f, KD};
        {
            // decompose q - m = i + WS*j with |i| <= wv: column part is lane-constant
            int a_ = xs - mx_ + wv;       // i + wv  (before borrow)
            int jadj = 0;
            if (a_ < 0) {
                a_ += WS;
                jadj = -1;
            } else if (a_ >= WS) {
                a_ -= WS;
                jadj = 1;
            }
            const bool col_in = a_ <= 2 * wv;
            const bool zero_hit = (m - wv - wv * WS) <= 0;             // clamp sends some index to 0
            const bool last_hit = (m + wv + wv * WS) >= KD - 1;        // ... or to KD-1
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int ysft = decltype(kc)::value;
                constexpr int y = (ysft + WS / 2) % WS;
                const int q = ysft * WS + xs;
                const int jj = ysft - my_ + wv + jadj;
                bool excl = col_in && (jj >= 0) && (jj <= 2 * wv);
                excl = excl || (q == 0 && zero_hit) || (q == KD - 1 && last_hit);
                const float v = x[FFT_POS<y, WS>].x;
                if (!excl && v > second.v) {
                    second.v = v;
                    second.idx = q;
                }
            });
        }
        second = group_argmax<WS>(second, scratch);

        // ---- sub-pixel fit (B:385-407): lanes 0..4 of the window take one logarithm each
        {
            int left = m + 1, right = m - 1, top = m + WS, bot = m - WS;
            if (left >= KD - 1) left = m;
            if (right <= 0) right = m;
            if (top >= KD - 1) top = m;
            if (bot <= 0) bot = m;
            int q = m;
            q = (c == 1) ? left : q;
            q = (c == 2) ? right : q;
            q = (c == 3) ? top : q;
            q = (c == 4) ? bot : q;
            q = (c == 5) ? (second.idx < KD ? second.idx : m) : q;
            double val = (double)my_map[q];
            double lg = log(val);
            // gather the five logs and the two raw values to lane 0 of the window
            const int base = (tid & 63) - (c & 63);     // first lane of this window in its wavefront
            const double lm = __shfl(lg, base + 0, 64);
            const double ll = __shfl(lg, base + 1, 64);
            const double lr = __shfl(lg, base + 2, 64);
            const double lt = __shfl(lg, base + 3, 64);
            const double lb = __shfl(lg, base + 4, 64);
            const double cm = __shfl(val, base + 0, 64);
            const double c2 = __shfl(val, base + 5, 64);
            if (c == 0 && active) {
                const double nom1 = lr - ll;
                const double den1 = 2 * (ll + lr) - 4 * lm;
                const double nom2 = lb - lt;
                const double den2 = 2 * (lb + lt) - 4 * lm;
                double du = (double)mx_ + nom1 / den1 - (double)(WS / 2);
                double dv = (double)my_ + nom2 / den2 - (double)(WS / 2);
                du = nan_to_num(du);
                dv = nan_to_num(dv);
                bool invalid = (cm / c2) < p.val_ratio;                  // B:411
                if constexpr (MODE == MODE_PASS1) {
                    if (dead) {             // all-NaN map in the reference: u = v = 0, "valid"
                        du = 0.0;
                        dv = 0.0;
                        invalid = false;
                    }
                    p.u[fidx] = du;
                    p.v[fidx] = dv;
                    p.val[fidx] = invalid ? 1 : 0;
                } else {
                    // multipass combine (B:728-738 / B:800-810)
                    const double u0 = p.u0[fidx], v0 = p.v0[fidx];
                    const double u2 = p.u2[fidx], v2 = p.v2[fidx];
                    double u = 2 * u2 + du;
                    double v = 2 * v2 + dv;
                    const bool mask_u = ((du > u0) && (rint(u0) > 0)) || invalid;
                    const bool mask_v = ((dv > v0) && (rint(v0) > 0)) || invalid;
                    if (mask_u) u = u0;
                    if (mask_v) v = v0;
                    p.u[fidx] = u;
                    p.v[fidx] = v;
                    p.val[fidx] = invalid ? 1 : 0;
                    if (p.du != nullptr) {
                        p.du[fidx] = du;
                        p.dv[fidx] = dv;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <int WS, int MODE>
static hipError_t launch_one(const PassParams& p, int n_cu, hipStream_t stream) {
    using G = Geo<WS>;
    const int N = p.n_rows * p.n_cols;
    const long long groups = (N + G::WPB - 1) / G::WPB;
    const long long items = (long long)p.batch * groups;
    long long blocks = items;
    const long long cap = (long long)n_cu * 16;     // grid-stride above this
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;                  // the XCD remap needs a multiple of 8
    hipLaunchKernelGGL((xcorr_kernel<WS, MODE>), dim3((unsigned)blocks), dim3(G::TPB), 0, stream, p);
    return hipGetLastError();
}


// one translation unit per tile size instantiates this (xcorr_ws*.hip)
template <int WS>
hipError_t launch_xcorr_ws(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    switch (mode) {
        case MODE_PASS1: return launch_one<WS, MODE_PASS1>(p, n_cu, stream);
        case MODE_DWS: return launch_one<WS, MODE_DWS>(p, n_cu, stream);
        case MODE_CWS: return launch_one<WS, MODE_CWS>(p, n_cu, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace tpiv
