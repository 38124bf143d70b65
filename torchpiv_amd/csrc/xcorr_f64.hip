// First pass at the reference's precision (TPIV_PREC_REFERENCE): float64 windows, transforms, map
// and peak analysis for the power-of-two tile sizes 8..64.
//
// The reference promotes pass 1 to float64 (PIVbackend.py:513-514: aa = a / mean(a) as float64, then
// correalte_fft B:249-257 runs rfft2 / irfft2 in complex128, `corr - corr.min()` B:518 and
// correlation_to_displacement B:360-422 on the float64 map).  The float32 tile kernels
// (xcorr_tile.hpp) hold a whole line per lane in registers; a 64-point complex float64 line would
// need 256 VGPRs, so this kernel keeps the packed tile Z = a/mean(a) + i b/mean(b) in LDS as two
// planes of doubles (WS = 64: 2 x 33 KB, two workgroups per CU) and transforms it IN PLACE:
//
//   forward  decimation in frequency, radix 4 (radix 2 first when log2 WS is odd): natural order in,
//            digit-reversed order out -- along x, then along y;
//   spectrum the pair {k, -k} is found through the digit-reversal table and handled by ONE thread:
//            P(k) = conj(A) B from Z(k), Z(-k) (packed real transforms), P(-k) = conj P(k);
//   inverse  the exact inverse of the forward stages in reverse order (decimation in time, conjugate
//            twiddles): digit-reversed in, natural order out -- along y, then along x.
//
// No reordering pass and no second buffer; every stage reads and writes the same four (two) cells per
// butterfly, so only a workgroup barrier separates the stages.  Peak analysis follows the reference on
// the float64 map (first flat index on ties, flat-index neighbours and fix-ups, 7x7 flat-index
// exclusion with row wrap and clamps); the 8-double record goes to finalize_kernel<true>.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "piv_kernels.h"

namespace tpiv {

namespace {

struct cd {
    double x, y;
};
__device__ __forceinline__ cd cmul(cd a, cd w) { return cd{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ cd cmulc(cd a, cd w) { return cd{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }   // a * conj(w)
__device__ __forceinline__ cd cadd_(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub_(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }

constexpr bool radix2_first_d(int n) { return n == 2 || n == 8 || n == 32 || n == 128; }
// position of bin k after the in-place DIF transform of length n (same rule as fft_pos, fft_inreg.hpp)
constexpr int pos_of(int k, int n) {
    if (n <= 1) return 0;
    if (radix2_first_d(n)) return (k % 2) * (n / 2) + pos_of(k / 2, n / 2);
    return (k % 4) * (n / 4) + pos_of(k / 4, n / 4);
}

template <int WS>
struct F64Geo {
    static constexpr int NT = WS >= 32 ? 256 : 64;      // threads per workgroup (one window)
    static constexpr int P = WS + 1;                    // plane pitch in doubles
    static constexpr int NW = NT / 64;                  // wavefronts
};

template <int WS>
struct F64Shared {
    double re[WS * (WS + 1)];
    double im[WS * (WS + 1)];
    double twc[WS], tws[WS];          // exp(-2 pi i k / WS) = twc - i * (-tws) ... stored as (cos, -sin)
    int pos[WS];                      // bin -> position
    int bin[WS];                      // position -> bin
    double redd[8];
    int redi[8];
    unsigned redu[8];
};

// ---- workgroup reductions (every thread gets the result) ------------------------------------------
template <int NW, typename T, typename OP>
__device__ __forceinline__ T wg_reduce(T v, OP op, T* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
    if constexpr (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = op(r, red[w]);
    return r;
}

// ---- one radix-R stage on sub-length L, along x (ALONG_Y = false) or y, forward (DIF) or inverse (DIT)
template <int WS, int L, int R, bool ALONG_Y, bool FWD>
__device__ __forceinline__ void stage(F64Shared<WS>& sm) {
    using G = F64Geo<WS>;
    constexpr int P = G::P;
    constexpr int Q = L / R;                   // butterflies per block
    constexpr int TWS = WS / L;                // twiddle stride in the length-WS table
    constexpr int NB = WS * (WS / R);          // butterflies of the whole tile
    for (int b = threadIdx.x; b < NB; b += G::NT) {
        int line, t;
        if constexpr (ALONG_Y) {
            line = b % WS;                     // lanes walk along a row of the planes
            t = b / WS;
        } else {
            line = b / (WS / R);
            t = b % (WS / R);
        }
        const int blk = t / Q, j = t % Q;
        const int e0 = blk * L + j;
        auto addr = [&](int q) { return ALONG_Y ? (e0 + q * Q) * P + line : line * P + e0 + q * Q; };
        auto tw = [&](int q) { const int i = (q * j * TWS) % WS; return cd{sm.twc[i], sm.tws[i]}; };
        if constexpr (R == 2) {
            const int a0 = addr(0), a1 = addr(1);
            cd a{sm.re[a0], sm.im[a0]}, c{sm.re[a1], sm.im[a1]};
            cd y0, y1;
            if constexpr (FWD) {
                y0 = cadd_(a, c);
                y1 = cmul(csub_(a, c), tw(1));
            } else {
                const cd v1 = cmulc(c, tw(1));
                y0 = cadd_(a, v1);
                y1 = csub_(a, v1);
            }
            sm.re[a0] = y0.x, sm.im[a0] = y0.y;
            sm.re[a1] = y1.x, sm.im[a1] = y1.y;
        } else {
            const int a0 = addr(0), a1 = addr(1), a2 = addr(2), a3 = addr(3);
            cd x0{sm.re[a0], sm.im[a0]}, x1{sm.re[a1], sm.im[a1]}, x2{sm.re[a2], sm.im[a2]}, x3{sm.re[a3], sm.im[a3]};
            cd y0, y1, y2, y3;
            if constexpr (FWD) {
                const cd t0 = cadd_(x0, x2), t1 = csub_(x0, x2), t2 = cadd_(x1, x3), bd = csub_(x1, x3);
                const cd t3{bd.y, -bd.x};                      // -i (x1 - x3)
                y0 = cadd_(t0, t2);
                y1 = cmul(cadd_(t1, t3), tw(1));
                y2 = cmul(csub_(t0, t2), tw(2));
                y3 = cmul(csub_(t1, t3), tw(3));
            } else {
                const cd v1 = cmulc(x1, tw(1)), v2 = cmulc(x2, tw(2)), v3 = cmulc(x3, tw(3));
                const cd s0 = cadd_(x0, v2), s1 = csub_(x0, v2), s2 = cadd_(v1, v3), d = csub_(v1, v3);
                const cd s3{-d.y, d.x};                        // +i (v1 - v3)
                y0 = cadd_(s0, s2);
                y1 = cadd_(s1, s3);
                y2 = csub_(s0, s2);
                y3 = csub_(s1, s3);
            }
            sm.re[a0] = y0.x, sm.im[a0] = y0.y;
            sm.re[a1] = y1.x, sm.im[a1] = y1.y;
            sm.re[a2] = y2.x, sm.im[a2] = y2.y;
            sm.re[a3] = y3.x, sm.im[a3] = y3.y;
        }
    }
    __syncthreads();
}

// all stages of one dimension: forward = largest block first, inverse = the same stages in reverse order
template <int WS, int L, bool ALONG_Y, bool FWD>
__device__ __forceinline__ void transform(F64Shared<WS>& sm) {
    if constexpr (L >= 2) {
        constexpr int R = radix2_first_d(L) ? 2 : 4;
        if constexpr (FWD) {
            stage<WS, L, R, ALONG_Y, true>(sm);
            transform<WS, L / R, ALONG_Y, true>(sm);
        } else {
            transform<WS, L / R, ALONG_Y, false>(sm);
            stage<WS, L, R, ALONG_Y, false>(sm);
        }
    }
}

template <int WS>
__global__ __launch_bounds__(F64Geo<WS>::NT) void xcorr_f64_kernel(PassParams p) {
    using G = F64Geo<WS>;
    constexpr int NT = G::NT, P = G::P, NN = WS * WS;
    constexpr int NDW = NN / 4;                 // dwords per frame window
    constexpr int DPT = (NDW + NT - 1) / NT;    // dwords per thread
    __shared__ F64Shared<WS> sm;
    const int tid = threadIdx.x;

    for (int k = tid; k < WS; k += NT) {
        double s, c;
        sincospi(2.0 * (double)k / (double)WS, &s, &c);
        sm.twc[k] = c;
        sm.tws[k] = -s;
        const int q = pos_of(k, WS);            // (evaluated at run time: small recursion, once per workgroup)
        sm.pos[k] = q;
        sm.bin[q] = k;
    }
    __syncthreads();

    const int N = p.n_rows * p.n_cols;
    const long long items = (long long)p.batch * N;
    const int st = p.ws - p.ov;
    const int HW = p.H * p.W;
    // XCD-aware static order: workgroups b, b+8, ... share an XCD and walk one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < items ? lo + chunk : items;

    for (long long item = lo + slot; item < hi; item += per_xcd) {
        const int pair = (int)(item / N), win = (int)(item % N);
        const int y0 = (win / p.n_cols) * st, x0 = (win % p.n_cols) * st;
        const uint8_t* __restrict__ fa = p.A + (size_t)pair * HW + (size_t)y0 * p.W + x0;
        const uint8_t* __restrict__ fb = p.B + (size_t)pair * HW + (size_t)y0 * p.W + x0;
        const size_t fidx = (size_t)item;

        // ---- stage 0: pixels (4 per dword load, any alignment), exact integer window sums
        uint32_t da[DPT], db[DPT];
        unsigned ia = 0, ib = 0;
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int i = tid + q * NT;
            da[q] = 0;
            db[q] = 0;
            if (i < NDW) {
                const int y = i / (WS / 4), x4 = (i % (WS / 4)) * 4;
                __builtin_memcpy(&da[q], fa + (size_t)y * p.W + x4, 4);
                __builtin_memcpy(&db[q], fb + (size_t)y * p.W + x4, 4);
                ia = __builtin_amdgcn_sad_u8(da[q], 0u, ia);
                ib = __builtin_amdgcn_sad_u8(db[q], 0u, ib);
            }
        }
        auto uadd = [](unsigned a, unsigned b) { return a + b; };
        ia = wg_reduce<G::NW>(ia, uadd, sm.redu);
        ib = wg_reduce<G::NW>(ib, uadd, sm.redu + 4);
        const bool dead = ia == 0u || ib == 0u;          // zero-mean window: 0/0 = NaN map in the reference
        const double ma = (double)ia / (double)NN, mb = (double)ib / (double)NN;      // torch.mean: exact sum / n
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int i = tid + q * NT;
            if (i < NDW) {
                const int y = i / (WS / 4), x4 = (i % (WS / 4)) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double a = (double)((da[q] >> (8 * k)) & 0xffu), b = (double)((db[q] >> (8 * k)) & 0xffu);
                    sm.re[y * P + x4 + k] = dead ? 0.0 : a / ma;       // B:513-514
                    sm.im[y * P + x4 + k] = dead ? 0.0 : b / mb;
                }
            }
        }
        __syncthreads();

        // ---- forward 2-D transform of a + i b, in place: bin (ky, kx) ends at [pos(ky)][pos(kx)]
        transform<WS, WS, false, true>(sm);
        transform<WS, WS, true, true>(sm);

        // ---- cross-spectrum P = conj(A) B / n^2 of the packed transform; one thread per pair {k, -k}
        {
            constexpr double SC = 0.25 / (double)NN;       // 1/4 of the un-packing, 1/n^2 of the inverse (exact)
            for (int e = tid; e < NN; e += NT) {
                const int py = e / WS, px = e % WS;
                const int ky = sm.bin[py], kx = sm.bin[px];
                const int qy = sm.pos[(WS - ky) % WS], qx = sm.pos[(WS - kx) % WS];
                const int e2 = qy * WS + qx;
                if (e2 < e) continue;                      // the partner's thread writes both cells
                const int a1 = py * P + px, a2 = qy * P + qx;
                const double a_ = sm.re[a1], b_ = sm.im[a1], c_ = sm.re[a2], d_ = sm.im[a2];
                // Z(k) = a + ib, Z(-k) = c + id:  4 P(k) = 2 (a d + b c) + i ((c^2 - a^2) + (d^2 - b^2))
                const double pr = (a_ * d_ + b_ * c_) * (2.0 * SC);
                const double pi = ((c_ * c_ - a_ * a_) + (d_ * d_ - b_ * b_)) * SC;
                sm.re[a1] = pr;
                sm.im[a1] = pi;
                if (e2 != e) {                             // P(-k) = conj P(k)
                    sm.re[a2] = pr;
                    sm.im[a2] = -pi;
                }
            }
            __syncthreads();
        }

        // ---- inverse 2-D transform: natural order out; the map is the real plane
        transform<WS, WS, true, false>(sm);
        transform<WS, WS, false, false>(sm);

        // ---- peak analysis in fftshift coordinates (y' = (y + WS/2) % WS, x' likewise), float64
        auto dmin = [](double a, double b) { return a < b ? a : b; };
        double cmin = 1.7e308;
        for (int e = tid; e < NN; e += NT) cmin = dmin(cmin, sm.re[(e / WS) * P + e % WS]);
        cmin = wg_reduce<G::NW>(cmin, dmin, sm.redd);
        double bv = -1.0;
        int bf = NN;
        for (int e = tid; e < NN; e += NT) {
            const int y = e / WS, x = e % WS;
            const double v = __dadd_rn(__dsub_rn(sm.re[y * P + x], cmin), 1e-7);      // B:518, B:381
            sm.re[y * P + x] = v;
            const int f = ((y + WS / 2) % WS) * WS + (x + WS / 2) % WS;
            if (v > bv || (v == bv && f < bf)) {
                bv = v;
                bf = f;
            }
        }
        const double gmax = wg_reduce<G::NW>(bv, [](double a, double b) { return a > b ? a : b; }, sm.redd + 4);
        auto imin = [](int a, int b) { return a < b ? a : b; };
        const int m = wg_reduce<G::NW>(bv == gmax ? bf : NN, imin, sm.redi);           // first flat index (B:383)
        // (the two reductions above also order the map writes before the reads below)
        const int wv = p.val_win;
        double sv = -1.0;
        for (int e = tid; e < NN; e += NT) {
            const int y = e / WS, x = e % WS;
            const int f = ((y + WS / 2) % WS) * WS + (x + WS / 2) % WS;
            bool excl = false;                             // f in {clamp(m + i + WS j)}: B:352-357
            for (int j = -wv; j <= wv; ++j) {
                const int t = f - m - WS * j;
                if (t >= -wv && t <= wv) excl = true;
            }
            if (f == 0 && (m - wv - wv * WS) <= 0) excl = true;
            if (f == NN - 1 && (m + wv + wv * WS) >= NN - 1) excl = true;
            const double v = sm.re[y * P + x];
            if (!excl && v > sv) sv = v;
        }
        sv = wg_reduce<G::NW>(sv, [](double a, double b) { return a > b ? a : b; }, sm.redd);
        if (tid < 8) {
            int left = m + 1, right = m - 1, top = m + WS, bot = m - WS;      // B:385-392 (flat index)
            if (left >= NN - 1) left = m;
            if (right <= 0) right = m;
            if (top >= NN - 1) top = m;
            if (bot <= 0) bot = m;
            int q = m;
            q = (tid == 1) ? left : q;
            q = (tid == 2) ? right : q;
            q = (tid == 3) ? top : q;
            q = (tid == 4) ? bot : q;
            const int ys = q / WS, xs = q % WS;                               // shifted -> stored coordinates
            double outv = sm.re[((ys + WS / 2) % WS) * P + (xs + WS / 2) % WS];
            // nothing left outside the exclusion zone: the reference's second arg-max runs over the zeroed
            // map, whose float64 storage `cor` aliases in pass 1 (B:382): c[m2] = 0, ratio = +inf
            outv = (tid == 5) ? (sv >= 0.0 ? sv : 0.0) : outv;
            outv = (tid == 6) ? (double)m : outv;
            outv = (tid == 7) ? (dead ? 1.0 : 0.0) : outv;
            reinterpret_cast<double*>(p.peak_raw)[fidx * 8 + tid] = outv;
        }
        __syncthreads();                                   // planes free for the next window
    }
}

template <int WS>
hipError_t launch_f64(const PassParams& p, int n_cu, hipStream_t stream) {
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    if (items <= 0) return hipErrorInvalidValue;
    const int per_cu = WS == 64 ? 2 : (WS == 32 ? 4 : 8);       // LDS: 66.6 KB / 17 KB / ... per workgroup
    long long blocks = items < (long long)n_cu * per_cu ? items : (long long)n_cu * per_cu;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((xcorr_f64_kernel<WS>), dim3((unsigned)blocks), dim3(F64Geo<WS>::NT), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_xcorr_f64(const PassParams& p, int n_cu, hipStream_t stream) {
    switch (p.ws) {
        case 8: return launch_f64<8>(p, n_cu, stream);
        case 16: return launch_f64<16>(p, n_cu, stream);
        case 32: return launch_f64<32>(p, n_cu, stream);
        case 64: return launch_f64<64>(p, n_cu, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace tpiv
