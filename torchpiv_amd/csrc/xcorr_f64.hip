// First pass at the reference's precision (TPIV_PREC_REFERENCE): float64 windows, transforms, map
// and peak analysis for the power-of-two tile sizes 8..64.
//
// The reference promotes pass 1 to float64 (PIVbackend.py:513-514: aa = a / mean(a) as float64, then
// correalte_fft B:249-257 runs rfft2 / irfft2 in complex128, `corr - corr.min()` B:518 and
// correlation_to_displacement B:360-422 on the float64 map).  The float32 tile kernels
// (xcorr_tile.hpp) hold a whole line per lane in registers; a 64-point complex float64 line would
// need 256 VGPRs, so this kernel keeps the packed tile Z = a/mean(a) + i b/mean(b) in LDS (16-byte
// complex elements, WS = 64: 66.6 KB, two workgroups per CU) and transforms it IN PLACE:
//
//   forward  decimation in frequency, radix 8 (then 4 or 2 for what is left of the length): natural
//            order in, digit-reversed order out -- along x, then along y; two LDS passes per
//            dimension for WS = 64, and only the first of them carries non-trivial twiddles;
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
#include "xcorr_tile.hpp"      // grp_reduce: wavefront reductions in the VALU (DPP / permlane swaps)
#include "xcorr_f64_split.hpp" // 64x64: lines split over two lanes, 32-point in-register codelets (second generation)

namespace tpiv {

namespace {

// (the LDS-resident scheme below -- 8x8 ... 32x32 windows -- has its own 16-byte complex type)
struct alignas(16) cz {       // 16-byte alignment: ds_read_b128 / ds_write_b128 instead of ds_read2_b64 / ds_write2_b64
    double x, y;
};
__device__ __forceinline__ cz cmul(cz a, cz w) { return cz{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ cz cmulc(cz a, cz w) { return cz{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }   // a * conj(w)
__device__ __forceinline__ cz cadd_(cz a, cz b) { return cz{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cz csub_(cz a, cz b) { return cz{a.x - b.x, a.y - b.y}; }
// a * (-i) for the forward transform, a * (+i) for the inverse
template <bool FWD>
__device__ __forceinline__ cz rot90(cz a) { return FWD ? cz{a.y, -a.x} : cz{-a.y, a.x}; }

// radix of the stage that works on blocks of length L: 8 while it fits, then what is left (4 or 2)
constexpr int radix_of(int L) { return L >= 8 ? 8 : L; }
// position of bin k after the in-place DIF transform of length n
constexpr int pos_of(int k, int n) {
    if (n <= 1) return 0;
    const int r = radix_of(n);
    return (k % r) * (n / r) + pos_of(k / r, n / r);
}

// the closed form the cross-spectrum uses: two stages (radix R1 = min(8, n), then n / R1) make the digit
// reversal a swap of two digits
constexpr bool digit_swap_ok(int n) {
    const int r1 = n >= 8 ? 8 : n, r2 = n / r1;
    for (int k = 0; k < n; ++k)
        if (pos_of(k, n) != (k % r1) * r2 + k / r1) return false;
    return true;
}
static_assert(digit_swap_ok(8) && digit_swap_ok(16) && digit_swap_ok(32) && digit_swap_ok(64), "digit reversal closed form");

template <int WS>
struct F64Geo {
    // threads per workgroup (one window).  64x64: 512, i.e. 16 wavefronts per CU with the two workgroups the
    // LDS admits -- at 256 threads the two wavefronts per SIMD could not hide the LDS round trips
    static constexpr int NT = WS >= 64 ? 512 : (WS >= 32 ? 256 : 64);
    static constexpr int P = WS + 1;                    // tile pitch in complex elements
    static constexpr int NW = NT / 64;                  // wavefronts
    static constexpr int OCC = WS >= 64 ? 4 : 1;        // wavefronts per SIMD the register budget must allow
};

template <int WS>
struct F64Shared {
    cz z[WS * (WS + 1)];              // the packed tile a/mean(a) + i b/mean(b): 16-byte elements (ds_*_b128)
    cz tw[WS];                        // exp(-2 pi i k / WS)
    double redd[16];
    int redi[8];
    unsigned long long redu[8];
};

// Workgroup barrier for LDS exchanges only: __syncthreads() also waits for vmcnt(0), which would drain
// the next window's pixel prefetch at the first of the ~25 barriers per window.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- workgroup reductions (every thread gets the result) ------------------------------------------
// (the wavefront step runs in the VALU -- DPP row operations and permlane swaps, xcorr_tile.hpp -- not
//  through ds_bpermute: six dependent LDS round trips per reduction were a sixth of the window time)
template <int NW, typename T, typename OP>
__device__ __forceinline__ T wg_reduce(T v, OP op, T* red) {
    v = grp_reduce<64>(v, op);
    if constexpr (NW == 1) return v;
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    T r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = op(r, red[w]);
    return r;
}

// R-point DFT of x[0..R) in registers, natural order in and out; FWD: exp(-2 pi i pq/R), else the conjugate kernel
template <int R, bool FWD>
__device__ __forceinline__ void dft_small(cz (&x)[R]) {
    if constexpr (R == 2) {
        const cz a = x[0], b = x[1];
        x[0] = cadd_(a, b);
        x[1] = csub_(a, b);
    } else if constexpr (R == 4) {
        const cz t0 = cadd_(x[0], x[2]), t1 = csub_(x[0], x[2]), t2 = cadd_(x[1], x[3]);
        const cz t3 = rot90<FWD>(csub_(x[1], x[3]));
        x[0] = cadd_(t0, t2);
        x[1] = cadd_(t1, t3);
        x[2] = csub_(t0, t2);
        x[3] = csub_(t1, t3);
    } else {
        static_assert(R == 8, "radix 2, 4 or 8");
        constexpr double H = 0.70710678118654752440;
        // first layer: a_p = x_p + x_{p+4} (even outputs), b_p = (x_p - x_{p+4}) w8^p (odd outputs)
        cz a[4], b[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            a[p] = cadd_(x[p], x[p + 4]);
            b[p] = csub_(x[p], x[p + 4]);
        }
        // w8^1 = (1 -+ i)/sqrt2, w8^2 = -+i, w8^3 = (-1 -+ i)/sqrt2   (upper sign: forward)
        b[1] = FWD ? cz{(b[1].x + b[1].y) * H, (b[1].y - b[1].x) * H} : cz{(b[1].x - b[1].y) * H, (b[1].y + b[1].x) * H};
        b[2] = rot90<FWD>(b[2]);
        b[3] = FWD ? cz{(b[3].y - b[3].x) * H, -(b[3].x + b[3].y) * H} : cz{-(b[3].x + b[3].y) * H, (b[3].x - b[3].y) * H};
        dft_small<4, FWD>(a);
        dft_small<4, FWD>(b);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            x[2 * p] = a[p];
            x[2 * p + 1] = b[p];
        }
    }
}

// ---- one radix-R stage on sub-length L of every line, along x (ALONG_Y = false) or y, forward
//      (decimation in frequency: butterfly, then twiddle) or inverse (conjugate twiddle, then butterfly).
// Thread mapping: a line has WS / R butterflies (index t = block * (L/R) + j).  The twiddles depend on j
// only, so a thread keeps ONE t and walks the lines: it fetches its R - 1 twiddles once per stage.
// The LW lanes that share a t are consecutive LINES.  Along y that is LW consecutive 16-byte elements of
// a tile row; along x it is LW consecutive rows of one column, 65 elements = 260 dwords = 4 banks
// (mod 64) apart -- both patterns are conflict-free for ds_read/write_b128.  (Mapping the lanes of a
// wavefront to the butterflies of one row instead cost 3 bank-conflict cycles per LDS cycle:
// SQ_LDS_BANK_CONFLICT 2.7e10 against SQ_ACTIVE_INST_LDS 8.4e9 per launch.)
template <int WS, int L, int R, bool ALONG_Y, bool FWD>
__device__ __forceinline__ void stage(F64Shared<WS>& sm) {
    using G = F64Geo<WS>;
    constexpr int P = G::P, NT = G::NT;
    constexpr int Q = L / R;                   // butterflies per block = stride between a butterfly's elements
    constexpr int BPL = WS / R;                // butterflies per line
    constexpr int TWS = WS / L;                // twiddle stride in the length-WS table
    constexpr int LW = WS < NT * 8 / WS ? WS : NT * 8 / WS;        // lanes that share a butterfly index
    constexpr int TPP = NT / LW;                                   // butterfly indices handled per pass
    const int tid = threadIdx.x;
    const int t0 = tid / LW;
    const int line0 = tid % LW;
    constexpr int LSTEP = LW;                                      // lines advanced per pass
    for (int t = t0; t < BPL; t += TPP) {                          // (one iteration unless NT / LW < WS / R)
        const int blk = t / Q, j = t % Q;
        const int e0 = blk * L + j;
        cz w[R];
        if constexpr (Q > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) w[q] = sm.tw[(q * j * TWS) % WS];
        }
        constexpr int NIT = (WS + LSTEP - 1) / LSTEP;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int line = line0 + it * LSTEP;
            if (LSTEP * NIT != WS && line >= WS) break;
            if (WS < LSTEP && line >= WS) break;
            cz x[R];
#pragma unroll
            for (int q = 0; q < R; ++q) x[q] = ALONG_Y ? sm.z[(e0 + q * Q) * P + line] : sm.z[line * P + e0 + q * Q];
            if constexpr (FWD) {
                dft_small<R, true>(x);
                if constexpr (Q > 1) {
#pragma unroll
                    for (int q = 1; q < R; ++q) x[q] = cmul(x[q], w[q]);
                }
            } else {
                if constexpr (Q > 1) {
#pragma unroll
                    for (int q = 1; q < R; ++q) x[q] = cmulc(x[q], w[q]);
                }
                dft_small<R, false>(x);
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if constexpr (ALONG_Y) sm.z[(e0 + q * Q) * P + line] = x[q];
                else sm.z[line * P + e0 + q * Q] = x[q];
            }
        }
    }
    lds_barrier();
}

// all stages of one dimension: forward = largest block first, inverse = the same stages in reverse order
template <int WS, int L, bool ALONG_Y, bool FWD>
__device__ __forceinline__ void transform(F64Shared<WS>& sm) {
    if constexpr (L >= 2) {
        constexpr int R = radix_of(L);
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
__global__ __launch_bounds__(F64Geo<WS>::NT, F64Geo<WS>::OCC) void xcorr_f64_kernel(PassParams p) {
    using G = F64Geo<WS>;
    constexpr int NT = G::NT, P = G::P, NN = WS * WS;
    constexpr int NDW = NN / 4;                 // dwords per frame window
    constexpr int DPT = (NDW + NT - 1) / NT;    // dwords per thread
    __shared__ F64Shared<WS> sm;
    const int tid = threadIdx.x;

    for (int k = tid; k < WS; k += NT) {
        double s, c;
        sincospi(2.0 * (double)k / (double)WS, &s, &c);
        sm.tw[k] = cz{c, -s};
    }
    lds_barrier();

    const int N = p.n_rows * p.n_cols;
    const long long items = (long long)p.batch * N;
    const int st = p.ws - p.ov;
    const int HW = p.H * p.W;
    // XCD-aware static order: workgroups b, b+8, ... share an XCD and walk one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < items ? lo + chunk : items;

    // the pixels of the next window are fetched while the current one is transformed (4 dwords per thread)
    uint32_t da[DPT], db[DPT];
    auto fetch = [&](long long it) {
        const int pair_ = (int)(it / N), win_ = (int)(it % N);
        const int yy0 = (win_ / p.n_cols) * st, xx0 = (win_ % p.n_cols) * st;
        const uint8_t* __restrict__ fa = p.A + (size_t)pair_ * HW + (size_t)yy0 * p.W + xx0;
        const uint8_t* __restrict__ fb = p.B + (size_t)pair_ * HW + (size_t)yy0 * p.W + xx0;
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int i = tid + q * NT;
            da[q] = 0;
            db[q] = 0;
            if (i < NDW) {
                const int y = i / (WS / 4), x4 = (i % (WS / 4)) * 4;
                __builtin_memcpy(&da[q], fa + (size_t)y * p.W + x4, 4);      // (unaligned dword loads are fine in global memory)
                __builtin_memcpy(&db[q], fb + (size_t)y * p.W + x4, 4);
            }
        }
    };
    if (lo + slot < hi) fetch(lo + slot);
    for (long long item = lo + slot; item < hi; item += per_xcd) {
        const size_t fidx = (size_t)item;

        // ---- stage 0: exact integer window sums of the fetched pixels
        unsigned ia = 0, ib = 0;
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            ia = __builtin_amdgcn_sad_u8(da[q], 0u, ia);
            ib = __builtin_amdgcn_sad_u8(db[q], 0u, ib);
        }
        {      // both window sums in one reduction (each is below 2^21)
            auto uadd = [](unsigned long long a, unsigned long long b) { return a + b; };
            const unsigned long long s2 = wg_reduce<G::NW>((unsigned long long)ia | ((unsigned long long)ib << 32), uadd, sm.redu);
            ia = (unsigned)s2;
            ib = (unsigned)(s2 >> 32);
        }
        const bool dead = ia == 0u || ib == 0u;          // zero-mean window: 0/0 = NaN map in the reference
        const double ma = (double)ia / (double)NN, mb = (double)ib / (double)NN;      // torch.mean: exact sum / n
        // a / mean(a) (B:513-514) as one reciprocal per window plus a residual correction per pixel:
        // q = a r, q += (a - q m) r  -- the correctly rounded quotient (the residual is exact in an fma)
        // at a fifth of the instructions of 32 float64 divisions per thread
        const double ra = dead ? 0.0 : 1.0 / ma, rb = dead ? 0.0 : 1.0 / mb;
        auto quot = [](double a, double m, double r) {
            const double q = a * r;
            return __fma_rn(__fma_rn(-q, m, a), r, q);
        };
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            const int i = tid + q * NT;
            if (i < NDW) {
                const int y = i / (WS / 4), x4 = (i % (WS / 4)) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double a = (double)((da[q] >> (8 * k)) & 0xffu), b = (double)((db[q] >> (8 * k)) & 0xffu);
                    sm.z[y * P + x4 + k] = cz{quot(a, ma, ra), quot(b, mb, rb)};
                }
            }
        }
        lds_barrier();
        if (item + per_xcd < hi) fetch(item + per_xcd);      // in flight during the transforms

        // ---- forward 2-D transform of a + i b, in place: bin (ky, kx) ends at [pos(ky)][pos(kx)]
        transform<WS, WS, false, true>(sm);
        transform<WS, WS, true, true>(sm);

        // ---- cross-spectrum P = conj(A) B / n^2 of the packed transform; one thread per pair {k, -k}
        {
            constexpr double SC = 0.25 / (double)NN;       // 1/4 of the un-packing, 1/n^2 of the inverse (exact)
            // digit reversal in closed form (two stages: radix R1 = 8, then WS / 8): bin <-> position are digit
            // swaps, so the partner cell needs no table look-up (four dependent LDS reads per element before)
            constexpr int R1 = WS >= 8 ? 8 : WS, R2 = WS / R1;
            auto bin_of = [](int p_) { return (p_ % R2) * R1 + p_ / R2; };
            auto pos_of_ = [](int k_) { return (k_ % R1) * R2 + k_ / R1; };
            for (int e = tid; e < NN; e += NT) {
                const int py = e / WS, px = e % WS;
                const int ky = bin_of(py), kx = bin_of(px);
                const int qy = pos_of_((WS - ky) % WS), qx = pos_of_((WS - kx) % WS);
                const int e2 = qy * WS + qx;
                if (e2 < e) continue;                      // the partner's thread writes both cells
                const int a1 = py * P + px, a2 = qy * P + qx;
                const cz zk = sm.z[a1], zm = sm.z[a2];
                const double a_ = zk.x, b_ = zk.y, c_ = zm.x, d_ = zm.y;
                // Z(k) = a + ib, Z(-k) = c + id:  4 P(k) = 2 (a d + b c) + i ((c^2 - a^2) + (d^2 - b^2))
                const double pr = (a_ * d_ + b_ * c_) * (2.0 * SC);
                const double pi = ((c_ * c_ - a_ * a_) + (d_ * d_ - b_ * b_)) * SC;
                sm.z[a1] = cz{pr, pi};
                if (e2 != e) sm.z[a2] = cz{pr, -pi};       // P(-k) = conj P(k)
            }
            lds_barrier();
        }

        // ---- inverse 2-D transform: natural order out; the map is the real plane
        transform<WS, WS, true, false>(sm);
        transform<WS, WS, false, false>(sm);

        // ---- peak analysis in fftshift coordinates (y' = (y + WS/2) % WS, x' likewise), float64.
        //      A thread's cells (e = tid + k NT) stay in registers through the three scans: one read of the
        //      real parts and one write of the finished map instead of four passes over the tile.
        auto dmin = [](double a, double b) { return a < b ? a : b; };
        constexpr int CPT = (NN + NT - 1) / NT;            // cells per thread
        double cv[CPT];
        double cmin = 1.7e308;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = tid + k * NT;
            cv[k] = e < NN ? sm.z[(e / WS) * P + e % WS].x : 1.7e308;
            cmin = dmin(cmin, cv[k]);
        }
        cmin = wg_reduce<G::NW>(cmin, dmin, sm.redd);
        double bv = -1.0;
        int bf = NN;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = tid + k * NT;
            if (e >= NN) break;
            const int y = e / WS, x = e % WS;
            const double v = __dadd_rn(__dsub_rn(cv[k], cmin), 1e-7);      // B:518, B:381
            cv[k] = v;
            sm.z[y * P + x].x = v;
            const int f = ((y + WS / 2) % WS) * WS + (x + WS / 2) % WS;
            if (v > bv || (v == bv && f < bf)) {
                bv = v;
                bf = f;
            }
        }
        const double gmax = wg_reduce<G::NW>(bv, [](double a, double b) { return a > b ? a : b; }, sm.redd + 8);
        auto imin = [](int a, int b) { return a < b ? a : b; };
        const int m = wg_reduce<G::NW>(bv == gmax ? bf : NN, imin, sm.redi);           // first flat index (B:383)
        // (two cheap VALU reductions; a fused (value, index) key would need a three-dword exchange per step)
        // (the reductions above also order the map writes before the neighbour reads below)
        const int wv = p.val_win;
        const int my = m / WS, mx = m % WS;
        double sv = -1.0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int e = tid + k * NT;
            if (e >= NN) break;
            const int y = e / WS, x = e % WS;
            const int fy = (y + WS / 2) % WS, fx = (x + WS / 2) % WS;
            const int f = fy * WS + fx;
            // f in {clamp(m + i + WS j), |i|, |j| <= wv} (B:352-357): in row fy the columns mx + i (j = fy - my),
            // mx + i + WS (j = fy - my - 1 ... the row wrap) and mx + i - WS, plus the two clamps
            const int dj = fy - my;
            bool excl = false;
            if (dj >= -wv && dj <= wv) excl |= (fx >= mx - wv && fx <= mx + wv);
            if (dj + 1 >= -wv && dj + 1 <= wv) excl |= (fx >= mx - wv + WS && fx <= mx + wv + WS);
            if (dj - 1 >= -wv && dj - 1 <= wv) excl |= (fx >= mx - wv - WS && fx <= mx + wv - WS);
            if (f == 0 && (m - wv - wv * WS) <= 0) excl = true;
            if (f == NN - 1 && (m + wv + wv * WS) >= NN - 1) excl = true;
            if (!excl && cv[k] > sv) sv = cv[k];
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
            double outv = sm.z[((ys + WS / 2) % WS) * P + (xs + WS / 2) % WS].x;
            // nothing left outside the exclusion zone: the reference's second arg-max runs over the zeroed
            // map, whose float64 storage `cor` aliases in pass 1 (B:382): c[m2] = 0, ratio = +inf
            outv = (tid == 5) ? (sv >= 0.0 ? sv : 0.0) : outv;
            outv = (tid == 6) ? (double)m : outv;
            outv = (tid == 7) ? (dead ? 1.0 : 0.0) : outv;
            reinterpret_cast<double*>(p.peak_raw)[fidx * 8 + tid] = outv;
        }
        lds_barrier();                                   // planes free for the next window
    }
}


// =====================================================================================================
// 64x64 windows, second generation: every 64-point line split over two lanes, 32-point in-register codelets,
// the radix-2 steps folded into planar LDS transposes (scheme and per-thread arithmetic: xcorr_f64_split.hpp,
// which the CPU suite runs thread by thread against numpy).  One window per 128-thread workgroup, one 33 KB
// float64 plane, four workgroups per CU (two wavefronts per SIMD, <= 256 VGPRs).  The first generation above
// kept the 16-byte elements in LDS and walked radix-8 stages over them: 35 ms per 256 pairs at 43 % LDS bank
// conflicts and ~25 barriers per window (profiles/r02); it still serves 8 ... 32 pixel windows.
// =====================================================================================================
template <int W>
struct F64SplitShared {
    double plane[W * (W + 1)];
    double redd[16];
    unsigned long long redu[4];
    int redi[4];
};

// W = 64: 128 threads (two wavefronts: wavefront = half of every line), 33 KB, four workgroups per CU, <= 256 VGPRs.
// W = 128: 256 threads (wavefronts 0-1 = first halves of lines 0..63 / 64..127, wavefronts 2-3 = second halves), 132 KB:
//          one workgroup per CU and one wavefront per SIMD with up to 512 registers (64-point codelets hold 256).
template <int W>
__global__ __launch_bounds__(2 * W, W == 64 ? 2 : 1) void xcorr_f64_split_kernel(PassParams p) {
    using S = f64s::Split<W>;
    using f64s::dmax2;
    using f64s::dmin2;
    using f64s::peak_shifted;
    constexpr int M = S::M, PL = S::PL, NDW = S::NDW, NWV = 2 * W / 64;      // wavefronts per workgroup
    constexpr int LB = W == 64 ? 6 : 7;                                          // log2 W: line = t & (W - 1), half = t >> LB
    __shared__ F64SplitShared<W> sm;
    double* const plane = sm.plane;
    const int tid = threadIdx.x;

    const int N = p.n_rows * p.n_cols;
    const long long items = (long long)p.batch * N;
    const int st = p.ws - p.ov;
    const int HW = p.H * p.W;
    // XCD-aware static order: workgroups b, b+8, ... share an XCD and walk one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < items ? lo + chunk : items;

    // both threads of a line load the whole image row; the next window's rows are fetched while the peak analysis of
    // the current one runs
    uint32_t da[NDW], db[NDW];
    auto fetch = [&](long long it) TPIV_LAMBDA_INLINE {
        const int pair_ = (int)(it / N), win_ = (int)(it % N);
        const int yy0 = (win_ / p.n_cols) * st, xx0 = (win_ % p.n_cols) * st;
        const size_t off = (size_t)pair_ * HW + (size_t)(yy0 + (tid & (W - 1))) * p.W + xx0;
        load_dwords<NDW>(p.A + off, da);
        load_dwords<NDW>(p.B + off, db);
    };
    // exchange between the wavefronts of the workgroup, ONE barrier: every exchange has its own LDS slots (re-used a
    // window later, many barriers on)
    auto exchange = [&](auto v, auto op, auto* slots) TPIV_LAMBDA_INLINE {
        v = grp_reduce<64>(v, op);
        if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = v;
        lds_barrier();
        auto r = slots[0];
#pragma unroll
        for (int w_ = 1; w_ < NWV; ++w_) r = op(r, slots[w_]);
        return r;
    };
    if (lo + slot < hi) fetch(lo + slot);
    TPIV_STAMP_DECL
    TPIV_STAMP_START;
    for (long long item = lo + slot; item < hi; item += per_xcd) {
#ifdef TPIV_STAMPS
        ++st_iter;
#endif
        // (lane-derived values are re-made from an opaque copy of the thread index at every phase, so that the
        //  loop-invariant LDS addresses are not hoisted out of the item loop into registers)
#define TPIV_F64_TID() [&]() TPIV_LAMBDA_INLINE { int t_ = tid; asm volatile("" : "+v"(t_)); return t_; }()
        // ---- window sums: exact integers.  64x64: every wavefront holds all 64 rows (no exchange); 128x128: the rows of
        //      wavefronts 0 and 1 together
        unsigned ia = 0, ib = 0;
#pragma unroll
        for (int q = 0; q < NDW; ++q) {
            ia = __builtin_amdgcn_sad_u8(da[q], 0u, ia);
            ib = __builtin_amdgcn_sad_u8(db[q], 0u, ib);
        }
        {
            auto uadd = [](unsigned long long a, unsigned long long b) TPIV_LAMBDA_INLINE { return a + b; };
            unsigned long long s2 = grp_reduce<64>((unsigned long long)ia | ((unsigned long long)ib << 32), uadd);
            if constexpr (W == 128) {
                if ((threadIdx.x & 63) == 0) sm.redu[threadIdx.x >> 6] = s2;
                lds_barrier();
                s2 = sm.redu[0] + sm.redu[1];
            }
            ia = (unsigned)s2;
            ib = (unsigned)(s2 >> 32);
        }
        const bool dead = ia == 0u || ib == 0u;          // zero-mean window: 0/0 = NaN map in the reference
        // a / mean(a), b / mean(b) (B:513-514) as ONE factor on the correlation map (xcorr_f64_split.hpp, rows_forward):
        // n^2 / sum(a) * n^2 / sum(b), times 1/n^2 of the inverse transform and the 1/4 of the cross-spectrum algebra;
        // sum(a) sum(b) < 2^44 is exact, so this is one correctly rounded division per window
        const double map_scale = dead ? 0.0 : ((double)(W * W) * 0.25) / ((double)ia * (double)ib);

        // ---- R: rows forward
        cd x[M];
        S::rows_forward(da, db, TPIV_F64_TID() >> LB, x);
        TPIV_STAMP(0);      // window sums, rows forward

        // ---- T1 + C: transposition with the DIF step of the column transform, columns forward
        cd u[M];
        lds_barrier();                               // plane free (the previous window's record reads)
        {
            const int t_ = TPIV_F64_TID();
            S::template t1_write<0>(x, t_ & (W - 1), t_ >> LB, plane);
        }
        lds_barrier();
        {
            const int t_ = TPIV_F64_TID();
            S::template t1_read<0>(u, t_ & (W - 1), 1 - (t_ >> LB), plane);
        }
        lds_barrier();
        {
            const int t_ = TPIV_F64_TID();
            S::template t1_write<1>(x, t_ & (W - 1), t_ >> LB, plane);
        }
        lds_barrier();
        {
            const int t_ = TPIV_F64_TID();
            S::template t1_read<1>(u, t_ & (W - 1), 1 - (t_ >> LB), plane);
        }
        TPIV_STAMP(1);      // transposition 1 (4 barriers)
        const int g = 1 - (TPIV_F64_TID() >> LB);    // parity of the column bins this thread owns (wave-uniform)
        S::cols_forward(u, g);
        TPIV_STAMP(2);      // columns forward

        // ---- X: cross-spectrum
        if constexpr (W == 64) {
            // the mirrored bin sits in lane (64 - k) % 64 of the same wavefront
            const int partner = (64 - (TPIV_F64_TID() & 63)) & 63;
            auto sh = [](double v, int, int, int pt) TPIV_LAMBDA_INLINE { return __shfl(v, pt, 64); };
            if (g == 0) S::template cross_spectrum_g<0>(u, partner, sh);
            else S::template cross_spectrum_g<1>(u, partner, sh);
        } else {
            // the mirrored column lives in another wavefront: through the plane, one component at a time
            double mre[M];
            lds_barrier();                           // every thread has read its T1 column
            {
                const int t_ = TPIV_F64_TID();
                S::template cross_write<0>(u, t_ & (W - 1), 1 - (t_ >> LB), plane);
            }
            lds_barrier();
            {
                const int t_ = TPIV_F64_TID();
                S::cross_read(mre, t_ & (W - 1), 1 - (t_ >> LB), plane);
            }
            lds_barrier();
            {
                const int t_ = TPIV_F64_TID();
                S::template cross_write<1>(u, t_ & (W - 1), 1 - (t_ >> LB), plane);
            }
            lds_barrier();
            double mim[M];
            {
                const int t_ = TPIV_F64_TID();
                S::cross_read(mim, t_ & (W - 1), 1 - (t_ >> LB), plane);
            }
            S::cross_finish(u, mre, mim);
        }

        TPIV_STAMP(3);      // cross-spectrum incl. the exchange
        // ---- Ci + T2: columns inverse, transposition with the DIT step
        cd t[M];
        S::cols_inverse(u, g, t);
        TPIV_STAMP(4);      // columns inverse
        cd Y[M + 1];
        lds_barrier();                               // every thread has read what it needs from the plane
        {
            const int t_ = TPIV_F64_TID();
            S::template t2_write<0>(t, t_ & (W - 1), 1 - (t_ >> LB), plane);
        }
        lds_barrier();
        S::template t2_read<0>(Y, TPIV_F64_TID() & (W - 1), plane);
        lds_barrier();
        {
            const int t_ = TPIV_F64_TID();
            S::template t2_write<1>(t, t_ & (W - 1), 1 - (t_ >> LB), plane);
        }
        lds_barrier();
        S::template t2_read<1>(Y, TPIV_F64_TID() & (W - 1), plane);

        TPIV_STAMP(5);      // transposition 2 (4 barriers)
        // ---- Ri: rows inverse (c2r over the thread pair)
        double c[M];
        S::rows_inverse(Y, TPIV_F64_TID() >> LB, c);
        TPIV_STAMP(6);      // rows inverse

        // ---- P: peak analysis on the float64 map.  Three exchanges between the wavefronts, ONE barrier each:
        //      (min, raw max) -> map -> first row of the maximum -> second peak.
        auto dmin = [](double a, double b) TPIV_LAMBDA_INLINE { return dmin2(a, b); };
        auto dmax = [](double a, double b) TPIV_LAMBDA_INLINE { return dmax2(a, b); };
        auto imin = [](int a, int b) TPIV_LAMBDA_INLINE { return a < b ? a : b; };
        double cmin, rraw;
        S::peak_local_minmax(c, cmin, rraw);
        // prefetch: the last iteration re-loads its own window (no branch around the loads)
        fetch(item + per_xcd < hi ? item + per_xcd : item);
        {
            const double mn = grp_reduce<64>(cmin, dmin), mx = grp_reduce<64>(rraw, dmax);
            const int t_ = TPIV_F64_TID();
            if ((t_ & 63) == 0) {
                sm.redd[t_ >> 6] = mn;
                sm.redd[4 + (t_ >> 6)] = mx;
            }
            lds_barrier();                            // (also: every thread has read its T2 row -> the map may be written)
        }
        double graw = sm.redd[4];
        cmin = sm.redd[0];
#pragma unroll
        for (int w_ = 1; w_ < NWV; ++w_) {
            cmin = dmin2(cmin, sm.redd[w_]);
            graw = dmax2(graw, sm.redd[4 + w_]);
        }
        // the maximum of the shifted map is the shifted raw maximum (monotonic, same roundings)
        const double gmax = peak_shifted(graw, cmin, map_scale);
        {
            const int t_ = TPIV_F64_TID();
            S::peak_shift_and_write(c, cmin, map_scale, t_ & (W - 1), t_ >> LB, plane);
        }
        // arg-max = FIRST flat index holding the maximum (B:383): the smallest shifted row whose maximum is the global
        // one, then the first column of that row -- lane = column, LDS reads and ballots (every wavefront does it: same
        // row, same result, no exchange)
        int ywin;
        {
            const int t_ = TPIV_F64_TID();
            const int fy = ((t_ & (W - 1)) + W / 2) & (W - 1);
            ywin = exchange(peak_shifted(rraw, cmin, map_scale) == gmax ? fy : W - 1, imin, sm.redi);      // (also: map complete)
        }
        int xwin = W - 1;
        {
            const int l_ = TPIV_F64_TID() & 63;
#pragma unroll
            for (int part = W / 64 - 1; part >= 0; --part) {
                const unsigned long long hit = __ballot(plane[ywin * PL + 64 * part + l_] == gmax);
                xwin = hit ? 64 * part + (int)__builtin_ctzll(hit) : xwin;
            }
        }
        const int m = ywin * W + xwin;
        double sv;
        {
            const int t_ = TPIV_F64_TID();
            sv = exchange(S::peak_second_local(c, t_ & (W - 1), t_ >> LB, m, p.val_win), dmax, sm.redd + 8);
        }
        if (tid < 8) reinterpret_cast<double*>(p.peak_raw)[(size_t)item * 8 + tid] = S::peak_record_slot(tid, m, sv, dead, plane);
        TPIV_STAMP(7);      // peak analysis incl. the issue of the next window's loads
#undef TPIV_F64_TID
    }
    TPIV_STAMP_FLUSH(p);
}

template <int W>
static hipError_t launch_f64_split(const PassParams& p, int n_cu, hipStream_t stream) {
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    if (items <= 0) return hipErrorInvalidValue;
    const int per_cu = W == 64 ? 4 : 1;              // LDS: 33.4 KB / 132.3 KB per workgroup
    long long blocks = items < (long long)n_cu * per_cu ? items : (long long)n_cu * per_cu;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((xcorr_f64_split_kernel<W>), dim3((unsigned)blocks), dim3(2 * W), 0, stream, p);
    return hipGetLastError();
}

// =====================================================================================================
// 8 ... 32 pixel windows, second generation: lane = one image row of a window, the whole line (WS complex float64
// samples, frame a in the real part, frame b in the imaginary part) in registers -- the float32 tile kernel's scheme
// (xcorr_tile.hpp) in float64.  A workgroup is ONE wavefront = 64 / WS windows: no barriers anywhere; LDS only
// transposes between the row and the column transform, one float64 plane of WS x (WS + 1) per window at a time
// (32x32: 16.9 KB per wavefront, eight wavefronts per CU = two per SIMD at <= 256 VGPRs), and holds the map for the
// data-dependent look-ups of the peak analysis.  The last transform is a c2r one (only spectrum columns 0 .. WS/2 cross
// the LDS the second time); the normalisation is one factor on the map (xcorr_f64_split.hpp, rows_forward).
// =====================================================================================================
template <int WS>
struct F64TileGeo {
    static constexpr int WPW = 64 / WS;             // windows per wavefront
    static constexpr int P = WS + 1;                // plane pitch in doubles
    static constexpr int PLANE = WS * P;            // doubles per window
    static constexpr int NDW = WS / 4;
    static constexpr int M = WS / 2;
};

// 8-value batches of explicit ds_read_b64 (see xcorr_f64_split.hpp: merged ds_read2_b64 pairs run at half rate)
template <int N, int STRIDE_BYTES, typename F>
__device__ __forceinline__ void lds_read_seq(unsigned base, F&& sink) {
    // sink(index, value) for index = 0 .. N-1 at byte offsets index * STRIDE_BYTES
    constexpr int NB = (N + 7) / 8;
    double v[2][8];
    auto issue = [&](auto bc) TPIV_LAMBDA_INLINE {
        constexpr int b_ = decltype(bc)::value;
        static_for<0, 8>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int i = 8 * b_ + decltype(ic)::value;
            if constexpr (i < N) v[b_ & 1][decltype(ic)::value] = f64s::lds_rd<i * STRIDE_BYTES>(base);
            else v[b_ & 1][decltype(ic)::value] = 0.0;
        });
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, NB>([&](auto bc) TPIV_LAMBDA_INLINE {
        constexpr int b_ = decltype(bc)::value;
        if constexpr (b_ + 1 < NB) issue(std::integral_constant<int, b_ + 1>{});
        constexpr int next = (b_ + 1 < NB) ? ((N - 8 * (b_ + 1)) < 8 ? (N - 8 * (b_ + 1)) : 8) : 0;
        f64s::lds_wait<next>(v[b_ & 1]);
        static_for<0, 8>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int i = 8 * b_ + decltype(ic)::value;
            if constexpr (i < N) sink(std::integral_constant<int, i>{}, v[b_ & 1][decltype(ic)::value]);
        });
    });
}

template <int WS>
__global__ __launch_bounds__(64, 2) void xcorr_f64_tile_kernel(PassParams p) {
    using G = F64TileGeo<WS>;
    constexpr int P = G::P, M = G::M, NDW = G::NDW, WPW = G::WPW;
    __shared__ double tile[WPW * G::PLANE];

    const int N = p.n_rows * p.n_cols;
    const int groups = (N + WPW - 1) / WPW;
    const long long items = (long long)p.batch * groups;
    const int st = p.ws - p.ov;
    const int HW = p.H * p.W;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < items ? lo + chunk : items;

    // item -> (pair, window) of this lane; lanes of a window slot past the last window of a pair re-do that last window
    // (their stores are suppressed)
    auto geom = [&](long long it, int& pair_, int& win_, bool& active) TPIV_LAMBDA_INLINE {
        const int w_ = (int)(threadIdx.x / WS);
        pair_ = (int)(it / groups);
        const int raw = (int)(it % groups) * WPW + w_;
        active = raw < N;
        win_ = active ? raw : N - 1;
    };
    uint32_t da[NDW], db[NDW];
    auto fetch = [&](long long it) TPIV_LAMBDA_INLINE {
        int pair_, win_;
        bool act_;
        geom(it, pair_, win_, act_);
        const int yy0 = (win_ / p.n_cols) * st, xx0 = (win_ % p.n_cols) * st;
        const size_t off = (size_t)pair_ * HW + (size_t)(yy0 + (int)(threadIdx.x % WS)) * p.W + xx0;
        load_dwords<NDW>(p.A + off, da);
        load_dwords<NDW>(p.B + off, db);
    };
    if (lo + slot < hi) fetch(lo + slot);
    for (long long item = lo + slot; item < hi; item += per_xcd) {
        int pair, win;
        bool active;
        geom(item, pair, win, active);
        const size_t fidx = (size_t)pair * N + win;
        const int lane = fresh_lane();
        const int w = lane / WS, r = lane % WS;
        double* const pl = tile + w * G::PLANE;          // this window's plane

        // ---- window sums (exact integers), map scale
        unsigned ia = 0, ib = 0;
#pragma unroll
        for (int q = 0; q < NDW; ++q) {
            ia = __builtin_amdgcn_sad_u8(da[q], 0u, ia);
            ib = __builtin_amdgcn_sad_u8(db[q], 0u, ib);
        }
        {
            auto uadd = [](unsigned long long a, unsigned long long b) TPIV_LAMBDA_INLINE { return a + b; };
            const unsigned long long s2 = grp_reduce<WS>((unsigned long long)ia | ((unsigned long long)ib << 32), uadd);
            ia = (unsigned)s2;
            ib = (unsigned)(s2 >> 32);
        }
        const bool dead = ia == 0u || ib == 0u;
        const double map_scale = dead ? 0.0 : ((double)(WS * WS) * 0.25) / ((double)ia * (double)ib);

        // ---- rows forward: x[k] = a[k] + i b[k], bin kx at x[FFT_POS<kx>]
        cd x[WS];
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            x[k] = cd{(double)byte_f<k, NDW>(da), (double)byte_f<k, NDW>(db)};
        });
        fft_inreg_d<WS, 1>(x);

        // ---- transposition 1 (one plane at a time): lane (w, kx) gets x[y] = X[y][kx]
        {
            const unsigned rd_base = f64s::lds_addr(pl + r);
            wave_sync();
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                pl[r * P + k] = x[FFT_POS<k, WS>].x;
            });
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double xr[WS];
            lds_read_seq<WS, P * 8>(rd_base, [&](auto ic, double v) TPIV_LAMBDA_INLINE { xr[decltype(ic)::value] = v; });
            wave_sync();
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                pl[r * P + k] = x[FFT_POS<k, WS>].y;
            });
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_read_seq<WS, P * 8>(rd_base, [&](auto ic, double v) TPIV_LAMBDA_INLINE {
                constexpr int i = decltype(ic)::value;
                x[i] = cd{xr[i], v};
            });
        }
        // ---- columns forward: Z(ky, kx = lane) at x[FFT_POS<ky>]
        fft_inreg_d<WS, 1>(x);

        // ---- cross-spectrum: Z(-ky, -kx) sits in lane (-kx mod WS) of the window, register (-ky mod WS)
        {
            const int partner = (lane - r) + ((WS - r) % WS);
            static_for<0, WS / 2 + 1>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int ky = decltype(kc)::value;
                constexpr int nky = (WS - ky) % WS;
                constexpr int p1 = FFT_POS<ky, WS>, p2 = FFT_POS<nky, WS>;
                cd z1 = x[p1];
                f64s::pin(z1);
                if constexpr (ky == nky) {
                    const cd m1{__shfl(z1.x, partner, 64), __shfl(z1.y, partner, 64)};
                    x[p1] = f64s::cross_bin(z1, m1);
                } else {
                    cd z2 = x[p2];
                    f64s::pin(z2);
                    const cd m1{__shfl(z2.x, partner, 64), __shfl(z2.y, partner, 64)};       // Z(-ky, -kx)
                    const cd m2{__shfl(z1.x, partner, 64), __shfl(z1.y, partner, 64)};       // Z(+ky, -kx)
                    x[p1] = f64s::cross_bin(z1, m1);
                    x[p2] = f64s::cross_bin(z2, m2);
                }
            });
        }

        // ---- columns inverse (natural-order input: rename), row y at t[FFT_POS<y>]
        cd t[WS];
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int ky = decltype(kc)::value;
            t[ky] = x[FFT_POS<ky, WS>];
        });
        fft_inreg_d<WS, -1>(t);

        // ---- transposition 2, half: the map rows are real, so lane (w, y) needs spectrum columns 0 .. WS/2 only
        cd hs[M + 1];
        {
            const unsigned rd_base = f64s::lds_addr(pl + r);
            cd b[WS];
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                b[k] = t[FFT_POS<k, WS>];
                asm volatile("" : "+v"(b[k].x), "+v"(b[k].y));       // (keeps the tail of the transform out of the branch)
            });
            wave_sync();
            if (r <= M) {
                static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    pl[r * P + k] = b[k].x;                            // row kx, column y
                });
            }
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_read_seq<M + 1, P * 8>(rd_base, [&](auto ic, double v) TPIV_LAMBDA_INLINE { hs[decltype(ic)::value].x = v; });
            wave_sync();
            if (r <= M) {
                static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    pl[r * P + k] = b[k].y;
                });
            }
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_read_seq<M + 1, P * 8>(rd_base, [&](auto ic, double v) TPIV_LAMBDA_INLINE { hs[decltype(ic)::value].y = v; });
            wave_sync();
        }
        // ---- rows inverse (c2r): corr(y = r, 2m) + i corr(y, 2m + 1) at hs[FFT_POS<m, M>]
        c2r_pre_d<WS>(hs);
        cd z[M];
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            z[m] = hs[m];
        });
        fft_inreg_d<M, -1>(z);
        double c[WS];                                  // c[x'] in fftshift coordinates
        static_for<0, WS>([&](auto xc) TPIV_LAMBDA_INLINE {
            constexpr int xs = decltype(xc)::value;
            constexpr int xo = (xs + WS / 2) % WS;
            c[xs] = (xo & 1) ? z[FFT_POS<xo / 2, M>].y : z[FFT_POS<xo / 2, M>].x;
        });

        // ---- next item's rows fly during the peak analysis (the last iteration re-loads its own)
        fetch(item + per_xcd < hi ? item + per_xcd : item);

        // ---- peak analysis on the float64 map (B:346-358, B:381-392, B:518)
        {
            auto dmin = [](double a_, double b_) TPIV_LAMBDA_INLINE { return f64s::dmin2(a_, b_); };
            auto dmax = [](double a_, double b_) TPIV_LAMBDA_INLINE { return f64s::dmax2(a_, b_); };
            auto imin = [](int a_, int b_) TPIV_LAMBDA_INLINE { return a_ < b_ ? a_ : b_; };
            double cmin = c[0], rraw = c[0];
#pragma unroll
            for (int k = 1; k < WS; ++k) {
                cmin = f64s::dmin2(cmin, c[k]);
                rraw = f64s::dmax2(rraw, c[k]);
            }
            cmin = grp_reduce<WS>(cmin, dmin);
            const int ys = (r + WS / 2) % WS;
            wave_sync();
            static_for<0, WS>([&](auto xc) TPIV_LAMBDA_INLINE {
                constexpr int xs = decltype(xc)::value;
                const double v = f64s::peak_shifted(c[xs], cmin, map_scale);
                c[xs] = v;
                pl[ys * P + xs] = v;
            });
            const double rmax = f64s::peak_shifted(rraw, cmin, map_scale);
            const double gmax = grp_reduce<WS>(rmax, dmax);
            const int ywin = grp_reduce<WS>(rmax == gmax ? ys : WS - 1, imin);       // first row holding the maximum
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int xwin = grp_reduce<WS>(pl[ywin * P + r] == gmax ? r : WS - 1, imin);   // lane r = column r
            const int KD = WS * WS;
            const int m = ywin * WS + xwin;
            const int wv = p.val_win;
            // second peak: maximum outside the flat-index exclusion zone (see xcorr_tile.hpp, peak_analysis)
            double sv;
            {
                const int dj = ys - ywin;
                unsigned ex = 0u;
                auto span = [&](int lo_, int hi_) TPIV_LAMBDA_INLINE {
                    lo_ = lo_ < 0 ? 0 : lo_;
                    hi_ = hi_ > WS - 1 ? WS - 1 : hi_;
                    if (lo_ > hi_) return 0u;
                    const int len = hi_ - lo_ + 1;
                    const unsigned ones = len >= 32 ? ~0u : ((1u << len) - 1u);
                    return ones << lo_;
                };
                if (dj >= -wv && dj <= wv) ex |= span(xwin - wv, xwin + wv);
                if (dj + 1 >= -wv && dj + 1 <= wv) ex |= span(xwin - wv + WS, xwin + wv + WS);
                if (dj - 1 >= -wv && dj - 1 <= wv) ex |= span(xwin - wv - WS, xwin + wv - WS);
                if (ys == 0 && (m - wv - wv * WS) <= 0) ex |= 1u;
                if (ys == WS - 1 && (m + wv + wv * WS) >= KD - 1) ex |= 1u << (WS - 1);
                sv = -1.0;
                static_for<0, WS>([&](auto xc) TPIV_LAMBDA_INLINE {
                    constexpr int xs = decltype(xc)::value;
                    const int kill = __builtin_amdgcn_sbfe((int)ex, xs, 1);
                    const double v = __hiloint2double(__double2hiint(c[xs]) | (kill & (int)0x80000000), __double2loint(c[xs]));
                    sv = f64s::dmax2(sv, v);
                });
                sv = grp_reduce<WS>(sv, dmax);
            }
            if (r < 8 && active) {
                int left = m + 1, right = m - 1, top = m + WS, bot = m - WS;      // B:385-392 (flat index)
                if (left >= KD - 1) left = m;
                if (right <= 0) right = m;
                if (top >= KD - 1) top = m;
                if (bot <= 0) bot = m;
                int q = m;
                q = (r == 1) ? left : q;
                q = (r == 2) ? right : q;
                q = (r == 3) ? top : q;
                q = (r == 4) ? bot : q;
                double outv = pl[(q / WS) * P + (q % WS)];
                outv = (r == 5) ? (sv >= 0.0 ? sv : 0.0) : outv;
                outv = (r == 6) ? (double)m : outv;
                outv = (r == 7) ? (dead ? 1.0 : 0.0) : outv;
                reinterpret_cast<double*>(p.peak_raw)[fidx * 8 + r] = outv;
            }
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // record look-ups done before the plane is reused
        }
    }
}

template <int WS>
static hipError_t launch_f64_tile(const PassParams& p, int n_cu, hipStream_t stream) {
    using G = F64TileGeo<WS>;
    const long long groups = ((long long)p.n_rows * p.n_cols + G::WPW - 1) / G::WPW;
    const long long items = (long long)p.batch * groups;
    if (items <= 0) return hipErrorInvalidValue;
    long long blocks = items < (long long)n_cu * 8 ? items : (long long)n_cu * 8;       // two wavefronts per SIMD
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((xcorr_f64_tile_kernel<WS>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

template <int WS>
hipError_t launch_f64(const PassParams& p, int n_cu, hipStream_t stream) {
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    if (items <= 0) return hipErrorInvalidValue;
    const int per_cu = WS == 64 ? 2 : (WS == 32 ? 8 : 16);      // LDS: 67.7 KB / 17.5 KB / ... per workgroup
    long long blocks = items < (long long)n_cu * per_cu ? items : (long long)n_cu * per_cu;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((xcorr_f64_kernel<WS>), dim3((unsigned)blocks), dim3(F64Geo<WS>::NT), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

// TPIV_F64_GEN1=1: the first-generation LDS-resident kernel (A/B runs)
static bool gen1() {
    static const bool v = [] {
        const char* e = getenv("TPIV_F64_GEN1");
        return e && e[0] == '1';
    }();
    return v;
}

hipError_t launch_xcorr_f64(const PassParams& p, int n_cu, hipStream_t stream) {
    switch (p.ws) {
        case 8: return gen1() ? launch_f64<8>(p, n_cu, stream) : launch_f64_tile<8>(p, n_cu, stream);
        case 16: return gen1() ? launch_f64<16>(p, n_cu, stream) : launch_f64_tile<16>(p, n_cu, stream);
        case 32: return gen1() ? launch_f64<32>(p, n_cu, stream) : launch_f64_tile<32>(p, n_cu, stream);
        case 64: return gen1() ? launch_f64<64>(p, n_cu, stream) : launch_f64_split<64>(p, n_cu, stream);
        case 128: return launch_f64_split<128>(p, n_cu, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace tpiv
