// Second-generation kernel for 128x128 interrogation windows, first pass (no window shift).
//
// A 128-point line does not fit one lane (128 complex samples = 256 VGPRs), so a line is split over
// TWO threads by sample parity and every 128-point transform becomes a 64-point in-register codelet
// (fft_inreg.hpp) plus one radix-2 combine, and the combines are folded into the LDS transposes that
// are needed anyway:
//
//   workgroup = 256 threads = one window;  thread t:  line = t & 127,  parity = t >> 7
//   stage 0   rows:    thread (y, h) takes samples x = 2j + h of image row y (a in .x, b in .y),
//                      window means by a block reduction, normalise (B:513-514)
//   stage 1   rows:    64-point forward FFT              F_h[y][k1]
//   stage 2   T1:      plane[y][64h + k1] <- F_h;   thread (k, g) reads rows y = 2i + g and forms
//                      X[y][k] = F_0[y][k1] +- w^k1 F_1[y][k1]          (decimation in time, k1 = k mod 64)
//   stage 3   columns: 64-point forward FFT over i       G_g[ky1]       (column k, row parity g)
//   stage 4   spectrum: plane[ky1 + 64g][k] <- (g ? w^ky1 G_1 : G_0);  thread (k, a) forms
//                      Z[ky1 + 64a][k] = G_0 +- w^ky1 G_1  and the mirrored bin Z[-ky][-k] from the
//                      same plane, then the cross-spectrum P = conj(A) B of the packed transform
//   stage 5   columns^-1: partner exchange through the plane (decimation in frequency):
//                      u = P[ky1] + P[ky1+64]  (even rows, a = 0),  (P[ky1] - P[ky1+64]) w^-ky1  (odd rows)
//                      and a 64-point inverse FFT  ->  Y[2i + a][k]
//   stage 6   T2 + rows^-1: plane[2i + a][k] <- Y;  thread (y, h) reads its row, forms the DIF
//                      split over k (bins 0..32 only: Hermitian) and runs a c2r codelet -> corr[y][2j + h]
//   stage 7   peak analysis of the 128x128 map in LDS (same record for finalize_kernel as the tile
//                      kernel: B:346-358, B:381-392, B:518)
//
// The LDS plane holds ONE float component of the 128x128 tile (66 KB): two workgroups per CU, two
// wavefronts per SIMD, 256 VGPRs per thread.  Real and imaginary planes pass one after the other.
#pragma once
#include "xcorr_tile.hpp"

namespace tpiv {

constexpr int BW = 128;               // window edge
constexpr int BP = 129;               // plane pitch (floats): conflict-free rows and columns
constexpr int BH = 64;                // codelet length

struct BigShared {
    float plane[BW * BP];
    float redf[8];                  // (min, raw max) of the map
    int redi[8];                    // row of the maximum / second peak
    unsigned long long redu[8];     // window sums; CAND: [4..7] the sums of squares
    // candidate-cell variant (precision "exact", xcorr_exact.hip): cells inside the band of the maximum (count only),
    // of the second peak and of the minimum
    int redc[4];
    int n_second, n_min;
    int cand_second[EXACT_MAX_SECOND], cand_min[EXACT_MAX_MIN];
};

// Workgroup barrier for LDS exchanges only: __syncthreads() also waits for vmcnt(0), which would
// drain the next window's prefetch at every one of the ~30 barriers per window.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Reduction over the 256 threads of the workgroup with ONE barrier; every thread gets the result.  Every call site has
// its own four slots, which are written again a whole window later (many barriers on), so no barrier is needed in
// front of the write.
template <typename T, typename OP>
__device__ __forceinline__ T block_reduce(T v, OP op, T* slots, int wave, int lane) {
    v = grp_reduce<64>(v, op);
    if (lane == 0) slots[wave] = v;
    wg_barrier();
    return op(op(slots[0], slots[1]), op(slots[2], slots[3]));
}

// ---- stage 7 of the 128x128 kernel: peak analysis of one window's map (B:346-358, B:381-392, B:518).
// in: thread (line = t & 127, par = t >> 7) holds c[j] = corr(row line, column 2j + par) in UN-shifted
// coordinates; the map is parked in sm.plane in fftshift coordinates.  `prefetch` runs once after the
// thread-local minimum (the kernel issues the next window's row loads there).
// CAND: instead of the record, the flat indices of the cells the exact refinement evaluates (see peak_candidates in
// xcorr_tile.hpp for the rules; here the whole map sits in LDS and every thread looks through its own 64 cells)
// band_abs (CAND): the proven part of the decision band, 2 Gamma(128) (1 + 1/16) E+ (piv_kernels.h, "The band")
template <bool CAND = false, typename PREFETCH>
__device__ __forceinline__ void big_peak_stage(const PassParams& p, float (&c)[BH], BigShared& sm, int t, size_t fidx,
                                               bool dead, PREFETCH&& prefetch, float band_abs = 0.f) {
    float* const plane = sm.plane;
    const int lane = t & 63, wave = t >> 6, par = t >> 7;
    {
        int tq = t;
        asm volatile("" : "+v"(tq));
        const int ys = ((tq & 127) + 64) & 127;
        const int KD = BW * BW;
        // one pass over the raw cells gives their minimum AND their maximum: v = (c - min) + 1e-7 is monotonic in c,
        // so the maximum of the shifted cells is the shifted raw maximum (the same two roundings)
        float cmin = c[0], rraw = c[0];
#pragma unroll
        for (int j = 1; j < BH; ++j) {
            cmin = fminf(cmin, c[j]);
            rraw = fmaxf(rraw, c[j]);
        }
        const float cmin_mine = cmin;
        prefetch();
        if constexpr (CAND) {
            if (t == 0) {
                sm.n_second = 0;
                sm.n_min = 0;
            }
        }
        auto fmin_ = [](float a, float b) TPIV_LAMBDA_INLINE { return fminf(a, b); };
        auto fmax_ = [](float a, float b) TPIV_LAMBDA_INLINE { return fmaxf(a, b); };
        auto imin_ = [](int a, int b) TPIV_LAMBDA_INLINE { return a < b ? a : b; };
        auto imax_ = [](int a, int b) TPIV_LAMBDA_INLINE { return a > b ? a : b; };
        float graw;
        {   // (min, raw max) with one barrier  (also: plane reads of stage 6 done)
            const float mn = grp_reduce<64>(cmin, fmin_), mx = grp_reduce<64>(rraw, fmax_);
            if (lane == 0) {
                sm.redf[wave] = mn;
                sm.redf[4 + wave] = mx;
            }
            wg_barrier();
            cmin = fminf(fminf(sm.redf[0], sm.redf[1]), fminf(sm.redf[2], sm.redf[3]));
            graw = fmaxf(fmaxf(sm.redf[4], sm.redf[5]), fmaxf(sm.redf[6], sm.redf[7]));
        }
        const float gmax = __fadd_rn(__fsub_rn(graw, cmin), 1e-7f);
        static_for<0, BH>([&](auto jc) TPIV_LAMBDA_INLINE {
            constexpr int j = decltype(jc)::value;
            constexpr int xe = (2 * j + 64) & 127;
            const float v = __fadd_rn(__fsub_rn(c[j], cmin), 1e-7f);       // B:518, B:381
            c[j] = v;
            plane[ys * BP + xe + par] = v;
        });
        const float rmax = __fadd_rn(__fsub_rn(rraw, cmin), 1e-7f);
        // arg-max = FIRST flat index holding the maximum (B:383): the smallest row whose maximum is the global one, then
        // the first column of that row -- every wavefront looks the row up itself (ballots), no further exchange
        const int ywin = block_reduce(rmax == gmax ? ys : BW - 1, imin_, sm.redi, wave, lane);   // (also: map complete)
        int xwin = BW - 1;
#pragma unroll
        for (int part = 1; part >= 0; --part) {
            const unsigned long long hit = __ballot(plane[ywin * BP + 64 * part + lane] == gmax);
            xwin = hit ? 64 * part + (int)__builtin_ctzll(hit) : xwin;
        }
        const int m = ywin * BW + xwin;
        if (p.dbg_corr != nullptr) {
            float* d = p.dbg_corr + fidx * KD + ys * BW + par;
            static_for<0, BH>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                constexpr int xe = (2 * j + 64) & 127;
                d[xe] = c[j];
            });
        }
        // second peak: maximum outside the flat-index neighbourhood of m (B:346-358), see peak_analysis
        const int wv = p.val_win;
        int smax = 0;
        {
            const int dj = ys - ywin;
            unsigned long long exl = 0ull, exh = 0ull;     // excluded columns 0..63 / 64..127 of this row
            auto span = [&](int lo_, int hi_) TPIV_LAMBDA_INLINE {      // columns lo_ .. hi_, clipped to the row (closed form)
                lo_ = lo_ < 0 ? 0 : lo_;
                hi_ = hi_ > BW - 1 ? BW - 1 : hi_;
                const int ll = lo_, hl = hi_ < 63 ? hi_ : 63;               // low word
                if (ll <= hl) exl |= (~0ull >> (63 - (hl - ll))) << ll;
                const int lh = (lo_ > 64 ? lo_ : 64) - 64, hh = hi_ - 64;   // high word
                if (lh <= hh) exh |= (~0ull >> (63 - (hh - lh))) << lh;
            };
            if (dj >= -wv && dj <= wv) span(xwin - wv, xwin + wv);
            if (dj + 1 >= -wv && dj + 1 <= wv) span(xwin - wv + BW, xwin + wv + BW);
            if (dj - 1 >= -wv && dj - 1 <= wv) span(xwin - wv - BW, xwin + wv - BW);
            if (ys == 0 && (m - wv - wv * BW) <= 0) exl |= 1ull;
            if (ys == BW - 1 && (m + wv + wv * BW) >= KD - 1) exh |= 1ull << 63;
            // this thread's columns have parity `par`: shift it away, bit positions become even constants
            const unsigned e0 = (unsigned)(exl >> par), e1 = (unsigned)(exl >> (32 + par));
            const unsigned e2 = (unsigned)(exh >> par), e3 = (unsigned)(exh >> (32 + par));
            static_for<0, BH>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                constexpr int xe = (2 * j + 64) & 127;
                constexpr int wsel = xe >> 5, bit = xe & 31;
                const unsigned word = wsel == 0 ? e0 : (wsel == 1 ? e1 : (wsel == 2 ? e2 : e3));
                const int kill = __builtin_amdgcn_sbfe((int)word, bit, 1);
                const int cand = __float_as_int(c[j]) | kill;
                smax = cand > smax ? cand : smax;
            });
        }
        const int smax_mine = smax;
        smax = block_reduce(smax, imax_, sm.redi + 4, wave, lane);
        if constexpr (CAND) {
            // (all values are the shifted cells v = (c - min) + 1e-7 > 0: differences are those of the raw map)
            const float band = fmaxf(p.exact_band_range * __fsub_rn(graw, cmin), band_abs);
            bool open = !(band > 0.0f) || !(graw > cmin);
            // The few threads whose row extremum sits inside a band walk their 64 cells over the map in LDS
            // (plane[ys][xe + par], the shifted cells just parked there) in two steps: a branch-free pass that only sets one bit
            // per matching cell (sixteen reads in flight), then the bits -- one to three -- are visited.  (Two earlier forms: unrolled
            // over the register copy with the append inside, 4 800 lines of code; a rolled loop with a branch per cell, one LDS
            // round trip per cell = 8 000 cycles per walk with the whole workgroup waiting at the next barrier: the locating pass
            // cost 20-25 % more than the plain float32 pass 1 either way.)
            const float* const my_cells = plane + ys * BP + par;
            auto walk = [&](auto&& pred) TPIV_LAMBDA_INLINE {
                unsigned lo = 0u, hi = 0u;
#pragma unroll 1
                for (int jb = 0; jb < BH; jb += 16) {             // sixteen reads issued together, then their tests
                    float v_[16];
                    static_for<0, 16>([&](auto kc) TPIV_LAMBDA_INLINE {
                        constexpr int k = decltype(kc)::value;
                        v_[k] = my_cells[(2 * (jb + k) + 64) & 127];
                    });
                    unsigned bits = 0u;
                    static_for<0, 16>([&](auto kc) TPIV_LAMBDA_INLINE {
                        constexpr int k = decltype(kc)::value;
                        bits |= (pred(v_[k]) ? 1u : 0u) << k;
                    });
                    if (jb < 32) lo |= bits << jb;
                    else hi |= bits << (jb - 32);
                }
                return ((unsigned long long)hi << 32) | lo;
            };
            // arg-max: exactly one cell inside the band of the maximum (then it is the cell found above)
            int cnt = 0;
            if (rmax >= gmax - band) cnt = __popcll(walk([&](float v_) TPIV_LAMBDA_INLINE { return v_ >= gmax - band; }));
            // second peak: the cells outside the exclusion zone inside the band of their maximum
            if (smax > 0 && smax_mine > 0 && __int_as_float(smax_mine) >= __int_as_float(smax) - band) {
                const float thr = __int_as_float(smax) - band;
                const int dj = ys - ywin;
                unsigned long long mk = walk([&](float v_) TPIV_LAMBDA_INLINE { return v_ >= thr; });
                while (mk) {
                    const int j = (int)__builtin_ctzll(mk);
                    mk &= mk - 1;
                    const int q = ys * BW + ((2 * j + 64) & 127) + par;
                    // B:346-358 for one cell: q = clamp(m + i + BW jj), |i|, |jj| <= wv
                    bool ex = false;
                    for (int jj = dj - 1; jj <= dj + 1; ++jj) {
                        const int i = q - m - BW * jj;
                        ex = ex || (jj >= -wv && jj <= wv && i >= -wv && i <= wv);
                    }
                    ex = ex || (q == 0 && m - wv - wv * BW <= 0) || (q == KD - 1 && m + wv + wv * BW >= KD - 1);
                    if (!ex) {
                        const int k = atomicAdd(&sm.n_second, 1);
                        if (k < EXACT_MAX_SECOND) sm.cand_second[k] = q;
                    }
                }
            }
            // minimum: the cells inside the band of it
            if (__fadd_rn(__fsub_rn(cmin_mine, cmin), 1e-7f) <= 1e-7f + band) {
                const float thr = 1e-7f + band;
                unsigned long long mk = walk([&](float v_) TPIV_LAMBDA_INLINE { return v_ <= thr; });
                while (mk) {
                    const int j = (int)__builtin_ctzll(mk);
                    mk &= mk - 1;
                    const int k = atomicAdd(&sm.n_min, 1);
                    if (k < EXACT_MAX_MIN) sm.cand_min[k] = ys * BW + ((2 * j + 64) & 127) + par;
                }
            }
            auto iadd_ = [](int a, int b) TPIV_LAMBDA_INLINE { return a + b; };
            cnt = block_reduce(cnt, iadd_, sm.redc, wave, lane);          // (also: the lists are complete)
            if (t == 0) {
                const int ns = sm.n_second, nn = sm.n_min;
                // (more minimum cells than the record holds: three of them and -2 -- see peak_candidates, xcorr_tile.hpp)
                open = open || cnt != 1 || ns > EXACT_MAX_SECOND;
                if (nn > EXACT_MAX_MIN) sm.cand_min[EXACT_MAX_MIN - 1] = -2;
                auto get = [](const int* l, int n, int k) TPIV_LAMBDA_INLINE { return k < n ? l[k] : -1; };
                auto pack = [](int lo_, int hi_) TPIV_LAMBDA_INLINE { return ((unsigned)lo_ & 0xffffu) | ((unsigned)hi_ << 16); };
                const int m_out = dead ? -2 : (open ? -1 : m);
                uint4 rec;
                rec.x = pack(m_out, get(sm.cand_second, ns, 0));
                rec.y = pack(get(sm.cand_second, ns, 1), get(sm.cand_second, ns, 2));
                rec.z = pack(get(sm.cand_min, nn, 0), get(sm.cand_min, nn, 1));
                rec.w = pack(get(sm.cand_min, nn, 2), get(sm.cand_min, nn, 3));
                p.cand[fidx] = rec;
            }
            return;
        }
        const float second_v = smax > 0 ? __int_as_float(smax) : gmax;
        if (t < 8) {
            int left = m + 1, right = m - 1, top = m + BW, bot = m - BW;     // B:385-392 (flat index)
            if (left >= KD - 1) left = m;
            if (right <= 0) right = m;
            if (top >= KD - 1) top = m;
            if (bot <= 0) bot = m;
            int q = m;
            q = (t == 1) ? left : q;
            q = (t == 2) ? right : q;
            q = (t == 3) ? top : q;
            q = (t == 4) ? bot : q;
            float outv = plane[(q / BW) * BP + (q % BW)];
            outv = (t == 5) ? second_v : outv;
            outv = (t == 6) ? __int_as_float(m) : outv;
            outv = (t == 7) ? __int_as_float(dead ? 1 : 0) : outv;
            p.peak_raw[fidx * 8 + t] = outv;
        }
    }
}

// PAR (round 5): 0 / 1 = the whole window loop instantiated per parity behind ONE wave-uniform branch at the top of the kernel
// (the float64 kernels lost all their scratch that way, xcorr_f64.hip); -1 = the parity a run-time value (round 4)
#ifndef TPIV_BIG_PER_PAR
#define TPIV_BIG_PER_PAR 1
#endif
template <bool CAND, int PAR = -1>
__device__ __forceinline__ void xcorr_big128_body(const PassParams& p, BigShared& sm) {
    float* const plane = sm.plane;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int line_c = t & 127;       // row y (stages 0-1, 6-7) or column k (stages 2-5)
    const int par = PAR >= 0 ? PAR : t >> 7;      // sample / line parity handled by this thread (wave-uniform)
    const float sgn = par ? -1.f : 1.f;

    const int N = p.n_rows * p.n_cols;
    const long long items = (long long)p.batch * N;
    const int st = p.ws - p.ov;
    const int HW = p.H * p.W;

    // forward twiddle of the row combine (stage 2): w^k1 = exp(-2 pi i k1 / 128), sign of the half folded in
    const int k1r_c = line_c & 63;
    const float twr = (float)cospi((double)k1r_c / 64.0), twi = (float)(-sinpi((double)k1r_c / 64.0));
    const float s2 = line_c < 64 ? 1.f : -1.f;
    const float cwr = s2 * twr, cwi = s2 * twi;

    // XCD-aware static order: workgroups b, b+8, ... share an XCD and walk one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < items ? lo + chunk : items;

    // rows of the next window are fetched while the peak analysis of the current one runs
    uint32_t da[32], db[32];
    auto issue_loads = [&](long long it) TPIV_LAMBDA_INLINE {
        const int pair_ = (int)(it / N), win_ = (int)(it % N);
        const int yy0 = (win_ / p.n_cols) * st, xx0 = (win_ % p.n_cols) * st;
        const size_t off = (size_t)pair_ * HW + (size_t)(yy0 + line_c) * p.W + xx0;
        load_dwords<32>(p.A + off, da);
        load_dwords<32>(p.B + off, db);
    };
    if (lo + slot < hi) issue_loads(lo + slot);
    for (long long item = lo + slot; item < hi; item += per_xcd) {
        // LDS addresses are rebuilt from an opaque copy of the line index at every plane pass (OPQ): the
        // backend pairs the plane accesses into ds_read2/ds_write2 (8-bit offsets), which needs a base
        // register per pair; left alone it computes all of them once (hoisted out of the loop or to
        // its head) and spills ~80-130 registers.
#define TPIV_OPQ_T() [&]() TPIV_LAMBDA_INLINE { int t_ = t; asm volatile("" : "+v"(t_)); return t_; }()
        const int line = TPIV_OPQ_T() & 127;
        const int pair = (int)(item / N), win = (int)(item % N);
        const int y0 = (win / p.n_cols) * st, x0 = (win % p.n_cols) * st;
        const size_t fidx = (size_t)item;

        // ---------------- stage 0: samples 2j + par of row `line`
        cf x[BH];
        float sa, sb;
        unsigned sqa = 0u, sqb = 0u;
        {
            const uint32_t mask = par ? 0xff00ff00u : 0x00ff00ffu;
            const int sh = 8 * par;
            unsigned ia = 0, ib = 0, iaa = 0, ibb = 0;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                ia = __builtin_amdgcn_sad_u8(da[q] & mask, 0u, ia);
                ib = __builtin_amdgcn_sad_u8(db[q] & mask, 0u, ib);
                if constexpr (CAND) {     // sums of squares of this thread's bytes: E+, the scale of the decision band
                    iaa = __builtin_amdgcn_udot4(da[q] & mask, da[q] & mask, iaa, false);
                    ibb = __builtin_amdgcn_udot4(db[q] & mask, db[q] & mask, ibb, false);
                }
                da[q] >>= sh;
                db[q] >>= sh;
            }
            static_for<0, BH>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;          // byte 2j (+par, shifted away) of the row
                x[j].x = byte_f<2 * j, 32>(da);
                x[j].y = byte_f<2 * j, 32>(db);
            });
            // window sums: exact integers (< 2^24), both in one reduction with one barrier
            auto uadd = [](unsigned long long a, unsigned long long b) TPIV_LAMBDA_INLINE { return a + b; };
            if constexpr (CAND) {     // (window totals < 2^31: the packed halves do not carry into each other; same barrier)
                const unsigned long long q2 = grp_reduce<64>((unsigned long long)iaa | ((unsigned long long)ibb << 32), uadd);
                if (lane == 0) sm.redu[4 + wave] = q2;
            }
            const unsigned long long s2_ =
                block_reduce((unsigned long long)ia | ((unsigned long long)ib << 32), uadd, sm.redu, wave, lane);
            sa = (float)(unsigned)s2_;
            sb = (float)(unsigned)(s2_ >> 32);
            if constexpr (CAND) {
                const unsigned long long q2 = (sm.redu[4] + sm.redu[5]) + (sm.redu[6] + sm.redu[7]);
                sqa = (unsigned)q2;
                sqb = (unsigned)(q2 >> 32);
            }
        }
        if (p.dbg_win != nullptr) {       // test hook: the staged window
            float* d = p.dbg_win + fidx * 2 * BW * BW + line * BW + par;
#pragma unroll
            for (int j = 0; j < BH; ++j) {
                d[2 * j] = x[j].x;
                d[BW * BW + 2 * j] = x[j].y;
            }
        }
        const bool dead = (sa == 0.f) || (sb == 0.f);   // zero-mean window: 0/0 = NaN map in the reference
        int band_s = 0;                                 // CAND: the decision band (float bits, scalar)
        {
            constexpr float PRE = 0.5f / (float)BW;     // 1/n^2 and the 1/4 of the cross-spectrum, exact
            const float ma = sa * (1.0f / (BW * BW)), mb = sb * (1.0f / (BW * BW));
            const float ka = dead ? 0.f : 1.0f / ma, kb = dead ? 0.f : 1.0f / mb;
            const float oa = -ma * ka, ob = -mb * kb;
            const float kas = ka * PRE, kbs = kb * PRE, oas = oa * PRE, obs = ob * PRE;
            if constexpr (CAND) {     // E+ = (|a'|^2 + |b'|^2) / 2, |a'|^2 = (n sum a^2 - (sum a)^2) ka^2 / n  (see xcorr_tile.hpp)
                constexpr double NN = (double)(BW * BW);
                const double da_ = (double)sa, db_ = (double)sb;
                const float ea = (float)__fma_rn(-da_, da_, NN * (double)sqa), eb = (float)__fma_rn(-db_, db_, NN * (double)sqb);
                const float e_plus = (0.5f / (float)(BW * BW)) * (ea * (ka * ka) + eb * (kb * kb));
                band_s = __builtin_amdgcn_readfirstlane(__float_as_int(dead ? 0.f : p.exact_band * e_plus));
            }
#pragma unroll
            for (int j = 0; j < BH; ++j) {
                x[j].x = fmaf(x[j].x, kas, oas);
                x[j].y = fmaf(x[j].y, kbs, obs);
            }
        }

        // ---------------- stage 1: forward row codelet; F_par[line][k1] at x[FFT_POS<k1>]
        fft_inreg<BH, 1>(x);

        // ---------------- stage 2: transposition + row combine -> x[i] = X[2i + par][k = line]
        {
            float A_[BH], B_[BH];
            wg_barrier();                                   // plane free (previous item's map)
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k1 = decltype(kc)::value;
                    plane[ln * BP + 64 * pr + k1] = x[FFT_POS<k1, BH>].x;
                });
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int k1r = tq & 63, pr = PAR >= 0 ? PAR : tq >> 7;
#pragma unroll
                for (int i = 0; i < BH; ++i) {
                    const float E = plane[(2 * i + pr) * BP + k1r], O = plane[(2 * i + pr) * BP + 64 + k1r];
                    A_[i] = fmaf(cwr, O, E);                   // Re: E.re + wr O.re (- wi O.im later)
                    B_[i] = cwi * O;                           // Im: wi O.re (+ E.im + wr O.im later)
                }
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k1 = decltype(kc)::value;
                    plane[ln * BP + 64 * pr + k1] = x[FFT_POS<k1, BH>].y;
                });
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int k1r = tq & 63, pr = PAR >= 0 ? PAR : tq >> 7;
#pragma unroll
                for (int i = 0; i < BH; ++i) {
                    const float E = plane[(2 * i + pr) * BP + k1r], O = plane[(2 * i + pr) * BP + 64 + k1r];
                    x[i].x = fmaf(-cwi, O, A_[i]);
                    x[i].y = fmaf(cwr, O, E + B_[i]);
                }
            }
        }

        // ---------------- stage 3: forward column codelet; G_par[ky1] at x[FFT_POS<ky1>] (column `line`)
        fft_inreg<BH, 1>(x);

        // ---------------- stage 4: column combine, mirrored bin, cross-spectrum
        //   thread (k, a = par) owns bins ky = ky1 + 64a:  Z = G_0 + s w^ky1 G_1,  s = a ? -1 : +1
        {
            if (par) {       // wave-uniform: the odd-row threads contribute w^ky1 G_1
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int ky1 = decltype(kc)::value;
                    x[FFT_POS<ky1, BH>] = twmul<ky1, BW, 1>(x[FFT_POS<ky1, BH>]);
                });
            }
            float za[BH], zc[BH];                              // Re Z(k), Re Z(-k)
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int ky1 = decltype(kc)::value;
                    plane[(ky1 + 64 * pr) * BP + ln] = x[FFT_POS<ky1, BH>].x;
                });
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                const int mk = (BW - ln) & (BW - 1);           // mirrored column
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int ky1 = decltype(kc)::value;
                    // mirrored row of ky = ky1 + 64 par:  mky = (128 - ky) mod 128 = mky1 + 64 ma  with
                    // mky1 = (64 - ky1) mod 64 for either parity, and  ma = par (ky1 = 0),  1 - par (else)
                    constexpr int mky1 = (64 - ky1) & 63;
                    const float ms = ky1 == 0 ? sgn : -sgn;
                    za[ky1] = fmaf(sgn, plane[(64 + ky1) * BP + ln], plane[ky1 * BP + ln]);
                    zc[ky1] = fmaf(ms, plane[(64 + mky1) * BP + mk], plane[mky1 * BP + mk]);
                });
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int ky1 = decltype(kc)::value;
                    plane[(ky1 + 64 * pr) * BP + ln] = x[FFT_POS<ky1, BH>].y;
                });
            }
            wg_barrier();
            // with Z(k) = a + ib, Z(-k) = c + id:  re = 2 (a d + b c),  im = (c^2 - a^2) + (d^2 - b^2)
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                const int mk = (BW - ln) & (BW - 1);
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int ky1 = decltype(kc)::value;
                    constexpr int mky1 = (64 - ky1) & 63;
                    const float ms = ky1 == 0 ? sgn : -sgn;
                    const float b_ = fmaf(sgn, plane[(64 + ky1) * BP + ln], plane[ky1 * BP + ln]);
                    const float d_ = fmaf(ms, plane[(64 + mky1) * BP + mk], plane[mky1 * BP + mk]);
                    const float a_ = za[ky1], c_ = zc[ky1];
                    x[ky1].x = (a_ * d_ + b_ * c_) * 2.0f;      // P[ky1 + 64 par][k], natural order
                    x[ky1].y = (c_ * c_ - a_ * a_) + (d_ * d_ - b_ * b_);
                });
            }
        }

        // ---------------- stage 5: inverse column transform (DIF split over the thread pair)
        {
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
#pragma unroll
                for (int q = 0; q < BH; ++q) plane[(q + 64 * pr) * BP + ln] = x[q].x;
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
#pragma unroll
                for (int q = 0; q < BH; ++q) x[q].x = fmaf(sgn, x[q].x, plane[(q + 64 * (1 - pr)) * BP + ln]);
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
#pragma unroll
                for (int q = 0; q < BH; ++q) plane[(q + 64 * pr) * BP + ln] = x[q].y;
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
#pragma unroll
                for (int q = 0; q < BH; ++q) x[q].y = fmaf(sgn, x[q].y, plane[(q + 64 * (1 - pr)) * BP + ln]);
            }
            // a = 0: P[ky1] + P[ky1+64];  a = 1: P[ky1] - P[ky1+64] (own is the +64 bin: partner - own), times w^-ky1
            if (par) {
                static_for<0, BH>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int ky1 = decltype(kc)::value;
                    x[ky1] = twmul<ky1, BW, -1>(x[ky1]);
                });
            }
            fft_inreg<BH, -1>(x);                              // Y[2i + par][k] at x[FFT_POS<i>]
        }

        // ---------------- stage 6: transposition + inverse row transform (DIF split over k)
        //   v[k1] = Y[k1] +- Y[k1 + 64] (odd samples: times w^-k1) is Hermitian in k1 because the row is
        //   real, so bins 0..32 suffice and the last transform is a c2r one (32-point complex codelet)
        cf zc[BH / 2];
        {
            cf hs[BH / 2 + 1];
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                static_for<0, BH>([&](auto ic) TPIV_LAMBDA_INLINE {
                    constexpr int i = decltype(ic)::value;
                    plane[(2 * i + pr) * BP + ln] = x[FFT_POS<i, BH>].x;
                });
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127;
#pragma unroll
                for (int q = 0; q <= BH / 2; ++q) hs[q].x = fmaf(sgn, plane[ln * BP + 64 + q], plane[ln * BP + q]);
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127, pr = PAR >= 0 ? PAR : tq >> 7;
                static_for<0, BH>([&](auto ic) TPIV_LAMBDA_INLINE {
                    constexpr int i = decltype(ic)::value;
                    plane[(2 * i + pr) * BP + ln] = x[FFT_POS<i, BH>].y;
                });
            }
            wg_barrier();
            {
                const int tq = TPIV_OPQ_T();
                const int ln = tq & 127;
#pragma unroll
                for (int q = 0; q <= BH / 2; ++q) hs[q].y = fmaf(sgn, plane[ln * BP + 64 + q], plane[ln * BP + q]);
            }
            if (par) {
                static_for<0, BH / 2 + 1>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k1 = decltype(kc)::value;
                    hs[k1] = twmul<k1, BW, -1>(hs[k1]);
                });
            }
            c2r_inreg<BH>(hs, zc);          // corr[line][2 (2m) + par], corr[line][2 (2m+1) + par] = zc[FFT_POS<m, 32>].x, .y
        }

        // ---------------- stage 7: peak analysis on the map in LDS (fftshift coordinates)
        {
            float c[BH];                                       // c[j]: column 2j + par of row `line`
            static_for<0, BH>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                c[j] = (j & 1) ? zc[FFT_POS<j / 2, BH / 2>].y : zc[FFT_POS<j / 2, BH / 2>].x;
            });
            // prefetch: the last iteration re-loads its own window (no branch around the loads)
            big_peak_stage<CAND>(p, c, sm, t, fidx, dead, [&]() TPIV_LAMBDA_INLINE {
                issue_loads(item + per_xcd < hi ? item + per_xcd : item);
            }, __int_as_float(band_s));
        }
    }
}

template <bool CAND>
__device__ __forceinline__ void xcorr_big128_entry(const PassParams& p) {
    __shared__ BigShared sm;
    // (the candidate form only: its 6 spilled registers go down to 2 and the locating pass of configs[4] from 3.76 to 3.60 ms
    //  per 64 pairs; the plain float32 kernel has no spills to lose and its doubled code ran 2 % slower -- same box, A B A B)
    if constexpr (TPIV_BIG_PER_PAR && CAND) {
        if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 128) xcorr_big128_body<CAND, 1>(p, sm);
        else xcorr_big128_body<CAND, 0>(p, sm);
    } else {
        xcorr_big128_body<CAND, -1>(p, sm);
    }
}
__global__ __launch_bounds__(256, 2) void xcorr_big128_kernel(PassParams p) { xcorr_big128_entry<false>(p); }
// float32 first pass of the exact scheme (xcorr_exact.hip): same transforms, candidate cells out
__global__ __launch_bounds__(256, 2) void xcorr_big128_cand_kernel(PassParams p) { xcorr_big128_entry<true>(p); }

// test hook: hand-made maps [n_maps, 128, 128] float32 in fftshift layout through stage 7
__global__ __launch_bounds__(256, 2) void peak_debug_big_kernel(PassParams p, const float* maps, int n_maps) {
    __shared__ BigShared sm;
    const int t = threadIdx.x;
    const int line = t & 127, par = t >> 7;
    const int win = blockIdx.x;
    if (win >= n_maps) return;
    float c[BH];
    const int ys = (line + 64) & 127;
#pragma unroll
    for (int j = 0; j < BH; ++j) c[j] = maps[((size_t)win * BW + ys) * BW + ((2 * j + par + 64) & 127)];
    big_peak_stage(p, c, sm, t, (size_t)win, false, []() TPIV_LAMBDA_INLINE {});
}

inline hipError_t launch_peak_debug_big(const PassParams& p, const float* maps, int n_maps, hipStream_t stream) {
    hipLaunchKernelGGL(peak_debug_big_kernel, dim3(n_maps), dim3(256), 0, stream, p, maps, n_maps);
    return hipGetLastError();
}

inline hipError_t launch_xcorr_big128(const PassParams& p, int n_cu, hipStream_t stream, bool cand = false) {
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    if (items <= 0 || (cand && p.cand == nullptr)) return hipErrorInvalidValue;
    long long blocks = items < (long long)n_cu * 16 ? items : (long long)n_cu * 16;
    blocks = (blocks + 7) / 8 * 8;
    if (cand) hipLaunchKernelGGL(xcorr_big128_cand_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(xcorr_big128_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace tpiv
