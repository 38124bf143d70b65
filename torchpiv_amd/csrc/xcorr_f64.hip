// First pass in float64 (TPIV_PREC_F64 / TPIV_PREC_REFERENCE) for the power-of-two window sizes 8 ... 128.
//
// The reference promotes pass 1 to float64 (PIVbackend.py:513-514: aa = a / mean(a) as float64, then correalte_fft
// B:249-257 runs rfft2 / irfft2 in complex128, `corr - corr.min()` B:518 and correlation_to_displacement B:360-422 on
// the float64 map).  Everything here keeps the lines of the packed tile a + i b IN REGISTERS and uses LDS only to
// transpose between the row and the column transform, one float64 plane at a time:
//
//   8 ... 32    xcorr_f64_tile_kernel    lane = one image row, the whole line (<= 32 complex float64 = 128 VGPRs) per
//                                        lane, single-wavefront workgroups (no barriers), c2r last transform;
//   64, 128     xcorr_f64_split_kernel   a 64- / 128-point line does not fit a lane: two threads per line, 32- / 64-point
//                                        codelets, the radix-2 steps folded into the LDS transposes (per-thread arithmetic
//                                        in xcorr_f64_split.hpp, which the CPU suite runs thread by thread against numpy).
//
// The window normalisation is one factor on the correlation map (rows_forward in xcorr_f64_split.hpp).  Peak analysis
// follows the reference on the float64 map (first flat index on ties, flat-index neighbours and fix-ups, 7x7 flat-index
// exclusion with row wrap and clamps); the 8-double record goes to finalize_kernel<true>.
//
// History: the first generation (round 2) kept the 16-byte complex elements of the tile in LDS and walked radix-8 stages
// over them -- 35.0 ms per 256 pairs for the 64x64 pass 1 of configs[1] at 43 % LDS bank conflicts and ~25 barriers per
// window (profiles/r02), 597 us per 4096^2 pair for the 32x32 pass 1 of configs[3]; same-box A/B against this file:
// 17.2 ms and 196 us.  128x128 windows ran the generic-size DFT kernel: 15.4 ms per pair, now 0.117 ms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "piv_kernels.h"
#include "xcorr_tile.hpp"      // grp_reduce: wavefront reductions in the VALU (DPP / permlane swaps)
#include "xcorr_f64_split.hpp" // 64 / 128: lines split over two threads, 32- / 64-point in-register codelets

#ifndef TPIV_F64_WIDE64
#define TPIV_F64_WIDE64 1      // 64x64: the parity branch spans the column stages through the T2 write (A/B: 0 = only the cross-spectrum)
#endif

namespace tpiv {

namespace {

// Workgroup barrier for LDS exchanges only: __syncthreads() also waits for vmcnt(0), which would drain
// the next window's pixel prefetch at the first barrier of a window.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// =====================================================================================================
// 64x64 and 128x128 windows: every line split over two threads, 32- / 64-point in-register codelets, the radix-2
// steps folded into the LDS transposes (scheme and per-thread arithmetic: xcorr_f64_split.hpp).
// =====================================================================================================
template <int W>
struct __attribute__((aligned(16))) F64SplitShared {
    double plane[f64s::Split<W>::PLANE];                        // T1: one component of the tile, column-major; T2: complex elements
    double zone[f64s::Split<W>::ZR * f64s::Split<W>::ZP];       // raw map rows around the maximum (peak analysis)
    double rmx[2 * W];                                          // every thread's raw row maximum
    double redd[8];                                             // per wavefront: minimum [0..3], maximum [4..7]
    unsigned long long redu[4];                                 // 128x128: per-wavefront window sums
};

// W = 64: 128 threads (two wavefronts: wavefront = half of every line), 39.7 KB, four workgroups per CU, <= 256 VGPRs.
// W = 128: 256 threads (wavefronts 0-1 = first halves of lines 0..63 / 64..127, wavefronts 2-3 = second halves), 143 KB:
//          one workgroup per CU and one wavefront per SIMD with up to 512 registers (64-point codelets hold 256).
// Seven workgroup barriers per window: T1 three (real plane written / read / imaginary plane written), T2 two (plane free /
// written), peak analysis two.
// LIST: the windows are those of PassParams::fb_list (precision "exact": the ones its float32 pass left undecided)
// HALF (round 5): 0 / 1 = the whole window loop instantiated per line half (two straight-line loop bodies
// behind ONE wave-uniform branch at the top of the kernel: no merge point for the register allocator anywhere inside);
// -1 = one loop, the half a run-time (wave-uniform) value with branches around the parity-specific stages (round 4).
// 64x64: 256 VGPRs + 58 spilled (220 B of scratch per lane, 10.7 x the algorithmic HBM traffic) -> 185 VGPRs, ScratchSize 0;
// 16.1 -> 14.7 ms per 256 pairs (same box, A B A B).  128x128 (TPIV_F64_PER_HALF=2): 256 + 96 AGPRs, ScratchSize 0 on
// paper -- and a memory fault on the GPU in its first test (golden g3, 128x128): left on the round-4 form.
#ifndef TPIV_F64_PER_HALF
#define TPIV_F64_PER_HALF 1
#endif
template <int W, bool LIST, int HALF = -1>
__device__ __forceinline__ void xcorr_f64_split_body(const PassParams& p, F64SplitShared<W>& sm) {
    using S = f64s::Split<W>;
    using f64s::dmax2;
    using f64s::dmin2;
    using f64s::peak_shifted;
    constexpr int M = S::M, NDW = S::NDW, NWV = 2 * W / 64;                      // wavefronts per workgroup
    constexpr int LB = W == 64 ? 6 : 7;                                          // log2 W: line = t & (W - 1), half = t >> LB
    constexpr int NJ = W / 64;                                                   // map rows per lane in the row scans of the peak stage
    double* const plane = sm.plane;
    const int tid = threadIdx.x;
    // The thread index at the point of use.  128x128: lane id (two VALU instructions) + the wavefront's base in an SGPR --
    // a copy of threadIdx.x kept across the loop is spilled and re-loaded from scratch in front of every phase there.
    // 64x64: an opaque copy of the register (the recomputation costs more than the re-loads).
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    auto thread_index = [&]() TPIV_LAMBDA_INLINE {
        if constexpr (W == 64) {
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            return t_;
        } else {
            return fresh_lane() | wave_base;
        }
    };
#define TPIV_F64_TID() thread_index()
    const int half = HALF >= 0 ? HALF : wave_base >> LB;      // wave-uniform: 0 = first halves of the lines, 1 = second halves
    const int g = 1 - half;                      // parity of the column bins this thread owns

    const int N = p.n_rows * p.n_cols;
    // (128x128: ONE instance serves both forms, the list decided at run time -- a second instance is a second register
    //  allocation for tools/check_lds_inflight.py to pass, and the first one it got for the list form did not)
    const bool listed = W == 128 ? p.fb_list != nullptr : LIST;
    const long long items = listed ? (long long)*p.fb_count : (long long)p.batch * N;
    const int st = p.ws - p.ov;
    const int HW = p.H * p.W;
    const int wv = p.val_win;
    const int nz = 2 * wv + 3;                   // map rows around the maximum that the record / the exclusion zone can touch
    const bool zone_in_plane = nz > S::ZR;       // (val_win > 4: the zone rows go through the plane, at the price of one more barrier)
    double* const zone = zone_in_plane ? plane : sm.zone;
    // XCD-aware static order: workgroups b, b+8, ... share an XCD and walk one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < items ? lo + chunk : items;

    // both threads of a line load the whole image row; the next window's rows are fetched while the peak analysis of
    // the current one runs
    uint32_t da[NDW], db[NDW];
    auto window_of = [&](long long it) TPIV_LAMBDA_INLINE { return listed ? (long long)p.fb_list[it] : it; };
    auto fetch = [&](long long it_) TPIV_LAMBDA_INLINE {
        const long long it = window_of(it_);
        const int pair_ = (int)(it / N), win_ = (int)(it % N);
        const int yy0 = (win_ / p.n_cols) * st, xx0 = (win_ % p.n_cols) * st;
        const int t_ = W == 64 ? tid : TPIV_F64_TID();
        const size_t off = (size_t)pair_ * HW + (size_t)(yy0 + (t_ & (W - 1))) * p.W + xx0;
        load_dwords<NDW>(p.A + off, da);
        load_dwords<NDW>(p.B + off, db);
    };
    if (lo + slot < hi) fetch(lo + slot);
    TPIV_STAMP_DECL
    TPIV_STAMP_START;
    for (long long item = lo + slot; item < hi; item += per_xcd) {
#ifdef TPIV_STAMPS
        ++st_iter;
#endif
        // (lane-derived values are re-made from the thread index at every phase -- TPIV_F64_TID() -- so that the
        //  loop-invariant LDS addresses are not hoisted out of the item loop into registers)
        // ---- window sums: exact integers, kept in scalar registers until the peak stage.  64x64: every wavefront holds
        //      all 64 rows; 128x128: the rows of wavefronts 0 and 1 together (added behind the first barrier of the peak stage)
        unsigned long long s2;
        {
            unsigned ia = 0, ib = 0;
#pragma unroll
            for (int q = 0; q < NDW; ++q) {
                ia = __builtin_amdgcn_sad_u8(da[q], 0u, ia);
                ib = __builtin_amdgcn_sad_u8(db[q], 0u, ib);
            }
            auto uadd = [](unsigned long long a, unsigned long long b) TPIV_LAMBDA_INLINE { return a + b; };
            s2 = grp_reduce<64>((unsigned long long)ia | ((unsigned long long)ib << 32), uadd);
            s2 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(s2 >> 32)) << 32) |
                 (unsigned)__builtin_amdgcn_readfirstlane((unsigned)s2);
            if constexpr (W == 128) {
                if (fresh_lane() == 0) sm.redu[wave_base >> 6] = s2;
            }
        }

        // ---- R: rows forward
        cd x[M];
        S::rows_forward(da, db, half, x);
        TPIV_STAMP(0);      // window sums, rows forward

        // ---- T1 + C: transposition with the DIF step of the column transform, columns forward
        cd u[M];
        if (zone_in_plane) lds_barrier();            // (the previous window's zone reads)
        {
            const int y_ = TPIV_F64_TID() & (W - 1);
            if (half) S::template t1_write<0, 1>(x, y_, plane);
            else S::template t1_write<0, 0>(x, y_, plane);
        }
        lds_barrier();
        S::template t1_read<0>(u, TPIV_F64_TID() & (W - 1), g, plane);
        lds_barrier();
        {
            const int y_ = TPIV_F64_TID() & (W - 1);
            if (half) S::template t1_write<1, 1>(x, y_, plane);
            else S::template t1_write<1, 0>(x, y_, plane);
        }
        lds_barrier();
        S::template t1_read<1>(u, TPIV_F64_TID() & (W - 1), g, plane);
        TPIV_STAMP(1);      // transposition 1 (3 barriers)
        // ---- C, X, Ci: the column stages, one straight-line instance per parity (wave-uniform branch: the parity-0
        //      instance carries no twiddle products, and nothing is common to the two instances that the compiler could hoist
        //      in front of the branch and keep in registers)
        // ---- C, X, Ci and the T2 write: one straight-line instance per parity behind a wave-uniform branch that ends
        //      only where nothing of it is live any more (its results sit in LDS): a merge in between would have to bring
        //      the 128 registers of the spectrum together from both sides, and the register allocator answers that with a
        //      second set of registers and scratch traffic.  (Every wavefront passes exactly one barrier inside.)
        auto column_stages = [&](auto gc) TPIV_LAMBDA_INLINE {
            constexpr int G = decltype(gc)::value;
            S::cols_forward(u, G);
            __builtin_amdgcn_sched_barrier(0);
            TPIV_STAMP(2);      // columns forward
            // cross-spectrum; the mirrored column sits in the neighbouring lane (lanes 0, 1 of the line order: their own)
            {
                const bool own = (TPIV_F64_TID() & (W - 1)) < 2;
                auto sh = [&](double v, int, int) TPIV_LAMBDA_INLINE {
                    const double r = f64s::dpp_xor1(v);
                    return own ? v : r;
                };
                S::template cross_spectrum_g<G>(u, u, sh);
            }
            __builtin_amdgcn_sched_barrier(0);
            TPIV_STAMP(3);      // cross-spectrum incl. the exchange
            // columns inverse, transposition with the DIT step
            cd t[M];
            S::cols_inverse(u, G, t);
            __builtin_amdgcn_sched_barrier(0);
            TPIV_STAMP(4);      // columns inverse
            lds_barrier();                               // every thread has read its T1 column
            S::t2_write(t, TPIV_F64_TID() & (W - 1), G, plane);
        };
        if constexpr (W == 64 && TPIV_F64_WIDE64) {
            if (g) column_stages(std::integral_constant<int, 1>{});
            else column_stages(std::integral_constant<int, 0>{});
        } else {
            // 128x128 (one wavefront per SIMD, 512 registers): the branch only around the cross-spectrum, transforms shared
            S::cols_forward(u, g);
            {
                const bool own = (TPIV_F64_TID() & (W - 1)) < 2;
                auto sh = [&](double v, int, int) TPIV_LAMBDA_INLINE {
                    const double r = f64s::dpp_xor1(v);
                    return own ? v : r;
                };
                if (g) S::template cross_spectrum_g<1>(u, u, sh);
                else S::template cross_spectrum_g<0>(u, u, sh);
            }
            cd t[M];
            S::cols_inverse(u, g, t);
            lds_barrier();
            S::t2_write(t, TPIV_F64_TID() & (W - 1), g, plane);
        }
        cd Y[M + 1];
        lds_barrier();
        S::t2_read(Y, TPIV_F64_TID() & (W - 1), plane);
        TPIV_STAMP(5);      // transposition 2 (2 barriers)

        // ---- Ri: rows inverse (c2r over the thread pair)
        double c[M];
        S::rows_inverse(Y, half, c);
        TPIV_STAMP(6);      // rows inverse

        // ---- P: peak analysis on the raw float64 cells (xcorr_f64_split.hpp, P).  Two exchanges, ONE barrier each.
        auto dmin = [](double a, double b) TPIV_LAMBDA_INLINE { return dmin2(a, b); };
        auto dmax = [](double a, double b) TPIV_LAMBDA_INLINE { return dmax2(a, b); };
        auto imin = [](int a, int b) TPIV_LAMBDA_INLINE { return a < b ? a : b; };
        double cmin, graw;
        {
            double mn_, mx_;
            S::peak_local_minmax(c, mn_, mx_);
            // prefetch: the last iteration re-loads its own window (no branch around the loads)
            fetch(item + per_xcd < hi ? item + per_xcd : item);
            const int t_ = TPIV_F64_TID();
            sm.rmx[t_] = mx_;
            mn_ = grp_reduce<64>(mn_, dmin);
            mx_ = grp_reduce<64>(mx_, dmax);
            if ((t_ & 63) == 0) {
                sm.redd[t_ >> 6] = mn_;
                sm.redd[4 + (t_ >> 6)] = mx_;
            }
            lds_barrier();                            // (also: every thread has read its T2 row -> the plane is free)
            cmin = sm.redd[0];
            graw = sm.redd[4];
#pragma unroll
            for (int w_ = 1; w_ < NWV; ++w_) {
                cmin = dmin2(cmin, sm.redd[w_]);
                graw = dmax2(graw, sm.redd[4 + w_]);
            }
        }
        if constexpr (W == 128) s2 = sm.redu[0] + sm.redu[1];
        const unsigned ia = (unsigned)s2, ib = (unsigned)(s2 >> 32);
        const bool dead = ia == 0u || ib == 0u;          // zero-mean window: 0/0 = NaN map in the reference
        // a / mean(a), b / mean(b) (B:513-514) as ONE factor on the correlation map (xcorr_f64_split.hpp, rows_forward):
        // n^2 / sum(a) * n^2 / sum(b), times 1/n^2 of the inverse transform and the 1/4 of the cross-spectrum algebra;
        // sum(a) sum(b) < 2^44 is exact, so this is one correctly rounded division per window
        const double map_scale = dead ? 0.0 : ((double)(W * W) * 0.25) / ((double)ia * (double)ib);
        // the maximum of the shifted map is the shifted raw maximum (monotonic, same roundings)
        const double gmax = peak_shifted(graw, cmin, map_scale);
        // arg-max = FIRST flat index holding the maximum (B:383): the smallest shifted row whose maximum is the global
        // one, then the first column of that row.  Every wavefront looks at all rows (lane l: rows l, l + 64, ...; both
        // halves of a row) -- same result in every wavefront, no exchange.
        double rm[NJ];
        int ywin = W - 1;
        {
            const int l_ = TPIV_F64_TID() & 63;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                rm[j] = dmax2(sm.rmx[64 * j + l_], sm.rmx[W + 64 * j + l_]);
                const int fy = S::frow(64 * j + l_);
                ywin = ((int)(peak_shifted(rm[j], cmin, map_scale) == gmax) & (int)(fy < ywin)) ? fy : ywin;
            }
            ywin = grp_reduce<64>(ywin, imin);
            ywin = __builtin_amdgcn_readfirstlane(ywin);
        }
        const int zlo = ywin - wv - 1;
        {
            const int t_ = TPIV_F64_TID();
            S::peak_zone_write(c, t_ & (W - 1), t_ >> LB, zlo, nz, zone);
        }
        lds_barrier();                                // zone rows complete
        int xwin = W - 1;
        {
            const int l_ = TPIV_F64_TID() & 63;
            const double* zrow = zone + (wv + 1) * S::ZP;
#pragma unroll
            for (int part = NJ - 1; part >= 0; --part) {
                const unsigned long long hit = __ballot(peak_shifted(zrow[64 * part + l_], cmin, map_scale) == gmax);
                xwin = hit ? 64 * part + (int)__builtin_ctzll(hit) : xwin;
            }
        }
        const int m = ywin * W + xwin;
        // second peak (B:346-358): the largest cell outside the exclusion zone = max(row maxima of the rows outside the
        // zone rows, cells of the zone rows that are not excluded); every wavefront scans all of it (9 cells per lane)
        double sv = f64s::PEAK_NONE;
        {
            const int l_ = TPIV_F64_TID() & 63;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int zr = S::frow(64 * j + l_) - zlo;
                sv = dmax2(sv, (unsigned)zr >= (unsigned)nz ? rm[j] : f64s::PEAK_NONE);
            }
            // straight-line code: the cells of the ZR buffer rows are loaded unconditionally (compile-time offsets; rows
            // beyond nz hold stale cells and are masked out).  The test "cell (fy, fx) = m + i + W j with |i|, |j| <= wv, or one
            // of the two clamps" (S::peak_excluded, which the CPU suite checks against the oracle) is taken apart: the column
            // part depends on the lane only -- computed once per 64-column piece of a row --, the row part on the wave-uniform
            // row and the lane's wrap adjustment.
            const int mx = xwin;
            const int nk = nz * NJ;                               // 64-cell pieces of the zone rows
            constexpr int KU = S::ZR * NJ;
            double raw[KU];
            static_for<0, KU>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                raw[k] = zone[(k / NJ) * S::ZP + 64 * (k % NJ) + l_];
            });
            int q[NJ];
            bool colok[NJ];
#pragma unroll
            for (int c_ = 0; c_ < NJ; ++c_) colok[c_] = S::peak_col_ok(64 * c_ + l_, mx, wv, q[c_]);
            const bool clamp_lo = m - wv - wv * W <= 0, clamp_hi = m + wv + wv * W >= W * W - 1;      // wave-uniform
            static_for<0, KU>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                constexpr int zr = k / NJ, c_ = k % NJ;
                const int fy = zlo + zr;                                                          // wave-uniform
                const bool row_ok = (k < nk) & ((unsigned)fy < (unsigned)W);                      // wave-uniform
                const bool ex = S::peak_excluded_zr(colok[c_], q[c_], zr, fy, 64 * c_ + l_, wv, clamp_lo, clamp_hi);
                sv = dmax2(sv, (row_ok & !ex) ? raw[k] : f64s::PEAK_NONE);
            });
            for (int k = KU; k < nk; ++k) {                        // (val_win > 4 only: the zone rows sit in the plane)
                const int fy = zlo + k / NJ, fx = 64 * (k % NJ) + l_;
                const int use = (int)((unsigned)fy < (unsigned)W) & (S::peak_excluded(fy * W + fx, m, wv) ^ 1);
                sv = dmax2(sv, use ? zone[(k / NJ) * S::ZP + fx] : f64s::PEAK_NONE);
            }
            sv = grp_reduce<64>(sv, dmax);
        }
        {
            const int t_ = W == 64 ? tid : TPIV_F64_TID();
            if (t_ < 8)
                reinterpret_cast<double*>(p.peak_raw)[(size_t)window_of(item) * 8 + t_] =
                    S::peak_record_slot(t_, m, sv, dead, zone, zlo, cmin, map_scale);
        }
        TPIV_STAMP(7);      // peak analysis incl. the issue of the next window's loads
#undef TPIV_F64_TID
    }
    TPIV_STAMP_FLUSH(p);
}

template <int W, bool LIST>
__device__ __forceinline__ void xcorr_f64_split_entry(const PassParams& p) {
    __shared__ F64SplitShared<W> sm;
    if constexpr (TPIV_F64_PER_HALF == 2 || (TPIV_F64_PER_HALF && W == 64)) {
        if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= W) xcorr_f64_split_body<W, LIST, 1>(p, sm);
        else xcorr_f64_split_body<W, LIST, 0>(p, sm);
    } else {
        xcorr_f64_split_body<W, LIST, -1>(p, sm);
    }
}
template <int W>
__global__ __launch_bounds__(2 * W, W == 64 ? 2 : 1) void xcorr_f64_split_kernel(PassParams p) {
    xcorr_f64_split_entry<W, false>(p);
}
template <int W>
__global__ __launch_bounds__(2 * W, W == 64 ? 2 : 1) void xcorr_f64_list_kernel(PassParams p) {
    xcorr_f64_split_entry<W, true>(p);
}

template <int W>
static hipError_t launch_f64_split(const PassParams& p, int n_cu, hipStream_t stream) {
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    if (items <= 0) return hipErrorInvalidValue;
    // LDS: 39.7 KB / 143 KB per workgroup  (TPIV_F64_PER_CU: A/B runs of the residency)
    static const int per_cu_env = [] { const char* e_ = getenv("TPIV_F64_PER_CU"); return e_ ? atoi(e_) : 0; }();
    const int per_cu = W == 64 ? (per_cu_env > 0 ? per_cu_env : 4) : 1;
    long long blocks = items < (long long)n_cu * per_cu ? items : (long long)n_cu * per_cu;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((xcorr_f64_split_kernel<W>), dim3((unsigned)blocks), dim3(2 * W), 0, stream, p);
    return hipGetLastError();
}

// =====================================================================================================
// 8 ... 32 pixel windows: lane = one image row of a window, the whole line (WS complex float64
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

// LIST: the windows are those of PassParams::fb_list, WPW consecutive entries per item (precision "exact")
template <int WS, bool LIST>
__device__ __forceinline__ void xcorr_f64_tile_body(const PassParams& p) {
    using G = F64TileGeo<WS>;
    constexpr int P = G::P, M = G::M, NDW = G::NDW, WPW = G::WPW;
    __shared__ double tile[WPW * G::PLANE];

    const int N = p.n_rows * p.n_cols;
    const long long listed = LIST ? (long long)*p.fb_count : 0;
    const int groups = (N + WPW - 1) / WPW;
    const long long items = LIST ? (listed + WPW - 1) / WPW : (long long)p.batch * groups;
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
        if constexpr (LIST) {
            const long long li = it * WPW + w_;
            active = li < listed;
            const int e_ = p.fb_list[active ? li : listed - 1];
            pair_ = e_ / N;
            win_ = e_ % N;
        } else {
            pair_ = (int)(it / groups);
            const int raw = (int)(it % groups) * WPW + w_;
            active = raw < N;
            win_ = active ? raw : N - 1;
        }
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
            const unsigned wr_base = f64s::lds_addr(pl + r * P);      // explicit ds_write_b64: merged write2 pairs are slower
            wave_sync();
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                f64s::lds_wr<8 * k>(wr_base, x[FFT_POS<k, WS>].x);
            });
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double xr[WS];
            lds_read_seq<WS, P * 8>(rd_base, [&](auto ic, double v) TPIV_LAMBDA_INLINE { xr[decltype(ic)::value] = v; });
            wave_sync();
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                f64s::lds_wr<8 * k>(wr_base, x[FFT_POS<k, WS>].y);
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
            const unsigned wr_base2 = f64s::lds_addr(pl + r * P);
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
                    f64s::lds_wr<8 * k>(wr_base2, b[k].x);                            // row kx, column y
                });
            }
            wave_sync();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_read_seq<M + 1, P * 8>(rd_base, [&](auto ic, double v) TPIV_LAMBDA_INLINE { hs[decltype(ic)::value].x = v; });
            wave_sync();
            if (r <= M) {
                static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    f64s::lds_wr<8 * k>(wr_base2, b[k].y);
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
                f64s::lds_wr<8 * xs>(f64s::lds_addr(pl + ys * P), v);
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
__global__ __launch_bounds__(64, 2) void xcorr_f64_tile_kernel(PassParams p) {
    xcorr_f64_tile_body<WS, false>(p);
}
template <int WS>
__global__ __launch_bounds__(64, 2) void xcorr_f64_tile_list_kernel(PassParams p) {
    xcorr_f64_tile_body<WS, true>(p);
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

}  // namespace

// the float64 transform for the windows of PassParams::fb_list (their number is known on the device only: the grid is the
// resident set, workgroups without a window leave at once)
hipError_t launch_xcorr_f64_list(const PassParams& p, int n_cu, hipStream_t stream) {
    if (p.fb_list == nullptr || p.fb_count == nullptr) return hipErrorInvalidValue;
    const long long items = (long long)p.batch * p.n_rows * p.n_cols;
    const int per_cu = p.ws <= 32 ? 8 : (p.ws == 64 ? 4 : 1);
    long long blocks = items < (long long)n_cu * per_cu ? items : (long long)n_cu * per_cu;
    blocks = (blocks + 7) / 8 * 8;
    switch (p.ws) {
        case 8: hipLaunchKernelGGL((xcorr_f64_tile_list_kernel<8>), dim3((unsigned)blocks), dim3(64), 0, stream, p); break;
        case 16: hipLaunchKernelGGL((xcorr_f64_tile_list_kernel<16>), dim3((unsigned)blocks), dim3(64), 0, stream, p); break;
        case 32: hipLaunchKernelGGL((xcorr_f64_tile_list_kernel<32>), dim3((unsigned)blocks), dim3(64), 0, stream, p); break;
        case 64: hipLaunchKernelGGL((xcorr_f64_list_kernel<64>), dim3((unsigned)blocks), dim3(128), 0, stream, p); break;
        case 128: hipLaunchKernelGGL((xcorr_f64_split_kernel<128>), dim3((unsigned)blocks), dim3(256), 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_xcorr_f64(const PassParams& p, int n_cu, hipStream_t stream) {
    switch (p.ws) {
        case 8: return launch_f64_tile<8>(p, n_cu, stream);
        case 16: return launch_f64_tile<16>(p, n_cu, stream);
        case 32: return launch_f64_tile<32>(p, n_cu, stream);
        case 64: return launch_f64_split<64>(p, n_cu, stream);
        case 128: return launch_f64_split<128>(p, n_cu, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace tpiv
