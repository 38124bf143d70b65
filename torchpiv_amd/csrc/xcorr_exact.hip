// Exact first pass for every even window size 8 ... 128 (precision "exact"): the values behind the sub-pixel fit and the
// validity ratio as EXACT integer correlation sums of the uint8 windows.
//
// The reference (PIVbackend.py:513-518) divides both windows by their means in float64, correlates them through a
// float64 FFT, subtracts the map minimum and adds 1e-7; every cell of that map is
//     (S(d) - S_min) n^4 / (sum a  sum b) + 1e-7,      S(d) = sum_p a[p] b[(p + d) mod W]   (integers < 2^30)
// up to the rounding of the transform.  Only a handful of cells per window ever reach the result (B:383-411): the
// arg-max, its four flat-index neighbours, the second peak, and the minimum.  So:
//   1. xcorr_tile_cand_kernel<8 | 16 | 32 | 64> (xcorr_tile.hpp, peak_candidates) / xcorr_big128_cand_kernel (xcorr_big.hpp) /
//      the CAND forms of the generic-size kernels (xcorr_generic.hip, map_candidates): the float32 FFT pass LOCATES those
//      cells, with a PROVEN error band around every decision (piv_kernels.h, "The band");
//   2. xcorr_exact_refine_kernel<W> (here; W = 16, 32, 64, 128): the lanes of a window (16: a quarter of a wavefront, 32: half,
//      64: one, 128: two) evaluate S at the located cells -- lane = window row,
//      frame-a row in registers, frame b's rows parked twice over in LDS so that a row rotated by dx is one contiguous
//      span; W / 4 v_dot4_u32_u8 per lane and cell --, re-checks the decisions on the exact values (no neighbour or second
//      candidate above the arg-max, no evaluated cell below the minimum) and writes the 8-double record of
//      finalize_kernel<true>.  Undecided windows are appended to a list.
//      xcorr_exact_refine_any_kernel<G> (here): the same for any even W <= 128 (8, the generic sizes 12 ... 126) with lane =
//      CELL: both windows in LDS, the lane of a cell walks all rows of its rotation;
//   3. xcorr_f64_list_kernel<W> / xcorr_f64_tile_list_kernel<W> (xcorr_f64.hip) / the float64 generic kernel run the float64
//      transform for the listed windows only.
// tests/test_exact_scheme.py is the numpy statement of the scheme (checked against the oracle: 7e-15 px);
// tests/test_gpu_exact.py compares this file with it and with the float64 kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "piv_kernels.h"
#include "xcorr_tile.hpp"      // grp_reduce, wave_sync, load_dwords

namespace tpiv {

namespace {

template <int W>
struct XGeo {
    static constexpr int NDW = W / 4;                        // dwords per window row
    static constexpr int XP = 2 * NDW + 1;                   // dwords per parked row: the row twice over + 1 (odd pitch: the lane = row reads hit all banks)
    static constexpr int GROUP = W < 64 ? W : 64;            // lanes of one window inside a wavefront
    static constexpr int WPW = 64 / GROUP;                   // windows per wavefront (32x32: two)
    static constexpr int PARTS = W / GROUP;                  // wavefronts per window (128x128: two)
    static constexpr int WAVES = W == 128 ? 2 : 4;           // wavefronts per workgroup
    static constexpr int WINS = WAVES * WPW / PARTS;         // windows per workgroup: 8 / 4 / 1
    static constexpr int KD = W * W;
};
constexpr int XCELLS = 5 + EXACT_MAX_SECOND + EXACT_MAX_MIN;      // cells a window can ask for

// N consecutive dwords of a parked row as N single ds_read_b32: left to itself the backend pairs them into ds_read2_b32,
// which deliver a fifth of the bytes per clock (tools/micro/lds_pair.hip, DESIGN.md 9 "Paired LDS accesses") -- and this
// kernel is nothing but LDS reads and dot products.  Hand-issued reads become usable behind the wait only; the asm
// statements are volatile and tie every destination to the wait (tools/check_lds_inflight.py walks this unit too).
#if defined(__HIP_DEVICE_COMPILE__)
template <int OFF>
__device__ __forceinline__ uint32_t lds_rd32(unsigned addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_landed() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_tie(uint32_t& v) { asm volatile("" : "+v"(v)); }
#else       // host pass of the translation unit: the kernels' bodies are parsed, never run
template <int OFF>
__host__ __device__ inline uint32_t lds_rd32(unsigned) { return 0u; }
__host__ __device__ inline void lds_landed() {}
__host__ __device__ inline void lds_tie(uint32_t&) {}
#endif
template <int N>
__device__ __forceinline__ void read_span(const uint32_t* src, uint32_t (&w)[N]) {
    const unsigned addr = (unsigned)(uintptr_t)src;
    static_for<0, N>([&](auto ic) TPIV_LAMBDA_INLINE { w[decltype(ic)::value] = lds_rd32<4 * decltype(ic)::value>(addr); });
    lds_landed();
    static_for<0, N>([&](auto ic) TPIV_LAMBDA_INLINE { lds_tie(w[decltype(ic)::value]); });
}

// ---- the decisions, re-checked on the exact values, and the record.  Called by the 16 first lanes of a window (all of
// them: the reductions are row-wide DPP steps): lane r0 < XCELLS holds S of cell r0 (`have`: the cell exists), every lane
// the window sums.  (No contrast guard any more: round 4 sent low-contrast windows to the float64 transform because its band
// was a measured constant times the map range; the band of the locating passes is now the proven bound on their float32
// error, piv_kernels.h "The band", whatever the contrast.)
__device__ __forceinline__ void decide_and_store(const PassParams& p, const uint4& rec, int r0, bool go, bool writer, bool have,
                                                 int m, unsigned S, unsigned sa, unsigned sb, double kd, double* out, long long it) {
    auto hi16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v >> 16); };
    auto umin = [](unsigned x, unsigned y) TPIV_LAMBDA_INLINE { return x < y ? x : y; };
    auto umax = [](unsigned x, unsigned y) TPIV_LAMBDA_INLINE { return x > y ? x : y; };
    auto uadd = [](unsigned x, unsigned y) TPIV_LAMBDA_INLINE { return x + y; };
    constexpr int S0 = 5, N0 = 5 + EXACT_MAX_SECOND;         // first second-peak / minimum slot
    S = r0 < XCELLS ? S : 0u;
    // (the cells sit in lanes 0..11 of the window and the results are used by lanes 0..7: reductions over its first 16 lanes)
    const unsigned s_min = grp_reduce<16>((have && r0 >= N0 && r0 < XCELLS) ? S : 0xffffffffu, umin);
    const unsigned s_low = grp_reduce<16>((have && r0 < XCELLS) ? S : 0xffffffffu, umin);
    const unsigned s_top = grp_reduce<16>((have && r0 < N0) ? S : 0u, umax);
    const unsigned s_m = grp_reduce<16>(r0 == 0 ? S : 0u, umax);
    const unsigned s_second = grp_reduce<16>((have && r0 >= S0 && r0 < N0) ? S : 0u, umax);
    const unsigned n_second = grp_reduce<16>((have && r0 >= S0 && r0 < N0) ? 1u : 0u, uadd);
    // (the first two cannot happen while the float32 map stays inside the band; sa, sb: the locating pass marks those)
    // more minimum candidates than the record holds (-2 in its last slot): S >= 0 everywhere, so the evaluated ones settle
    // it exactly when one of them is 0 (true-zero backgrounds); anything else goes to the float64 transform
    const bool min_overflow = hi16(rec.w) == -2;
    const bool redo = s_top > s_m || s_low < s_min || s_min == 0xffffffffu || sa == 0u || sb == 0u ||
                      (min_overflow && s_min != 0u);
    if (go && redo && r0 == 0 && writer) p.fb_list[atomicAdd(p.fb_count, 1u)] = (int)it;
    // (S - S_min) n^4 / (sum a sum b) + 1e-7: the integer difference is exact, sum a * sum b < 2^44 is exact
    const double scale = (kd * kd) / ((double)sa * (double)sb);
    const unsigned mine = r0 == 5 ? s_second : S;
    double v = __fma_rn((double)(mine - s_min), scale, 1e-7);
    // B:410-411 with every cell inside the exclusion zone (2 val_win + 1 >= W): the reference zeroes its float64 map in
    // place and divides by the 0 it then finds -- ratio inf, valid; the float64 kernels store 0 there too
    v = (r0 == 5 && n_second == 0u) ? 0.0 : v;
    v = r0 == 6 ? (double)m : v;
    v = r0 == 7 ? 0.0 : v;
    if (go && !redo && r0 < 8 && writer) out[r0] = v;
}

// Everything below is per-lane data with predicated control flow: for 32x32 the two windows of a wavefront take their
// decisions independently.  (64x64: one window per wavefront; 128x128: one window per workgroup of two wavefronts, partial
// sums joined through LDS behind ONE barrier that both wavefronts reach or neither does.)
template <int W>
__global__ __launch_bounds__(64 * XGeo<W>::WAVES) void xcorr_exact_refine_kernel(PassParams p) {
    using G = XGeo<W>;
    constexpr int NDW = G::NDW, XP = G::XP, KD = G::KD;
    __shared__ uint32_t parked[G::WINS][W * XP];
    __shared__ unsigned joined[2][16];                       // 128x128: per wavefront the partial S of 12 cells + 4 window sums
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int g = lane / G::GROUP, r0 = lane % G::GROUP;     // window inside the wavefront, lane inside the window
    const int part = G::PARTS == 2 ? wave : 0;
    const int row = r0 + 64 * part;                          // window row of this lane
    const int wslot = G::PARTS == 2 ? 0 : wave * G::WPW + g;
    uint32_t* const rows_b = parked[wslot];
    const int N = p.n_rows * p.n_cols;
    const long long total = (long long)p.batch * N;
    // PERSISTENT (round 5): the grid is the resident set; workgroups b, b+8, ... share an XCD (and its L2) and walk one
    // contiguous run of windows together, G::WINS windows per step.  The rows and the candidate record of the NEXT step are
    // requested before the current one is evaluated: with one window per wavefront LIFETIME (round 4) the kernel was a chain
    // of memory latencies and workgroup launches -- 1.26 ms per 1 016 064 windows for 0.35 ms worth of instruction issue.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (int)(gridDim.x >> 3);
    const long long chunk = (total + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < total ? lo + chunk : total;
    const long long stride = (long long)per_xcd * G::WINS;
    const int st = p.ws - p.ov;
    auto lo16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v & 0xffffu); };
    auto hi16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v >> 16); };
    // window `itr` of this lane (clamped into the run: a slot past its end re-reads the last window and stores nothing)
    auto fetch = [&](long long itr, uint32_t (&a_)[NDW], uint32_t (&b_)[NDW], uint4& rec_) TPIV_LAMBDA_INLINE {
        const unsigned itu = (unsigned)(itr < hi ? itr : hi - 1);        // (total < 2^31: 32-bit divisions)
        const int pair = (int)(itu / (unsigned)N), win = (int)(itu - (unsigned)pair * (unsigned)N);
        const size_t off = (size_t)pair * p.H * p.W + (size_t)((win / p.n_cols) * st + row) * p.W + (size_t)(win % p.n_cols) * st;
        load_dwords<NDW>(p.A + off, a_);
        load_dwords<NDW>(p.B + off, b_);
        rec_ = p.cand[itu];
    };
    long long base = lo + (long long)slot * G::WINS;         // first window of the workgroup's step
    if (base >= hi) return;
    uint32_t a[NDW], b[NDW];
    uint4 rec;
    fetch(base + wslot, a, b, rec);
    for (; base < hi; base += stride) {
    const long long it_raw = base + wslot;
    const bool valid = it_raw < hi;
    const long long it = valid ? it_raw : hi - 1;
    // ---- the next step's rows and record: in flight while this step is evaluated
    uint32_t na[NDW], nb[NDW];
    uint4 nrec;
    fetch(base + stride + wslot, na, nb, nrec);
    auto advance = [&]() TPIV_LAMBDA_INLINE {
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            a[i] = na[i];
            b[i] = nb[i];
        }
        rec = nrec;
    };
    const int m = valid ? lo16(rec.x) : -3;
    double* const out = reinterpret_cast<double*>(p.peak_raw) + (size_t)it * 8;
    const bool writer = part == 0;                           // the wavefront of a window that stores its results
    auto to_f64_kernel = [&]() TPIV_LAMBDA_INLINE {
        if (r0 == 0 && writer) p.fb_list[atomicAdd(p.fb_count, 1u)] = (int)it;
    };
    if (m == -1) to_f64_kernel();
    if (m == -2 && r0 < 8 && writer) out[r0] = r0 == 6 ? 0.0 : 1.0;      // zero-mean window (B:513: NaN map): finalize_kernel looks at the flag [7] only
    const bool go = m >= 0;
    if (__ballot(go) == 0ull) {                              // (128x128: the same decision in both wavefronts of the window)
        advance();
        continue;
    }

    // ---- the cells: lane j of the window holds flat index q_j (fftshift layout) or -1
    int q = -1;
    if (go) {
        int left = m + 1, right = m - 1, top = m + W, bot = m - W;      // B:385-392
        if (left >= KD - 1) left = m;
        if (right <= 0) right = m;
        if (top >= KD - 1) top = m;
        if (bot <= 0) bot = m;
        q = r0 == 0 ? m : q;
        q = r0 == 1 ? left : q;
        q = r0 == 2 ? right : q;
        q = r0 == 3 ? top : q;
        q = r0 == 4 ? bot : q;
        q = r0 == 5 ? hi16(rec.x) : q;
        q = r0 == 6 ? lo16(rec.y) : q;
        q = r0 == 7 ? hi16(rec.y) : q;
        q = r0 == 8 ? lo16(rec.z) : q;
        q = r0 == 9 ? hi16(rec.z) : q;
        q = r0 == 10 ? lo16(rec.w) : q;
        q = r0 == 11 ? hi16(rec.w) : q;
    }

    // per-lane partial sums P[0..11]: S at the requested cells, P[12..13]: window sums (< 2^22 per window)
    unsigned P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) P[k] = 0u;
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        P[12] = __builtin_amdgcn_sad_u8(a[i], 0u, P[12]);
        P[13] = __builtin_amdgcn_sad_u8(b[i], 0u, P[13]);
        rows_b[row * XP + i] = b[i];
        rows_b[row * XP + NDW + i] = b[i];
    }
    if constexpr (G::PARTS == 2) __syncthreads();       // the other wavefront's rows are parked
    else wave_sync();

    // ---- S at every requested cell
    static_for<0, XCELLS>([&](auto cc) TPIV_LAMBDA_INLINE {
        constexpr int c = decltype(cc)::value;
        int qc = __builtin_amdgcn_readlane(q, c);
        bool any = qc >= 0;
        static_for<1, G::WPW>([&](auto wc) TPIV_LAMBDA_INLINE {          // (the other windows of the wavefront)
            constexpr int k = decltype(wc)::value;
            const int qk = __builtin_amdgcn_readlane(q, k * G::GROUP + c);
            any = any || qk >= 0;
            qc = g == k ? qk : qc;
        });
        if (any) {                                      // (wave-uniform)
            const bool on = qc >= 0;
            const int qq = on ? qc : 0;
            const int dy = qq / W - W / 2, dx = qq % W - W / 2;
            const int brow = (row + dy) & (W - 1), bx = dx & (W - 1);
            // (aligned dwords + v_alignbyte: reading the rotated row at its byte address -- unaligned ds_read_b32, no alignbyte --
            //  is correct on this chip and seven times slower: 9.4 instead of 1.28 ms per 1 016 064 windows)
            const uint32_t* src = rows_b + brow * XP + (bx >> 2);
            const unsigned sh = (unsigned)(bx & 3);
            uint32_t w[NDW + 1];
            read_span<NDW + 1>(src, w);
            unsigned acc = 0;
#pragma unroll
            for (int i = 0; i < NDW; ++i) acc = __builtin_amdgcn_udot4(a[i], __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh), acc, false);
            P[c] = on ? acc : 0u;
        }
    });
    // ---- ONE transposing reduction of the 16 partials over the window's lanes (instead of 16 butterflies): two halving
    //      steps inside the quads -- lane l keeps the values 4j + (l & 3) --, then plain sums over the quads of a row
    //      (row rotations by 8 and 4 keep l & 3), over the rows (permlane swaps)
    unsigned T[4];
    {
        const bool odd = lane & 1, hi = lane & 2;
        unsigned Q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned keep = odd ? P[2 * j + 1] : P[2 * j], send = odd ? P[2 * j] : P[2 * j + 1];
            Q[j] = keep + (unsigned)dpp_i<DPP_XOR1>((int)send);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned keep = hi ? Q[2 * j + 1] : Q[2 * j], send = hi ? Q[2 * j] : Q[2 * j + 1];
            unsigned t = keep + (unsigned)dpp_i<DPP_XOR2>((int)send);
            t += (unsigned)dpp_i<0x128>((int)t);        // row_ror:8
            t += (unsigned)dpp_i<0x124>((int)t);        // row_ror:4
            if constexpr (G::GROUP >= 32) {
                auto r = __builtin_amdgcn_permlane16_swap(t, t, false, false);
                t = r[0] + r[1];
            }
            if constexpr (G::GROUP >= 64) {
                auto r = __builtin_amdgcn_permlane32_swap(t, t, false, false);
                t = r[0] + r[1];
            }
            T[j] = t;
        }
    }
    // lane j < 12 of the window: S(q_j) = value j = T[j >> 2]; lanes 12..15: the four window sums = T[3]
    unsigned S = r0 < 4 ? T[0] : (r0 < 8 ? T[1] : (r0 < 12 ? T[2] : T[3]));
    if constexpr (G::PARTS == 2) {
        // join the two halves of the window: lanes 0..11 carry the partial S, lanes 12..15 the partial window sums
        if (lane < 16) joined[wave][lane] = S;
        __syncthreads();
        S = lane < 16 ? joined[0][lane] + joined[1][lane] : 0u;
        T[3] = S;                                       // (lanes 12..15; the quad broadcasts below read exactly those)
    }
    // the window sums to every lane of their quad (lanes 12..15 hold them; every lane r0 < 16 needs them: same row)
    unsigned sa, sb;
    {
        // row_bcast within the row of 16 lanes: lane 12 / 13 / 14 / 15 of the row through a row rotation + quad broadcast
        const unsigned t3 = (r0 & 12) == 12 ? T[3] : 0u;                  // only the last quad of the first row carries sums
        unsigned rowsum = t3;                                             // value at lane 12 + (l & 3), spread over the row:
        rowsum += (unsigned)dpp_i<0x128>((int)rowsum);                    // row_ror:8
        rowsum += (unsigned)dpp_i<0x124>((int)rowsum);                    // row_ror:4  -> every quad of the row holds the four sums
        sa = (unsigned)dpp_i<0x00>((int)rowsum);                          // quad_perm [0,0,0,0]
        sb = (unsigned)dpp_i<0x55>((int)rowsum);                          // quad_perm [1,1,1,1]
    }
    S = r0 < XCELLS ? S : 0u;

    decide_and_store(p, rec, r0, go, writer, q >= 0, m, S, sa, sb, (double)KD, out, it);
    if constexpr (G::PARTS == 2) __syncthreads();       // (`joined` and the parked rows are free for the next step)
    else wave_sync();
    advance();
    }
}

// ---- any even window size up to 128: lane = CELL.  Both windows sit in LDS (frame a's rows as zero-padded dwords,
// frame b's rows twice over at byte granularity, so that a row rotated by dx is one contiguous byte span for any W); the
// lane of cell j walks the rows of its rotation: per dword of a row one broadcast read of a, one read of b, v_alignbyte,
// v_dot4_u32_u8.  G = 4: four windows per wavefront, 16 lanes each (W <= 32: 8x8 first passes, the small generic sizes);
// G = 1: the 64 lanes are 4 row slices x 16 cells of one window, joined by two permlane swaps.  One wavefront per workgroup:
// LDS exchanges need no barrier.
template <int G>
__global__ __launch_bounds__(64) void xcorr_exact_refine_any_kernel(PassParams p, int ndw, int xp) {
    extern __shared__ uint32_t any_smem[];
    constexpr int LPW = 64 / G, SL = LPW / 16;               // lanes per window, row slices per cell
    const int W = p.ws, KD = W * W;
    const int lane = (int)threadIdx.x;
    const int g = lane / LPW, c = lane & 15, k = (lane % LPW) >> 4;
    const int N = p.n_rows * p.n_cols;
    const long long total = (long long)p.batch * N;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long chunk = (total + 7) / 8;
    const int st = p.ws - p.ov;
    const bool dwords = (W & 3) == 0;
    const float rcp_ndw = 1.0f / (float)ndw;

    // ---- staging, window after window, all 64 lanes (the loads of the G windows are independent of each other)
    unsigned sa_w[G], sb_w[G];
    long long it_w[G];
    bool valid_w[G];
    static_for<0, G>([&](auto gc) TPIV_LAMBDA_INLINE {
        constexpr int gi = decltype(gc)::value;
        const long long in_chunk = (long long)slot * G + gi;
        const long long it_raw = (long long)xcd * chunk + in_chunk;
        valid_w[gi] = in_chunk < chunk && it_raw < total;
        it_w[gi] = valid_w[gi] ? it_raw : 0;
        const unsigned itu = (unsigned)it_w[gi];
        const int pair = (int)(itu / (unsigned)N), win = (int)(itu - (unsigned)pair * (unsigned)N);
        const size_t off = (size_t)pair * p.H * p.W + (size_t)((win / p.n_cols) * st) * p.W + (size_t)(win % p.n_cols) * st;
        const uint8_t* __restrict__ A0 = p.A + off;
        const uint8_t* __restrict__ B0 = p.B + off;
        uint32_t* const ag = any_smem + gi * W * ndw;
        uint32_t* const bg = any_smem + G * W * ndw + gi * W * xp;
        uint8_t* const bgb = reinterpret_cast<uint8_t*>(bg);
        unsigned pa = 0u, pb = 0u;
        for (int idx = lane; idx < W * ndw; idx += 64) {
            const int row = (int)(((float)idx + 0.5f) * rcp_ndw), i = idx - row * ndw;
            const uint8_t* ra = A0 + (size_t)row * p.W + 4 * i;
            const uint8_t* rb = B0 + (size_t)row * p.W + 4 * i;
            uint32_t da = 0u, db = 0u;
            const int nb = W - 4 * i < 4 ? W - 4 * i : 4;   // bytes of this dword inside the window row
            if (nb == 4) {
                __builtin_memcpy(&da, ra, 4);
                __builtin_memcpy(&db, rb, 4);
            } else {                                        // (the tail of a row whose length is not a multiple of four)
                for (int t = 0; t < nb; ++t) {
                    da |= (uint32_t)ra[t] << (8 * t);
                    db |= (uint32_t)rb[t] << (8 * t);
                }
            }
            ag[row * ndw + i] = da;
            pa = __builtin_amdgcn_sad_u8(da, 0u, pa);
            pb = __builtin_amdgcn_sad_u8(db, 0u, pb);
            if (dwords) {
                bg[row * xp + i] = db;
                bg[row * xp + ndw + i] = db;
            } else {
                for (int t = 0; t < nb; ++t) {
                    const uint8_t v = (uint8_t)(db >> (8 * t));
                    bgb[row * xp * 4 + 4 * i + t] = v;
                    bgb[row * xp * 4 + W + 4 * i + t] = v;
                }
            }
        }
        auto uadd = [](unsigned x, unsigned y) TPIV_LAMBDA_INLINE { return x + y; };
        sa_w[gi] = grp_reduce<64>(pa, uadd);
        sb_w[gi] = grp_reduce<64>(pb, uadd);
    });
    wave_sync();

    // ---- this lane's window
    long long it = it_w[0];
    bool valid = valid_w[0];
    unsigned sa = sa_w[0], sb = sb_w[0];
    static_for<1, G>([&](auto gc) TPIV_LAMBDA_INLINE {
        constexpr int gi = decltype(gc)::value;
        it = g == gi ? it_w[gi] : it;
        valid = g == gi ? valid_w[gi] : valid;
        sa = g == gi ? sa_w[gi] : sa;
        sb = g == gi ? sb_w[gi] : sb;
    });
    const uint4 rec = p.cand[it];
    auto lo16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v & 0xffffu); };
    auto hi16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v >> 16); };
    const int m = valid ? lo16(rec.x) : -3;
    double* const out = reinterpret_cast<double*>(p.peak_raw) + (size_t)it * 8;
    const bool writer = k == 0;
    if (m == -1 && c == 0 && writer) p.fb_list[atomicAdd(p.fb_count, 1u)] = (int)it;
    if (m == -2 && c < 8 && writer) out[c] = c == 6 ? 0.0 : 1.0;          // zero-mean window (B:513): finalize_kernel looks at the flag [7] only
    const bool go = m >= 0;
    if (__ballot(go) == 0ull) return;
    int q = -1;
    if (go) {
        int left = m + 1, right = m - 1, top = m + W, bot = m - W;      // B:385-392
        if (left >= KD - 1) left = m;
        if (right <= 0) right = m;
        if (top >= KD - 1) top = m;
        if (bot <= 0) bot = m;
        q = c == 0 ? m : q;
        q = c == 1 ? left : q;
        q = c == 2 ? right : q;
        q = c == 3 ? top : q;
        q = c == 4 ? bot : q;
        q = c == 5 ? hi16(rec.x) : q;
        q = c == 6 ? lo16(rec.y) : q;
        q = c == 7 ? hi16(rec.y) : q;
        q = c == 8 ? lo16(rec.z) : q;
        q = c == 9 ? hi16(rec.z) : q;
        q = c == 10 ? lo16(rec.w) : q;
        q = c == 11 ? hi16(rec.w) : q;
    }
    // ---- S(q): the rows k, k + SL, ... of this lane's rotation
    unsigned S = 0u;
    {
        const bool on = q >= 0;
        const int qq = on ? q : 0;
        const int qy = qq / W, qx = qq - qy * W;
        int dy = qy - W / 2, bx = qx - W / 2;
        bx += bx < 0 ? W : 0;
        const unsigned sh = (unsigned)(bx & 3);
        const uint32_t* const ag = any_smem + g * W * ndw;
        const uint32_t* const bg = any_smem + G * W * ndw + g * W * xp + (bx >> 2);
        for (int y = k; y < W; y += SL) {
            int by = y + dy;
            by += by < 0 ? W : 0;
            by -= by >= W ? W : 0;
            const uint32_t* ar = ag + y * ndw;
            const uint32_t* br = bg + by * xp;
            uint32_t prev = br[0];
            for (int i = 0; i < ndw; ++i) {
                const uint32_t nxt = br[i + 1];
                S = __builtin_amdgcn_udot4(ar[i], __builtin_amdgcn_alignbyte(nxt, prev, sh), S, false);
                prev = nxt;
            }
        }
        S = on ? S : 0u;
    }
    if constexpr (SL == 4) {        // the four row slices of a cell sit in the four rows of 16 lanes
        auto r = __builtin_amdgcn_permlane16_swap(S, S, false, false);
        S = r[0] + r[1];
        auto r2 = __builtin_amdgcn_permlane32_swap(S, S, false, false);
        S = r2[0] + r2[1];
    }
    decide_and_store(p, rec, c, go, writer, q >= 0, m, S, sa, sb, (double)KD, out, it);
}

}  // namespace

// cand / fb_list / fb_count are set by the caller (launch_xcorr); the records go where the float64 kernel puts them
template <int W>
static hipError_t launch_refine(const PassParams& p, int n_cu, hipStream_t stream) {
    using G = XGeo<W>;
    const long long total = (long long)p.batch * p.n_rows * p.n_cols;
    const long long chunk = (total + 7) / 8;
    const long long slots = (chunk + G::WINS - 1) / G::WINS;
    // the resident set: LDS (parked rows) and wavefront slots per CU; TPIV_REFINE_PER_CU overrides for experiments
    static const int per_cu_env = [] {
        const char* e = getenv("TPIV_REFINE_PER_CU");
        return e ? atoi(e) : 0;
    }();
    const size_t lds = sizeof(uint32_t) * G::WINS * W * G::XP + 128;
    int per_cu = (int)((160 * 1024) / lds);
    constexpr int WPS = W == 128 ? 2 : (W == 64 ? 4 : (W == 32 ? 5 : 8));    // wavefronts per SIMD by the VGPR count (188 / 107 / 82 / 61)
    const int by_waves = 4 * WPS / G::WAVES;
    per_cu = per_cu > by_waves ? by_waves : per_cu;
    if (per_cu_env > 0) per_cu = per_cu_env;
    long long blocks = (long long)(n_cu > 0 ? n_cu : 256) * per_cu / 8 * 8;
    if (blocks > slots * 8) blocks = slots * 8;
    if (blocks < 8) blocks = 8;
    hipLaunchKernelGGL((xcorr_exact_refine_kernel<W>), dim3((unsigned)blocks), dim3(64 * G::WAVES), 0, stream, p);
    return hipGetLastError();
}

template <int G>
static hipError_t launch_refine_any(const PassParams& p, hipStream_t stream) {
    const int W = p.ws, ndw = (W + 3) / 4;
    const int xp = ((2 * W + 6 + 3) / 4) | 1;              // dwords per parked row: reads reach byte 2 W + 5; odd pitch
    const long long total = (long long)p.batch * p.n_rows * p.n_cols;
    const long long chunk = (total + 7) / 8;
    const long long slots = (chunk + G - 1) / G;
    const size_t smem = (size_t)G * W * (ndw + xp) * sizeof(uint32_t);
    if (smem > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL((xcorr_exact_refine_any_kernel<G>), dim3((unsigned)(slots * 8)), dim3(64), smem, stream, p, ndw, xp);
    return hipGetLastError();
}

// window sizes the exact scheme covers: every even size from 8 to 128 (odd sizes have the reference's ws x (ws - 1) map,
// B:255 -- not a plain circular correlation --, larger ones do not fit the refinement's LDS: both run the float64 kernels)
bool exact_refine_size(int ws) { return ws >= 8 && ws <= 128 && (ws & 1) == 0; }

hipError_t launch_exact_refine(const PassParams& p, int n_cu, hipStream_t stream) {
    const long long total = (long long)p.batch * p.n_rows * p.n_cols;
    if (total <= 0 || total >= (1ll << 31) || p.cand == nullptr || p.fb_list == nullptr || p.fb_count == nullptr ||
        !exact_refine_size(p.ws))
        return hipErrorInvalidValue;
    switch (p.ws) {
        case 16: return launch_refine<16>(p, n_cu, stream);
        case 32: return launch_refine<32>(p, n_cu, stream);
        case 64: return launch_refine<64>(p, n_cu, stream);
        case 128: return launch_refine<128>(p, n_cu, stream);
        default: return p.ws <= 32 ? launch_refine_any<4>(p, stream) : launch_refine_any<1>(p, stream);
    }
}

}  // namespace tpiv
