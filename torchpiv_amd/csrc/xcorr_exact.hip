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
    // rows are LOADED by the window's lanes together (W >= 32): 16-byte chunks, chunk ci = row * P16 + part, lane t of the
    // window takes chunks t, t + LW, ... -- consecutive lanes read consecutive pieces of a row
    static constexpr bool COOP = W >= 32;
    static constexpr int P16 = W / 16;                       // 16-byte chunks per row = chunks per lane
    static constexpr int LW = GROUP * PARTS;                 // lanes per window
    static constexpr int AP = NDW + 4;                       // dwords per row of the frame-a hand-over tile (16-byte reads, all banks)
    static_assert(!COOP || (W * AP <= W * XP && (LW * P16) == W * P16), "the hand-over tile lives in the parked rows' memory");
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

// Append window `it` to the float64 list for every lane that wants it: ONE atomic per wavefront (the lanes take consecutive
// slots by their rank among the appending lanes).  Per-lane atomics on the one counter serialise: a first pass of sparse 8x8
// windows on a noise-free background leaves 8 % of 4.2 M windows undecided -- 335 000 atomics took 3.4 of the refinement's 3.8 ms.
__device__ __forceinline__ void list_append(const PassParams& p, bool want, long long it) {
    const unsigned long long mk = __ballot(want);
    if (mk == 0ull) return;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int leader = (int)__builtin_ctzll(mk);
    unsigned base = 0u;
    if (lane == leader) base = atomicAdd(p.fb_count, (unsigned)__popcll(mk));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    if (want) p.fb_list[base + (unsigned)__popcll(mk & ((1ull << lane) - 1ull))] = (int)it;
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
    list_append(p, go && redo && r0 == 0 && writer, it);
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
    __shared__ __attribute__((aligned(16))) uint32_t parked[G::WINS][W * XP];
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
    // XCD-aware static order: workgroups b, b+8, ... share an XCD (and its L2) and cover one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long chunk = (total + 7) / 8;
    const long long in_chunk = (long long)slot * G::WINS + wslot;
    const long long it_raw = (long long)xcd * chunk + in_chunk;
    const bool valid = in_chunk < chunk && it_raw < total;
    const long long it = valid ? it_raw : 0;

    // ---- window rows.  Issued together with the candidate record, IN FRONT of the decisions that depend on it: the kernel
    //      is a chain of memory latencies (one window per wavefront lifetime), and the addresses do not depend on the record.
    //      (The 0.1 % of windows that do not go load their rows for nothing: in-bounds, unused.)
    // Round 5: W >= 32 loads by CHUNKS.  With lane = row every load instruction touched 64 different cache lines -- 512 tag
    // look-ups of the vector L1 per 64x64 window, one per CU cycle: 1.0 of the kernel's 1.26 ms per 1 016 064 windows was that
    // (a persistent form with the next window's rows prefetched was SLOWER, 1.6 ms: latency was never the limit).  Now four
    // consecutive lanes cover one 64-byte row (16 rows per instruction: a quarter of the look-ups); frame b's chunks go
    // straight to their parked place, frame a's pass through a hand-over tile in the same LDS (before b is parked) and come
    // back as the lane's row.  (The scatter of frame b's chunks into the odd-pitch rows is bank-conflicted -- 48 % of the kernel's
    // LDS-active cycles, PMC: no odd pitch lets 16 rows x 4 chunk quarters tile the 64 banks -- and still the fastest form: with
    // frame b through the hand-over tile as well, conflict-free, the kernel took 1.23 instead of 1.17 ms.)
    const unsigned itu = (unsigned)it;                       // (total < 2^31: 32-bit divisions)
    const int pair = (int)(itu / (unsigned)N), win = (int)(itu - (unsigned)pair * (unsigned)N);
    const int st = p.ws - p.ov;
    const size_t off0 = (size_t)pair * p.H * p.W + (size_t)((win / p.n_cols) * st) * p.W + (size_t)(win % p.n_cols) * st;
    uint32_t a[NDW], b[NDW];
    uint32_t ca[G::COOP ? G::P16 : 1][4], cb[G::COOP ? G::P16 : 1][4];
    const int tw = r0 + 64 * part;                           // lane inside the window
    if constexpr (G::COOP) {
#pragma unroll
        for (int k = 0; k < G::P16; ++k) {
            const int ci = tw + G::LW * k, j = ci / G::P16, pt = ci % G::P16;
            const size_t o_ = off0 + (size_t)j * p.W + 16 * pt;
            load_dwords<4>(p.A + o_, ca[k]);
            load_dwords<4>(p.B + o_, cb[k]);
        }
    } else {
        const size_t off = off0 + (size_t)row * p.W;
        load_dwords<NDW>(p.A + off, a);
        load_dwords<NDW>(p.B + off, b);
    }
    const uint4 rec = p.cand[it];
    auto lo16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v & 0xffffu); };
    auto hi16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v >> 16); };
    const int m = valid ? lo16(rec.x) : -3;
    double* const out = reinterpret_cast<double*>(p.peak_raw) + (size_t)it * 8;
    const bool writer = part == 0;                           // the wavefront of a window that stores its results
    list_append(p, m == -1 && r0 == 0 && writer, it);
    if (m == -2 && r0 < 8 && writer) out[r0] = r0 == 6 ? 0.0 : 1.0;      // zero-mean window (B:513: NaN map): finalize_kernel looks at the flag [7] only
    const bool go = m >= 0;
    if (__ballot(go) == 0ull) return;                        // (128x128: the same decision in both wavefronts of the window)

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
    auto lds_sync = [&]() TPIV_LAMBDA_INLINE {
        if constexpr (G::PARTS == 2) __syncthreads();   // (128x128: both wavefronts of the window)
        else wave_sync();
    };
    if constexpr (G::COOP) {
        // frame a: chunks -> hand-over tile (pitch AP dwords: 16-byte reads of 16 lanes cover all banks) -> the lane's row
        uint4* const ta = reinterpret_cast<uint4*>(rows_b);
#pragma unroll
        for (int k = 0; k < G::P16; ++k) {
            const int ci = tw + G::LW * k, j = ci / G::P16, pt = ci % G::P16;
            ta[j * (G::AP / 4) + pt] = make_uint4(ca[k][0], ca[k][1], ca[k][2], ca[k][3]);
        }
        lds_sync();
#pragma unroll
        for (int k = 0; k < NDW / 4; ++k) {
            const uint4 t = ta[row * (G::AP / 4) + k];
            a[4 * k] = t.x, a[4 * k + 1] = t.y, a[4 * k + 2] = t.z, a[4 * k + 3] = t.w;
        }
        lds_sync();
        // frame b: every chunk twice into its parked row (odd pitch: single dwords)
#pragma unroll
        for (int k = 0; k < G::P16; ++k) {
            const int ci = tw + G::LW * k, j = ci / G::P16, pt = ci % G::P16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                P[13] = __builtin_amdgcn_sad_u8(cb[k][c], 0u, P[13]);
                rows_b[j * XP + 4 * pt + c] = cb[k][c];
                rows_b[j * XP + NDW + 4 * pt + c] = cb[k][c];
            }
        }
#pragma unroll
        for (int i = 0; i < NDW; ++i) P[12] = __builtin_amdgcn_sad_u8(a[i], 0u, P[12]);
    } else {
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            P[12] = __builtin_amdgcn_sad_u8(a[i], 0u, P[12]);
            P[13] = __builtin_amdgcn_sad_u8(b[i], 0u, P[13]);
            rows_b[row * XP + i] = b[i];
            rows_b[row * XP + NDW + i] = b[i];
        }
    }
    lds_sync();                                         // the rows are parked

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
}

// ---- 8x8 and 16x16 windows: lane = WINDOW (round 5).  A first pass of 8-pixel windows has a million windows per 4096^2
// pair; with a wavefront per four windows the refinement was bound by workgroup launches (3.8 ms per 4.2 M windows against
// 0.6 ms for the locating pass).  Here a lane owns a whole window: frame a's rows in registers (W x W / 4 dwords), frame b's
// rows parked twice over in the lane's own LDS strip (odd strip pitch: the lanes' same-offset reads hit different banks), every
// cell W x (W / 4) dot products with compile-time register indices, every decision per lane -- no cross-lane traffic at all.
template <int W>
struct SGeo {
    static constexpr int NDW = W / 4, XP = 2 * NDW + 1, STRIP = (W * XP) | 1, KD = W * W;
};
// SW wavefronts per workgroup (8x8: four -- the float64-list appends of a workgroup go out as ONE atomic, see the end)
template <int W>
struct SBlock {
    static constexpr int SW = W == 8 ? 4 : 1;
};
template <int W>
__global__ __launch_bounds__(64 * SBlock<W>::SW) void xcorr_exact_refine_small_kernel(PassParams p) {
    using G = SGeo<W>;
    constexpr int NDW = G::NDW, XP = G::XP, KD = G::KD, SW = SBlock<W>::SW, NT = 64 * SW;
    __shared__ uint32_t strips[NT * G::STRIP];
    __shared__ unsigned blk_n, blk_base;
    const int lane = (int)threadIdx.x;                       // (thread of the workgroup = window slot)
    uint32_t* const mine = strips + lane * G::STRIP;
    if (SW > 1 && lane == 0) blk_n = 0u;
    const int N = p.n_rows * p.n_cols;
    const long long total = (long long)p.batch * N;
    // XCD-aware static order: workgroups b, b+8, ... share an XCD and cover one contiguous run of windows, 64 per workgroup
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long chunk = (total + 7) / 8;
    const long long in_chunk = (long long)slot * NT + lane;
    const long long it_raw = (long long)xcd * chunk + in_chunk;
    const bool valid = in_chunk < chunk && it_raw < total;
    const long long it = valid ? it_raw : 0;
    const unsigned itu = (unsigned)it;
    const int pair = (int)(itu / (unsigned)N), win = (int)(itu - (unsigned)pair * (unsigned)N);
    const int st = p.ws - p.ov;
    const size_t off = (size_t)pair * p.H * p.W + (size_t)((win / p.n_cols) * st) * p.W + (size_t)(win % p.n_cols) * st;
    const uint4 rec = p.cand[it];
    uint32_t a[W][NDW];
    unsigned sa = 0u, sb = 0u;
#pragma unroll
    for (int y = 0; y < W; ++y) {
        uint32_t b[NDW];
        load_dwords<NDW>(p.A + off + (size_t)y * p.W, a[y]);
        load_dwords<NDW>(p.B + off + (size_t)y * p.W, b);
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            sa = __builtin_amdgcn_sad_u8(a[y][i], 0u, sa);
            sb = __builtin_amdgcn_sad_u8(b[i], 0u, sb);
            mine[y * XP + i] = b[i];
            mine[y * XP + NDW + i] = b[i];
        }
    }
    auto lo16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v & 0xffffu); };
    auto hi16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v >> 16); };
    const int m = valid ? lo16(rec.x) : -3;
    double* const out = reinterpret_cast<double*>(p.peak_raw) + (size_t)it * 8;
    if (m == -2) {                                           // zero-mean window (B:513): finalize_kernel looks at the flag [7] only
#pragma unroll
        for (int r = 0; r < 8; ++r) out[r] = r == 6 ? 0.0 : 1.0;
    }
    const bool go = m >= 0;
    bool redo = false;
    if (__ballot(go) != 0ull) {
    int q[XCELLS];
    {
        int left = m + 1, right = m - 1, top = m + W, bot = m - W;      // B:385-392
        if (left >= KD - 1) left = m;
        if (right <= 0) right = m;
        if (top >= KD - 1) top = m;
        if (bot <= 0) bot = m;
        q[0] = m, q[1] = left, q[2] = right, q[3] = top, q[4] = bot;
        q[5] = hi16(rec.x), q[6] = lo16(rec.y), q[7] = hi16(rec.y);
        q[8] = lo16(rec.z), q[9] = hi16(rec.z), q[10] = lo16(rec.w), q[11] = hi16(rec.w);
    }
    wave_sync();                                             // (a lane reads its own strip only; program order)
    unsigned S[XCELLS];
    static_for<0, XCELLS>([&](auto cc) TPIV_LAMBDA_INLINE {
        constexpr int c = decltype(cc)::value;
        const bool on = go && q[c] >= 0;
        unsigned acc = 0u;
        if (__ballot(on) != 0ull) {
            const int qq = on ? q[c] : 0;
            const int dy = qq / W - W / 2, bx = (qq % W - W / 2) & (W - 1);
            const unsigned sh = (unsigned)(bx & 3);
            const uint32_t* const src = mine + (bx >> 2);
#pragma unroll
            for (int y = 0; y < W; ++y) {
                const uint32_t* rowp = src + ((y + dy) & (W - 1)) * XP;
                uint32_t w[NDW + 1];
#pragma unroll
                for (int i = 0; i <= NDW; ++i) w[i] = rowp[i];
#pragma unroll
                for (int i = 0; i < NDW; ++i) acc = __builtin_amdgcn_udot4(a[y][i], __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh), acc, false);
            }
        }
        S[c] = on ? acc : 0u;
    });
    // ---- the decisions on the exact values (decide_and_store, per lane)
    constexpr int S0 = 5, N0 = 5 + EXACT_MAX_SECOND;
    unsigned s_min = 0xffffffffu, s_low = 0xffffffffu, s_top = 0u, s_second = 0u, n_second = 0u;
#pragma unroll
    for (int c = 0; c < XCELLS; ++c) {
        const bool have = q[c] >= 0;
        if (c >= N0) s_min = (have && S[c] < s_min) ? S[c] : s_min;
        s_low = (have && S[c] < s_low) ? S[c] : s_low;
        if (c < N0) s_top = (have && S[c] > s_top) ? S[c] : s_top;
        if (c >= S0 && c < N0) {
            s_second = (have && S[c] > s_second) ? S[c] : s_second;
            n_second += have ? 1u : 0u;
        }
    }
    const unsigned s_m = S[0];
    const bool min_overflow = hi16(rec.w) == -2;
    redo = s_top > s_m || s_low < s_min || s_min == 0xffffffffu || sa == 0u || sb == 0u || (min_overflow && s_min != 0u);
    if (go && !redo) {
        const double scale = ((double)KD * (double)KD) / ((double)sa * (double)sb);
#pragma unroll
        for (int r = 0; r < 5; ++r) out[r] = __fma_rn((double)(S[r] - s_min), scale, 1e-7);
        out[5] = n_second == 0u ? 0.0 : __fma_rn((double)(s_second - s_min), scale, 1e-7);
        out[6] = (double)m;
        out[7] = 0.0;
    }
    }
    // ---- the float64 list: undecided by the locating pass, or a decision that did not survive the exact values
    const bool want = m == -1 || (go && redo);
    if constexpr (SW == 1) {
        list_append(p, want, it);
    } else {
        // one GLOBAL atomic per workgroup of 256 windows: the wavefronts reserve their slots in an LDS counter first (a
        // noise-free 8x8 first pass appends 8 % of 4.2 M windows: 1.06 ms of the kernel with one atomic per wavefront)
        const unsigned long long mk = __ballot(want);
        const int wl = lane & 63;
        unsigned wbase = 0u;
        __syncthreads();                                     // (blk_n = 0 is visible)
        if (mk != 0ull && wl == 0) wbase = atomicAdd(&blk_n, (unsigned)__popcll(mk));
        wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)wbase);
        __syncthreads();
        if (lane == 0 && blk_n != 0u) blk_base = atomicAdd(p.fb_count, blk_n);
        __syncthreads();
        if (want) p.fb_list[blk_base + wbase + (unsigned)__popcll(mk & ((1ull << wl) - 1ull))] = (int)it;
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
    list_append(p, m == -1 && c == 0 && writer, it);
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
static hipError_t launch_refine(const PassParams& p, hipStream_t stream) {
    using G = XGeo<W>;
    const long long total = (long long)p.batch * p.n_rows * p.n_cols;
    const long long chunk = (total + 7) / 8;
    const long long slots = (chunk + G::WINS - 1) / G::WINS;
    hipLaunchKernelGGL((xcorr_exact_refine_kernel<W>), dim3((unsigned)(slots * 8)), dim3(64 * G::WAVES), 0, stream, p);
    return hipGetLastError();
}

template <int W>
static hipError_t launch_refine_small(const PassParams& p, hipStream_t stream) {
    const long long total = (long long)p.batch * p.n_rows * p.n_cols;
    const long long chunk = (total + 7) / 8;
    constexpr int NT = 64 * SBlock<W>::SW;
    const long long slots = (chunk + NT - 1) / NT;
    hipLaunchKernelGGL((xcorr_exact_refine_small_kernel<W>), dim3((unsigned)(slots * 8)), dim3(NT), 0, stream, p);
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
    // (TPIV_REFINE_SMALL=0: the lane-per-row / lane-per-cell kernels for 16 / 8 as well -- A/B runs)
    static const bool small_on = [] { const char* e = getenv("TPIV_REFINE_SMALL"); return !(e && e[0] == '0'); }();
    switch (p.ws) {
        case 8:
            if (small_on) return launch_refine_small<8>(p, stream);
            return launch_refine_any<4>(p, stream);
        case 16: return small_on ? launch_refine_small<16>(p, stream) : launch_refine<16>(p, stream);
        case 32: return launch_refine<32>(p, stream);
        case 64: return launch_refine<64>(p, stream);
        case 128: return launch_refine<128>(p, stream);
        default: return p.ws <= 32 ? launch_refine_any<4>(p, stream) : launch_refine_any<1>(p, stream);
    }
}

}  // namespace tpiv
