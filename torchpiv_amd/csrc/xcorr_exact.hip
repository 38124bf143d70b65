// Exact first pass for 64x64 windows (precision "exact"): the values behind the sub-pixel fit and the validity ratio
// as EXACT integer correlation sums of the uint8 windows.
//
// The reference (PIVbackend.py:513-518) divides both windows by their means in float64, correlates them through a
// float64 FFT, subtracts the map minimum and adds 1e-7; every cell of that map is
//     (S(d) - S_min) n^4 / (sum a  sum b) + 1e-7,      S(d) = sum_p a[p] b[(p + d) mod 64]   (integers < 2^28)
// up to the rounding of the transform.  Only a handful of cells per window ever reach the result (B:383-411): the
// arg-max, its four flat-index neighbours, the second peak, and the minimum.  So:
//   1. xcorr_tile_cand_kernel<64> (xcorr_tile.hpp): the float32 FFT pass LOCATES those cells, with an error band around
//      every decision (peak_candidates);
//   2. exact_refine_kernel (here): one wavefront per window evaluates S at the located cells -- lane = window row,
//      frame-a row in registers, frame b's rows parked twice over in LDS so that a row rotated by dx is one contiguous
//      span; 16 v_dot4_u32_u8 per lane and cell --, re-checks the decisions on the exact values (no neighbour or second
//      candidate above the arg-max, no evaluated cell below the minimum) and writes the 8-double record of
//      finalize_kernel<true>.  Undecided windows are appended to a list;
//   3. xcorr_f64_split_kernel<64, true> (xcorr_f64.hip) runs the float64 transform for the listed windows only.
// tests/test_exact_scheme.py is the numpy statement of the scheme (checked against the oracle: 7e-15 px);
// tests/test_gpu_exact.py compares this file with it and with the float64 kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "piv_kernels.h"
#include "xcorr_tile.hpp"      // grp_reduce, wave_sync, load_dwords

namespace tpiv {

namespace {

constexpr int XW = 64;                  // window edge
constexpr int XP = 33;                  // dwords per parked row: 2 x 16 (the row twice over) + 1 (odd pitch: lane = row reads hit 64 banks)
constexpr int XWAVES = 4;               // windows per workgroup
constexpr int XCELLS = 5 + EXACT_MAX_SECOND + EXACT_MAX_MIN;      // cells a window can ask for

__global__ __launch_bounds__(64 * XWAVES) void exact_refine_kernel(PassParams p) {
    __shared__ uint32_t parked[XWAVES][XW * XP];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    uint32_t* const rows_b = parked[wave];
    const int N = p.n_rows * p.n_cols;
    const long long total = (long long)p.batch * N;
    // XCD-aware static order: workgroups b, b+8, ... share an XCD (and its L2) and cover one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long long chunk = (total + 7) / 8;
    const long long in_chunk = (long long)slot * XWAVES + wave;
    const long long it = (long long)xcd * chunk + in_chunk;
    if (in_chunk >= chunk || it >= total) return;

    const uint4 rec = p.cand[it];
    auto lo16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v & 0xffffu); };
    auto hi16 = [](unsigned v) TPIV_LAMBDA_INLINE { return (int)(short)(v >> 16); };
    const int m = lo16(rec.x);
    double* const out = reinterpret_cast<double*>(p.peak_raw) + (size_t)it * 8;
    auto to_f64_kernel = [&]() TPIV_LAMBDA_INLINE {
        if (lane == 0) p.fb_list[atomicAdd(p.fb_count, 1u)] = (int)it;
    };
    if (m == -1) {
        to_f64_kernel();
        return;
    }
    if (m == -2) {               // zero-mean window (B:513: NaN map): finalize_kernel looks at the flag only
        if (lane < 8) out[lane] = lane == 7 ? 1.0 : (lane == 6 ? 0.0 : 1.0);
        return;
    }
    constexpr int KD = XW * XW;
    // ---- the cells: lane j holds flat index q_j (fftshift layout) or -1
    int q;
    {
        int left = m + 1, right = m - 1, top = m + XW, bot = m - XW;      // B:385-392
        if (left >= KD - 1) left = m;
        if (right <= 0) right = m;
        if (top >= KD - 1) top = m;
        if (bot <= 0) bot = m;
        q = -1;
        q = lane == 0 ? m : q;
        q = lane == 1 ? left : q;
        q = lane == 2 ? right : q;
        q = lane == 3 ? top : q;
        q = lane == 4 ? bot : q;
        q = lane == 5 ? hi16(rec.x) : q;
        q = lane == 6 ? lo16(rec.y) : q;
        q = lane == 7 ? hi16(rec.y) : q;
        q = lane == 8 ? lo16(rec.z) : q;
        q = lane == 9 ? hi16(rec.z) : q;
        q = lane == 10 ? lo16(rec.w) : q;
        q = lane == 11 ? hi16(rec.w) : q;
    }

    // ---- window rows: lane = row
    const int pair = (int)(it / N), win = (int)(it % N);
    const int st = p.ws - p.ov;
    const size_t off = (size_t)pair * p.H * p.W + (size_t)((win / p.n_cols) * st + lane) * p.W + (size_t)(win % p.n_cols) * st;
    uint32_t a[XW / 4], b[XW / 4];
    load_dwords<XW / 4>(p.A + off, a);
    load_dwords<XW / 4>(p.B + off, b);
    unsigned sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < XW / 4; ++i) {
        sa = __builtin_amdgcn_sad_u8(a[i], 0u, sa);
        sb = __builtin_amdgcn_sad_u8(b[i], 0u, sb);
        rows_b[lane * XP + i] = b[i];
        rows_b[lane * XP + XW / 4 + i] = b[i];
    }
    auto uadd = [](unsigned x, unsigned y) TPIV_LAMBDA_INLINE { return x + y; };
    sa = grp_reduce<64>(sa, uadd);
    sb = grp_reduce<64>(sb, uadd);
    wave_sync();

    // ---- S at every requested cell; lane j keeps S(q_j)
    unsigned S = 0;
    for (int c = 0; c < XCELLS; ++c) {
        const int qc = __builtin_amdgcn_readlane(q, c);
        if (qc < 0) continue;
        const int dy = (qc >> 6) - XW / 2, dx = (qc & 63) - XW / 2;
        const int brow = (lane + dy) & 63, bx = dx & 63;
        const uint32_t* src = rows_b + brow * XP + (bx >> 2);
        const unsigned sh = (unsigned)(bx & 3);
        uint32_t w[XW / 4 + 1];
#pragma unroll
        for (int i = 0; i <= XW / 4; ++i) w[i] = src[i];
        unsigned acc = 0;
#pragma unroll
        for (int i = 0; i < XW / 4; ++i) acc = __builtin_amdgcn_udot4(a[i], __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh), acc, false);
        acc = grp_reduce<64>(acc, uadd);
        S = lane == c ? acc : S;
    }

    // ---- the decisions, re-checked on the exact values
    const bool have = q >= 0;
    auto umin = [](unsigned x, unsigned y) TPIV_LAMBDA_INLINE { return x < y ? x : y; };
    auto umax = [](unsigned x, unsigned y) TPIV_LAMBDA_INLINE { return x > y ? x : y; };
    const unsigned s_min = grp_reduce<64>((have && lane >= 5 + EXACT_MAX_SECOND && lane < XCELLS) ? S : 0xffffffffu, umin);
    const unsigned s_low = grp_reduce<64>((have && lane < XCELLS) ? S : 0xffffffffu, umin);
    const unsigned s_top = grp_reduce<64>((have && lane < 5 + EXACT_MAX_SECOND) ? S : 0u, umax);
    const unsigned s_m = (unsigned)__builtin_amdgcn_readlane((int)S, 0);
    const unsigned long long seconds = __ballot(have && lane >= 5 && lane < 5 + EXACT_MAX_SECOND);
    const unsigned s_second = grp_reduce<64>((have && lane >= 5 && lane < 5 + EXACT_MAX_SECOND) ? S : 0u, umax);
    if (s_top > s_m || s_low < s_min || s_min == 0xffffffffu || sa == 0u || sb == 0u) {
        to_f64_kernel();          // (cannot happen while the float32 map stays inside the band; sa, sb: the float32 kernel marks those)
        return;
    }
    // (S - S_min) n^4 / (sum a sum b) + 1e-7: the integer difference is exact, sum a * sum b < 2^40 is exact
    const double scale = ((double)KD * (double)KD) / ((double)sa * (double)sb);
    const unsigned mine = lane == 5 ? (seconds ? s_second : s_m) : S;      // B:411: no cell left -> the first peak itself
    double v = __fma_rn((double)(mine - s_min), scale, 1e-7);
    v = lane == 6 ? (double)m : v;
    v = lane == 7 ? 0.0 : v;
    if (lane < 8) out[lane] = v;
}

}  // namespace

// cand / fb_list / fb_count are set by the caller (launch_xcorr); the records go where the float64 kernel puts them
hipError_t launch_exact_refine(const PassParams& p, hipStream_t stream) {
    const long long total = (long long)p.batch * p.n_rows * p.n_cols;
    if (p.ws != XW || total <= 0 || total >= (1ll << 31) || p.cand == nullptr || p.fb_list == nullptr || p.fb_count == nullptr)
        return hipErrorInvalidValue;
    const long long chunk = (total + 7) / 8;
    const long long slots = (chunk + XWAVES - 1) / XWAVES;
    hipLaunchKernelGGL(exact_refine_kernel, dim3((unsigned)(slots * 8)), dim3(64 * XWAVES), 0, stream, p);
    return hipGetLastError();
}

}  // namespace tpiv
