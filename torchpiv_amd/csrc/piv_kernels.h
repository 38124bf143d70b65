// Internal launch interface between the C-ABI layer (c_api.cpp) and the kernels
// (piv_kernels.hip).  Not part of the public boundary (include/torchpiv_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tpiv {

// MODE_CWSF: the reference's piv_iteration_CWS_Fast (B:599-675): bicubic resampling of every window inside
// itself; runs the generic-size kernel only
enum { MODE_PASS1 = 0, MODE_DWS = 1, MODE_CWS = 2, MODE_CWSF = 3 };

struct PassParams {
    const uint8_t* A;      // [batch, H, W] frame a
    const uint8_t* B;      // [batch, H, W] frame b
    int batch, H, W;
    int ws, ov, n_rows, n_cols;
    // predictor inputs of passes >= 2, [batch, n_rows, n_cols] float64
    const double* u0;      // predictor after invalid-zeroing (fallback value)
    const double* v0;
    const double* u2;      // window half-shift: DWS rint(u0/2), CWS u0_prezero/2
    const double* v2;
    // Compact hand-off from the plan's predictor (tpiv_plan_run): pmask != nullptr means u0 / v0 hold the RAW
    // predictor (before the invalid-zeroing), pmask its thresholded mask, and u2 / v2 are not used -- the
    // zeroing and the half shift are formed where they are read (pred_half_shift / pred_fallback below): two
    // float64 fields and a byte per window cross HBM instead of four fields.
    const uint8_t* pmask;
    // outputs [batch, n_rows, n_cols]
    double* u;
    double* v;
    uint8_t* val;          // 1 = invalid (peak ratio below val_ratio)
    double* du;            // optional: raw displacement of this pass (passes >= 2)
    double* dv;
    double val_ratio;
    int val_win;
    int precision;         // pass 1 only: 0 = float32 kernels, 1 = float64 (TPIV_PREC_REFERENCE), 3 = exact sums (64x64)
    // test hooks (nullptr in production)
    float* dbg_win;        // [batch, N, 2, ws, ws] staged windows (after the shift)
    float* dbg_corr;       // [batch, N, ws, ws] corr - min + eps, fftshift layout
    unsigned long long* stamps;   // diagnostic build (-DTPIV_STAMPS) only: per-phase cycle sums
    // workspace [batch, N, 8] float32 between the tile kernel and finalize_kernel:
    // {c[m], c[left], c[right], c[top], c[bot], c[m2], bits(m), bits(dead)} per window
    // (float64 pass 1: 8 doubles per window, m and dead stored as values)
    float* peak_raw;
    unsigned* work_ctr;      // 8 x 16 dwords: per-XCD item counters of the tile kernel (set by launch_xcorr)
    // 64x64 CWS pass in two launches (round 5): the fast-path-only kernel appends the items that need the per-pixel staging
    // path to slow_list (count: slow_count); the full kernel then runs with list_mode = 1 over exactly those items
    int* slow_list;
    unsigned* slow_count;
    int list_mode;
    // precision "exact" (64x64 pass 1, xcorr_exact.hip; all set by launch_xcorr): the float32 kernel's candidate cells
    // per window (8 x int16: arg-max, 3 second-peak cells, 4 minimum cells; -1 = none, arg-max -1 = undecided, -2 = dead),
    // and the windows that go to the float64 kernel instead (undecided ones)
    uint4* cand;
    int* fb_list;
    unsigned* fb_count;
    float exact_band;        // width of the decision band of the locating pass per unit of E+ (exact_band_coef(ws); "The band" below)
    float exact_band_range;  // optional floor of the band relative to the map range (TPIV_EXACT_BAND_RANGE experiments; 0 by default)
    hipEvent_t* sub_events;  // host side only, optional: [3] recorded behind the locating pass, the refinement, the float64 list pass
    // n / d for n < 2^31 as (n * magic) >> shift (set by the tile launcher; keeps the per-item index
    // arithmetic in the scalar unit instead of a hoisted float reciprocal that occupies a VGPR)
    unsigned groups_magic, ncols_magic;
    int groups_shift, ncols_shift;
};

// ---- precision "exact": constants shared by the locating passes (xcorr_tile.hpp peak_candidates, xcorr_big.hpp,
// xcorr_generic.hip map_candidates), the refinement (xcorr_exact.hip) and the launcher
// The band.  A decision of the locating pass is right whenever every cell of its float32 map is within band / 2 of the
// exact one (up to a common offset and a common factor 1 + O(u), which change no order).  The float32 map's cell error is
// BOUNDED -- DESIGN.md 3.4b derives it from the standard FFT error analysis (Higham, Accuracy and Stability of Numerical
// Algorithms, 2nd ed., Thm 24.2), u = 2^-24:
//     |map32(d) - map(d)|  <=  Gamma E+,     E+ = (|a'|^2 + |b'|^2) / 2  >=  |a'| |b'|,     a' = a / mean(a) - 1,
//     Gamma = 2 (F + 1) u + 5 u + I u
//   F u: normwise relative error of the forward 2-D transform of a' + i b' (+ 1 u for the rounding of its inputs), carried
//        through the bilinear cross-spectrum by Cauchy-Schwarz: sum_k |dP_k| <= |dZ| |Z| = (F + 1) u N (|a'|^2 + |b'|^2);
//   5 u: the cross-spectrum's own arithmetic (re = 2 (ad + bc), im = (c^2 - a^2) + (d^2 - b^2): <= 5 u (|z_k|^2 + |z_-k|^2) / 4 per bin);
//   I u: the inverse 2-D transform, componentwise: |dy_d| <= I u sum_k |P_k| <= I u N |a'| |b'|.
// Per 1-D transform of length n (a 2-D transform is two of them):
//   radix-2/4 codelets (fft_inreg.hpp; tile kernels and 128x128): F = I = eta log2 n, eta = 6.66 per radix-2 level
//       (twiddle error + 4 roundings; a radix-4 butterfly with its one twiddle per two levels is below two radix-2 levels);
//   two-factor mixed-radix transforms (fft_mixed.hpp, radix_pass: n = n1 n2, direct small DFTs of radix r <= 8, each
//       (r + 3) sqrt(r) u normwise, + 4 u for the twiddle between them): F = I <= 64 for every pair;
//   plain O(n^2) DFTs (first-generation generic kernel): componentwise (n + 3) u sum |x|: I = n + 3, F = (n + 3) sqrt(n).
// 64 x 64: Gamma = 247 u = 1.47e-5 (measured over tools/research/exact_band.py's families and its adversarial search: <= 8e-7).
// The locating pass forms E+ from the exact integer window sums (sum a, sum a^2: v_sad_u8 / v_dot4_u32_u8 on the bytes it
// staged) and uses band = 2 Gamma (1 + 1/16) E+ -- the 1/16 covers the float32 rounding of E+ itself and of the band
// comparisons.
constexpr double EXACT_ETA = 6.66;
enum { EXACT_FFT_RADIX2 = 0, EXACT_FFT_MIXED = 1, EXACT_FFT_PLAIN = 2 };
inline double exact_gamma_u(int ws, int kind) {       // Gamma in units of u = 2^-24
    double lg = 0.0, rt = 1.0;
    for (int n = 1; n < ws; n *= 2) lg += 1.0;         // ceil(log2 ws)
    while ((rt + 1.0) * (rt + 1.0) <= (double)ws) rt += 1.0;
    rt += 1.0;                                         // >= sqrt(ws)
    const double f1 = kind == EXACT_FFT_PLAIN ? (ws + 3) * rt : (kind == EXACT_FFT_MIXED ? 64.0 : lg * EXACT_ETA);
    const double i1 = kind == EXACT_FFT_PLAIN ? (double)(ws + 3) : f1;
    return 2.0 * (2.0 * f1 + 1.0) + 5.0 + 2.0 * i1;
}
inline float exact_band_coef(int ws, int kind = EXACT_FFT_RADIX2) {
    return (float)(2.0 * exact_gamma_u(ws, kind) * (1.0 + 1.0 / 16) * 5.9604644775390625e-08);
}
constexpr int EXACT_MAX_SECOND = 3, EXACT_MAX_MIN = 4;

struct PredictParams {
    int batch, mode;
    int nrc, ncc, nrf, ncf;       // coarse / fine grid
    const double* Ay;             // [nrf, nrc]
    const double* Ax;             // [ncf, ncc]
    const double* u_c;            // [batch, nrc, ncc]
    const double* v_c;
    const uint8_t* val_c;
    double* T;                    // workspace [batch, 3, nrc, ncf]
    double* u0;                   // [batch, nrf, ncf]
    double* v0;
    double* u2;
    double* v2;
};

// Banded form of the predictor (plan path, predict_mfma.hip).  The rows of the spline operator decay like
// 0.268^|j - j0|, so everything outside a 65-tap band is below float64 rounding of the sum.  Per block of
// 32 fine rows / columns: a dense K x 32 weight tile over the union of the block's bands (zero elsewhere),
// K a multiple of 8.
struct BandedPredictParams {
    int batch, mode;
    int nrc, ncc, nrf, ncf;
    int KY, KX;                   // taps per row / column block
    int nrfp;                     // nrf rounded up to a multiple of 32: pitch of the transposed T1
    const double* Wy32;           // [ceil(nrf / 32)][KY][32]
    const int* k0y32;             // first coarse row of the block's tile
    const double* Ax32;           // [ceil(ncf / 32)][KX][32]
    const int* k0x32;             // first coarse column of the block's tile
    const double* u_c;            // [batch, nrc, ncc]
    const double* v_c;
    const uint8_t* val_c;
    double* T1;                   // workspace [batch, 3, ncc, nrfp]: the row operator's result, transposed
    double* u0;                   // [batch, nrf, ncf]
    double* v0;
    double* u2;
    double* v2;
    uint8_t* mask_out;            // compact output (PassParams::pmask): u0 / v0 receive the RAW predictor, u2 / v2 nothing
};

// post-validation on the device (postval.hip)
struct PostvalParams {
    double* u;               // [batch, n_rows, n_cols], modified in place (border interpolation, determined fills)
    double* v;
    const uint8_t* invalid;  // [batch, n_rows, n_cols] 1 = invalid vector
    uint8_t* cls;            // out [batch, n_rows, n_cols]: 0 valid, 2 hole filled here, 3 ambiguous hole,
                             //     4 general hole, 5 ring (valid cell next to a hole)
    int* counts;             // out [batch, 4]: holes, ring cells, ambiguous holes, general holes
    int batch, n_rows, n_cols;
};
hipError_t launch_postval(const PostvalParams& p, hipStream_t stream);
// ring / hole lists of the pairs that need the host triangulation, packed pair after pair in np.argwhere order
hipError_t launch_postval_compact(const double* u, const double* v, const uint8_t* cls, const int* counts, int batch, int n_rows,
                                  int n_cols, int* offsets, int* ring_rc, double* ring_uv, int* hole_rc, hipStream_t stream);
// B:894-898 for a batch: flip along the rows, sign of v, unit scaling (the reference's expression, bit-identical)
hipError_t launch_finish_fields(const double* u, const double* v, int batch, int n_rows, int n_cols, double scale, double dt,
                                double* fu, double* fv, hipStream_t stream);
// ensemble statistics (postval.hip): out [5][cells] = mean u, mean v, <u'u'>, <v'v'>, <u'v'> of n stacked fields
hipError_t launch_ensemble_moments(const double* U, const double* V, int n, long long cells, double* out, hipStream_t stream);

// image ingest (ingest.hip): raw uncompressed BMP files -> uint8 frames
hipError_t launch_bmp_unpack(const uint8_t* raw, const long long* desc, const uint8_t* lut, int n_files, int H, int W,
                             uint8_t* out, hipStream_t stream);

hipError_t launch_xcorr(const PassParams& p, int mode, int n_cu, hipStream_t stream);
// bytes of the tile kernels' work-queue counters (8 x one 64-byte line), and of the slow-item list header behind them
constexpr size_t TILE_CTR_BYTES = 8 * 16 * sizeof(unsigned), TILE_SLOW_HDR_BYTES = 256;
// 64x64 CWS pass at the fast operation order: fast-path-only kernel at three wavefronts per SIMD + a second launch of the full
// kernel over the items it set aside (TPIV_SPLIT64=0: the single launch of rounds 1-4)
#ifndef TPIV_SPLIT64
#define TPIV_SPLIT64 1
#endif
constexpr bool tile_split64(int ws, int mode) { return TPIV_SPLIT64 && ws == 64 && mode == MODE_CWS; }
// wavefronts per SIMD the tile kernel of (ws, mode) is built for (the OCC template argument of
// xcorr_tile_kernel); 0 for sizes that run another kernel
#ifndef TPIV_OCC64C
#define TPIV_OCC64C 2      // wavefronts per SIMD of the 64x64 CWS kernel (experiments: 3)
#endif
constexpr int tile_occ_c(int ws, int mode) {
    return (ws != 8 && ws != 16 && ws != 32 && ws != 64) ? 0
           : (ws == 16 ? 4 : ((ws == 32 || (ws == 64 && (mode != MODE_CWS || TPIV_OCC64C == 3))) ? 3 : 2));
}
inline int tile_occ(int ws, int mode) { return tile_occ_c(ws, mode); }
#ifdef __HIPCC__
// The predictor as the kernels of a shifted pass read it.  MODE is MODE_DWS or MODE_CWS.
//   half shift (B:705-706 CWS: halves taken BEFORE the zeroing; B:782-785 DWS: rint(u0 / 2) AFTER it)
// (The loads are unconditional -- the form is chosen by selecting POINTERS, wave-uniformly: the kernels fetch the
//  next item's shift one iteration ahead, and a load under `if` makes the compiler wait right behind it.)
template <int MODE>
__device__ __forceinline__ void pred_half_shift(const PassParams& p, size_t i, double& sx, double& sy) {
    const bool compact = p.pmask != nullptr;
    const double* __restrict__ us = compact ? p.u0 : p.u2;
    const double* __restrict__ vs = compact ? p.v0 : p.v2;
    double u = us[i], v = vs[i];
    if constexpr (MODE == MODE_DWS) {
        const uint8_t* mp = compact ? p.pmask + i : reinterpret_cast<const uint8_t*>(us + i);    // (any readable byte)
        const bool zero = *mp != 0 && compact;
        const double uz = zero ? 0.0 : u, vz = zero ? 0.0 : v;
        sx = compact ? rint(uz / 2) : u;
        sy = compact ? rint(vz / 2) : v;
    } else {
        sx = compact ? u / 2 : u;
        sy = compact ? v / 2 : v;
    }
}
//   the CWS half shift as the float32 the staging uses (B:714-715 casts u2, v2 to float32): halving is exact in
//   either precision, so float(u / 2) == float(u) * 0.5f -- one float32 multiply instead of float64 work
__device__ __forceinline__ void pred_half_shift_cws_f32(const PassParams& p, size_t i, float& vx, float& vy) {
    const bool compact = p.pmask != nullptr;
    const double* __restrict__ us = compact ? p.u0 : p.u2;
    const double* __restrict__ vs = compact ? p.v0 : p.v2;
    const float k = compact ? 0.5f : 1.0f;
    vx = (float)us[i] * k;
    vy = (float)vs[i] * k;
}
//   fallback value: the predictor after the invalid-zeroing (B:711-713 / B:778-780)
__device__ __forceinline__ void pred_fallback(const PassParams& p, size_t i, double& u0, double& v0) {
    const bool compact = p.pmask != nullptr;
    const uint8_t* mp = compact ? p.pmask + i : reinterpret_cast<const uint8_t*>(p.u0 + i);
    const bool zero = *mp != 0 && compact;
    u0 = zero ? 0.0 : p.u0[i];
    v0 = zero ? 0.0 : p.v0[i];
}
#endif

// symbol-like name of the kernel launch_xcorr picks for (ws, mode) -- for bench / profile labels
const char* xcorr_kernel_name(int ws, int mode, int precision, char* buf, int len);
// bytes of PassParams::peak_raw a pass needs (records, work-queue counters, generic-size DFT scratch)
size_t peak_raw_bytes(int ws, int batch, int n_windows, int precision, bool force_generic = false);
// precision "exact": window sizes whose first pass runs the exact scheme (every even size from 8 to 128; xcorr_exact.hip)
bool exact_refine_size(int ws);
// precision "exact": byte offset of the float64-list counter (one unsigned) inside PassParams::peak_raw
size_t exact_fallback_count_offset(int batch, int n_windows);
// test hook: peak stage + finalize on caller-made maps; planar selects the LDS layout variant of the tile kernel
hipError_t launch_peaks_from_maps(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream);
hipError_t launch_predict_mfma(const BandedPredictParams& q, hipStream_t stream);
hipError_t launch_predict(const PredictParams& q, hipStream_t stream);

}  // namespace tpiv
