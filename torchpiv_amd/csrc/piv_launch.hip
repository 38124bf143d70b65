// Predictor kernels, finalize kernel and the tile-size dispatch (tile kernels: xcorr_tile.hpp, xcorr_big.hpp,
// xcorr_generic.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "piv_kernels.h"

namespace tpiv {

hipError_t launch_xcorr_ws8(const PassParams& p, int mode, int n_cu, hipStream_t stream);
hipError_t launch_xcorr_w8(const PassParams& p, int mode, int n_cu, hipStream_t stream);       // xcorr_w8.hip: one window per lane
hipError_t launch_peak_debug_w8(const PassParams& p, const float* maps, int n_maps, hipStream_t stream);
hipError_t launch_xcorr_ws16(const PassParams& p, int mode, int n_cu, hipStream_t stream);
hipError_t launch_xcorr_ws32(const PassParams& p, int mode, int n_cu, hipStream_t stream);
hipError_t launch_xcorr_ws64(const PassParams& p, int mode, int n_cu, hipStream_t stream);
hipError_t launch_xcorr_big128_pass1(const PassParams& p, int n_cu, hipStream_t stream);
hipError_t launch_peak_debug_ws8(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream);
hipError_t launch_peak_debug_ws16(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream);
hipError_t launch_peak_debug_ws32(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream);
hipError_t launch_peak_debug_ws64(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream);
hipError_t launch_peak_debug_ws128(const PassParams& p, const float* maps, int n_maps, hipStream_t stream);
hipError_t launch_xcorr_generic(const PassParams& p, int mode, int n_cu, void* scratch, hipStream_t stream);
int generic_blocks(int ws, long long items, int n_cu, int elem_bytes);
bool generic_ct_usable(int ws);
int generic_ct_register_size(int ws, int mode);
hipError_t launch_xcorr_f64(const PassParams& p, int n_cu, hipStream_t stream);      // xcorr_f64.hip: 8..64, pass 1
// precision "exact" (pass 1): float32 candidate pass, exact integer refinement, float64 pass for the undecided windows
hipError_t launch_xcorr_cand_ws8(const PassParams& p, int n_cu, hipStream_t stream);        // xcorr_tile.hpp
hipError_t launch_xcorr_cand_ws16(const PassParams& p, int n_cu, hipStream_t stream);
hipError_t launch_xcorr_cand_ws32(const PassParams& p, int n_cu, hipStream_t stream);
hipError_t launch_xcorr_cand_ws64(const PassParams& p, int n_cu, hipStream_t stream);
hipError_t launch_xcorr_cand_ws128(const PassParams& p, int n_cu, hipStream_t stream);      // xcorr_big.hpp
hipError_t launch_exact_refine(const PassParams& p, int n_cu, hipStream_t stream);                    // xcorr_exact.hip
bool exact_refine_size(int ws);                                                             // every even size 8 ... 128
hipError_t launch_xcorr_f64_list(const PassParams& p, int n_cu, hipStream_t stream);        // xcorr_f64.hip

// ---- finalize: sub-pixel fit, validation and multipass combine, one thread per window -----------
// PIVbackend.py:385-422 (correlation_to_displacement) and B:728-738 / B:800-810 (combine).  Input:
// PassParams::peak_raw written by the tile kernel.
__device__ __forceinline__ double nan_to_num_f(double x) {      // torch.nan_to_num_ defaults, B:418-419
    if (x != x) return 0.0;
    if (x > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (x < -1.7976931348623157e308) return -1.7976931348623157e308;
    return x;
}

// F64: the records are 8 doubles (float64 first pass), else 8 floats
template <bool F64>
__global__ __launch_bounds__(256) void finalize_kernel(PassParams p, int mode) {
    const size_t total = (size_t)p.batch * p.n_rows * p.n_cols;
    // the map is d = ws rows by k columns; k = ws - 1 for odd window sizes (the reference's irfft2 quirk, see
    // xcorr_generic.hip), and the reference's formulas take the column from m % k but the row from m // d
    const int d_ = p.ws, k_ = (p.ws & 1) ? p.ws - 1 : p.ws;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        double cm, cl, cr, ct, cb, c2;
        int m;
        bool dead;
        if constexpr (F64) {
            const double* __restrict__ rw = reinterpret_cast<const double*>(p.peak_raw) + i * 8;
            cm = rw[0], cl = rw[1], cr = rw[2], ct = rw[3], cb = rw[4], c2 = rw[5];
            m = (int)rw[6];
            dead = rw[7] != 0.0;
        } else {
            const float4* __restrict__ rw = reinterpret_cast<const float4*>(p.peak_raw + i * 8);
            const float4 r0 = rw[0], r1 = rw[1];
            cm = (double)r0.x, cl = (double)r0.y, cr = (double)r0.z, ct = (double)r0.w;
            cb = (double)r1.x, c2 = (double)r1.y;
            m = __float_as_int(r1.z);
            dead = __float_as_int(r1.w) != 0;
        }
        const double lm = log(cm), ll = log(cl), lr = log(cr), lt = log(ct), lb = log(cb);
        const double nom1 = lr - ll;                           // B:399-402
        const double den1 = 2 * (ll + lr) - 4 * lm;
        const double nom2 = lb - lt;
        const double den2 = 2 * (lb + lt) - 4 * lm;
        double du = (double)(m % k_) + nom1 / den1 - (double)(k_ / 2);      // B:407, B:416-417
        double dv = (double)(m / d_) + nom2 / den2 - (double)(d_ / 2);      // B:404-406, B:415
        du = nan_to_num_f(du);
        dv = nan_to_num_f(dv);
        bool invalid = (cm / c2) < p.val_ratio;                // B:411
        if (mode == MODE_PASS1) {
            if (dead) {              // zero-mean window: all-NaN map in the reference -> u = v = 0, "valid"
                du = 0.0;
                dv = 0.0;
                invalid = false;
            }
            // Identical windows (frame b == frame a over the window): the correlation is an autocorrelation -- even by
            // construction -- so with the peak at zero displacement the three-point fit is EXACTLY zero in exact
            // arithmetic; what a transform returns instead is its rounding noise, +-1e-17 px (float32: +-1e-8) with a
            // random sign.  That sign matters downstream: the reference's window shift treats an exactly integral
            // coordinate differently from one a rounding error away (B:170 / B:193) and its flat-index wrap turns the
            // sign of a 1e-17 px shift at column 0 into another pixel -- a +-1e-17 perturbation of pass 1 moves 16 % of
            // the final vectors of such a pair by up to 3 px.  The reference's own float64 transform happens to return
            // exact zeros there (MKL keeps the map symmetric bit for bit); so does this: a window whose fit is within
            // 1e-4 px of zero with its peak at the centre is compared byte for byte, and identical windows get 0.
            // (Never taken on real pairs: two exposures differ by their noise.)
            if (!dead && (m % k_) == k_ / 2 && (m / d_) == d_ / 2 && fabs(du) < 1e-4 && fabs(dv) < 1e-4 && p.A != nullptr) {
                const int pair = (int)(i / ((size_t)p.n_rows * p.n_cols)), win = (int)(i % ((size_t)p.n_rows * p.n_cols));
                const int st = p.ws - p.ov;
                const size_t off = (size_t)pair * p.H * p.W + (size_t)(win / p.n_cols) * st * p.W + (size_t)(win % p.n_cols) * st;
                // (16 bytes per load and frame, any alignment; the row loop ends at the first difference -- a static scene
                //  sends every window here, so the comparison must not be a byte loop)
                bool same = true;
                for (int r = 0; r < p.ws && same; ++r) {
                    const uint8_t* __restrict__ ra = p.A + off + (size_t)r * p.W;
                    const uint8_t* __restrict__ rb = p.B + off + (size_t)r * p.W;
                    int c = 0;
                    for (; c + 16 <= p.ws; c += 16) {
                        uint4 va, vb;
                        __builtin_memcpy(&va, ra + c, 16);
                        __builtin_memcpy(&vb, rb + c, 16);
                        same = same && va.x == vb.x && va.y == vb.y && va.z == vb.z && va.w == vb.w;
                    }
                    for (; c < p.ws; ++c) same = same && ra[c] == rb[c];
                }
                if (same) {
                    du = 0.0;
                    dv = 0.0;
                }
            }
            p.u[i] = du;
            p.v[i] = dv;
            p.val[i] = invalid ? 1 : 0;
        } else {                     // multipass combine (B:728-738 / B:800-810)
            double u0, v0;
            pred_fallback(p, i, u0, v0);
            double u, v;
            if (mode == MODE_CWSF) {             // B:663-664: the windows were resampled by -/+ u0/2
                u = u0 + du;
                v = v0 + dv;
            } else {
                double sx, sy;
                if (mode == MODE_DWS) pred_half_shift<MODE_DWS>(p, i, sx, sy);
                else pred_half_shift<MODE_CWS>(p, i, sx, sy);
                u = 2 * sx + du;
                v = 2 * sy + dv;
            }
            const bool mask_u = ((du > u0) && (rint(u0) > 0)) || invalid;
            const bool mask_v = ((dv > v0) && (rint(v0) > 0)) || invalid;
            if (mask_u) u = u0;
            if (mask_v) v = v0;
            p.u[i] = u;
            p.v[i] = v;
            p.val[i] = invalid ? 1 : 0;
            if (p.du != nullptr) {
                p.du[i] = du;
                p.dv[i] = dv;
            }
        }
    }
}

static bool tile_size(int ws) { return ws == 8 || ws == 16 || ws == 32 || ws == 64; }
static size_t peak_bytes(int batch, int n_windows, int precision = 0) {
    return (((size_t)batch * n_windows * 8 * (precision ? sizeof(double) : sizeof(float))) + 255) / 256 * 256;
}

// peak records, followed (generic sizes only) by the DFT scratch tiles of the resident workgroups
// test hook: peak analysis + finalize (pass-1 semantics) on caller-supplied correlation maps
hipError_t launch_peaks_from_maps(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream) {
    hipError_t e;
    switch (p.ws) {
        case 8:          // planar == 2: the one-window-per-lane kernel's peak stage (xcorr_w8.hip)
            e = planar == 2 ? launch_peak_debug_w8(p, maps, n_maps, stream) : launch_peak_debug_ws8(p, maps, n_maps, planar, stream);
            break;
        case 16: e = launch_peak_debug_ws16(p, maps, n_maps, planar, stream); break;
        case 32: e = launch_peak_debug_ws32(p, maps, n_maps, planar, stream); break;
        case 64: e = launch_peak_debug_ws64(p, maps, n_maps, planar, stream); break;
        case 128: e = launch_peak_debug_ws128(p, maps, n_maps, stream); break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finalize_kernel<false>, dim3((n_maps + 255) / 256), dim3(256), 0, stream, p, (int)MODE_PASS1);
    return hipGetLastError();
}

static constexpr size_t WORK_CTR_BYTES = TILE_CTR_BYTES;
static size_t work_ctr_offset(int batch, int n_windows) { return (peak_bytes(batch, n_windows) + 127) / 128 * 128; }

// precision "exact": float64 records | candidate cells (16 B per window) | float64 list | its counter | tile work counters
// | generic sizes: the DFT scratch tiles of the first-generation kernel (float32 locating pass and float64 list pass)
struct ExactLayout {
    size_t cand, list, count, ctr, scratch, total;
};
static ExactLayout exact_layout(int ws, int batch, int n_windows) {
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    ExactLayout l;
    l.cand = peak_bytes(batch, n_windows, 1);
    l.list = l.cand + up((size_t)batch * n_windows * sizeof(uint4));
    l.count = l.list + up((size_t)batch * n_windows * sizeof(int));
    l.ctr = l.count + 256;
    l.scratch = l.ctr + WORK_CTR_BYTES;
    l.total = l.scratch;
    if (!tile_size(ws) && ws != 128)
        l.total += (size_t)generic_blocks(ws, (long long)batch * n_windows, 256, 8) * 2 * ws * ws * 2 * 8;
    return l;
}
// window sizes whose first pass runs the exact scheme: every even size from 8 to 128 (xcorr_exact.hip: exact_refine_size)
static bool exact_size(int ws) { return exact_refine_size(ws); }
size_t exact_fallback_count_offset(int batch, int n_windows) { return exact_layout(64, batch, n_windows).count; }

size_t peak_raw_bytes(int ws, int batch, int n_windows, int precision, bool force_generic) {
    if (precision == 3) {
        if (exact_size(ws) && !force_generic) return exact_layout(ws, batch, n_windows).total;
        precision = 1;                                                              // other sizes: the float64 kernels
    }
    if (tile_size(ws) && !force_generic) {
        if (precision) return peak_bytes(batch, n_windows, 1);                      // float64 records only
        // records + item counters (+ 64x64: the list of the items the fast-path-only CWS kernel sets aside)
        return work_ctr_offset(batch, n_windows) + WORK_CTR_BYTES +
               (ws == 64 ? TILE_SLOW_HDR_BYTES + (size_t)batch * n_windows * sizeof(int) : 0);
    }
    // generic sizes (and shifted 128x128 passes): records + the DFT scratch tiles of the resident workgroups
    const int eb = precision ? 8 : 4;
    return peak_bytes(batch, n_windows, precision) +
           (size_t)generic_blocks(ws, (long long)batch * n_windows, 256, eb) * 2 * ws * ws * 2 * eb;
}

// (in the demangled form rocprofv3 prints, so that profile rows can be matched by substring; the MODE
//  template argument is the tpiv::MODE_* value: 0 pass 1, 1 DWS, 2 CWS)
// TPIV_PREC_F64 (2): float64 pass 1, fast operation order in the shifted passes
// TPIV_PREC_EXACT (3): exact integer sums in pass 1 (64x64; float64 kernels for the other sizes), shifted passes as (2)
static int pass_precision(int precision, int mode, int ws) {
    if (precision == 2) return mode == MODE_PASS1 ? 1 : 0;
    if (precision == 3) return mode == MODE_PASS1 ? (exact_size(ws) ? 3 : 1) : 0;
    return precision;
}

const char* xcorr_kernel_name(int ws, int mode, int precision, char* buf, int len) {
    precision = pass_precision(precision, mode, ws);
    if (precision == 3 && mode == MODE_PASS1) {
        if (ws == 128) snprintf(buf, len, "xcorr_big128_cand_kernel");
        else if (tile_size(ws)) snprintf(buf, len, "xcorr_tile_cand_kernel<%d>", ws);
        else if (generic_ct_usable(ws)) snprintf(buf, len, "xcorr_generic_ct_kernel<0, %d> (cand)", generic_ct_register_size(ws, mode));
        else snprintf(buf, len, "xcorr_generic_kernel<0, float> (cand)");
    } else if (precision && mode == MODE_PASS1) {
        if (ws == 64 || ws == 128) snprintf(buf, len, "xcorr_f64_split_kernel<%d>", ws);
        else if (tile_size(ws)) snprintf(buf, len, "xcorr_f64_tile_kernel<%d>", ws);
        else snprintf(buf, len, "xcorr_generic_kernel<0, double>");
    } else if (ws == 8) {
        snprintf(buf, len, "xcorr_w8_kernel<%d, %s>", mode, (precision && mode != MODE_PASS1) ? "false" : "true");
    } else if (tile_size(ws)) {
        snprintf(buf, len, "xcorr_tile_kernel<%d, %d, %d, %s>", ws, mode, tile_occ(ws, mode),
                 (precision && mode != MODE_PASS1) ? "false" : "true");
    } else if (ws == 128 && mode == MODE_PASS1) {
        snprintf(buf, len, "xcorr_big128_kernel");
    } else {
        if (generic_ct_usable(ws)) snprintf(buf, len, "xcorr_generic_ct_kernel<%d, %d>", mode, generic_ct_register_size(ws, mode));
        else snprintf(buf, len, "xcorr_generic_kernel<%d, float>", mode);
    }
    if (mode == MODE_CWSF) snprintf(buf, len, generic_ct_usable(ws) ? "xcorr_generic_ct_kernel<3, 0>" : "xcorr_generic_kernel<3, float>");
    return buf;
}

hipError_t launch_xcorr(const PassParams& p_in, int mode, int n_cu, hipStream_t stream) {
    hipError_t e;
    PassParams p = p_in;
    p.precision = pass_precision(p.precision, mode, p.ws);
    // float64 exists for pass 1 only (the reference's later passes are float32, B:249-257); for shifted
    // passes precision != 0 selects the tile kernel's reference-order arithmetic (xcorr_tile.hpp)
    const bool f64 = p.precision != 0 && mode == MODE_PASS1;
    auto generic = [&]() {
        if (p.ws < 2 || p.ws > 256) return hipErrorInvalidValue;
        void* scratch = reinterpret_cast<char*>(p.peak_raw) + peak_bytes(p.batch, p.n_rows * p.n_cols, f64 ? 1 : 0);
        PassParams q = p;
        q.precision = f64 ? 1 : 0;
        return launch_xcorr_generic(q, mode, 256, scratch, stream);
    };
    if (mode == MODE_CWSF) {
        e = generic();
    } else if (p.precision == 3) {
        const ExactLayout l = exact_layout(p.ws, p.batch, p.n_rows * p.n_cols);
        char* const base = reinterpret_cast<char*>(p.peak_raw);
        p.cand = reinterpret_cast<uint4*>(base + l.cand);
        p.fb_list = reinterpret_cast<int*>(base + l.list);
        p.fb_count = reinterpret_cast<unsigned*>(base + l.count);
        p.work_ctr = reinterpret_cast<unsigned*>(base + l.ctr);
        // the decision band of the locating pass: 2 Gamma (1 + 1/16) E+ (piv_kernels.h, "The band"), Gamma by the kind of
        // transform the size's locating kernel runs
        // (TPIV_EXACT_BAND_SCALE / TPIV_EXACT_BAND_RANGE: experiments with the band, tools/research/exact_band.py)
        static const float band_scale = [] {
            const char* env = getenv("TPIV_EXACT_BAND_SCALE");
            return env ? (float)atof(env) : 1.0f;
        }();
        static const float band_range = [] {
            const char* env = getenv("TPIV_EXACT_BAND_RANGE");
            return env ? (float)atof(env) : 0.0f;
        }();
        const int fft_kind = (tile_size(p.ws) || p.ws == 128) ? EXACT_FFT_RADIX2 : (generic_ct_usable(p.ws) ? EXACT_FFT_MIXED : EXACT_FFT_PLAIN);
        p.exact_band = band_scale * exact_band_coef(p.ws, fft_kind);
        p.exact_band_range = band_range;
        e = hipMemsetAsync(p.fb_count, 0, 256 + WORK_CTR_BYTES, stream);
        if (e != hipSuccess) return e;
        auto mark = [&](int i) {
            if (p.sub_events) (void)hipEventRecord(p.sub_events[i], stream);
        };
        PassParams q = p;                    // generic sizes: the float32 kernels with `cand` set, the float64 one with the list
        q.precision = 0;
        void* const gscratch = base + l.scratch;
        switch (p.ws) {
            case 8: e = launch_xcorr_cand_ws8(p, n_cu, stream); break;
            case 16: e = launch_xcorr_cand_ws16(p, n_cu, stream); break;
            case 32: e = launch_xcorr_cand_ws32(p, n_cu, stream); break;
            case 64: e = launch_xcorr_cand_ws64(p, n_cu, stream); break;
            case 128: e = launch_xcorr_cand_ws128(p, n_cu, stream); break;
            default: e = launch_xcorr_generic(q, MODE_PASS1, 256, gscratch, stream); break;
        }
        if (e != hipSuccess) return e;
        mark(0);
        e = launch_exact_refine(p, n_cu, stream);
        if (e != hipSuccess) return e;
        mark(1);
        if (tile_size(p.ws) || p.ws == 128) {
            e = launch_xcorr_f64_list(p, n_cu, stream);
        } else {
            q.precision = 1;
            q.cand = nullptr;
            e = launch_xcorr_generic(q, MODE_PASS1, 256, gscratch, stream);
        }
        mark(2);
    } else if (f64) {
        // (TPIV_F64_GENERIC128=1: the generic-size DFT kernel for 128x128, as before the split kernel existed -- A/B runs)
        static const bool gen128 = [] {
            const char* env = getenv("TPIV_F64_GENERIC128");
            return env && env[0] == '1';
        }();
        e = (tile_size(p.ws) || (p.ws == 128 && !gen128)) ? launch_xcorr_f64(p, n_cu, stream) : generic();
    } else if (tile_size(p.ws)) {       // per-XCD work queue of the tile kernel: counters behind the peak records
        p.work_ctr = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.peak_raw) +
                                                 work_ctr_offset(p.batch, p.n_rows * p.n_cols));
        p.list_mode = 0;
        p.slow_list = nullptr;
        p.slow_count = nullptr;
        const bool split = tile_split64(p.ws, mode) && p.precision == 0;
        if (split) {        // (the list lives behind the counters: peak_raw_bytes)
            p.slow_count = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.work_ctr) + WORK_CTR_BYTES);
            p.slow_list = reinterpret_cast<int*>(reinterpret_cast<char*>(p.slow_count) + TILE_SLOW_HDR_BYTES);
        }
        e = hipMemsetAsync(p.work_ctr, 0, WORK_CTR_BYTES + (split ? TILE_SLOW_HDR_BYTES : 0), stream);
        if (e != hipSuccess) return e;
        // 8x8: one window per lane (xcorr_w8.hip); TPIV_W8=0 selects the lane-per-row tile kernel for A/B runs
        static const bool w8 = [] {
            const char* env = getenv("TPIV_W8");
            return !(env && env[0] == '0');
        }();
        switch (p.ws) {
            case 8: e = w8 ? launch_xcorr_w8(p, mode, n_cu, stream) : launch_xcorr_ws8(p, mode, n_cu, stream); break;
            case 16: e = launch_xcorr_ws16(p, mode, n_cu, stream); break;
            case 32: e = launch_xcorr_ws32(p, mode, n_cu, stream); break;
            default: e = launch_xcorr_ws64(p, mode, n_cu, stream); break;
        }
    } else if (p.ws == 128 && mode == MODE_PASS1) {
        e = launch_xcorr_big128_pass1(p, n_cu, stream);
    } else {
        e = generic();          // generic sizes, and shifted 128x128 passes
    }
    if (e != hipSuccess) return e;
    const size_t total = (size_t)p.batch * p.n_rows * p.n_cols;
    size_t blocks = (total + 255) / 256;
    if (blocks > (size_t)n_cu * 32) blocks = (size_t)n_cu * 32;
    if (f64) hipLaunchKernelGGL(finalize_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, p, mode);
    else hipLaunchKernelGGL(finalize_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, p, mode);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------
// predictor: Z_f = Ay * Z_c * Ax^T  (float64), PIVbackend.py:700-713 / 769-785
// ----------------------------------------------------------------------------
// step 1: T[b, f, rc, cf] = sum_cc Z[b, f, rc, cc] * Ax[cf, cc]
__global__ void predict_cols_kernel(PredictParams q) {
    const long long total = (long long)q.batch * 3 * q.nrc * q.ncf;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cfi = (int)(i % q.ncf);
        const int rc = (int)((i / q.ncf) % q.nrc);
        const int f = (int)((i / ((long long)q.ncf * q.nrc)) % 3);
        const int b = (int)(i / ((long long)q.ncf * q.nrc * 3));
        const double* ax = q.Ax + (size_t)cfi * q.ncc;
        const size_t zoff = ((size_t)b * q.nrc + rc) * q.ncc;
        double acc = 0.0;
        if (f == 2) {
            const uint8_t* z = q.val_c + zoff;
            for (int k = 0; k < q.ncc; ++k) acc += (double)z[k] * ax[k];
        } else {
            const double* z = (f == 0 ? q.u_c : q.v_c) + zoff;
            for (int k = 0; k < q.ncc; ++k) acc += z[k] * ax[k];
        }
        q.T[i] = acc;
    }
}

// step 2: out[b, f, rf, cf] = sum_rc Ay[rf, rc] * T[b, f, rc, cf], then the per-mode
// predictor post-processing.
__global__ void predict_rows_kernel(PredictParams q) {
    const long long total = (long long)q.batch * q.nrf * q.ncf;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cfi = (int)(i % q.ncf);
        const int rf = (int)((i / q.ncf) % q.nrf);
        const int b = (int)(i / ((long long)q.ncf * q.nrf));
        const double* ay = q.Ay + (size_t)rf * q.nrc;
        const double* t0 = q.T + ((size_t)b * 3 + 0) * q.nrc * q.ncf + cfi;
        const double* t1 = q.T + ((size_t)b * 3 + 1) * q.nrc * q.ncf + cfi;
        const double* t2 = q.T + ((size_t)b * 3 + 2) * q.nrc * q.ncf + cfi;
        double u0 = 0.0, v0 = 0.0, vm = 0.0;
        for (int k = 0; k < q.nrc; ++k) {
            const double a = ay[k];
            u0 += a * t0[(size_t)k * q.ncf];
            v0 += a * t1[(size_t)k * q.ncf];
            vm += a * t2[(size_t)k * q.ncf];
        }
        const bool val = vm >= 0.5;                 // B:711 / B:778
        double u2, v2;
        if (q.mode == MODE_CWS) {                   // B:705-706: halves taken BEFORE the zeroing
            u2 = u0 / 2;
            v2 = v0 / 2;
        }
        if (val) {
            u0 = 0.0;
            v0 = 0.0;
        }
        if (q.mode == MODE_DWS) {                   // B:782-785: AFTER the zeroing, half-even
            u2 = rint(u0 / 2);
            v2 = rint(v0 / 2);
        }
        q.u0[i] = u0;
        q.v0[i] = v0;
        q.u2[i] = u2;
        q.v2[i] = v2;
    }
}

hipError_t launch_predict(const PredictParams& q, hipStream_t stream) {
    {
        const long long total = (long long)q.batch * 3 * q.nrc * q.ncf;
        long long blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(predict_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    {
        const long long total = (long long)q.batch * q.nrf * q.ncf;
        long long blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(predict_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, q);
        return hipGetLastError();
    }
}

}  // namespace tpiv
