// C ABI of the PIV engine (include/torchpiv_hip.h): argument checking, geometry,
// the predictor's spline operators (host, float64) and the multipass plan.
// Compiled with hipcc as host code; the kernels live in xcorr_ws*.hip / piv_launch.hip.
#include "../../include/torchpiv_hip.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "piv_kernels.h"

namespace {

thread_local std::string g_err;
#ifdef TPIV_STAMPS
unsigned long long* g_stamps = nullptr;      // diagnostic builds (make stamps) only: device buffer for phase stamps
#else
constexpr unsigned long long* g_stamps = nullptr;      // the production library keeps no state
#endif

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(TPIV_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

// The function-level seam (tpiv_pass1 / tpiv_iter / tpiv_debug_*) works in a caller-provided buffer:
// the library keeps no device state between calls.
int check_work(const void* work, size_t have, size_t need) {
    if (need == 0) return TPIV_OK;
    if (!work || have < need)
        return fail(TPIV_EINVAL, "work buffer too small: need " + std::to_string(need) + " bytes (tpiv_work_bytes), got " +
                                     std::to_string(have));
    return TPIV_OK;
}

// 8/16/32/64: tile kernel (xcorr_tile.hpp); 128 pass 1: xcorr_big.hpp; anything else in 2..256 (and
// shifted 128x128 passes): generic-size DFT kernel (xcorr_generic.hip)
bool supported_ws(int ws) { return ws >= 2 && ws <= 256; }

// B:503-507 argument checks, then what the kernels cover
int check_window(int H, int W, int ws, int ov, int val_win) {
    if (ov >= ws) return fail(TPIV_EINVAL, "Overlap has to be smaller than the window_size");
    if (ws > H || ws > W) return fail(TPIV_EINVAL, "window size cannot be larger than the image");
    if (ws <= 0 || ov < 0 || H <= 0 || W <= 0) return fail(TPIV_EINVAL, "non-positive size");
    if (!supported_ws(ws))
        return fail(TPIV_EUNSUPPORTED, "window size must be in 2..256 (got " + std::to_string(ws) + ")");
    // (odd sizes: the generic kernel reproduces the reference's ws x (ws-1) irfft2 map; ws = 1 has no map)
    if (ws % 2 != 0 && ws < 3) return fail(TPIV_EUNSUPPORTED, "window size 1 has an empty correlation map in the reference");
    const bool pow2 = ws == 8 || ws == 16 || ws == 32 || ws == 64 || ws == 128;
    if (val_win < 0 || (pow2 && 2 * val_win >= ws))
        return fail(TPIV_EUNSUPPORTED, "validation half-window must satisfy 2*val_win < window size");
    if (((long long)H + 32) * W >= (1LL << 30)) return fail(TPIV_EUNSUPPORTED, "frame too large");   // 32-bit flat indices
    return TPIV_OK;
}

int n_cu_of_current_device(int* n_cu) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static thread_local int cached_dev = -1, cached_cu = 0;
    if (cached_dev != dev) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        cached_dev = dev;
        cached_cu = prop.multiProcessorCount;
    }
    *n_cu = cached_cu;
    return TPIV_OK;
}

void field_shape(int H, int W, int ws, int ov, int* nr, int* nc) {
    *nr = (H - ws) / (ws - ov) + 1;
    *nc = (W - ws) / (ws - ov) + 1;
}

// floor division for the (possibly negative) centring shift of B:582-592
long long floordiv(long long a, long long b) {
    long long q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

void coords_1d(int size, int n, int ws, int ov, double* out) {
    const long long shift = floordiv((long long)size - 1 - ((long long)(n - 1) * (ws - ov) + (ws - 1)), 2);
    for (int i = 0; i < n; ++i) out[i] = (double)i * (ws - ov) + ws / 2.0 + (double)shift;
}

// ---- cubic B-spline interpolation operator (FITPACK regrid, s = 0) -------------
// knots: 4x x[0], x[2..n-3], 4x x[n-1]  (not-a-knot at x[1] and x[n-2])
struct Bspl {
    std::vector<long double> t;
    int n;
    explicit Bspl(int n_, const double* x) : t(n_ + 4), n(n_) {
        for (int i = 0; i < 4; ++i) t[i] = x[0];
        for (int i = 2; i <= n - 3; ++i) t[i + 2] = x[i];
        for (int i = 0; i < 4; ++i) t[n + i] = x[n - 1];
    }
    // span l with t[l] <= xx < t[l+1], 3 <= l <= n-1; the 4 non-zero cubic B-splines
    // N_{l-3..l}(xx) by the de Boor-Cox recurrence
    int eval(long double xx, long double h[4]) const {
        int l = 3;
        while (l < n - 1 && xx >= t[l + 1]) ++l;
        long double hh[4];
        h[0] = 1.0L;
        for (int j = 1; j <= 3; ++j) {
            for (int i = 0; i < j; ++i) hh[i] = h[i];
            h[0] = 0.0L;
            for (int i = 0; i < j; ++i) {
                const int li = l + i + 1, lj = li - j;
                const long double f = hh[i] / (t[li] - t[lj]);
                h[i] += f * (t[li] - xx);
                h[i + 1] = f * (xx - t[lj]);
            }
        }
        return l;
    }
};

int spline_matrix(int nc, const double* xc, int nf, const double* xf, double* A) {
    if (nc < 4) return fail(TPIV_EINVAL, "the spline predictor needs at least 4 coarse points per axis");
    for (int i = 1; i < nc; ++i)
        if (!(xc[i] > xc[i - 1])) return fail(TPIV_EINVAL, "coarse coordinates must increase strictly");
    Bspl bs(nc, xc);
    const int n = nc;
    // collocation matrix C[i][j] = B_j(x_i): rows have 4 non-zeros at columns l-3..l
    std::vector<long double> C((size_t)n * n, 0.0L);
    for (int i = 0; i < n; ++i) {
        long double h[4];
        const int l = bs.eval((long double)xc[i], h);
        for (int k = 0; k < 4; ++k) C[(size_t)i * n + (l - 3 + k)] = h[k];
    }
    // Inv = C^-1: banded LU (forward elimination with partial pivoting inside the band,
    // lower bandwidth <= 3) applied to the identity, then back-substitution (the upper
    // bandwidth grows to <= 6 through the row swaps).
    std::vector<long double> Inv((size_t)n * n, 0.0L);
    for (int i = 0; i < n; ++i) Inv[(size_t)i * n + i] = 1.0L;
    const int kl = 3, ku = 6;
    for (int col = 0; col < n; ++col) {
        int piv = col;
        long double best = fabsl(C[(size_t)col * n + col]);
        for (int r = col + 1; r < n && r <= col + kl; ++r) {
            const long double v = fabsl(C[(size_t)r * n + col]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
        if (best == 0.0L) return fail(TPIV_EINVAL, "singular spline collocation matrix");
        if (piv != col) {
            for (int k = 0; k < n; ++k) {
                std::swap(C[(size_t)piv * n + k], C[(size_t)col * n + k]);
                std::swap(Inv[(size_t)piv * n + k], Inv[(size_t)col * n + k]);
            }
        }
        const int k_hi = col + ku + 1 < n ? col + ku + 1 : n;
        for (int r = col + 1; r < n && r <= col + kl; ++r) {
            const long double f = C[(size_t)r * n + col] / C[(size_t)col * n + col];
            if (f == 0.0L) continue;
            for (int k = col; k < k_hi; ++k) C[(size_t)r * n + k] -= f * C[(size_t)col * n + k];
            for (int k = 0; k < n; ++k) Inv[(size_t)r * n + k] -= f * Inv[(size_t)col * n + k];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        const int k_hi = i + ku + 1 < n ? i + ku + 1 : n;
        const long double d = 1.0L / C[(size_t)i * n + i];
        for (int j = 0; j < n; ++j) {
            long double acc = Inv[(size_t)i * n + j];
            for (int k = i + 1; k < k_hi; ++k) acc -= C[(size_t)i * n + k] * Inv[(size_t)k * n + j];
            Inv[(size_t)i * n + j] = acc * d;
        }
    }
    for (int f = 0; f < nf; ++f) {
        long double xx = xf[f];
        if (xx < xc[0]) xx = xc[0];            // FITPACK fpbisp clamps the arguments
        if (xx > xc[n - 1]) xx = xc[n - 1];
        long double h[4];
        const int l = bs.eval(xx, h);
        for (int j = 0; j < n; ++j) {
            long double acc = 0.0L;
            for (int k = 0; k < 4; ++k) acc += h[k] * Inv[(size_t)(l - 3 + k) * n + j];
            A[(size_t)f * n + j] = (double)acc;
        }
    }
    return TPIV_OK;
}

bool valid_precision(int p) { return p == TPIV_PREC_FAST || p == TPIV_PREC_REFERENCE || p == TPIV_PREC_F64 || p == TPIV_PREC_EXACT; }

struct PassGeo {
    int ws, ov, n_rows, n_cols;
    std::vector<double> x, y;
};

}  // namespace

struct tpiv_plan {
    int H = 0, W = 0, n_pass = 0, mode = 0, max_batch = 0, val_win = 3, device = 0, precision = 0;
    double val_ratio = 1.2;
    std::vector<PassGeo> geo;
    // device workspace
    std::vector<double*> u, v;           // per pass (all but the last): [max_batch, N_p]
    std::vector<uint8_t*> val;
    std::vector<double*> Ay, Ax;         // per pass p >= 1: dense operators from pass p-1 to p (debug)
    // banded form used by tpiv_plan_run (piv_kernels.h: BandedPredictParams)
    std::vector<double*> Wy32, Ax32;
    std::vector<int*> k0y32, k0x32;
    std::vector<int> KY, KX;
    double *u0 = nullptr, *v0 = nullptr, *T = nullptr;      // predictor hand-off: raw fields ...
    uint8_t* pmask = nullptr;                               // ... and the thresholded mask (PassParams::pmask)
    float* peak_raw = nullptr;           // [max_batch, max N_p, 8] hand-off tile kernel -> finalize
    size_t peak_raw_bytes = 0;
    int last_batch = 0;                  // batch of the last tpiv_plan_run (tpiv_plan_exact_fallbacks)
    unsigned* exact_count = nullptr;     // TPIV_PREC_EXACT: the float64-list length of the last run's first pass (copied out
                                         // of peak_raw, which the later passes reuse)
    std::vector<void*> allocs;
    // optional per-kernel timing: events[run][2*slot + {0,1}]
    bool timing = false;
    std::vector<std::vector<hipEvent_t>> events;
    size_t runs_recorded = 0;

    ~tpiv_plan() {
        for (void* p : allocs) (void)hipFree(p);
        for (auto& r : events)
            for (hipEvent_t e : r) (void)hipEventDestroy(e);
    }
    int n_slots() const { return 2 * n_pass - 1; }
    // TPIV_PREC_EXACT: three more events per run inside slot 0 (behind the locating pass, the refinement, the float64 list pass)
    double exact_ms[4] = {0, 0, 0, 0};   // their mean durations at the last tpiv_plan_get_timing (+ finalize)
    hipEvent_t* sub_events() {
        if (!exact_count || !ev(0, 0)) return nullptr;
        return events[runs_recorded].data() + 2 * n_slots();
    }
    // event for (current run, slot, begin/end); nullptr when timing is off or the record is full
    hipEvent_t ev(int slot, int end) {
        if (!timing || runs_recorded >= 512) return nullptr;
        if (events.size() <= runs_recorded) {
            std::vector<hipEvent_t> r(2 * n_slots() + 3, nullptr);
            for (auto& e : r)
                if (hipEventCreate(&e) != hipSuccess) return nullptr;
            events.push_back(std::move(r));
        }
        return events[runs_recorded][2 * slot + end];
    }
    template <typename T_>
    int alloc(T_** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, count * sizeof(T_) + 16);
        if (e != hipSuccess) return fail(TPIV_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        allocs.push_back(p);
        *out = static_cast<T_*>(p);
        return TPIV_OK;
    }
};

namespace {

// band of one operator row: PRED_BW taps around the largest entry (clipped to the matrix); returns
// the largest magnitude left outside so that the caller can verify the truncation is harmless
constexpr int PRED_BW = 65;

int band_start(const double* row, int nc, int bw) {
    int c = 0;
    double best = -1.0;
    for (int j = 0; j < nc; ++j)
        if (std::fabs(row[j]) > best) {
            best = std::fabs(row[j]);
            c = j;
        }
    int st = c - bw / 2;
    if (st < 0) st = 0;
    if (st + bw > nc) st = nc - bw;
    return st;
}

template <typename T>
int upload(tpiv_plan* pl, T** dst, const std::vector<T>& src) {
    int rc = pl->alloc(dst, src.size());
    if (rc) return rc;
    hipError_t e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy(banded operator)");
    return TPIV_OK;
}

// The two spline operators in the form predict_mfma.hip takes: per block of 32 fine rows / columns a dense
// K x 32 weight tile over the union of the block's 65-tap bands (zero elsewhere).
int build_banded(tpiv_plan* pl, int p, int nrc, int ncc, int nrf, int ncf, const std::vector<double>& ay,
                 const std::vector<double>& ax) {
    double leak = 0.0;
    // band starts, monotone in the fine index; `leak` = the largest weight left outside a band
    auto starts = [&leak](const std::vector<double>& op, int nf, int nc, int bw) {
        std::vector<int> st(nf);
        for (int f = 0; f < nf; ++f) {
            const double* row = &op[(size_t)f * nc];
            int s = band_start(row, nc, bw);
            if (f > 0 && s < st[f - 1]) s = st[f - 1];
            st[f] = s;
            for (int j = 0; j < nc; ++j)
                if ((j < s || j >= s + bw) && std::fabs(row[j]) > leak) leak = std::fabs(row[j]);
        }
        return st;
    };
    auto tiles32 = [](const std::vector<double>& op, const std::vector<int>& st, int nf, int nc, int bw,
                      std::vector<double>& tile, std::vector<int>& k0v) {
        const int nb = (nf + 31) / 32;
        int K = 0;
        k0v.resize(nb);
        for (int b = 0; b < nb; ++b) {
            const int f1 = std::min(32 * b + 31, nf - 1);
            k0v[b] = st[32 * b];
            K = std::max(K, st[f1] + bw - st[32 * b]);
        }
        K = (K + 7) / 8 * 8;                                     // the kernels take two K-steps of 4 per trip
        tile.assign((size_t)nb * K * 32, 0.0);
        for (int f = 0; f < nf; ++f)
            for (int j = st[f]; j < st[f] + bw; ++j)
                tile[((size_t)(f / 32) * K + (j - k0v[f / 32])) * 32 + f % 32] = op[(size_t)f * nc + j];
        return K;
    };
    const int bwy = nrc < PRED_BW ? nrc : PRED_BW, bwx = ncc < PRED_BW ? ncc : PRED_BW;
    const std::vector<int> sy = starts(ay, nrf, nrc, bwy), sx = starts(ax, ncf, ncc, bwx);
    if (leak > 1e-17) return fail(TPIV_EUNSUPPORTED, "spline operator does not fit the 65-tap band");
    std::vector<double> wy32, ax32;
    std::vector<int> k0y32, k0x32;
    pl->KY[p] = tiles32(ay, sy, nrf, nrc, bwy, wy32, k0y32);
    pl->KX[p] = tiles32(ax, sx, ncf, ncc, bwx, ax32, k0x32);
    int rc = upload(pl, &pl->Wy32[p], wy32);
    if (!rc) rc = upload(pl, &pl->Ax32[p], ax32);
    if (!rc) rc = upload(pl, &pl->k0y32[p], k0y32);
    if (!rc) rc = upload(pl, &pl->k0x32[p], k0x32);
    return rc;
}

}  // namespace

extern "C" {

int tpiv_version(void) { return TPIV_VERSION; }

const char* tpiv_last_error(void) { return g_err.c_str(); }

int tpiv_field_shape(int H, int W, int ws, int ov, int* n_rows, int* n_cols) {
    if (ov >= ws) return fail(TPIV_EINVAL, "Overlap has to be smaller than the window_size");
    if (ws > H || ws > W) return fail(TPIV_EINVAL, "window size cannot be larger than the image");
    field_shape(H, W, ws, ov, n_rows, n_cols);
    return TPIV_OK;
}

int tpiv_coordinates(int H, int W, int ws, int ov, double* x, double* y) {
    int nr, nc;
    int rc = tpiv_field_shape(H, W, ws, ov, &nr, &nc);
    if (rc) return rc;
    coords_1d(W, nc, ws, ov, x);
    coords_1d(H, nr, ws, ov, y);
    return TPIV_OK;
}

int tpiv_spline_matrix(int nc, const double* xc, int nf, const double* xf, double* A) {
    return spline_matrix(nc, xc, nf, xf, A);
}

static int pass1_impl(const uint8_t* a, const uint8_t* b, int batch, int H, int W, int ws, int ov,
                      double val_ratio, int val_win, int precision, double* u, double* v, uint8_t* invalid,
                      void* work, size_t work_bytes, float* dbg_win, float* dbg_corr, void* stream,
                      hipEvent_t* sub_events = nullptr) {
    int rc = check_window(H, W, ws, ov, val_win);
    if (rc) return rc;
    if (!valid_precision(precision)) return fail(TPIV_EINVAL, "precision must be TPIV_PREC_FAST, TPIV_PREC_REFERENCE, TPIV_PREC_F64 or TPIV_PREC_EXACT");
    if (batch <= 0) return TPIV_OK;
    tpiv::PassParams p{};
    p.dbg_win = dbg_win;
    p.dbg_corr = dbg_corr;
    p.A = a;
    p.B = b;
    p.batch = batch;
    p.H = H;
    p.W = W;
    p.ws = ws;
    p.ov = ov;
    field_shape(H, W, ws, ov, &p.n_rows, &p.n_cols);
    p.u = u;
    p.v = v;
    p.val = invalid;
    p.val_ratio = val_ratio;
    p.val_win = val_win;
    p.stamps = g_stamps;
    p.precision = precision;
    p.sub_events = sub_events;
    rc = check_work(work, work_bytes, tpiv::peak_raw_bytes(ws, batch, p.n_rows * p.n_cols, precision));
    if (rc) return rc;
    p.peak_raw = static_cast<float*>(work);
    int n_cu;
    rc = n_cu_of_current_device(&n_cu);
    if (rc) return rc;
    HIP_TRY(tpiv::launch_xcorr(p, tpiv::MODE_PASS1, n_cu, (hipStream_t)stream));
    return TPIV_OK;
}

int tpiv_pass1(const uint8_t* a, const uint8_t* b, int batch, int H, int W, int ws, int ov,
               double val_ratio, int val_win, int precision, double* u, double* v, uint8_t* invalid,
               void* work, size_t work_bytes, void* stream) {
    return pass1_impl(a, b, batch, H, W, ws, ov, val_ratio, val_win, precision, u, v, invalid, work, work_bytes,
                      nullptr, nullptr, stream);
}

size_t tpiv_work_bytes(int H, int W, int ws, int ov, int batch) {
    if (ov >= ws || ws > H || ws > W || ws <= 0 || ov < 0 || batch <= 0 || !supported_ws(ws)) return 0;
    int nr, nc;
    field_shape(H, W, ws, ov, &nr, &nc);
    const size_t a = tpiv::peak_raw_bytes(ws, batch, nr * nc, TPIV_PREC_EXACT);          // the largest of the precisions
    const size_t a1 = tpiv::peak_raw_bytes(ws, batch, nr * nc, TPIV_PREC_REFERENCE);
    const size_t b = tpiv::peak_raw_bytes(ws, batch, nr * nc, TPIV_PREC_FAST, true);     // TPIV_MODE_CWS_FAST: generic kernel
    return std::max(a, std::max(a1, b));
}

int tpiv_predict(int mode, int batch, int nrc, int ncc, int nrf, int ncf, const double* Ay,
                 const double* Ax, const double* u_c, const double* v_c, const uint8_t* invalid_c,
                 double* work, double* u0, double* v0, double* u2, double* v2, void* stream) {
    if (mode != TPIV_MODE_DWS && mode != TPIV_MODE_CWS) return fail(TPIV_EKEY, "unknown multipass mode");
    if (batch <= 0) return TPIV_OK;
    tpiv::PredictParams q{};
    q.batch = batch;
    q.mode = mode;
    q.nrc = nrc;
    q.ncc = ncc;
    q.nrf = nrf;
    q.ncf = ncf;
    q.Ay = Ay;
    q.Ax = Ax;
    q.u_c = u_c;
    q.v_c = v_c;
    q.val_c = invalid_c;
    q.T = work;
    q.u0 = u0;
    q.v0 = v0;
    q.u2 = u2;
    q.v2 = v2;
    HIP_TRY(tpiv::launch_predict(q, (hipStream_t)stream));
    return TPIV_OK;
}

static int run_iter(int mode, int precision, const uint8_t* a, const uint8_t* b, int batch, int H, int W, int ws,
                    int ov, const double* u0, const double* v0, const double* u2, const double* v2,
                    double val_ratio, int val_win, double* u, double* v, uint8_t* invalid, double* du,
                    double* dv, float* dbg_win, float* dbg_corr, void* work, size_t work_bytes, void* stream,
                    const uint8_t* pmask = nullptr) {
    if (mode != TPIV_MODE_DWS && mode != TPIV_MODE_CWS && mode != TPIV_MODE_CWS_FAST)
        return fail(TPIV_EKEY, "unknown multipass mode");
    int rc = check_window(H, W, ws, ov, val_win);
    if (rc) return rc;
    if (mode == TPIV_MODE_CWS_FAST && ws < 2) return fail(TPIV_EINVAL, "window too small");
    if (mode != TPIV_MODE_CWS_FAST && !pmask && (!u2 || !v2)) return fail(TPIV_EINVAL, "tpiv_iter: u2 / v2 missing");
    if (!u0 || !v0) return fail(TPIV_EINVAL, "tpiv_iter: u0 / v0 missing");
    if (!valid_precision(precision)) return fail(TPIV_EINVAL, "precision must be TPIV_PREC_FAST, TPIV_PREC_REFERENCE, TPIV_PREC_F64 or TPIV_PREC_EXACT");
    if (batch <= 0) return TPIV_OK;
    tpiv::PassParams p{};
    p.A = a;
    p.B = b;
    p.batch = batch;
    p.H = H;
    p.W = W;
    p.ws = ws;
    p.ov = ov;
    p.precision = precision;
    field_shape(H, W, ws, ov, &p.n_rows, &p.n_cols);
    p.u0 = u0;
    p.v0 = v0;
    p.u2 = u2;
    p.v2 = v2;
    p.pmask = pmask;           // plan path: raw predictor in u0 / v0 + mask (piv_kernels.h)
    p.u = u;
    p.v = v;
    p.val = invalid;
    p.du = du;
    p.dv = dv;
    p.val_ratio = val_ratio;
    p.val_win = val_win;
    p.dbg_win = dbg_win;
    p.dbg_corr = dbg_corr;
    p.stamps = g_stamps;
    rc = check_work(work, work_bytes,
                    tpiv::peak_raw_bytes(ws, batch, p.n_rows * p.n_cols, TPIV_PREC_FAST, mode == TPIV_MODE_CWS_FAST));
    if (rc) return rc;
    p.peak_raw = static_cast<float*>(work);
    int n_cu;
    rc = n_cu_of_current_device(&n_cu);
    if (rc) return rc;
    HIP_TRY(tpiv::launch_xcorr(p, mode, n_cu, (hipStream_t)stream));
    return TPIV_OK;
}

int tpiv_iter(int mode, const uint8_t* a, const uint8_t* b, int batch, int H, int W, int ws, int ov,
              const double* u0, const double* v0, const double* u2, const double* v2,
              double val_ratio, int val_win, int precision, double* u, double* v, uint8_t* invalid, double* du,
              double* dv, void* work, size_t work_bytes, void* stream) {
    return run_iter(mode, precision, a, b, batch, H, W, ws, ov, u0, v0, u2, v2, val_ratio, val_win, u, v, invalid,
                    du, dv, nullptr, nullptr, work, work_bytes, stream);
}

int tpiv_debug_pass(int mode, int precision, const uint8_t* a, const uint8_t* b, int batch, int H, int W, int ws,
                    int ov, const double* u2, const double* v2, const double* zero, double* u, double* v,
                    uint8_t* invalid, float* win, float* corr, void* work, size_t work_bytes, void* stream) {
    if (mode == 0)
        return pass1_impl(a, b, batch, H, W, ws, ov, 1.2, 3, TPIV_PREC_FAST, u, v, invalid, work, work_bytes, win,
                          corr, stream);
    if (!zero || !u2 || !v2) return fail(TPIV_EINVAL, "tpiv_debug_pass: shifted passes need u2, v2 and a zero field");
    return run_iter(mode, precision, a, b, batch, H, W, ws, ov, zero, zero, u2, v2, 1.2, 3, u, v, invalid, nullptr,
                    nullptr, win, corr, work, work_bytes, stream);
}

int tpiv_debug_peaks(const float* maps, int n_maps, int ws, int planar, double val_ratio, int val_win, double* u,
                     double* v, uint8_t* invalid, void* work, size_t work_bytes, void* stream) {
    if (ws != 8 && ws != 16 && ws != 32 && ws != 64 && ws != 128)
        return fail(TPIV_EUNSUPPORTED, "tpiv_debug_peaks handles 8, 16, 32, 64 and 128 pixel maps");
    if (n_maps <= 0) return TPIV_OK;
    tpiv::PassParams p{};
    p.batch = 1;
    p.ws = ws;
    p.n_rows = n_maps;
    p.n_cols = 1;
    p.u = u;
    p.v = v;
    p.val = invalid;
    p.val_ratio = val_ratio;
    p.val_win = val_win;
    int rc = check_work(work, work_bytes, (size_t)n_maps * 8 * sizeof(float));
    if (rc) return rc;
    p.peak_raw = static_cast<float*>(work);
    HIP_TRY(tpiv::launch_peaks_from_maps(p, maps, n_maps, planar, (hipStream_t)stream));
    return TPIV_OK;
}

int tpiv_plan_create(tpiv_plan** out, int H, int W, int ws, int ov, int n_pass, int mode,
                     double pass_scale, double val_ratio, int val_win, int max_batch, int precision) {
    if (!out) return fail(TPIV_EINVAL, "null plan pointer");
    *out = nullptr;
    if (n_pass < 1) n_pass = 1;     // range(multipass - 1) is empty for multipass <= 1 (B:855)
    if (n_pass > 1 && mode != TPIV_MODE_DWS && mode != TPIV_MODE_CWS)
        return fail(TPIV_EKEY, "unknown multipass mode");
    if (max_batch < 1) return fail(TPIV_EINVAL, "max_batch must be >= 1");
    if (!(pass_scale > 0)) return fail(TPIV_EINVAL, "multipass_scale must be positive");
    if (!valid_precision(precision)) return fail(TPIV_EINVAL, "precision must be TPIV_PREC_FAST, TPIV_PREC_REFERENCE, TPIV_PREC_F64 or TPIV_PREC_EXACT");
    tpiv_plan* pl = new tpiv_plan();
    pl->precision = precision;
    pl->H = H;
    pl->W = W;
    pl->n_pass = n_pass;
    pl->mode = mode;
    pl->max_batch = max_batch;
    pl->val_ratio = val_ratio;
    pl->val_win = val_win;
    int rc = TPIV_OK;
    hipError_t he = hipGetDevice(&pl->device);
    if (he != hipSuccess) {
        delete pl;
        return hip_fail(he, "hipGetDevice");
    }
    int w = ws, o = ov;
    for (int p = 0; p < n_pass && rc == TPIV_OK; ++p) {
        if (p > 0) {
            w = (int)std::floor((double)w / pass_scale);   // int(ws // scale), B:856-857
            o = (int)std::floor((double)o / pass_scale);
        }
        rc = check_window(H, W, w, o, val_win);
        if (rc) break;
        PassGeo g;
        g.ws = w;
        g.ov = o;
        field_shape(H, W, w, o, &g.n_rows, &g.n_cols);
        g.x.resize(g.n_cols);
        g.y.resize(g.n_rows);
        coords_1d(W, g.n_cols, w, o, g.x.data());
        coords_1d(H, g.n_rows, w, o, g.y.data());
        pl->geo.push_back(std::move(g));
    }
    size_t max_fine = 0, max_T = 0;
    pl->Ay.assign(n_pass, nullptr);
    pl->Ax.assign(n_pass, nullptr);
    pl->Wy32.assign(n_pass, nullptr);
    pl->Ax32.assign(n_pass, nullptr);
    pl->k0y32.assign(n_pass, nullptr);
    pl->k0x32.assign(n_pass, nullptr);
    pl->KY.assign(n_pass, 0);
    pl->KX.assign(n_pass, 0);
    pl->u.assign(n_pass, nullptr);
    pl->v.assign(n_pass, nullptr);
    pl->val.assign(n_pass, nullptr);
    for (int p = 0; p < n_pass && rc == TPIV_OK; ++p) {
        const PassGeo& g = pl->geo[p];
        const size_t N = (size_t)g.n_rows * g.n_cols;
        if (p < n_pass - 1) {
            rc = pl->alloc(&pl->u[p], N * max_batch);
            if (!rc) rc = pl->alloc(&pl->v[p], N * max_batch);
            if (!rc) rc = pl->alloc(&pl->val[p], N * max_batch);
        }
        if (p > 0 && rc == TPIV_OK) {
            const PassGeo& c = pl->geo[p - 1];
            std::vector<double> ay((size_t)g.n_rows * c.n_rows), ax((size_t)g.n_cols * c.n_cols);
            rc = spline_matrix(c.n_rows, c.y.data(), g.n_rows, g.y.data(), ay.data());
            if (!rc) rc = spline_matrix(c.n_cols, c.x.data(), g.n_cols, g.x.data(), ax.data());
            if (!rc) rc = pl->alloc(&pl->Ay[p], ay.size());
            if (!rc) rc = pl->alloc(&pl->Ax[p], ax.size());
            if (!rc) {
                he = hipMemcpy(pl->Ay[p], ay.data(), ay.size() * sizeof(double), hipMemcpyHostToDevice);
                if (he == hipSuccess)
                    he = hipMemcpy(pl->Ax[p], ax.data(), ax.size() * sizeof(double), hipMemcpyHostToDevice);
                if (he != hipSuccess) rc = hip_fail(he, "hipMemcpy(spline operators)");
            }
            if (!rc) rc = build_banded(pl, p, c.n_rows, c.n_cols, g.n_rows, g.n_cols, ay, ax);
            if (N > max_fine) max_fine = N;
            const size_t t = (size_t)3 * ((g.n_rows + 31) / 32 * 32) * c.n_cols;     // T1 of the banded path (transposed, padded pitch)
            const size_t t_dense = (size_t)3 * c.n_rows * g.n_cols;
            if (t > max_T) max_T = t;
            if (t_dense > max_T) max_T = t_dense;
        }
    }
    if (rc == TPIV_OK) {
        size_t raw = 0;
        for (size_t i = 0; i < pl->geo.size(); ++i) {
            const PassGeo& g = pl->geo[i];
            const size_t b = tpiv::peak_raw_bytes(g.ws, max_batch, g.n_rows * g.n_cols,
                                                  i == 0 ? precision : (int)TPIV_PREC_FAST);
            if (b > raw) raw = b;
        }
        pl->peak_raw_bytes = raw;
        if (raw) rc = pl->alloc(&pl->peak_raw, raw / sizeof(float) + 64);
        if (rc == TPIV_OK && precision == TPIV_PREC_EXACT && tpiv::exact_refine_size(pl->geo[0].ws))
            rc = pl->alloc(&pl->exact_count, 4);
    }
    if (rc == TPIV_OK && n_pass > 1) {
        rc = pl->alloc(&pl->u0, max_fine * max_batch);
        if (!rc) rc = pl->alloc(&pl->v0, max_fine * max_batch);
        if (!rc) rc = pl->alloc(&pl->pmask, max_fine * max_batch);
        if (!rc) rc = pl->alloc(&pl->T, max_T * max_batch);
    }
    if (rc) {
        const std::string keep = g_err;
        delete pl;
        g_err = keep;
        return rc;
    }
    *out = pl;
    return TPIV_OK;
}

void tpiv_plan_destroy(tpiv_plan* plan) { delete plan; }

int tpiv_plan_n_pass(const tpiv_plan* plan) { return plan ? plan->n_pass : 0; }

int tpiv_plan_pass_geometry(const tpiv_plan* plan, int pass, int* ws, int* ov, int* n_rows, int* n_cols) {
    if (!plan || pass < 0 || pass >= plan->n_pass) return fail(TPIV_EINVAL, "bad plan / pass index");
    const PassGeo& g = plan->geo[pass];
    if (ws) *ws = g.ws;
    if (ov) *ov = g.ov;
    if (n_rows) *n_rows = g.n_rows;
    if (n_cols) *n_cols = g.n_cols;
    return TPIV_OK;
}

const char* tpiv_plan_kernel_name(const tpiv_plan* plan, int pass, char* buf, int len) {
    if (!buf || len <= 0) return buf;
    buf[0] = 0;
    if (!plan || pass < 0 || pass >= plan->n_pass) return buf;
    const int mode = pass == 0 ? (int)tpiv::MODE_PASS1 : plan->mode;
    return tpiv::xcorr_kernel_name(plan->geo[pass].ws, mode, plan->precision, buf, len);
}

int tpiv_plan_exact_fallbacks(tpiv_plan* plan, long long* n_windows) {
    if (!plan || !n_windows) return fail(TPIV_EINVAL, "null argument");
    if (!plan->exact_count)
        return fail(TPIV_EINVAL, "not a TPIV_PREC_EXACT plan with an even first-pass window size from 8 to 128");
    if (plan->last_batch <= 0) return fail(TPIV_EINVAL, "the plan has not run yet");
    HIP_TRY(hipDeviceSynchronize());
    unsigned n = 0;
    HIP_TRY(hipMemcpy(&n, plan->exact_count, sizeof(n), hipMemcpyDeviceToHost));
    *n_windows = (long long)n;
    return TPIV_OK;
}

int tpiv_plan_pass_fields(const tpiv_plan* plan, int pass, double** u, double** v, uint8_t** invalid) {
    if (!plan || pass < 0 || pass >= plan->n_pass - 1)
        return fail(TPIV_EINVAL, "only the passes before the last keep their fields in the plan");
    if (u) *u = plan->u[pass];
    if (v) *v = plan->v[pass];
    if (invalid) *invalid = plan->val[pass];
    return TPIV_OK;
}

static int run_banded_predict(tpiv_plan* plan, int p, int batch, const double* u_c, const double* v_c,
                              const uint8_t* val_c, double* u0, double* v0, double* u2, double* v2,
                              hipStream_t st, uint8_t* mask_out = nullptr) {
    const PassGeo& g = plan->geo[p];
    const PassGeo& c = plan->geo[p - 1];
    tpiv::BandedPredictParams q{};
    q.batch = batch;
    q.mode = plan->mode;
    q.nrc = c.n_rows;
    q.ncc = c.n_cols;
    q.nrf = g.n_rows;
    q.ncf = g.n_cols;
    q.KY = plan->KY[p];
    q.KX = plan->KX[p];
    q.nrfp = (g.n_rows + 31) / 32 * 32;
    q.Wy32 = plan->Wy32[p];
    q.k0y32 = plan->k0y32[p];
    q.Ax32 = plan->Ax32[p];
    q.k0x32 = plan->k0x32[p];
    q.u_c = u_c;
    q.v_c = v_c;
    q.val_c = val_c;
    q.T1 = plan->T;
    q.u0 = u0;
    q.v0 = v0;
    q.u2 = u2;
    q.v2 = v2;
    q.mask_out = mask_out;
    hipError_t he = tpiv::launch_predict_mfma(q, st);
    return he == hipSuccess ? TPIV_OK : hip_fail(he, "launch_predict_mfma");
}

int tpiv_plan_debug_predict(tpiv_plan* plan, int pass, int batch, const double* u_c, const double* v_c,
                            const uint8_t* invalid_c, double* u0, double* v0, double* u2, double* v2,
                            void* stream) {
    if (!plan || pass < 1 || pass >= plan->n_pass) return fail(TPIV_EINVAL, "bad plan / pass index");
    if (batch < 1 || batch > plan->max_batch) return fail(TPIV_EINVAL, "batch exceeds the plan's max_batch");
    return run_banded_predict(plan, pass, batch, u_c, v_c, invalid_c, u0, v0, u2, v2, (hipStream_t)stream);
}

int tpiv_plan_run(tpiv_plan* plan, const uint8_t* a, const uint8_t* b, int batch, double* u, double* v,
                  uint8_t* invalid, void* stream) {
    if (!plan) return fail(TPIV_EINVAL, "null plan");
    if (batch < 0 || batch > plan->max_batch) return fail(TPIV_EINVAL, "batch exceeds the plan's max_batch");
    if (batch == 0) return TPIV_OK;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != plan->device) return fail(TPIV_EINVAL, "plan was created on another device");
    const int last = plan->n_pass - 1;
    plan->last_batch = batch;
    hipStream_t st = (hipStream_t)stream;
    auto mark = [&](int slot, int end) {
        hipEvent_t e = plan->ev(slot, end);
        if (e) (void)hipEventRecord(e, st);
    };
    for (int p = 0; p <= last; ++p) {
        const PassGeo& g = plan->geo[p];
        double* pu = p == last ? u : plan->u[p];
        double* pv = p == last ? v : plan->v[p];
        uint8_t* pval = p == last ? invalid : plan->val[p];
        int rc;
        if (p == 0) {
            mark(0, 0);
            rc = pass1_impl(a, b, batch, plan->H, plan->W, g.ws, g.ov, plan->val_ratio, plan->val_win,
                            plan->precision, pu, pv, pval, plan->peak_raw, plan->peak_raw_bytes, nullptr, nullptr,
                            stream, plan->sub_events());
            mark(0, 1);
            if (!rc && plan->exact_count) {
                const size_t off = tpiv::exact_fallback_count_offset(batch, g.n_rows * g.n_cols);
                HIP_TRY(hipMemcpyAsync(plan->exact_count, reinterpret_cast<const char*>(plan->peak_raw) + off,
                                       sizeof(unsigned), hipMemcpyDeviceToDevice, st));
            }
        } else {
            const PassGeo& c = plan->geo[p - 1];
            mark(2 * p - 1, 0);
            // compact hand-off: raw predictor (u0, v0) + mask byte; zeroing and half shift are formed by the readers
            rc = run_banded_predict(plan, p, batch, plan->u[p - 1], plan->v[p - 1], plan->val[p - 1], plan->u0,
                                    plan->v0, nullptr, nullptr, st, plan->pmask);
            mark(2 * p - 1, 1);
            mark(2 * p, 0);
            if (!rc)
                rc = run_iter(plan->mode, plan->precision, a, b, batch, plan->H, plan->W, g.ws, g.ov, plan->u0, plan->v0,
                              nullptr, nullptr, plan->val_ratio, plan->val_win, pu, pv, pval, nullptr,
                              nullptr, nullptr, nullptr, plan->peak_raw, plan->peak_raw_bytes, stream, plan->pmask);
            mark(2 * p, 1);
        }
        if (rc) return rc;
    }
    if (plan->timing && plan->runs_recorded < 512) plan->runs_recorded++;
    return TPIV_OK;
}

int tpiv_postval(double* u, double* v, const uint8_t* invalid, int batch, int n_rows, int n_cols, uint8_t* cls,
                 int32_t* counts, void* stream) {
    if (batch < 0 || n_rows < 1 || n_cols < 1) return fail(TPIV_EINVAL, "tpiv_postval: empty grid");
    if ((long long)n_rows * n_cols >= (1LL << 31)) return fail(TPIV_EUNSUPPORTED, "field too large");
    if (batch == 0) return TPIV_OK;
    if (!u || !v || !invalid || !cls || !counts) return fail(TPIV_EINVAL, "tpiv_postval: null pointer");
    tpiv::PostvalParams q{};
    q.u = u;
    q.v = v;
    q.invalid = invalid;
    q.cls = cls;
    q.counts = counts;
    q.batch = batch;
    q.n_rows = n_rows;
    q.n_cols = n_cols;
    HIP_TRY(tpiv::launch_postval(q, (hipStream_t)stream));
    return TPIV_OK;
}

int tpiv_postval_compact(const double* u, const double* v, const uint8_t* cls, const int32_t* counts, int batch, int n_rows,
                         int n_cols, int32_t* offsets, int32_t* ring_rc, double* ring_uv, int32_t* hole_rc, void* stream) {
    if (batch < 0 || n_rows < 1 || n_cols < 1) return fail(TPIV_EINVAL, "tpiv_postval_compact: empty grid");
    if ((long long)batch * n_rows * n_cols >= (1LL << 30)) return fail(TPIV_EUNSUPPORTED, "batch of fields too large");
    if (batch == 0) return TPIV_OK;
    if (!u || !v || !cls || !counts || !offsets || !ring_rc || !ring_uv || !hole_rc)
        return fail(TPIV_EINVAL, "tpiv_postval_compact: null pointer");
    HIP_TRY(tpiv::launch_postval_compact(u, v, cls, counts, batch, n_rows, n_cols, offsets, ring_rc, ring_uv, hole_rc,
                                         (hipStream_t)stream));
    return TPIV_OK;
}

int tpiv_finish_fields(const double* u, const double* v, int batch, int n_rows, int n_cols, double scale, double dt,
                       double* fu, double* fv, void* stream) {
    if (batch < 0 || n_rows < 1 || n_cols < 1) return fail(TPIV_EINVAL, "tpiv_finish_fields: empty grid");
    if (batch == 0) return TPIV_OK;
    if (!u || !v || !fu || !fv) return fail(TPIV_EINVAL, "tpiv_finish_fields: null pointer");
    HIP_TRY(tpiv::launch_finish_fields(u, v, batch, n_rows, n_cols, scale, dt, fu, fv, (hipStream_t)stream));
    return TPIV_OK;
}

int tpiv_ensemble_moments(const double* u, const double* v, int n, long long cells, double* out, void* stream) {
    if (n <= 0 || cells <= 0) return fail(TPIV_EINVAL, "tpiv_ensemble_moments: needs at least one field");
    if (!u || !v || !out) return fail(TPIV_EINVAL, "tpiv_ensemble_moments: null pointer");
    HIP_TRY(tpiv::launch_ensemble_moments(u, v, n, cells, out, (hipStream_t)stream));
    return TPIV_OK;
}

int tpiv_bmp_unpack(const uint8_t* raw, const int64_t* desc, const uint8_t* lut, int n_files, int H, int W,
                    uint8_t* out, void* stream) {
    if (n_files < 0 || H <= 0 || W <= 0) return fail(TPIV_EINVAL, "tpiv_bmp_unpack: bad shape");
    if (n_files == 0) return TPIV_OK;
    if (!raw || !desc || !lut || !out) return fail(TPIV_EINVAL, "tpiv_bmp_unpack: null pointer");
    HIP_TRY(tpiv::launch_bmp_unpack(raw, reinterpret_cast<const long long*>(desc), lut, n_files, H, W, out,
                                    (hipStream_t)stream));
    return TPIV_OK;
}

namespace {
// One file into a slot of at most slot_bytes: bytes read, or -1 (cannot open / not a regular file / too big / short read).
int64_t read_one_file(const char* path, uint8_t* out, size_t slot_bytes) {
    const int fd = path ? ::open(path, O_RDONLY | O_CLOEXEC) : -1;
    if (fd < 0) return -1;
    int64_t res = -1;
    struct stat st;
    if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size <= slot_bytes) {
        size_t got = 0;
        const size_t want = (size_t)st.st_size;
        while (got < want) {
            const ssize_t r = ::read(fd, out + got, want - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got == want) res = (int64_t)want;
    }
    ::close(fd);
    return res;
}
}  // namespace

// Read-ahead ring of the file path: the whole run's file list is handed over once, reader threads fill the caller's
// staging buffers batch by batch (in order, up to n_bufs batches ahead of the consumer) and the consumer blocks in
// tpiv_reader_next without holding any interpreter lock.
struct tpiv_reader {
    std::vector<std::string> paths;
    std::vector<uint8_t*> bufs;
    std::vector<int64_t> sizes;          // per file
    std::vector<int> done;               // per batch: files finished
    size_t slot_bytes = 0;
    long files_per_batch = 0, n_batches = 0;
    std::atomic<long> next_file{0};
    long released = 0;                   // batches handed back by the consumer (in order)
    long cursor = 0;                     // next batch tpiv_reader_next hands out
    bool stop = false;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::thread> pool;

    long files_in(long b) const {
        const long n = (long)paths.size();
        return std::min(files_per_batch, n - b * files_per_batch);
    }
    void work() {
        const long n = (long)paths.size();
        for (;;) {
            const long i = next_file.fetch_add(1);
            if (i >= n) return;
            const long b = i / files_per_batch;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || b < released + (long)bufs.size(); });
                if (stop) return;
            }
            uint8_t* out = bufs[b % (long)bufs.size()] + (size_t)(i - b * files_per_batch) * slot_bytes;
            const int64_t got = read_one_file(paths[i].c_str(), out, slot_bytes);
            std::lock_guard<std::mutex> lk(m);
            sizes[i] = got;
            if (++done[b] == files_in(b)) cv.notify_all();
        }
    }
};

tpiv_reader* tpiv_reader_open(const char* const* paths, int64_t n_files, int files_per_batch, uint8_t* const* bufs,
                              int n_bufs, size_t slot_bytes, int n_threads) {
    if (n_files < 0 || files_per_batch < 1 || n_bufs < 1 || slot_bytes == 0 || !bufs || (n_files > 0 && !paths)) {
        fail(TPIV_EINVAL, "tpiv_reader_open: bad arguments");
        return nullptr;
    }
    auto* r = new tpiv_reader;
    for (int64_t i = 0; i < n_files; ++i) r->paths.emplace_back(paths[i] ? paths[i] : "");
    for (int k = 0; k < n_bufs; ++k) {
        if (!bufs[k]) {
            delete r;
            fail(TPIV_EINVAL, "tpiv_reader_open: null staging buffer");
            return nullptr;
        }
        r->bufs.push_back(bufs[k]);
    }
    r->slot_bytes = slot_bytes;
    r->files_per_batch = files_per_batch;
    r->n_batches = (long)((n_files + files_per_batch - 1) / files_per_batch);
    r->sizes.assign((size_t)n_files, -1);
    r->done.assign((size_t)r->n_batches, 0);
    if (n_threads < 1) n_threads = 1;
    for (int t = 0; t < n_threads; ++t) r->pool.emplace_back([r] { r->work(); });
    return r;
}

int tpiv_reader_next(tpiv_reader* r, int* n_files, int* buf_index, int64_t* sizes) {
    if (!r || !n_files || !buf_index || !sizes) return fail(TPIV_EINVAL, "tpiv_reader_next: null argument");
    std::unique_lock<std::mutex> lk(r->m);
    *n_files = 0;
    if (r->cursor >= r->n_batches) return TPIV_OK;
    const long b = r->cursor;
    const long cnt = r->files_in(b);
    if (b >= r->released + (long)r->bufs.size())
        return fail(TPIV_EINVAL, "tpiv_reader_next: every staging buffer is held (release one first)");
    r->cv.wait(lk, [&] { return r->done[b] == cnt; });
    for (long k = 0; k < cnt; ++k) sizes[k] = r->sizes[(size_t)(b * r->files_per_batch + k)];
    *buf_index = (int)(b % (long)r->bufs.size());
    *n_files = (int)cnt;
    r->cursor = b + 1;
    return TPIV_OK;
}

int tpiv_reader_release(tpiv_reader* r) {
    if (!r) return fail(TPIV_EINVAL, "tpiv_reader_release: null reader");
    {
        std::lock_guard<std::mutex> lk(r->m);
        if (r->released >= r->cursor) return fail(TPIV_EINVAL, "tpiv_reader_release: nothing handed out");
        ++r->released;
    }
    r->cv.notify_all();
    return TPIV_OK;
}

void tpiv_reader_close(tpiv_reader* r) {
    if (!r) return;
    {
        std::lock_guard<std::mutex> lk(r->m);
        r->stop = true;
    }
    r->cv.notify_all();
    for (auto& t : r->pool) t.join();
    delete r;
}

int tpiv_read_files(const char* const* paths, int n_files, uint8_t* dst, size_t slot_bytes, int n_threads,
                    int64_t* sizes) {
    if (n_files < 0 || (n_files > 0 && (!paths || !dst || !sizes)) || slot_bytes == 0)
        return fail(TPIV_EINVAL, "tpiv_read_files: bad arguments");
    if (n_files == 0) return TPIV_OK;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_files) n_threads = n_files;
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_files) return;
            sizes[i] = read_one_file(paths[i], dst + (size_t)i * slot_bytes, slot_bytes);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    return TPIV_OK;
}

#ifdef TPIV_STAMPS
// Diagnostic builds only (make stamps): device buffer of 32 uint64 that the tile kernels add their
// per-phase s_memtime deltas to.  The production library has neither this setter nor any other state.
int tpiv_debug_set_stamps(void* dev_buffer) {
    g_stamps = static_cast<unsigned long long*>(dev_buffer);
    return TPIV_OK;
}
#endif

int tpiv_plan_set_timing(tpiv_plan* plan, int enable) {
    if (!plan) return fail(TPIV_EINVAL, "null plan");
    plan->timing = enable != 0;
    plan->runs_recorded = 0;
    return TPIV_OK;
}

int tpiv_plan_get_timing(tpiv_plan* plan, double* avg_ms, int n_slots, int* n_runs) {
    if (!plan || !avg_ms || n_slots != plan->n_slots()) return fail(TPIV_EINVAL, "bad timing query");
    for (int s = 0; s < n_slots; ++s) avg_ms[s] = 0.0;
    const size_t n = plan->runs_recorded;
    for (size_t r = 0; r < n; ++r) {
        for (int s = 0; s < n_slots; ++s) {
            float ms = 0.f;
            HIP_TRY(hipEventSynchronize(plan->events[r][2 * s + 1]));
            HIP_TRY(hipEventElapsedTime(&ms, plan->events[r][2 * s], plan->events[r][2 * s + 1]));
            avg_ms[s] += ms;
        }
    }
    if (n)
        for (int s = 0; s < n_slots; ++s) avg_ms[s] /= (double)n;
    for (double& m : plan->exact_ms) m = 0.0;
    if (plan->exact_count && n) {
        for (size_t r = 0; r < n; ++r) {
            const hipEvent_t* sub = plan->events[r].data() + 2 * n_slots;
            const hipEvent_t chain[5] = {plan->events[r][0], sub[0], sub[1], sub[2], plan->events[r][1]};
            for (int s = 0; s < 4; ++s) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, chain[s], chain[s + 1]));
                plan->exact_ms[s] += ms / (double)n;
            }
        }
    }
    if (n_runs) *n_runs = (int)n;
    plan->runs_recorded = 0;
    return TPIV_OK;
}

int tpiv_plan_exact_timing(const tpiv_plan* plan, double* ms4) {
    if (!plan || !ms4) return fail(TPIV_EINVAL, "null argument");
    if (!plan->exact_count) return fail(TPIV_EINVAL, "not a TPIV_PREC_EXACT plan with an even first-pass window size from 8 to 128");
    for (int s = 0; s < 4; ++s) ms4[s] = plan->exact_ms[s];
    return TPIV_OK;
}

}  // extern "C"
