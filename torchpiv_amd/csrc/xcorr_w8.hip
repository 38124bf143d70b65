// 8x8 interrogation windows, ONE WINDOW PER LANE (configs[3]'s last pass: 1 046 529 windows per 4096^2 pair).
//
// The tile kernel of xcorr_tile.hpp gives a lane one image row of a window: for 8-pixel rows the per-lane
// fixed costs (row classification, weight staging, queue, the cross-lane reductions and LDS transposes
// between 8-point transforms) were 36 % of its instructions -- 96 wave-instructions per window.  Here a lane
// owns a whole window: the 8 x 8 packed tile a + i b lives in 128 VGPRs, BOTH transforms, the cross-spectrum
// (partner bin in the same lane), the inverse (Hermitian: columns 0 and 4 packed into one complex transform,
// rows by a c2r codelet) and the peak scans are straight-line code on registers with compile-time indices.
// No LDS transposes, no cross-lane traffic, no barriers; LDS only holds the finished map for the five
// data-dependent neighbour look-ups of the sub-pixel fit, and parks the samples of the rare windows that
// need the per-pixel staging path (flat-index clamps at the frame ends, "integral coordinate" quirk).
// Adjacent lanes are adjacent windows (4 px apart), so their row loads coalesce.
//
// Same reference semantics as the tile kernel (PIVbackend.py:147-216 staging, 249-257 correlation,
// 346-358 / 381-392 / 518 peak rules); same 8-float record for finalize_kernel.
#include "xcorr_tile.hpp"

namespace tpiv {

namespace {

constexpr int W8 = 8;
template <int K>
inline constexpr int P8 = fft_pos(K, 8);

// one frame of the CWS patch: rows py .. py+8, columns px .. px+8 of frame f (all inside the frame), resampled
// with the reference's float32 weights (B:162-193).  IS_B: frame b (imaginary parts).  All indices are
// compile-time constants (static_for): the patch and the samples stay in registers.
template <bool FAST, bool IS_B>
__device__ __forceinline__ void stage_cws_fast(const uint8_t* __restrict__ f, int W, int q0, int gx, int gy, float vx,
                                               float vy, cf (&x)[8][8]) {
#pragma clang fp contract(off)
    uint32_t rows[9][3];
    static_for<0, 9>([&](auto rc) TPIV_LAMBDA_INLINE {
        constexpr int r = decltype(rc)::value;
        load_dwords<3>(f + q0 + r * W, rows[r]);
    });
    float wxu[8], wxd[8];
    static_for<0, 8>([&](auto kc) TPIV_LAMBDA_INLINE {
        constexpr int k = decltype(kc)::value;
        const float nx = (float)(gx + k) + vx;
        wxu[k] = ceilf(nx) - nx;
        wxd[k] = nx - floorf(nx);
    });
    float h0[8], h1[8];          // FAST: x-lerped rows r and r + 1
    if constexpr (FAST) {
        static_for<0, 8>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            h0[k] = __builtin_fmaf(byte_f<k + 1, 3>(rows[0]), wxd[k], byte_f<k, 3>(rows[0]) * wxu[k]);
        });
    }
    static_for<0, 8>([&](auto rc) TPIV_LAMBDA_INLINE {
        constexpr int r = decltype(rc)::value;
        const float ny = (float)(gy + r) + vy;
        const float wyu = ceilf(ny) - ny, wyd = ny - floorf(ny);
        static_for<0, 8>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            float v;
            if constexpr (FAST) {
                h1[k] = __builtin_fmaf(byte_f<k + 1, 3>(rows[r + 1]), wxd[k], byte_f<k, 3>(rows[r + 1]) * wxu[k]);
                v = __builtin_fmaf(h1[k], wyd, h0[k] * wyu);
                h0[k] = h1[k];
            } else {
                v = bilerp_ref(byte_f<k, 3>(rows[r]), byte_f<k + 1, 3>(rows[r]), byte_f<k, 3>(rows[r + 1]),
                               byte_f<k + 1, 3>(rows[r + 1]), wxu[k], wxd[k], wyu, wyd, false);
            }
            if constexpr (IS_B) x[r][k].y = v;
            else x[r][k].x = v;
        });
    });
}

// the per-pixel form of the tile kernel's slow path, for one frame: values go to LDS ([pixel][lane]) in a rolled
// loop and come back with compile-time indices
template <int MODE, bool IS_B>
__device__ __forceinline__ void stage_slow(const PassParams& p, const uint8_t* __restrict__ f, int gx, int gy, float vx,
                                           float vy, long long sh, float* lds, int lane, cf (&x)[8][8]) {
    const int HW = p.H * p.W;
    for (int i = 0; i < 64; ++i) {
        const int r = i >> 3, k = i & 7;
        float v;
        if constexpr (MODE == MODE_DWS) {
            v = fetch_clamped_t(f, (long long)(gy + r) * p.W + gx + k + sh, HW);
        } else {
            const float nx = (float)(gx + k) + vx, ny = (float)(gy + r) + vy;
            const float ux_f = ceilf(nx), dx_f = floorf(nx), uy_f = ceilf(ny), dy_f = floorf(ny);
            const int ux = f2i_sat_t(ux_f), dx = f2i_sat_t(dx_f), uy = f2i_sat_t(uy_f), dy = f2i_sat_t(dy_f);
            v = bilerp_ref(fetch_clamped_t(f, (long long)dy * p.W + dx, HW), fetch_clamped_t(f, (long long)dy * p.W + ux, HW),
                           fetch_clamped_t(f, (long long)uy * p.W + dx, HW), fetch_clamped_t(f, (long long)uy * p.W + ux, HW),
                           ux_f - nx, nx - dx_f, uy_f - ny, ny - dy_f, (ux == dx) || (uy == dy));
        }
        lds[i * 64 + lane] = v;
    }
    static_for<0, 64>([&](auto ic) TPIV_LAMBDA_INLINE {
        constexpr int i = decltype(ic)::value;
        if constexpr (IS_B) x[i >> 3][i & 7].y = lds[i * 64 + lane];
        else x[i >> 3][i & 7].x = lds[i * 64 + lane];
    });
}

// ---- peak analysis of one window per lane (B:346-358, 381-392, 518): c[y][x] in un-shifted coordinates.
// SCALED: the constant factor of the map is applied here (see xcorr_tile.hpp peak_analysis).
template <bool FASTN>
__device__ __forceinline__ void w8_peak(const PassParams& p, const float (&c)[8][8], float* lds, int lane, bool active,
                                        bool dead, size_t fidx, float end_scale) {
    // ---------------- peak analysis in fftshift coordinates: flat index f = y' * 8 + x', y' = (y + 4) % 8
    float cmin = 3.4e38f, raw_max = -3.4e38f;
    static_for<0, 64>([&](auto ic) TPIV_LAMBDA_INLINE {
        constexpr int i = decltype(ic)::value;
        cmin = fminf(cmin, c[i >> 3][i & 7]);
        raw_max = fmaxf(raw_max, c[i >> 3][i & 7]);
    });
    const float ncs = -(cmin * end_scale);
    auto shifted = [&](float v_) TPIV_LAMBDA_INLINE {            // B:518 corr - min; B:381 corr += eps
        if constexpr (FASTN) return __fadd_rn(fmaf(v_, end_scale, ncs), 1e-7f);
        else return __fadd_rn(__fsub_rn(v_, cmin), 1e-7f);
    };
    const float gmax = shifted(raw_max);                       // (monotonic: the maximum of the shifted map)
    float v[64];                                               // v[f], shifted flat order
    static_for<0, 64>([&](auto fc) TPIV_LAMBDA_INLINE {
        constexpr int f = decltype(fc)::value;
        constexpr int y = ((f >> 3) + 4) % 8, xx = ((f & 7) + 4) % 8;
        v[f] = shifted(c[y][xx]);
        lds[f * 64 + lane] = v[f];
    });
    // arg-max = FIRST flat index holding the maximum (B:383)
    int m = 63;
    static_for<0, 64>([&](auto fc) TPIV_LAMBDA_INLINE {
        constexpr int f = 63 - decltype(fc)::value;
        m = (v[f] == gmax) ? f : m;
    });
    // second peak: maximum outside {clamp(m + i + 8 j), |i|, |j| <= wv} (B:346-358); mask of the flat indices
    const int wv = p.val_win;
    unsigned long long ex = 0ull;
    {
        const unsigned long long run = (2 * wv + 1) >= 64 ? ~0ull : ((1ull << (2 * wv + 1)) - 1ull);
        for (int j = -wv; j <= wv; ++j) {
            const int s = m - wv + 8 * j;                      // first index of this run
            if (s >= 64 || s + 2 * wv < 0) continue;
            ex |= s >= 0 ? (run << s) : (run >> (-s));
        }
        if (m - wv - 8 * wv <= 0) ex |= 1ull;                  // clamp to 0
        if (m + wv + 8 * wv >= 63) ex |= 1ull << 63;           // clamp to k*d - 1
    }
    const int exl = (int)(unsigned)ex, exh = (int)(unsigned)(ex >> 32);
    int smax = 0;                                              // float bits; positive floats order like ints
    static_for<0, 64>([&](auto fc) TPIV_LAMBDA_INLINE {
        constexpr int f = decltype(fc)::value;
        const int kill = __builtin_amdgcn_sbfe(f < 32 ? exl : exh, f & 31, 1);
        const int cand = __float_as_int(v[f]) | kill;
        smax = cand > smax ? cand : smax;
    });
    const float second_v = smax > 0 ? __int_as_float(smax) : gmax;
    // the five values of the fit (flat-index neighbours and fix-ups of B:385-392) from the parked map
    int left = m + 1, right = m - 1, top = m + 8, bot = m - 8;
    if (left >= 63) left = m;
    if (right <= 0) right = m;
    if (top >= 63) top = m;
    if (bot <= 0) bot = m;
    if (active) {
        float4 r0, r1;
        r0.x = gmax;
        r0.y = lds[left * 64 + lane];
        r0.z = lds[right * 64 + lane];
        r0.w = lds[top * 64 + lane];
        r1.x = lds[bot * 64 + lane];
        r1.y = second_v;
        r1.z = __int_as_float(m);
        r1.w = __int_as_float(dead ? 1 : 0);
        float4* out = reinterpret_cast<float4*>(p.peak_raw + fidx * 8);
        out[0] = r0;
        out[1] = r1;
        if (p.dbg_corr != nullptr) {
            float* d = p.dbg_corr + fidx * 64;
            static_for<0, 64>([&](auto fc) TPIV_LAMBDA_INLINE { d[decltype(fc)::value] = v[decltype(fc)::value]; });
        }
    }
}

template <int MODE, bool FAST>
__global__ __launch_bounds__(64, 2) void xcorr_w8_kernel(PassParams p) {
    __shared__ float lds[64 * 64];
    const int lane = threadIdx.x;
    const int N = p.n_rows * p.n_cols;
    const int groups = (N + 63) / 64;
    const long long items = (long long)p.batch * groups;
    const int st = p.ws - p.ov;
    const int HW = p.H * p.W;
    // XCD-aware static order: workgroups b, b+8, ... share an XCD and walk one contiguous run of windows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long long chunk = (items + 7) / 8;
    const long long lo = (long long)xcd * chunk;
    const long long hi = lo + chunk < items ? lo + chunk : items;
    constexpr bool FASTN = FAST && MODE != MODE_PASS1;        // (pass 1 keeps the mean removal in front, as in the tile kernel)

    for (long long item = lo + slot; item < hi; item += per_xcd) {
        const int pair = (int)(item / groups), gi = (int)(item % groups);
        const int win_raw = gi * 64 + lane;
        const bool active = win_raw < N;
        const int win = active ? win_raw : N - 1;
        const int wrow = fast_div(win, p.ncols_magic, p.ncols_shift);
        const int y0 = wrow * st, x0 = (win - wrow * p.n_cols) * st;
        const size_t fidx = (size_t)pair * N + win;
        const uint8_t* __restrict__ fa = p.A + (size_t)pair * HW;
        const uint8_t* __restrict__ fb = p.B + (size_t)pair * HW;
        const int base = y0 * p.W + x0;

        cf x[8][8];
        // ---------------- staging
        if constexpr (MODE == MODE_PASS1) {
            static_for<0, 8>([&](auto rc) TPIV_LAMBDA_INLINE {
                constexpr int r = decltype(rc)::value;
                uint32_t da[2], db[2];
                load_dwords<2>(fa + base + r * p.W, da);
                load_dwords<2>(fb + base + r * p.W, db);
                static_for<0, 8>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    x[r][k].x = byte_f<k, 2>(da);
                    x[r][k].y = byte_f<k, 2>(db);
                });
            });
        } else if constexpr (MODE == MODE_DWS) {
            // integer shift on the FLAT index (B:213-215): a at idx - (vy W + vx), b at idx + (...), clamped per pixel
            double sx, sy;
            pred_half_shift<MODE_DWS>(p, fidx, sx, sy);
            const long long sh = (long long)sy * p.W + (long long)sx;
            const long long qa = (long long)base - sh, qb = (long long)base + sh;
            const long long last = (long long)7 * p.W + 7;
            const bool fast = qa >= 0 && qb >= 0 && qa + last <= (long long)HW - 1 && qb + last <= (long long)HW - 1;
            if (__all(fast)) {
                static_for<0, 8>([&](auto rc) TPIV_LAMBDA_INLINE {
                    constexpr int r = decltype(rc)::value;
                    uint32_t da[2], db[2];
                    load_dwords<2>(fa + qa + r * p.W, da);
                    load_dwords<2>(fb + qb + r * p.W, db);
                    static_for<0, 8>([&](auto kc) TPIV_LAMBDA_INLINE {
                        constexpr int k = decltype(kc)::value;
                        x[r][k].x = byte_f<k, 2>(da);
                        x[r][k].y = byte_f<k, 2>(db);
                    });
                });
            } else {
                stage_slow<MODE_DWS, false>(p, fa, x0, y0, 0.f, 0.f, -sh, lds, lane, x);
                stage_slow<MODE_DWS, true>(p, fb, x0, y0, 0.f, 0.f, sh, lds, lane, x);
            }
        } else {
            float vx, vy;                                                        // the float32 cast of B:714-715
            pred_half_shift_cws_f32(p, fidx, vx, vy);
            // Fast path: floor(float(g) + v) == g + floor(v) and no exactly integral coordinate (the "nearest
            // sample" quirk, B:170/193) for every column and row of the window -- checked coordinate by coordinate
            // with the reference's own float32 sums (a threshold on frac(v) wide enough for float32 rounding at
            // W = 4096 would send 0.8 % of the windows, i.e. 40 % of the 64-window wavefronts, down the per-pixel
            // path) -- and both 9 x 9 source patches inside the frame with room for the 12-byte row loads.
            const float fvx = floorf(vx), fvy = floorf(vy);
            const int ivx = f2i_sat_t(fminf(fmaxf(fvx, -(float)p.W), (float)p.W));
            const int ivy = f2i_sat_t(fminf(fmaxf(fvy, -(float)p.H), (float)p.H));
            bool coords_ok = true;
            static_for<0, 8>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                // frame a samples at g - v (floor = g - floor(v) - 1), frame b at g + v (floor = g + floor(v))
                const float xa = (float)(x0 + k) - vx, xb = (float)(x0 + k) + vx;
                const float ya = (float)(y0 + k) - vy, yb = (float)(y0 + k) + vy;
                const float fxa = floorf(xa), fxb = floorf(xb), fya = floorf(ya), fyb = floorf(yb);
                coords_ok = coords_ok && fxa == (float)(x0 + k - ivx - 1) && fxb == (float)(x0 + k + ivx) &&
                            fya == (float)(y0 + k - ivy - 1) && fyb == (float)(y0 + k + ivy) && fxa != xa && fxb != xb &&
                            fya != ya && fyb != yb;
            });
            // frame a uses -v: floor(-v) = -floor(v) - 1 when frac != 0
            const long long qa0 = (long long)(y0 - ivy - 1) * p.W + (x0 - ivx - 1);
            const long long qb0 = (long long)(y0 + ivy) * p.W + (x0 + ivx);
            const long long lastb = (long long)8 * p.W + 12;
            const bool fast = coords_ok && fabsf(vx) < (float)p.W && fabsf(vy) < (float)p.H && qa0 >= 0 && qb0 >= 0 &&
                              qa0 + lastb <= (long long)HW && qb0 + lastb <= (long long)HW;
            // (a patch that crosses a row end wraps into the neighbouring image row exactly as the reference's
            //  flat index does, B:177-180: only the two ends of the frame need the clamp of the slow path)
            if (__all(fast)) {           // (idle lanes repeat the last window: same verdict as its own lane)
                stage_cws_fast<FAST, false>(fa, p.W, (int)qa0, x0, y0, -vx, -vy, x);
                stage_cws_fast<FAST, true>(fb, p.W, (int)qb0, x0, y0, vx, vy, x);
            } else {
                stage_slow<MODE_CWS, false>(p, fa, x0, y0, -vx, -vy, 0, lds, lane, x);
                stage_slow<MODE_CWS, true>(p, fb, x0, y0, vx, vy, 0, lds, lane, x);
            }
        }
        if (p.dbg_win != nullptr && active) {
            float* d = p.dbg_win + fidx * 2 * 64;
            static_for<0, 64>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int i = decltype(ic)::value;
                d[i] = x[i >> 3][i & 7].x;
                d[64 + i] = x[i >> 3][i & 7].y;
            });
        }

        // ---------------- mean handling (see xcorr_tile.hpp): pass 1 divides by the mean (B:513-514); the
        //                  shifted passes only remove it -- FAST: by zeroing the DC bin of the cross-spectrum
        bool dead = false;
        float end_scale = 1.0f;
        if constexpr (!FASTN) {
            float sa = 0.f, sb = 0.f;
            static_for<0, 64>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int i = decltype(ic)::value;
                sa += x[i >> 3][i & 7].x;
                sb += x[i >> 3][i & 7].y;
            });
            const float ma = sa * (1.0f / 64), mb = sb * (1.0f / 64);
            float ka = 1.f, kb = 1.f;
            if constexpr (MODE == MODE_PASS1) {
                dead = (sa == 0.f) || (sb == 0.f);
                ka = dead ? 0.f : 1.0f / ma;
                kb = dead ? 0.f : 1.0f / mb;
            }
            constexpr float PRE = 0.5f / 8.0f;             // 1/n^2 and the 1/4 of the packed spectrum, exact
            const float kas = ka * PRE, kbs = kb * PRE, oas = -ma * ka * PRE, obs = -mb * kb * PRE;
            static_for<0, 64>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int i = decltype(ic)::value;
                x[i >> 3][i & 7].x = fmaf(x[i >> 3][i & 7].x, kas, oas);
                x[i >> 3][i & 7].y = fmaf(x[i >> 3][i & 7].y, kbs, obs);
            });
        } else {
            end_scale = 0.25f / 64.0f;
        }

        // ---------------- forward 2-D transform: rows, then columns (register renames); Z(ky, kx) at x[P8<ky>][P8<kx>]
        static_for<0, 8>([&](auto rc) TPIV_LAMBDA_INLINE { fft_inreg<8, 1>(x[decltype(rc)::value]); });
        static_for<0, 8>([&](auto cc) TPIV_LAMBDA_INLINE {
            constexpr int c = decltype(cc)::value;
            cf col[8];
            static_for<0, 8>([&](auto yc) TPIV_LAMBDA_INLINE { col[decltype(yc)::value] = x[decltype(yc)::value][c]; });
            fft_inreg<8, 1>(col);
            static_for<0, 8>([&](auto yc) TPIV_LAMBDA_INLINE { x[decltype(yc)::value][c] = col[decltype(yc)::value]; });
        });

        // ---------------- cross-spectrum, partner bin in the same lane.  With Z(k) = a + ib, Z(-k) = c + id:
        //                  4 P = 2 (a d + b c) + i ((c^2 - a^2) + (d^2 - b^2));  P(-k) = conj P(k)
        static_for<0, 64>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int ky = decltype(ic)::value >> 3, kx = decltype(ic)::value & 7;
            constexpr int nky = (8 - ky) % 8, nkx = (8 - kx) % 8;
            if constexpr (ky * 8 + kx <= nky * 8 + nkx) {
                const cf zk = x[P8<ky>][P8<kx>], zm = x[P8<nky>][P8<nkx>];
                cf pr;
                pr.x = (zk.x * zm.y + zk.y * zm.x) * 2.0f;
                pr.y = (zm.x * zm.x - zk.x * zk.x) + (zm.y * zm.y - zk.y * zk.y);
                if constexpr (FASTN && ky == 0 && kx == 0) pr = cf{0.f, 0.f};        // the window means, removed here
                x[P8<ky>][P8<kx>] = pr;
                if constexpr (ky != nky || kx != nkx) x[P8<nky>][P8<nkx>] = cf{pr.x, -pr.y};
            }
        });

        // ---------------- inverse: columns kx = 0 and 4 (real results) packed into one transform, kx = 1..3, then a
        //                  c2r transform per row (the map is real: columns 5..7 are mirrors)
        cf g[5][8];            // g[kx][y]; kx = 0: r0(y) + i r4(y)
        static_for<0, 4>([&](auto jc) TPIV_LAMBDA_INLINE {
            constexpr int j = decltype(jc)::value;               // 0: packed (0, 4); 1..3: kx = j
            cf t[8];
            static_for<0, 8>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int ky = decltype(kc)::value;
                if constexpr (j == 0) {
                    const cf p0 = x[P8<ky>][P8<0>], p4 = x[P8<ky>][P8<4>];
                    t[ky] = cf{p0.x - p4.y, p0.y + p4.x};          // P(ky, 0) + i P(ky, 4)
                } else {
                    t[ky] = x[P8<ky>][P8<j>];
                }
            });
            fft_inreg<8, -1>(t);
            static_for<0, 8>([&](auto yc) TPIV_LAMBDA_INLINE { g[j][decltype(yc)::value] = t[P8<decltype(yc)::value>]; });
        });
        float c[8][8];         // corr(y, x), un-shifted coordinates
        static_for<0, 8>([&](auto yc) TPIV_LAMBDA_INLINE {
            constexpr int y = decltype(yc)::value;
            cf Y[5] = {cf{g[0][y].x, 0.f}, g[1][y], g[2][y], g[3][y], cf{g[0][y].y, 0.f}};
            cf h[4];
            c2r_inreg<8>(Y, h);
            static_for<0, 8>([&](auto xc) TPIV_LAMBDA_INLINE {
                constexpr int xx = decltype(xc)::value;
                c[y][xx] = (xx & 1) ? h[FFT_POS<xx / 2, 4>].y : h[FFT_POS<xx / 2, 4>].x;
            });
        });

        w8_peak<FASTN>(p, c, lds, lane, active, dead, fidx, end_scale);
    }
}

// test hook: hand-made maps [n_maps, 8, 8] float32 in fftshift layout through w8_peak (one map per lane)
__global__ __launch_bounds__(64, 2) void peak_debug_w8_kernel(PassParams p, const float* maps, int n_maps) {
    __shared__ float lds[64 * 64];
    const int lane = threadIdx.x;
    const int win_raw = blockIdx.x * 64 + lane;
    const bool active = win_raw < n_maps;
    const int win = active ? win_raw : n_maps - 1;
    float c[8][8];
    static_for<0, 64>([&](auto ic) TPIV_LAMBDA_INLINE {
        constexpr int y = decltype(ic)::value >> 3, xx = decltype(ic)::value & 7;
        c[y][xx] = maps[(size_t)win * 64 + ((y + 4) % 8) * 8 + (xx + 4) % 8];
    });
    w8_peak<false>(p, c, lds, lane, active, false, (size_t)win, 1.0f);
}

template <int MODE>
hipError_t launch_w8_mode(const PassParams& p_in, int n_cu, hipStream_t stream) {
    PassParams p = p_in;
    const int N = p.n_rows * p.n_cols;
    const long long groups = (N + 63) / 64;
    const long long items = (long long)p.batch * groups;
    if (items <= 0 || (long long)p.batch * N >= (1ll << 31)) return hipErrorInvalidValue;
    fast_div_setup((unsigned)p.n_cols, p.ncols_magic, p.ncols_shift);
    long long blocks = items < (long long)n_cu * 16 ? items : (long long)n_cu * 16;
    blocks = (blocks + 7) / 8 * 8;
    if (p.precision != 0 && MODE != MODE_PASS1)
        hipLaunchKernelGGL((xcorr_w8_kernel<MODE, false>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL((xcorr_w8_kernel<MODE, true>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_peak_debug_w8(const PassParams& p, const float* maps, int n_maps, hipStream_t stream) {
    hipLaunchKernelGGL(peak_debug_w8_kernel, dim3((n_maps + 63) / 64), dim3(64), 0, stream, p, maps, n_maps);
    return hipGetLastError();
}

hipError_t launch_xcorr_w8(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    switch (mode) {
        case MODE_PASS1: return launch_w8_mode<MODE_PASS1>(p, n_cu, stream);
        case MODE_DWS: return launch_w8_mode<MODE_DWS>(p, n_cu, stream);
        case MODE_CWS: return launch_w8_mode<MODE_CWS>(p, n_cu, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace tpiv
