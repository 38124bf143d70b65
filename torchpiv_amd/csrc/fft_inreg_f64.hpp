// float64 forms of the in-register FFT codelets of fft_inreg.hpp (same structure: compile-time-unrolled
// radix-4 / leading radix-2 decimation-in-frequency butterflies on a register array, twiddles as
// compile-time constants, output digit-reversed at fft_pos(k, N)).  Used by the float64 first pass
// (xcorr_f64.hip), where a 64-point line is split over two lanes and every transform is a 32-point
// (the last one a 16-point) codelet.  Compiles as plain host C++ too (tests/test_host_logic.py).
#pragma once
#include "fft_inreg.hpp"

namespace tpiv {

#include "twiddles_f64.inc"

struct cd {
    double x, y;
};

TPIV_HD cd cadd(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
TPIV_HD cd csub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }

// a * exp(-DIR * 2*pi*i * K / N)   (DIR = +1 forward, -1 inverse)
template <int K, int N, int DIR>
TPIV_HD cd twmul_d(cd a) {
    constexpr int idx = (((K % N) + N) % N * (128 / N)) % 128;
    if constexpr (idx == 0) {
        return a;
    } else if constexpr (idx == 32) {
        return DIR > 0 ? cd{a.y, -a.x} : cd{-a.y, a.x};
    } else if constexpr (idx == 64) {
        return cd{-a.x, -a.y};
    } else if constexpr (idx == 96) {
        return DIR > 0 ? cd{-a.y, a.x} : cd{a.y, -a.x};
    } else {
        constexpr double c = TWD_COS[idx];
        constexpr double s = DIR > 0 ? -TWD_SIN[idx] : TWD_SIN[idx];
        return cd{a.x * c - a.y * s, a.x * s + a.y * c};
    }
}

template <int N, int OFF, int DIR, int TOTAL>
struct FFTStageD {
    static TPIV_HD void run(cd (&x)[TOTAL]) {
        if constexpr (N == 2) {
            cd a = x[OFF], b = x[OFF + 1];
            x[OFF] = cadd(a, b);
            x[OFF + 1] = csub(a, b);
        } else if constexpr (radix2_first(N)) {
            static_for<0, N / 2>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                cd a = x[OFF + j], b = x[OFF + j + N / 2];
                x[OFF + j] = cadd(a, b);
                x[OFF + j + N / 2] = twmul_d<j, N, DIR>(csub(a, b));
            });
            FFTStageD<N / 2, OFF, DIR, TOTAL>::run(x);
            FFTStageD<N / 2, OFF + N / 2, DIR, TOTAL>::run(x);
        } else if constexpr (N >= 4) {
            static_for<0, N / 4>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                cd a = x[OFF + j], b = x[OFF + j + N / 4], c = x[OFF + j + N / 2], d = x[OFF + j + 3 * N / 4];
                cd t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), bd = csub(b, d);
                cd t3 = DIR > 0 ? cd{bd.y, -bd.x} : cd{-bd.y, bd.x};      // -+ i (b - d)
                x[OFF + j] = cadd(t0, t2);
                x[OFF + j + N / 4] = twmul_d<j, N, DIR>(cadd(t1, t3));
                x[OFF + j + N / 2] = twmul_d<2 * j, N, DIR>(csub(t0, t2));
                x[OFF + j + 3 * N / 4] = twmul_d<3 * j, N, DIR>(csub(t1, t3));
            });
            if constexpr (N > 4) {
                FFTStageD<N / 4, OFF, DIR, TOTAL>::run(x);
                FFTStageD<N / 4, OFF + N / 4, DIR, TOTAL>::run(x);
                FFTStageD<N / 4, OFF + N / 2, DIR, TOTAL>::run(x);
                FFTStageD<N / 4, OFF + 3 * N / 4, DIR, TOTAL>::run(x);
            }
        }
    }
};

// Unnormalised N-point transform of x[0..N) in place; bin k ends at x[fft_pos(k, N)].
template <int N, int DIR>
TPIV_HD void fft_inreg_d(cd (&x)[N]) {
    FFTStageD<N, 0, DIR, N>::run(x);
}

// First half of c2r_inreg (fft_inreg.hpp) in float64, in place: Y[0..M] (M = N/2, natural order, Hermitian half
// spectrum of a real N-point row) -> h[0..M) with  z = IDFT_M(h),  z[m] = r[2m] + i r[2m+1];  Y[M] is consumed.
template <int N>
TPIV_HD void c2r_pre_d(cd (&Y)[N / 2 + 1]) {
    constexpr int M = N / 2;
    const cd y0 = Y[0], yM = Y[M];
    Y[0] = cd{y0.x + yM.x, y0.x - yM.x};
    static_for<1, M / 2>([&](auto kc) TPIV_LAMBDA_INLINE {
        constexpr int k = decltype(kc)::value;
        const cd A = Y[k], B = Y[M - k];
        const cd S{A.x + B.x, A.y - B.y};
        const cd T = twmul_d<k, N, -1>(cd{A.x - B.x, A.y + B.y});
        Y[k] = cd{S.x - T.y, S.y + T.x};
        Y[M - k] = cd{S.x + T.y, T.x - S.y};
    });
    Y[M / 2] = cd{2.0 * Y[M / 2].x, -2.0 * Y[M / 2].y};
}

}  // namespace tpiv
