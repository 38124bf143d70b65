// In-register complex FFT codelets for window sizes that are not powers of two (42, 28, 18, 12, 56, 24, 14, ...: what
// `multipass_scale` values other than 2 produce, PIVbackend.py:846-858).
//
// One lane holds a whole line of N = n1 * n2 complex samples (2 <= n1 <= n2 <= 8) in registers and transforms it in place
// with two steps of small DFTs -- n2 butterflies of radix n1 over the samples n2 a + b, a twiddle w_N^(b k1), n1 butterflies
// of radix n2 -- all indices and constants compile-time, so the array lives in VGPRs (fft_inreg.hpp does the same for powers
// of two with radix-4 stages).  Output bin k = k1 + n1 k2 sits at slot n2 k1 + k2: mixed_pos(k, N).
// The odd radices use the symmetry of their roots (inputs j and R - j paired: (R - 1)^2 / 2 real multiplies per butterfly
// instead of (R - 1)^2 complex ones), 6 = 2 x 3 and 8 = 2 x 4 by one decimation step.
//
// Compiles as plain host C++ too: tests/host/fft_mixed_harness.cpp checks every supported size against numpy.fft.
#pragma once
#include "fft_inreg.hpp"

namespace tpiv {
namespace fmx {

// ---- constexpr sin / cos of 2 pi k / n (double precision series on [0, pi/4] behind an exact-as-it-gets reduction)
constexpr double PI_D = 3.14159265358979323846264338327950288;
constexpr double series_sin(double x) {
    const double x2 = x * x;
    double term = x, sum = x;
    for (int i = 1; i < 14; ++i) {
        term *= -x2 / (double)((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double series_cos(double x) {
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < 14; ++i) {
        term *= -x2 / (double)((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
struct SinCos {
    double s, c;
};
constexpr SinCos sincos_2pi(int k, int n) {
    k %= n;
    if (k < 0) k += n;
    // quarter turns are exact: 4 k = q n + r, angle = q pi/2 + (pi/2) r / n with 0 <= r < n
    const int q = (4 * k) / n, r = (4 * k) % n;
    double s = 0, c = 0;
    if (2 * r <= n) {                    // (pi/2) r / n <= pi/4
        const double t = PI_D * 0.5 * (double)r / (double)n;
        s = series_sin(t);
        c = series_cos(t);
    } else {                             // sin(t) = cos(pi/2 - t), cos(t) = sin(pi/2 - t)
        const double t = PI_D * 0.5 * (double)(n - r) / (double)n;
        s = series_cos(t);
        c = series_sin(t);
    }
    switch (q & 3) {
        case 0: return SinCos{s, c};
        case 1: return SinCos{c, -s};
        case 2: return SinCos{-s, -c};
        default: return SinCos{-c, s};
    }
}

// a * exp(-DIR * 2 pi i K / N)
template <int K, int N, int DIR>
TPIV_HD cf twm(cf a) {
    constexpr int k = ((K % N) + N) % N;
    if constexpr (k == 0) {
        return a;
    } else if constexpr (4 * k == N) {
        return DIR > 0 ? cf{a.y, -a.x} : cf{-a.y, a.x};
    } else if constexpr (2 * k == N) {
        return cf{-a.x, -a.y};
    } else if constexpr (4 * k == 3 * N) {
        return DIR > 0 ? cf{-a.y, a.x} : cf{a.y, -a.x};
    } else {
        constexpr SinCos sc = sincos_2pi(k, N);
        constexpr float c = (float)sc.c;
        constexpr float s = DIR > 0 ? (float)(-sc.s) : (float)sc.s;
        return cf{a.x * c - a.y * s, a.x * s + a.y * c};
    }
}

// ---- small DFTs, in place, natural order in and out: v[k] <- sum_j v[j] exp(-DIR 2 pi i j k / R)
template <int R, int DIR>
struct Dft;

template <int DIR>
struct Dft<2, DIR> {
    static TPIV_HD void run(cf (&v)[2]) {
        const cf a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};
template <int DIR>
struct Dft<4, DIR> {
    static TPIV_HD void run(cf (&v)[4]) {
        const cf s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]), s13 = cadd(v[1], v[3]), d13 = csub(v[1], v[3]);
        const cf r = DIR > 0 ? cf{d13.y, -d13.x} : cf{-d13.y, d13.x};      // -i d13 (forward) / +i d13
        v[0] = cadd(s02, s13);
        v[2] = csub(s02, s13);
        v[1] = cadd(d02, r);
        v[3] = csub(d02, r);
    }
};
// odd radix by the symmetry of the roots: s_j = v_j + v_{R-j}, d_j = v_j - v_{R-j};  X_k = A_k -+ i B_k, X_{R-k} = A_k +- i B_k
// with A_k = v_0 + sum_j cos(2 pi j k / R) s_j, B_k = sum_j sin(2 pi j k / R) d_j
template <int R, int DIR>
struct DftOdd {
    static constexpr int H = (R - 1) / 2;
    static TPIV_HD void run(cf (&v)[R]) {
        cf s[H], d[H];
        static_for<0, H>([&](auto jc) TPIV_LAMBDA_INLINE {
            constexpr int j = decltype(jc)::value;
            s[j] = cadd(v[j + 1], v[R - 1 - j]);
            d[j] = csub(v[j + 1], v[R - 1 - j]);
        });
        const cf v0 = v[0];
        cf sum = v0;
        static_for<0, H>([&](auto jc) TPIV_LAMBDA_INLINE { sum = cadd(sum, s[decltype(jc)::value]); });
        v[0] = sum;
        static_for<1, H + 1>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            cf A = v0, B = cf{0.f, 0.f};
            static_for<1, H + 1>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                constexpr SinCos sc = sincos_2pi(j * k, R);
                constexpr float c = (float)sc.c, sn = (float)sc.s;
                A.x += c * s[j - 1].x;
                A.y += c * s[j - 1].y;
                if constexpr (j == 1) {
                    B.x = sn * d[0].x;
                    B.y = sn * d[0].y;
                } else {
                    B.x += sn * d[j - 1].x;
                    B.y += sn * d[j - 1].y;
                }
            });
            // forward: X_k = A - i B = (A.x + B.y, A.y - B.x); X_{R-k} = A + i B.  Inverse: the two swapped.
            const cf lo = cf{A.x + B.y, A.y - B.x}, hi = cf{A.x - B.y, A.y + B.x};
            v[k] = DIR > 0 ? lo : hi;
            v[R - k] = DIR > 0 ? hi : lo;
        });
    }
};
template <int DIR>
struct Dft<3, DIR> : DftOdd<3, DIR> {};
template <int DIR>
struct Dft<5, DIR> : DftOdd<5, DIR> {};
template <int DIR>
struct Dft<7, DIR> : DftOdd<7, DIR> {};
// even radix 2 h: DFTs of the even and of the odd samples, X_k = E_k + w^k O_k, X_{k+h} = E_k - w^k O_k
template <int R, int DIR>
struct DftEven {
    static constexpr int H = R / 2;
    static TPIV_HD void run(cf (&v)[R]) {
        cf e[H], o[H];
        static_for<0, H>([&](auto jc) TPIV_LAMBDA_INLINE {
            constexpr int j = decltype(jc)::value;
            e[j] = v[2 * j];
            o[j] = v[2 * j + 1];
        });
        Dft<H, DIR>::run(e);
        Dft<H, DIR>::run(o);
        static_for<0, H>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            const cf t = twm<k, R, DIR>(o[k]);
            v[k] = cadd(e[k], t);
            v[k + H] = csub(e[k], t);
        });
    }
};
template <int DIR>
struct Dft<6, DIR> : DftEven<6, DIR> {};
template <int DIR>
struct Dft<8, DIR> : DftEven<8, DIR> {};

// ---- N = n1 n2 with 2 <= n1 <= n2 <= 8, n1 as large as possible (the rule of ct_factors in xcorr_generic.hip)
constexpr int factor1(int n) {
    int n1 = 0;
    for (int a = 2; a * a <= n; ++a)
        if (n % a == 0 && n / a <= 8) n1 = a;
    return n1;
}
constexpr bool mixed_usable(int n) { return factor1(n) >= 2 && n / factor1(n) <= 8; }
// register slot of output bin k
constexpr int mixed_pos(int k, int n) { return (n / factor1(n)) * (k % factor1(n)) + k / factor1(n); }
template <int K, int N>
inline constexpr int MIXED_POS = mixed_pos(K, N);

template <int N, int DIR>
TPIV_HD void fft_mixed(cf (&x)[N]) {
    static_assert(mixed_usable(N), "N = n1 n2 with 2 <= n1 <= n2 <= 8");
    constexpr int N1 = factor1(N), N2 = N / N1;
    // step 1: over a (x[N2 a + b]), then the twiddle w_N^(b k1)
    static_for<0, N2>([&](auto bc) TPIV_LAMBDA_INLINE {
        constexpr int b = decltype(bc)::value;
        cf v[N1];
        static_for<0, N1>([&](auto ac) TPIV_LAMBDA_INLINE { v[decltype(ac)::value] = x[N2 * decltype(ac)::value + b]; });
        Dft<N1, DIR>::run(v);
        static_for<0, N1>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k1 = decltype(kc)::value;
            x[N2 * k1 + b] = twm<b * k1, N, DIR>(v[k1]);
        });
    });
    // step 2: over b (x[N2 k1 + b]): X[k1 + N1 k2] lands at slot N2 k1 + k2
    static_for<0, N1>([&](auto kc) TPIV_LAMBDA_INLINE {
        constexpr int k1 = decltype(kc)::value;
        cf v[N2];
        static_for<0, N2>([&](auto bc) TPIV_LAMBDA_INLINE { v[decltype(bc)::value] = x[N2 * k1 + decltype(bc)::value]; });
        Dft<N2, DIR>::run(v);
        static_for<0, N2>([&](auto qc) TPIV_LAMBDA_INLINE { x[N2 * k1 + decltype(qc)::value] = v[decltype(qc)::value]; });
    });
}

}  // namespace fmx
}  // namespace tpiv
