// In-register (per-lane) complex FFT codelets for the PIV tile kernels.
//
// One lane holds a whole line of N complex samples in registers (N = 8..128)
// and transforms it with compile-time-unrolled radix-4 (first stage radix-2
// when log2 N is odd) decimation-in-frequency butterflies.  All array indices
// and twiddles are compile-time constants, so the array lives in VGPRs and the
// trivial twiddles (1, -i, -1, i) cost nothing.  The output is left in
// digit-reversed order: output bin k sits at fft_pos(k, N).
//
// The header also compiles as plain host C++ (tests/test_fft_codelets.py builds
// it with g++ and checks it against numpy.fft).
#pragma once
#include <type_traits>

#if defined(__HIPCC__)
#define TPIV_HD __host__ __device__ __forceinline__
#else
#define TPIV_HD inline
#endif
// the bodies handed to static_for must be inlined, or the register arrays they touch get
// their address taken and end up in scratch memory
#define TPIV_LAMBDA_INLINE __attribute__((always_inline))

namespace tpiv {

#include "twiddles.inc"

struct cf {
    float x, y;
};

TPIV_HD cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
TPIV_HD cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }

constexpr bool radix2_first(int n) { return n == 2 || n == 8 || n == 32 || n == 128; }

// register position of output bin k after the size-n DIF transform
constexpr int fft_pos(int k, int n) {
    if (n <= 1) return 0;
    if (radix2_first(n)) return (k % 2) * (n / 2) + fft_pos(k / 2, n / 2);
    return (k % 4) * (n / 4) + fft_pos(k / 4, n / 4);
}

// compile-time constant form: indexing a register array with fft_pos(k, n) directly would
// leave a run-time call to the (recursive) function and push the array into scratch
template <int K, int N>
inline constexpr int FFT_POS = fft_pos(K, N);

template <int J, int END, typename F>
TPIV_HD void static_for(F&& f) {
    if constexpr (J < END) {
        f(std::integral_constant<int, J>{});
        static_for<J + 1, END>(f);
    }
}

// a * exp(-DIR * 2*pi*i * K / N)   (DIR = +1 forward, -1 inverse)
template <int K, int N, int DIR>
TPIV_HD cf twmul(cf a) {
    constexpr int idx = ((K % N) * (128 / N)) % 128;
    if constexpr (idx == 0) {
        return a;
    } else if constexpr (idx == 32) {
        return DIR > 0 ? cf{a.y, -a.x} : cf{-a.y, a.x};
    } else if constexpr (idx == 64) {
        return cf{-a.x, -a.y};
    } else if constexpr (idx == 96) {
        return DIR > 0 ? cf{-a.y, a.x} : cf{a.y, -a.x};
    } else {
        constexpr float c = TW_COS[idx];
        constexpr float s = DIR > 0 ? -TW_SIN[idx] : TW_SIN[idx];
        return cf{a.x * c - a.y * s, a.x * s + a.y * c};
    }
}

template <int N, int OFF, int DIR, int TOTAL>
struct FFTStage {
    static TPIV_HD void run(cf (&x)[TOTAL]) {
        if constexpr (N == 2) {
            cf a = x[OFF], b = x[OFF + 1];
            x[OFF] = cadd(a, b);
            x[OFF + 1] = csub(a, b);
        } else if constexpr (radix2_first(N)) {
            static_for<0, N / 2>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                cf a = x[OFF + j], b = x[OFF + j + N / 2];
                x[OFF + j] = cadd(a, b);
                x[OFF + j + N / 2] = twmul<j, N, DIR>(csub(a, b));
            });
            FFTStage<N / 2, OFF, DIR, TOTAL>::run(x);
            FFTStage<N / 2, OFF + N / 2, DIR, TOTAL>::run(x);
        } else if constexpr (N >= 4) {
            static_for<0, N / 4>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                cf a = x[OFF + j], b = x[OFF + j + N / 4], c = x[OFF + j + N / 2],
                   d = x[OFF + j + 3 * N / 4];
                cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), bd = csub(b, d);
                // forward: t3 = -i (b - d); inverse: t3 = +i (b - d)
                cf t3 = DIR > 0 ? cf{bd.y, -bd.x} : cf{-bd.y, bd.x};
                x[OFF + j] = cadd(t0, t2);
                x[OFF + j + N / 4] = twmul<j, N, DIR>(cadd(t1, t3));
                x[OFF + j + N / 2] = twmul<2 * j, N, DIR>(csub(t0, t2));
                x[OFF + j + 3 * N / 4] = twmul<3 * j, N, DIR>(csub(t1, t3));
            });
            if constexpr (N > 4) {
                FFTStage<N / 4, OFF, DIR, TOTAL>::run(x);
                FFTStage<N / 4, OFF + N / 4, DIR, TOTAL>::run(x);
                FFTStage<N / 4, OFF + N / 2, DIR, TOTAL>::run(x);
                FFTStage<N / 4, OFF + 3 * N / 4, DIR, TOTAL>::run(x);
            }
        }
    }
};

// Unnormalised N-point transform of x[0..N) in place; bin k ends at x[fft_pos(k, N)].
template <int N, int DIR>
TPIV_HD void fft_inreg(cf (&x)[N]) {
    FFTStage<N, 0, DIR, N>::run(x);
}

// (Not used by the tile kernels at present: measured there it trades ~3 % fewer VALU instructions for
//  more live registers and came out even; kept, with its host test, for a later round.)
// Real N-point inverse transform of a Hermitian spectrum through ONE N/2-point complex transform
// (the classic even/odd packing): in  Y[k], k = 0..N/2 (natural order; the imaginary parts of the
// DC and Nyquist bins are ignored, as in any c2r transform); out  r[2m] = h[fft_pos(m, N/2)].x,
// r[2m+1] = h[fft_pos(m, N/2)].y  with  r[n] = sum_k Y[k] exp(+2 pi i k n / N)  (unnormalised).
// With z[m] = r[2m] + i r[2m+1]:  z = IDFT_{N/2}(G),  G[k] = (Y[k] + conj Y[M-k]) + i w^k (Y[k] - conj Y[M-k]),
// w = exp(2 pi i / N), M = N/2; the pair (k, M-k) shares S = Y[k] + conj Y[M-k], T = w^k (Y[k] - conj Y[M-k]):
// G[k] = S + i T,  G[M-k] = conj(S) + i conj(T).
template <int N>
TPIV_HD void c2r_inreg(const cf (&Y)[N / 2 + 1], cf (&h)[N / 2]) {
    constexpr int M = N / 2;
    static_assert(M >= 2, "N >= 4");
    h[0] = cf{Y[0].x + Y[M].x, Y[0].x - Y[M].x};
    static_for<1, M / 2>([&](auto kc) TPIV_LAMBDA_INLINE {
        constexpr int k = decltype(kc)::value;
        const cf A = Y[k], B = Y[M - k];
        const cf S{A.x + B.x, A.y - B.y};
        const cf T = twmul<k, N, -1>(cf{A.x - B.x, A.y + B.y});
        h[k] = cf{S.x - T.y, S.y + T.x};
        h[M - k] = cf{S.x + T.y, T.x - S.y};
    });
    h[M / 2] = cf{2.0f * Y[M / 2].x, -2.0f * Y[M / 2].y};
    fft_inreg<M, -1>(h);
}

}  // namespace tpiv
