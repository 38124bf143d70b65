// In-register (per-lane) complex FFT codelets for the PIV tile kernels.
//
// One lane holds a whole line of N complex samples in registers (N = 8..128)
// and transforms it with compile-time-unrolled radix-4 (first stage radix-2
// when log2 N is odd) decimation-in-frequency butterflies.  All array indices
// and twiddles are compile-time constants, so the array lives in VGPRs and the
// trivial twiddles (1, -i, -1, i) cost nothing.  The output is left in
// digit-reversed order: output bin k sits at fft_pos(k, N).
//
// The header also compiles as plain host C++ (tests/test_host_logic.py::test_fft_codelets_on_host builds
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

// ---- where the twiddle constants live ------------------------------------------------------------
// TwLiteral: as 32-bit literals inside the instructions (8-byte encodings; a constant used by several
// instructions is moved to an SGPR by the compiler, and an SGPR operand halves the issue rate on
// MI355X).  TwRegs<NTOP>: the N/4 - 1 non-trivial first-quadrant magnitudes cos(2 pi k / NTOP) and
// their negatives in VGPRs, so that every twiddle product is a plain 4-byte v_mul_f32 / v_fmac_f32.
struct TwLiteral {};

template <int NTOP>
struct TwRegs {
    static constexpr int STEP = 128 / NTOP;             // stride in the 128-entry table
    static constexpr int NV = NTOP / 4 - 1;             // table entries STEP, 2 STEP, ..., 32 - STEP
    float pos[NV > 0 ? NV : 1], neg[NV > 0 ? NV : 1];
    TPIV_HD void init() {
        static_for<0, NV>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int i = decltype(ic)::value;
            pos[i] = TW_COS[(i + 1) * STEP];
            neg[i] = -TW_COS[(i + 1) * STEP];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(pos[i]), "+v"(neg[i]));      // opaque: stays a register operand
#endif
        });
    }
    template <int R, bool NEG>
    TPIV_HD float m() const {                            // (+-) cos(2 pi R / 128), R a multiple of STEP in 1..31
        static_assert(R % STEP == 0 && R > 0 && R < 32, "not a twiddle of this transform");
        return NEG ? neg[R / STEP - 1] : pos[R / STEP - 1];
    }
};

template <int K, int N, int DIR, typename TW>
TPIV_HD cf twmul_t(cf a, const TW& tw) {
    constexpr int idx = ((K % N) * (128 / N)) % 128;
    if constexpr (std::is_same<TW, TwLiteral>::value || idx % 32 == 0) {
        return twmul<K, N, DIR>(a);
    } else {
        // cos = (+-) M[rc], sin = (+-) M[rs] with M[r] = cos(2 pi r / 128) = sin(2 pi (32 - r) / 128)
        constexpr int q = idx / 32, r = idx % 32;
        constexpr int rc = (q % 2 == 0) ? r : 32 - r;
        constexpr int rs = (q % 2 == 0) ? 32 - r : r;
        constexpr bool cneg = (q == 1 || q == 2);
        constexpr bool sneg0 = (q == 2 || q == 3);
        constexpr bool sneg = DIR > 0 ? !sneg0 : sneg0;   // multiply by exp(-DIR * i * theta)
        const float C = tw.template m<rc, cneg>();
        const float S = tw.template m<rs, sneg>();
        const float nS = tw.template m<rs, !sneg>();
#if defined(__HIP_DEVICE_COMPILE__)
        return cf{__builtin_fmaf(a.y, nS, a.x * C), __builtin_fmaf(a.y, C, a.x * S)};
#else
        return cf{a.x * C + a.y * nS, a.x * S + a.y * C};
#endif
    }
}

template <int N, int OFF, int DIR, int TOTAL, typename TW = TwLiteral>
struct FFTStage {
    static TPIV_HD void run(cf (&x)[TOTAL], const TW& tw = TW{}) {
        if constexpr (N == 2) {
            cf a = x[OFF], b = x[OFF + 1];
            x[OFF] = cadd(a, b);
            x[OFF + 1] = csub(a, b);
        } else if constexpr (radix2_first(N)) {
            static_for<0, N / 2>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                cf a = x[OFF + j], b = x[OFF + j + N / 2];
                x[OFF + j] = cadd(a, b);
                x[OFF + j + N / 2] = twmul_t<j, N, DIR>(csub(a, b), tw);
            });
            FFTStage<N / 2, OFF, DIR, TOTAL, TW>::run(x, tw);
            FFTStage<N / 2, OFF + N / 2, DIR, TOTAL, TW>::run(x, tw);
        } else if constexpr (N >= 4) {
            static_for<0, N / 4>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                cf a = x[OFF + j], b = x[OFF + j + N / 4], c = x[OFF + j + N / 2],
                   d = x[OFF + j + 3 * N / 4];
                cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), bd = csub(b, d);
                // forward: t3 = -i (b - d); inverse: t3 = +i (b - d)
                cf t3 = DIR > 0 ? cf{bd.y, -bd.x} : cf{-bd.y, bd.x};
                x[OFF + j] = cadd(t0, t2);
                x[OFF + j + N / 4] = twmul_t<j, N, DIR>(cadd(t1, t3), tw);
                x[OFF + j + N / 2] = twmul_t<2 * j, N, DIR>(csub(t0, t2), tw);
                x[OFF + j + 3 * N / 4] = twmul_t<3 * j, N, DIR>(csub(t1, t3), tw);
            });
            if constexpr (N > 4) {
                FFTStage<N / 4, OFF, DIR, TOTAL, TW>::run(x, tw);
                FFTStage<N / 4, OFF + N / 4, DIR, TOTAL, TW>::run(x, tw);
                FFTStage<N / 4, OFF + N / 2, DIR, TOTAL, TW>::run(x, tw);
                FFTStage<N / 4, OFF + 3 * N / 4, DIR, TOTAL, TW>::run(x, tw);
            }
        }
    }
};

// Unnormalised N-point transform of x[0..N) in place; bin k ends at x[fft_pos(k, N)].
template <int N, int DIR>
TPIV_HD void fft_inreg(cf (&x)[N]) {
    FFTStage<N, 0, DIR, N>::run(x);
}
// the same with the twiddle constants taken from `tw` (TwRegs<N>, or TwLiteral{} for the form above)
template <int N, int DIR, typename TW>
TPIV_HD void fft_inreg(cf (&x)[N], const TW& tw) {
    FFTStage<N, 0, DIR, N, TW>::run(x, tw);
}

// (Used by the tile kernels for WS <= 32: same-box A/B 2.4 % (32x32 CWS pass) to 5.8 % (32x32 pass 1)
//  faster than the full complex transform; 64x64 is at its register limit and keeps the complex form.)
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
