// Per-thread arithmetic of the float64 first pass for 64x64 and 128x128 windows (xcorr_f64.hip), written as host/device
// functions so that the CPU suite can run the WHOLE scheme -- index maps, splits, partner bins, combines --
// thread by thread against numpy (tests/host/f64_split_harness.cpp, tests/test_host_logic.py).
//
// A 64-point complex float64 line does not fit one lane (256 VGPRs), so every line is split over TWO threads
// and every 64-point transform is a 32-point in-register codelet (fft_inreg_f64.hpp) plus one radix-2 step
// that is folded into the LDS transposition in front of (decimation in frequency) or behind (decimation in
// time) the codelet -- adds and subtractions of two plane cells while reading, plane by plane:
//
//   workgroup = 128 threads = one window; thread t: lane = t & 63, wave wv = t >> 6
//   R   rows forward     thread (y = lane, h = wv) loads the whole image row y of both frames, forms the DIF halves
//                        u[j] = x[j] + x[j+32]  (h = 0)   /   (x[j] - x[j+32]) w64^j  (h = 1),  j = 0..31
//                        (integer sums / differences of the bytes, exact; a / mean, b / mean of B:513-514 are applied
//                        later as ONE factor on the map, see rows_forward) and transforms: X[y][2m + h] at
//                        x[fft_pos(m, 32)]
//   T1  transposition    plane[y][kx] <- X, one float64 component at a time (64 x 65 doubles = 33 KB: four
//                        workgroups per CU); thread (k = lane, g = 1 - wv) reads column k as the DIF halves over
//                        the rows: u[i] = X[i][k] +- X[i+32][k], the odd half times w64^i
//   C   columns forward  32-point codelet: Z[2m + g][k] at u[fft_pos(m, 32)]
//   X   cross-spectrum   P = conj(A) B of the packed transform Z = FFT2(a + i b); the mirrored bin Z(-ky, -k) has the
//                        parity of ky, i.e. it lives in the same wavefront, lane (64 - k) % 64: ds_bpermute, no LDS memory
//   Ci  columns inverse  the thread's own-parity bins through a 32-point inverse codelet: G_g[y1]; decimation in time:
//                        Y[y1 + 32 c][k] = G_0[y1] +- w64^-y1 G_1[y1]; the g = 1 threads multiply by w64^-y1
//   T2  transposition    plane[32 g + y1][k] <- G; thread (y = lane, q = wv) reads spectrum columns 0..32 of its row
//                        (the row is real: Hermitian spectrum) as G_0 +- G_1
//   Ri  rows inverse     the real 64-point row split over the thread pair by output parity: the even / odd samples are
//                        real 32-point rows themselves (c2r through a 16-point codelet): thread q holds corr[y][2n + q]
//   P   peak analysis    on the float64 map (B:346-358, B:381-392, B:518), 8-double record for finalize_kernel<true>
//
// The odd halves carry 30 twiddle products per stage; the parities are assigned so that wave 0 takes them in the
// column stages and wave 1 in the row stages.
//
// 128x128 windows run the same scheme with 64-point codelets (Split<128>: 256 threads, line = t % 128, half = t / 128;
// one 132 KB plane, i.e. one workgroup per CU and up to 512 registers per thread).  There the two halves of a line and
// the mirrored column -k live in different wavefronts, so the cross-spectrum fetches Z(-ky, -k) through the plane
// (cross_write / cross_read) instead of ds_bpermute.  The text above gives the numbers of the 64x64 case.
#pragma once
#include <math.h>
#include <stdint.h>

#include "fft_inreg_f64.hpp"

namespace tpiv {
namespace f64s {

#if defined(__HIP_DEVICE_COMPILE__)
// byte SEL of a (+ / -) byte SEL of b as a 32-bit integer, one instruction (sub-dword addressing of both sources)
template <int SEL>
__device__ __forceinline__ int sdwa_add(uint32_t a, uint32_t b) {
    int r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_%3 src1_sel:BYTE_%3"
        : "=v"(r) : "v"(a), "v"(b), "n"(SEL));
    return r;
}
template <int SEL>
__device__ __forceinline__ int sdwa_sub(uint32_t a, uint32_t b) {
    int r;
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_%3 src1_sel:BYTE_%3"
        : "=v"(r) : "v"(a), "v"(b), "n"(SEL));
    return r;
}
// all four byte lanes of a dword pair of frame a and of frame b in ONE asm block (the backend pads every asm block with
// a wait state of its own: eight single-instruction blocks cost eight s_nop)
#define TPIV_SDWA4(OP)                                                                                                  \
    asm(OP " %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"                       \
        OP " %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"                       \
        OP " %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n\t"                       \
        OP " %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"                       \
        OP " %4, %10, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"                     \
        OP " %5, %10, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n\t"                     \
        OP " %6, %10, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n\t"                     \
        OP " %7, %10, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3"                          \
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])        \
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1))
template <bool SUB>
__device__ __forceinline__ void sdwa_quad(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, int (&r)[8]) {
    if constexpr (SUB) TPIV_SDWA4("v_sub_u32_sdwa");
    else TPIV_SDWA4("v_add_u32_sdwa");
}
#undef TPIV_SDWA4
#endif

template <int K, int NDW>
TPIV_HD float byte_of(const uint32_t (&d)[NDW]) {
    return (float)((d[K >> 2] >> (8 * (K & 3))) & 0xffu);
}

// ---- LDS reads of the transposes.  Device: explicit ds_read_b64 (the backend would merge neighbouring reads into
// ds_read2_b64 pairs, which run at HALF the LDS rate: 8 cycles for 2 x 8 bytes per lane against 2 + 2), issued in
// batches of eight with the next batch in flight while the current one is consumed (lgkmcnt counts at most 15).
// Host: plain loads.
#if defined(__HIP_DEVICE_COMPILE__)
template <int OFF>
__device__ __forceinline__ double lds_rd(unsigned addr) {
    double v;
    if constexpr (OFF < 65536) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    } else {                  // the offset field has 16 bits (the 128x128 plane is 132 KB): second base, shared by the reads
        const unsigned hi = addr + (unsigned)(OFF & ~0xffff);
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(hi), "n"(OFF & 0xffff));
    }
    return v;
}
// explicit ds_write_b64 (the backend merges neighbouring writes into ds_write2_b64 pairs)
template <int OFF>
__device__ __forceinline__ void lds_wr(unsigned addr, double v) {
    if constexpr (OFF < 65536) {
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
    } else {
        const unsigned hi = addr + (unsigned)(OFF & ~0xffff);
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(hi), "v"(v), "n"(OFF & 0xffff) : "memory");
    }
}
// wait until at most LEFT LDS operations are outstanding; the eight values become usable only behind it
template <int LEFT>
__device__ __forceinline__ void lds_wait(double (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(LEFT));
}
__device__ __forceinline__ unsigned lds_addr(const double* p) { return (unsigned)(uintptr_t)p; }     // low dword of a flat LDS address
#elif defined(__HIPCC__)
// host pass of a HIP translation unit: the kernels' bodies are parsed, never run
template <int OFF>
inline double lds_rd(unsigned) { return 0.0; }
template <int LEFT>
inline void lds_wait(double (&)[8]) {}
template <int OFF>
inline void lds_wr(unsigned, double) {}
inline unsigned lds_addr(const double*) { return 0u; }
#endif

// (device: an empty asm on a value keeps computations that depend on it where they are written -- left alone, the
//  compiler hoists the squares a^2, b^2 of ALL bins out of both parity branches of the cross-spectrum and spills 48 of them)
TPIV_HD void pin(cd& z) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(z.x), "+v"(z.y));
#else
    (void)z;
#endif
}
// cross-spectrum of one bin.  With Z(k) = a + ib, Z(-k) = c + id:  4 P(k) = 2 (a d + b c) + i ((c^2 - a^2) + (d^2 - b^2))
// (P = conj(A) B of the packed transform; the factor 0.25 / n^2 rides in the map scale, peak_shifted)
TPIV_HD cd cross_bin(cd zk, cd zm) {
    const double a_ = zk.x, b_ = zk.y, c_ = zm.x, d_ = zm.y;
    // imaginary part as |Z(-k)|^2 - |Z(k)|^2: one multiply and three fused multiply-adds
#if defined(__HIP_DEVICE_COMPILE__)
    const double n1 = __fma_rn(a_, a_, b_ * b_);
    const double im = __fma_rn(c_, c_, __fma_rn(d_, d_, -n1));
#else
    const double im = (c_ * c_ + d_ * d_) - (a_ * a_ + b_ * b_);
#endif
    const double re = a_ * d_ + b_ * c_;
    return cd{re + re, im};
}
// min / max of two finite doubles as ONE instruction (fmin / fmax are compiled to v_min_f64 / v_max_f64 plus a
// canonicalising v_max_f64 x, x per operand -- sNaN quieting the map values never need: 100 float64 instructions per window)
TPIV_HD double dmin2(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a < b ? a : b;
#endif
}
TPIV_HD double dmax2(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a > b ? a : b;
#endif
}
// (corr - min) + 1e-7 of B:518 / B:381 on the normalised map: `scale` = 1 / (mean(a) mean(b)) times the constant factors of
// the transforms (rows_forward)
TPIV_HD double peak_shifted(double c, double cmin, double scale) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fma_rn(__dsub_rn(c, cmin), scale, 1e-7);
#else
    return (c - cmin) * scale + 1e-7;
#endif
}

// W = window edge (64 or 128); every function is the work of ONE thread
template <int W>
struct Split {
    static constexpr int WS = W;
    static constexpr int M = W / 2;          // codelet length
    static constexpr int PL = W + 1;         // plane pitch in doubles: conflict-free ds_read_b64 / ds_write_b64 along rows and columns
    static constexpr int NDW = W / 4;        // dwords per window row
    static constexpr int NT = 2 * W;         // threads per window: line = t % W, half = t / W

    // ---- R: thread (y, h).  da / db: row y of frame a / b.  The samples go in as they are: the normalisation a / mean(a),
    // b / mean(b) of B:513-514 is linear, so it is ONE factor 1 / (mean(a) mean(b)) on the whole correlation map, applied
    // where the map is shifted by its minimum (peak_shifted) -- W multiplies per thread and one of two divisions per window
    // less, at rounding-level differences (1e-16 relative) from scaling every sample first.
    static TPIV_HD void rows_forward(const uint32_t (&da)[NDW], const uint32_t (&db)[NDW], int h, cd (&x)[M]) {
#if defined(__HIP_DEVICE_COMPILE__)
        // sums / differences of two bytes that sit in the same byte lane of two dwords (M is a multiple of 4): ONE
        // sub-dword-addressed integer add / subtract each (SDWA byte selects), then one exact conversion -- two
        // instructions per real sample instead of four (byte -> float, byte -> float, fma, float -> double)
        if (h) {
            static_for<0, M / 4>([&](auto qc) TPIV_LAMBDA_INLINE {
                constexpr int q = decltype(qc)::value;
                int r[8];
                sdwa_quad<true>(da[q], da[q + M / 4], db[q], db[q + M / 4], r);
                static_for<0, 4>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    x[4 * q + k] = cd{(double)r[k], (double)r[4 + k]};
                });
            });
            static_for<1, M>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                x[j] = twmul_d<j, W, 1>(x[j]);
            });
        } else {
            static_for<0, M / 4>([&](auto qc) TPIV_LAMBDA_INLINE {
                constexpr int q = decltype(qc)::value;
                int r[8];
                sdwa_quad<false>(da[q], da[q + M / 4], db[q], db[q + M / 4], r);
                static_for<0, 4>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    x[4 * q + k] = cd{(double)r[k], (double)r[4 + k]};
                });
            });
        }
#else
        const float sg = h ? -1.0f : 1.0f;
        static_for<0, M>([&](auto jc) TPIV_LAMBDA_INLINE {
            constexpr int j = decltype(jc)::value;
            // sums / differences of two bytes: exact in float32, converted once
            const float sa = byte_of<j, NDW>(da) + sg * byte_of<j + M, NDW>(da);
            const float sb = byte_of<j, NDW>(db) + sg * byte_of<j + M, NDW>(db);
            x[j] = cd{(double)sa, (double)sb};
        });
        if (h) {
            static_for<1, M>([&](auto jc) TPIV_LAMBDA_INLINE {
                constexpr int j = decltype(jc)::value;
                x[j] = twmul_d<j, W, 1>(x[j]);
            });
        }
#endif
        fft_inreg_d<M, 1>(x);          // X[y][2m + h] at x[FFT_POS<m, M>]
    }

    // ---- T1 write: component COMP (0 = real, 1 = imaginary) of X[y][2m + h]
    template <int COMP>
    static TPIV_HD void t1_write(const cd (&x)[M], int y, int h, double* plane) {
        double* row = plane + y * PL + h;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(row);
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            lds_wr<16 * m>(base, COMP ? x[FFT_POS<m, M>].y : x[FFT_POS<m, M>].x);
        });
#else
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            row[2 * m] = COMP ? x[FFT_POS<m, M>].y : x[FFT_POS<m, M>].x;
        });
#endif
    }
    // ---- T1 read: thread (k, g): u[i] = X[i][k] +- X[i + M][k]
    template <int COMP>
    static TPIV_HD void t1_read(cd (&u)[M], int k, int g, const double* plane) {
        const double sg = g ? -1.0 : 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(plane + k);
        constexpr int NB = M / 4;                                   // batches of 4 rows i = 8 reads
        double v[2][8];
        auto issue = [&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            static_for<0, 4>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int i = 4 * b_ + decltype(ic)::value;
                v[b_ & 1][2 * decltype(ic)::value] = lds_rd<i * PL * 8>(base);
                v[b_ & 1][2 * decltype(ic)::value + 1] = lds_rd<(i + M) * PL * 8>(base);
            });
        };
        issue(std::integral_constant<int, 0>{});
        static_for<0, NB>([&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            if constexpr (b_ + 1 < NB) issue(std::integral_constant<int, b_ + 1>{});
            lds_wait<(b_ + 1 < NB) ? 8 : 0>(v[b_ & 1]);
            static_for<0, 4>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int q = decltype(ic)::value, i = 4 * b_ + q;
                const double r = v[b_ & 1][2 * q] + sg * v[b_ & 1][2 * q + 1];
                if constexpr (COMP) u[i].y = r;
                else u[i].x = r;
            });
        });
#else
        const double* col = plane + k;
        for (int i = 0; i < M; ++i) {
            const double r = col[i * PL] + sg * col[(i + M) * PL];
            if (COMP) u[i].y = r;
            else u[i].x = r;
        }
#endif
    }
    // ---- C: the odd half's twiddles, then the codelet: Z[2m + g][k] at u[FFT_POS<m, M>]
    static TPIV_HD void cols_forward(cd (&u)[M], int g) {
        if (g) {
            static_for<1, M>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int i = decltype(ic)::value;
                u[i] = twmul_d<i, W, 1>(u[i]);
            });
        }
        fft_inreg_d<M, 1>(u);
    }

    // ---- X (64x64): cross-spectrum in place with the mirrored bins fetched by ds_bpermute.  sh(value, reg, comp, partner)
    // returns the PARTNER thread's value of register `reg` (device: ds_bpermute of `value`; the host harness looks it up).
    // bin ky = 2m + G sits at z[FFT_POS<m>]; its mirror -ky = 2 m' + G with m' = (M - m) % M (G = 0), M - 1 - m (G = 1)
    template <int G, typename SH>
    static TPIV_HD void cross_spectrum_g(cd (&z)[M], int partner, SH&& sh) {
        constexpr int NPAIR = G ? M / 2 : M / 2 + 1;
        static_for<0, NPAIR>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            constexpr int m2 = G ? M - 1 - m : (M - m) % M;
            constexpr int p1 = FFT_POS<m, M>, p2 = FFT_POS<m2, M>;
            cd z1 = z[p1];
            pin(z1);
            if constexpr (m == m2) {
                const cd m1{sh(z1.x, p1, 0, partner), sh(z1.y, p1, 1, partner)};
                z[p1] = cross_bin(z1, m1);
            } else {
                cd z2 = z[p2];
                pin(z2);
                const cd m1{sh(z2.x, p2, 0, partner), sh(z2.y, p2, 1, partner)};      // Z(-ky, -k)
                const cd m2v{sh(z1.x, p1, 0, partner), sh(z1.y, p1, 1, partner)};     // Z(+ky, -k): the mirror of bin -ky
                z[p1] = cross_bin(z1, m1);
                z[p2] = cross_bin(z2, m2v);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            // keep the exchange of a bin pair together: hoisted ahead, the 8 permutes of every pair hold their results
            // in registers next to the 128 of the spectrum
            if constexpr (m % 2 == 1) __builtin_amdgcn_sched_barrier(0);
#endif
        });
    }
    // ---- X (128x128): the mirrored bins through the plane.  cross_write<COMP>: plane[ky][k] <- Z(ky, k); cross_read<COMP>:
    // mir[m] = component of Z(-ky, -k) for the thread's bins ky = 2m + g (row (W - ky) % W, column (W - k) % W).
    template <int COMP>
    static TPIV_HD void cross_write(const cd (&z)[M], int k, int g, double* plane) {
        double* col = plane + g * PL + k;
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            col[2 * m * PL] = COMP ? z[FFT_POS<m, M>].y : z[FFT_POS<m, M>].x;
        });
    }
    static TPIV_HD void cross_read(double (&mir)[M], int k, int g, const double* plane) {
        const int mk = (W - k) & (W - 1);
        // row of bin m: (W - 2m - g) % W = W - 2m - g for m >= 1; m = 0: (W - g) % W
        const double* c0 = plane + mk - g * PL;                  // + (W - 2m) * PL for m >= 1
        mir[0] = plane[((W - g) & (W - 1)) * PL + mk];
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(c0);
        constexpr int NB = (M - 1 + 7) / 8;
        double v[2][8];
        auto issue = [&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            static_for<0, 8>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int m = 1 + 8 * b_ + decltype(ic)::value;
                if constexpr (m < M) v[b_ & 1][decltype(ic)::value] = lds_rd<(W - 2 * m) * PL * 8>(base);
                else v[b_ & 1][decltype(ic)::value] = 0.0;
            });
        };
        issue(std::integral_constant<int, 0>{});
        static_for<0, NB>([&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            if constexpr (b_ + 1 < NB) issue(std::integral_constant<int, b_ + 1>{});
            constexpr int left = M - 1 - 8 * (b_ + 1);               // reads issued behind this batch
            lds_wait<(b_ + 1 < NB) ? (left < 8 ? left : 8) : 0>(v[b_ & 1]);
            static_for<0, 8>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int m = 1 + 8 * b_ + decltype(ic)::value;
                if constexpr (m < M) mir[m] = v[b_ & 1][decltype(ic)::value];
            });
        });
#else
        for (int m = 1; m < M; ++m) mir[m] = c0[(W - 2 * m) * PL];
#endif
    }
    static TPIV_HD void cross_finish(cd (&z)[M], const double (&mre)[M], const double (&mim)[M]) {
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            z[FFT_POS<m, M>] = cross_bin(z[FFT_POS<m, M>], cd{mre[m], mim[m]});
        });
    }

    // ---- Ci: natural-order rename, inverse codelet, decimation-in-time twiddle of the odd half: G_g[y1] at t[FFT_POS<y1, M>]
    static TPIV_HD void cols_inverse(const cd (&z)[M], int g, cd (&t)[M]) {
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            t[m] = z[FFT_POS<m, M>];
        });
        fft_inreg_d<M, -1>(t);
        if (g) {
            static_for<1, M>([&](auto yc) TPIV_LAMBDA_INLINE {
                constexpr int y1 = decltype(yc)::value;
                t[FFT_POS<y1, M>] = twmul_d<y1, W, -1>(t[FFT_POS<y1, M>]);
            });
        }
    }
    // ---- T2 write: thread (k, g): plane[M g + y1][k] <- G_g[y1][k]
    template <int COMP>
    static TPIV_HD void t2_write(const cd (&t)[M], int k, int g, double* plane) {
        double* col = plane + (M * g) * PL + k;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(col);
        static_for<0, M>([&](auto yc) TPIV_LAMBDA_INLINE {
            constexpr int y1 = decltype(yc)::value;
            lds_wr<y1 * PL * 8>(base, COMP ? t[FFT_POS<y1, M>].y : t[FFT_POS<y1, M>].x);
        });
#else
        static_for<0, M>([&](auto yc) TPIV_LAMBDA_INLINE {
            constexpr int y1 = decltype(yc)::value;
            col[y1 * PL] = COMP ? t[FFT_POS<y1, M>].y : t[FFT_POS<y1, M>].x;
        });
#endif
    }
    // ---- T2 read: thread (y, q): Y[kx] = G_0[y % M][kx] +- G_1[y % M][kx], kx = 0..M
    template <int COMP>
    static TPIV_HD void t2_read(cd (&Y)[M + 1], int y, const double* plane) {
        const double sg = y < M ? 1.0 : -1.0;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(plane + (y & (M - 1)) * PL);
        constexpr int NB = M / 4 + 1;                               // M / 4 batches of 4 columns + column M
        double v[2][8];
        auto issue = [&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            static_for<0, 4>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int kx = 4 * b_ + decltype(ic)::value;
                if constexpr (kx <= M) {
                    v[b_ & 1][2 * decltype(ic)::value] = lds_rd<kx * 8>(base);
                    v[b_ & 1][2 * decltype(ic)::value + 1] = lds_rd<(M * PL + kx) * 8>(base);
                }
            });
        };
        issue(std::integral_constant<int, 0>{});
        static_for<0, NB>([&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            if constexpr (b_ + 1 < NB) issue(std::integral_constant<int, b_ + 1>{});
            // outstanding behind this batch: the next one (8 reads, the last batch holds 2)
            lds_wait<(b_ + 1 < NB) ? (b_ + 2 == NB ? 2 : 8) : 0>(v[b_ & 1]);
            static_for<0, 4>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int q = decltype(ic)::value, kx = 4 * b_ + q;
                if constexpr (kx <= M) {
                    const double r = v[b_ & 1][2 * q] + sg * v[b_ & 1][2 * q + 1];
                    if constexpr (COMP) Y[kx].y = r;
                    else Y[kx].x = r;
                }
            });
        });
#else
        const double* r0 = plane + (y & (M - 1)) * PL;
        const double* r1 = r0 + M * PL;
        for (int kx = 0; kx <= M; ++kx) {
            const double r = r0[kx] + sg * r1[kx];
            if (COMP) Y[kx].y = r;
            else Y[kx].x = r;
        }
#endif
    }
    // ---- Ri: the real W-point row from its Hermitian half spectrum Y[0..M], split over the thread pair by OUTPUT parity
    // (decimation in time): with w = exp(2 pi i / W),
    //     even samples  r[2n]   = IDFT_M(E)[n],  E[k] = Y[k] + Y[k + M]          = Y[k] + conj(Y[M - k])
    //     odd  samples  r[2n+1] = IDFT_M(O)[n],  O[k] = (Y[k] - Y[k + M]) w^k    = (Y[k] - conj(Y[M - k])) w^k
    // and E, O are Hermitian over length M again, i.e. each half is a real M-point row: c2r through an M/2-point complex
    // codelet from the bins 0 .. M/2 (c2r_pre_d<M>).  Thread q takes the samples of parity q -- 90 float64 instructions
    // per thread fewer than a W-point c2r pre-processing done by both threads followed by a split M-point transform.
    // Out: c[2n + e] = corr[y][4n + 2e + q], n = 0..M/2-1, e = 0, 1 (un-shifted column index).
    static TPIV_HD void rows_inverse(const cd (&Y)[M + 1], int q, double (&c)[M]) {
        constexpr int H = M / 2;
        cd g[H + 1];
        const double sq = q ? -1.0 : 1.0;
        static_for<0, H + 1>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            // Y[k] +- conj(Y[M - k])
            g[k] = cd{Y[k].x + sq * Y[M - k].x, Y[k].y - sq * Y[M - k].y};
        });
        if (q) {
            static_for<1, H + 1>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                g[k] = twmul_d<k, W, -1>(g[k]);
            });
        }
        c2r_pre_d<M>(g);                        // packed values in g[0..H)
        cd e[H];
        static_for<0, H>([&](auto jc) TPIV_LAMBDA_INLINE {
            constexpr int j = decltype(jc)::value;
            e[j] = g[j];
        });
        fft_inreg_d<H, -1>(e);                  // r_q[2m] + i r_q[2m + 1] at e[FFT_POS<m, H>],  r_q[n] = corr[y][2n + q]
        static_for<0, H>([&](auto nc) TPIV_LAMBDA_INLINE {
            constexpr int n = decltype(nc)::value;
            c[2 * n] = e[FFT_POS<n, H>].x;
            c[2 * n + 1] = e[FFT_POS<n, H>].y;
        });
    }

    // ---- P: peak analysis in fftshift coordinates (y' = (y + W/2) % W, x' likewise).  Thread (y, q) holds
    // c[2n + e] = corr[y][x = 4n + 2e + q].
    // one pass over the raw cells gives their minimum AND their maximum: v = (c - min) + 1e-7 is monotonic in c, so the
    // maximum of the shifted cells is the shifted maximum (exactly: the same two roundings)
    static TPIV_HD void peak_local_minmax(const double (&c)[M], double& mn, double& mx) {
        mn = c[0];
        mx = c[0];
#pragma unroll
        for (int i = 1; i < M; ++i) {
            mn = dmin2(mn, c[i]);
            mx = dmax2(mx, c[i]);
        }
    }
    // v = (c - min) + 1e-7, written to the map (plane, shifted coordinates).  (No index is tracked: the arg-max position
    // comes from the map afterwards -- first the smallest row whose maximum is the global one, then the first column of
    // that row -- which keeps the scan free of compare / select chains.)
    static TPIV_HD void peak_shift_and_write(double (&c)[M], double cmin, double scale, int y, int q, double* plane) {
        const int fy = (y + W / 2) & (W - 1);
        double* row = plane + fy * PL + q;
        static_for<0, M>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int i = decltype(ic)::value;
            constexpr int fx0 = (4 * (i >> 1) + 2 * (i & 1) + W / 2) & (W - 1);   // + q: stays inside its pair
            const double v = peak_shifted(c[i], cmin, scale);
            c[i] = v;
#if defined(__HIP_DEVICE_COMPILE__)
            lds_wr<fx0 * 8>(lds_addr(row), v);
#else
            row[fx0] = v;
#endif
        });
    }
    // second peak: the thread's maximum outside the flat-index exclusion zone of m (B:346-358): f in {clamp(m + i + W j),
    // |i|, |j| <= wv}: in row fy the columns mx + i (j = fy - my), mx + i + W (the row wrap, j = fy - my - 1) and
    // mx + i - W (j = fy - my + 1), plus the two clamps.  Negative if every cell of the thread is excluded (an excluded
    // cell takes part with its sign bit set: every map value is >= 1e-7).
    static TPIV_HD double peak_second_local(const double (&c)[M], int y, int q, int m, int wv) {
        constexpr int NW = W / 64;
        const int fy = (y + W / 2) & (W - 1);
        const int my = m / W, mx = m % W;
        const int dj = fy - my;
        unsigned long long ex[NW] = {};                // bit fx set = excluded
        auto span = [&](int lo_, int hi_) TPIV_LAMBDA_INLINE {
            lo_ = lo_ < 0 ? 0 : lo_;
            hi_ = hi_ > W - 1 ? W - 1 : hi_;
            for (int b_ = lo_; b_ <= hi_; ++b_) ex[b_ >> 6] |= 1ull << (b_ & 63);     // at most 2 wv + 1 columns
        };
        if (dj >= -wv && dj <= wv) span(mx - wv, mx + wv);
        if (dj + 1 >= -wv && dj + 1 <= wv) span(mx - wv + W, mx + wv + W);
        if (dj - 1 >= -wv && dj - 1 <= wv) span(mx - wv - W, mx + wv - W);
        if (fy == 0 && (m - wv - wv * W) <= 0) ex[0] |= 1ull;
        if (fy == W - 1 && (m + wv + wv * W) >= W * W - 1) ex[NW - 1] |= 1ull << 63;
        // this thread's columns are fx0 + q: shift the mask down by q, the bit positions become compile-time constants
        unsigned word[2 * NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            unsigned long long w_ = ex[i] >> q;
            if (i + 1 < NW && q) w_ |= ex[i + 1] << 63;
            word[2 * i] = (unsigned)w_;
            word[2 * i + 1] = (unsigned)(w_ >> 32);
        }
        double sv = -1.0;
        static_for<0, M>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int i = decltype(ic)::value;
            constexpr int fx0 = (4 * (i >> 1) + 2 * (i & 1) + W / 2) & (W - 1);
#if defined(__HIP_DEVICE_COMPILE__)
            // sign-extended exclusion bit (0 / -1) -> sign bit of the value: excluded cells lose every comparison
            const int kill = __builtin_amdgcn_sbfe((int)word[fx0 >> 5], fx0 & 31, 1);
            const double v = __hiloint2double(__double2hiint(c[i]) | (kill & (int)0x80000000), __double2loint(c[i]));
#else
            const bool excl = (word[fx0 >> 5] >> (fx0 & 31)) & 1u;
            const double v = excl ? -c[i] : c[i];
#endif
            sv = dmax2(sv, v);
        });
        return sv;
    }
    // record for finalize_kernel<true>: slot 0..7 = c[m], c[left], c[right], c[top], c[bot], c[m2], m, dead
    // (flat-index neighbours and fix-ups of B:385-392; `sv` < 0: nothing left outside the exclusion zone -- the
    // reference's second arg-max then runs over the zeroed map: c[m2] = 0, ratio = +inf)
    static TPIV_HD double peak_record_slot(int slot, int m, double sv, bool dead, const double* plane) {
        const int KD = W * W;
        int left = m + 1, right = m - 1, top = m + W, bot = m - W;
        if (left >= KD - 1) left = m;
        if (right <= 0) right = m;
        if (top >= KD - 1) top = m;
        if (bot <= 0) bot = m;
        int qi = m;
        qi = slot == 1 ? left : qi;
        qi = slot == 2 ? right : qi;
        qi = slot == 3 ? top : qi;
        qi = slot == 4 ? bot : qi;
        double outv = plane[(qi / W) * PL + (qi % W)];
        outv = slot == 5 ? (sv >= 0.0 ? sv : 0.0) : outv;
        outv = slot == 6 ? (double)m : outv;
        outv = slot == 7 ? (dead ? 1.0 : 0.0) : outv;
        return outv;
    }
};

}  // namespace f64s
}  // namespace tpiv
