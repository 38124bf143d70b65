// Per-thread arithmetic of the float64 first pass for 64x64 and 128x128 windows (xcorr_f64.hip), written as host/device
// functions so that the CPU suite can run the WHOLE scheme -- index maps, splits, partner bins, combines, peak logic --
// thread by thread against numpy (tests/host/f64_split_harness.cpp, tests/test_host_logic.py).
//
// A W-point complex float64 line does not fit one lane, so every line is split over TWO threads and every W-point
// transform is a W/2-point in-register codelet (fft_inreg_f64.hpp) plus one radix-2 step that is folded into the LDS
// transposition in front of (decimation in frequency) or behind (decimation in time) the codelet.  Third generation
// (round 4); what changed against the second one is marked (r4):
//
//   workgroup = 2 W threads = one window; thread t: line = t % W, half = t / W  (W = 64: half = wavefront)
//   R   rows forward     thread (y = line, h = half) loads the whole image row y of both frames, forms the DIF halves
//                        u[j] = x[j] + x[j+W/2]  (h = 0)   /   (x[j] - x[j+W/2]) w^j  (h = 1)
//                        (integer sums / differences of the bytes, exact; a / mean, b / mean of B:513-514 are applied
//                        later as ONE factor on the map, see rows_forward) and transforms: X[y][2m + h] at
//                        x[fft_pos(m)]
//   T1  transposition    one float64 component at a time through the plane, which is stored COLUMN-major in the lane
//                        order of the column stages (r4): plane[pos(kx)][y], pitch W + 2.  The writer's lanes are
//                        consecutive rows y (contiguous, conflict-free); the reader (position L = line, g = 1 - half)
//                        gets two consecutive rows of its column with ONE ds_read_b128 (lane stride W + 2 doubles:
//                        conflict-free for the 16-lane groups of a 128-bit read) and forms the DIF halves over the
//                        rows: u[i] = X[i][k] +- X[i+W/2][k], the odd half times w^i
//   C   columns forward  W/2-point codelet: Z[2m + g][k] at u[fft_pos(m)]
//   X   cross-spectrum   P = conj(A) B of the packed transform Z = FFT2(a + i b) needs the mirrored bin Z(-ky, -k).
//                        (r4) The columns sit in the lanes in the order 0, W/2, 1, W-1, 2, W-2, ...: column k and its
//                        mirror W - k are NEIGHBOURING lanes, so the mirrored bin comes through a quad-permute DPP move
//                        (full-rate VALU, no LDS, no wait; it was 128 ds_bpermute per thread for W = 64 and two more
//                        passes through the plane with four barriers for W = 128).  Lanes 0 and 1 (columns 0 and W/2
//                        are their own mirrors) keep their own value.
//   Ci  columns inverse  the thread's own-parity bins through a W/2-point inverse codelet: G_g[y1]; decimation in time:
//                        Y[y1 + W/2 c][k] = G_0[y1] +- w^-y1 G_1[y1]; the g = 1 threads multiply by w^-y1
//   T2  transposition    (r4) only the spectrum columns 0 .. W/2 are needed (the map rows are real), so both components
//                        fit the plane at once: 16-byte complex elements plane2[W/2 g + y1][kx], pitch W/2 + 1
//                        elements, ONE write phase and ONE read phase (two barriers instead of four);
//                        thread (y = line, q = half) reads its row as G_0 +- G_1 with ds_read_b128
//   Ri  rows inverse     the real W-point row split over the thread pair by output parity: the even / odd samples are
//                        real W/2-point rows themselves (c2r through a W/4-point codelet): thread q holds corr[y][2n + q]
//   P   peak analysis    on the float64 map (B:346-358, B:381-392, B:518), 8-double record for finalize_kernel<true>.
//                        (r4) The map is never shifted or stored as a whole: v = (c - min) scale + 1e-7 is monotonic in
//                        c, so minimum, maximum, arg-max candidates and the second peak are found on the RAW cells and
//                        only the six cells of the record are shifted.  Exchange 1: per-wavefront (min, max) and every
//                        thread's row maximum -> first row that holds the maximum.  Then the threads of the 2 wv + 3
//                        rows around it (everything the flat-index neighbours and the exclusion zone of B:346-358 can
//                        touch) put their cells into a small zone buffer; exchange 2; first column of the maximum by
//                        ballot, second peak = max(row maxima outside the zone rows, zone cells outside the exclusion
//                        zone), record.  Two barriers, none at the top of the next window.
//
// The odd halves carry W/2 - 2 twiddle products per stage; the parities are assigned so that half 0 takes them in the
// column stages and half 1 in the row stages.
#pragma once
#include <math.h>
#include <stdint.h>

#include "fft_inreg_f64.hpp"

namespace tpiv {
namespace f64s {

typedef double d2 __attribute__((ext_vector_type(2)));

#if defined(__HIP_DEVICE_COMPILE__)
// all four byte lanes of a dword pair of frame a and of frame b in ONE asm block (the backend pads every asm block with
// a wait state of its own: eight single-instruction blocks cost eight s_nop).  Sums / differences of two bytes that sit
// in the same byte lane of two dwords: ONE sub-dword-addressed integer instruction each.
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

// ---- LDS accesses of the transposes.  Device: explicit ds_read_b64 / ds_read_b128 / ds_write_b64 / ds_write_b128 (the
// backend would merge neighbouring 8-byte accesses into ds_read2_b64 / ds_write2_b64 pairs, which run at HALF the LDS
// rate), reads issued in batches with the next batch in flight while the current one is consumed (lgkmcnt counts at
// most 15).  The value of a read is usable only behind lds_wait; the asm statements are volatile and clobber memory, so
// the compiler keeps them in program order among themselves and against ordinary LDS accesses.  (Constraint, kept by
// the zero-scratch build and checked by tests/test_host_logic.py + the full-size parity tests on every toolchain bump:
// the compiler must not copy or spill a destination register between the read and its wait -- it cannot know that the
// value has not landed yet.)  Host: plain loads.
#if defined(__HIP_DEVICE_COMPILE__)
template <int OFF>
__device__ __forceinline__ double lds_rd(unsigned addr) {
    double v;
    if constexpr (OFF < 65536) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    } else {                  // the offset field has 16 bits (the 128x128 plane is 130 KB): second base, shared by the reads
        const unsigned hi = addr + (unsigned)(OFF & ~0xffff);
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(hi), "n"(OFF & 0xffff) : "memory");
    }
    return v;
}
template <int OFF>
__device__ __forceinline__ d2 lds_rd2(unsigned addr) {          // 16-byte aligned address
    d2 v;
    if constexpr (OFF < 65536) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    } else {
        const unsigned hi = addr + (unsigned)(OFF & ~0xffff);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(hi), "n"(OFF & 0xffff) : "memory");
    }
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_wr(unsigned addr, double v) {
    if constexpr (OFF < 65536) {
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
    } else {
        const unsigned hi = addr + (unsigned)(OFF & ~0xffff);
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(hi), "v"(v), "n"(OFF & 0xffff) : "memory");
    }
}
template <int OFF>
__device__ __forceinline__ void lds_wr2(unsigned addr, d2 v) {   // 16-byte aligned address
    if constexpr (OFF < 65536) {
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
    } else {
        const unsigned hi = addr + (unsigned)(OFF & ~0xffff);
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(hi), "v"(v), "n"(OFF & 0xffff) : "memory");
    }
}
// wait until at most LEFT LDS operations are outstanding; the values become usable only behind it
template <int LEFT>
__device__ __forceinline__ void lds_wait(double (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(LEFT)
                 : "memory");
}
template <int LEFT>
__device__ __forceinline__ void lds_wait(d2 (&v)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(LEFT) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const double* p) { return (unsigned)(uintptr_t)p; }     // low dword of a flat LDS address
// the value of the NEIGHBOURING lane (lane ^ 1): one quad-permute DPP move per dword
__device__ __forceinline__ double dpp_xor1(double v) {
    // (bound_ctrl with old = 0: no lane of a quad permute reads out of bounds, and the compiler needs no move for `old`)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true);      // quad_perm:[1,0,3,2]
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#elif defined(__HIPCC__)
// host pass of a HIP translation unit: the kernels' bodies are parsed, never run
template <int OFF>
inline double lds_rd(unsigned) { return 0.0; }
template <int OFF>
inline d2 lds_rd2(unsigned) { return d2{0.0, 0.0}; }
template <int LEFT>
inline void lds_wait(double (&)[8]) {}
template <int LEFT>
inline void lds_wait(d2 (&)[4]) {}
template <int OFF>
inline void lds_wr(unsigned, double) {}
template <int OFF>
inline void lds_wr2(unsigned, d2) {}
inline unsigned lds_addr(const double*) { return 0u; }
inline double dpp_xor1(double v) { return v; }
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
// "no cell": below every map value (raw correlation sums are bounded by 255^2 W^4 < 1e14)
constexpr double PEAK_NONE = -1.0e300;

constexpr int ilog2_c(int n) { return n <= 1 ? 0 : 1 + ilog2_c(n / 2); }

// W = window edge (64 or 128); every function is the work of ONE thread
template <int W>
struct Split {
    static constexpr int WS = W;
    static constexpr int LOGW = ilog2_c(W);
    static constexpr int M = W / 2;          // codelet length
    static constexpr int NDW = W / 4;        // dwords per window row
    static constexpr int NT = 2 * W;         // threads per window: line = t % W, half = t / W
    static constexpr int P1 = W + 2;         // T1: doubles per stored spectrum column (pitch of plane[pos][y])
    static constexpr int P2 = M + 1;         // T2: complex elements per stored row (kx = 0 .. W/2)
    static constexpr int PLANE = W * P1;     // doubles; = 2 * W * P2
    static constexpr int ZP = W + 1;         // zone buffer: doubles per map row
    static constexpr int ZR = 11;            // zone rows the buffer holds (2 wv + 3 <= ZR, i.e. wv <= 4; else the plane serves)
    static_assert(PLANE == 2 * W * P2, "T1 and T2 share the plane");

    // lane order of the spectrum columns in the column stages: 0, W/2, 1, W-1, 2, W-2, ... (mirror pairs side by side)
    static constexpr TPIV_HD int col_of(int L) { return L == 0 ? 0 : (L == 1 ? W / 2 : ((L & 1) ? W - (L >> 1) : (L >> 1))); }
    static constexpr TPIV_HD int pos_of(int c) { return c == 0 ? 0 : (c == W / 2 ? 1 : (c < W / 2 ? 2 * c : 2 * (W - c) + 1)); }

    // ---- R: thread (y, h).  da / db: row y of frame a / b.  The samples go in as they are: the normalisation a / mean(a),
    // b / mean(b) of B:513-514 is linear, so it is ONE factor 1 / (mean(a) mean(b)) on the whole correlation map, applied
    // where map cells are shifted by the minimum (peak_shifted) -- W multiplies per thread and one of two divisions per
    // window less, at rounding-level differences (1e-16 relative) from scaling every sample first.
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

    // ---- T1 write: component COMP (0 = real, 1 = imaginary) of X[y][kx = 2m + h] -> plane[pos(kx)][y].  h is a
    // compile-time argument (the positions are offsets in the instructions); the kernel branches on it wave-uniformly.
    template <int COMP, int H_>
    static TPIV_HD void t1_write(const cd (&x)[M], int y, double* plane) {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(plane + y);
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            lds_wr<pos_of(2 * m + H_) * P1 * 8>(base, COMP ? x[FFT_POS<m, M>].y : x[FFT_POS<m, M>].x);
        });
#else
        static_for<0, M>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            plane[pos_of(2 * m + H_) * P1 + y] = COMP ? x[FFT_POS<m, M>].y : x[FFT_POS<m, M>].x;
        });
#endif
    }
    // ---- T1 read: thread (position L, parity g): u[i] = X[i][k] +- X[i + M][k], k = col_of(L)
    template <int COMP>
    static TPIV_HD void t1_read(cd (&u)[M], int L, int g, const double* plane) {
        const double sg = g ? -1.0 : 1.0;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(plane + L * P1);
        constexpr int NB = M / 4;                                   // batches of 4 rows i = 4 reads of 16 bytes
        d2 v[2][4];
        auto issue = [&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            v[b_ & 1][0] = lds_rd2<(4 * b_) * 8>(base);
            v[b_ & 1][1] = lds_rd2<(4 * b_ + M) * 8>(base);
            v[b_ & 1][2] = lds_rd2<(4 * b_ + 2) * 8>(base);
            v[b_ & 1][3] = lds_rd2<(4 * b_ + 2 + M) * 8>(base);
        };
        issue(std::integral_constant<int, 0>{});
        static_for<0, NB>([&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            if constexpr (b_ + 1 < NB) issue(std::integral_constant<int, b_ + 1>{});
            lds_wait<(b_ + 1 < NB) ? 4 : 0>(v[b_ & 1]);
            static_for<0, 4>([&](auto ic) TPIV_LAMBDA_INLINE {
                constexpr int q = decltype(ic)::value, i = 4 * b_ + q;
                const double lo_ = (q & 1) ? v[b_ & 1][q & 2].y : v[b_ & 1][q & 2].x;
                const double hi_ = (q & 1) ? v[b_ & 1][(q & 2) + 1].y : v[b_ & 1][(q & 2) + 1].x;
                const double r = lo_ + sg * hi_;
                if constexpr (COMP) u[i].y = r;
                else u[i].x = r;
            });
        });
#else
        const double* col = plane + L * P1;
        for (int i = 0; i < M; ++i) {
            const double r = col[i] + sg * col[i + M];
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

    // ---- X: cross-spectrum z -> pz.  sh(value, reg, comp) returns the MIRROR thread's value of register `reg`: the
    // neighbouring lane's (device: quad-permute DPP of `value`; the host harness looks it up), the thread's own in the
    // lanes of the self-mirrored columns 0 and W/2.
    // bin ky = 2m + G sits at z[FFT_POS<m>]; its mirror -ky = 2 m' + G with m' = (M - m) % M (G = 0), M - 1 - m (G = 1)
    template <int G, typename SH>
    static TPIV_HD void cross_spectrum_g(const cd (&z)[M], cd (&pz)[M], SH&& sh) {
        constexpr int NPAIR = G ? M / 2 : M / 2 + 1;
        static_for<0, NPAIR>([&](auto mc) TPIV_LAMBDA_INLINE {
            constexpr int m = decltype(mc)::value;
            constexpr int m2 = G ? M - 1 - m : (M - m) % M;
            constexpr int p1 = FFT_POS<m, M>, p2 = FFT_POS<m2, M>;
            cd z1 = z[p1];
            pin(z1);
            if constexpr (m == m2) {
                const cd m1{sh(z1.x, p1, 0), sh(z1.y, p1, 1)};
                pz[p1] = cross_bin(z1, m1);
                pin(pz[p1]);
            } else {
                cd z2 = z[p2];
                pin(z2);
                const cd m1{sh(z2.x, p2, 0), sh(z2.y, p2, 1)};      // Z(-ky, -k)
                const cd m2v{sh(z1.x, p1, 0), sh(z1.y, p1, 1)};     // Z(+ky, -k): the mirror of bin -ky
                pz[p1] = cross_bin(z1, m1);
                pz[p2] = cross_bin(z2, m2v);
                // (the empty asm statements execute in program order: with the inputs AND the results of a pair passing
                //  through one, the pairs stay one after the other -- left alone, instruction selection emits the 128
                //  moves of all pairs first and the register allocator spills them)
                pin(pz[p1]);
                pin(pz[p2]);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            // keep the exchange of a bin pair together: hoisted ahead, the moves of every pair hold their results in
            // registers next to the 128 of the spectrum
            __builtin_amdgcn_sched_barrier(0);
#endif
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
    // ---- T2 write: thread (position L, g), k = col_of(L): plane2[M g + y1][k] <- G_g[y1][k] for the columns k <= W/2
    // (the others are the mirrors nobody reads: the map rows are real)
    static TPIV_HD void t2_write(const cd (&t)[M], int L, int g, double* plane) {
        const int k = col_of(L);
        if (k > M) return;
        double* el = plane + 2 * ((M * g) * P2 + k);
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(el);
        static_for<0, M>([&](auto yc) TPIV_LAMBDA_INLINE {
            constexpr int y1 = decltype(yc)::value;
            lds_wr2<y1 * P2 * 16>(base, d2{t[FFT_POS<y1, M>].x, t[FFT_POS<y1, M>].y});
        });
#else
        static_for<0, M>([&](auto yc) TPIV_LAMBDA_INLINE {
            constexpr int y1 = decltype(yc)::value;
            el[2 * y1 * P2] = t[FFT_POS<y1, M>].x;
            el[2 * y1 * P2 + 1] = t[FFT_POS<y1, M>].y;
        });
#endif
    }
    // ---- T2 read: thread (y, q): Y[kx] = G_0[y % M][kx] +- G_1[y % M][kx], kx = 0..M
    static TPIV_HD void t2_read(cd (&Y)[M + 1], int y, const double* plane) {
        const double sg = y < M ? 1.0 : -1.0;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(plane + 2 * (y & (M - 1)) * P2);
        constexpr int NB = (M + 2) / 2;                              // batches of 2 columns = 4 reads of 16 bytes; the last holds column M alone
        d2 v[2][4];
        auto issue = [&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            v[b_ & 1][0] = lds_rd2<(2 * b_) * 16>(base);
            v[b_ & 1][1] = lds_rd2<(M * P2 + 2 * b_) * 16>(base);
            if constexpr (2 * b_ + 1 <= M) {
                v[b_ & 1][2] = lds_rd2<(2 * b_ + 1) * 16>(base);
                v[b_ & 1][3] = lds_rd2<(M * P2 + 2 * b_ + 1) * 16>(base);
            }
        };
        issue(std::integral_constant<int, 0>{});
        static_for<0, NB>([&](auto bc) TPIV_LAMBDA_INLINE {
            constexpr int b_ = decltype(bc)::value;
            if constexpr (b_ + 1 < NB) issue(std::integral_constant<int, b_ + 1>{});
            // outstanding behind this batch: the next one (4 reads, the last batch holds 2)
            lds_wait<(b_ + 1 < NB) ? (b_ + 2 == NB ? 2 : 4) : 0>(v[b_ & 1]);
            Y[2 * b_] = cd{v[b_ & 1][0].x + sg * v[b_ & 1][1].x, v[b_ & 1][0].y + sg * v[b_ & 1][1].y};
            if constexpr (2 * b_ + 1 <= M)
                Y[2 * b_ + 1] = cd{v[b_ & 1][2].x + sg * v[b_ & 1][3].x, v[b_ & 1][2].y + sg * v[b_ & 1][3].y};
        });
#else
        const double* r0 = plane + 2 * (y & (M - 1)) * P2;
        const double* r1 = r0 + 2 * M * P2;
        for (int kx = 0; kx <= M; ++kx)
            Y[kx] = cd{r0[2 * kx] + sg * r1[2 * kx], r0[2 * kx + 1] + sg * r1[2 * kx + 1]};
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

    // ---- P: peak analysis in fftshift coordinates (fy = (y + W/2) % W, fx likewise) on the RAW cells.  Thread (y, q)
    // holds c[i] = corr[y][x = 4 (i / 2) + 2 (i % 2) + q], i.e. the shifted column fx0(i) + q.
    static constexpr TPIV_HD int fx0(int i) { return (4 * (i >> 1) + 2 * (i & 1) + W / 2) & (W - 1); }
    static TPIV_HD int frow(int y) { return (y + W / 2) & (W - 1); }
    // one pass over the raw cells gives their minimum AND their maximum: v = (c - min) scale + 1e-7 is monotonic in c, so
    // the maximum of the shifted cells is the shifted maximum (exactly: the same two roundings)
    static TPIV_HD void peak_local_minmax(const double (&c)[M], double& mn, double& mx) {
        mn = c[0];
        mx = c[0];
#pragma unroll
        for (int i = 1; i < M; ++i) {
            mn = dmin2(mn, c[i]);
            mx = dmax2(mx, c[i]);
        }
    }
    // The rows whose cells the record and the second peak may need one by one: the flat-index neighbours of the arg-max m
    // (B:385-392) lie in the map rows my - 1 .. my + 1, the exclusion zone {clamp(m + i + W j), |i|, |j| <= wv} (B:346-358)
    // in the rows my - wv - 1 .. my + wv + 1 (row wrap of m + i) -- the clamps to cell 0 / cell W^2 - 1 only occur when
    // those rows reach the first / last map row.  Zone row zr <-> map row zlo + zr, zlo = my - wv - 1 (may be negative:
    // rows outside the map are never touched), nz = 2 wv + 3.
    static TPIV_HD void peak_zone_write(const double (&c)[M], int y, int q, int zlo, int nz, double* zone) {
        const int zr = frow(y) - zlo;
        if (zr < 0 || zr >= nz) return;
        double* row = zone + zr * ZP + q;
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned base = lds_addr(row);
        static_for<0, M>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int i = decltype(ic)::value;
            lds_wr<fx0(i) * 8>(base, c[i]);
        });
#else
        static_for<0, M>([&](auto ic) TPIV_LAMBDA_INLINE {
            constexpr int i = decltype(ic)::value;
            row[fx0(i)] = c[i];
        });
#endif
    }
    // is the flat cell f inside the exclusion zone of m?  f = m + i + W j with |i|, |j| <= wv (wv < W/2: unique), or one
    // of the two clamps
    // (0 / 1, written without short-circuit operators: the kernel evaluates it per lane inside straight-line code)
    static TPIV_HD int peak_excluded(int f, int m, int wv) {
        constexpr int KD = W * W;
        const int d = f - m;
        const int j0 = (d + W / 2) >> LOGW;             // floor((d + W/2) / W): arithmetic shift
        const int r = d - (j0 << LOGW);                 // -W/2 <= r < W/2
        int ex = (int)((unsigned)(r + wv) <= (unsigned)(2 * wv)) & (int)((unsigned)(j0 + wv) <= (unsigned)(2 * wv));
        ex |= (int)(f == 0) & (int)(m - wv - wv * W <= 0);
        ex |= (int)(f == KD - 1) & (int)(m + wv + wv * W >= KD - 1);
        return ex;
    }
    // The same test taken apart for the kernel's scan of the zone rows (zone row zr <-> map row fy = my - wv - 1 + zr): with
    // dx = fx - mx and q = floor((dx + W/2) / W) in {-1, 0, 1} the cell is m + i + W j with i = dx - q W, j = (fy - my) + q =
    // zr - wv - 1 + q.  The column part depends on the lane only, the row part on the wave-uniform row and q.
    static TPIV_HD bool peak_col_ok(int fx, int mx, int wv, int& q) {
        const int dx = fx - mx;
        q = (dx + W / 2) >> LOGW;
        return (unsigned)(dx - (q << LOGW) + wv) <= (unsigned)(2 * wv);
    }
    static TPIV_HD bool peak_excluded_zr(bool colok, int q, int zr, int fy, int fx, int wv, bool clamp_lo, bool clamp_hi) {
        bool ex = colok & ((unsigned)(zr - 1 + q) <= (unsigned)(2 * wv));
        ex |= (clamp_lo & (fy == 0)) & (fx == 0);              // clamp_lo: m - wv - wv W <= 0
        ex |= (clamp_hi & (fy == W - 1)) & (fx == W - 1);      // clamp_hi: m + wv + wv W >= W^2 - 1
        return ex;
    }
    // record for finalize_kernel<true>: slot 0..7 = c[m], c[left], c[right], c[top], c[bot], c[m2], m, dead
    // (flat-index neighbours and fix-ups of B:385-392; sv_raw == PEAK_NONE: nothing left outside the exclusion zone -- the
    // reference's second arg-max then runs over the zeroed map: c[m2] = 0, ratio = +inf).  The cells come out of the
    // zone buffer and are shifted here.
    static TPIV_HD double peak_record_slot(int slot, int m, double sv_raw, bool dead, const double* zone, int zlo, double cmin,
                                           double scale) {
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
        double raw = zone[((qi >> LOGW) - zlo) * ZP + (qi & (W - 1))];
        raw = slot == 5 ? sv_raw : raw;
        double outv = peak_shifted(raw, cmin, scale);
        outv = (slot == 5 && sv_raw == PEAK_NONE) ? 0.0 : outv;
        outv = slot == 6 ? (double)m : outv;
        outv = slot == 7 ? (dead ? 1.0 : 0.0) : outv;
        return outv;
    }
};

}  // namespace f64s
}  // namespace tpiv
