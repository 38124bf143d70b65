// Fused PIV tile kernel, second generation (tile sizes 8..64) for gfx950 (MI355X).
//
// Replaces the ATen kernel sequences of the reference's extended_search_area_piv
// (PIVbackend.py:459-520), piv_iteration_DWS/CWS.__call__ (B:757-812 / B:690-740),
// interpolation_DWS (B:197-216), biliniar_interpolation_CWS (B:147-194), correalte_fft
// (B:249-257), correlation_to_displacement (B:360-422) and peak2peak_secondpeak (B:346-358)
// with ONE launch per pass: uint8 frames in HBM -> (u, v, invalid) per window.
//
// Mapping (one 64-lane wavefront = one workgroup = 64/WS windows):
//   * lane = one image ROW of one window.  The row (WS bytes per frame) is fetched with
//     16-byte loads straight into VGPRs; overlapping windows re-read through L2 (a per-XCD work
//     queue keeps the wavefronts of an XCD on adjacent windows).  Both frames are packed as a + i*b.
//     Bilinear (CWS) passes fetch the (WS+1)^2 source patch of a window with all its lanes together and
//     hand the rows over through LDS (CoopGeo below).
//   * all 1-D FFTs (WS points) run per lane, entirely in registers (fft_inreg.hpp).
//   * LDS is used only to transpose between the row and the column transform, in 32x33 (or
//     WSx(WS+1)) tiles; a 64x64 tile is transposed as four 32x32 blocks after a
//     v_permlane32_swap of the off-diagonal blocks.  The last transform is a c2r one: only spectrum
//     columns 0..WS/2 cross the LDS the second time (transpose_half).  Kernels at three wavefronts per SIMD move one
//     float plane at a time (8.4 KB per wavefront), the others complex elements (16.9 KB).
//   * the k <-> -k partner of the packed spectrum is fetched with ds_bpermute (no LDS memory).
//   * peak search: plain max scans per lane, DPP / permlane reductions, one LDS row lookup for the
//     arg-max position, sign-bit exclusion for the second peak (compares and selects cost twice a
//     plain fp32 instruction on MI355X); an 8-float record per window goes to finalize_kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fft_inreg.hpp"
#include "piv_kernels.h"

namespace tpiv {

template <int WS, bool PLANAR_ = false>
struct TileGeo {
    static constexpr int WPW = 64 / WS;                 // windows per wavefront
    static constexpr int TS = WS > 32 ? 32 : WS;        // transposition tile edge
    static constexpr int PITCH = TS + 1;                // complex elements per tile row
    static constexpr int NT = 64 / TS;                  // tiles per wavefront
    static constexpr int TILE = TS * PITCH;             // complex elements per tile
    // PLANAR: the real and the imaginary plane pass through the tile one after the other -- half
    // the LDS per wavefront (8.4 KB), which is what lets a third wavefront fit per SIMD.
    static constexpr bool PLANAR = PLANAR_;
    static constexpr int HROWS = WS / 2 + 1;            // spectrum columns 0..WS/2 (c2r last transform)
    static constexpr int HTILE = HROWS * PITCH;         // elements per half-transposition tile (WS <= 32)
    static constexpr int LDS_FLOATS = NT * TILE * (PLANAR ? 1 : 2);
    static constexpr int MAP_PITCH = WS + 1;            // floats per row of the correlation map
    static constexpr int NDW = WS / 4;                  // dwords per window row
};

// ---- reductions over the WS lanes of one window, all lanes receive the result.
// Cross-lane steps stay in the VALU (DPP quad/row permutes, v_permlane16/32_swap): the usual
// __shfl_xor goes through ds_bpermute, i.e. one LDS round trip per step, and five dependent
// round trips per reduction dominated the latency of the peak search.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(dpp_i<CTRL>(__float_as_int(v)));
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7-i inside each 8
constexpr int DPP_ROW_MIRROR = 0x140;   // lane i <-> 15-i inside each 16

// partner values for the reduction step that spans 2*HALF lanes (any pairing of the two halves works)
template <int HALF>
__device__ __forceinline__ void partner2(int a, int& pa) {
    if constexpr (HALF == 1) pa = dpp_i<DPP_XOR1>(a);
    else if constexpr (HALF == 2) pa = dpp_i<DPP_XOR2>(a);
    else if constexpr (HALF == 4) pa = dpp_i<DPP_HALF_MIRROR>(a);
    else if constexpr (HALF == 8) pa = dpp_i<DPP_ROW_MIRROR>(a);
}

template <int WS, typename T, typename OP>
__device__ __forceinline__ T grp_reduce(T v, OP op) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "one or two dwords");
    constexpr int NW = sizeof(T) / 4;
    union U {
        T t;
        int w[NW];
    };
    static_for<0, 4>([&](auto sc) TPIV_LAMBDA_INLINE {
        constexpr int half = 1 << decltype(sc)::value;
        if constexpr (half < WS) {
            U a, b;
            a.t = v;
#pragma unroll
            for (int q = 0; q < NW; ++q) partner2<half>(a.w[q], b.w[q]);
            v = op(v, b.t);
        }
    });
    if constexpr (WS >= 32) {           // rows of 16 lanes: {R0,R1,R2,R3} -> {R0,R0,R2,R2} (+) {R1,R1,R3,R3}
        U a, x, y;
        a.t = v;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            auto r = __builtin_amdgcn_permlane16_swap((unsigned)a.w[q], (unsigned)a.w[q], false, false);
            x.w[q] = (int)r[0];
            y.w[q] = (int)r[1];
        }
        v = op(x.t, y.t);
    }
    if constexpr (WS >= 64) {           // halves of 32 lanes
        U a, x, y;
        a.t = v;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            auto r = __builtin_amdgcn_permlane32_swap((unsigned)a.w[q], (unsigned)a.w[q], false, false);
            x.w[q] = (int)r[0];
            y.w[q] = (int)r[1];
        }
        v = op(x.t, y.t);
    }
    return v;
}

template <int WS>
__device__ __forceinline__ float grp_sum(float v) {
    return grp_reduce<WS>(v, [](float a, float b) TPIV_LAMBDA_INLINE { return a + b; });
}

template <int WS>
__device__ __forceinline__ float grp_min(float v) {
    return grp_reduce<WS>(v, [](float a, float b) TPIV_LAMBDA_INLINE { return fminf(a, b); });
}

__device__ __forceinline__ float fetch_clamped_t(const uint8_t* __restrict__ f, long long q, int HW) {
    q = q < 0 ? 0 : (q > (long long)(HW - 1) ? (long long)(HW - 1) : q);      // B:177-180, B:214
    return (float)f[q];
}

__device__ __forceinline__ int f2i_sat_t(float v) {
    v = fminf(fmaxf(v, -1073741824.f), 1073741824.f);
    return (int)v;
}

// byte k of a row held as dwords (compile-time k -> v_cvt_f32_ubyteN)
template <int K, int N>
__device__ __forceinline__ float byte_f(const uint32_t (&d)[N]) {
    return (float)((d[K >> 2] >> (8 * (K & 3))) & 0xffu);
}

// ... the same byte REINTERPRETED as a float32 (a denormal: b x 2^-149), for use as a multiplicand: the compiler folds
// the byte extraction into the multiply (v_mul_f32_sdwa src_sel:BYTE_k) -- one instruction instead of v_cvt_f32_ubyte +
// v_mul_f32.  The other factor carries 2^125, so the product is b x w x 2^-24: exact scaling, no rounding differs (the
// kernels run with float32 denormals on: .amdhsa_float_denorm_mode_32 3).  gfx950 has no SDWA form of v_fmac / v_fma.
// OFF: measured in round 5 (same box, A B A B; results bit-identical, 82 parity tests green): the 66 v_mul_f32_sdwa per item
// replace 66 conversions -- and the 32x32 CWS pass takes 7.27 / 7.33 ms with them against 7.19 / 7.26 without, the 64x64 CWS
// pass 2.35 against 2.27, the 16x16 pass 1.93 against 1.91: the multiply with a denormal (or SDWA) operand is not a
// full-rate instruction on this chip.
#ifndef TPIV_SDWA_LERP
#define TPIV_SDWA_LERP 0
#endif
constexpr float SDWA_UP = 0x1p125f, SDWA_DN = 0x1p-24f, SDWA_INV = 0x1p24f;      // 2^125 x 2^-149 = 2^-24
template <int K, int N>
__device__ __forceinline__ float byte_d(const uint32_t (&d)[N]) {
    return __uint_as_float((d[K >> 2] >> (8 * (K & 3))) & 0xffu);
}

// N dwords from a byte address of any alignment (global memory takes unaligned dword loads)
template <int N>
__device__ __forceinline__ void load_dwords(const uint8_t* __restrict__ p, uint32_t (&d)[N]) {
    __builtin_memcpy(&d[0], p, 4 * N);
}

// One bilinear sample exactly as PIVbackend.py:187-193 evaluates it in float32: every product
// and sum rounded separately, left to right -- FMA contraction must stay off in here.
__device__ __forceinline__ float bilerp_ref(float f11, float f21, float f12, float f22, float wx_up,
                                            float wx_dn, float wy_up, float wy_dn, bool degenerate) {
#pragma clang fp contract(off)
    float r = (f11 * wx_up) * wy_up;
    r = r + (f21 * wx_dn) * wy_up;
    r = r + (f12 * wx_up) * wy_dn;
    r = r + (f22 * wx_dn) * wy_dn;
    return degenerate ? f11 : r;          // B:170, B:193: either coordinate integral -> f(floor y, floor x)
}

// The same with the quirk as a per-lane bit mask (0 or ~0): one v_bfi_b32 instead of a select through
// an SGPR lane mask (a select costs two plain instructions on MI355X).
__device__ __forceinline__ float bilerp_ref_m(float f11, float f21, float f12, float f22, float wx_up,
                                              float wx_dn, float wy_up, float wy_dn, unsigned degenerate_mask) {
#pragma clang fp contract(off)
    float r = (f11 * wx_up) * wy_up;
    r = r + (f21 * wx_dn) * wy_up;
    r = r + (f12 * wx_up) * wy_dn;
    r = r + (f22 * wx_dn) * wy_dn;
    const unsigned bits = (degenerate_mask & __float_as_uint(f11)) | (~degenerate_mask & __float_as_uint(r));
    return __uint_as_float(bits);
}

// ---- diagnostic build only: in-kernel phase stamps (s_memtime), summed per wavefront in scalar
// registers and added to p.stamps at the end.  Compiled out of the production library.
#ifdef TPIV_STAMPS
#define TPIV_STAMP_DECL unsigned long long st_acc[16] = {}; unsigned long long st_prev = 0; unsigned st_iter = 0; \
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define TPIV_STAMP_START                                                                  \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");    \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#define TPIV_STAMP(i)                                                                     \
    do {                                                                                  \
        unsigned long long st_t;                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                \
        st_acc[i] += st_t - st_prev;                                                      \
        st_prev = st_t;                                                                   \
    } while (0)
#define TPIV_STAMP_FLUSH(pp)                                                              \
    do {                                                                                  \
        if ((pp).stamps != nullptr && threadIdx.x == 0) {                                 \
            for (int q_ = 0; q_ < 16; ++q_) atomicAdd(&(pp).stamps[q_], st_acc[q_]);      \
            atomicAdd(&(pp).stamps[16], (unsigned long long)st_iter);                     \
            atomicAdd(&(pp).stamps[17], __builtin_amdgcn_s_memtime() - st_t0);            \
            atomicAdd(&(pp).stamps[18], __builtin_amdgcn_s_memrealtime() - st_r0);        \
        }                                                                                 \
    } while (0)
#else
#define TPIV_STAMP_DECL
#define TPIV_STAMP_START
#define TPIV_STAMP(i)
#define TPIV_STAMP_FLUSH(pp)
#endif

// n / d for 0 <= n < 2^31 with a host-made (magic, shift): exact (magic = ceil(2^shift / d),
// shift = 31 + ceil(log2 d)); two scalar multiplies for wave-uniform n, v_mul_hi otherwise.
__device__ __forceinline__ int fast_div(int n, unsigned magic, int shift) {
    return (int)(((unsigned long long)(unsigned)n * magic) >> shift);
}
inline void fast_div_setup(unsigned d, unsigned& magic, int& shift) {
    int s_ = 0;
    while ((1ull << s_) < d) ++s_;
    shift = 31 + s_;
    magic = (unsigned)(((1ull << shift) + d - 1) / d);
}

// The lane id, recomputed (two VALU instructions) and opaque to CSE: values derived from it at the
// point of use do not have to stay in registers (or get spilled) across the transforms.
__device__ __forceinline__ int fresh_lane() {
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// A workgroup is ONE wavefront, and the LDS executes one wavefront's DS instructions in order,
// so exchanging data between lanes through LDS needs no s_barrier and no s_waitcnt: only the
// compiler must keep the program order of the LDS accesses.  (A real __syncthreads() would also
// emit vmcnt(0) and drain the next item's prefetch loads.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- transposition of the wavefront's tile(s) through LDS ---------------------------------
// in:  lane (w, i) holds line i of its window, element k at in[POS(k)]  (POS = digit-reversed
//      position when DIGITREV, else k)
// out: lane (w, j) holds element j of every line: out[i] = element (line i, position j)
template <int WS, bool DIGITREV, bool PLANAR>
__device__ __forceinline__ void transpose_tile(cf (&a)[WS], float* lds, int lane) {
    using G = TileGeo<WS, PLANAR>;
    constexpr int P = G::PITCH;
    cf* tile = reinterpret_cast<cf*>(lds);
    if constexpr (PLANAR && WS > 32) {
        // 64x64, one float plane at a time through two 32x33 float tiles (8.4 KB per wavefront, which
        // is what lets three wavefronts per SIMD fit in the LDS): swap the off-diagonal blocks
        // between the lane halves, then transpose the four blocks of each plane, two at a time.
        static_for<0, 32>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            constexpr int lo = DIGITREV ? FFT_POS<k, WS> : k;
            constexpr int hi = DIGITREV ? FFT_POS<k + 32, WS> : k + 32;
            auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[lo].x), __float_as_uint(a[hi].x),
                                                       false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[lo].y), __float_as_uint(a[hi].y),
                                                       false, false);
            a[lo].x = __uint_as_float(rx[0]);
            a[hi].x = __uint_as_float(rx[1]);
            a[lo].y = __uint_as_float(ry[0]);
            a[hi].y = __uint_as_float(ry[1]);
        });
        float* t = lds + (lane >> 5) * G::TILE;
        const int i = lane & 31;
        auto plane = [&](auto comp) TPIV_LAMBDA_INLINE {
            constexpr bool Y = decltype(comp)::value;
            float lowhalf[32];
            wave_sync();
            static_for<0, 32>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                constexpr int src = DIGITREV ? FFT_POS<k, WS> : k;
                t[i * P + k] = Y ? a[src].y : a[src].x;
            });
            wave_sync();
#pragma unroll
            for (int r = 0; r < 32; ++r) lowhalf[r] = t[r * P + i];
            wave_sync();
            static_for<0, 32>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                constexpr int src = DIGITREV ? FFT_POS<k + 32, WS> : k + 32;
                t[i * P + k] = Y ? a[src].y : a[src].x;
            });
            wave_sync();
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if constexpr (Y) {
                    a[32 + r].y = t[r * P + i];
                    a[r].y = lowhalf[r];
                } else {
                    a[32 + r].x = t[r * P + i];
                    a[r].x = lowhalf[r];
                }
            }
        };
        plane(std::false_type{});
        plane(std::true_type{});
    } else if constexpr (PLANAR) {
        float* t = lds + (lane / WS) * G::TILE;
        const int i = lane % WS;
        wave_sync();
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            constexpr int src = DIGITREV ? FFT_POS<k, WS> : k;
            t[i * P + k] = a[src].x;
        });
        wave_sync();
#pragma unroll
        for (int r = 0; r < WS; ++r) a[r].x = t[r * P + i];
        wave_sync();
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            constexpr int src = DIGITREV ? FFT_POS<k, WS> : k;
            t[i * P + k] = a[src].y;
        });
        wave_sync();
#pragma unroll
        for (int r = 0; r < WS; ++r) a[r].y = t[r * P + i];
    } else if constexpr (WS <= 32) {
        cf* t = tile + (lane / WS) * G::TILE;
        const int i = lane % WS;
        wave_sync();
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            constexpr int src = DIGITREV ? FFT_POS<k, WS> : k;
            t[i * P + k] = a[src];
        });
        wave_sync();
#pragma unroll
        for (int r = 0; r < WS; ++r) a[r] = t[r * P + i];
    } else {
        // 64x64 = 2x2 blocks of 32x32: swap the off-diagonal blocks between the lane halves,
        // then transpose the four blocks, two at a time, through the two 32x33 tiles
        static_for<0, 32>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            constexpr int lo = DIGITREV ? FFT_POS<k, WS> : k;
            constexpr int hi = DIGITREV ? FFT_POS<k + 32, WS> : k + 32;
            auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[lo].x), __float_as_uint(a[hi].x),
                                                       false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[lo].y), __float_as_uint(a[hi].y),
                                                       false, false);
            a[lo].x = __uint_as_float(rx[0]);
            a[hi].x = __uint_as_float(rx[1]);
            a[lo].y = __uint_as_float(ry[0]);
            a[hi].y = __uint_as_float(ry[1]);
        });
        cf* t = tile + (lane >> 5) * G::TILE;
        const int i = lane & 31;
        cf lowhalf[32];
        wave_sync();
        static_for<0, 32>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            constexpr int src = DIGITREV ? FFT_POS<k, WS> : k;
            t[i * P + k] = a[src];
        });
        wave_sync();
#pragma unroll
        for (int r = 0; r < 32; ++r) lowhalf[r] = t[r * P + i];
        wave_sync();
        static_for<0, 32>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            constexpr int src = DIGITREV ? FFT_POS<k + 32, WS> : k + 32;
            t[i * P + k] = a[src];
        });
        wave_sync();
#pragma unroll
        for (int r = 0; r < 32; ++r) a[32 + r] = t[r * P + i];
#pragma unroll
        for (int r = 0; r < 32; ++r) a[r] = lowhalf[r];
    }
}

// ---- half transposition for a real (c2r) last transform ---------------------------------------------
// in:  lane (w, kx) holds column kx of its window's half-transformed spectrum, row y at in[FFT_POS<y>]
// out: lane (w, y) holds g[kx] = element (row y, column kx) for kx = 0..WS/2 only (the spectrum of a
//      real row is Hermitian).  Only lanes kx <= WS/2 write.
template <int WS, bool PLANAR>
__device__ __forceinline__ void transpose_half(const cf (&a)[WS], cf (&g)[WS / 2 + 1], float* lds, int lane) {
    static_assert(WS <= 32 || WS == 64, "one tile per window");
    using G = TileGeo<WS, PLANAR>;
    constexpr int P = G::PITCH, M = WS / 2;
    const int i = lane % WS;
    if constexpr (WS == 64) {
        // 64x64, planar: element (row y, column kx) at y * 33 + kx -- 64 x 33 floats, exactly the two 32x33 tiles;
        // lanes kx <= 32 write with stride 1, lane y reads with stride 33: conflict-free, and no lane-half swaps
        constexpr int Q = M + 1;
        static_assert(WS * Q * (PLANAR ? 1 : 2) <= G::LDS_FLOATS, "half-spectrum plane fits the tile");
        // (the values are pinned in front of the conditional stores: left alone, the compiler sinks the end of the
        //  column transform into the branch and keeps its inputs alive across it -- 32 spills at the register cap)
        cf b[WS];
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            b[k] = a[FFT_POS<k, WS>];
            asm volatile("" : "+v"(b[k].x), "+v"(b[k].y));
        });
        wave_sync();
        if constexpr (!PLANAR) {              // complex elements (two wavefronts per SIMD: 16.9 KB tile)
            cf* t = reinterpret_cast<cf*>(lds);
            if (i <= M) {
                static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    t[k * Q + i] = b[k];
                });
            }
            wave_sync();
#pragma unroll
            for (int r = 0; r <= M; ++r) g[r] = t[i * Q + r];
        } else {
        if (i <= M) {
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                lds[k * Q + i] = b[k].x;
            });
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r <= M; ++r) g[r].x = lds[i * Q + r];
        wave_sync();
        if (i <= M) {
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                lds[k * Q + i] = b[k].y;
            });
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r <= M; ++r) g[r].y = lds[i * Q + r];
        }
    } else if constexpr (PLANAR) {
        float* t = lds + (lane / WS) * G::HTILE;
        wave_sync();
        if (i <= M) {
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                t[i * P + k] = a[FFT_POS<k, WS>].x;
            });
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r <= M; ++r) g[r].x = t[r * P + i];
        wave_sync();
        if (i <= M) {
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                t[i * P + k] = a[FFT_POS<k, WS>].y;
            });
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r <= M; ++r) g[r].y = t[r * P + i];
    } else {
        cf* t = reinterpret_cast<cf*>(lds) + (lane / WS) * G::HTILE;
        wave_sync();
        if (i <= M) {
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                t[i * P + k] = a[FFT_POS<k, WS>];
            });
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r <= M; ++r) g[r] = t[r * P + i];
    }
}

// ---- stage 0: window rows, split into "issue the loads" and "turn them into samples" so that
//      the loads of item i+1 can be in flight while item i is transformed (software prefetch)
struct ItemGeom {
    int pair, win, y0, x0;
    int active;          // int, not bool: byte members of loop-carried structs get copied through scratch
    size_t fidx;
};

// first pass: rows loaded in 16-byte chunks by the window's lanes together (issue_rows / convert_rows, MODE_PASS1).
// OFF: measured in round 5 (same box, A B A B): locating pass of configs[1] 6.86 / 6.90 ms without, 7.05 / 7.16 ms with; 32x32
// (configs[3]) 1.58 against 1.61 ms -- the hand-over through the tile costs more than the address unit saves (the shifted
// CWS passes, whose patches are (WS + 1)^2 and fetched twice per row otherwise, keep their chunk loads: CoopGeo).
#ifndef TPIV_COOP1
#define TPIV_COOP1 0
#endif
template <int WS>
struct Coop1 {
    static constexpr bool ON = TPIV_COOP1 && (WS == 64 || WS == 32);
    static constexpr int P16 = WS / 16;                       // chunks per row = chunks per lane
    static constexpr int PITCH4 = WS / 16 + 1;                // hand-over tile: 16-byte units per row (16-byte reads cover all banks)
};
template <int WS, int MODE>
struct RawRows;
template <int WS>
struct RawRows<WS, MODE_PASS1> {
    uint32_t a[WS / 4], b[WS / 4];
};
template <int WS>
struct RawRows<WS, MODE_DWS> {
    uint32_t a[WS / 4], b[WS / 4];
    long long qa, qb;
    int reg;             // wave-uniform: every lane's rows can take the wide-load path
    int cls;             // per lane, 2 bits per row: 0 in frame, 1 entirely before it, 2 entirely behind it
    int fix;             // wave-uniform: some lane has a row of class 1 or 2
};
// CWS source patches, loaded by the window's lanes TOGETHER (tile sizes 16 and 32).  A shifted window reads
// (WS+1) x (WS+1) source pixels per frame; with lane = row every load instruction touched 64 different rows,
// i.e. 64-128 cache lines (TCP_TOTAL_CACHE_ACCESSES: 151 per load instruction against 58 in pass 1, about
// one tag look-up per CU cycle over the whole kernel), each row was fetched twice (as the upper row of one
// lane and the lower row of its neighbour), and at a power-of-two row pitch all those lines fall into the
// same cache banks (2048-pixel frames: the 32x32 CWS pass took 32.8 us per pair, 28.9 us at a pitch of
// 2064).  Here the patch is cut into 16-byte chunks, chunk ci = row * RPG + part, and lane r of the window
// loads chunks r, r + WS, ...: consecutive lanes read consecutive pieces of a row.  convert_rows parks the
// chunks in LDS (row pitch an odd number of 16-byte units: conflict-free 16-byte reads at any row offset) and every lane reads
// the two rows it interpolates between.
template <int WS>
struct CoopGeo {
#ifndef TPIV_COOP
#define TPIV_COOP 1
#endif
    static constexpr bool ON = TPIV_COOP && (WS == 16 || WS == 32 || WS == 64);
    static constexpr int NROW = WS + 1;                       // patch rows
    static constexpr int RPG = (WS + 1 + 15) / 16;            // 16-byte chunks loaded per row
    static constexpr int LOADED = 16 * RPG;                   // bytes loaded per row
    static constexpr int T = NROW * RPG;                      // chunks per patch
    static constexpr int NCH = (T + WS - 1) / WS;             // chunks per lane
    static constexpr int PITCH4 = RPG | 1;                    // LDS row pitch in 16-byte units (odd: see above)
    static constexpr int PATCH4 = NROW * PITCH4;              // 16-byte units per patch
    // The two patches of a window form a block whose size is a multiple of 256 bytes: a 16-byte row read covers the
    // 64 banks with 16 lanes, and for 16x16 (32x32) tiles those lanes belong to two windows -- with the blocks a
    // multiple of 256 bytes apart the rows of both interleave on the same 12-dword lattice (conflict-free); at the
    // natural size (1632 bytes for 16x16) they collided: SQ_LDS_BANK_CONFLICT 35 % of the LDS-active cycles of the
    // 16x16 CWS pass (profiles/r02).
    static constexpr int BLOCK4 = (2 * PATCH4 + 15) / 16 * 16;
    static constexpr int FLOATS = (64 / WS) * BLOCK4 * 4;     // all patches of a wavefront, in floats
    static constexpr int LDS_FLOATS = FLOATS + (64 / WS) * (WS + 1) * 4;      // + the x-weight tables
    static_assert(!ON || RPG <= PITCH4, "patch row pitch");
    __device__ static __forceinline__ void chunk(int ci, int& j, int& part) {
        if constexpr (RPG == 2) {
            j = ci >> 1;
            part = ci & 1;
        } else if constexpr (RPG == 3) {
            j = (ci * 171) >> 9;                              // ci / 3 for ci < 171
            part = ci - 3 * j;
        } else {
            static_assert(RPG == 5, "chunks per row");
            j = (ci * 205) >> 10;                             // ci / 5 for ci < 404
            part = ci - 5 * j;
        }
    }
};
// loop-invariant part of a lane's chunk addresses: row * W + 16 * part (0 for the idle slots of the last round)
template <int WS>
struct CoopOff {
    int o[CoopGeo<WS>::ON ? CoopGeo<WS>::NCH : 1];
    __device__ __forceinline__ void init(int r, int W) {
        if constexpr (CoopGeo<WS>::ON) {
#pragma unroll
            for (int k = 0; k < CoopGeo<WS>::NCH; ++k) {
                int j, part;
                CoopGeo<WS>::chunk(r + WS * k, j, part);
                o[k] = r + WS * k < CoopGeo<WS>::T ? j * W + 16 * part : 0;
            }
        }
    }
};
template <int WS>
struct RawRows<WS, MODE_CWS> {
    static constexpr int NB = WS / 4 + 1;
    // WS+1 bytes per row, or the lane's share of the two patches
    uint32_t a0[CoopGeo<WS>::ON ? 1 : NB], a1[CoopGeo<WS>::ON ? 1 : NB], b0[CoopGeo<WS>::ON ? 1 : NB], b1[CoopGeo<WS>::ON ? 1 : NB];
    uint32_t ca[CoopGeo<WS>::ON ? CoopGeo<WS>::NCH : 1][4], cb[CoopGeo<WS>::ON ? CoopGeo<WS>::NCH : 1][4];
    int reg;
    int cls;             // see above (rows a0, a1, b0, b1 in bits 0-1, 2-3, 4-5, 6-7)
    int fix;
};

// per-lane row geometry of the bilinear (CWS) shift, exactly as PIVbackend.py:162-172 computes it
struct CwsRow {
    float wya_up, wya_dn, wyb_up, wyb_dn;
    int dya, uya, dyb, uyb;
    int ydeg_a, ydeg_b;
};
__device__ __forceinline__ CwsRow cws_row(int gy, float vy) {
    CwsRow c;
    const float gyf = (float)gy;
    const float nya = gyf - vy, nyb = gyf + vy;          // frame a by -(vx, vy), frame b by +(vx, vy)
    const float uya_f = ceilf(nya), dya_f = floorf(nya);
    const float uyb_f = ceilf(nyb), dyb_f = floorf(nyb);
    c.uya = f2i_sat_t(uya_f);
    c.dya = f2i_sat_t(dya_f);
    c.uyb = f2i_sat_t(uyb_f);
    c.dyb = f2i_sat_t(dyb_f);
    c.wya_up = uya_f - nya;
    c.wya_dn = nya - dya_f;
    c.wyb_up = uyb_f - nyb;
    c.wyb_dn = nyb - dyb_f;
    c.ydeg_a = c.uya == c.dya;
    c.ydeg_b = c.uyb == c.dyb;
    return c;
}

// Row classes under the flat-index clamp (B:177-180, B:214): a run of `used` pixels starting at
// flat index q that lies entirely before pixel 0 reads f[0] everywhere, entirely behind the last
// pixel reads f[HW-1] everywhere; such rows (top / bottom border windows) stay on the wide-load
// path: they load a safe address and the loaded dwords are overwritten with the broadcast pixel.
// Class 3 (the run straddles either end of the frame) needs the per-pixel path.
__device__ __forceinline__ int classify_row(long long q, int used, int loaded, int HW, long long& qload) {
    const long long lim = (long long)HW - loaded;
    if (q >= 0 && q <= lim) {
        qload = q;
        return 0;
    }
    if (q + used - 1 <= 0) {
        qload = 0;
        return 1;
    }
    if (q >= (long long)HW - 1) {
        qload = lim;
        return 2;
    }
    qload = 0;
    return 3;
}

__device__ __forceinline__ int clamp_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The same classification in 32-bit arithmetic (callers clamp the row index first, see issue_rows):
// a 64-bit compare or select is two to four VALU instructions, and this runs four times per item.
__device__ __forceinline__ int classify_row32(int q, int used, int loaded, int HW, int& qload) {
    const int lim = HW - loaded;
    const bool inside = (unsigned)q <= (unsigned)lim;          // 0 <= q <= lim
    const bool before = q + used - 1 <= 0;
    const bool behind = q >= HW - 1;
    qload = inside ? q : (behind && !before ? lim : 0);
    return inside ? 0 : (before ? 1 : (behind ? 2 : 3));
}

template <int N>
__device__ __forceinline__ void fix_row(uint32_t (&d)[N], int cls) {
    const uint32_t px = (cls == 1 ? (d[0] & 0xffu) : (d[N - 1] >> 24)) * 0x01010101u;
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = cls != 0 ? px : d[i];
}
// (patch rows: the last loaded byte of a row lies beyond the N dwords a lane holds; `last` = that dword)
template <int N>
__device__ __forceinline__ void fix_row(uint32_t (&d)[N], int cls, uint32_t last) {
    const uint32_t px = (cls == 1 ? (d[0] & 0xffu) : (last >> 24)) * 0x01010101u;
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = cls != 0 ? px : d[i];
}

// The loads are issued UNCONDITIONALLY: a load under `if` makes the loaded registers the target of
// a PHI copy, and the compiler then waits for the data right behind the load -- which would
// cancel the prefetch.  Rows that need the slow path (frame border / rounding corner cases) load
// from the un-shifted row instead (always inside the frame) and ignore the data.
template <int WS, int MODE, bool FAST = false>
__device__ __forceinline__ void issue_rows(const PassParams& p, const ItemGeom& g, int r, float vx, float vy,
                                           RawRows<WS, MODE>& raw, const CoopOff<WS>& coff) {
    const int HW = p.H * p.W;
    const uint8_t* __restrict__ fa = p.A + (size_t)g.pair * HW;
    const uint8_t* __restrict__ fb = p.B + (size_t)g.pair * HW;
    const long long base = (long long)(g.y0 + r) * p.W + g.x0;
    if constexpr (MODE == MODE_PASS1) {
        if constexpr (Coop1<WS>::ON) {
            // rows LOADED by the window's lanes together (round 5): with lane = row every load instruction touches WS cache
            // lines and the CU's address unit takes them one by one -- the stamps put 18 % of the 64x64 iteration on the issue
            // of these eight loads.  Chunk ci = row * P16 + part of 16 bytes, lane r takes chunks r, r + WS, ...: consecutive
            // lanes read consecutive pieces of a row (16 rows per instruction for 64-pixel rows); convert_rows hands the
            // chunks over through the (idle) transposition tile and every lane gets its row back.
            using C1 = Coop1<WS>;
            const unsigned o0 = (unsigned)(g.y0 * p.W + g.x0);
#pragma unroll
            for (int k = 0; k < C1::P16; ++k) {
                const int ci = r + WS * k, j = ci / C1::P16, part = ci % C1::P16;
                const unsigned o_ = o0 + (unsigned)(j * p.W + 16 * part);
                load_dwords<4>(fa + o_, *reinterpret_cast<uint32_t(*)[4]>(&raw.a[4 * k]));
                load_dwords<4>(fb + o_, *reinterpret_cast<uint32_t(*)[4]>(&raw.b[4 * k]));
            }
        } else {
#ifdef TPIV_EXP_P1LOADS      // timing experiment only (wrong results): TPIV_EXP_P1LOADS 16-byte loads per row and frame
            load_dwords<4 * TPIV_EXP_P1LOADS>(fa + base, *reinterpret_cast<uint32_t(*)[4 * TPIV_EXP_P1LOADS]>(&raw.a[0]));
            load_dwords<4 * TPIV_EXP_P1LOADS>(fb + base, *reinterpret_cast<uint32_t(*)[4 * TPIV_EXP_P1LOADS]>(&raw.b[0]));
#pragma unroll
            for (int k = 4 * TPIV_EXP_P1LOADS; k < WS / 4; ++k) {
                raw.a[k] = 0x11213141u * (unsigned)(r + 1 + k);
                raw.b[k] = 0x31112141u * (unsigned)(r + 3 + k);
            }
#else
            load_dwords<WS / 4>(fa + base, raw.a);
            load_dwords<WS / 4>(fb + base, raw.b);
#endif
        }
    } else if constexpr (MODE == MODE_DWS) {
        // integer shift on the FLAT index (B:213-215): a at idx - (vy*W + vx), b at idx + (...)
        const long long sh = (long long)vy * p.W + (long long)vx;
        raw.qa = base - sh;
        raw.qb = base + sh;
        long long la, lb;
        const int ca = classify_row(raw.qa, WS, WS, HW, la), cb = classify_row(raw.qb, WS, WS, HW, lb);
        raw.reg = __all(ca != 3 && cb != 3) ? 1 : 0;
        raw.cls = ca | (cb << 2);
        raw.fix = (raw.reg && __any(raw.cls != 0)) ? 1 : 0;
        load_dwords<WS / 4>(fa + (raw.reg ? la : base), raw.a);
        load_dwords<WS / 4>(fb + (raw.reg ? lb : base), raw.b);
    } else {
        constexpr int NB = WS / 4 + 1;
        const CwsRow c = cws_row(g.y0 + r, vy);
        // Fast path: floor(float(gx) + vx) == gx + floor(vx) for every column, no column exactly integral
        // (the "nearest sample" quirk, B:170/193), and all four source rows inside the frame, so that a row is
        // WS+1 consecutive bytes.  The column condition is checked exactly, with the reference's own float32
        // sums: lane r tests column r of its window (the WS lanes of a window cover its WS columns) -- a
        // threshold on frac(vx) wide enough for float32 rounding at W = 4096 sent 0.4 % of the windows, i.e.
        // 1.6 % of the 16x16 wavefronts, down the 4.6x slower per-pixel path.
        const float fvx = floorf(vx);
        // 32-bit flat indices: |floor(vx)| <= W on the fast path, and a source row further than 8
        // rows outside the frame classifies exactly like row -8 / H+7 (the whole run stays before
        // pixel 0 / behind the last pixel), so the clamped values give the same class and address.
        const int W_ = p.W;
        const int ivx = clamp_i(f2i_sat_t(fvx), -W_, W_);
        const int rlo = -8, rhi = p.H + 7;
        const int dya = clamp_i(c.dya, rlo, rhi), uya = clamp_i(c.uya, rlo, rhi);
        const int dyb = clamp_i(c.dyb, rlo, rhi), uyb = clamp_i(c.uyb, rlo, rhi);
        // frame a uses -vx: floor(-vx) = -floor(vx) - 1 when frac != 0
        const int xa = g.x0 - ivx - 1, xb = g.x0 + ivx;
        const int qa0 = dya * W_ + xa, qa1 = uya * W_ + xa;
        const int qb0 = dyb * W_ + xb, qb1 = uyb * W_ + xb;
        using CG = CoopGeo<WS>;
        constexpr int LOADED = CG::ON ? CG::LOADED : 4 * NB;       // bytes a row load reads
        // (FAST: an integral row coordinate -- B:170, B:193 return the nearest sample -- is left to the
        //  per-pixel path, which implements the quirk; the lerp form of convert_rows does not)
        const float gxr = (float)(g.x0 + r);
        const float nxa_r = gxr - vx, nxb_r = gxr + vx;
        const float fxa_r = floorf(nxa_r), fxb_r = floorf(nxb_r);
        const bool col_ok = fxa_r == (float)(g.x0 + r - ivx - 1) && fxb_r == (float)(g.x0 + r + ivx) &&
                            fxa_r != nxa_r && fxb_r != nxb_r;
        bool reg = col_ok && fabsf(vx) < (float)p.W && !(FAST && (c.ydeg_a | c.ydeg_b));
        int la0 = 0, la1 = 0, lb0 = 0, lb1 = 0, cls = 0;
        auto classify_own = [&]() TPIV_LAMBDA_INLINE {       // the lane's four rows under the flat-index clamp
            const int c0 = classify_row32(qa0, WS + 1, LOADED, HW, la0), c1 = classify_row32(qa1, WS + 1, LOADED, HW, la1);
            const int c2 = classify_row32(qb0, WS + 1, LOADED, HW, lb0), c3 = classify_row32(qb1, WS + 1, LOADED, HW, lb1);
            reg = reg && c0 != 3 && c1 != 3 && c2 != 3 && c3 != 3;
            cls = c0 | (c1 << 2) | (c2 << 4) | (c3 << 6);
        };
        if constexpr (CG::ON) {
            // the window's patches start at the (clamped) lower source row of its row 0; a lane's two rows must
            // lie inside them (they do unless float32 rounding makes the row coordinates jump)
            const float gy0f = (float)g.y0;
            const int base_a = clamp_i(f2i_sat_t(floorf(gy0f - vy)), rlo, rhi);
            const int base_b = clamp_i(f2i_sat_t(floorf(gy0f + vy)), rlo, rhi);
            reg = reg && (unsigned)(dya - base_a) <= (unsigned)WS && (unsigned)(uya - base_a) <= (unsigned)WS &&
                  (unsigned)(dyb - base_b) <= (unsigned)WS && (unsigned)(uyb - base_b) <= (unsigned)WS;
            // Wavefronts whose patches lie inside the frame with LOADED bytes to spare in every row (all but the
            // border windows) add the loop-invariant chunk offsets to the patch origin -- and every row a lane
            // may use is an ordinary row (class 0); the others classify every chunk's row like a lane's own rows
            // (the first version did that always: +160 VALU instructions per item).  Only ADDRESSES differ
            // between the branches: the loads stay unconditional.
            const bool inside = base_a >= 0 && base_a + WS < p.H && base_b >= 0 && base_b + WS < p.H &&
                                xa >= 0 && xa + LOADED <= W_ && xb >= 0 && xb + LOADED <= W_;
            const bool all_in = __all(inside);
            if (!all_in) classify_own();
            raw.reg = __all(reg) ? 1 : 0;
            raw.cls = cls;
            raw.fix = (raw.reg && __any(raw.cls != 0)) ? 1 : 0;
            unsigned oa[CG::NCH], ob[CG::NCH];
            if (all_in) {
                const int qa_o = base_a * W_ + xa, qb_o = base_b * W_ + xb;
#pragma unroll
                for (int k = 0; k < CG::NCH; ++k) {
                    oa[k] = (unsigned)(qa_o + coff.o[k]);
                    ob[k] = (unsigned)(qb_o + coff.o[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < CG::NCH; ++k) {
                    const int ci = r + WS * k;
                    int j, part;
                    CG::chunk(ci, j, part);
                    int qa_l, qb_l;
                    const int ra = clamp_i(base_a + j, rlo, rhi), rb = clamp_i(base_b + j, rlo, rhi);
                    classify_row32(ra * W_ + xa, WS + 1, LOADED, HW, qa_l);
                    classify_row32(rb * W_ + xb, WS + 1, LOADED, HW, qb_l);
                    const bool live = ci < CG::T;                  // (the last chunk round is partly idle)
                    oa[k] = live ? (unsigned)(qa_l + 16 * part) : 0u;
                    ob[k] = live ? (unsigned)(qb_l + 16 * part) : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < CG::NCH; ++k) {
                load_dwords<4>(fa + oa[k], raw.ca[k]);
                load_dwords<4>(fb + ob[k], raw.cb[k]);
            }
        } else {
        classify_own();
        raw.reg = __all(reg) ? 1 : 0;
        raw.cls = cls;
        raw.fix = (raw.reg && __any(raw.cls != 0)) ? 1 : 0;
        const int lim = HW - 4 * NB;
        const int base32 = (g.y0 + r) * W_ + g.x0;
        const int safe = base32 < lim ? base32 : lim;
        // (unsigned 32-bit offsets from the wave-uniform frame pointers: scalar base + VGPR offset loads)
        load_dwords<NB>(fa + (unsigned)(raw.reg ? la0 : safe), raw.a0);
        load_dwords<NB>(fa + (unsigned)(raw.reg ? la1 : safe), raw.a1);
        load_dwords<NB>(fb + (unsigned)(raw.reg ? lb0 : safe), raw.b0);
        load_dwords<NB>(fb + (unsigned)(raw.reg ? lb1 : safe), raw.b1);
        }
    }
}

// first dword of a 16-byte LDS unit, read AS a ds_read_b128 (the empty asm keeps all four components "used": left
// alone, the compiler narrows the load to a ds_read_b32, which is bank-conflicted at the patch row pitch)
__device__ __forceinline__ uint32_t read_unit_x(const uint4* src) {
    uint4 t = *src;
    asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
    return t.x;
}

// raw rows -> float samples x[k] = (a, b) and the lane's partial sums
// RBH: the per-pixel slow paths park a row in LDS in RBH pieces (2 when the planar 64x64 layout
// leaves only 64 x 33 floats)
// FAST (precision "fast"): the CWS sample is formed as a lerp between the two source rows followed by a
// lerp between neighbouring columns, with the SAME float32 weights (4 instead of 11 multiply/add
// instructions per sample; differs from the reference's evaluation order by float32 rounding only), and the per-sample window sums are not formed here (the mean is
// removed in the DC bin after the row transform, see the kernel).  !FAST keeps the reference's
// operation order: staged windows bit-identical to biliniar_interpolation_CWS (B:187-193).
// SQ (locating pass of precision "exact"): also the lane's sums of squares of the raw bytes (exact integers), from which
// the kernel forms E+, the scale of its decision band (piv_kernels.h, "The band")
struct RowSquares {
    unsigned aa, bb;
};
// PATH: 0 = the staging path of the item is a run-time (wave-uniform) choice, 1 = the wide-load path only (the per-pixel
// path is not compiled: the caller has sent those items elsewhere)
template <int WS, int MODE, int RBH = 1, bool FAST = false, bool SQ = false, int PATH = 0>
__device__ __forceinline__ void convert_rows(const PassParams& p, const ItemGeom& g, int r, int lane, float vx,
                                             float vy, RawRows<WS, MODE>& raw, cf (&x)[WS], float& sa,
                                             float& sb, float* lds, RowSquares* sq = nullptr) {
    const int HW = p.H * p.W;
    const uint8_t* __restrict__ fa = p.A + (size_t)g.pair * HW;
    const uint8_t* __restrict__ fb = p.B + (size_t)g.pair * HW;
    static constexpr int RBL = WS / RBH;       // row-buffer length
    float* rowbuf = lds + lane * (RBL + 1);    // slow paths only
    // The rare per-pixel paths must not set the register budget of the small tiles (unrolled, their
    // gathers keep a 64-bit address per load in flight: 135 VGPRs for an 8x8 kernel whose fast path
    // needs about 60), so they are fully rolled there.
    static constexpr int UNR_DWS = WS <= 16 ? 1 : 4, UNR_CWS = WS <= 16 ? 1 : 2;
    if constexpr (MODE == MODE_PASS1) {
        if constexpr (Coop1<WS>::ON) {
            // chunks -> tile -> the lane's own row, one frame at a time (the tile is idle until the first transposition)
            using C1 = Coop1<WS>;
            uint4* const t4 = reinterpret_cast<uint4*>(lds) + (lane / WS) * (WS * C1::PITCH4);
            auto hand_over = [&](uint32_t (&d)[WS / 4]) TPIV_LAMBDA_INLINE {
                wave_sync();
#pragma unroll
                for (int k = 0; k < C1::P16; ++k) {
                    const int ci = r + WS * k, j = ci / C1::P16, part = ci % C1::P16;
                    t4[j * C1::PITCH4 + part] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
                }
                wave_sync();
#pragma unroll
                for (int k = 0; k < C1::P16; ++k) {
                    const uint4 v = t4[r * C1::PITCH4 + k];
                    d[4 * k] = v.x, d[4 * k + 1] = v.y, d[4 * k + 2] = v.z, d[4 * k + 3] = v.w;
                }
            };
            hand_over(raw.a);
            hand_over(raw.b);
            wave_sync();
        }
        unsigned ia = 0, ib = 0;
#pragma unroll
        for (int q = 0; q < WS / 4; ++q) {
            ia = __builtin_amdgcn_sad_u8(raw.a[q], 0u, ia);
            ib = __builtin_amdgcn_sad_u8(raw.b[q], 0u, ib);
        }
        sa = (float)ia;
        sb = (float)ib;
        if constexpr (SQ) {
            unsigned iaa = 0, ibb = 0;
#pragma unroll
            for (int q = 0; q < WS / 4; ++q) {
                iaa = __builtin_amdgcn_udot4(raw.a[q], raw.a[q], iaa, false);
                ibb = __builtin_amdgcn_udot4(raw.b[q], raw.b[q], ibb, false);
            }
            sq->aa = iaa;
            sq->bb = ibb;
        }
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int k = decltype(kc)::value;
            x[k].x = byte_f<k, WS / 4>(raw.a);
            x[k].y = byte_f<k, WS / 4>(raw.b);
        });
    } else if constexpr (MODE == MODE_DWS) {
        if (raw.reg) {
            if (raw.fix) {            // border windows: rows entirely outside the frame
                fix_row(raw.a, raw.cls & 3);
                fix_row(raw.b, (raw.cls >> 2) & 3);
            }
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                x[k].x = byte_f<k, WS / 4>(raw.a);
                x[k].y = byte_f<k, WS / 4>(raw.b);
            });
        } else {          // a row touches the first/last pixel of the frame: per-pixel clamp.
            // Rare: kept as a rolled loop that parks the row in LDS (no code blow-up).
            wave_sync();
            static_for<0, RBH>([&](auto hc) TPIV_LAMBDA_INLINE {
                constexpr int k0 = decltype(hc)::value * RBL;
#pragma unroll UNR_DWS
                for (int k = 0; k < RBL; ++k) rowbuf[k] = fetch_clamped_t(fa, raw.qa + k0 + k, HW);
#pragma unroll
                for (int k = 0; k < RBL; ++k) x[k0 + k].x = rowbuf[k];
#pragma unroll UNR_DWS
                for (int k = 0; k < RBL; ++k) rowbuf[k] = fetch_clamped_t(fb, raw.qb + k0 + k, HW);
#pragma unroll
                for (int k = 0; k < RBL; ++k) x[k0 + k].y = rowbuf[k];
            });
            wave_sync();
        }
        sa = 0.f;
        sb = 0.f;
        if constexpr (!FAST) {
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                sa += x[k].x;
                sb += x[k].y;
            }
        }
    } else {
        constexpr int NB = WS / 4 + 1;
        const CwsRow c = cws_row(g.y0 + r, vy);
        const float gx0f = (float)g.x0;
        using CG = CoopGeo<WS>;
        if (PATH == 1 || raw.reg) {
            uint32_t ra0[NB], ra1[NB], rb0[NB], rb1[NB];
            const uint4 *lz0 = nullptr, *lz1 = nullptr, *lz2 = nullptr, *lz3 = nullptr;
            if constexpr (CG::ON) {
                // the chunks go to LDS, the lane's rows come back (same wavefront: program order is enough)
                uint4* pl = reinterpret_cast<uint4*>(lds);
                const int w_ = lane / WS;
                uint4* pa = pl + w_ * CG::BLOCK4;
                uint4* pb = pa + CG::PATCH4;
                wave_sync();
#pragma unroll
                for (int k = 0; k < CG::NCH; ++k) {
                    const int ci = r + WS * k;
                    int j, part;
                    CG::chunk(ci, j, part);
                    if (ci < CG::T) {
                        pa[j * CG::PITCH4 + part] = make_uint4(raw.ca[k][0], raw.ca[k][1], raw.ca[k][2], raw.ca[k][3]);
                        pb[j * CG::PITCH4 + part] = make_uint4(raw.cb[k][0], raw.cb[k][1], raw.cb[k][2], raw.cb[k][3]);
                    }
                }
                wave_sync();
                const int rlo = -8, rhi = p.H + 7;
                const float gy0f = (float)g.y0;
                const int base_a = clamp_i(f2i_sat_t(floorf(gy0f - vy)), rlo, rhi);
                const int base_b = clamp_i(f2i_sat_t(floorf(gy0f + vy)), rlo, rhi);
                const uint4* s0 = pa + (clamp_i(c.dya, rlo, rhi) - base_a) * CG::PITCH4;
                const uint4* s1 = pa + (clamp_i(c.uya, rlo, rhi) - base_a) * CG::PITCH4;
                const uint4* s2 = pb + (clamp_i(c.dyb, rlo, rhi) - base_b) * CG::PITCH4;
                const uint4* s3 = pb + (clamp_i(c.uyb, rlo, rhi) - base_b) * CG::PITCH4;
                auto fetch = [&](const uint4* src, uint32_t (&d)[NB]) TPIV_LAMBDA_INLINE {
#pragma unroll
                    for (int q = 0; q < NB / 4; ++q) {
                        const uint4 t = src[q];
                        d[4 * q] = t.x, d[4 * q + 1] = t.y, d[4 * q + 2] = t.z, d[4 * q + 3] = t.w;
                    }
                    // NB = 4 m + 1: the last dword comes with one more 16-byte read (the row holds RPG = m + 1 units) -- a
                    // ds_read_b32 at a row pitch of 12 dwords is 4-way bank-conflicted
                    static_assert(CG::RPG * 4 >= NB, "patch rows hold the whole 16-byte unit of the last dword");
                    d[NB - 1] = read_unit_x(src + NB / 4);
                };
                if (raw.fix) {
                    // border windows: a source row entirely outside the frame reads the first / last pixel of the
                    // frame everywhere (flat-index clamp).  Its chunks were loaded from the start / the end of the
                    // frame; the lanes that use such a row overwrite it with that pixel (several lanes may write
                    // the same row: same bytes, and the pixel survives the overwrite)
                    constexpr int LAST = CG::LOADED / 4 - 1;       // dword that holds the last loaded byte
                    auto fix_lds = [&](const uint4* src, int cls) TPIV_LAMBDA_INLINE {
                        if (cls != 0) {
                            const uint32_t* sw = reinterpret_cast<const uint32_t*>(src);
                            const uint32_t px = (cls == 1 ? (sw[0] & 0xffu) : (sw[LAST] >> 24)) * 0x01010101u;
                            uint4* dst = const_cast<uint4*>(src);
#pragma unroll
                            for (int q = 0; q < CG::RPG; ++q) dst[q] = make_uint4(px, px, px, px);
                        }
                    };
                    fix_lds(s0, raw.cls & 3);
                    fix_lds(s1, (raw.cls >> 2) & 3);
                    fix_lds(s2, (raw.cls >> 4) & 3);
                    fix_lds(s3, (raw.cls >> 6) & 3);
                    wave_sync();
                }
                // (64x64, fast form: the rows come out of LDS 16 pixels at a time inside the sampling loop --
                //  four whole rows are 68 registers next to the 128 of the samples)
                constexpr bool LAZY = FAST && WS == 64;
                if constexpr (LAZY) {
                    lz0 = s0, lz1 = s1, lz2 = s2, lz3 = s3;
                } else {
                    fetch(s0, ra0);
                    fetch(s1, ra1);
                    fetch(s2, rb0);
                    fetch(s3, rb1);
                }
            } else {
#pragma unroll
                for (int q = 0; q < NB; ++q) ra0[q] = raw.a0[q], ra1[q] = raw.a1[q], rb0[q] = raw.b0[q], rb1[q] = raw.b1[q];
                if (raw.fix) {            // border windows: rows entirely outside the frame
                    fix_row(ra0, raw.cls & 3);
                    fix_row(ra1, (raw.cls >> 2) & 3);
                    fix_row(rb0, (raw.cls >> 4) & 3);
                    fix_row(rb1, (raw.cls >> 6) & 3);
                }
            }
            // The x-direction weights depend on the column only (same for every row of the
            // window): lane r evaluates them for column r exactly as B:164-171 does and parks
            // them in LDS; every lane then reads the WS columns back (broadcast reads).
            // In the fast path frac(vx) is away from 0/1, so ceil = floor + 1 and the column is
            // never "integral" (only the row can be, ydeg_*).
            // (one spare float4 per window: without it the 64 / WS windows' tables start a multiple of 256 bytes
            //  apart and their broadcast reads hit the same banks -- a 4-way conflict on every read for 16x16)
            float4* wbuf = reinterpret_cast<float4*>(lds + (CG::ON ? CG::FLOATS : 0)) + (lane / WS) * (WS + 1);
            {
                const float gxf = gx0f + (float)r;               // exact: small integers
                const float nxa = gxf - vx, nxb = gxf + vx;
                const float uxa_f = ceilf(nxa), dxa_f = floorf(nxa);
                const float uxb_f = ceilf(nxb), dxb_f = floorf(nxb);
                wave_sync();
                wbuf[r] = make_float4(uxa_f - nxa, nxa - dxa_f, uxb_f - nxb, nxb - dxb_f);
                wave_sync();
            }
            // (FAST: a wavefront that holds a lane with an integral row coordinate -- the "nearest sample"
            //  quirk of B:170, B:193 -- never gets here: issue_rows sends it down the per-pixel path)
            if constexpr (FAST) {
                // rows first, then columns: the WS + 1 column values of the row lerp are each used by two
                // output samples (130 instead of 192 multiply/add instructions per frame and lane)
                constexpr bool LAZY = CG::ON && WS == 64;
                auto pull = [&](auto qc) TPIV_LAMBDA_INLINE {      // 16-byte unit q of the four rows (or the last dword)
                    constexpr int q = decltype(qc)::value;
                    auto one = [&](const uint4* src, uint32_t (&d)[NB]) TPIV_LAMBDA_INLINE {
                        if constexpr (4 * q + 3 < NB) {
                            const uint4 t = src[q];
                            d[4 * q] = t.x, d[4 * q + 1] = t.y, d[4 * q + 2] = t.z, d[4 * q + 3] = t.w;
                        } else {
                            d[NB - 1] = read_unit_x(src + NB / 4);
                        }
                    };
                    one(lz0, ra0), one(lz1, ra1), one(lz2, rb0), one(lz3, rb1);
                };
                if constexpr (LAZY) {
                    pull(std::integral_constant<int, 0>{});
                    pull(std::integral_constant<int, 1>{});
                }
                // (TPIV_SDWA_LERP: the upper row's byte enters its product as a float32 denormal -- see byte_d -- and every sample
                //  of the wavefront carries the factor 2^-24 from here on: folded into end_scale by the kernel)
                constexpr bool SD = TPIV_SDWA_LERP != 0;
                const float wau = SD ? c.wya_up * SDWA_UP : c.wya_up, wad = SD ? c.wya_dn * SDWA_DN : c.wya_dn;
                const float wbu = SD ? c.wyb_up * SDWA_UP : c.wyb_up, wbd = SD ? c.wyb_dn * SDWA_DN : c.wyb_dn;
                auto row_lerp = [&](auto kc, const uint32_t (&r0_)[NB], const uint32_t (&r1_)[NB], float wu_, float wd_) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (SD) return fmaf(byte_f<k, NB>(r1_), wd_, byte_d<k, NB>(r0_) * wu_);
                    else return fmaf(byte_f<k, NB>(r1_), wd_, byte_f<k, NB>(r0_) * wu_);
                };
                float va = row_lerp(std::integral_constant<int, 0>{}, ra0, ra1, wau, wad);
                float vb = row_lerp(std::integral_constant<int, 0>{}, rb0, rb1, wbu, wbd);
                static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k % 8 == 0 && k > 0) __builtin_amdgcn_sched_barrier(0);
                    // dwords 4b .. 4b + 4 serve pixels 16b .. 16b + 15 (+1): unit b + 1 arrives at pixel 16b
                    if constexpr (LAZY && k % 16 == 0 && k > 0) pull(std::integral_constant<int, k / 16 + 1>{});
                    const float4 wx = wbuf[k];
                    const float na = row_lerp(std::integral_constant<int, k + 1>{}, ra0, ra1, wau, wad);
                    const float nb = row_lerp(std::integral_constant<int, k + 1>{}, rb0, rb1, wbu, wbd);
#ifdef TPIV_MUTANT_LERP
                    // tests only (tools/diag/libtorchpiv_hip_mutant.so, never shipped): one weight of the column
                    // lerp off by 1e-3 -- the parity gates must notice (tests/test_gpu_gates.py)
                    x[k].x = fmaf(na, wx.y * 1.001f, va * wx.x);
#else
                    x[k].x = fmaf(na, wx.y, va * wx.x);
#endif
                    x[k].y = fmaf(nb, wx.w, vb * wx.z);
                    va = na;
                    vb = nb;
                });
            } else {
            unsigned dmask_a = c.ydeg_a ? ~0u : 0u, dmask_b = c.ydeg_b ? ~0u : 0u;
            asm volatile("" : "+v"(dmask_a), "+v"(dmask_b));       // keep them masks (not re-derived selects)
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int k = decltype(kc)::value;
                // keep the scheduler from hoisting all WS weight reads (4 VGPRs each) to the top
                if constexpr (k % 8 == 0 && k > 0) __builtin_amdgcn_sched_barrier(0);
                const float4 wx = wbuf[k];
                x[k].x = bilerp_ref_m(byte_f<k, NB>(ra0), byte_f<k + 1, NB>(ra0), byte_f<k, NB>(ra1),
                                      byte_f<k + 1, NB>(ra1), wx.x, wx.y, c.wya_up, c.wya_dn, dmask_a);
                x[k].y = bilerp_ref_m(byte_f<k, NB>(rb0), byte_f<k + 1, NB>(rb0), byte_f<k, NB>(rb1),
                                      byte_f<k + 1, NB>(rb1), wx.z, wx.w, c.wyb_up, c.wyb_dn, dmask_b);
            });
            }
            wave_sync();
        } else {          // generic per-pixel gather with the flat-index clamp (rare: rolled loops
                          // that park the row in LDS)
            wave_sync();
            static_for<0, RBH>([&](auto hc) TPIV_LAMBDA_INLINE {
            constexpr int k0 = decltype(hc)::value * RBL;
#pragma unroll UNR_CWS
            for (int k = 0; k < RBL; ++k) {
                const float nxa = (gx0f + (float)(k0 + k)) - vx;
                const float uxa_f = ceilf(nxa), dxa_f = floorf(nxa);
                const int uxa = f2i_sat_t(uxa_f), dxa = f2i_sat_t(dxa_f);
                rowbuf[k] = bilerp_ref(fetch_clamped_t(fa, (long long)c.dya * p.W + dxa, HW),
                                       fetch_clamped_t(fa, (long long)c.dya * p.W + uxa, HW),
                                       fetch_clamped_t(fa, (long long)c.uya * p.W + dxa, HW),
                                       fetch_clamped_t(fa, (long long)c.uya * p.W + uxa, HW), uxa_f - nxa,
                                       nxa - dxa_f, c.wya_up, c.wya_dn, c.ydeg_a || (uxa == dxa));
            }
#pragma unroll
            for (int k = 0; k < RBL; ++k) x[k0 + k].x = (FAST && TPIV_SDWA_LERP) ? rowbuf[k] * SDWA_DN : rowbuf[k];
#pragma unroll UNR_CWS
            for (int k = 0; k < RBL; ++k) {
                const float nxb = (gx0f + (float)(k0 + k)) + vx;
                const float uxb_f = ceilf(nxb), dxb_f = floorf(nxb);
                const int uxb = f2i_sat_t(uxb_f), dxb = f2i_sat_t(dxb_f);
                rowbuf[k] = bilerp_ref(fetch_clamped_t(fb, (long long)c.dyb * p.W + dxb, HW),
                                       fetch_clamped_t(fb, (long long)c.dyb * p.W + uxb, HW),
                                       fetch_clamped_t(fb, (long long)c.uyb * p.W + dxb, HW),
                                       fetch_clamped_t(fb, (long long)c.uyb * p.W + uxb, HW), uxb_f - nxb,
                                       nxb - dxb_f, c.wyb_up, c.wyb_dn, c.ydeg_b || (uxb == dxb));
            }
#pragma unroll
            for (int k = 0; k < RBL; ++k) x[k0 + k].y = (FAST && TPIV_SDWA_LERP) ? rowbuf[k] * SDWA_DN : rowbuf[k];
            });
            wave_sync();
        }
        sa = 0.f;
        sb = 0.f;
        if constexpr (!FAST) {
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                sa += x[k].x;
                sb += x[k].y;
            }
        }
    }
}

// columns of map row ys (fftshift layout) that B:346-358 zeroes around the first peak m: q = clamp(m + i + WS j),
// |i|, |j| <= wv -- in row y' the columns mx+i (j = y'-my), mx+i+WS (j = y'-my+1), mx+i-WS (j = y'-my-1), plus the clamps
template <int WS>
__device__ __forceinline__ unsigned long long exclusion_row_mask(int m, int ys, int wv) {
    static_assert(WS <= 64, "one 64-bit mask per row");
    const int my_ = m / WS, mx_ = m % WS, KD = WS * WS;
    const int dj = ys - my_;
    unsigned long long ex = 0ull;
    auto span = [&](int lo_, int hi_) TPIV_LAMBDA_INLINE {
        lo_ = lo_ < 0 ? 0 : lo_;
        hi_ = hi_ > WS - 1 ? WS - 1 : hi_;
        if (lo_ > hi_) return 0ull;
        const unsigned long long ones = (hi_ - lo_ + 1) >= 64 ? ~0ull : ((1ull << (hi_ - lo_ + 1)) - 1ull);
        return ones << lo_;
    };
    if (dj >= -wv && dj <= wv) ex |= span(mx_ - wv, mx_ + wv);
    if (dj + 1 >= -wv && dj + 1 <= wv) ex |= span(mx_ - wv + WS, mx_ + wv + WS);
    if (dj - 1 >= -wv && dj - 1 <= wv) ex |= span(mx_ - wv - WS, mx_ + wv - WS);
    if (ys == 0 && (m - wv - wv * WS) <= 0) ex |= 1ull;
    if (ys == WS - 1 && (m + wv + wv * WS) >= KD - 1) ex |= 1ull << (WS - 1);
    return ex;
}

// maxima that pass over quiet NaNs (IEEE mode, the default of compute kernels: the maximum of a number and a quiet NaN is the
// number; all NaN in, NaN out).  Inline assembly: for fmaxf() of a value of unknown origin the compiler adds a canonicalising
// v_max_f32 x, x per operand.
__device__ __forceinline__ float max3_skip_nan(float a, float b, float c) {
    float o;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
__device__ __forceinline__ float max_skip_nan(float a, float b) {
    float o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}

// ---- peak analysis of one correlation row per lane (B:346-358, B:381-392, B:518) -----------------
// in: lane (w, r) holds row y = r of its window's circular correlation, value at column x in row[x]
// (un-shifted coordinates; the callers' copies into `row` are register renames).  Writes the window's 8-float record for finalize_kernel.
template <int WS, bool PLANAR, bool SCALED = false>
__device__ __forceinline__ void peak_analysis(const PassParams& p, const float (&row)[WS], float* tile, int w, int r,
                                              bool active, bool dead, size_t fidx, float scale = 1.0f) {
    using G = TileGeo<WS, PLANAR>;
    // ---- correlation map in fftshift coordinates: y' = (r + WS/2) % WS, x' = (x + WS/2) % WS
    // SMALLMAP (planar 64x64, 8.4 KB of LDS): only the rows ywin-1..ywin+1 around the peak are parked in
    // LDS (everything the lookups below touch); otherwise the whole map is, written during the scan.
    constexpr bool SMALLMAP = G::WPW * WS * G::MAP_PITCH > G::LDS_FLOATS;
    static_assert(G::WPW * 3 * G::MAP_PITCH <= G::LDS_FLOATS, "map rows must fit the tile LDS");
    float* my_map = tile + w * ((SMALLMAP ? 3 : WS) * G::MAP_PITCH);
    const int ys = (r + WS / 2) % WS;
    // one pass over the raw row gives its minimum AND its maximum: v = (row - min) + eps is monotonic
    // in row, so the row maximum of v is v(max row) -- no second scan for it
    float cmin = 3.4e38f, rraw = -3.4e38f;
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        cmin = fminf(cmin, row[k]);
        rraw = fmaxf(rraw, row[k]);
    }
    cmin = grp_min<WS>(cmin);
    // The scans below avoid per-element compare/select chains (SGPR-pair results and masks, and the
    // registers that tracking an index per lane costs): the arg-max is a plain max per row, and its
    // position is found afterwards with one row of the LDS map spread over the lanes.  (A further
    // variant that also replaced the integer sign-bit scan of the second peak by a lane = column pass
    // over the 2 wv + 1 band rows measured no faster in a same-box A/B: integer ops interleaved with
    // fp32 ops issue for free on MI355X, tools/micro/gen_issue_rate.py.)
    // B:518 corr - min; B:381 corr += eps (float32 arithmetic in passes >= 2).  SCALED: the transforms ran
    // on un-normalised samples and the constant factor of the map (1/n^2, the 1/4 of the packed spectrum,
    // pass 1: 1/(mean a * mean b)) is applied here, inside the same two instructions.
    const float ncs = -(cmin * scale);
    auto shifted = [&](float c_) TPIV_LAMBDA_INLINE {
        if constexpr (SCALED) return __fadd_rn(fmaf(c_, scale, ncs), 1e-7f);
        else return __fadd_rn(__fsub_rn(c_, cmin), 1e-7f);
    };
    // The map goes to LDS and through the second-peak scan as RAW values (fftshift column order): `shifted` is monotonic, so
    // every maximum below is taken over raw values and shifted afterwards, and only the handful of cells that reach the
    // record are shifted at all (round 5: 2 x WS instructions per lane less; same bits -- max(shifted) = shifted(max), and the
    // arg-max compares SHIFTED values, so two raw values that round to the same shifted maximum still tie as they do in the
    // reference's float32 map).
    if constexpr (!SMALLMAP) {
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int xsft = decltype(kc)::value;     // ascending shifted column
            my_map[ys * G::MAP_PITCH + xsft] = row[(xsft + WS / 2) % WS];
        });
    }
    const float rmax = shifted(rraw);                 // = max of the row's shifted values (monotonic); every one is >= 1e-7 > 0
    const float gmax = grp_reduce<WS>(rmax, [](float a, float b) TPIV_LAMBDA_INLINE { return fmaxf(a, b); });
    auto imin = [](int a, int b) TPIV_LAMBDA_INLINE { return a < b ? a : b; };
    // arg-max = FIRST flat index holding the maximum (torch.argmax, B:383): the smallest row y' whose
    // maximum equals it, then the smallest column of that row
    const int ywin = grp_reduce<WS>(rmax == gmax ? ys : WS - 1, imin);      // (WS - 1: NaN maps stay in range)
    const int row0 = SMALLMAP ? ywin - 1 : 0;     // map row held in LDS row 0
    if constexpr (SMALLMAP) {
        wave_sync();
        const int slot = ys - row0;
        if (slot >= 0 && slot <= 2) {
            static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int xsft = decltype(kc)::value;
                my_map[slot * G::MAP_PITCH + xsft] = row[(xsft + WS / 2) % WS];
            });
        }
    }
    wave_sync();                                  // map complete

    if (p.dbg_corr != nullptr && active) {
        int yy = ys;                 // opaque: keeps the WS store addresses out of the item loop's registers
        asm volatile("" : "+v"(yy));
        float* d = p.dbg_corr + fidx * WS * WS + yy * WS;
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int xsft = decltype(kc)::value;
            d[xsft] = shifted(row[(xsft + WS / 2) % WS]);
        });
    }
    const int xwin = grp_reduce<WS>(shifted(my_map[(ywin - row0) * G::MAP_PITCH + r]) == gmax ? r : WS - 1, imin);   // lane r = column r

    // ---- second peak: maximum outside the (2*wv+1)^2 FLAT-index neighbourhood (B:346-358):
    //      excluded q = clamp(m + i + WS*j), |i|,|j| <= wv, i.e. in row y' the columns
    //      mx+i (j = y'-my), mx+i+WS (j = y'-my+1) and mx+i-WS (j = y'-my-1), plus the clamps.
    //      Only its VALUE is needed (the validity ratio); "none left" falls back to the first peak.
    const int m = ywin * WS + xwin;
    const int KD = WS * WS;
    const int wv = p.val_win;
    float sraw;                                       // maximum of the raw values outside the neighbourhood; NaN = none left
    {
        const unsigned long long ex = exclusion_row_mask<WS>(m, ys, wv);      // bit x' set = excluded in this lane's row
        const int exl = (int)(unsigned)ex, exh = (int)(unsigned)(ex >> 32);
        // an excluded value becomes the quiet NaN 0xffffffff (OR with the sign-extended exclusion bit), which v_max3_f32 passes
        // over (IEEE mode: the maximum of a number and a quiet NaN is the number).  Issued as inline assembly: for fmaxf() of a
        // value of unknown origin the compiler adds a canonicalising v_max_f32 x, x per element.
        float acc = __int_as_float(-1);
        static_for<0, WS / 2>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int x0 = 2 * decltype(kc)::value, x1 = x0 + 1;
            const int k0 = __builtin_amdgcn_sbfe(x0 < 32 ? exl : exh, x0 & 31, 1);
            const int k1 = __builtin_amdgcn_sbfe(x1 < 32 ? exl : exh, x1 & 31, 1);
            const float v0 = __int_as_float(__float_as_int(row[(x0 + WS / 2) % WS]) | k0);
            const float v1 = __int_as_float(__float_as_int(row[(x1 + WS / 2) % WS]) | k1);
            acc = max3_skip_nan(acc, v0, v1);
        });
        sraw = acc;
    }
    sraw = grp_reduce<WS>(sraw, [](float a, float b) TPIV_LAMBDA_INLINE { return max_skip_nan(a, b); });
    const float second_v = sraw != sraw ? gmax : shifted(sraw);

    // ---- hand-off to finalize_kernel (piv_launch.hip): the float64 logarithms, divisions and the
    //      multipass combine of B:385-422 / B:728-738 need ONE lane per window, so they run in a
    //      separate thread-per-window kernel; lanes 0..7 of the window store the raw peak data.
    {
        int left = m + 1, right = m - 1, top = m + WS, bot = m - WS;     // B:385-392 (flat index)
        if (left >= KD - 1) left = m;
        if (right <= 0) right = m;
        if (top >= KD - 1) top = m;
        if (bot <= 0) bot = m;
        int q = m;
        q = (r == 1) ? left : q;
        q = (r == 2) ? right : q;
        q = (r == 3) ? top : q;
        q = (r == 4) ? bot : q;
        float outv = shifted(my_map[(q / WS - row0) * G::MAP_PITCH + (q % WS)]);     // rows ywin-1..ywin+1 only
        outv = (r == 5) ? second_v : outv;
        outv = (r == 6) ? __int_as_float(m) : outv;
        outv = (r == 7) ? __int_as_float(dead ? 1 : 0) : outv;
        if (r < 8 && active) p.peak_raw[fidx * 8 + r] = outv;
    }
}


// ---- exact first pass (precision "exact", xcorr_exact.hip): the float32 map only LOCATES cells ------------------
// Instead of the record of peak_analysis the window gets the flat indices (fftshift layout) of
//   * the arg-max, if exactly one cell lies within EXACT_BAND x (max - min) of the maximum,
//   * up to EXACT_MAX_SECOND cells outside the exclusion zone of B:346-358 within the band of their maximum,
//   * up to EXACT_MAX_MIN cells within the band of the minimum;
// xcorr_exact_refine_kernel then evaluates those cells (and the arg-max's flat-index neighbours) as exact integer sums of the
// uint8 windows, and re-checks every decision on the exact values.
// A window with more cells inside a band than the record holds, or a flat / NaN map, is marked undecided (-1): it goes
// through the float64 transform (xcorr_f64_split_kernel<64, true>).  Dead windows (B:513: zero mean) are marked -2.
// One window per wavefront (WS = 64): every ballot below spans exactly the window.
// (EXACT_BAND, EXACT_MIN_CONTRAST, EXACT_MAX_SECOND / _MIN: piv_kernels.h)

// band_abs: the proven part of the decision band, 2 Gamma(WS) (1 + 1/16) E+ of this lane's window (piv_kernels.h, "The band")
template <int WS>
__device__ __forceinline__ void peak_candidates(const PassParams& p, const float (&row)[WS], float* tile, int w, int r,
                                                bool active, bool dead, size_t fidx, float band_abs) {
    static_assert(WS == 8 || WS == 16 || WS == 32 || WS == 64, "64 / WS windows per wavefront");
    // ballot over the lanes of this lane's window (WS < 64: the windows of the wavefront decide independently --
    // everything below is per-lane data and predicated control flow, uniform only within a window)
    auto wballot = [&](bool pred) TPIV_LAMBDA_INLINE {
        const unsigned long long b_ = __ballot(pred);
        if constexpr (WS == 64) return b_;
        else return (b_ >> (w * WS)) & ((1ull << WS) - 1ull);
    };
    // "some window of the wavefront still has a bit set" / "this lane's window has": with one window per wavefront the
    // masks are wave-uniform and both are the plain scalar test
    auto any_left = [](unsigned long long mk) TPIV_LAMBDA_INLINE {
        if constexpr (WS == 64) return mk != 0ull;
        else return __ballot(mk != 0ull) != 0ull;
    };
    auto mine_left = [](unsigned long long mk) TPIV_LAMBDA_INLINE {
        if constexpr (WS == 64) return true;
        else return mk != 0ull;
    };
    float* const my_row = tile + w * WS;              // one parked map row per window
    const int ys = (r + WS / 2) % WS;
    float rmin = 3.4e38f, rmax = -3.4e38f;
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        rmin = fminf(rmin, row[k]);
        rmax = fmaxf(rmax, row[k]);
    }
    const float cmin = grp_min<WS>(rmin);
    const float gmax = grp_reduce<WS>(rmax, [](float a, float b) TPIV_LAMBDA_INLINE { return fmaxf(a, b); });
    const float band = fmaxf(p.exact_band_range * (gmax - cmin), band_abs);
    bool open = !(band > 0.0f) || !(gmax > cmin);     // flat or NaN map
    // map row of lane rl of the window (fftshift column order) through LDS: lane r receives column r
    auto park = [&](int rl) TPIV_LAMBDA_INLINE {
        wave_sync();
        if (r == rl) {
#pragma unroll
            for (int k = 0; k < WS; ++k) my_row[(k + WS / 2) % WS] = row[k];
        }
        wave_sync();
        return my_row[r];
    };
    auto row_of_lane = [](int rl) TPIV_LAMBDA_INLINE { return (rl + WS / 2) % WS; };
    // ---- arg-max: one row, one column inside the band, or undecided
    int m = 0;
    {
        const unsigned long long rows = wballot(rmax >= gmax - band);
        open = open || __popcll(rows) != 1;
        const int rl = rows ? (int)__builtin_ctzll(rows) : 0;
        const unsigned long long cols = wballot(park(rl) >= gmax - band);
        open = open || __popcll(cols) != 1;
        m = row_of_lane(rl) * WS + (cols ? (int)__builtin_ctzll(cols) : 0);
    }
    const int wv = p.val_win;
    // ---- second peak: this lane's row maximum outside the exclusion zone, on c = row - min >= 0 (non-negative floats
    //      order like their bit patterns; excluded cells become -1, so a result >= 0 means "a cell exists")
    int s[EXACT_MAX_SECOND] = {-1, -1, -1};
    {
        const unsigned long long ex = exclusion_row_mask<WS>(m, ys, wv);
        const int exl = (int)(unsigned)ex, exh = (int)(unsigned)(ex >> 32);
        int smax = -1;
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int xsft = decltype(kc)::value;
            constexpr int xo = (xsft + WS / 2) % WS;
            const int kill = __builtin_amdgcn_sbfe(xsft < 32 ? exl : exh, xsft & 31, 1);
            const int cnd = __float_as_int(__fsub_rn(row[xo], cmin)) | kill;
            smax = cnd > smax ? cnd : smax;
        });
        const int sall = grp_reduce<WS>(smax, [](int a, int b) TPIV_LAMBDA_INLINE { return a > b ? a : b; });
        const float thr = __int_as_float(sall) - band;
        unsigned long long rows = sall >= 0 ? wballot(smax >= 0 && __int_as_float(smax) >= thr) : 0ull;
        open = open || __popcll(rows) > EXACT_MAX_SECOND;
        int ns = 0;
        for (int it = 0; it < EXACT_MAX_SECOND; ++it) {
            if (!any_left(rows)) break;                         // (no window of the wavefront has a row left)
            const bool on = mine_left(rows);
            const int rl = on ? (int)__builtin_ctzll(rows) : 0;
            rows &= rows - 1;
            const int yr = row_of_lane(rl);
            const unsigned long long exr = exclusion_row_mask<WS>(m, yr, wv);
            unsigned long long cols = wballot(__fsub_rn(park(rl), cmin) >= thr) & ~exr;
            cols = on ? cols : 0ull;
            for (int k = 0; k < WS; ++k) {                      // (at most a handful of bits are set)
                if (!any_left(cols)) break;
                const bool has = mine_left(cols);
                const int q = yr * WS + (has ? (int)__builtin_ctzll(cols) : 0);
                cols &= cols - 1;
                s[0] = (has && ns == 0) ? q : s[0];
                s[1] = (has && ns == 1) ? q : s[1];
                s[2] = (has && ns == 2) ? q : s[2];
                ns += has ? 1 : 0;
            }
        }
        open = open || ns > EXACT_MAX_SECOND;
    }
    // ---- minimum: the cells within the band of it (only the VALUE of the exact minimum is needed).  More cells than the
    //      record holds is NOT undecided by itself: frames with a true-zero background (background-subtracted recordings)
    //      have hundreds of cells where no particle pair overlaps, all at S = 0 -- and S >= 0 for every cell, so an evaluated
    //      cell with S = 0 IS the minimum.  The record then carries three cells and -2 in the fourth slot; the refinement
    //      accepts it when their exact minimum is 0 and sends the window to the float64 transform otherwise.
    int n[EXACT_MAX_MIN] = {-1, -1, -1, -1};
    {
        const float thr = cmin + band;
        unsigned long long rows = wballot(rmin <= thr);
        bool over = __popcll(rows) > EXACT_MAX_MIN;
        int nn = 0;
        for (int it = 0; it < EXACT_MAX_MIN; ++it) {
            if (!any_left(rows)) break;
            const bool on = mine_left(rows);
            const int rl = on ? (int)__builtin_ctzll(rows) : 0;
            rows &= rows - 1;
            const int yr = row_of_lane(rl);
            unsigned long long cols = wballot(park(rl) <= thr);
            cols = on ? cols : 0ull;
            over = over || nn + __popcll(cols) > EXACT_MAX_MIN;
            for (int k = 0; k < EXACT_MAX_MIN; ++k) {           // (the first cells of the row are as good as any)
                if (!any_left(cols)) break;
                const bool has = mine_left(cols) && cols != 0ull && nn < EXACT_MAX_MIN;
                const int q = yr * WS + (cols ? (int)__builtin_ctzll(cols) : 0);
                cols &= cols - 1;
                n[0] = (has && nn == 0) ? q : n[0];
                n[1] = (has && nn == 1) ? q : n[1];
                n[2] = (has && nn == 2) ? q : n[2];
                n[3] = (has && nn == 3) ? q : n[3];
                nn += has ? 1 : 0;
            }
        }
        n[3] = over ? -2 : n[3];
    }
    if (r == 0 && active) {
        const int m_out = dead ? -2 : (open ? -1 : m);
        auto pack = [](int lo_, int hi_) TPIV_LAMBDA_INLINE { return ((unsigned)lo_ & 0xffffu) | ((unsigned)hi_ << 16); };
        uint4 rec;
        rec.x = pack(m_out, s[0]);
        rec.y = pack(s[1], s[2]);
        rec.z = pack(n[0], n[1]);
        rec.w = pack(n[2], n[3]);
        p.cand[fidx] = rec;
    }
}

// ---------------------------------------------------------------------------------------------
// FAST (PassParams::precision == 0): cheaper arithmetic that differs from the !FAST form by float32
// rounding only -- see convert_rows, the mean handling after the row transform and peak_analysis<.., SCALED>.
// CAND: the peak stage writes candidate cells for the exact refinement (peak_candidates) instead of the 8-float record
// ROLE 1 (64x64 CWS, round 5): the per-pixel staging path is NOT part of this instance -- with it the kernel needs 100 spilled
// registers at the 168 three wavefronts per SIMD allow, without it 7 -- and the items that need it (frame borders, integral
// row coordinates: ~0.3 %) are appended to PassParams::slow_list instead of being processed; a second launch of the full
// kernel (ROLE 0, PassParams::list_mode = 1) runs exactly those.
template <int WS, int MODE, int OCC, bool FAST, bool CAND, int ROLE = 0>
__device__ __forceinline__ void xcorr_tile_body(const PassParams& p) {
    constexpr bool PLANAR = OCC > 2;
    using G = TileGeo<WS, PLANAR>;
    static_assert(WS == 8 || WS == 16 || WS == 32 || WS == 64, "tile sizes of this kernel");
    constexpr int RBH = 64 * (WS + 1) <= G::LDS_FLOATS ? 1 : 2;      // slow-path row buffer pieces
    static_assert(MODE == MODE_PASS1 || 64 * (WS / RBH + 1) <= G::LDS_FLOATS,
                  "slow-path row buffer (shifted passes) must fit the tile LDS");
    constexpr int LDSF = (MODE == MODE_CWS && CoopGeo<WS>::ON && CoopGeo<WS>::LDS_FLOATS > G::LDS_FLOATS)
                             ? CoopGeo<WS>::LDS_FLOATS : G::LDS_FLOATS;
    __shared__ __attribute__((aligned(16))) float tile[LDSF];

    // 16x16 has registers to spare: its twiddle constants live in VGPRs (plain 4-byte VOP2 multiplies
    // instead of 8-byte literal forms and half-rate SGPR operands): 169 -> 158 us/pair
    // for the 16x16 CWS pass at 4096^2.  For 32x32 the same change measured 1.3 % SLOWER in a
    // same-box A/B (36.1 vs 35.6 us/pair; 14 more live VGPRs), so it stays off there.
#ifndef TPIV_TWREG32
#define TPIV_TWREG32 0
#endif
    using TW = std::conditional_t<((WS == 32 && TPIV_TWREG32) || WS == 16), TwRegs<WS>, TwLiteral>;
    TW tw;
    if constexpr (!std::is_same<TW, TwLiteral>::value) tw.init();

    const int N = p.n_rows * p.n_cols;
    const int groups = (N + G::WPW - 1) / G::WPW;
    // list mode: the items are the entries of slow_list, their number is on the device
    const bool listed = ROLE == 0 && MODE != MODE_PASS1 && p.list_mode != 0;
    const int items = listed ? __builtin_amdgcn_readfirstlane((int)*p.slow_count) : p.batch * groups;   // < 2^31 (checked by the launcher)
    const int st = p.ws - p.ov;
    auto real_item = [&](int pos) TPIV_LAMBDA_INLINE { return listed ? __builtin_amdgcn_readfirstlane(p.slow_list[pos]) : pos; };

    // XCD-aware item order: workgroups b, b+8, ... share an XCD (and its L2); every XCD owns one
    // contiguous run of windows and its wavefronts pull the next item(s) from a per-XCD counter
    // (PassParams::work_ctr, zeroed by the host before the launch).  The wavefronts resident on an XCD
    // therefore always work on ADJACENT windows -- the overlapping halves and the rows shared with
    // the window row below come out of that XCD's L2 instead of HBM -- and the tail balances itself.
    // Small tiles take QCH items per atomic to keep the counter below its ~88 dequeues/us limit.
    constexpr int QCH = WS <= 16 ? 4 : 1;
    const int xcd = blockIdx.x & 7;
    const int chunk = (int)(((long long)items + 7) / 8);
    const int lo = __builtin_amdgcn_readfirstlane(xcd * chunk < items ? xcd * chunk : items);
    const int hi = __builtin_amdgcn_readfirstlane(items - lo > chunk ? lo + chunk : items);
    unsigned* const ctr = p.work_ctr + xcd * 16;          // one 64-byte line per counter
    // The dequeue is split into issue (atomic goes out at the loop head) and take (its result is read
    // after the sample conversion), so that its round trip hides behind the conversion.
    int q_base = 0;
    int q_left = 0;
    auto q_issue = [&]() TPIV_LAMBDA_INLINE -> unsigned {      // returns lane 0's counter value (a VGPR)
        unsigned v = 0;
        if (QCH == 1 || q_left == 0) {
            if (fresh_lane() == 0) v = atomicAdd(ctr, (unsigned)QCH);
        }
        return v;
    };
    auto q_take = [&](unsigned q_raw) TPIV_LAMBDA_INLINE -> int {
        if (QCH == 1 || q_left == 0) {
            // (a counter value past the end of the run can be anything below 2^31 + grid size: clamp)
            const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)q_raw);
            q_base = v < (unsigned)(hi - lo) ? lo + (int)v : hi;
            q_left = QCH;
        }
        const int it = q_base + (QCH - q_left);
        --q_left;
        return it;
    };

    // per-lane geometry of an item; `w` = the lane's window slot (a fresh copy at every call site)
    auto geom_of = [&](int pos, int w) TPIV_LAMBDA_INLINE {
        ItemGeom g;
        const int item = real_item(pos);
        g.pair = fast_div(item, p.groups_magic, p.groups_shift);
        const int gi = item - g.pair * groups;
        const int win_raw = gi * G::WPW + w;
        g.active = win_raw < N ? 1 : 0;
        g.win = g.active ? win_raw : N - 1;
        const int wrow = fast_div(g.win, p.ncols_magic, p.ncols_shift);
        g.y0 = wrow * st;
        g.x0 = (g.win - wrow * p.n_cols) * st;
        g.fidx = (size_t)g.pair * N + g.win;
        return g;
    };
    auto shift_of = [&](const ItemGeom& g, float& vx, float& vy) TPIV_LAMBDA_INLINE {
        if constexpr (MODE == MODE_PASS1) {
            vx = 0.f;
            vy = 0.f;
        } else {      // DWS: exact small integers; CWS: the float32 cast of B:714-715
            if constexpr (MODE == MODE_CWS) {
                pred_half_shift_cws_f32(p, g.fidx, vx, vy);
            } else {
                double sx, sy;
                pred_half_shift<MODE>(p, g.fidx, sx, sy);
                vx = (float)sx;
                vy = (float)sy;
            }
        }
    };

    int item = q_take(q_issue());
    if (item >= hi) return;
    int nitem = q_take(q_issue());                       // the queue runs two items ahead of the FFTs
    // software pipeline: rows of item i+1 in flight during item i (its shifts are fetched at the loop
    // head and have the conversion of item i to land)
    // Tiles of 32 and 64 are register-bound: they re-derive the per-lane geometry where it is needed
    // (scalar magic division + a few VALU instructions).  Small tiles have registers to spare and
    // short iterations, so they carry it instead (recomputing cost the 16x16 pass 5 %).
    constexpr bool RECOMPUTE = WS >= 32;
    float vx, vy;
    RawRows<WS, MODE> raw;
    ItemGeom gcur;
    CoopOff<WS> coff;
    if constexpr (MODE == MODE_CWS) coff.init(fresh_lane() % WS, p.W);
    {
        const int l0 = fresh_lane();
        gcur = geom_of(item, l0 / WS);
        shift_of(gcur, vx, vy);
        issue_rows<WS, MODE, FAST>(p, gcur, l0 % WS, vx, vy, raw, coff);
    }

    TPIV_STAMP_DECL
    TPIV_STAMP_START;
    for (int nnitem; item < hi; item = nitem, nitem = nnitem) {
        const unsigned q_raw = q_issue();
#ifdef TPIV_STAMPS
        ++st_iter;
#endif
        const int lane = fresh_lane();
        const int w = lane / WS;          // window slot inside the wavefront
        const int r = lane % WS;          // image row of the window held by this lane
        const ItemGeom g = RECOMPUTE ? geom_of(item, w) : gcur;
        const bool active = g.active != 0;
        const size_t fidx = g.fidx;
        // (the last iteration simply re-loads its own item: no branch around the prefetch)
        const int nit = nitem < hi ? nitem : item;
        const ItemGeom gnext = geom_of(nit, w);
        float nvx, nvy;
        shift_of(gnext, nvx, nvy);

        if constexpr (ROLE == 1) {
            if (!raw.reg) {                // (wave-uniform) an item for the full kernel: note it, fetch the next one's rows, go on
                if (lane == 0) p.slow_list[atomicAdd(p.slow_count, 1u)] = item;
                nnitem = q_take(q_raw);
                issue_rows<WS, MODE, FAST>(p, gnext, r, nvx, nvy, raw, coff);
                vx = nvx;
                vy = nvy;
                if constexpr (!RECOMPUTE) gcur = gnext;
                continue;
            }
        }
        cf x[WS];
        float sa, sb;                      // window sums (for the mean)
        // CAND: the decision band of the lane's window (float bits).  Register-bound tiles keep it in scalar registers, one
        // per window of the wavefront; small tiles have registers to spare
        constexpr int NBS = WS >= 32 ? G::WPW : 1;
        int band_s[NBS];
        TPIV_STAMP(0);      // loop head: geometry, next shifts, combine loads
        RowSquares sq;
        convert_rows<WS, MODE, RBH, FAST, CAND, ROLE == 1 ? 1 : 0>(p, g, r, lane, vx, vy, raw, x, sa, sb, tile, &sq);
        TPIV_STAMP(1);      // wait for the rows + conversion / bilinear sampling
        // 64x64 (register-bound): the dequeue issued at the loop head has landed by now; move it to a
        // scalar register (kept in a VGPR to the loop end it would be spilled, and the reload would
        // wait behind the prefetch).  Smaller tiles: the conversion is too short to cover the atomic's
        // round trip (the stamps showed the wait), and a VGPR to the loop end costs nothing there.
        if constexpr (WS >= 64) nnitem = q_take(q_raw);
        // small tiles: the row registers are free again, so the next item's loads go out now and
        // have the whole iteration to land (64x64 is register-bound: it waits until the peak search)
        if constexpr (WS <= 32) issue_rows<WS, MODE, FAST>(p, gnext, r, nvx, nvy, raw, coff);

        if (p.dbg_win != nullptr && active) {     // test hook: the staged (shifted) windows
            // (the row index is made opaque here: otherwise the WS loop-invariant store addresses get
            //  hoisted out of the item loop and occupy WS VGPRs for the whole kernel)
            int rr = r;
            asm volatile("" : "+v"(rr));
            float* d = p.dbg_win + fidx * 2 * WS * WS + rr * WS;
            // (fast-order CWS samples carry 2^-24, see byte_d)
            constexpr float UNS = (MODE == MODE_CWS && FAST && TPIV_SDWA_LERP) ? SDWA_INV : 1.0f;
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                d[k] = x[k].x * UNS;
                d[WS * WS + k] = x[k].y * UNS;
            }
        }

        // ---- mean removal: any constant offset leaves corr - min(corr) unchanged and only
        //      conditions the float32 transform.  Pass 1 also divides by the mean (B:513-514).
        bool dead = false;        // pass 1: zero-mean window -> 0/0 = NaN map in the reference
        float end_scale = 1.0f;
        // Pass 1 is compared with a float64 reference and its normalised maps have peak neighbours AT the
        // map minimum in sparse 8x8 windows, where the log-ratio fit amplifies every rounding error: it
        // keeps the mean removal in front of the transform (128 fma per 64x64 window row, 2 % of the pass).
        // The shifted passes are float32 in the reference itself, with the full pedestal in both transforms.
        constexpr bool FASTN = FAST && MODE != MODE_PASS1;
        if constexpr (!FASTN) {
            sa = grp_sum<WS>(sa);
            sb = grp_sum<WS>(sb);
            const float ma = sa * (1.0f / (WS * WS)), mb = sb * (1.0f / (WS * WS));
            float ka = 1.f, kb = 1.f;
            if constexpr (MODE == MODE_PASS1) {
                dead = (sa == 0.f) || (sb == 0.f);
                ka = dead ? 0.f : 1.0f / ma;
                kb = dead ? 0.f : 1.0f / mb;
            }
            if constexpr (CAND) {
                // E+ = (|a'|^2 + |b'|^2) / 2 with a' = a / mean(a) - 1:  |a'|^2 = n (n sum a^2 - (sum a)^2) / (sum a)^2, the
                // bracket from exact integers (< 2^42: exact in float64), the rest in float32 (relative error ~1e-6 against
                // the 1/16 margin of exact_band_coef).  Kept in scalar registers to the peak stage: one per window.
                auto uadd = [](unsigned a_, unsigned b_) TPIV_LAMBDA_INLINE { return a_ + b_; };
                const unsigned saa = grp_reduce<WS>(sq.aa, uadd), sbb = grp_reduce<WS>(sq.bb, uadd);
                constexpr double NN = (double)(WS * WS);
                const double da = (double)sa, db = (double)sb;
                const float ea = (float)__fma_rn(-da, da, NN * (double)saa), eb = (float)__fma_rn(-db, db, NN * (double)sbb);
                const float e_plus = (0.5f / (float)(WS * WS)) * (ea * (ka * ka) + eb * (kb * kb));       // ka = n / sum a
                const float bnd = dead ? 0.f : p.exact_band * e_plus;
                if constexpr (WS >= 32) {
                    static_for<0, G::WPW>([&](auto wc) TPIV_LAMBDA_INLINE {
                        band_s[decltype(wc)::value] = __builtin_amdgcn_readlane(__float_as_int(bnd), decltype(wc)::value * WS);
                    });
                } else {
                    band_s[0] = __float_as_int(bnd);
                }
            }
            // The 1/n^2 of the inverse transform and the 1/4 of the cross-spectrum algebra are applied
            // HERE, as the power of two 0.5/WS on both inputs: exact (no rounding anywhere changes), and
            // it saves the per-bin scaling multiplies of the cross-spectrum.
            constexpr float PRE = 0.5f / (float)WS;
            const float oa = -ma * ka, ob = -mb * kb;
            const float kas = ka * PRE, kbs = kb * PRE, oas = oa * PRE, obs = ob * PRE;
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                x[k].x = fmaf(x[k].x, kas, oas);        // (x - mean) * k * 0.5/WS
                x[k].y = fmaf(x[k].y, kbs, obs);
            }
        }

        TPIV_STAMP(2);      // mean reduction + normalisation
        // ---- forward 2-D transform of a + i*b: rows in registers, transpose, columns in registers
        fft_inreg<WS, 1>(x, tw);                          // over x; bin kx at x[FFT_POS<kx>]
        if constexpr (FASTN) {
            // The samples went in as they are.  Bin kx = 0 of a lane's row is the row sum (a in .x, b in .y):
            // the window mean is removed THERE (one subtraction per lane instead of one per sample; only
            // the row transform has seen the pedestal), and the constant factor of the map -- 1/n^2 and the
            // 1/4 of the packed spectrum -- is applied by peak_analysis.
            constexpr int P0 = FFT_POS<0, WS>;
            const float ta = grp_sum<WS>(x[P0].x);
            const float tb = grp_sum<WS>(x[P0].y);
            x[P0].x -= ta * (1.0f / WS);
            x[P0].y -= tb * (1.0f / WS);
            end_scale = 0.25f / (float)(WS * WS);
            if constexpr (MODE == MODE_CWS && TPIV_SDWA_LERP) end_scale *= SDWA_INV * SDWA_INV;     // both frames' samples carry 2^-24
        }
        TPIV_STAMP(3);      // forward row FFT
        transpose_tile<WS, true, PLANAR>(x, tile, fresh_lane());  // lane = kx, x[y] natural
        TPIV_STAMP(4);      // transposition 1
        fft_inreg<WS, 1>(x, tw);                          // over y; Z(ky, kx = lane) at x[FFT_POS<ky>]
        TPIV_STAMP(5);      // forward column FFT

        // ---- cross-spectrum.  A = (Z(k) + conj Z(-k))/2, B = (Z(k) - conj Z(-k))/(2i),
        //      P = conj(A) * B / n^2.  Z(-ky, -kx) sits in lane (-kx mod WS), register (-ky mod WS).
        {
            const int lane_c = fresh_lane();
            const int r_c = lane_c % WS;
            const int partner = (lane_c - r_c) + ((WS - r_c) % WS);
            // with zk = a + ib, zm = Z(-k) = c + id:  4 P = conj(2A) * (2B)
            //   re = (a+c)(b+d) + (b-d)(c-a) = 2 (a d + b c),   im = (c^2 - a^2) + (d^2 - b^2)
            // (the factor 0.25 / WS^2 is already in the inputs, see the normalisation above)
            auto cross = [&](cf zk, cf zm) TPIV_LAMBDA_INLINE {
                const float a_ = zk.x, b_ = zk.y, c_ = zm.x, d_ = zm.y;
                cf pr;
                pr.x = (a_ * d_ + b_ * c_) * 2.0f;
                pr.y = (c_ * c_ - a_ * a_) + (d_ * d_ - b_ * b_);
                return pr;
            };
            static_for<0, WS / 2 + 1>([&](auto kc) TPIV_LAMBDA_INLINE {
                constexpr int ky = decltype(kc)::value;
                constexpr int nky = (WS - ky) % WS;
                constexpr int p1 = FFT_POS<ky, WS>, p2 = FFT_POS<nky, WS>;
                const cf z1 = x[p1];
                if constexpr (ky == nky) {
                    cf m1;
                    m1.x = __shfl(z1.x, partner, 64);
                    m1.y = __shfl(z1.y, partner, 64);
                    x[p1] = cross(z1, m1);
                } else {
                    const cf z2 = x[p2];
                    cf m1, m2;
                    m1.x = __shfl(z2.x, partner, 64);     // Z(-ky, -kx)
                    m1.y = __shfl(z2.y, partner, 64);
                    m2.x = __shfl(z1.x, partner, 64);     // Z(+ky, -kx), the partner of bin -ky
                    m2.y = __shfl(z1.y, partner, 64);
                    x[p1] = cross(z1, m1);
                    x[p2] = cross(z2, m2);
                }
            });
        }

        TPIV_STAMP(6);      // cross-spectrum incl. the bpermute exchange
        // ---- inverse: columns (natural-order input: rename registers), transpose, rows
        cf t[WS];
        static_for<0, WS>([&](auto kc) TPIV_LAMBDA_INLINE {
            constexpr int ky = decltype(kc)::value;
            t[ky] = x[FFT_POS<ky, WS>];
        });
        fft_inreg<WS, -1>(t, tw);                         // over ky; row y at t[FFT_POS<y>]
        TPIV_STAMP(7);      // inverse column FFT
#ifndef TPIV_C2R
#define TPIV_C2R 1
#endif
        float crow[WS];
        // (64x64: the planar three-wavefront kernels take the c2r form too -- round 1 and the first try of round 2
        //  ended in 32-42 spills at the 168-VGPR cap, which were the compiler sinking the tail of the column
        //  transform into the conditional stores of transpose_half; with the values pinned in front of the branch
        //  the kernel keeps its one spill and pass 1 went from 6.94 to 6.42 ms per 256 pairs.  The complex-tile
        //  64x64 CWS kernel keeps the complex last transform.)
#ifndef TPIV_C2R64
#define TPIV_C2R64 1
#endif
        if constexpr (TPIV_C2R && (WS <= 32 || (TPIV_C2R64 && WS == 64))) {
            // the map rows are real: only spectrum columns 0..WS/2 cross the LDS and a WS/2-point complex
            // transform yields the row as z[m] = corr(y, 2m) + i corr(y, 2m + 1)  (c2r_inreg)
            cf hs[WS / 2 + 1], z[WS / 2];
            transpose_half<WS, PLANAR>(t, hs, tile, fresh_lane());
            TPIV_STAMP(8);      // transposition 2
            c2r_inreg<WS>(hs, z);
            static_for<0, WS>([&](auto xc) TPIV_LAMBDA_INLINE {
                constexpr int x_ = decltype(xc)::value;
                crow[x_] = (x_ & 1) ? z[FFT_POS<x_ / 2, WS / 2>].y : z[FFT_POS<x_ / 2, WS / 2>].x;
            });
        } else {
            transpose_tile<WS, true, PLANAR>(t, tile, fresh_lane());  // lane = y, t[kx] natural
            TPIV_STAMP(8);      // transposition 2
            fft_inreg<WS, -1>(t, tw);                     // over kx; corr(y = lane, x) at t[FFT_POS<x>].x
            static_for<0, WS>([&](auto xc) TPIV_LAMBDA_INLINE {
                constexpr int x_ = decltype(xc)::value;
                crow[x_] = t[FFT_POS<x_, WS>].x;
            });
        }
        wave_sync();                                  // tile reads done: it becomes the map
        TPIV_STAMP(9);      // inverse row FFT

        auto band_of = [&](int w_) TPIV_LAMBDA_INLINE {
            int b_ = band_s[0];
            static_for<1, NBS>([&](auto wc) TPIV_LAMBDA_INLINE { b_ = w_ == decltype(wc)::value ? band_s[decltype(wc)::value] : b_; });
            return __int_as_float(b_);
        };
        // ---- prefetch: the next item's row loads fly while this item's peak search runs
        if constexpr (WS > 32) {
            const int lane_p = fresh_lane();
            issue_rows<WS, MODE, FAST>(p, geom_of(nit, lane_p / WS), lane_p % WS, nvx, nvy, raw, coff);
        }
        vx = nvx;
        vy = nvy;
        if constexpr (!RECOMPUTE) gcur = gnext;

        TPIV_STAMP(10);     // issue of the next item's row loads
        if constexpr (RECOMPUTE) {
            // the window index is re-derived from the (wave-uniform) item position, see above
            const int lane_e = fresh_lane();
            const int w_e = lane_e / WS, r_e = lane_e % WS;
            const int item_e = real_item(item);
            const int pair_e = fast_div(item_e, p.groups_magic, p.groups_shift);
            const int win_raw_e = (item_e - pair_e * groups) * G::WPW + w_e;
            const bool active_e = win_raw_e < N;
            const int win_e = active_e ? win_raw_e : N - 1;
            const size_t fidx_e = (size_t)pair_e * N + win_e;
            if constexpr (CAND) peak_candidates<WS>(p, crow, tile, w_e, r_e, active_e, dead, fidx_e, band_of(w_e));
            else peak_analysis<WS, PLANAR, FASTN>(p, crow, tile, w_e, r_e, active_e, dead, fidx_e, end_scale);
        } else {
            if constexpr (CAND) peak_candidates<WS>(p, crow, tile, w, r, active, dead, fidx, band_of(w));
            else peak_analysis<WS, PLANAR, FASTN>(p, crow, tile, w, r, active, dead, fidx, end_scale);
        }
        wave_sync();
        if constexpr (WS < 64) nnitem = q_take(q_raw);
        TPIV_STAMP(13);     // sub-pixel fit, combine, stores
    }
    TPIV_STAMP_FLUSH(p);
}

template <int WS, int MODE, int OCC, bool FAST>
__global__ __launch_bounds__(64, OCC) void xcorr_tile_kernel(PassParams p) {
    xcorr_tile_body<WS, MODE, OCC, FAST, false>(p);
}
// 64x64 CWS, fast-order arithmetic: the instance without the per-pixel staging path, three wavefronts per SIMD (ROLE 1)
template <int WS, int MODE>
__global__ __launch_bounds__(64, 3) void xcorr_tile_fastpath_kernel(PassParams p) {
    xcorr_tile_body<WS, MODE, 3, true, false, 1>(p);
}
// float32 first pass of the exact scheme: same transforms, candidate cells out
template <int WS>
__global__ __launch_bounds__(64, tile_occ_c(WS, MODE_PASS1)) void xcorr_tile_cand_kernel(PassParams p) {
    xcorr_tile_body<WS, MODE_PASS1, tile_occ_c(WS, MODE_PASS1), true, true>(p);
}

// ---- test hook: feed hand-made correlation maps straight into peak_analysis ------------------------
// maps: [n_windows, WS, WS] float32 in fftshift layout (what correlation_to_displacement receives,
// before its `+= eps`); one wavefront handles 64/WS maps.  Output: peak_raw records.
template <int WS, bool PLANAR>
__global__ __launch_bounds__(64, 2) void peak_debug_kernel(PassParams p, const float* maps, int n_maps) {
    using G = TileGeo<WS, PLANAR>;
    __shared__ float tile[G::LDS_FLOATS];
    const int lane = threadIdx.x;
    const int w = lane / WS, r = lane % WS;
    const int win_raw = blockIdx.x * G::WPW + w;
    const bool active = win_raw < n_maps;
    const int win = active ? win_raw : n_maps - 1;
    float crow[WS];
    const int ys = (r + WS / 2) % WS;
#pragma unroll
    for (int xo = 0; xo < WS; ++xo) crow[xo] = maps[((size_t)win * WS + ys) * WS + (xo + WS / 2) % WS];
    peak_analysis<WS, PLANAR>(p, crow, tile, w, r, active, false, (size_t)win);
}

// planar != 0: the LDS layout of the three-wavefront kernels (for 64x64: the three-row SMALLMAP form)
template <int WS>
hipError_t launch_peak_debug(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream) {
    using G = TileGeo<WS, false>;
    const int blocks = (n_maps + G::WPW - 1) / G::WPW;
    if (planar)
        hipLaunchKernelGGL((peak_debug_kernel<WS, true>), dim3(blocks), dim3(64), 0, stream, p, maps, n_maps);
    else
        hipLaunchKernelGGL((peak_debug_kernel<WS, false>), dim3(blocks), dim3(64), 0, stream, p, maps, n_maps);
    return hipGetLastError();
}

template <int WS, int MODE>
static hipError_t launch_tile(const PassParams& p_in, int n_cu, hipStream_t stream) {
    using G = TileGeo<WS>;
    PassParams p = p_in;
    const int N = p.n_rows * p.n_cols;
    const long long groups = (N + G::WPW - 1) / G::WPW;
    const long long items = (long long)p.batch * groups;
    if (items <= 0 || items >= (1ll << 31) - 64) return hipErrorInvalidValue;     // 32-bit item arithmetic
    fast_div_setup((unsigned)groups, p.groups_magic, p.groups_shift);
    fast_div_setup((unsigned)p.n_cols, p.ncols_magic, p.ncols_shift);
    long long blocks = items;
    // Up to 64 single-wavefront workgroups per CU; each pulls items from its XCD's counter until the
    // run is empty, so surplus workgroups exit at once and the grid size only has to cover the
    // resident set.  (History: with a static grid-stride order 64/CU beat the exact resident set,
    // 98.2 vs 104.5 us/pair, because of dynamic balancing, but spread every XCD over its whole run:
    // FETCH_SIZE was 1.5x (pass 1) to 3x (pass 2) the algorithmic bytes.  The queue keeps both.)
    // TPIV_WG_PER_CU overrides for experiments.
    static const int wg_per_cu = [] {
        const char* e = getenv("TPIV_WG_PER_CU");
        return e ? atoi(e) : 0;
    }();
    const long long cap = (long long)n_cu * (wg_per_cu > 0 ? wg_per_cu : 64);
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;                  // the XCD remap needs a multiple of 8
    // Register budget (wavefronts per SIMD).  32x32, 16x16 and 64x64 pass 1 run three wavefronts per
    // SIMD (<= 168 VGPRs, planar LDS tiles of 8.4 KB per wavefront).  Measured vs two wavefronts:
    // 32x32 DWS pass 33.6 -> 27.4 us/pair, 32x32 CWS pass 48.0 -> 43.9, 16x16 CWS pass (4096^2)
    // 213.9 -> 189.6, 64x64 pass 1 32.3 -> 29.1; four (possible for 32x32 pass 1 / DWS and 16x16 since
    // the peak search stopped using compare/select chains) gains nothing more: the VALU is saturated.
    // 16x16 is built for four (its CWS variant needs 129 VGPRs unconstrained, one more than four
    // wavefronts allow: 165 -> 155 us/pair at 4096^2); 8x8 is built for two but small enough (107
    // VGPRs) to run four.  64x64 pass 1 and DWS run three (planar tiles, three-row map, slow-path row
    // buffer in two pieces); the 64x64 CWS variant needs 230 VGPRs and stays at two.
    // Only the chosen variant is instantiated (tile_occ_c, piv_kernels.h); -DTPIV_EXPERIMENT builds all
    // of them and lets TPIV_OCC=2|3|4 pick one at run time.
    constexpr int OCC = tile_occ_c(WS, MODE);
    // precision "reference": shifted passes keep the reference's operation order (bit-identical staged
    // windows); pass 1 at that precision is the float64 kernel (xcorr_f64.hip), so the float32 pass 1
    // exists in the fast form only
    if constexpr (MODE != MODE_PASS1) {
        if (p.precision != 0) {
            hipLaunchKernelGGL((xcorr_tile_kernel<WS, MODE, OCC, false>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
            return hipGetLastError();
        }
    }
#ifdef TPIV_EXPERIMENT
    static const int occ_env = [] {
        const char* e = getenv("TPIV_OCC");
        return e ? atoi(e) : 0;
    }();
    if (occ_env == 2 && OCC != 2) {
        hipLaunchKernelGGL((xcorr_tile_kernel<WS, MODE, 2, true>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
        return hipGetLastError();
    }
    if constexpr (WS <= 32 || MODE != MODE_CWS) {
        if (occ_env == 3 && OCC != 3) {
            hipLaunchKernelGGL((xcorr_tile_kernel<WS, MODE, 3, true>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
            return hipGetLastError();
        }
    }
    if constexpr (WS <= 32) {
        if (occ_env == 4 && OCC != 4) {
            hipLaunchKernelGGL((xcorr_tile_kernel<WS, MODE, 4, true>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
            return hipGetLastError();
        }
    }
#endif
    if constexpr (tile_split64(WS, MODE)) {
        // two launches: everything that takes the wide-load path at three wavefronts per SIMD (no per-pixel code in that
        // instance), then the full kernel over the items it set aside (their number is on the device: the grid is a
        // resident set, wavefronts without an item leave at once)
        if (p.slow_list != nullptr && p.slow_count != nullptr) {
            hipLaunchKernelGGL((xcorr_tile_fastpath_kernel<WS, MODE>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
            hipError_t e_ = hipGetLastError();
            if (e_ != hipSuccess) return e_;
            e_ = hipMemsetAsync(p.work_ctr, 0, TILE_CTR_BYTES, stream);
            if (e_ != hipSuccess) return e_;
            PassParams q = p;
            q.list_mode = 1;
            long long blocks2 = (long long)n_cu * 8;
            if (blocks2 > blocks) blocks2 = blocks;
            blocks2 = (blocks2 + 7) / 8 * 8;
            hipLaunchKernelGGL((xcorr_tile_kernel<WS, MODE, OCC, true>), dim3((unsigned)blocks2), dim3(64), 0, stream, q);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((xcorr_tile_kernel<WS, MODE, OCC, true>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// the candidate-cell variant of the float32 first pass (precision "exact"); same grid and work queue as launch_tile
template <int WS>
hipError_t launch_xcorr_tile_cand_ws(const PassParams& p_in, int n_cu, hipStream_t stream) {
    using G = TileGeo<WS>;
    PassParams p = p_in;
    const int N = p.n_rows * p.n_cols;
    const long long groups = (N + G::WPW - 1) / G::WPW;
    const long long items = (long long)p.batch * groups;
    if (items <= 0 || items >= (1ll << 31) - 64 || p.cand == nullptr) return hipErrorInvalidValue;
    fast_div_setup((unsigned)groups, p.groups_magic, p.groups_shift);
    fast_div_setup((unsigned)p.n_cols, p.ncols_magic, p.ncols_shift);
    long long blocks = items < (long long)n_cu * 64 ? items : (long long)n_cu * 64;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((xcorr_tile_cand_kernel<WS>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
    return hipGetLastError();
}

// one translation unit per tile size instantiates this (xcorr_ws*.hip)
template <int WS>
hipError_t launch_xcorr_tile_ws(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    switch (mode) {
        case MODE_PASS1: return launch_tile<WS, MODE_PASS1>(p, n_cu, stream);
        case MODE_DWS: return launch_tile<WS, MODE_DWS>(p, n_cu, stream);
        case MODE_CWS: return launch_tile<WS, MODE_CWS>(p, n_cu, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace tpiv
