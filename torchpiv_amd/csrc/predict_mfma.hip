// Predictor of the shifted passes as two banded float64 GEMMs on the matrix cores.
//
// Reference (PIVbackend.py:692-711 CWS, :765-785 DWS): RectBivariateSpline(coarse grid)(fine grid) of the
// previous pass's u, v and validity mask.  Interpolating splines are linear in the data, so the plan holds
// the two operators  out = Ay . Z . Ax^T  (c_api.cpp: the reference's own FITPACK calls applied to unit
// vectors, in float64), and their rows decay like 0.268^|j - j0|: everything outside 65 taps is below the
// rounding of the sum.  Per block of 16 fine rows / columns the band is a dense K x 16 tile (K = band +
// the drift of the band start over the block, rounded up to a multiple of 16; zeros elsewhere):
//
//   rows:  T1t[b, f, cc, rf] = sum_k Z_f[b, k0y + k, cc] * Wy16[rf / 16][k][rf % 16]        (f = u, v, mask)
//   cols:  out[b, rf, cf]    = sum_k T1t[b, f, k0x + k, rf] * Ax16[cf / 16][k][cf % 16]  -> u0, v0, u2, v2
//
// Both are v_mfma_f64_16x16x4_f64 chains, one wavefront per 16 x 16 output tile and the three fields side
// by side (three accumulators, so consecutive MFMAs never wait for each other).  T1 is kept TRANSPOSED
// ([cc][rf], pitch nrfp = nrf rounded up to 16): with the operand layout of the instruction
//     lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16], and holds D[4 r + l / 16][l % 16] in register r
// (found with tools/micro/mfma_f64_layout.hip) every global access of both kernels — operands, weights,
// results — is a run of 16 consecutive doubles per quarter wavefront: no LDS staging, no transposition.
// The vector form of round 1 needed one LDS read per 4 FMAs and one 32-byte weight load per 24 and sat at
// 15 % of the float64 rate; measured rates: DESIGN.md 3.3.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "piv_kernels.h"

namespace tpiv {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// Operands come through buffer descriptors: a 32-bit byte offset per lane (no 64-bit address pairs: 16 loads
// are in flight per trip), and the range check returns 0 for taps beyond the matrix (their weights are 0).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}

// A wavefront owns a 32 x 32 output tile of the three fields: 2 x 2 MFMA tiles x 3 = 12 accumulators.  Per
// K-step it loads 6 operand values and 2 weights for 12 MFMAs.  (With one 16 x 16 tile per wavefront every
// MFMA needed 1.33 loads of 512 bytes and the kernels were bound by the 64 B/clk of the CU's vector L1 at
// 42 % matrix-core utilisation.)  One trip = two K-steps = 16 loads.
struct Trip {
    double x[2][3][2];      // [step][field][row half]
    double w[2][2];         // [step][column half]
};
struct Acc {
    d4 a[3][2][2];          // [field][row half][column half]
};

__device__ __forceinline__ void run_trip(const Trip& t, Acc& c) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
#pragma unroll
            for (int hj = 0; hj < 2; ++hj)
#pragma unroll
                for (int f = 0; f < 3; ++f) c.a[f][hi][hj] = mfma(t.x[j][f][hi], t.w[j][hj], c.a[f][hi][hj]);
}

// The K loop, two trips in flight: the loads of trip s + 1 are issued before the MFMAs of trip s (the
// counter waits are in order, so the MFMAs of s wait for their own 16 loads only).  `issue(t, s)` fills t;
// trip 0 is already on its way in `a` when the loop starts.
template <typename Issue>
__device__ __forceinline__ void k_loop(int trips, Issue issue, Trip& a, Acc& c) {
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int h = 0; h < 4; ++h) c.a[f][h >> 1][h & 1] = d4{0, 0, 0, 0};
    Trip b;
    for (int s = 0;;) {
        if (s + 1 < trips) issue(b, s + 1);
        run_trip(a, c);
        if (++s >= trips) break;
        if (s + 1 < trips) issue(a, s + 1);
        run_trip(b, c);
        if (++s >= trips) break;
    }
}

// A wavefront walks several tiles (stride gridDim over the tile index that keeps its weights the same), and
// the first trip of the NEXT tile is requested before the results of this one are stored: the stores then
// drain under the next tile's MFMAs.  (One tile per wavefront: all wavefronts of a SIMD share the matrix core
// round-robin, finish together and store together -- the store phase, 540 MB per launch of configs[3]'s last
// predictor, ran with the matrix cores idle: 274 us against 167 us without the stores.)

// grid (x slices, ceil(n_blk_y / 4), batch), 4 wavefronts = 4 consecutive blocks of 32 fine rows; a wavefront
// takes the blocks of 32 coarse columns x, x + gridDim.x, ...
__global__ __launch_bounds__(256, 2) void predict_rows_mfma_kernel(BandedPredictParams q) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int rb = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (rb * 32 >= q.nrf) return;                                   // wave-uniform
    const int b = blockIdx.z;
    const int k0 = q.k0y32[rb];
    const size_t zn = (size_t)q.nrc * q.ncc;
    const __amdgpu_buffer_rsrc_t rw = rsrc(q.Wy32 + (size_t)rb * q.KY * 32, (size_t)q.KY * 256);
    const __amdgpu_buffer_rsrc_t ru = rsrc(q.u_c + b * zn, zn * 8), rv = rsrc(q.v_c + b * zn, zn * 8);
    const __amdgpu_buffer_rsrc_t rm = rsrc(q.val_c + b * zn, zn);
    const unsigned wo = (lk * 32 + li) * 8;
    // element offset; rows >= nrc read as 0, columns >= ncc read the next row: their results are not stored
    const unsigned zo = (unsigned)(k0 + lk) * q.ncc + li;
    const unsigned zstep = 4u * q.ncc;
    const size_t plane = (size_t)q.ncc * q.nrfp;
    const int nbc = (q.ncc + 31) / 32;
    auto issue = [&](Trip& t, int s, int cc0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned o = zo + cc0 + (2 * s + j) * zstep + 16 * h;
                t.w[j][h] = bload(rw, wo + (2 * s + j) * 1024 + 128 * h);
                t.x[j][0][h] = bload(ru, o * 8);
                t.x[j][1][h] = bload(rv, o * 8);
                t.x[j][2][h] = (double)__builtin_amdgcn_raw_buffer_load_b8(rm, o, 0, 0);
            }
        }
    };
    Trip a;
    issue(a, 0, blockIdx.x * 32);
    for (int cb = blockIdx.x; cb < nbc; cb += gridDim.x) {
        const int cc0 = cb * 32;
        Acc c;
        k_loop(q.KY >> 3, [&](Trip& t, int s) { issue(t, s, cc0); }, a, c);
        if (cb + (int)gridDim.x < nbc) issue(a, 0, cc0 + 32 * gridDim.x);
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cc = cc0 + 16 * hi + 4 * r + lk;
                if (cc >= q.ncc) continue;
#pragma unroll
                for (int hj = 0; hj < 2; ++hj) {
                    const size_t o = (size_t)b * 3 * plane + (size_t)cc * q.nrfp + rb * 32 + 16 * hj + li;    // < nrfp: padding rows get zeros
#pragma unroll
                    for (int f = 0; f < 3; ++f) q.T1[o + f * plane] = c.a[f][hi][hj][r];
                }
            }
    }
}

// grid (ceil(n_blk_x / 4), y slices, batch), 4 wavefronts = 4 consecutive blocks of 32 fine columns; a
// wavefront takes the blocks of 32 fine rows y, y + gridDim.y, ...
__global__ __launch_bounds__(256, 2) void predict_cols_mfma_kernel(BandedPredictParams q) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g * 32 >= q.ncf) return;                                    // wave-uniform
    const int b = blockIdx.z;
    const int k0 = q.k0x32[g];
    const size_t plane = (size_t)q.ncc * q.nrfp;
    const double* tb = q.T1 + (size_t)b * 3 * plane;
    const __amdgpu_buffer_rsrc_t rw = rsrc(q.Ax32 + (size_t)g * q.KX * 32, (size_t)q.KX * 256);
    const __amdgpu_buffer_rsrc_t rt[3] = {rsrc(tb, plane * 8), rsrc(tb + plane, plane * 8), rsrc(tb + 2 * plane, plane * 8)};
    const unsigned wo = (lk * 32 + li) * 8;
    const unsigned to = ((unsigned)(k0 + lk) * q.nrfp + li) * 8;    // byte offset; rows >= ncc read as 0
    const unsigned tstep = 32u * q.nrfp;
    const int nbr = q.nrfp >> 5;
    auto issue = [&](Trip& t, int s, int rf0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned o = to + rf0 * 8 + (2 * s + j) * tstep + 128 * h;
                t.w[j][h] = bload(rw, wo + (2 * s + j) * 1024 + 128 * h);
#pragma unroll
                for (int f = 0; f < 3; ++f) t.x[j][f][h] = bload(rt[f], o);
            }
        }
    };
    Trip a;
    issue(a, 0, blockIdx.y * 32);
    for (int rbk = blockIdx.y; rbk < nbr; rbk += gridDim.y) {
        const int rf0 = rbk * 32;
        Acc c;
        k_loop(q.KX >> 3, [&](Trip& t, int s) { issue(t, s, rf0); }, a, c);
        if (rbk + (int)gridDim.y < nbr) issue(a, 0, rf0 + 32 * gridDim.y);
#pragma unroll
        for (int hj = 0; hj < 2; ++hj) {
            const int cf = g * 32 + 16 * hj + li;
            if (cf >= q.ncf) continue;
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rf = rf0 + 16 * hi + 4 * r + lk;
                    if (rf >= q.nrf) continue;
                    double u0 = c.a[0][hi][hj][r], v0 = c.a[1][hi][hj][r];
                    const bool val = c.a[2][hi][hj][r] >= 0.5;      // B:711 / B:778
                    if (q.mask_out != nullptr) {                    // compact hand-off: raw predictor + mask byte
                        const size_t o = ((size_t)b * q.nrf + rf) * q.ncf + cf;
                        q.u0[o] = u0;
                        q.v0[o] = v0;
                        q.mask_out[o] = val ? 1 : 0;
                        continue;
                    }
                    double u2 = 0.0, v2 = 0.0;
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
                    const size_t o = ((size_t)b * q.nrf + rf) * q.ncf + cf;
                    q.u0[o] = u0;
                    q.v0[o] = v0;
                    q.u2[o] = u2;
                    q.v2[o] = v2;
                }
        }
    }
}

}  // namespace

hipError_t launch_predict_mfma(const BandedPredictParams& q, hipStream_t stream) {
    const int nby = (q.nrf + 31) / 32, nbx = (q.ncf + 31) / 32, nbc = (q.ncc + 31) / 32;
    if (q.KY % 8 || q.KX % 8 || q.nrfp != nby * 32 || (size_t)3 * q.ncc * q.nrfp >= (1ull << 28) ||
        (size_t)q.nrc * q.ncc >= (1ull << 28))
        return hipErrorInvalidValue;                              // 32-bit byte offsets inside a pair
    // slices of the walked tile index: about two workgroups per CU in all, each wavefront several tiles
    auto slices = [](int walked, int others) {
        int n = 512 / (others > 0 ? others : 1);
        return n < 1 ? 1 : (n > walked ? walked : n);
    };
    const int gy = (nby + 3) / 4, gx = (nbx + 3) / 4;
    hipLaunchKernelGGL(predict_rows_mfma_kernel, dim3(slices(nbc, gy * q.batch), gy, q.batch), dim3(256), 0, stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(predict_cols_mfma_kernel, dim3(gx, slices(nby, gx * q.batch), q.batch), dim3(256), 0, stream, q);
    return hipGetLastError();
}

}  // namespace tpiv
