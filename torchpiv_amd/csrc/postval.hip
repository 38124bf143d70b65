// Post-validation of the final fields on the device, batched over pairs.
//
// Reference (PIVbackend.py:884-892, per pair, on the host):  u[val] = v[val] = NaN;
// interpolate_boarders (B:328-344): 1-D np.interp along the first / last row and the first / last
// column, in that order, an all-NaN edge is left alone;  fillMissingValues (B:284-308):
// ring = valid cells 4-adjacent to a NaN cell (getPixelsForInterp B:266-282: OpenCV 3x3
// MORPH_ELLIPSE = the cross, zero border); if 2 * #ring >= size / 2 the pair is dropped ("to many
// false vectors"); otherwise scipy's LinearNDInterpolator over the Delaunay triangulation (Qhull) of
// the ring points fills the NaN cells, and ANY exception drops the pair -- in particular "no NaN
// cell at all" (zero points), the reference's dropped-clean-pair quirk.
//
// What runs here, for the whole batch at once:
//   postval_border_kernel    the four border interpolations with np.interp's arithmetic
//                            (slope * (x - xp[j]) + fp[j], end values outside the valid range) and the
//                            reference's edge order (later edges see the corners filled by earlier ones);
//   postval_classify_kernel  ring / hole census per pair (-> both drop decisions need only these counts,
//                            so dropped pairs never leave the device) and the hole fill WHERE THE
//                            DELAUNAY-LINEAR VALUE DOES NOT DEPEND ON THE TRIANGULATION:
//
//     a hole cell whose N and S neighbours are ring points while E and W are not BOTH ring points lies
//     on the segment N-S, and that segment is an edge of EVERY Delaunay triangulation of the ring
//     (the circle through N and S centred a quarter cell towards the missing side contains no other
//     lattice point), so the interpolant there is (N + S) / 2; likewise (E + W) / 2.  This covers every
//     straight run of invalid vectors.
//
//     A hole with all four neighbours valid (an isolated invalid vector) is the centre of a
//     CO-CIRCULAR diamond: the triangulation may take either diagonal, the value is (N+S)/2 or
//     (E+W)/2, and Qhull decides by the order in which it happened to insert the four vertices
//     ('Qt' fans a non-simplicial facet from its highest vertex id) -- a function of the whole point
//     set.  Such cells (class AMBIGUOUS) are only counted here: pairs that contain any go to the host
//     triangulation (torchpiv_amd/backend.py), and the fallbacks are counted.
//
//   postval_rules_kernel     (round 4) cells of wider holes (class GENERAL: corners of L / T / S shaped groups of
//                            invalid vectors, ends of thick bars ...): a triangle of ring points that contains the
//                            cell and whose CLOSED circumdisc holds no other ring point belongs to EVERY Delaunay
//                            triangulation (strict empty-circle property), so the cell's value is the barycentric
//                            sum over that triangle whatever Qhull does elsewhere.  postval_rules.inc lists the 40
//                            such triangles with circumradius^2 <= 5/2 (generated and checked against SciPy by
//                            tools/research/fill_rules.py): three vertex offsets, the weights, and the other lattice
//                            cells of the disc, none of which may be a ring cell.  Cells no rule covers stay GENERAL.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "piv_kernels.h"

namespace tpiv {

namespace {

// cell classes in PostvalParams::cls (one byte per cell); "hole" <=> 1 <= c <= 4
enum : uint8_t { PV_VALID = 0, PV_HOLE = 1, PV_FILLED = 2, PV_AMBIGUOUS = 3, PV_GENERAL = 4, PV_RING = 5 };

__device__ __forceinline__ bool is_hole(uint8_t c) { return c >= PV_HOLE && c <= PV_GENERAL; }

// one edge of one field pair: index i of the edge -> flat cell offset base + i * stride
__device__ void border_edge(double* __restrict__ u, double* __restrict__ v, uint8_t* __restrict__ cls, int base,
                            int stride, int L) {
#pragma clang fp contract(off)
    // all-NaN edge: left alone (B:332, 335, 338, 341)
    int any_valid = 0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) any_valid |= !is_hole(cls[base + i * stride]);
    any_valid = __syncthreads_or(any_valid);
    if (!any_valid) return;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        if (!is_hole(cls[base + i * stride])) continue;
        int l = i - 1, r = i + 1;
        while (l >= 0 && is_hole(cls[base + l * stride])) --l;
        while (r < L && is_hole(cls[base + r * stride])) ++r;
        double* const f[2] = {u, v};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double val;
            if (l < 0) val = f[q][base + r * stride];                  // np.interp: left of xp[0] -> fp[0]
            else if (r >= L) val = f[q][base + l * stride];            //            right of xp[-1] -> fp[-1]
            else {
                const double fl = f[q][base + l * stride], fr = f[q][base + r * stride];
                const double slope = (fr - fl) / ((double)r - (double)l);
                val = slope * ((double)i - (double)l) + fl;
                if (val != val) val = slope * ((double)i - (double)r) + fr;      // numpy: NaN one way -> try the other
            }
            f[q][base + i * stride] = val;
        }
    }
    __syncthreads();          // every scan of this edge is done before its cells turn valid
    for (int i = threadIdx.x; i < L; i += blockDim.x)
        if (is_hole(cls[base + i * stride])) cls[base + i * stride] = PV_VALID;
    __syncthreads();
}

__global__ __launch_bounds__(256) void postval_border_kernel(PostvalParams p) {
    const size_t off = (size_t)blockIdx.x * p.n_rows * p.n_cols;
    double* u = p.u + off;
    double* v = p.v + off;
    uint8_t* cls = p.cls + off;
    const uint8_t* inv = p.invalid + off;
    const int n = p.n_rows * p.n_cols;
    int any = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint8_t h = inv[i] ? PV_HOLE : PV_VALID;
        cls[i] = h;
        any |= h;
    }
    any = __syncthreads_or(any);
    if (!any) return;                                            // B:329-330
    border_edge(u, v, cls, 0, 1, p.n_cols);                                   // first row
    border_edge(u, v, cls, (p.n_rows - 1) * p.n_cols, 1, p.n_cols);           // last row
    border_edge(u, v, cls, 0, p.n_cols, p.n_rows);                            // first column
    border_edge(u, v, cls, p.n_cols - 1, p.n_cols, p.n_rows);                 // last column
}

// grid (blocks per pair, batch): a block never spans two pairs, so the census needs one atomic per wave
__global__ __launch_bounds__(256) void postval_classify_kernel(PostvalParams p) {
    const int pair = blockIdx.y;
    const size_t off = (size_t)pair * p.n_rows * p.n_cols;
    double* __restrict__ u = p.u + off;
    double* __restrict__ v = p.v + off;
    uint8_t* cls = p.cls + off;
    const int nr = p.n_rows, nc = p.n_cols, n = nr * nc;
    int n_hole = 0, n_ring = 0, n_amb = 0, n_gen = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / nc, c = i - r * nc;
        // neighbour is a usable (valid) cell?  out of the field: no (zero border, B:278-279)
        const bool hN = r > 0 && is_hole(cls[i - nc]), hS = r < nr - 1 && is_hole(cls[i + nc]);
        const bool hW = c > 0 && is_hole(cls[i - 1]), hE = c < nc - 1 && is_hole(cls[i + 1]);
        if (!is_hole(cls[i])) {
            if (hN || hS || hW || hE) {
                cls[i] = PV_RING;
                ++n_ring;
            }
            continue;
        }
        ++n_hole;
        const bool vN = r > 0 && !hN, vS = r < nr - 1 && !hS, vW = c > 0 && !hW, vE = c < nc - 1 && !hE;
        if (vN && vS && vW && vE) {
            cls[i] = PV_AMBIGUOUS;
            ++n_amb;
        } else if (vN && vS) {
            u[i] = (u[i - nc] + u[i + nc]) * 0.5;
            v[i] = (v[i - nc] + v[i + nc]) * 0.5;
            cls[i] = PV_FILLED;
        } else if (vW && vE) {
            u[i] = (u[i - 1] + u[i + 1]) * 0.5;
            v[i] = (v[i - 1] + v[i + 1]) * 0.5;
            cls[i] = PV_FILLED;
        } else {
            cls[i] = PV_GENERAL;
            ++n_gen;
        }
    }
    int vals[4] = {n_hole, n_ring, n_amb, n_gen};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int s = vals[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&p.counts[pair * 4 + q], s);
    }
}

#include "postval_rules.inc"

__global__ __launch_bounds__(256) void postval_rules_kernel(PostvalParams p) {
    const int pair = blockIdx.y;
    const size_t off = (size_t)pair * p.n_rows * p.n_cols;
    double* __restrict__ u = p.u + off;
    double* __restrict__ v = p.v + off;
    uint8_t* cls = p.cls + off;
    const int nr = p.n_rows, nc = p.n_cols, n = nr * nc;
    int n_filled = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (cls[i] != PV_GENERAL) continue;
        const int r = i / nc, c = i - r * nc;
        // (ring cells are valid cells: this kernel neither changes them nor their values; it only turns GENERAL into
        //  FILLED, both of which are "hole" to every test below)
        auto ring = [&](int dr, int dc) {
            const int rr = r + dr, cc = c + dc;
            return rr >= 0 && rr < nr && cc >= 0 && cc < nc && cls[rr * nc + cc] == PV_RING;
        };
        for (int k = 0; k < PV_N_RULES; ++k) {
            const PvRule& R = PV_RULES[k];
            bool ok = ring(R.v[0][0], R.v[0][1]) && ring(R.v[1][0], R.v[1][1]) && ring(R.v[2][0], R.v[2][1]);
            for (int b = 0; ok && b < R.nb; ++b) ok = !ring(R.b[b][0], R.b[b][1]);
            if (!ok) continue;
            const int i0 = i + R.v[0][0] * nc + R.v[0][1], i1 = i + R.v[1][0] * nc + R.v[1][1], i2 = i + R.v[2][0] * nc + R.v[2][1];
            u[i] = R.w[0] * u[i0] + R.w[1] * u[i1] + R.w[2] * u[i2];
            v[i] = R.w[0] * v[i0] + R.w[1] * v[i1] + R.w[2] * v[i2];
            cls[i] = PV_FILLED;
            ++n_filled;
            break;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_filled += __shfl_xor(n_filled, o, 64);
    if ((threadIdx.x & 63) == 0 && n_filled) atomicSub(&p.counts[pair * 4 + 3], n_filled);
}

// ---- hand-over to the host triangulation: what fillMissingValues (B:284-308) feeds to the interpolator, cut out of
// the batch on the device.  A pair "needs the host" when it is kept (ring > 0, 4 ring < cells) and holds an ambiguous
// or general hole; for those pairs the ring cells (B:298 np.argwhere(neighbours): ROW-MAJOR order -- Qhull's result
// depends on the insertion order, so the compaction is order-preserving) with their (u, v) and all hole cells
// (np.argwhere(invalid_mask), same order) are packed pair after pair into three flat lists.
__global__ void postval_offsets_kernel(const int* __restrict__ counts, int batch, int cells, int* __restrict__ offsets) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int r = 0, h = 0;
    for (int i = 0; i < batch; ++i) {
        const int holes = counts[4 * i], ring = counts[4 * i + 1], amb = counts[4 * i + 2], gen = counts[4 * i + 3];
        const bool need = ring > 0 && 4LL * ring < (long long)cells && (amb + gen) > 0;
        offsets[i] = r;
        offsets[batch + 1 + i] = h;
        r += need ? ring : 0;
        h += need ? holes : 0;
    }
    offsets[batch] = r;
    offsets[2 * batch + 1] = h;
}

__global__ __launch_bounds__(256) void postval_compact_kernel(const double* __restrict__ U, const double* __restrict__ V,
                                                              const uint8_t* __restrict__ CLS, const int* __restrict__ offsets,
                                                              int batch, int n_rows, int n_cols, int* __restrict__ ring_rc,
                                                              double* __restrict__ ring_uv, int* __restrict__ hole_rc) {
    const int pair = blockIdx.x;
    const int r0 = offsets[pair], r1 = offsets[pair + 1];
    const int h0 = offsets[batch + 1 + pair];
    if (r1 == r0) return;                                  // dropped or complete on the device: nothing to hand over
    const size_t off = (size_t)pair * n_rows * n_cols;
    const double* u = U + off;
    const double* v = V + off;
    const uint8_t* cls = CLS + off;
    const int n = n_rows * n_cols;
    __shared__ int wsum[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int done_r = 0, done_h = 0;                            // cells of both kinds in the chunks before this one
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        const uint8_t c = i < n ? cls[i] : (uint8_t)PV_VALID;
        const bool is_r = c == PV_RING, is_h = is_hole(c);
        const unsigned long long br = __ballot(is_r), bh = __ballot(is_h);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (lane == 0) {
            wsum[0][wave] = __popcll(br);
            wsum[1][wave] = __popcll(bh);
        }
        __syncthreads();
        int pr = done_r + __popcll(br & below), ph = done_h + __popcll(bh & below);
        int tr = 0, th = 0;
#pragma unroll
        for (int w_ = 0; w_ < 4; ++w_) {
            pr += w_ < wave ? wsum[0][w_] : 0;
            ph += w_ < wave ? wsum[1][w_] : 0;
            tr += wsum[0][w_];
            th += wsum[1][w_];
        }
        if (i < n) {
            const int r = i / n_cols, cc = i - r * n_cols;
            if (is_r) {
                ring_rc[2 * (size_t)(r0 + pr)] = r;
                ring_rc[2 * (size_t)(r0 + pr) + 1] = cc;
                ring_uv[2 * (size_t)(r0 + pr)] = u[i];
                ring_uv[2 * (size_t)(r0 + pr) + 1] = v[i];
            }
            if (is_h) {
                hole_rc[2 * (size_t)(h0 + ph)] = r;
                hole_rc[2 * (size_t)(h0 + ph) + 1] = cc;
            }
        }
        done_r += tr;
        done_h += th;
        __syncthreads();                                   // wsum is rewritten by the next chunk
    }
}

// ---- B:894-898 for a batch: u = flip(u, rows) * scale / dt * 1000, v = -flip(v, rows) * scale / dt * 1000 -- the
// reference's float64 expression evaluated left to right, three correctly rounded operations per value (no contraction:
// numpy rounds every one of them), so the result is bit-identical to the host's.
__global__ __launch_bounds__(256) void finish_fields_kernel(const double* __restrict__ U, const double* __restrict__ V,
                                                            int batch, int n_rows, int n_cols, double scale, double dt,
                                                            double* __restrict__ FU, double* __restrict__ FV) {
#pragma clang fp contract(off)
    const long long cells = (long long)n_rows * n_cols;
    const long long total = cells * batch;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pair = i / cells;
        const int rc = (int)(i - pair * cells);
        const int r = rc / n_cols, c = rc - r * n_cols;
        const long long src = pair * cells + (long long)(n_rows - 1 - r) * n_cols + c;
        const double u = U[src], v = -V[src];
        FU[i] = u * scale / dt * 1000.0;
        FV[i] = v * scale / dt * 1000.0;
    }
}

// ---- ensemble moments (workers.py:85-96): one thread per grid cell walks the stack of fields IN ORDER,
// like numpy's reduction along the stack axis: mean = ((f0 + f1) + f2 ...) / n, then the two-pass
// central moments sum_k (f_k - mean)^2 / n.  No contraction: numpy rounds the product and the sum apart.
__global__ __launch_bounds__(256) void ensemble_moments_kernel(const double* __restrict__ U, const double* __restrict__ V,
                                                               int n, long long cells, double* __restrict__ out) {
#pragma clang fp contract(off)
    const long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (c >= cells) return;
    double su = 0.0, sv = 0.0;
    for (int k = 0; k < n; ++k) {
        su = su + U[(long long)k * cells + c];
        sv = sv + V[(long long)k * cells + c];
    }
    const double mu = su / (double)n, mv = sv / (double)n;
    double suu = 0.0, svv = 0.0, suv = 0.0;
    for (int k = 0; k < n; ++k) {
        const double du = U[(long long)k * cells + c] - mu, dv = V[(long long)k * cells + c] - mv;
        const double a = du * du, b = dv * dv, ab = du * dv;
        suu = suu + a;
        svv = svv + b;
        suv = suv + ab;
    }
    out[c] = mu;
    out[cells + c] = mv;
    out[2 * cells + c] = suu / (double)n;
    out[3 * cells + c] = svv / (double)n;
    out[4 * cells + c] = suv / (double)n;
}

}  // namespace

hipError_t launch_ensemble_moments(const double* U, const double* V, int n, long long cells, double* out, hipStream_t stream) {
    if (n <= 0 || cells <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ensemble_moments_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, U, V, n, cells, out);
    return hipGetLastError();
}

hipError_t launch_finish_fields(const double* u, const double* v, int batch, int n_rows, int n_cols, double scale, double dt,
                                double* fu, double* fv, hipStream_t stream) {
    const long long total = (long long)batch * n_rows * n_cols;
    if (total <= 0) return hipErrorInvalidValue;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(finish_fields_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, u, v, batch, n_rows, n_cols, scale, dt,
                       fu, fv);
    return hipGetLastError();
}

hipError_t launch_postval_compact(const double* u, const double* v, const uint8_t* cls, const int* counts, int batch, int n_rows,
                                  int n_cols, int* offsets, int* ring_rc, double* ring_uv, int* hole_rc, hipStream_t stream) {
    hipLaunchKernelGGL(postval_offsets_kernel, dim3(1), dim3(64), 0, stream, counts, batch, n_rows * n_cols, offsets);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(postval_compact_kernel, dim3(batch), dim3(256), 0, stream, u, v, cls, offsets, batch, n_rows, n_cols,
                       ring_rc, ring_uv, hole_rc);
    return hipGetLastError();
}

hipError_t launch_postval(const PostvalParams& p, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(p.counts, 0, (size_t)p.batch * 4 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(postval_border_kernel, dim3(p.batch), dim3(256), 0, stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = p.n_rows * p.n_cols;
    int bpp = (n + 256 * 8 - 1) / (256 * 8);          // ~8 cells per thread
    if (bpp < 1) bpp = 1;
    if (bpp > 256) bpp = 256;
    hipLaunchKernelGGL(postval_classify_kernel, dim3(bpp, p.batch), dim3(256), 0, stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(postval_rules_kernel, dim3(bpp, p.batch), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace tpiv
