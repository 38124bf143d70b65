/* torchpiv_hip.h -- C ABI of the MI355X-native PIV cross-correlation engine.
 *
 * The reference (NikNazarov/TorchPIV) has no FFI layer: its boundary is the Python
 * class OfflinePIV (src/torchPIV/PIVbackend.py:824-903) calling pure functions on
 * torch tensors.  This library replaces the device part of those functions; the
 * Python host (torchpiv_amd/backend.py) binds it with ctypes and keeps the
 * reference's class / function signatures.  Each entry point cites the reference
 * code it replaces (B: = src/torchPIV/PIVbackend.py).
 *
 * Conventions
 *   - every pointer named *_dev is device memory on the CURRENT HIP device, owned by
 *     the caller (PyTorch-ROCm tensors: tensor.data_ptr()); the run functions only enqueue
 *     work on `stream` (a hipStream_t passed as void*; NULL = the null stream) and return at
 *     once.  No entry point allocates device memory or synchronises: a plan owns its workspace
 *     (allocated once by tpiv_plan_create), and the function-level entry points (tpiv_pass1 /
 *     tpiv_iter) take a caller-provided work buffer of tpiv_work_bytes() bytes (hand-off records
 *     between the tile kernel and the finalize kernel).  The library keeps no state between
 *     calls, so calls on different streams (with different work buffers) may overlap;
 *   - fields are row-major [batch, n_rows, n_cols]; frames are uint8 [batch, H, W];
 *   - return value: TPIV_OK or an error code; tpiv_last_error() gives the message
 *     of the calling thread's last failure.  Nothing is thrown across the ABI.
 */
#ifndef TORCHPIV_HIP_H
#define TORCHPIV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPIV_VERSION 2

enum tpiv_status {
    TPIV_OK = 0,
    TPIV_EINVAL = 1,  /* bad window / overlap / shape: the reference raises ValueError (B:503-507) */
    TPIV_EKEY = 2,    /* unknown multipass mode: the reference raises KeyError (B:850) */
    TPIV_EHIP = 3,    /* a HIP runtime call failed */
    TPIV_ENOMEM = 4,
    TPIV_EUNSUPPORTED = 5 /* valid for the reference but outside what the kernels cover */
};

enum tpiv_mode {
    TPIV_MODE_DWS = 1, /* discrete window shift,  piv_iteration_DWS  B:744-812 */
    TPIV_MODE_CWS = 2, /* continuous window shift, piv_iteration_CWS B:677-740 */
    TPIV_MODE_CWS_FAST = 3 /* piv_iteration_CWS_Fast B:599-675 (bicubic grid_sample of every window inside itself,
                              u = u0 + du); the reference's OfflinePIV cannot reach it (absent from IterModMap):
                              tpiv_iter only, generic-size kernel; u2 / v2 unused (may be NULL), u0 / v0 are the
                              predictor AFTER the invalid-zeroing (tpiv_predict's u0 / v0) */
};

enum tpiv_precision {
    TPIV_PREC_FAST = 0,      /* pass 1 in float32 (observed deviation ~1e-6 px); shifted passes form the CWS
                                sample as row lerps + a column lerp with the reference's float32 weights
                                (float32 rounding differences against B:187-193) */
    TPIV_PREC_REFERENCE = 1, /* pass 1 in float64 like the reference (B:513-514 promotes the windows to
                                float64 before the FFT); shifted passes evaluate B:187-193 operation by
                                operation, so the staged windows are bit-identical to the reference's
                                (the transforms of passes >= 2 are float32 in the reference itself,
                                B:249-257, with a float64 epilogue, B:382) */
    TPIV_PREC_F64 = 2,       /* the reference's ARITHMETIC TYPES in every pass -- pass 1 in float64 (as
                                TPIV_PREC_REFERENCE), shifted passes in float32 with the float64 epilogue -- with
                                the cheaper operation order of TPIV_PREC_FAST in the shifted passes (row lerps +
                                column lerp of the CWS sample: float32 rounding differences, <= 1e-4 grey levels,
                                against B:187-193). */
    TPIV_PREC_EXACT = 3      /* as TPIV_PREC_F64, with the map cells that reach the result of pass 1 (arg-max, its
                                neighbours, second peak, minimum: B:383-411, B:518) evaluated as EXACT integer
                                correlation sums of the uint8 windows instead of through a float64 FFT: a float32
                                FFT pass locates the cells inside an error band (the proven bound on its rounding
                                error), windows it cannot decide run the float64 transform
                                (xcorr_exact.hip).  Every even first-pass window size from 8 to 128; other sizes run
                                as TPIV_PREC_F64.  Within ~1e-14 px of the reference's float64
                                pass 1 (whose own transform rounding is the difference).  The default of the Python
                                drop-in (OfflinePIV). */
};

typedef struct tpiv_plan tpiv_plan;

int tpiv_version(void);
const char* tpiv_last_error(void);

/* ---- host-side geometry (no GPU needed) -------------------------------------- */

/* get_field_shape, B:425-456: (size - ws)//(ws - ov) + 1 per axis. */
int tpiv_field_shape(int H, int W, int ws, int ov, int* n_rows, int* n_cols);

/* get_coordinates, B:522-597: window-centre coordinates along each axis
 * (x: n_cols values, y: n_rows values); the reference returns their meshgrid. */
int tpiv_coordinates(int H, int W, int ws, int ov, double* x, double* y);

/* The predictor operator of scipy.interpolate.RectBivariateSpline(kx=ky=3, s=0)
 * as the reference calls it (B:700-704, 710-711, 769-773, 777-778): row-major
 * A[nf, nc] such that fine = A_y * coarse * A_x^T.  1-D not-a-knot cubic spline
 * interpolation from the nc coarse coordinates xc to the nf fine coordinates xf,
 * evaluation points clamped to [xc[0], xc[nc-1]] (FITPACK does not extrapolate).
 * Needs nc >= 4 (as scipy does). */
int tpiv_spline_matrix(int nc, const double* xc, int nf, const double* xf, double* A);

/* ---- function-level seam (device pointers) ---------------------------------- */

/* Tensor part of extended_search_area_piv(frame_a, frame_b, window_size, overlap,
 * validate=True, validation_ratio), B:459-520 (+ correalte_fft B:249-257,
 * correlation_to_displacement B:360-422, peak2peak_secondpeak B:346-358).
 * Outputs: u, v float64 and invalid uint8 (1 = peak ratio < val_ratio). */
int tpiv_pass1(const uint8_t* a_dev, const uint8_t* b_dev, int batch, int H, int W,
               int ws, int ov, double val_ratio, int val_win, int precision,
               double* u_dev, double* v_dev, uint8_t* invalid_dev,
               void* work_dev, size_t work_bytes, void* stream);

/* Bytes of device work buffer tpiv_pass1 / tpiv_iter / tpiv_debug_pass need for `batch` pairs of
 * H x W frames at (ws, ov) (either precision); 0 if the geometry is invalid. */
size_t tpiv_work_bytes(int H, int W, int ws, int ov, int batch);

/* Predictor of one multipass iteration (host-side scipy calls in the reference,
 * B:700-717 CWS / B:769-790 DWS): spline-upsample u, v and the invalid mask from
 * the coarse grid [nrc, ncc] to the fine grid [nrf, ncf] with the operators of
 * tpiv_spline_matrix (Ay [nrf, nrc], Ax [ncf, ncc], device memory), threshold the
 * mask at 0.5, zero the predictor where invalid and form the window half-shift
 * (CWS: u0/2 taken before the zeroing; DWS: rint(u0/2) after it).
 * work_dev: batch*3*nrc*ncf float64 of scratch. */
int tpiv_predict(int mode, int batch, int nrc, int ncc, int nrf, int ncf,
                 const double* Ay_dev, const double* Ax_dev,
                 const double* u_c_dev, const double* v_c_dev, const uint8_t* invalid_c_dev,
                 double* work_dev, double* u0_dev, double* v0_dev, double* u2_dev, double* v2_dev,
                 void* stream);

/* Tensor part of piv_iteration_DWS.__call__ (B:791-810, interpolation_DWS B:197-216)
 * or piv_iteration_CWS.__call__ (B:719-738, biliniar_interpolation_CWS B:147-194):
 * shift the windows of frame a by -(u2, v2) and of frame b by +(u2, v2), correlate,
 * find the peak, validate and combine with the predictor
 * (u = 2*u2 + du, fallback to u0 where (du > u0 and rint(u0) > 0) or invalid).
 * du_dev / dv_dev (optional, may be NULL) receive the raw displacement of this pass. */
int tpiv_iter(int mode, const uint8_t* a_dev, const uint8_t* b_dev, int batch, int H, int W,
              int ws, int ov,
              const double* u0_dev, const double* v0_dev, const double* u2_dev, const double* v2_dev,
              double val_ratio, int val_win, int precision,
              double* u_dev, double* v_dev, uint8_t* invalid_dev,
              double* du_dev, double* dv_dev, void* work_dev, size_t work_bytes, void* stream);

/* ---- plan: the whole multipass pipeline of OfflinePIV.__call__ for a batch ---- */

/* Mirrors OfflinePIV.__init__ (B:825-858): pass p > 0 uses ws_p = int(ws_{p-1} // pass_scale),
 * ov_p = int(ov_{p-1} // pass_scale).  Allocates (on the current device) the spline
 * operators and the per-pass field workspace for up to max_batch pairs.
 * precision: enum tpiv_precision (arithmetic of pass 1). */
int tpiv_plan_create(tpiv_plan** out, int H, int W, int ws, int ov, int n_pass, int mode,
                     double pass_scale, double val_ratio, int val_win, int max_batch, int precision);
void tpiv_plan_destroy(tpiv_plan* plan);
int tpiv_plan_n_pass(const tpiv_plan* plan);
int tpiv_plan_pass_geometry(const tpiv_plan* plan, int pass, int* ws, int* ov, int* n_rows, int* n_cols);

/* Name of the cross-correlation kernel pass `pass` launches (for bench / profile labels), e.g.
 * "xcorr_tile_kernel<32, 2, 3, true>" (the demangled form profilers print; second argument: 0 pass 1, 1 DWS, 2 CWS; last: fast arithmetic).  Returns buf. */
const char* tpiv_plan_kernel_name(const tpiv_plan* plan, int pass, char* buf, int len);

/* Pass 1 and every further pass (B:873-882) for `batch` <= max_batch pairs; the last
 * pass writes straight into u_dev / v_dev / invalid_dev ([batch, n_rows_last, n_cols_last]). */
int tpiv_plan_run(tpiv_plan* plan, const uint8_t* a_dev, const uint8_t* b_dev, int batch,
                  double* u_dev, double* v_dev, uint8_t* invalid_dev, void* stream);

/* Device pointers to the fields pass `pass` (< n_pass - 1) left in the plan's workspace
 * during the last run (for parity tests of the intermediate passes). */
int tpiv_plan_pass_fields(const tpiv_plan* plan, int pass, double** u_dev, double** v_dev,
                          uint8_t** invalid_dev);

/* ---- post-validation (B:884-892) ------------------------------------------------- */

/* Device part of the reference's per-pair host post-processing, for a whole batch:
 *   u[val] = v[val] = NaN (B:885-886; `invalid_dev` plays the NaN mask, u/v are not overwritten with NaN),
 *   interpolate_boarders (B:328-344) on u and v in place,
 *   the ring / hole census of getPixelsForInterp (B:266-282) -> counts_dev [batch, 4] int32 =
 *   {holes, ring cells, ambiguous holes, general holes}.  Both drop decisions of fillMissingValues
 *   (B:284-308) follow from the counts: ring == 0 (nothing to interpolate from: the interpolator raises ->
 *   pair dropped, including the "no invalid vector" quirk) and 4 * ring >= n_rows * n_cols ("to many false
 *   vectors").
 *   Holes whose Delaunay-linear value does not depend on the triangulation (both N and S, or both E and W,
 *   neighbours valid, but not all four) are filled in place: (N + S) / 2 resp. (E + W) / 2.  Holes with all
 *   four neighbours valid (co-circular diamond: Qhull's tie-break decides between the two) and holes inside
 *   wider gaps are only classified; a pair with ambiguous + general > 0 needs the host triangulation.
 * cls_dev [batch, n_rows, n_cols] uint8: 0 valid, 2 hole filled here, 3 ambiguous hole, 4 general hole,
 * 5 ring cell.  Needs n_rows, n_cols >= 2. */
int tpiv_postval(double* u_dev, double* v_dev, const uint8_t* invalid_dev, int batch, int n_rows, int n_cols,
                 uint8_t* cls_dev, int32_t* counts_dev, void* stream);

/* What fillMissingValues (B:296-302) hands to the interpolator, cut out of a batch on the device after tpiv_postval: for
 * every pair that is kept (ring > 0, 4 ring < cells) AND holds an ambiguous or general hole, the ring cells
 * (np.argwhere(neighbours), B:298: row-major order -- Qhull's triangulation depends on the insertion order, so the
 * compaction preserves it) with their values, and all hole cells (np.argwhere(invalid_mask), B:297), packed pair after
 * pair into flat lists:
 *   offsets_dev [2, batch + 1] int32: row 0 = start of every pair's ring cells (last entry: total), row 1 = holes;
 *   ring_rc_dev [total ring, 2] int32 (row, column), ring_uv_dev [total ring, 2] float64 (u, v), hole_rc_dev
 *   [total holes, 2] int32.  The caller sizes ring_* for batch * ceil(cells / 4) entries and hole_rc for batch * cells. */
int tpiv_postval_compact(const double* u_dev, const double* v_dev, const uint8_t* cls_dev, const int32_t* counts_dev,
                         int batch, int n_rows, int n_cols, int32_t* offsets_dev, int32_t* ring_rc_dev,
                         double* ring_uv_dev, int32_t* hole_rc_dev, void* stream);

/* B:894-898 for a batch of final fields [batch, n_rows, n_cols]: fu = flip(u, axis 0) * scale / dt * 1000,
 * fv = -flip(v, axis 0) * scale / dt * 1000 -- the reference's float64 expression, left to right, three correctly
 * rounded operations per value: bit-identical to numpy's.  (x, y are not flipped, B:899-900.) */
int tpiv_finish_fields(const double* u_dev, const double* v_dev, int batch, int n_rows, int n_cols, double scale,
                       double dt, double* fu_dev, double* fv_dev, void* stream);

/* ---- ensemble statistics (workers.py:85-96 of the reference's job runner) ------------ */

/* Mean and two-pass central moments of n stacked fields u_dev, v_dev [n, cells] float64 (dataset
 * order): out_dev [5, cells] = mean(u), mean(v), mean((u-U)^2), mean((v-V)^2), mean((u-U)(v-V)),
 * accumulated along the stack IN ORDER like numpy's np.mean(axis=0) -- bit-identical to the reference's
 * np.mean(u_inst, axis=0), np.mean((u_inst - avg_u)**2, axis=0), ... */
int tpiv_ensemble_moments(const double* u_dev, const double* v_dev, int n, long long cells, double* out_dev,
                          void* stream);

/* ---- image ingest (PIVDataset.__getitem__, B:129-144) ------------------------------- */

/* Unpacks n_files uncompressed BMP files that were uploaded as RAW FILE BYTES into uint8 frames
 * out_dev [n_files, H, W] (top-down rows, what cv2.imdecode(..., IMREAD_GRAYSCALE) returns): header
 * skip, bottom-up row flip, row-padding strip, palette look-up (1 byte per pixel) or OpenCV's
 * fixed-point BGR -> gray weights (3 / 4 bytes per pixel).  The host parses the 54-byte headers
 * (torchpiv_amd/io.py) and passes, per file, desc_dev[f][6] int64 = {offset of the file in raw_dev,
 * offset of the pixel data in the file, row stride in bytes, bytes per pixel (1, 3, 4), rows stored
 * bottom-up (0/1), reserved} and lut_dev[f][256] (palette entries already converted to gray). */
int tpiv_bmp_unpack(const uint8_t* raw_dev, const int64_t* desc_dev, const uint8_t* lut_dev, int n_files,
                    int H, int W, uint8_t* out_dev, void* stream);

/* Host side of the ingest (no GPU involved): reads n_files files into dst + i * slot_bytes (page-locked staging memory
 * of the caller, at most slot_bytes each) with up to n_threads native reader threads -- what PIVDataset.__getitem__
 * (B:129-144) does file by file with np.fromfile, here for a whole batch without the interpreter in the loop.
 * sizes[i] = bytes read, or -1 when the file cannot be opened / read or does not fit the slot (the caller then takes
 * its per-file path, which skips an undecodable pair like B:138-139).  Returns TPIV_OK (per-file failures are not errors). */
int tpiv_read_files(const char* const* paths, int n_files, uint8_t* dst, size_t slot_bytes, int n_threads,
                    int64_t* sizes);

/* Read-ahead form of the same for a whole run (the loader of OfflinePIV.batched): the file list is handed over once;
 * n_threads reader threads fill the caller's n_bufs page-locked staging buffers batch after batch -- batch k (files
 * [k * files_per_batch, (k + 1) * files_per_batch), file j of it at bufs[k % n_bufs] + j * slot_bytes) -- running at most
 * n_bufs batches ahead of the consumer.  tpiv_reader_next blocks until the next batch is complete and gives its number
 * of files in *n_files (0: end of the list) with *buf_index and sizes[j] (bytes, or -1 as for tpiv_read_files);
 * tpiv_reader_release hands the OLDEST outstanding batch's buffer back for refilling (call it once the upload of that
 * buffer is through); tpiv_reader_close stops the threads (also mid-run) and frees the handle. */
typedef struct tpiv_reader tpiv_reader;
tpiv_reader* tpiv_reader_open(const char* const* paths, int64_t n_files, int files_per_batch, uint8_t* const* bufs,
                              int n_bufs, size_t slot_bytes, int n_threads);
int tpiv_reader_next(tpiv_reader* reader, int* n_files, int* buf_index, int64_t* sizes);
int tpiv_reader_release(tpiv_reader* reader);
void tpiv_reader_close(tpiv_reader* reader);

/* ---- measurement ------------------------------------------------------------------ */

/* Per-kernel timing with hipEvents recorded on the run's own stream (torch.cuda.Event only
 * sees torch's current stream).  While enabled, every tpiv_plan_run brackets each launch
 * with an event pair (no host synchronisation; up to 512 runs are kept).  Slots:
 * 0 = pass-1 tile kernel; for pass p >= 1: 2p-1 = predictor kernels, 2p = tile kernel. */
int tpiv_plan_set_timing(tpiv_plan* plan, int enable);
/* Waits for the recorded events, writes the MEAN duration in milliseconds of every slot over
 * the runs recorded since the last call into avg_ms[0..n_slots) and the number of runs into
 * n_runs, then clears the record.  n_slots must be 2*n_pass - 1. */
int tpiv_plan_get_timing(tpiv_plan* plan, double* avg_ms, int n_slots, int* n_runs);
/* TPIV_PREC_EXACT plans with an even first-pass window size from 8 to 128: the number of windows of the LAST tpiv_plan_run whose first pass
 * went through the float64 transform (undecided by the float32 locating pass).  Waits for the device.  TPIV_EINVAL for
 * other plans or before the first run.  (Diagnostics: bench.py reports the share.) */
int tpiv_plan_exact_fallbacks(tpiv_plan* plan, long long* n_windows);
/* The same plans: slot 0 of tpiv_plan_get_timing taken apart, as of the last call of
 * that function: ms4 = mean milliseconds of {float32 locating pass, exact refinement, float64 pass of the undecided
 * windows, finalize}. */
int tpiv_plan_exact_timing(const tpiv_plan* plan, double* ms4);

/* ---- test hook ------------------------------------------------------------------ */

/* Runs one pass like tpiv_pass1 (mode 0, float32) / tpiv_iter (mode DWS/CWS at `precision`; zero_dev =
 * [batch, n_rows, n_cols] float64 zeros, used as u0 = v0) and additionally writes the staged
 * windows win_dev [batch, N, 2, ws, ws] float32 (frame a, frame b, after the shift) and the
 * correlation maps corr_dev [batch, N, ws, ws] float32 (corr - min + 1e-7, fftshift layout).
 * Either may be NULL. */
int tpiv_debug_pass(int mode, int precision, const uint8_t* a_dev, const uint8_t* b_dev, int batch, int H, int W,
                    int ws, int ov, const double* u2_dev, const double* v2_dev, const double* zero_dev,
                    double* u_dev, double* v_dev, uint8_t* invalid_dev,
                    float* win_dev, float* corr_dev, void* work_dev, size_t work_bytes, void* stream);

/* Peak analysis alone -- correlation_to_displacement (B:360-422) + peak2peak_secondpeak
 * (B:346-358) -- on caller-supplied correlation maps [n_maps, ws, ws] float32 in fftshift layout
 * (ws = 8, 16, 32, 64 or 128): runs the kernels' peak stage and finalize on them.  planar != 0
 * selects the LDS layout of the three-wavefront tile kernels (64x64: the three-row map); planar = 2 with
 * ws = 8 the peak stage of the one-window-per-lane kernel; ignored for ws = 128.  The kernel subtracts the map minimum first (B:518), so feed maps whose minimum is
 * 0 to compare with the reference function.  work_dev: n_maps * 32 bytes. */
int tpiv_debug_peaks(const float* maps_dev, int n_maps, int ws, int planar, double val_ratio, int val_win,
                     double* u_dev, double* v_dev, uint8_t* invalid_dev,
                     void* work_dev, size_t work_bytes, void* stream);

/* Runs the plan's own (banded) predictor of pass `pass` (1 <= pass < n_pass) on caller-supplied
 * coarse fields, exactly as tpiv_plan_run does between passes; same outputs as tpiv_predict.
 * Lets the tests compare the banded operator with the dense one. */
int tpiv_plan_debug_predict(tpiv_plan* plan, int pass, int batch,
                            const double* u_c_dev, const double* v_c_dev, const uint8_t* invalid_c_dev,
                            double* u0_dev, double* v0_dev, double* u2_dev, double* v2_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TORCHPIV_HIP_H */
