/*
 * rwh.h -- C ABI of librwh_hip.so: MI355X (gfx950) kernels for the
 * RANSAC-homography + backward-warp hot path of choice17/ransac_with_homography.
 *
 * The reference is pure Python/numpy and has no FFI of its own; the boundary a
 * maintainer would bind (ctypes) is the set of numpy calls on its hot path.
 * Each entry point below names the reference lines it replaces.  All pointers
 * named `d_*` are DEVICE pointers; everything else is host memory read before
 * the function returns.  `stream` is a hipStream_t (NULL = default stream).
 * Functions enqueue work and return without synchronising; they allocate
 * nothing and keep no state, so they may be captured into a hipGraph.
 *
 * Return value: 0 on success, a negative RWH_E_* code otherwise (never throws).
 */
#ifndef RWH_H
#define RWH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RWH_ABI_VERSION 3   /* 3 (round 4): rwh_ransac_run takes hyp_base and returns packed keys, RWH_HYP_DEGENERATE, rwh_score_interval;
                               2: blend modes of rwh_stitch_panorama, RWH_BATCH_EARLY_STOP (d_counts may hold -1), rwh_lab_clock_probe, RWH_HYP_ILLCOND */
#define RWH_API __attribute__((visibility("default")))

enum {
    RWH_OK = 0,
    RWH_E_INVALID = -1,      /* bad argument (NULL pointer, size <= 0, unknown enum) */
    RWH_E_UNSUPPORTED = -2,  /* valid but not implemented combination (dtype / channels) */
    RWH_E_LAUNCH = -3        /* HIP reported a launch / memset failure */
};

/* element types of image planes */
enum { RWH_U8 = 0, RWH_F32 = 1, RWH_F64 = 2 };
/* interpolators: reference homography.py:140 `convertfunc` keys */
enum { RWH_NEAREST = 0, RWH_BILINEAR = 1 };
/* RANSAC loss: reference ransac.py:84-98 `computeLoss` method */
enum { RWH_LOSS_FWD = 0, RWH_LOSS_BACKWARD = 1, RWH_LOSS_REPROJ = 2 };

/* flags for rwh_warp_backward */
#define RWH_WARP_ZERO_ORIGIN 1u /* write zeros into texel (0,0) of every source image first, exactly
                                   like homography.py:112-116 / 126-130 do to the caller's array */

#define RWH_WARP_EXACT 2u       /* reproduce the reference's float64 arithmetic operation by operation (dgemm
                                   k-order, IEEE divides, separately rounded float64 lerps): results bit-identical to
                                   numpy's; dst_dtype F64 (bilinear), U8 (bilinear, truncated) or the source dtype
                                   (nearest).  Several times slower than the default kernels. */

RWH_API int rwh_abi_version(void);
RWH_API const char* rwh_strerror(int code);

/*
 * Lab / TEST-ONLY hook, not part of the data path and not thread-safe (plain process-wide globals): pins a launch heuristic
 * (value 0 = back to the library's own choice).  RWH_TUNE_WARP_SHAPE: log2 of the fast bilinear kernel's patch width (5, 6, 7; 13, 14 = 32 x 16 / 64 x 8 patches staged by halves, the minification form of the uint8 RGB kernel); RWH_TUNE_SCORE_HPW:
 * hypotheses per wavefront of the scorer (1..64); RWH_TUNE_SCORE_EXACT: 1 = the scorer skips its reciprocal-based
 * filter and runs the two IEEE divisions for every pair (the filter only ever decides pairs that clear the threshold
 * by a proven error band, so counts and masks are the same either way); RWH_TUNE_WARP_FRAMES (round 4): frames per block of
 * the multi-frame form of the uint8 RGB bilinear kernel (one homography, batch >= 2: warp_rgb8_fast8m, a lab kernel that shares
 * a patch's coordinate / weight arithmetic between the frames of a batch) -- 0 / 1 = one frame per block (the product kernel),
 * 2..64 = that many.  Results never depend on any of them (tests/test_gpu_parity.py).
 */
enum { RWH_TUNE_WARP_SHAPE = 0, RWH_TUNE_SCORE_HPW = 1, RWH_TUNE_SCORE_EXACT = 2, RWH_TUNE_WARP_FRAMES = 3 };
RWH_API int rwh_lab_tune(int knob, int value);

/*
 * Lab / measurement hook, not part of the data path: ONE wavefront that stays resident for `milliseconds` (<= 2000) of
 * the 100 MHz constant clock and then writes d_out[0] = shader-clock cycles elapsed (s_memtime), d_out[1] = 100 MHz ticks
 * elapsed (s_memrealtime).  Launched on a side stream beside the kernels being timed, d_out[0] / d_out[1] * 100 MHz is the
 * shader clock the chip held under that load (bench.py: roofline.sclk_mhz).  d_out: 2 x uint64 on the device.
 */
RWH_API int rwh_lab_clock_probe(uint64_t* d_out, double milliseconds, void* stream);

/*
 * Backward perspective warp.  Replaces, per call, the body of
 *   wrapPerspective      homography.py:166-179  (grid -> inv(H) -> divide -> interpolate)
 *   wrapPerspectiveScan  homography.py:197-208
 *   nearestNeighbor      homography.py:108-121
 *   bilinear             homography.py:123-138
 * and, with dst_dtype == RWH_U8 on a bilinear warp, the `astype(np.uint8)`
 * truncation of transformImage / transformImageH (homography.py:225, 240-241).
 *
 * Source: `batch` images of src_h x src_w x channels, element type src_dtype,
 * rows contiguous, image b at d_src + b*src_image_stride (bytes).
 * Output pixel (r, c) of the full out_h x out_w grid sits at output coordinate
 *   x = (c == out_w-1 ? x_last : x0 + c*step_x),  y likewise
 * (numpy.linspace semantics, homography.py:166-167 / 197-198); its source
 * coordinate is inv_h (row-major 3x3, float64, = numpy.linalg.inv(H) computed by
 * the caller as in homography.py:172) applied to (x, y, 1) and dehomogenised in
 * float64.  Coordinates outside [0, bound_w-1] x [0, bound_h-1] give 0
 * (homography.py:117-118 / 131-132; scan mode passes the OUTPUT resolution here,
 * homography.py:208).  bound_w/bound_h are clipped to the source size.
 * Bilinear weights come from the float64 fraction; the blend itself runs in
 * float32 (tolerance vs the float64 reference: 1e-4 relative, typ. 3e-7).
 * Mask edge, fast kernels (no RWH_WARP_EXACT): source coordinates are rounded once onto a 2^-32 px grid before the bounds
 * test, so a coordinate inside (-2^-33, 0) or (bound-1, bound-1 + 2^-33) counts as ON the edge texel where the reference
 * (which tests the unrounded float64, homography.py:131) returns 0.  On a generic map that is one pixel in ~10^9; an
 * axis-aligned map whose inv(H) carries 1e-16 round-off (a rotation by numpy's pi) can put a whole border row or column
 * there.  RWH_WARP_EXACT reproduces the reference's decision bit for bit (tests: test_warp_image_edge_band_vs_oracle).
 * Non-finite inv_h entries: a pixel whose denominator is +-Inf maps to (0, 0) in IEEE arithmetic (finite / Inf = 0) and shows
 * source texel (0,0) in the reference and in the exact kernels; the fast bilinear kernels mask it (0).  The reference blanks
 * texel (0,0) before it samples (RWH_WARP_ZERO_ORIGIN), so the two differ only for a caller that skips the blanking.
 * A NaN coordinate (0 / 0, Inf / Inf) escapes the reference's comparisons and makes it raise IndexError (it indexes with
 * INT_MIN, homography.py:133-135); every kernel here returns 0 for such a pixel.
 *
 * Only rows [row_begin, row_end) are produced; d_dst points at row `row_begin`
 * of image 0 and image b at d_dst + b*dst_image_stride (bytes).  This is the
 * unit of multi-GPU sharding (output-row tiles or images; no collective).
 *
 * Supported: channels 3 or 4; src_dtype U8 or F32; dst_dtype == src_dtype for
 * RWH_NEAREST; dst_dtype U8 (truncating) or F32 for RWH_BILINEAR (F64 or U8 with RWH_WARP_EXACT).
 * n_h is 1 (one homography for the whole batch) or `batch` (inv_h holds batch
 * 3x3 matrices, image b uses the b-th; any batch size -- the library launches
 * groups of images that share a kernel configuration).
 */
RWH_API int rwh_warp_backward(const void* d_src, int src_h, int src_w, int channels, int src_dtype,
                      int64_t src_image_stride, int batch,
                      const double* inv_h, int n_h,
                      double x0, double step_x, double x_last,
                      double y0, double step_y, double y_last,
                      int out_h, int out_w, int bound_h, int bound_w, int interp,
                      void* d_dst, int dst_dtype, int64_t dst_image_stride,
                      int row_begin, int row_end, unsigned flags, void* stream);

/*
 * The interpolators on coordinates the caller computed: replaces convertfunc[convert](z_t, img, h, w, mh, mw), i.e.
 *   nearestNeighbor  homography.py:108-121   (z + 0.5 truncated to int32, mask on the integers, gather)
 *   bilinear         homography.py:123-138   (mask on the float64 coordinates, truncation, float64 lerps)
 * d_x / d_y: the n source coordinates (rows 0 and 1 of the reference's dehomogenised 3 x N z_t), float64, on the device;
 * d_out: n x channels, element type = the image's for RWH_NEAREST (dst_dtype == src_dtype), float64 for RWH_BILINEAR
 * (dst_dtype RWH_F64) -- the reference's float64 arithmetic operation by operation, bit-identical to numpy's;
 * (bound_h, bound_w): the `h, w` arguments of the reference's call (clipped to the image); flags: RWH_WARP_ZERO_ORIGIN
 * blanks texel (0,0) of the image first, as both interpolators do to the caller's array.  The +1 taps are clamped to the
 * image where the reference raises IndexError (their weight is 0 there).
 */
RWH_API int rwh_sample_points(const void* d_img, int src_h, int src_w, int channels, int src_dtype,
                      const double* d_x, const double* d_y, int64_t n, int bound_h, int bound_w, int interp,
                      void* d_out, int dst_dtype, unsigned flags, void* stream);

/*
 * Would the REFERENCE raise IndexError on this warp?  Its interpolators index the image with every coordinate its mask lets
 * through (homography.py:117-121, 131-135): bilinear reads texel x + 1, so a coordinate exactly ON the last column / row
 * (the identity homography) indexes one past the image; a scan-mode `res` larger than the image lets coordinates beyond it
 * through; a NaN coordinate passes the float mask and indexes with INT_MIN.  The kernels of this library clamp / mask such
 * pixels; a drop-in host layer that wants the reference's error behaviour asks here first.  Coordinates by the exact
 * kernels' arithmetic; no image access.  bound_h / bound_w: the reference's mask (NOT clipped to the source).
 * *d_flag (device int32) = OR of: 1 an index past the last column (axis 1), 2 past the last row (axis 0), 4 a NaN coordinate.
 */
RWH_API int rwh_warp_index_check(int src_h, int src_w, const double* inv_h, double x0, double step_x, double x_last,
                         double y0, double step_y, double y_last, int out_h, int out_w, int bound_h, int bound_w,
                         int interp, int* d_flag, void* stream);

/*
 * Which kernel rwh_warp_backward would launch for these arguments (same dispatch code, nothing is launched, no device
 * pointer is needed): writes the kernel's name as rocprofv3 prints it, e.g. "rwh::warp_rgb8_fast8<unsigned char, 6>"
 * (with one homography per image: the first group's kernel).  For reports (bench.py's roofline.kernel) and tests.
 */
RWH_API int rwh_warp_plan(int src_h, int src_w, int channels, int src_dtype, int batch, const double* inv_h, int n_h,
                  double x0, double step_x, double x_last, double y0, double step_y, double y_last,
                  int out_h, int out_w, int bound_h, int bound_w, int interp, int dst_dtype,
                  int row_begin, int row_end, unsigned flags, char* kernel_name, int name_len);

/*
 * Batched 4-point DLT hypothesis generator.  Replaces K calls of
 *   HomoModel.fit(X[:,idx], Y[:,idx])   ransac.py:178-180 -> 52
 *   calcHomography / calc_corresp        homography.py:71-88 / 4-14
 * d_pts_a, d_pts_b: M x 2 float32 (the `matchespoints` layout, points in rows);
 * d_idx: K x 4 int32 sample indices in [0, M) -- a PRECONDITION: device tables are not range-checked (the host entry points
 *   rwh_ransac_run / rwh_host_dlt4_svd and the Python mirror check theirs) -- (drawn by the caller from numpy's
 * legacy generator for parity, ransac.py:177);
 * d_h: K x 9 float32, row-major 3x3 with h[8] == 1;
 * d_flags: K bytes, bit 0 = repeated index in the sample, bit 1 = non-finite
 * result (singular system), bit 2 = ill-conditioned sample (RWH_HYP_ILLCOND below).  The 8x9 system is built from float32-rounded
 * products like the reference, solved in float64, scaled to unit norm, rounded
 * to float32 and divided by its 9th element in float32.
 */
#define RWH_HYP_REPEATED 1u
#define RWH_HYP_SINGULAR 2u
#define RWH_HYP_ILLCOND 4u   /* bit 2: ill-conditioned sample -- a pivot of the elimination below 1e-3 of its column's scale
                                (three collinear source points, equal coordinates at different indices, ...) or a unit null
                                vector whose 9th element is below 1e-7: H is finite but K1's elimination and LAPACK's SVD
                                (the reference's solver) may round to different float32 H.  RANSAC.run re-derives flagged
                                samples on the host with the reference's own solver; ~2 % of the samples on natural matches.
                                In searches that invert the hypotheses (rwh_ransac_search / _batched with 'backward' or 'reproj')
                                the bit is also set for a nearly singular H (|det| below 1e-6 of the sum of the six products'
                                magnitudes): its inverse must be numpy.linalg.inv's own (rwh_score_count_inv). */
#define RWH_HYP_DEGENERATE 8u /* bit 3 (round 4; implies bit 2): the ill-conditioned samples whose H says nothing about the
                                reference's -- a pivot below 1e-7 of its column's scale, |n[8]| < 1e-7, a nearly singular H in a
                                search that inverts.  Always host-solved.  The other RWH_HYP_ILLCOND samples are accurate
                                here and, under 'fwd', are bounded by rwh_score_interval instead. */
RWH_API int rwh_dlt4_batched(const float* d_pts_a, const float* d_pts_b, int m,
                     const int32_t* d_idx, int k,
                     float* d_h, uint8_t* d_flags, void* stream);

/*
 * Hypothesis x correspondence reprojection-error inlier scorer.  Replaces, per
 * hypothesis,
 *   computeLoss(X, Y, method)           ransac.py:182 -> 84-98 (fwd 55-64, reproj 66-76, dist 78-82)
 *   inliers_pos = err < th ; np.sum     ransac.py:183-184
 * and the accept rules of ransac.py:186-202 in order-independent form.
 * One wavefront scores a run of consecutive hypotheses: lanes stride over the M
 * correspondences (kept in registers; M > 256: one wavefront per chunk of 256, counts
 * accumulated with integer atomic adds after a memset node), `err < th` is ballotted and
 * popcounted.  A pair whose distance clears `th` by a proven error band is decided from a
 * reciprocal; a hypothesis with any pair inside the band (or a NaN) is redone with the
 * reference's two IEEE float32 divisions: counts and masks are bit-identical to the
 * reference's arithmetic either way.
 * d_counts: K int32.  d_masks: optional (may be NULL) K x ceil(M/64) uint64
 * inlier bitmasks (bit j of word w = correspondence 64*w + j).
 * d_best: 2 x uint64, accumulated with atomic max, so several calls (hypothesis
 * shards, `hyp_base` = global index of row 0) may share it after ONE memset:
 *   word 0 = (count << 32) | (0xFFFFFFFF - global_index)  -> max count, lowest index on ties
 *   word 1 = 0xFFFFFFFF - (lowest global_index with count >= need), 0 if none
 * which is also the payload of the one RCCL all-reduce(max) in the sharded run.
 * `th` is compared as (double)err < th.
 * d_err: optional (may be NULL) K x M float32, the per-correspondence loss itself
 * (what HomoModel.computeLoss returns, ransac.py:84-98).
 */
RWH_API int rwh_score_count(const float* d_h, const float* d_pts_a, const float* d_pts_b, int m, int k,
                    double th, int loss, int need, int64_t hyp_base,
                    int32_t* d_counts, uint64_t* d_masks, uint64_t* d_best, float* d_err, void* stream);

/*
 * The same with the INVERSE of every hypothesis supplied by the caller (d_hinv: K x 9 float32, row-major; NULL = as above).
 * 'backward' and 'reproj' project through numpy.linalg.inv(val) (ransac.py:74: float64 LAPACK dgesv on the identity, cast
 * to float32).  The kernels' own float64 elimination gives the same float32 matrix for every homography a sane sample
 * produces, but NOT for a nearly singular one (a sample drawn from two or three clusters): there the two round apart, and so
 * do the losses.  The settle step of RANSAC.run therefore inverts the hypotheses it re-scores with numpy's own routine
 * (rwh_host_inv3) and scores them through this entry point.
 */
RWH_API int rwh_score_count_inv(const float* d_h, const float* d_hinv, const float* d_pts_a, const float* d_pts_b, int m, int k,
                        double th, int loss, int need, int64_t hyp_base, int32_t* d_counts, uint64_t* d_masks,
                        uint64_t* d_best, float* d_err, void* stream);

/*
 * Project M points through one homography.  Replaces HomoModel.fwd (ransac.py:55-64,
 * inverse == 0) and HomoModel.reproj (ransac.py:66-76, inverse != 0: through the
 * float64 inverse rounded to float32).  d_h: 9 float32; d_pts: M x 2 float32;
 * d_out: 3 x M float32 = (val @ [x;y;1]) / (row2 + 1e-10), same rounding recipe as
 * rwh_score_count.
 */
RWH_API int rwh_project_points(const float* d_h, const float* d_pts, int m, int inverse,
                       float* d_out, void* stream);

/*
 * General form of the same two methods: d_h 9 elements, d_pts3 3 x M (rows x, y, w: the caller's third row, which a 3 x M
 * input keeps, ransac.py:58-62), d_out 3 x M, all float32 (dtype RWH_F32) or all float64 (RWH_F64: what numpy computes when
 * model.val -- the float64 refit after RANSAC.run -- or the points are float64).  For reproj the caller passes inv(val)
 * (numpy.linalg.inv on the host, as ransac.py:74 does).  Same k-order: rounded multiply, FMA, FMA; IEEE divides.
 */
RWH_API int rwh_project_points_ex(const void* d_h, const void* d_pts3, int m, int dtype, void* d_out, void* stream);

/*
 * The RANSAC search of ransac.py:176-202 as ONE call: (optional) reset of the two packed keys, then
 * rwh_dlt4_batched + rwh_score_count on the same stream with caller-provided workspaces (d_h K x 9,
 * d_flags K, d_counts K, d_masks K x ceil(M/64) or NULL).  Saves two host round trips per run; arguments
 * as in the two functions above.
 */
RWH_API int rwh_ransac_search(const float* d_pts_a, const float* d_pts_b, int m, const int32_t* d_idx, int k,
                      double th, int loss, int need, int64_t hyp_base,
                      float* d_h, uint8_t* d_flags, int32_t* d_counts, uint64_t* d_masks, uint64_t* d_best,
                      int reset_best, void* stream);

/* flag for rwh_ransac_batched */
#define RWH_BATCH_DEVICE_SAMPLING 1u /* fill d_idx on the device (Philox4x32-10) instead of reading the caller's table */
#define RWH_BATCH_EARLY_STOP 2u      /* the `break` of ransac.py:186-190 as saved work: once a hypothesis of a problem has reached
                                        its `need`, scorer waves of LATER hypotheses of that problem skip (their d_counts entry is
                                        -1, their mask words 0).  The winner is unchanged -- the lowest index that reaches `need`
                                        -- because a skipped hypothesis always has a lower-index exit on record. */

/*
 * Many independent RANSAC searches in one submission (no counterpart in the reference,
 * whose RANSAC.run handles one image pair per call, ransac.py:159-213): the loop of ransac.py:176-202 for P problems
 * x K hypotheses each, as three launches (sampling, 4-point DLT, scorer) regardless of P.
 *   d_pts_a, d_pts_b: the problems' correspondences concatenated, total x 2 float32;
 *   d_offsets: P+1 int32, problem p owns rows d_offsets[p] .. d_offsets[p+1]-1; m_max >= the largest problem;
 *   d_idx: P x K x 4 int32, indices LOCAL to each problem.  Without RWH_BATCH_DEVICE_SAMPLING the caller provides it
 *     (e.g. numpy's stream, then every problem's result equals rwh_ransac_search on that table bit for bit); with the
 *     flag the library fills it: Philox4x32-10, counter (hypothesis, problem_base + problem, 0, 0) -- `problem_base` is
 *     the global index of this call's first problem, so that a problem list sharded over several calls / GPUs draws
 *     the tables of the unsharded run --, key = seed, four DISTINCT indices
 *     per hypothesis by multiply-shift range reduction -- NOT the reference's sampler (ransac.py:177 draws with
 *     replacement from numpy's legacy generator), a documented non-parity mode; problems with fewer than 4
 *     correspondences get (0,0,0,0) and are flagged RWH_HYP_REPEATED;
 *   d_need: P int32, per-problem early-exit count (ceil(M*d/100 + n), ransac.py:169);
 *   d_h P*K x 9, d_flags P*K, d_counts P*K, d_masks P*K x ceil(m_max/64) (or NULL; words past a problem's own
 *     ceil(M/64) are written as 0);
 *   d_best: P x 2 uint64, reset here; per problem the two packed keys of rwh_score_count with the hypothesis index
 *     counted inside the problem.
 * Scoring arithmetic, tie-break and early-exit rules are those of rwh_score_count.
 */
RWH_API int rwh_ransac_batched(const float* d_pts_a, const float* d_pts_b, const int32_t* d_offsets, int n_problems,
                       int m_max, int k, int32_t* d_idx, uint64_t seed, int64_t problem_base, double th, int loss,
                       const int32_t* d_need, float* d_h, uint8_t* d_flags, int32_t* d_counts,
                       uint64_t* d_masks, uint64_t* d_best, unsigned flags, void* stream);

/*
 * HOST helper of the settle step (no device work, no stream): the reference's own 4-point solve for n samples,
 *   calc_corresp (homography.py:4-14: 8 x 9 float32 DLT matrix, float32 products) -> numpy.linalg.svd (LAPACK dgesdd,
 *   float64 inside) -> last row of V^T cast to float32 -> / its 9th element in float32   (homography.py:71-88),
 * on `threads` host threads.  pts_a / pts_b: m x 2 float32 HOST arrays; idx_rows: n x 4 int32 (HOST); out_h: n x 9 float32.
 * dgesdd_ilp64: address of the Fortran symbol dgesdd (64-bit integers) of the LAPACK the caller's numpy uses
 * (numpy >= 2: `scipy_dgesdd_64_` in numpy.libs/libscipy_openblas64_*.so): called with numpy's own arguments, so every H is
 * numpy.linalg.svd's bit for bit -- this entry point only moves the loop off the Python interpreter and onto several
 * cores.  The accept rules of RANSAC.run need this solver (LAPACK's null vector of a rank-deficient sample is arbitrary
 * but is what the reference uses, ransac.py:177-180) for samples K1 flags and for hypotheses near the decision.
 * Returns RWH_E_LAUNCH if LAPACK reported info != 0 for a sample (numpy would raise LinAlgError).
 */
RWH_API int rwh_host_dlt4_svd(const float* pts_a, const float* pts_b, int m, const int32_t* idx_rows, int n,
                      void* dgesdd_ilp64, int threads, float* out_h);

/*
 * HOST helper: n x numpy.linalg.inv of a float32 3 x 3 (ransac.py:74) -- float64 dgesv on the identity, the routine numpy
 * calls, by address (`scipy_dgesv_64_`), cast back to float32.  h, out: n x 9 float32 HOST arrays, row-major.  A singular
 * matrix (numpy raises LinAlgError) gives NaNs.
 */
RWH_API int rwh_host_inv3(const float* h, int n, void* dgesv_ilp64, float* out);

/*
 * The inlier count of listed hypotheses as an interval ('fwd' loss, ransac.py:55-64 + 78-82 + 183): for row d_rows[i] of d_h
 * (d_rows NULL: rows 0 .. n_rows-1), d_lo[i] = the pairs that are inliers for EVERY H within the perturbation budget of d_h's row,
 * d_hi[i] = the pairs that are inliers for SOME such H.  Budget: every entry may move by delta x its natural scale (rows 0 / 1:
 * s, s, s C; row 2: s / C, s / C, s with C = coord_scale >= 1, the magnitude of the coordinates, and s the largest scale-free
 * entry), delta = delta1 for rows whose d_flags byte has RWH_HYP_ILLCOND set, delta0 otherwise (d_flags may be NULL).  The settle
 * step of RANSAC.run uses it to decide which hypotheses need the reference's own solver: LAPACK's float32 H lies within a few
 * ulps (natural scale) of K1's, so d_lo == d_hi means the reference's count IS rwh_score_count's, and a d_hi below the best
 * d_lo cannot win.  th as in rwh_score_count.
 */
RWH_API int rwh_score_interval(const float* d_h, const int32_t* d_rows, int n_rows, const uint8_t* d_flags, const float* d_pts_a,
                       const float* d_pts_b, int m, double th, double coord_scale, double delta0, double delta1,
                       int32_t* d_lo, int32_t* d_hi, void* stream);

/*
 * Host code: `count` draws of numpy's LEGACY np.random.randint(0, m, ...) (ransac.py:177 samples with it) from the MT19937 state
 * the caller took with RandomState.get_state() -- key[624], pos, updated in place for set_state() --: the identical stream
 * (one 32-bit output per draw, masked, rejected above m - 1; m == 1 draws nothing), without the interpreter and the generator's
 * lock around every draw.  out32 (int32, the index table the kernels take) and / or out64 (what numpy returns) may be NULL.
 * 1 <= m < 2^31.  tests/test_settle_cpu.py holds it to numpy's own output.
 */
RWH_API int rwh_host_legacy_randint(uint32_t* key, int32_t* pos, int64_t m, int64_t count, int32_t* out32, int64_t* out64);

/*
 * The host driver of RANSAC.run (ransac.py:159-213 up to, not including, the final refit) as ONE native call: upload,
 * rwh_ransac_search, the settle step, and the accept rules: the first hypothesis with count >= need wins and ends the search,
 * else the first maximum.  Winner, count and inlier mask equal the reference loop's on the same index table (tests: every
 * RANSAC fixture the reference produced).
 * The settle step gives the reference's own H (rwh_host_dlt4_svd, re-scored by rwh_score_count) to every hypothesis whose K1 H
 * may not stand for it in the decision; the repeated-index samples are solved on host threads while the GPU searches.
 *   'fwd' (round 4): repeated-index / non-finite / RWH_HYP_DEGENERATE samples always; of the others -- every RWH_HYP_ILLCOND
 *   sample and every hypothesis within 32 counts of the best or of `need` get a count INTERVAL (rwh_score_interval, budgets of 16 /
 *   64 float32 ulps of natural scale) -- those whose interval is not a point AND reaches the best lower bound or `need`.
 *   'backward' / 'reproj': every flagged sample and every hypothesis within min(margin_cap, 3 + count / 16) of a decision.
 * What remains EMPIRICAL in both: that LAPACK's float32 H lies within the budget (resp. that an unflagged K1 count is within the
 * margin) of K1's -- measured on 13 problem families and ~30 000 soak cases (profiles/r04_lab_notes.txt), not proven.
 *   pts_a, pts_b: m x 2 float32, idx: k x 4 int32 -- HOST arrays (the reference's inputs are host arrays; idx = the first
 *   four columns of numpy's draws, ransac.py:177);
 *   d_ws / h_ws: device workspace and PAGE-LOCKED host workspace of the sizes rwh_ransac_run_layout reports (offsets[11] and
 *   offsets[19]); after the call d_ws holds K1's H (k x 9 float32 at offsets[3]), K2's counts (int32, offsets[4]), K1's flags
 *   (offsets[5]) and the masks (offsets[6]); h_ws holds K2's raw counts (offsets[13]), the flags (offsets[14]) and the counts
 *   after the settle step (offsets[15]);
 *   hyp_base (round 4): global index of hypothesis 0 -- a rank of a sharded search passes its slice of the table and its offset;
 *   out: 8 x int32 = winner index inside this table (-1: nothing ever scored > 0), early exit (0 / 1), winner's count, hypotheses
 *   solved on the host, settle rounds, samples flagged by K1, hypotheses given an interval, 0;
 *   out_keys (may be NULL): 2 x uint64, this table's packed keys as rwh_score_count packs them -- (count << 32) | (0xFFFFFFFF -
 *   (hyp_base + winner)) and 0xFFFFFFFF - (hyp_base + first hypothesis with count >= need), or 0 --: the payload of the ONE
 *   all-reduce(MAX) of a sharded search;  out_mask: ceil(m / 64) x uint64, the winner's inlier bitmask.
 *   dgesv_ilp64: address of LAPACK dgesv in the same library (numpy.linalg.inv's routine), or NULL: with it the hypotheses
 *   the settle step re-scores under 'backward' / 'reproj' are inverted by rwh_host_inv3 (see rwh_score_count_inv).
 * Synchronises `stream` (its results are host values).  rwh_ransac_run_layout: fills offsets[0 .. 27), returns 27.
 */
RWH_API int rwh_ransac_run_layout(int m, int k, long long* offsets, int n_offsets);
RWH_API int rwh_ransac_run(const float* pts_a, const float* pts_b, int m, const int32_t* idx, int k, double th, int loss,
                   int need, int margin_cap, void* dgesdd_ilp64, void* dgesv_ilp64, int threads, void* d_ws, void* h_ws,
                   int64_t hyp_base, int32_t* out, uint64_t* out_keys, uint64_t* out_mask, void* stream);

/*
 * Fused panorama compositor.  Replaces the body of stitchPanorama (homography.py:288-338) after its canvas
 * geometry (host, homography.py:303-321): addAlpha('Rate') + transformImageH + paste / alpha blend, in one pass
 * over the canvas, float64 arithmetic in the reference's order -> uint8 canvas bit-identical to the reference's.
 * d_img_t: imgT (t_h x t_w x 3 uint8, the image warped by H); d_img_q: imgQ (q_h x q_w x 3 uint8).
 * inv_h: inv(H); (grid_x0, grid_y0, warp_w, warp_h): wrapPerspective's output grid (min_x, min_y, max_w, max_h);
 * (tsx, tsy) / (qsx, qsy): where the warped imgT / imgQ sit on the canvas_h x canvas_w canvas.
 * blend == 0: paste imgQ over the warped imgT; 1: the 'Rate' alpha blend with blendrate `rate`; 2: the 'Gradient' blend
 * (alpha of imgT = the (x + y) / (w + h) / 2 ramp of homography.py:260-265, alpha of imgQ = 1; always the exact kernel);
 * 3 (round 4): any OTHER truthy `blending` of the reference -- addAlpha then leaves imgT's alpha plane at 0 (homography.py:250-266),
 * the alpha-weighted mean keeps imgQ / 0 everywhere and the warp only decides where the reference would raise (exact kernel).
 * Any other value: RWH_E_INVALID.
 * flags: RWH_WARP_ZERO_ORIGIN blanks texel (0,0) of imgT first, as bilinear() does to the caller's array;
 * RWH_STITCH_FAST: the staged float32-blend warp kernel with the compositor as its epilogue (rwh::warp_rgb8_comp) instead of
 * the exact float64 kernel (four pixels per lane; 0.24 ms paste / 0.57 ms blend for a 13 181 x 6 313 canvas): 1.5-2.6x faster, canvas
 * within 1 LSB of the reference's (the alpha plane's own
 * bilinear lerp is taken as constant: the one output pixel whose taps include the blanked texel (0,0) can differ more).
 */
#define RWH_STITCH_FAST 4u
RWH_API int rwh_stitch_panorama(const void* d_img_t, int t_h, int t_w, const void* d_img_q, int q_h, int q_w,
                        const double* inv_h, int grid_x0, int grid_y0, int warp_w, int warp_h,
                        int tsx, int tsy, int qsx, int qsy, int canvas_h, int canvas_w,
                        int blend, double rate, void* d_canvas, unsigned flags, void* stream);

/*
 * The same, canvas rows [row_begin, row_end) only (d_canvas still points at row 0; always the exact kernel): lets a host layer
 * that gets its images as host arrays compose a row tile as soon as the image rows it reads have arrived, and send it
 * back while later rows are still on their way up (full-duplex PCIe).  RWH_WARP_ZERO_ORIGIN: pass it with the FIRST tile
 * only (it writes texel (0,0) of imgT).
 */
RWH_API int rwh_stitch_panorama_rows(const void* d_img_t, int t_h, int t_w, const void* d_img_q, int q_h, int q_w,
                             const double* inv_h, int grid_x0, int grid_y0, int warp_w, int warp_h,
                             int tsx, int tsy, int qsx, int qsy, int canvas_h, int canvas_w,
                             int blend, double rate, void* d_canvas, int row_begin, int row_end, unsigned flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RWH_H */
