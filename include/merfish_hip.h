/* merfish_hip.h -- C-ABI of libmerfish_hip.so (MI355X / gfx950).
 *
 * The reference (emanuega/MERlin v0.1.6) is pure Python and has no FFI; its
 * extension point for this path is the PixelBasedDecoder object
 * (merlin/util/decoding.py:25).  This header is the C boundary a maintainer
 * binds (ctypes) beneath that object; every entry point names the reference
 * code it replaces.  See INTEGRATION.md for the binding stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name starts with h_ (host);
 *  - all device buffers are allocated and owned by the caller (PyTorch-ROCm
 *    tensors' data_ptr()); the library allocates no device memory;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *    call only enqueues work on that stream and does not synchronise;
 *  - return value: 0 on success, negative MF_E* code on error; the message is
 *    available from mf_last_error_string() (thread-local);
 *  - image volumes are [Z,H,W] row-major; "2-D mode" (conn3d = 0) treats the Z
 *    planes as independent images, as Decode does per z-slice
 *    (merlin/analysis/decode.py:118-127); "3-D mode" (conn3d = 1) is
 *    decode_3d (merlin/analysis/decode.py:129-161).
 */
#ifndef MERFISH_HIP_H
#define MERFISH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_OK            0
#define MF_EINVAL       -1   /* bad argument (shape, dtype, null pointer)   */
#define MF_EUNSUPPORTED -2   /* outside the built kernel set (B > 32, ...)  */
#define MF_EWORKSPACE   -3   /* workspace too small                         */
#define MF_EHIP         -4   /* a HIP runtime call failed                   */

#define MF_MAX_BITS     32   /* bits per codeword supported by the kernels  */
#define MF_MAX_RADIUS    6   /* blur radius of the fused decode kernel:
                              ksize <= 13, i.e. sigma <= 3                 */
#define MF_LOWPASS_MAX_RADIUS 32 /* mf_lowpass_images: ksize <= 65, sigma <= 16 */
#define MF_REC_COLS     12   /* doubles per region row, see mf_region_table */
#define MF_FRAME_COLS   10   /* float64 columns of mf_region_frame          */
#define MF_TYPED_F32_COLS 7  /* float32 columns of mf_region_frame_typed    */
#define MF_PP_MAX_RADIUS 8   /* preprocess high-pass radius: sigma <= 4     */
#define MF_HIST_BINS 65534   /* bins of np.arange(0, 65535), preprocess.py:114 */
#define MF_FRAME_STATIC_MAX_ROWS 1048576 /* row limit of mf_region_frame_static */

#define MF_CODEBOOK_UNIFORM 1 /* flag: all codewords share weight and value   */
#define MF_CODEBOOK_MINHD_SHIFT   8  /* flags bits 8-15: minimum Hamming distance
                                        of a uniform codebook (0 = not computed) */
#define MF_CODEBOOK_WEIGHT_SHIFT 16  /* flags bits 16-23: its codeword weight    */

#define MF_DECODE_TRACES_DECODED_ONLY 1 /* mf_decode_pixels flag, see there   */
#define MF_DECODE_KEEP_RUNNER_UP     2 /* mf_decode_pixels flag, see there   */

#define MF_DTYPE_U16     0
#define MF_DTYPE_F32     1
#define MF_DTYPE_F64     2

int mf_version(void);
const char *mf_last_error_string(void);

/* cv2.GaussianBlur kernel taps (merlin/util/decoding.py:94-97): the 16.16
 * fixed-point taps OpenCV 4.x uses for CV_16U (sum == 65536) and the float32
 * taps used for float input.  Host-only helper; h_fixed / h_float hold ksize
 * entries each; ksize odd, <= 2 * MF_LOWPASS_MAX_RADIUS + 1. */
int mf_gaussian_taps(int ksize, double sigma, uint32_t *h_fixed,
                     float *h_float);

/* cv2.GaussianBlur of every plane of a stack (merlin/util/decoding.py:93-97) as
 * a pass of its own, for lowPassSigma beyond the fused kernel of
 * mf_decode_pixels (sigma > 3): blur with this, then decode `out` with sigma =
 * 0 -- the same bits as the fused path would give.  uint16 planes take OpenCV
 * >= 4's exact fixed-point path, float32 planes the float32 kernel; border
 * REFLECT_101; ksize = 2 ceil(2 sigma) + 1 <= 65.
 *   image  [n_planes] planes of H x W (rows contiguous), plane_stride elements
 *          apart; out: contiguous [n_planes, H, W] of the same dtype.
 *   workspace: mf_lowpass_workspace_bytes(...) bytes of device memory. */
int64_t mf_lowpass_workspace_bytes(int image_dtype, int64_t n_planes, int H,
                                   int W);
int mf_lowpass_images(const void *image, int image_dtype, int64_t n_planes,
                      int H, int W, int64_t plane_stride, double sigma,
                      void *out, void *workspace, int64_t workspace_bytes,
                      void *stream);

/* Codebook table (replaces PixelBasedDecoder.__init__ /
 * _calculate_normalized_barcodes' float64 matrix, decoding.py:27-44, 254-276,
 * as the device-side form the kernels read).
 *
 *   h_codebook  HOST f64 [N,B]: L2-normalised rows exactly as the reference
 *               builds them.  MERFISH codewords are binary, so every non-zero
 *               entry of a row must carry the same value; anything else
 *               returns MF_EUNSUPPORTED.
 *   table       device buffer of mf_codebook_table_bytes(N, B) bytes
 * Synchronous (one hipMemcpy); call once per codebook.  Returns the codebook
 * flags (>= 0, MF_CODEBOOK_*) that mf_decode_pixels takes, or an error. */
int64_t mf_codebook_table_bytes(int N, int B);
int mf_codebook_prepare(const double *h_codebook, int N, int B, void *table,
                        int64_t table_bytes);

/* decode_pixels (merlin/util/decoding.py:46-131) for Z planes at once.
 *
 *   image      u16 (MF_DTYPE_U16) or f32 stack, element (b,z,y,x) at
 *              image[b*img_bit_stride + z*img_z_stride + y*W + x]
 *   codebook_table  device table from mf_codebook_prepare for N codewords
 *   codebook_flags  the value mf_codebook_prepare returned for that table
 *   h_scale, h_background   host float[B]: scale factors / backgrounds already
 *              rounded to float32 (legacy numpy value casting, decoding.py:103)
 *   decoded    int16 [Z,H,W]   barcode index or -1
 *   magnitudes f32   [Z,H,W]
 *   distances  f32 or f64 [Z,H,W] (dist_dtype), NOT masked by the thresholds
 *   traces     f32, element (z,b,y,x) at traces[z*tr_z_stride+b*tr_bit_stride
 *              + (y*W + x)*tr_pix_stride]; may be NULL (traces not materialised).
 *              The reference's layout is plane-major (tr_pix_stride = 1,
 *              tr_bit_stride = H*W); the FOV runner uses pixel-major
 *              (tr_bit_stride = 1, tr_pix_stride = B), which mf_region_table
 *              gathers one cache line per pixel.
 *   flags      MF_DECODE_TRACES_DECODED_ONLY: write the traces only of pixels
 *              whose decoded value is >= 0 (all that the region table reads);
 *              the other entries of `traces` are left untouched.
 *              MF_DECODE_KEEP_RUNNER_UP: always track the second-best codeword
 *              and take the exact-tie path, also for codebooks whose minimum
 *              Hamming distance makes a runner-up within the threshold
 *              impossible (MHD4 at thresholds <= 0.5176); same results, used
 *              to check that shortcut.
 *
 * Blur: ksize = 2*ceil(2*sigma)+1 <= 9, BORDER_REFLECT_101, exact integer
 * arithmetic for u16.  Nearest codeword replaces sklearn NearestNeighbors
 * (decoding.py:112-116): the winner is the exact float64 Euclidean 1-NN
 * (lowest index on exact ties) for every pixel that can pass the distance
 * threshold; for a pixel farther than the threshold whose two best float32
 * scores differ by < 1e-5 the reported winner may be either of them, so its
 * distance agrees with the exact one to 2^-18 = 4e-6 relative (the search
 * key carries its accumulator slot in the low four mantissa bits, which moves
 * a key by < 2^-19 of itself; measured worst case 2.4e-6). */
int mf_decode_pixels(const void *image, int image_dtype,
                     int B, int Z, int H, int W,
                     int64_t img_bit_stride, int64_t img_z_stride,
                     const void *codebook_table, int N, int codebook_flags,
                     const float *h_scale, const float *h_background,
                     double sigma, double distance_threshold,
                     float magnitude_threshold,
                     int16_t *decoded, float *magnitudes,
                     void *distances, int dist_dtype,
                     float *traces, int64_t tr_z_stride, int64_t tr_bit_stride,
                     int64_t tr_pix_stride, int flags, void *stream);

/* Connected-component labelling of the decoded image: replaces the N calls to
 * skimage.measure.label(decodedImage == i) (decoding.py:165-166, 324-325,
 * 360-361) by one multi-label pass.  Two pixels are connected iff they hold
 * the same barcode index >= 0 and are 8-neighbours (conn3d = 0, each z plane
 * on its own) or 26-neighbours (conn3d = 1).
 *
 *   labels  int32 [Z,H,W]: linear index (z*H*W + y*W + x) of the first pixel
 *           of the component in raster order; -1 for unassigned pixels;
 *           16-byte aligned
 *   counts  int64 [MF_LABEL_COUNTS_LEN] (device): counts[0] = assigned pixels,
 *           counts[1] = components; the caller reads these two back to size
 *           the tables below.  The rest is scratch: the workgroups add their
 *           partial counts into 64 slots, because thousands of atomics on one
 *           address serialise (17 ns each: a third of the labelling's last
 *           kernel), and a one-wave kernel sums the slots. */
#define MF_LABEL_COUNT_SLOTS 64
#define MF_LABEL_COUNTS_LEN (2 + 2 * MF_LABEL_COUNT_SLOTS)
int mf_label_components(const int16_t *decoded, int Z, int H, int W,
                        int conn3d, int32_t *labels, int64_t *counts,
                        void *stream);

/* Bytes of scratch mf_region_table needs for the given sizes. */
int64_t mf_region_workspace_bytes(int Z, int H, int W, int64_t n_pixels,
                                  int64_t n_regions);

/* regionprops-style aggregation (decoding.py:165-241, 323-334, 359-369) of
 * every component at once.  Rows are ordered as the reference emits them:
 * z plane (2-D mode only), then barcode index, then skimage label order
 * (raster order of the first pixel).
 *
 *   rec      f64 [n_regions, MF_REC_COLS]:
 *            0 barcode_id  1 z plane (2-D mode) / first-pixel z (3-D mode)
 *            2 area        3 mean_intensity      4 max_intensity
 *            5 mean_distance  6 min_distance
 *            7 centroid z  8 centroid row (y)    9 centroid column (x)
 *            10 first-pixel linear index
 *            11 reserved (0); the position of the row inside its (z plane,
 *               barcode) group is worked out by mf_region_frame
 *   inten    f32 [n_regions, B]  per-bit mean of the normalised traces
 *   mean_tm  f32 [n_regions, B]  per-bit mean of trace*magnitude, or NULL
 *   min_tm   f32 [n_regions, B]  per-bit min  of trace*magnitude, or NULL
 *            (extract_refactors / _extract_backgrounds inputs; 2-D mode)
 *   n_pixels / n_regions: the counts mf_label_components produced.
 *   n_codewords: the codebook's size N (every barcode in `decoded` is < N), or
 *   0 if unknown (then 32768): it only makes the regions' sort keys dense --
 *   planes x N <= 1024 sorts in one pass.
 *   decoded and labels must be 16-byte aligned (MF_EINVAL otherwise; any
 *   hipMalloc'ed or torch tensor base is): the run kernels read eight barcodes
 *   per access. */
int mf_region_table(const int16_t *decoded, const float *magnitudes,
                    const void *distances, int dist_dtype,
                    const float *traces, int64_t tr_z_stride,
                    int64_t tr_bit_stride, int64_t tr_pix_stride,
                    const int32_t *labels,
                    int B, int n_codewords, int Z, int H, int W, int conn3d,
                    int64_t n_pixels, int64_t n_regions,
                    double *rec, float *inten, float *mean_tm, float *min_tm,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* Crop / area filter and column gather of the region table: the tail of
 * extract_barcodes_with_index (decoding.py:222-252) for every row at once.
 * A row is kept iff  crop_width < x < crop_shape0 - crop_width  and
 * crop_width < y < crop_shape1 - crop_width  (the reference compares x with
 * shape[0] and y with shape[1]) and area >= minimum_area.  Kept rows keep their
 * order.
 *
 *   out64  f64 [MF_FRAME_COLS][n_regions] column-major, first *count entries of
 *          each column valid: barcode_id, mean_intensity, max_intensity, area,
 *          mean_distance, min_distance, x, y, z, index-in-group
 *   out32  f32 [B][n_regions]: the per-bit intensities, same convention
 *   count  int64 [1]: number of kept rows
 * out64 / out32 / count may be device memory (a full grid writes the rows;
 * copy them to the host afterwards) or page-locked host memory that is mapped
 * into the device (hipHostMalloc): four workgroups then write the rows straight
 * across PCIe, which slows kernels running beside them by about 5 %.  Read them
 * after the stream has been synchronised. */
int64_t mf_region_frame_workspace_bytes(int64_t n_regions);
int mf_region_frame(const double *rec, const float *inten, int64_t n_regions,
                    int B, int conn3d, double crop_width, double crop_shape0,
                    double crop_shape1, double minimum_area,
                    double *out64, float *out32, int64_t *count,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* mf_region_frame with the rows already in the barcode database's column
 * types (merlin/util/barcodedb.py:46-65; the astype of write_barcodes,
 * :264-266), so the record sink takes them without a host-side cast:
 *   out_u16   u16 [2][n_regions]  barcode_id, area (area wraps modulo 2^16 as
 *             numpy's astype does)
 *   out_f32   f32 [MF_TYPED_F32_COLS][n_regions]  mean_intensity, max_intensity,
 *             mean_distance, min_distance, x, y, z (float64 rounded to nearest)
 *   out_index i32 [n_regions]     position of the row inside its (plane, barcode)
 *             frame
 *   out_inten f32 [B][n_regions]  per-bit intensities
 *   out_xy64  f64 [2][n_regions]  x, y unrounded, for a global aligner
 *             (analysis/globalalign.py:141-148); may be NULL
 * Same filter, row order, workspace and count as mf_region_frame. */
int mf_region_frame_typed(const double *rec, const float *inten,
                          int64_t n_regions, int B, int conn3d,
                          double crop_width, double crop_shape0,
                          double crop_shape1, double minimum_area,
                          uint16_t *out_u16, float *out_f32, int32_t *out_index,
                          float *out_inten, double *out_xy64, int64_t *count,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* The per-barcode reductions of extract_refactors and _extract_backgrounds
 * (merlin/util/decoding.py:321-334, 355-370) from the region table, in float64
 * like the reference.  For every barcode b, over the rows of z plane `plane`
 * (-1: every plane, in plane order) with area >= area_threshold, in the
 * table's row order (= the reference's label order):
 *   seen[b]    = number of those rows                       (barcodesSeen)
 *   normalize: sums[b,:] = sum of  w / ||w|| / seen[b],  w = stat[r,:] - h_offset
 *              (stat = mean_tm; h_offset = the background refactors, decoding.py:
 *              328-332)
 *   else:      sums[b,:] = sum of  stat[r,:] - h_offset
 *              (stat = min_tm, area_threshold 5: decoding.py:362-369)
 *   rec, stat  the table of mf_region_table (2-D mode); h_offset HOST f64[B] or
 *              NULL; sums f64 [N,B]; seen f64 [N]
 * The caller finishes on the host ([N,B] arithmetic: NaN at off-bits, nanmean,
 * ratio).  Eight rows are accumulated side by side and added at the end, so
 * the sums agree with the reference's sequential ones to float64 rounding. */
int mf_refactor_sums(const double *rec, const float *stat, int64_t n_regions,
                     int B, int N, int Z, int plane, double area_threshold,
                     const double *h_offset, int normalize, double *sums,
                     double *seen, void *stream);

/* mf_region_frame without a workspace argument, for callers that cannot
 * allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arguments, results
 * and kernels as mf_region_frame; the scratch (kept-row indexes and chunk
 * counts, 4.004 MiB) is a static __device__ array of the library, one copy per
 * GPU held until the process ends, so the call itself allocates nothing.
 *   - n_regions > MF_FRAME_STATIC_MAX_ROWS returns MF_EWORKSPACE;
 *   - the device that holds `rec` must be the current device (MF_EINVAL else);
 *   - every call on a GPU shares that GPU's array: calls that overlap on one
 *     GPU (two streams, two threads) race -- serialise them, or use
 *     mf_region_frame with one workspace per stream.  Nothing in the array has
 *     to start at zero; each call overwrites what it reads. */
int mf_region_frame_static(const double *rec, const float *inten,
                           int64_t n_regions, int B, int conn3d,
                           double crop_width, double crop_shape0,
                           double crop_shape1, double minimum_area,
                           double *out64, float *out32, int64_t *count,
                           void *stream);

/* ---- preprocess stage: what produces the uint16 stack
 * mf_decode_pixels reads.  Image batches are [n_images, H, W] row-major. ---- */

/* high_pass_filter on uint16 (merlin/util/imagefilters.py:9-28 as called by
 * DeconvolutionPreprocess._high_pass_filter, analysis/preprocess.py:103-108):
 * low-pass = cv2.GaussianBlur(image, (window_size, window_size), sigma,
 * BORDER_REPLICATE), window_size odd and <= 17, in OpenCV 4.x's exact
 * fixed-point CV_16U arithmetic; out = image - lowpass, 0 where lowpass >
 * image.  Not in place. */
int mf_high_pass_filter(const uint16_t *image, int n_images, int H, int W,
                        int window_size, double sigma, uint16_t *out,
                        void *stream);

/* Scratch for the calls below; `batch` images are deconvolved together
 * (26 B/px each: three float64 images and the uint16 high-pass).  Larger
 * batches are faster (0.625 ms per 2048 x 2048 frame at 28 frames, 0.609 at
 * 112 = 12 GB: more strips to fill the chip evenly, longer strip segments); the
 * working set streams through HBM either way, it does not fit the 256 MiB
 * Infinity Cache beyond one frame. */
int64_t mf_preprocess_workspace_bytes(int batch, int H, int W);

/* deconvolve_lucyrichardson (merlin/util/deconvolve.py:70-133), odd
 * window_size <= 13 (the cv2.GaussianBlur branch; BORDER_REPLICATE), float64
 * arithmetic throughout like the reference.
 *
 *   image      u16 or f64 (image_dtype) [n_images,H,W]; not modified
 *   out        f64 (the reference's return value) or u16 (out_dtype): the
 *              `.astype(np.uint16)` of preprocess.py:139 (truncate toward
 *              zero, wrap modulo 2^16); must not alias `image`
 *   histogram  int64 [ceil(n_images/hist_group), MF_HIST_BINS] or NULL: counts
 *              of the uint16 output over np.arange(0, 65535)
 *              (preprocess.py:114-129) are ADDED to row image / hist_group;
 *              the caller zeroes it.  uint16 output only.
 *   lambdas    f64 [n_images, iterations] or NULL: the acceleration parameter
 *              every iteration used (deconvolve.py:110-111), for checking
 *   batch      images per pass; the workspace must hold
 *              mf_preprocess_workspace_bytes(batch, H, W)
 * The acceleration sums are reduced in a fixed order, so results are
 * identical from run to run; they differ from numpy's pairwise np.sum in the
 * last bits. */
int mf_deconvolve_lucyrichardson(const void *image, int image_dtype,
                                 int n_images, int H, int W, int window_size,
                                 double sigma, int iterations, void *out,
                                 int out_dtype, int64_t *histogram,
                                 int hist_group, double *lambdas, int batch,
                                 void *workspace, int64_t workspace_bytes,
                                 void *stream);

/* deconvolve_lucyrichardson with an EVEN windowSize (merlin/util/deconvolve.py:
 * 102-104, 118-119, 126-128): both blurs of an iteration are
 * scipy.ndimage.convolve(x, matlab_gauss2D((w, w), sigma), mode='constant') --
 * a non-separable kernel, zeros outside the image, ndimage's own tap order and
 * origin -- and the blurred ratio is zeroed where it exceeds 2^16.
 *   h_weights  HOST f64 [window_size, window_size]: the kernel as
 *              matlab_gauss2D (merlin/util/matlab.py:11-24) returns it,
 *              unflipped; window_size <= 16 (odd sizes are accepted and follow
 *              ndimage's rules too, but the reference sends odd windows to
 *              cv2.GaussianBlur = mf_deconvolve_lucyrichardson)
 * Other arguments as mf_deconvolve_lucyrichardson (same workspace). */
int mf_deconvolve_lucyrichardson_kernel2d(const void *image, int image_dtype,
                                          int n_images, int H, int W,
                                          int window_size,
                                          const double *h_weights,
                                          int iterations, void *out,
                                          int out_dtype, int64_t *histogram,
                                          int hist_group, double *lambdas,
                                          int batch, void *workspace,
                                          int64_t workspace_bytes,
                                          void *stream);

/* deconvolve_lucyrichardson_guo (merlin/util/deconvolve.py:136-174), what
 * DeconvolutionPreprocessGuo runs (analysis/preprocess.py:143-164): per
 * iteration two cv2.filter2D correlations (BORDER_REPLICATE) with the forward
 * and backward projectors of calculate_projectors (deconvolve.py:13-67).
 *   h_pf, h_pb  HOST f64 [window_size, window_size], window_size odd <= 13
 * Other arguments as mf_deconvolve_lucyrichardson (same workspace). */
int mf_deconvolve_lucyrichardson_guo(const void *image, int image_dtype,
                                     int n_images, int H, int W,
                                     int window_size, const double *h_pf,
                                     const double *h_pb, int iterations,
                                     void *out, int out_dtype,
                                     int64_t *histogram, int hist_group,
                                     int batch, void *workspace,
                                     int64_t workspace_bytes, void *stream);

/* DeconvolutionPreprocess._preprocess_image (analysis/preprocess.py:133-140)
 * for n_images camera frames: high-pass (highpass_sigma), Lucy-Richardson
 * (decon_filter_size, decon_sigma, decon_iterations), uint16 hand-over, plus
 * the pixel histogram of _run_analysis (:114-129) as above. */
int mf_preprocess_images(const uint16_t *image, int n_images, int H, int W,
                         double highpass_sigma, int decon_filter_size,
                         double decon_sigma, int decon_iterations,
                         uint16_t *out, int64_t *histogram, int hist_group,
                         int batch, void *workspace, int64_t workspace_bytes,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MERFISH_HIP_H */
