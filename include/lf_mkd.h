/*
 * lf_mkd.h -- C ABI of the MI355X-native MKD descriptor path (liblf_mkd.so).
 *
 * This is the drop-in boundary for ONE path of tnibler/local-features: the
 * "describe" half of LocalFeaturesVulkan::detect (local_features/src/vulkan/mod.rs:435-453),
 * i.e. the extract task graph (mod.rs:1277-1572):
 *   keypoint orientation -> patch sampling -> blur/gradients -> von-Mises x spatial-kernel
 *   pooling -> normalise -> PCA whitening -> L2.
 * Plain pointers and sizes only; no C++/torch types.  Each entry point cites the
 * reference interface it replaces.  The Rust-side binding a maintainer would add
 * is shown in INTEGRATION.md.
 *
 * Threading: a handle is NOT thread-safe (the reference takes &mut self on every
 * call, mod.rs:346-367); use one handle per device/stream.  Different handles may be
 * used from different host threads at the same time (one thread per handle).  All functions return
 * LF_MKD_OK (0) or a negative lf_mkd_status; they never abort.  The message for
 * the last failure on a handle is available from lf_mkd_last_error().
 *
 * Current device: every entry point makes the handle's device (lf_mkd_params.device) current
 * for its own duration and puts the calling thread's current HIP device back before it returns,
 * on every return path -- a process that drives one handle per GPU keeps the current device it
 * had (one hipGetDevice per call; hipSetDevice only when the two differ).
 */
#ifndef LF_MKD_H
#define LF_MKD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_MKD_PATCH_SIZE 32       /* lib.rs:15  PATCH_SIZE            */
#define LF_MKD_RAW_LEN 238         /* lib.rs:12  RAW_DESCRIPTOR_LEN    */
#define LF_MKD_DESC_LEN 128        /* lib.rs:13  DESCRIPTOR_LEN        */

typedef enum {
    LF_MKD_OK = 0,
    LF_MKD_ERR_BAD_ARG = -1,   /* null pointer, zero size, image larger than max_image_* ... */
    LF_MKD_ERR_HIP = -2,       /* a HIP runtime call failed; see lf_mkd_last_error         */
    LF_MKD_ERR_IO = -3,        /* cannot read / parse the PCA model file                   */
    LF_MKD_ERR_NO_IMAGE = -4,  /* describe_keypoints before set_image                      */
    LF_MKD_ERR_NO_DEVICE = -5, /* no usable gfx950 device                                  */
    LF_MKD_ERR_COMM = -6       /* RCCL is not loadable, or one of its calls failed         */
} lf_mkd_status;

/* Which PCA model to load from a model directory: enum MKDPCA, lib.rs:26-32. */
typedef enum { LF_MKD_PCA_LIBERTY = 0, LF_MKD_PCA_NOTREDAME = 1, LF_MKD_PCA_YOSEMITE = 2 } lf_mkd_pca;

/* Angle path of the gradient stage.
 * SHADER  : the reference's polynomial atan2 incl. its quirks (shaders/atan2.glsl:19-46). Default.
 *           (Domain: gx == -0.0 is gx == 0, as in the shader.  A gradient whose LARGER component is below 1e-30 in
 *           magnitude -- pixel values of that order; frames of [0, 1] have none -- gets the direction of
 *           min / max(larger, 1e-30) instead of min / larger: the hardware reciprocal has no denormal range.)
 * EXACT   : cos/sin of the gradient direction taken as gx/|g|, gy/|g| (what the polynomial
 *           approximates to 1e-5 rad; matches mkd_ref.rs:140, the CPU twin).
 * EXACT_ZERO : EXACT, except that a pixel with gx == 0 gets angle 0 as in the shader.  That is the one input
 *           where SHADER and EXACT are far apart (0 against +-pi/2), so this mode stays within 1e-4 relative L2 of
 *           the shader reference on EVERY patch (measured <= 3e-5) at EXACT's cost (about 11 % cheaper than SHADER). */
typedef enum { LF_MKD_ANGLE_SHADER = 0, LF_MKD_ANGLE_EXACT = 1, LF_MKD_ANGLE_EXACT_ZERO = 2 } lf_mkd_angle_mode;

/* Arithmetic of the pooling contraction (1024 px x 7 in-dims x 34 kernels per patch).
 * F16X3   : operands split into f16 hi+lo, three f16 MFMAs per product (hi*hi, hi*lo, lo*hi),
 *           f32 accumulate; ~2^-21 relative per product -- descriptors within 1e-5 relative L2 of the f32
 *           formulation (gate 1e-4).  THE DEFAULT (a zero-initialised lf_mkd_params selects it): ~250 M
 *           descriptors/s per MI355X in patch mode.
 * F32     : v_mfma_f32_16x16x4_f32, bit-for-bit an f32 fma chain; the verification mode, bound by the f32 MFMA
 *           rate at ~117 M descriptors/s (2.1x slower).
 * F16_FP6 : an experiment kept as a mode (round 4, NOTEBOOK.md section 11): hi*hi in f16 as above, the two cross terms of the
 *           harmonics' streams in ONE block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 per accumulator tile with e2m3
 *           operands (51 instead of 81 matrix instructions per wave-row).  e2m3 carries three bits below its block's
 *           maximum: descriptors within 3e-5 of the oracle (measured worst 2.95e-5, mean 2.0e-5, over the goldens and 4099
 *           patches x 3 angle modes x both kernel forms; F16X3: 4.2e-6 / 3.1e-6 -- seven times the error, inside the gate; the
 *           -m gpu test holds it below 4e-5) for +2 % of speed -- NOT the default, not used for any reported parity figure.  Patch mode only: keypoint
 *           entry points take the two-launch form in this mode. */
typedef enum { LF_MKD_POOL_DEFAULT = 0, LF_MKD_POOL_F16X3 = 1, LF_MKD_POOL_F32 = 2, LF_MKD_POOL_F16_FP6 = 3 } lf_mkd_pool_mode;

/* lf_mkd_params.flags */
#define LF_MKD_FLAG_KERNEL_TIMING 1u /* bracket every kernel launch with HIP events on its stream;
                                        read the sums with lf_mkd_kernel_times (bench.py's roofline) */
#define LF_MKD_FLAG_UNFUSED_KEYPOINTS 2u /* keypoint mode as two launches: the sampler writes the 32x32 patches to a staging
                                        buffer in HBM, the patch kernel reads them back (the verification form: same bits).
                                        By default patches are sampled inside the describe kernel -- producer waves of each
                                        workgroup fill its LDS row ring, patch_gradients.glsl:42-70 -- and never touch HBM.
                                        (LF_MKD_POOL_F32 always takes the two-launch form.) */

#define LF_MKD_FLAG_DETECT_STEPWISE 4u /* lf_mkd_detect / lf_mkd_detect_u8 stage by stage, every count fetched by the host before the
                                        next stage is sized (three waits): the verification form.  By default every count stays
                                        on the device, and the whole pipeline is ONE hipGraph launch, recorded the second time a
                                        (frame size, top_n, min_size, max_out, pixel type) is asked for and kept; same bits. */

/* Mirrors BuildTimeParams (lib.rs:54-75) + FeatureDetectParams (lib.rs:34-52) for this path.
 * Zero-initialise, then set what you need; 0 means "default". */
typedef struct {
    uint32_t max_image_width;   /* BuildTimeParams.max_image_width  (0: keypoint mode unused) */
    uint32_t max_image_height;  /* BuildTimeParams.max_image_height                            */
    uint32_t max_features;      /* BuildTimeParams.max_features: descriptors per internal batch;
                                   larger requests are processed in batches (default 2000 -> raised
                                   to a multiple of 64) */
    float patch_scale_factor;   /* FeatureDetectParams.patch_scale_factor (default 24)         */
    int32_t device;             /* HIP device ordinal                                          */
    int32_t angle_mode;         /* lf_mkd_angle_mode                                           */
    int32_t pool_mode;          /* lf_mkd_pool_mode (0 = LF_MKD_POOL_F16X3)                    */
    uint32_t flags;             /* LF_MKD_FLAG_*                                               */
    uint32_t max_frames;        /* frames of max_image_* size the pyramid store holds for the
                                   multi-frame entry points (default 1)                         */
    uint32_t n_scales;          /* BuildTimeParams.n_scales (lib.rs:60, default 4): keypoint orientation
                                   reads an a-trous stack of n_scales + 3 layers                */
    uint32_t max_blobs;         /* BuildTimeParams.max_blobs (lib.rs:58, default 8000): extrema the detector keeps
                                   per frame, rounded up to a multiple of 256 (mod.rs:279-286)   */
    uint32_t reserved[1];
} lf_mkd_params;

/* Keypoint as the path consumes it: struct Keypoint, lib.rs:17-24 (angle in DEGREES,
 * keypoint_orientation.glsl:162-167; response is carried through, not used). */
typedef struct {
    float x, y, size, angle, response;
} lf_mkd_keypoint;

/* Refined scale-space extremum as keypoint orientation reads it: the {x, y, scale, contrast} floats of
 * ExtremumLocations (shaders/common.glsl:45-81) for one index of FilteredExtrema (common.glsl:83-89),
 * gathered into an array of structs.  size = the interpolated scale written by refine_extrema. */
typedef struct {
    float x, y, size, response;
} lf_mkd_extremum;

#define LF_MKD_MAX_ANGLES_PER_EXTREMUM 18 /* strict local maxima of a circular 36-bin histogram */

typedef struct lf_mkd lf_mkd; /* opaque; owns all device memory */

/* Replaces new_vulkan() + upload_constant_data() for this path (lib.rs:94-100,
 * mod.rs:1587-1713).  PCA tensors are the three arrays of the reference's safetensors
 * model (mkd_ref.rs:352-391): mean[238], eigvals[238], eigvecs[238*238] row-major. */
int lf_mkd_create(const lf_mkd_params *params, const float *mean, const float *eigvals,
                  const float *eigvecs, lf_mkd **out);

/* Same, reading concat-pca-*.safetensors (models/mkd/, embedded by mkd_ref.rs:26-31). */
int lf_mkd_create_from_file(const lf_mkd_params *params, const char *safetensors_path,
                            lf_mkd **out);

void lf_mkd_destroy(lf_mkd *h);

const char *lf_mkd_last_error(const lf_mkd *h);

/* Patch mode: the extract graph from patch_gradients' blur onwards
 * (tasks_extract.rs:72-333; CPU twin Mkd::patch, mkd_ref.rs:57-77).
 * patches: [n][32][32] f32 row-major, out: [n][128] f32.  Host pointers; synchronous. */
int lf_mkd_describe_patches(lf_mkd *h, const float *patches, uint64_t n, float *out);

/* Same with DEVICE pointers, enqueued on `stream` (a hipStream_t, or NULL for the handle's
 * own stream); asynchronous.  This is what a caller that already holds patches in HBM uses. */
int lf_mkd_describe_patches_device(lf_mkd *h, const float *d_patches, uint64_t n, float *d_out,
                                   void *stream);

/* Debug/verification tap: un-whitened 238-D descriptor (RawDescriptorBuffer, common.glsl:133-139).
 * Device pointers, asynchronous. */
int lf_mkd_raw_descriptors_device(lf_mkd *h, const float *d_patches, uint64_t n, float *d_raw,
                                  void *stream);

/* Keypoint mode, step 1: upload one frame (f32 in [0,1], row-major, contiguous; mod.rs:368)
 * and build the patch pyramid (patch_pyramid.rs:38-156 + blur.glsl + swt.glsl level 0).
 * width/height must not exceed max_image_*; the pyramid mirrors at the content edge. */
int lf_mkd_set_image(lf_mkd *h, const float *image, uint32_t width, uint32_t height);
int lf_mkd_set_image_device(lf_mkd *h, const float *d_image, uint32_t width, uint32_t height,
                            void *stream);

/* The same from the 8-bit luma the reference's callers start from (`image::open(..).grayscale()`, then `convert()` to f32 / 255:
 * examples/match_images/src/main.rs:44-60; `u8 as f32 / 255.`: examples/webcam/src/main.rs:136): 1 byte per pixel over PCIe
 * instead of 4.  A pixel v becomes (float)v / 255.0f with a correctly rounded division on the device, i.e. the f32 frame the
 * host conversion gives, bit for bit: every result equals that of lf_mkd_set_image on the converted frame.  Row-major,
 * contiguous, width bytes per row. */
int lf_mkd_set_image_u8(lf_mkd *h, const uint8_t *image, uint32_t width, uint32_t height);
/* Device-pointer form, n_frames frames of one size, width * height bytes apart (n_frames <= max_frames). */
int lf_mkd_set_images_u8_device(lf_mkd *h, const uint8_t *d_images, uint32_t n_frames, uint32_t width, uint32_t height,
                                void *stream);

/* Multi-frame form of step 1 (BASELINE configs[2], [3]: hundreds of small frames): n_frames frames of one
 * size, contiguous [n_frames][height][width] in device memory, all pyramids built by one set of launches.
 * n_frames <= max_frames. */
int lf_mkd_set_images_device(lf_mkd *h, const float *d_images, uint32_t n_frames, uint32_t width,
                             uint32_t height, void *stream);

/* Keypoint mode, step 2: sample + describe (patch_gradients.glsl:42-70 onwards).
 * out: [n][128].  Host pointers; synchronous.
 * Forms of the launch, chosen by the size of the request (the capacity max_out for lf_mkd_detect* and lf_mkd_stream_*):
 *   whole-patch   one wave walks the 32 rows of its 16 patches and sums them in one chain: any size; ~72 us of latency
 *                 however few the keypoints;
 *   row-split     at most 4096 keypoints (16 x the chip's CUs; the reference's own settings are top_n 2000 / max_features
 *                 3000): the rows of a batch of 32 patches are shared by 4 workgroups (up to 2048 keypoints) or 2, whose
 *                 partial sums are added in a fixed order -- 43 us at 2000 keypoints, 50 at 3000.
 * Within a form a descriptor's bits depend on its keypoint and its frame alone (whatever else is in the request, however
 * often it is computed).  Between forms the pooled sums round differently: descriptors agree to ~3e-6 relative L2 (worst
 * measured 6.5e-6 over 32 000 soak launches; the tests hold 1e-5), both within the same distance of the reference.  LF_MKD_KP_SPLIT=1 in the
 * environment keeps every request in the whole-patch form (2 / 4: that row-split form wherever it fits).  (In the row-split
 * form a workgroup waits for its partners' partial sums; the wait is bounded, and a partial sum that never arrived -- a fault,
 * never observed -- is counted: lf_mkd_describe_keypoints then returns LF_MKD_ERR_HIP.) */
int lf_mkd_describe_keypoints(lf_mkd *h, const lf_mkd_keypoint *kps, uint64_t n, float *out);
int lf_mkd_describe_keypoints_device(lf_mkd *h, const lf_mkd_keypoint *d_kps, uint64_t n,
                                     float *d_out, void *stream);

/* Multi-frame form of step 2: d_frame_of_kp[i] (device, may be NULL = all frame 0) names the frame of
 * keypoint i among those given to lf_mkd_set_images_device.  One sampling launch + one describe launch
 * per internal batch, whatever the number of frames. */
int lf_mkd_describe_keypoints_frames_device(lf_mkd *h, const lf_mkd_keypoint *d_kps,
                                            const uint32_t *d_frame_of_kp, uint64_t n, float *d_out,
                                            void *stream);

/* Keypoint orientation: the first node of the extract graph (keypoint_orientation.glsl:36-171,
 * dispatched from mod.rs:1277-1344), i.e. what turns the detector's extrema into the keypoints the
 * describe entry points take.  Needs lf_mkd_set_image*: the first call after it extends the frame's
 * sigma-0.6 level into the a-trous stack of n_scales + 3 full-resolution layers (swt.glsl, mod.rs:1093-1130).
 * Every histogram peak >= 0.8 max yields one keypoint {x, y, size, angle = 360 - 10 bin, response}.
 * Output ORDER is defined here (the reference appends with an atomic counter, in no particular order):
 * by extremum index, then by ascending histogram bin.  At most max_out keypoints are written; *n_out
 * receives the number written and *n_dropped (may be NULL) how many more there were (the reference's
 * dropped_features, mod.rs:585).  Host pointers; synchronous. */
int lf_mkd_orient_keypoints(lf_mkd *h, const lf_mkd_extremum *extrema, uint64_t n,
                            lf_mkd_keypoint *out, uint64_t max_out, uint64_t *n_out,
                            uint64_t *n_dropped);

/* Same with DEVICE arrays.  d_frame_of_extremum (may be NULL = frame 0) names each extremum's frame
 * among those given to lf_mkd_set_images_device; d_frame_of_kp (may be NULL) receives the frame of
 * every keypoint written, ready for lf_mkd_describe_keypoints_frames_device.  n_out / n_dropped are
 * HOST pointers: the call waits for `stream` before returning so that the count is valid. */
int lf_mkd_orient_keypoints_device(lf_mkd *h, const lf_mkd_extremum *d_extrema,
                                   const uint32_t *d_frame_of_extremum, uint64_t n,
                                   lf_mkd_keypoint *d_out, uint32_t *d_frame_of_kp, uint64_t max_out,
                                   uint64_t *n_out, uint64_t *n_dropped, void *stream);

/* Detector: the detect task graph after the a-trous stack (swt_sub.glsl, scan_extrema.glsl; constants
 * border = 5, contrast threshold 0.035, skip_layers = 0: mod.rs:76,395-407).  Needs lf_mkd_set_image*.
 * Scans the DoG volume of every loaded frame for 3-D extrema, refines them and applies the edge test;
 * writes {x + dx, y + dy, size, contrast} in a DEFINED order (the reference appends atomically): by frame,
 * then 4x4x4 scan cube in raster order (z, y, x), then position x + 4 y + 16 z in the cube.  A cube keeps at
 * most 8 candidates like the reference, here the first 8 in that order.  At most max_out are written;
 * *n_out = written, *n_dropped (may be NULL) = found beyond max_out (dropped_blobs, mod.rs:625-633).
 * d_frame_of (may be NULL) receives each extremum's frame.  n_out / n_dropped are HOST pointers: waits for
 * `stream`. */
int lf_mkd_detect_extrema_device(lf_mkd *h, lf_mkd_extremum *d_out, uint32_t *d_frame_of, uint64_t max_out,
                                 uint64_t *n_out, uint64_t *n_dropped, void *stream);
int lf_mkd_detect_extrema(lf_mkd *h, lf_mkd_extremum *out, uint64_t max_out, uint64_t *n_out,
                          uint64_t *n_dropped);

/* The host blob filter of detect_top_n on the device (TopKContrastFilter, mod.rs:1753-1786): of the n
 * extrema of ONE frame keep those with size >= min_size and, if more than top_n remain, the top_n with the
 * largest contrast (ties at the cut resolved in index order); index order is preserved.  d_out [top_n]
 * receives the kept extrema, d_index (may be NULL) their indices into d_extrema.  *n_out on the host. */
int lf_mkd_filter_extrema_device(lf_mkd *h, const lf_mkd_extremum *d_extrema, uint64_t n, uint32_t top_n,
                                 float min_size, lf_mkd_extremum *d_out, uint32_t *d_index, uint64_t *n_out,
                                 void *stream);

/* LocalFeaturesVulkan::detect / detect_top_n (mod.rs:346-593) in one call, host pointers, synchronous:
 * image -> pyramid + a-trous stack -> extrema (at most max_blobs) -> [top_n filter if top_n > 0] ->
 * orientation -> sampling -> descriptors.  keypoints [max_out] and descriptors [max_out][128] receive
 * *n_out <= max_out results; *dropped_blobs and *dropped_features (may be NULL) as FeaturesResult
 * (lib.rs:77-83).  max_out beyond 18 x (top_n, or max_blobs when top_n == 0) -- more keypoints than can exist -- is treated as
 * that bound: it sizes no buffer and no copy.  The whole pipeline is one hipGraph launch, recorded the SECOND time these
 * arguments' frame size, top_n, min_size, max_out and pixel type are seen (up to 8 such recordings are kept per handle, the
 * least recently used one makes room): a replayed call costs the upload of the frame, the pipeline and the copy of *n_out
 * results, with one wait in between.  The first sighting of a request is served by the same launches without recording them
 * (one upload, the pipeline's ~20 launches, one wait; same bits) -- the reference's match_images detects each image once, at its
 * own size, and never pays a recording; a camera loop pays it on its second frame (LF_MKD_DETECT_RECORD_AFTER=k in the
 * environment at lf_mkd_create: k sightings before recording, 0 = record at once; cost of the three kinds of call:
 * INTEGRATION.md section 2).  Handles whose keypoint mode takes the two-launch form (LF_MKD_POOL_F32, LF_MKD_POOL_F16_FP6,
 * LF_MKD_FLAG_UNFUSED_KEYPOINTS) always take the stage-by-stage form here.  A frame of 6 MB or more crosses PCIe in pieces
 * (two to four, planned from a model of the link and of the pipeline's front when the request is recorded), and the front --
 * level 0, the a-trous layers, the extremum scan -- runs on the rows a piece completes while the next one is on its way
 * (LF_MKD_DETECT_BANDS=0 in the environment: one piece; LF_MKD_BAND_PIECES=k / LF_MKD_BAND_SPLIT=f1,f2,..: k equal pieces /
 * cuts at these fractions of the height, for tests); same results bit for bit.  Afterwards the handle holds the frame like
 * lf_mkd_set_image.  (benches/bench.rs on houses.jpg, 4096 x 3072, top 2000, n_scales 3: 1.25 ms, 0.91 of it the 50 MB upload
 * at the link's 56 GB/s; lf_mkd_detect_u8 on the same frame: 0.66 ms; n_scales 5: 1.30 / 0.80 ms.) */
int lf_mkd_detect(lf_mkd *h, const float *image, uint32_t width, uint32_t height, uint32_t top_n,
                  float min_size, lf_mkd_keypoint *keypoints, float *descriptors, uint64_t max_out,
                  uint64_t *n_out, uint64_t *dropped_blobs, uint64_t *dropped_features);

/* Host-only diagnostic (needs no device): where lf_mkd_detect / lf_mkd_detect_u8 would cut a frame of this size for its banded
 * upload -- the rows after which a piece ends, ascending, *n_cuts of them (0: one piece; at most max_cuts are written to cuts) --
 * and what the plan's model says: the time (us, from the start of the first copy) at which the pipeline's front has run on the
 * whole frame with these pieces and with one.  bytes_per_pixel 4 (f32 frame) or 1 (8-bit); n_scales as in lf_mkd_params (0: 4).
 * Honours LF_MKD_DETECT_BANDS / LF_MKD_BAND_SPLIT / LF_MKD_BAND_PIECES like the call itself.  modelled_us / one_piece_us may be NULL. */
int lf_mkd_plan_upload(uint32_t width, uint32_t height, uint32_t bytes_per_pixel, uint32_t n_scales, uint32_t *cuts,
                       uint32_t max_cuts, uint32_t *n_cuts, double *modelled_us, double *one_piece_us);

/* lf_mkd_detect on an 8-bit frame (see lf_mkd_set_image_u8): same results, a quarter of the upload. */
int lf_mkd_detect_u8(lf_mkd *h, const uint8_t *image, uint32_t width, uint32_t height, uint32_t top_n, float min_size,
                     lf_mkd_keypoint *keypoints, float *descriptors, uint64_t max_out, uint64_t *n_out,
                     uint64_t *dropped_blobs, uint64_t *dropped_features);

/* detect_top_n over a BATCH of frames (BASELINE configs[2]: hundreds of small frames): n_frames frames of one size,
 * contiguous in device memory, through the whole pipeline with every stage launched once for all frames --
 * pyramids, a-trous stacks, extremum scan, per-frame top_n filter (top_n = 0: every extremum, at most max_blobs
 * per frame), orientation, sampling, description.  Results are ordered by frame, then as lf_mkd_detect orders
 * them; d_frame_of_kp [max_out] names each keypoint's frame.  *dropped_blobs sums the extrema beyond max_blobs
 * over the frames; *dropped_features counts keypoints beyond max_out (a budget for the whole batch).
 * n_frames <= max_frames.  Counts come back to the host (the call waits for `stream`). */
int lf_mkd_detect_frames_device(lf_mkd *h, const float *d_images, uint32_t n_frames, uint32_t width,
                                uint32_t height, uint32_t top_n, float min_size, lf_mkd_keypoint *d_keypoints,
                                uint32_t *d_frame_of_kp, float *d_descriptors, uint64_t max_out,
                                uint64_t *n_out, uint64_t *dropped_blobs, uint64_t *dropped_features,
                                void *stream);

/* Per-frame pipeline as one hipGraph (BASELINE configs[4]: 4K stream, detect + describe per frame).
 * lf_mkd_stream_create records the launch sequence of lf_mkd_detect for frames of width x height -- pyramid,
 * a-trous stack, extremum scan, [top_n filter if top_n > 0], orientation, sampling, description -- with every
 * count handed from stage to stage in device memory, so a frame needs no host round trip.  The buffers are
 * fixed at creation, all DEVICE memory owned by the caller: d_image [height][width] f32 (write the next frame
 * there before each launch), d_keypoints [max_out], d_descriptors [max_out][128], d_counts [8] uint64:
 * [0] extrema found (capped at max_blobs), [1] dropped_blobs, [2] extrema after the top_n filter,
 * [3] keypoints written (valid rows of the two outputs), [4] dropped_features.
 * lf_mkd_stream_frame launches the graph on `stream` (NULL: the handle's stream), asynchronously.
 * One stream pipeline per handle; creating another replaces it.  Creating one also discards the frame loaded by
 * lf_mkd_set_image*: until the first lf_mkd_stream_frame the keypoint, orientation and verification entry points return
 * LF_MKD_ERR_NO_IMAGE; after it they see the frame the pipeline last processed (on the stream it was launched on). */
int lf_mkd_stream_create(lf_mkd *h, uint32_t width, uint32_t height, uint32_t top_n, float min_size,
                         uint64_t max_out, const float *d_image, lf_mkd_keypoint *d_keypoints,
                         float *d_descriptors, uint64_t *d_counts);
int lf_mkd_stream_frame(lf_mkd *h, void *stream);

/* Brute-force matcher: match_features of examples/match_images/src/main.rs:8-27.  For every row of a [na][128]:
 * similarity = dot product with every row of b [nb][128]; best = the largest (the HIGHEST index among equal maxima,
 * as the reference's stable sort leaves it), second = the next one down; match[i] = index of the best if
 * best * ratio > second (the reference uses ratio = 0.8), else -1; ratio <= 0 skips the test and returns the best
 * index as is.  d_best / d_second (may be NULL) receive the two similarities, with which a caller can apply another
 * acceptance rule -- e.g. the webcam example's inner-product-distance form, (1 - best) < 0.75 (1 - second)
 * (examples/webcam/src/main.rs:261-265).  d_exclude_lo / d_exclude_hi (may both be NULL): b rows [lo[i], hi[i]) are not candidates for
 * a row i -- the cross-image form of BASELINE configs[3], where b is the all-gathered descriptor set and a row
 * must not match its own image.  nb must be at least 2 (the reference indexes the second-to-last candidate).
 * Similarities come from the matrix cores in one of three forms, all within ~1e-7 of an f32 dot product, so that decisions
 * can differ from the reference's only where two similarities, or best*ratio and second, agree to that level:
 *   small   -- na * nb <= 2^23 and nb <= 4096 (the reference's own 2000 x 2000): ONE launch straight from the f32 rows, the
 *              scan's three terms formed in registers, no operand tiles and no scratch buffer;
 *   scan    -- every pair from f16 hi+lo splits of both sides (three MFMA terms; ~2^-21 relative for rows of unit norm or
 *              larger -- the lo parts of much smaller elements fall below the f16 grid); ~1.7e12 pairs/s on an MI355X;
 *              used for mid-sized problems;
 *   screen  -- two passes, for na >= 16384 and na * nb >= 2^29: every pair is screened with the f16 roundings of both sides
 *              (one term; error bounded by ~1e-3 |a||b|, from the rows' norms), every candidate within that bound of a row's
 *              second best is re-scored as an f32 dot product, and the decision is taken on the re-scored values, i.e. the
 *              result of an exhaustive f32 scan; ~5.9e12 pairs/s.  A row with more than 64 such candidates in one lane's
 *              share of b (hundreds of near-duplicates of its best match) is redone by the scan form inside the same call,
 *              decided on the device.
 * LF_MKD_MATCH=small, =scan or =screen in the environment forces a form (small: where it fits -- also nb >= 128 or na <= 4096 --
 * else scan).  Elements must be finite and below 65504 in magnitude (f16 range).
 * Device pointers, asynchronous on `stream`.  d_a and d_b must be 16-byte aligned (rows are read as 16-byte vectors; any
 * hipMalloc'd array or row offset into one is: a row is 512 bytes). */
int lf_mkd_match_device(lf_mkd *h, const float *d_a, uint64_t na, const float *d_b, uint64_t nb,
                        const uint32_t *d_exclude_lo, const uint32_t *d_exclude_hi, float ratio,
                        int32_t *d_match, float *d_best, float *d_second, void *stream);
/* Both directions of the reference's example in one call (examples/match_images/src/main.rs:113-116 matches image 1 against
 * image 2 and image 2 against image 1): d_match_ab [na] as lf_mkd_match_device(a, b) gives it, d_match_ba [nb] as
 * lf_mkd_match_device(b, a) does -- decision for decision.  Where both directions fit the one-launch form (the example's own
 * 2000 x 2000) they ARE one launch: the second direction costs no second launch.  Device pointers (16-byte aligned),
 * asynchronous on `stream`.  An empty side (na == 0 or nb == 0) is LF_MKD_OK and writes nothing, as lf_mkd_match_device with
 * na == 0; otherwise na, nb >= 2.  lf_mkd_match_overflowed afterwards reports the rows redone over BOTH directions. */
int lf_mkd_match_both_device(lf_mkd *h, const float *d_a, uint64_t na, const float *d_b, uint64_t nb, float ratio,
                             int32_t *d_match_ab, int32_t *d_match_ba, void *stream);
/* Host pointers, synchronous. */
int lf_mkd_match(lf_mkd *h, const float *a, uint64_t na, const float *b, uint64_t nb, float ratio,
                 int32_t *match);
/* Diagnostic: *n_rows = rows of a that the handle's latest match call had to redo by the full scan (0 in the ordinary
 * case).  Waits for that call to finish (synchronises `stream`, NULL = the handle's own). */
int lf_mkd_match_overflowed(lf_mkd *h, void *stream, uint64_t *n_rows);

/* ---- multi-GPU: the path's ONE collective (BASELINE configs[3]) -------------------------------------------------
 * Keypoint batches shard by image, one process and one handle per GPU, and nothing is exchanged while describing.  The
 * cross-image match stage needs every rank's descriptors on every rank: an all-gather of the descriptor shards over RCCL
 * (xGMI).  The reference has no counterpart (one device, one queue: vulkan/make_a_vulkan.rs:80-115); the Rust crate calls
 * these from LocalFeaturesHip::cross_image_match (bindings/rust/).  librccl is loaded when the first of these functions
 * is called (dlopen: a process that has torch's copy loaded gets that one), so the library itself does not depend on it.
 *
 * lf_mkd_comm_unique_id   rank 0 draws the identifier (ncclGetUniqueId) and ships its LF_MKD_COMM_ID_BYTES to the other
 *                         ranks by whatever channel the application has (a file, a socket, MPI, torch.distributed).
 * lf_mkd_comm_create      every rank, with the same identifier: ncclCommInitRank on the handle's device.  Collective.
 * lf_mkd_allgather_descriptors
 *     d_buf   [sum(counts)][128] f32 on the handle's device: the gathered set, rank r's rows at offset sum(counts[0..r));
 *             this rank's own rows are already in place (a producer that writes its descriptors straight there makes
 *             the gather copy-free on the sending side too).
 *     counts  [n_ranks] descriptors held by every rank (host array; shards may differ in size).
 *     mode    LF_MKD_GATHER_DIRECT: one group of point-to-point transfers (ncclGroupStart .. ncclSend / ncclRecv to and from
 *             every peer .. ncclGroupEnd): xGMI is a full point-to-point mesh, so a rank's n-1 sends leave on n-1 links at
 *             once.  LF_MKD_GATHER_RING: one ncclAllGather (in place; needs equal shards, else DIRECT is used).
 *     Asynchronous on `stream` (NULL: the handle's own).  Collective: every rank calls it with the same counts and mode.
 * lf_mkd_comm_info        RCCL's version code, this communicator's size and rank (any pointer may be NULL). */
#define LF_MKD_COMM_ID_BYTES 128
#define LF_MKD_GATHER_DIRECT 0
#define LF_MKD_GATHER_RING 1
typedef struct lf_mkd_comm lf_mkd_comm;
int lf_mkd_comm_unique_id(uint8_t *id);
int lf_mkd_comm_create(lf_mkd *h, const uint8_t *id, int32_t n_ranks, int32_t rank, lf_mkd_comm **out);
int lf_mkd_comm_destroy(lf_mkd_comm *c);
int lf_mkd_comm_info(const lf_mkd_comm *c, int32_t *rccl_version, int32_t *n_ranks, int32_t *rank);
int lf_mkd_allgather_descriptors(lf_mkd *h, lf_mkd_comm *c, const uint64_t *counts, float *d_buf, int32_t mode,
                                 void *stream);
/* Diagnostics of the gather.
 * lf_mkd_comm_loopback: one group with a send of n_rows rows of 128 f32 from d_src to THIS rank and the matching receive into
 *     d_dst (ncclGroupStart, ncclSend, ncclRecv, ncclGroupEnd; n_rows == 0 posts the empty group) -- through the same binding
 *     and the same posting routine as the DIRECT form, whose loop is empty on a one-rank communicator: what lets one process
 *     check that the point-to-point branch works against the librccl it finds, before a multi-rank job depends on it.
 *     d_src and d_dst are DEVICE arrays that must not overlap.  Asynchronous on `stream` (NULL: the handle's own).
 * lf_mkd_comm_last_form: LF_MKD_GATHER_DIRECT or LF_MKD_GATHER_RING, whichever the communicator's latest
 *     lf_mkd_allgather_descriptors took (RING falls back to DIRECT on unequal shards); -1 before the first gather. */
int lf_mkd_comm_loopback(lf_mkd *h, lf_mkd_comm *c, const float *d_src, float *d_dst, uint64_t n_rows, void *stream);
int lf_mkd_comm_last_form(const lf_mkd_comm *c);

/* The same stage on the reference's own buffer formats, for a caller that keeps the reference's detect graph and host
 * filter and swaps only the extract graph.  Host pointers; synchronous.
 *   extremum_data  ExtremumLocations.data: blocks of block_len (256, mod.rs:279) extrema laid out
 *                  [x * block_len][y * block_len][size * block_len][contrast * block_len] f32
 *                  (common.glsl:45-81 `_coord_idx`, host mirror shaders.rs:257-319), n_extrema entries in all;
 *   indices        FilteredExtrema.indices (common.glsl:83-89): the extrema the host filter kept, n_indices of them.
 * Outputs as KeypointIndices holds them (common.glsl:93-101, shaders.rs:321-353): kp_extremum_index[i] = index of
 * keypoint i's extremum (into extremum_data, i.e. one of the values in `indices`) and kp_orientation[i] in degrees,
 * for *n_out <= max_out keypoints, ordered by position in `indices`, then histogram bin.  keypoints (may be NULL) additionally receives
 * the assembled lf_mkd_keypoint records the describe entry points take. */
int lf_mkd_orient_keypoints_blocked(lf_mkd *h, const float *extremum_data, uint64_t n_extrema, uint32_t block_len,
                                    const uint32_t *indices, uint64_t n_indices, uint32_t *kp_extremum_index,
                                    float *kp_orientation, lf_mkd_keypoint *keypoints, uint64_t max_out,
                                    uint64_t *n_out, uint64_t *n_dropped);

/* Verification tap: copies layer `layer` (0 .. n_scales + 2) of frame 0's a-trous stack to a host
 * buffer of width x height floats, building the stack first if needed. */
int lf_mkd_get_coarse_layer(lf_mkd *h, uint32_t layer, float *out);

/* Verification taps for keypoint mode (device pointers). */
int lf_mkd_sample_patches_device(lf_mkd *h, const lf_mkd_keypoint *d_kps, uint64_t n,
                                 float *d_patches, void *stream);
/* Copies pyramid level `level` ((w>>level) x (h>>level) f32) to a host buffer. */
int lf_mkd_get_pyramid_level(lf_mkd *h, uint32_t level, float *out, uint32_t *w, uint32_t *hgt);
/* The same level as the sampler addresses it: with its apron of *apron texels on every side (48 on EVERY level, level 0 included: size
 * the buffer from the returned *apron, never from a constant), which holds
 * what MirroredRepeat addressing (mod.rs:940-943) would fetch there -- (hgt + 2 apron) rows of (w + 2 apron) floats.
 * Any of out / w / hgt / apron may be NULL. */
int lf_mkd_get_pyramid_level_apron(lf_mkd *h, uint32_t level, float *out, uint32_t *w, uint32_t *hgt, uint32_t *apron);

/* Host-only verification tap: the constants upload_constant_data (mod.rs:1587-1713) would place in
 * ConstantData (common.glsl:34-40), as this library builds them.  Any output pointer may be NULL.
 * gradient_angle[1024], embedding_polar[25*1024], embedding_cartesian[9*1024], w_t[128*238].
 * Needs no device. */
int lf_mkd_build_constants(const float *mean, const float *eigvals, const float *eigvecs,
                           float *gradient_angle, float *embedding_polar,
                           float *embedding_cartesian, float *w_t);

/* With LF_MKD_FLAG_KERNEL_TIMING: waits for the recorded launches, returns the summed device time
 * (ms) of the describe kernel (pool_ms; whiten_ms is 0 since the whitening stage was fused into it)
 * and the number of batches since the previous call, then resets the sums.  Any output pointer may
 * be NULL.  Covers the describe launches the patch and keypoint entry points make (and a stage-by-stage
 * lf_mkd_detect); a REPLAYED lf_mkd_detect / lf_mkd_stream_frame launches its describe node inside the
 * hipGraph, without events: lf_mkd_detect_times covers those. */
int lf_mkd_kernel_times(lf_mkd *h, double *pool_ms, double *whiten_ms, uint64_t *launches);

/* With LF_MKD_FLAG_KERNEL_TIMING: where the handle's latest lf_mkd_detect / lf_mkd_detect_u8 call spent its time -- the upload of
 * the frame and the recorded pipeline (HIP events on the handle's stream around each), and the wall time from the pipeline's
 * end to the return (the copy of the results to the caller's arrays).  Any output pointer may be NULL. */
int lf_mkd_detect_times(lf_mkd *h, double *upload_ms, double *pipeline_ms, double *readback_ms);

/* With LF_MKD_FLAG_KERNEL_TIMING: the shader clock the chip sustained during the handle's latest describe launch, from
 * stamps workgroup 0 of the kernel leaves on entry and exit (shader-clock counter / constant 100 MHz counter), and that
 * workgroup's lifetime in ms.  The chip lowers its clock under load, box by box: this is what makes two boxes' figures
 * comparable.  Waits for `stream` (NULL: the handle's own).  Either output pointer may be NULL. */
int lf_mkd_kernel_clock(lf_mkd *h, void *stream, double *shader_mhz, double *kernel_ms);

/* Diagnostic of lf_mkd_detect / lf_mkd_detect_u8 (host state only, no device work): how many recorded pipelines the handle holds
 * at the moment (at most 8), how many of them upload their frame in pieces, and how many distinct requests it remembers having
 * seen (at most 64; a request is recorded on its second sighting).  Any output pointer may be NULL. */
int lf_mkd_detect_recordings(const lf_mkd *h, uint32_t *n_recordings, uint32_t *n_banded, uint32_t *n_sightings);

/* Blocks until everything enqueued on the handle's own stream has finished. */
int lf_mkd_synchronize(lf_mkd *h);

/* Library identification: "lf_mkd <version> gfx950". */
const char *lf_mkd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LF_MKD_H */
