// Launch interface between the C ABI (lf_mkd.cpp) and the gfx950 kernels (mkd_describe.hip, mkd_pyramid.hip,
// mkd_orient.hip, mkd_detect.hip, mkd_match.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <stdint.h>

#define LF_ANGLE_SHADER 0
#define LF_ANGLE_EXACT 1
#define LF_ANGLE_EXACT_ZERO 2   // exact direction, but angle 0 where gx == 0 like the shader
#define LF_POOL_F16X3 1         // = LF_MKD_POOL_F16X3 (lf_mkd.h); the ABI's 0 ("default") is mapped to it at creation
#define LF_POOL_F32 2           // = LF_MKD_POOL_F32
#define LF_POOL_F16_FP6 3       // = LF_MKD_POOL_F16_FP6

namespace lfmkd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxPyrLevels = 16;

// Patch pyramid: levels packed back to back in one allocation; level l is w[l] x h[l] f32, rows pitch[l] floats apart.
// Every level carries an APRON of kPyrApron texels on every side, filled with the level's own texels under MirroredRepeat
// (mod.rs:940-943): a keypoint whose centre lies inside its level reaches at most 22.7 rem + 2 < 48 texels from it
// (rem < 2), so its whole footprint is addressed without any mirror arithmetic.  offset[l] is texel (0, 0).  (Level 0 is also
// a-trous layer 0 of the detector and the orientation stage: they read it with its pitch.)
constexpr int kPyrApron = 48;
struct PyramidDesc {
    int levels;
    int w[kMaxPyrLevels], h[kMaxPyrLevels];
    int pitch[kMaxPyrLevels];    // floats between rows
    int apron[kMaxPyrLevels];    // mirrored texels around the level
    long offset[kMaxPyrLevels];  // in floats, of texel (0, 0)
};

// Row bands (lf_mkd_detect's upload / compute overlap; two pieces in round 5, any number since round 6): the frame reaches the
// device in pieces, and the row-tiled kernels of the pipeline's front -- level 0, the a-trous layers, the extremum scan -- run
// once per piece on the rows that piece completes, while the next piece is still on its way over PCIe.  A launch sequence is
// given the rows [lo, hi) of every stage it is to produce: hi = what the frame's rows so far allow (plan_row_bands: level 0
// needs two raw rows below an output row, a-trous layer l + 1 with dilation d = 2^l needs 2 d rows of layer l, the scan one row
// of every layer below a tile's candidates), lo = the previous piece's hi.  The last piece has hi = everything and also runs
// what is not row-tiled.  Splits fall on multiples of 12 rows for level 0, of d rows for a layer of dilation d, of a tile row
// (8 candidate rows) for the scan; no piece reads a row a later piece produces.
struct RowBands {
    int last;             // 1: the final piece -- every stage to its end, then the rest of the pipeline
    int level0_lo, level0_hi;       // pyramid level 0 = a-trous layer 0: rows (multiples of 12; hi of the last piece: the height)
    int layer_lo[8], layer_hi[8];   // a-trous layer l + 1 (dilation 2^l): rows (multiples of 2^l)
    int scan_lo, scan_hi;           // extremum scan: tile rows (8 candidate rows each)
};
// the `hi` fields for a frame of which the first raw_rows rows have arrived (`lo` and `last` are the caller's); false: the
// frame's shape does not take row bands (the staged kernels' alignment), or raw_rows covers the frame
bool plan_row_bands(int raw_rows, int w, int h, int n_layers, int border, RowBands &bands);

// Device copies of HostConsts' device layouts (mkd_consts.hpp).
struct DeviceConsts {
    short *colmap = nullptr;        // [336]
    float *pool_b_f32 = nullptr;    // [32][15][2][64][4]
    uint16_t *pool_b_f16 = nullptr; // [32][15][2][64][8]
    uint16_t *pool_b_fp6 = nullptr; // the same row images with fp6 cross-term operands for the harmonics' tiles (mkd_consts.hpp)
    uint16_t *white_a_f16 = nullptr; // [11][8][2][64][8]
    float *white_a_f32 = nullptr;    // [21][4][8][64]
    float *white_bias = nullptr;     // [128]  -W mean
};

// patches [n][32][32] -> out [n][128] (and, when raw_out != nullptr, the un-whitened [n][238])
// (n_dev != nullptr: the count is read on the device, n only sizes the grid; same for the launchers below)
// waves: 0 = choose by size (4-wave workgroups when the request fits one round of them, else 8-wave), 4 / 8 = that form
void launch_describe(const float *patches, long n, const unsigned long long *n_dev, const DeviceConsts &dc,
                     int angle_mode, int pool_mode, float *out, float *raw_out, int num_cus, hipStream_t stream,
                     int waves = 0, unsigned long long *clk = nullptr);
// clk (nullable, device, 4 words): workgroup 0 leaves (shader clock, 100 MHz clock) on entry and on exit
// frame_of_kp == nullptr: every keypoint belongs to frame 0
// (frame indices beyond n_frames - 1 -- caller's data -- are clamped to it: never a read outside the pyramid store)
void launch_sample_patches(const float *pyr, long pyr_stride, const PyramidDesc &pd, const float *kps,
                           const unsigned *frame_of_kp, unsigned n_frames, long n, const unsigned long long *n_dev, float psf,
                           float *patches, hipStream_t stream);
// keypoint mode in ONE launch (csrc/mkd_describe.hip): patches are sampled by producer waves of the describe workgroup
// straight into its LDS row ring and never touch HBM.  f16x3 pooling only.
void launch_describe_keypoints(const float *pyr, long pyr_stride, const PyramidDesc &pd, const float *kps,
                               const unsigned *frame_of_kp, unsigned n_frames, long n, const unsigned long long *n_dev,
                               float psf, const DeviceConsts &dc, int angle_mode, float *out, int num_cus, hipStream_t stream,
                               unsigned long long *clk = nullptr, float *xchg = nullptr, unsigned *xchg_words = nullptr,
                               long form_n = 0);
// form_n (0: n): the request size the choice between the whole-patch and the row-split forms follows, when it is not n
// xchg / xchg_words (nullable: the whole-patch forms only): scratch of the row-split form a request of at most 16 x CUs
// keypoints takes -- kp_split_exchange_bytes(num_cus) bytes and kp_split_counter_words(num_cus) u32 words, the words zero
// before the first launch (they return to zero at the end of every launch; word 0 counts partial sums that never arrived)
size_t kp_split_exchange_bytes(int num_cus);
size_t kp_split_counter_words(int num_cus);
// rest_stream (nullable): levels >= 1 are built there -- after `fork`, recorded on `stream` once level 0 and a-trous layer 1
// exist -- and `join` is recorded behind them; the caller waits for `join` before it samples patches
void launch_build_pyramid(const float *image, long image_stride, float *pyr, long pyr_stride, float *tmp_a,
                          float *tmp_b, const PyramidDesc &pd, int frames, float *layer1, long layer1_stride,
                          hipStream_t stream, hipStream_t rest_stream = nullptr, hipEvent_t fork = nullptr,
                          hipEvent_t join = nullptr, const std::function<void()> &main_next = {},
                          const unsigned char *image_u8 = nullptr, const RowBands *bands = nullptr);
// (image_u8 != nullptr: the frames are 8-bit luma, image_stride BYTES apart, `image` is ignored; a pixel is (float)v / 255.0f)

// a-trous layers 1 .. n_layers-1 over layer0 (= pyramid level 0); tmp holds frames x w x h floats
void launch_build_coarse_stack(const float *layer0, long layer0_stride, int layer0_pitch, float *coarse, long coarse_stride,
                               long layer_stride, float *tmp, int n_layers, int first_layer, int w, int h, int frames,
                               hipStream_t stream, const RowBands *bands = nullptr);
// extrema [n][4] -> kps [<= max_out][5] ordered by extremum then bin; angles [n][18], counts [n], sums [n/1024+1]
// are scratch (sums may be null for n <= 8192);
// totals[0] = written, totals[1] = dropped
void launch_orient(const float *layer0, long layer0_stride, int layer0_pitch, const float *coarse, long coarse_stride, long layer_stride,
                   int n_layers, int w, int h, const float *extrema, const unsigned *frame_of, long n,
                   const unsigned long long *n_dev, float *angles, unsigned *counts, unsigned *sums, float *kps,
                   unsigned *frame_of_kp, unsigned long long max_out, unsigned long long *totals, hipStream_t stream);

// cubes of the extremum scan for a w x h frame with n_fine DoG layers (tasks_detect.rs:300-310)
void scan_grid(int w, int h, int n_fine, int border, int skip_layers, int &gx, int &gy, int &gz);
// a-trous stack -> extrema [<= max_out][4] of all frames, ordered by frame, cube (raster), lane; scratch: slots
// [frames*cubes][8][4], counts [frames*cubes], sums [ceil(frames*cubes/1024)]; totals[0] = written, [1] = dropped
void launch_detect_extrema(const float *layer0, long layer0_stride, int layer0_pitch, const float *coarse, long coarse_stride,
                           long layer_stride, int n_layers, int w, int h, int frames, int border, int skip_layers,
                           float contrast_threshold, float *slots, unsigned *counts, unsigned *sums, float *extrema,
                           unsigned *frame_of, unsigned *frame_start, unsigned long long max_out,
                           unsigned long long *totals, hipStream_t stream, const RowBands *bands = nullptr);
// per frame: blobs with size >= min_size, the n_keep best by contrast, index order; out [n_frames][n_keep][4]
// (the number of extrema is read from n_in on the device, or taken from n_host when n_in is null)
void launch_topk_filter(const float *extrema, const unsigned *seg_start, const unsigned long long *n_in,
                        unsigned long long n_host, unsigned n_frames, unsigned seg_cap, unsigned n_keep, float min_size,
                        float *out, unsigned *out_index, unsigned *out_count, unsigned long long *out_count64,
                        unsigned long long n_cap, unsigned *work, hipStream_t stream);
// u32 words of `work` for lists of up to n_cap extrema (work may be null: one workgroup per frame is used then)
size_t topk_work_words(unsigned long long n_cap);
// per-frame padded selections [frames][n_keep] + counts -> contiguous list + frame ids; totals[0] = entries,
// totals[1] = extrema dropped by the per-frame cap seg_cap
void launch_segments_compact(const float *padded, const unsigned *counts, const unsigned *seg_start,
                             const unsigned long long *n_total, unsigned n_frames, unsigned seg_cap, unsigned n_keep,
                             unsigned *offsets, float *out, unsigned *frame_of, unsigned long long *totals,
                             hipStream_t stream);

// brute-force matcher (csrc/mkd_match.hip): x [n][128] f32 -> f16 hi/lo operand tiles (match_tiles_bytes(n) bytes);
// a tiles against b tiles -> match [na] (index into b or -1) and optionally the best / second-best similarity.
// p_* hold splits x na partial results; excl_lo/hi (nullable): b rows [lo[i], hi[i]) are skipped for a row i.
size_t match_tiles_bytes(long n);
int match_splits(long na, long nb, int num_cus);
// norms [n] (nullable): |x_row| rounded up; max_norm_bits (nullable): running maximum of them as float bits
void launch_match_split(const float *x, long n, unsigned char *tiles, float *norms, unsigned *max_norm_bits,
                        hipStream_t stream);
// the three-term scan; n_over (nullable, device): as the fallback of the two-pass form the launch does nothing unless
// more rows overflowed than launch_match_few takes
void launch_match(const unsigned char *a_tiles, long na, const unsigned char *b_tiles, long nb, const unsigned *excl_lo,
                  const unsigned *excl_hi, float ratio, int splits, float *p_best, int *p_index, float *p_second,
                  int *match, float *best, float *second, const int *n_over, hipStream_t stream);
// small problems (match_small_fits: na * nb <= 2^23, nb <= 4096) in ONE launch straight from the f32 rows, no scratch; the
// three terms of the scan in its order.  overflowed_word (nullable): zeroed by the launch (this form redoes no row)
bool match_small_fits(long na, long nb);
void launch_match_small(const float *a, long na, const float *b, long nb, const unsigned *excl_lo, const unsigned *excl_hi,
                        float ratio, int *match, float *best, float *second, unsigned *overflowed_word, hipStream_t stream);
// both directions (a's rows against b, b's rows against a) in one launch; both must fit match_small_fits
void launch_match_small_both(const float *a, long na, const float *b, long nb, float ratio, int *match_ab, int *match_ba,
                             unsigned *overflowed_word, hipStream_t stream);
// the same scan over the overflowed rows alone (few_words: their indices first, written by launch_match_verify)
size_t match_few_tiles_bytes();
size_t match_few_words();
void launch_match_few(const float *a, const unsigned char *b_tiles, long nb, const unsigned *excl_lo,
                      const unsigned *excl_hi, float ratio, const int *n_over, unsigned char *few_tiles, unsigned *few_words,
                      int *match, float *best, float *second, hipStream_t stream);
// the two-pass form: screen (one-term scan, candidate records per a row) then verify (exact f32 re-scoring, decision);
// *n_over (device, zeroed by the caller) counts a rows whose records overflowed
size_t match_record_bytes(long na, int splits);
size_t match_count_bytes(long na, int splits);
// shared_floor [na] ints (nullable): scratch through which the b splits of a query tell each other their running bounds
void launch_match_screen(const unsigned char *a_tiles, long na, const unsigned char *b_tiles, long nb,
                         const unsigned *excl_lo, const unsigned *excl_hi, int splits, const float *a_norms,
                         const unsigned *b_max_norm_bits, void *rec, void *rec_info, hipStream_t stream,
                         int *shared_floor = nullptr);
void launch_match_verify(const float *a, long na, const float *b, const float *a_norms, const unsigned *b_max_norm_bits,
                         const void *rec, const void *rec_info, int splits, float ratio, int *match, float *best,
                         float *second, int *n_over, int *over_rows, hipStream_t stream);

#ifdef __HIPCC__
// Which tile a workgroup takes, for the row-tiled pyramid and a-trous kernels: workgroups are dealt to the 8 XCDs round-robin by their
// linear id, so the tiles above and below a tile -- which read the same halo rows -- would sit on other XCDs, behind other
// L2s, and every halo row would come from HBM once per tile.  The remap (bijective for any workgroup count) hands each
// group of workgroups that share an XCD a contiguous run of tiles in (x fastest, then y, then frame) order.  (Measured on the
// staged kernels of mkd_pyramid.hip: the frame is then read from HBM once, 0.315 instead of 0.459 GB per 256 frames, and
// the launch is ~5 % shorter; the extremum scan and the decimating kernel did not gain and keep blockIdx.)
struct TileId { unsigned x, y, z; };
__device__ __forceinline__ TileId xcd_tile() {
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = id % 8u, q = nwg / 8u, r = nwg % 8u;
    const unsigned t = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + id / 8u;
    const unsigned row = t / gridDim.x;
    return TileId{t - row * gridDim.x, row % gridDim.y, row / gridDim.y};
}

#endif

}  // namespace lfmkd
