// Launch interface between the C ABI (lf_mkd.cpp) and the gfx950 kernels (mkd_device.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LF_ANGLE_SHADER 0
#define LF_ANGLE_EXACT 1
#define LF_POOL_F32 0
#define LF_POOL_F16X3 1

namespace lfmkd {

constexpr int kMaxPyrLevels = 16;

// Patch pyramid: levels packed back to back in one allocation; level l is w[l] x h[l] f32.
struct PyramidDesc {
    int levels;
    int w[kMaxPyrLevels], h[kMaxPyrLevels];
    long offset[kMaxPyrLevels];  // in floats
};

// Device copies of HostConsts' device layouts (mkd_consts.hpp).
struct DeviceConsts {
    float *phi_cs = nullptr;        // [1024][2]
    short *colmap = nullptr;        // [336]
    float *pool_b_f32 = nullptr;    // [32][12][2][64][4]
    uint16_t *pool_b_f16 = nullptr; // [32][12][2][64][8]
    uint16_t *white_a_f16 = nullptr; // [11][8][2][64][8]
    float *white_a_f32 = nullptr;    // [21][4][8][64]
    float *white_bias = nullptr;     // [128]  -W mean
};

// patches [n][32][32] -> out [n][128] (and, when raw_out != nullptr, the un-whitened [n][238])
void launch_describe(const float *patches, long n, const DeviceConsts &dc, int angle_mode, int pool_mode, float *out,
                     float *raw_out, int num_cus, hipStream_t stream);
// frame_of_kp == nullptr: every keypoint belongs to frame 0
void launch_sample_patches(const float *pyr, long pyr_stride, const PyramidDesc &pd, const float *kps,
                           const unsigned *frame_of_kp, long n, float psf, float *patches, hipStream_t stream);
void launch_build_pyramid(const float *image, long image_stride, float *pyr, long pyr_stride, float *tmp_a,
                          float *tmp_b, const PyramidDesc &pd, int frames, hipStream_t stream);

// a-trous layers 1 .. n_layers-1 over layer0 (= pyramid level 0); tmp holds frames x w x h floats
void launch_build_coarse_stack(const float *layer0, long layer0_stride, float *coarse, long coarse_stride,
                               long layer_stride, float *tmp, int n_layers, int w, int h, int frames,
                               hipStream_t stream);
// extrema [n][4] -> kps [<= max_out][5] ordered by extremum then bin; angles [n][18], counts [n] are scratch;
// totals[0] = written, totals[1] = dropped
void launch_orient(const float *layer0, long layer0_stride, const float *coarse, long coarse_stride, long layer_stride,
                   int n_layers, int w, int h, const float *extrema, const unsigned *frame_of, long n, float *angles,
                   unsigned *counts, float *kps, unsigned *frame_of_kp, unsigned long long max_out,
                   unsigned long long *totals, hipStream_t stream);

}  // namespace lfmkd
