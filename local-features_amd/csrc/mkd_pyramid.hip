// MKD descriptor path, keypoint mode: patch pyramid, a-trous stack and patch sampling for gfx950.
//
//   pyr_*             blur.glsl, swt.glsl (all levels of the a-trous stack), blur_pyramid.glsl, patch_pyramid.rs blits
//   sample_patches    mkd/patch_gradients.glsl:42-70
// (paths under local_features/src/vulkan/)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>

#include "mkd_device.h"
#include "mkd_sample.h"

namespace lfmkd {

// ---------------------------------------------------------------------------------------------
// Keypoint mode: pyramid and sampling.  Sampler = linear filter, MirroredRepeat (mod.rs:940-943),
// restated with exact f32 weights; texel centres at i + 0.5 (mkd_sample.h).
// ---------------------------------------------------------------------------------------------
// All pyramid kernels work on a batch of frames of one size: blockIdx.z = frame, consecutive frames are
// in_stride / out_stride floats apart.
// blur.glsl:34-65 (sigma 0.6) and blur_pyramid.glsl horizontal pass share this shape:
// out = w0 * tex(c) + w1 * (tex(c - off) + tex(c + off)) along one axis.  The bilinear fetch is evaluated exactly as
// a bilinear fetch (mkd_sample.h) does, minus the terms that are multiplied by a weight of exactly 0: the centre tap sits on a texel
// centre (both fractions 0), the side taps have fraction 0 across the pass direction.
// MirroredRepeat index for an offset of at most two texels past either edge of a line of n >= 2 texels: one reflection
// (the same index mirror_idx gives there, in a third of its instructions); REFLECT = false: any offset, any n.
template <bool REFLECT>
__device__ __forceinline__ int edge_idx(int i, int n) {
    if (!REFLECT) return mirror_idx(i, n);
    i = i < 0 ? -1 - i : i;
    return i >= n ? 2 * n - 1 - i : i;
}

// The frame as the caller holds it: f32 in [0, 1], or the 8-bit luma it was made from (lf_mkd_set_image_u8: 1 B/px over PCIe
// and from HBM instead of 4).  An 8-bit pixel becomes (float)v / 255.0f with a true, correctly rounded division -- what the
// reference's callers compute on the host (`u8 as f32 / 255.`, examples/webcam/src/main.rs:136; the `image` crate's
// convert(), examples/match_images/src/main.rs:59-60) -- so both inputs give the same level 0 bit for bit.
__device__ __forceinline__ float px_f32(float v) { return v; }
__device__ __forceinline__ float px_f32(unsigned char v) { return (float)v / 255.0f; }
__device__ __forceinline__ f32x4 load_px4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 load_px4(const unsigned char *p) {   // p 4-byte aligned
    const unsigned v = *reinterpret_cast<const unsigned *>(p);
    return f32x4{px_f32((unsigned char)(v & 255u)), px_f32((unsigned char)((v >> 8) & 255u)),
                 px_f32((unsigned char)((v >> 16) & 255u)), px_f32((unsigned char)(v >> 24))};
}

template <bool REFLECT = false, typename PX = float>
__device__ __forceinline__ float sep3_pixel(const PX *__restrict__ in, int w, int h, int pitch, int x, int y, float w0,
                                            float w1, float off, int vertical) {
#pragma clang fp contract(off)   // the detector's decisions sit on these values: round like the restatement they are tested against
    const float c = (float)(vertical ? y : x) + 0.5f;
    const int n = vertical ? h : w;
    const long stride = vertical ? pitch : 1;
    const PX *line = vertical ? in + x : in + (size_t)y * pitch;
    float side[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float u = k == 0 ? c - off : c + off;
        const float fu = u - 0.5f;
        const float f0 = floorf(fu);
        const float a = fu - f0;
        const int i0 = edge_idx<REFLECT>((int)f0, n), i1 = edge_idx<REFLECT>((int)f0 + 1, n);
        side[k] = px_f32(line[i0 * stride]) * (1.f - a) + px_f32(line[i1 * stride]) * a;
    }
    float s = px_f32(in[(size_t)y * pitch + x]) * w0;
    s += (side[0] + side[1]) * w1;
    return s;
}

// A level's texel together with the texels of the mirrored apron (mkd_device.h) that take its value: the kernels that
// produce a level write its apron themselves where one reflection reaches every apron texel (w >= a and h >= a; the
// smaller levels are left to pyr_apron_fill).  (x, y) inside the level, `lvl0` at its texel (0, 0); a = 0: no apron written.
// x' in [-a, 0) mirrors to -1 - x', x' in [w, w + a) to 2 w - 1 - x'; a texel of a level narrower than 2 a can own both.
__device__ __forceinline__ void store_with_apron(float *__restrict__ lvl0, int pitch, int w, int h, int a, int x, int y,
                                                 float v) {
    float *row = lvl0 + (long)y * pitch;
    row[x] = v;
    if (a == 0) return;
    const bool xl = x < a, xh = x >= w - a;
    if (xl) row[-1 - x] = v;
    if (xh) row[2 * w - 1 - x] = v;
    if (y < a) {
        float *up = lvl0 + (long)(-1 - y) * pitch;
        up[x] = v;
        if (xl) up[-1 - x] = v;
        if (xh) up[2 * w - 1 - x] = v;
    }
    if (y >= h - a) {
        float *dn = lvl0 + (long)(2 * h - 1 - y) * pitch;
        dn[x] = v;
        if (xl) dn[-1 - x] = v;
        if (xh) dn[2 * w - 1 - x] = v;
    }
}

// blur.glsl's two passes (horizontal, then vertical) in one launch, for tap offsets in (1, 2): the same LDS tiling as
// pyr_swt_fused with dilation 1 -- a workgroup computes the horizontal pass of kSwtRows + 4 rows of a 256-column strip
// (slot m = virtual row y0 - 2 + m, holding the row that index mirrors to) and the vertical pass of the kSwtRows rows
// in the middle from them.  The vertical taps of row y blend rows floor(y - off) .. +1 and floor(y + off) .. +1, i.e.
// rows y-2 .. y+2.  Same arithmetic as the two pyr_sep3 dispatches: bit-identical.
// (waves_per_eu: left alone the scheduler aims at 8 waves per SIMD, 36 registers, and waits for every pair of loads of the
// horizontal pass; with ten in flight the kernel is 40 % faster)
template <typename PX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 6))) void pyr_sep3_fused(const PX *__restrict__ in, float *__restrict__ out, long in_stride,
                                                      long out_stride, int w, int h, int opitch, int oapron, float w0, float w1,
                                                      float off, int ty_base) {
#pragma clang fp contract(off)
    __shared__ float s_h[16][256];   // kSwtRows + 4 rows
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int y0 = ((int)blockIdx.y + ty_base) * 12;   // (ty_base: a launch over part of the frame's rows, see RowBands)
    const int xr = (int)blockIdx.x * 256 + (int)threadIdx.x, x = xr < w ? xr : w - 1;
#pragma unroll 4
    for (int m = 0; m < 16; ++m) s_h[m][threadIdx.x] = sep3_pixel<false, PX>(in, w, h, w, x, mirror_idx(y0 - 2 + m, h), w0, w1, off, 0);
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 12; ++k) {
        const int y = y0 + k;
        if (y >= h) break;
        const float c = (float)y + 0.5f;
        float side[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float u = t == 0 ? c - off : c + off;
            const float fu = u - 0.5f;
            const float f0 = floorf(fu);
            const float a = fu - f0;
            int m0 = (int)f0 - (y0 - 2);
            m0 = m0 < 0 ? 0 : (m0 > 14 ? 14 : m0);   // never binding for off in (1, 2)
            side[t] = s_h[m0][threadIdx.x] * (1.f - a) + s_h[m0 + 1][threadIdx.x] * a;
        }
        float sum = s_h[k + 2][threadIdx.x] * w0;
        sum += (side[0] + side[1]) * w1;
        if (xr < w) store_with_apron(out, opitch, w, h, oapron, xr, y, sum);
    }
}

// blur.glsl's two passes with the input rows staged in LDS (see pyr_swt_staged below for the why): 16-byte requests of whole
// row segments, horizontal taps from LDS, horizontal results in registers, vertical pass on them.  A workgroup writes 248
// columns (256 minus 4 texels of halo a side) and 12 rows.  Widths that are multiples of 4; tap offsets in (1, 2): the
// vertical taps of output row y0 + k blend slots k, k + 1 and k + 3, k + 4 (floor(y -+ off) = y - 2, y + 1), which is what
// lets the horizontal results live in registers.  Same arithmetic in the same order as pyr_sep3_fused: bit-identical.
template <typename PX>
__global__ __launch_bounds__(256) void pyr_sep3_staged(const PX *__restrict__ in, float *__restrict__ out, long in_stride,
                                                       long out_stride, int w, int h, int opitch, int oapron, float w0, float w1,
                                                       float off, int ty_base) {
#pragma clang fp contract(off)
    constexpr int kSlots = 16, kH4 = 4, kOutCols = 256 - 2 * kH4;
    __shared__ __attribute__((aligned(16))) float s_raw[kSlots][256];
    const TileId tile = xcd_tile();
    in += tile.z * in_stride;
    out += tile.z * out_stride;
    const int y0 = ((int)tile.y + ty_base) * 12, xs = (int)tile.x * kOutCols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = xs - kH4 + 4 * lane;
    const bool whole = c0 >= 0 && c0 + 3 < w;
    int cm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = mirror_idx(c0 + j, w);
    f32x4 seg[kSlots / 4];
#pragma unroll
    for (int i = 0; i < kSlots / 4; ++i) {
        const PX *row = in + (size_t)mirror_idx(y0 - 2 + wave + 4 * i, h) * w;
        if (whole) seg[i] = load_px4(row + c0);
        else seg[i] = f32x4{px_f32(row[cm[0]]), px_f32(row[cm[1]]), px_f32(row[cm[2]]), px_f32(row[cm[3]])};
    }
#pragma unroll
    for (int i = 0; i < kSlots / 4; ++i) *reinterpret_cast<f32x4 *>(&s_raw[wave + 4 * i][4 * lane]) = seg[i];
    __syncthreads();
    const int t = (int)threadIdx.x, xr = xs + t;
    if (t >= kOutCols || xr >= w) return;
    // horizontal pass (sep3_pixel, horizontal): the side taps blend the columns floor(x -+ off) and the next one
    int j0[2];
    float a[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float c = (float)xr + 0.5f, u = k == 0 ? c - off : c + off;
        const float fu = u - 0.5f, f0 = floorf(fu);
        a[k] = fu - f0;
        const int j = (int)f0 - (xs - kH4);
        j0[k] = j < 0 ? 0 : (j > 254 ? 254 : j);   // never binding for off in (1, 2)
    }
    float hres[kSlots];
#pragma unroll
    for (int m = 0; m < kSlots; ++m) {
        const float *row = s_raw[m];
        const float side0 = row[j0[0]] * (1.f - a[0]) + row[j0[0] + 1] * a[0];
        const float side1 = row[j0[1]] * (1.f - a[1]) + row[j0[1] + 1] * a[1];
        float sum = row[t + kH4] * w0;
        sum += (side0 + side1) * w1;
        hres[m] = sum;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int y = y0 + k;
        if (y >= h) break;
        const float c = (float)y + 0.5f;
        const float fl = (c - off) - 0.5f, fh = (c + off) - 0.5f;
        const float al = fl - floorf(fl), ah = fh - floorf(fh);
        const float side0 = hres[k] * (1.f - al) + hres[k + 1] * al;
        const float side1 = hres[k + 3] * (1.f - ah) + hres[k + 4] * ah;
        float sum = hres[k + 2] * w0;
        sum += (side0 + side1) * w1;
        store_with_apron(out, opitch, w, h, oapron, xr, y, sum);
    }
}

// swt.glsl:24-58, both passes in one launch: [1 4 6 4 1]/16 at texel centres, taps d = 2^in_level apart, mirrored,
// horizontal pass then vertical pass.  The reference runs them as two dispatches through a scratch layer; here a
// workgroup keeps the horizontal results it needs in LDS, so a layer costs one read and one write of the frame
// instead of two of each.  A workgroup owns a 256-column strip and kSwtRows output rows of ONE residue class modulo d
// (rows r, r + d, r + 2d, ...): their vertical taps are rows of the same class, so kSwtRows + 4 horizontal rows
// serve kSwtRows outputs whatever the dilation.  Slot m of the LDS tile stands for the virtual row r + m d and holds
// the horizontal pass of the row that index mirrors to -- which is the row the two-pass form would have read.
// Operation order and rounding are those of the two dispatches (no contraction): results are bit-identical.
constexpr int kSwtRows = 12, kSwtCols = 256;

__global__ __launch_bounds__(256) void pyr_swt_fused(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                     long out_stride, int w, int h, int ipitch, int d, int blocks_per_class,
                                                     int k_first, int k_end) {
#pragma clang fp contract(off)
    __shared__ float s_h[kSwtRows + 4][kSwtCols];
    const float k0 = 6.f / 16.f, k1 = 4.f / 16.f, k2 = 1.f / 16.f;
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int r = blockIdx.y / blocks_per_class;                    // residue class of the rows
    // first lattice index of this workgroup: the launch covers lattice rows [k_first, k_end) of every class (row bands)
    const int kb = k_first + (blockIdx.y - r * blocks_per_class) * kSwtRows;
    const int xs = (int)blockIdx.x * kSwtCols;   // signed: xs - 2 d must be able to go negative
    const int xr = xs + (int)threadIdx.x, x = xr < w ? xr : w - 1;
    const bool interior = xs - 2 * d >= 0 && xs + kSwtCols - 1 + 2 * d < w;
    int xi[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) xi[t] = interior ? x + (t - 2) * d : mirror_idx(x + (t - 2) * d, w);
    // horizontal pass of the kSwtRows + 4 rows this workgroup's outputs reach
#pragma unroll 4
    for (int m = 0; m < kSwtRows + 4; ++m) {
        const int v = r + (kb + m - 2) * d;                // virtual row of slot m
        const float *row = in + (size_t)mirror_idx(v, h) * ipitch;
        float sum = row[xi[2]] * k0;
        sum += row[xi[0]] * k2;
        sum += row[xi[1]] * k1;
        sum += row[xi[3]] * k1;
        sum += row[xi[4]] * k2;
        s_h[m][threadIdx.x] = sum;
    }
    __syncthreads();
    // vertical pass: output row r + (kb + k) d reads slots k .. k + 4
#pragma unroll 4
    for (int k = 0; k < kSwtRows; ++k) {
        const int y = r + (kb + k) * d;
        if (y >= h || kb + k >= k_end) break;
        float sum = s_h[k + 2][threadIdx.x] * k0;
        sum += s_h[k + 1][threadIdx.x] * k1;
        sum += s_h[k][threadIdx.x] * k2;
        sum += s_h[k + 4][threadIdx.x] * k2;
        sum += s_h[k + 3][threadIdx.x] * k1;
        if (xr < w) out[(size_t)y * w + xr] = sum;
    }
}

// The same layer with the input rows staged in LDS: a wave requests whole 1 KB row segments, 16 bytes per lane, four rows
// in flight (tools/micro/tile_copy.hip: this request shape streams frames at 5.6 TB/s, one dword per lane and five
// overlapping taps per output at 4.0), the five horizontal taps are then LDS reads, the horizontal results stay in
// registers and the vertical pass reads them there.  The segment is 256 texels -- the H4 texels of halo on either side
// (H4 >= 2 d, a multiple of 4 so that the requests stay 16-byte aligned) come out of the strip's width: a workgroup writes
// 256 - 2 H4 columns.  For d <= 32, widths and pitches that are multiples of 4 (pyr_swt_fused serves the rest); same
// arithmetic in the same order: bit-identical.  Per layer of 256 frames 640 x 480: 174 -> 133 us (d = 1), 256 -> 187 (d = 32);
// of a 4K frame: 19 -> 14 us.
// `blit` (dilation 1, even frame sizes): the Nearest blit of patch_pyramid.rs:251-285 picks texel (2x + 1, 2y + 1) of this
// layer for pyramid level 1 -- stored from here, with its apron, instead of by a launch of pyr_decimate that reads the layer back.
struct SwtBlit { float *out; long stride; int pitch, apron; };

template <int H4>
__global__ __launch_bounds__(256) void pyr_swt_staged(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                      long out_stride, int w, int h, int ipitch, int d, int blocks_per_class,
                                                      SwtBlit blit, int k_first, int k_end) {
#pragma clang fp contract(off)
    constexpr int kSlots = kSwtRows + 4, kOutCols = kSwtCols - 2 * H4;
    __shared__ __attribute__((aligned(16))) float s_raw[kSlots][kSwtCols];
    const float k0 = 6.f / 16.f, k1 = 4.f / 16.f, k2 = 1.f / 16.f;
    const TileId tile = xcd_tile();
    in += tile.z * in_stride;
    out += tile.z * out_stride;
    const int r = tile.y / blocks_per_class;
    const int kb = k_first + (tile.y - r * blocks_per_class) * kSwtRows;   // (lattice rows [k_first, k_end): see pyr_swt_fused)
    const int xs = (int)tile.x * kOutCols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this lane's four texels of a segment: columns c0 .. c0 + 3 (virtual: mirrored where they leave the frame)
    const int c0 = xs - H4 + 4 * lane;
    const bool whole = c0 >= 0 && c0 + 3 < w;
    int cm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = mirror_idx(c0 + j, w);
    f32x4 seg[kSlots / 4];
#pragma unroll
    for (int i = 0; i < kSlots / 4; ++i) {
        const int m = wave + 4 * i;
        const float *row = in + (size_t)mirror_idx(r + (kb + m - 2) * d, h) * ipitch;
        if (whole) seg[i] = *reinterpret_cast<const f32x4 *>(row + c0);
        else seg[i] = f32x4{row[cm[0]], row[cm[1]], row[cm[2]], row[cm[3]]};
    }
#pragma unroll
    for (int i = 0; i < kSlots / 4; ++i) *reinterpret_cast<f32x4 *>(&s_raw[wave + 4 * i][4 * lane]) = seg[i];
    __syncthreads();
    const int t = (int)threadIdx.x, xr = xs + t;
    if (t >= kOutCols || xr >= w) return;
    float hres[kSlots];
#pragma unroll
    for (int m = 0; m < kSlots; ++m) {
        const float *row = &s_raw[m][t + H4];
        float sum = row[0] * k0;
        sum += row[-2 * d] * k2;
        sum += row[-d] * k1;
        sum += row[d] * k1;
        sum += row[2 * d] * k2;
        hres[m] = sum;
    }
#pragma unroll
    for (int k = 0; k < kSwtRows; ++k) {
        const int y = r + (kb + k) * d;
        if (y >= h || kb + k >= k_end) break;
        float sum = hres[k + 2] * k0;
        sum += hres[k + 1] * k1;
        sum += hres[k] * k2;
        sum += hres[k + 4] * k2;
        sum += hres[k + 3] * k1;
        out[(size_t)y * w + xr] = sum;
        if (H4 == 4 && blit.out && (xr & 1) && (y & 1))
            store_with_apron(blit.out + tile.z * blit.stride, blit.pitch, w / 2, h / 2, blit.apron, xr >> 1, y >> 1, sum);
    }
}

// Nearest blit [0,w)x[0,h) -> [0,w/2)x[0,h/2): patch_pyramid.rs:251-285.
__global__ void pyr_decimate(const float *__restrict__ in, float *__restrict__ out, long in_stride, long out_stride,
                             int w, int h, int ow, int oh, int opitch, int oapron) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= ow || y >= oh) return;
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    int sx = (int)floorf(((float)x + 0.5f) * (float)w / (float)(w / 2));
    int sy = (int)floorf(((float)y + 0.5f) * (float)h / (float)(h / 2));
    sx = sx > w - 1 ? w - 1 : sx;
    sy = sy > h - 1 ? h - 1 : sy;
    store_with_apron(out, opitch, ow, oh, oapron, x, y, in[(size_t)sy * w + sx]);
}

// Level 1 in one launch when nobody else needs a-trous layer 1 (describe-only callers: the detector and the orientation
// stage are what share it): the a-trous pass (swt.glsl:24-58, dilation 1) evaluated only at the texels the Nearest blit
// (patch_pyramid.rs:251-285) picks -- the horizontal pass at the picked columns of the rows the picked rows reach, kept in
// LDS, the vertical pass at the picked rows.  Same pixel arithmetic in the same order as pyr_swt_fused + pyr_decimate:
// bit-identical, a quarter of the vertical work, and layer 1 (a full frame) is neither written nor read back.
constexpr int kL1Cols = 128;                     // output columns per workgroup (level 1 of a 640-wide frame: 320 = 2.5 x 128)

__device__ __forceinline__ int blit_src(int i, int n) {   // pyr_decimate's source index, clamped the same way
    int s = (int)floorf(((float)i + 0.5f) * (float)n / (float)(n / 2));
    return s > n - 1 ? n - 1 : s;
}

// ROWS = output rows per workgroup: 16 for batches of frames (less halo per output row), 8 for a single frame (more
// workgroups); they reach 2 ROWS + 8 source rows (odd heights step by 3 now and then)
template <int ROWS>
__global__ __launch_bounds__(kL1Cols) void pyr_level1_fused(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                        long out_stride, int w, int h, int ipitch, int ow, int oh,
                                                        int opitch, int oapron) {
#pragma clang fp contract(off)
    constexpr int kL1Rows = ROWS, kL1Slots = 2 * ROWS + 8;
    __shared__ float s_h[kL1Slots][kL1Cols];
    const float k0 = 6.f / 16.f, k1 = 4.f / 16.f, k2 = 1.f / 16.f;
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int y0 = (int)blockIdx.y * kL1Rows;
    const int xr = (int)blockIdx.x * kL1Cols + (int)threadIdx.x, x = xr < ow ? xr : ow - 1;
    const int sx = blit_src(x, w);
    int xi[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) xi[t] = mirror_idx(sx + t - 2, w);
    const int y_last = min(y0 + kL1Rows, oh) - 1;
    const int v0 = blit_src(y0, h) - 2;                       // virtual source row of slot 0
    const int n_slots = min(blit_src(y_last, h) + 2 - v0 + 1, kL1Slots);
    for (int m = 0; m < n_slots; ++m) {
        const float *row = in + (size_t)mirror_idx(v0 + m, h) * ipitch;
        float sum = row[xi[2]] * k0;
        sum += row[xi[0]] * k2;
        sum += row[xi[1]] * k1;
        sum += row[xi[3]] * k1;
        sum += row[xi[4]] * k2;
        s_h[m][threadIdx.x] = sum;
    }
    __syncthreads();
    for (int y = y0; y <= y_last; ++y) {
        const int k = blit_src(y, h) - 2 - v0;                  // slot of source row sy - 2
        float sum = s_h[k + 2][threadIdx.x] * k0;
        sum += s_h[k + 1][threadIdx.x] * k1;
        sum += s_h[k][threadIdx.x] * k2;
        sum += s_h[k + 4][threadIdx.x] * k2;
        sum += s_h[k + 3][threadIdx.x] * k1;
        if (xr < ow) store_with_apron(out, opitch, ow, oh, oapron, xr, y, sum);
    }
}

// pyr_level1_fused with the source rows staged in LDS by 16-byte requests (pyr_swt_staged has the why), for even heights and
// widths that are multiples of 4 (the blit then picks texel (2x + 1, 2y + 1) exactly).  A workgroup of two waves writes 124
// columns and ROWS rows: the 256-texel segment starts at source column 2 x0 - 4, output column x0 + t reads its five taps at
// segment texels 2t + 3 .. 2t + 7; slot m holds source row 2 y0 - 1 + m (mirrored), output row y0 + k reads slots 2k .. 2k + 4.
// Same arithmetic in the same order: bit-identical.
template <int ROWS>
__global__ __launch_bounds__(128) void pyr_level1_staged(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                         long out_stride, int w, int h, int ipitch, int ow, int oh, int opitch,
                                                         int oapron) {
#pragma clang fp contract(off)
    constexpr int kSlots = 2 * ROWS + 4, kOutCols = 124;
    __shared__ __attribute__((aligned(16))) float s_raw[kSlots][256];
    const float k0 = 6.f / 16.f, k1 = 4.f / 16.f, k2 = 1.f / 16.f;
    const TileId tile = xcd_tile();
    in += tile.z * in_stride;
    out += tile.z * out_stride;
    const int y0 = (int)tile.y * ROWS, x0 = (int)tile.x * kOutCols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = 2 * x0 - 4 + 4 * lane, v0 = 2 * y0 - 1;
    const bool whole = c0 >= 0 && c0 + 3 < w;
    int cm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = mirror_idx(c0 + j, w);
    f32x4 seg[kSlots / 2];
#pragma unroll
    for (int i = 0; i < kSlots / 2; ++i) {
        const float *row = in + (size_t)mirror_idx(v0 + wave + 2 * i, h) * ipitch;
        if (whole) seg[i] = *reinterpret_cast<const f32x4 *>(row + c0);
        else seg[i] = f32x4{row[cm[0]], row[cm[1]], row[cm[2]], row[cm[3]]};
    }
#pragma unroll
    for (int i = 0; i < kSlots / 2; ++i) *reinterpret_cast<f32x4 *>(&s_raw[wave + 2 * i][4 * lane]) = seg[i];
    __syncthreads();
    const int t = (int)threadIdx.x, xr = x0 + t;
    if (t >= kOutCols || xr >= ow) return;
    float hres[kSlots];
#pragma unroll
    for (int m = 0; m < kSlots; ++m) {
        const float *row = &s_raw[m][2 * t + 5];
        float sum = row[0] * k0;
        sum += row[-2] * k2;
        sum += row[-1] * k1;
        sum += row[1] * k1;
        sum += row[2] * k2;
        hres[m] = sum;
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int y = y0 + k;
        if (y >= oh) break;
        float sum = hres[2 * k + 2] * k0;
        sum += hres[2 * k + 1] * k1;
        sum += hres[2 * k] * k2;
        sum += hres[2 * k + 4] * k2;
        sum += hres[2 * k + 3] * k1;
        store_with_apron(out, opitch, ow, oh, oapron, xr, y, sum);
    }
}

// (Round 5 built levels 0 AND 1 from one read of the frame, twice -- one column per thread like the kernels above, then four
// columns per lane with two-wide packed arithmetic -- bit-identical both times and no faster than pyr_sep3_staged followed by
// pyr_level1_staged (128 frames of 1080p: 704 us for the two launches, 700 and 775 us fused): each of the two kernels already
// sits at its HBM time AND at its instruction issue time, so the fused kernel pays the sum of their arithmetic on less traffic.
// profiles/r05_ab_pyramid.txt; removed.)

// blur_pyramid.glsl:36-49 vertical pass: binomial taps centred on texel (2x, 2y) of the H result.
template <bool REFLECT = false>
__device__ __forceinline__ float down_v_pixel(const float *__restrict__ in, int w, int h, int x, int y) {
#pragma clang fp contract(off)
    // taps centred on texel (2x, 2y): same arithmetic as a bilinear fetch, zero-weight terms left out (see sep3_pixel)
    const int sx = edge_idx<REFLECT>(2 * x, w);
    const float cy = 2.f * (float)y + 0.5f;
    float side[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float u = k == 0 ? cy - 1.2f : cy + 1.2f;
        const float fu = u - 0.5f;
        const float f0 = floorf(fu);
        const float a = fu - f0;
        const int i0 = edge_idx<REFLECT>((int)f0, h), i1 = edge_idx<REFLECT>((int)f0 + 1, h);
        side[k] = in[(size_t)i0 * w + sx] * (1.f - a) + in[(size_t)i1 * w + sx] * a;
    }
    float s = in[(size_t)edge_idx<REFLECT>(2 * y, h) * w + sx] * 0.375f;
    s += (side[0] + side[1]) * 0.3125f;
    return s;
}

// blur_pyramid.glsl's two passes for one level in one launch: the horizontal pass is only needed at the even columns and
// at the 2 kDownRows + 3 rows around the output rows (taps at 2y -+ 1.2 blend rows 2y-2 .. 2y+2), kept in LDS; the
// two-dispatch form computes it for every texel of level l-1 and writes it out.  Same pixel arithmetic: bit-identical.
constexpr int kDownRows = 6;

__global__ __launch_bounds__(256) void pyr_down_fused(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                      long out_stride, int pw, int ph, int ppitch, int ow, int oh,
                                                      int opitch, int oapron) {
#pragma clang fp contract(off)
    __shared__ float s_h[2 * kDownRows + 3][256];
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int y0 = (int)blockIdx.y * kDownRows;
    const int xr = (int)blockIdx.x * 256 + (int)threadIdx.x, x = xr < ow ? xr : ow - 1;
    const int sx = mirror_idx(2 * x, pw);
    const int v0 = 2 * y0 - 2;   // virtual row of slot 0
#pragma unroll 5
    for (int m = 0; m < 2 * kDownRows + 3; ++m)
        s_h[m][threadIdx.x] = sep3_pixel(in, pw, ph, ppitch, sx, mirror_idx(v0 + m, ph), 0.375f, 0.3125f, 1.2f, 0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kDownRows; ++k) {
        const int y = y0 + k;
        if (y >= oh) break;
        const float cy = 2.f * (float)y + 0.5f;
        float side[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float u = t == 0 ? cy - 1.2f : cy + 1.2f;
            const float fu = u - 0.5f;
            const float f0 = floorf(fu);
            const float a = fu - f0;
            int m0 = (int)f0 - v0;
            m0 = m0 < 0 ? 0 : (m0 > 2 * kDownRows + 1 ? 2 * kDownRows + 1 : m0);   // never binding
            side[t] = s_h[m0][threadIdx.x] * (1.f - a) + s_h[m0 + 1][threadIdx.x] * a;
        }
        float sum = s_h[2 * k + 2][threadIdx.x] * 0.375f;
        sum += (side[0] + side[1]) * 0.3125f;
        if (xr < ow) store_with_apron(out, opitch, ow, oh, oapron, xr, y, sum);
    }
}

// pyr_down_fused with the source rows staged in LDS by 16-byte requests (round 5; pyr_swt_staged has the why: the fused form
// gathers every other texel of level l-1 with one dword per lane, 75 such loads per thread).  A workgroup of two waves writes 124
// columns and ROWS rows of level l: the 256-texel segment starts at source column 2 x0 - 4 (virtual: mirrored where it leaves
// the level), output column x0 + t has its centre tap at segment texel 2 t + 4 and its side taps at floor(2 x -+ 1.2) and the
// texel after; slot m holds source row 2 y0 - 2 + m (mirrored), output row y0 + k reads slots 2k .. 2k + 4.  The tap weights
// are computed from the coordinates exactly as sep3_pixel and pyr_down_fused compute them.  For level widths (of l-1) that are
// multiples of 4 with 16-byte aligned rows.  Same arithmetic in the same order: bit-identical.
template <int ROWS>
__global__ __launch_bounds__(128) void pyr_down_staged(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                       long out_stride, int pw, int ph, int ppitch, int ow, int oh, int opitch,
                                                       int oapron) {
#pragma clang fp contract(off)
    constexpr int kSlots = 2 * ROWS + 4, kOutCols = 124;      // (2 ROWS + 3 are read; an even count keeps the two waves' loops alike)
    __shared__ __attribute__((aligned(16))) float s_raw[kSlots][256];
    const TileId tile = xcd_tile();
    in += tile.z * in_stride;
    out += tile.z * out_stride;
    const int y0 = (int)tile.y * ROWS, x0 = (int)tile.x * kOutCols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = 2 * x0 - 4 + 4 * lane, v0 = 2 * y0 - 2;
    const bool whole = c0 >= 0 && c0 + 3 < pw;
    int cm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = mirror_idx(c0 + j, pw);
    f32x4 seg[kSlots / 2];
#pragma unroll
    for (int i = 0; i < kSlots / 2; ++i) {
        const float *row = in + (size_t)mirror_idx(v0 + wave + 2 * i, ph) * ppitch;
        if (whole) seg[i] = *reinterpret_cast<const f32x4 *>(row + c0);
        else seg[i] = f32x4{row[cm[0]], row[cm[1]], row[cm[2]], row[cm[3]]};
    }
#pragma unroll
    for (int i = 0; i < kSlots / 2; ++i) *reinterpret_cast<f32x4 *>(&s_raw[wave + 2 * i][4 * lane]) = seg[i];
    __syncthreads();
    const int t = (int)threadIdx.x, xr = x0 + t;
    if (t >= kOutCols || xr >= ow) return;
    // horizontal pass at source column sx = 2 xr (sep3_pixel, horizontal, w0 = 0.375, w1 = 0.3125, off = 1.2)
    const int sx = 2 * xr, base = 2 * x0 - 4;
    int j0[2];
    float a[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float c = (float)sx + 0.5f, u = k == 0 ? c - 1.2f : c + 1.2f;
        const float fu = u - 0.5f, f0 = floorf(fu);
        a[k] = fu - f0;
        const int j = (int)f0 - base;
        j0[k] = j < 0 ? 0 : (j > 254 ? 254 : j);   // never binding: the taps are texels 2t + 2 .. 2t + 6 of the segment
    }
    float hres[2 * ROWS + 3];
#pragma unroll
    for (int m = 0; m < 2 * ROWS + 3; ++m) {
        const float *row = s_raw[m];
        const float side0 = row[j0[0]] * (1.f - a[0]) + row[j0[0] + 1] * a[0];
        const float side1 = row[j0[1]] * (1.f - a[1]) + row[j0[1] + 1] * a[1];
        float sum = row[sx - base] * 0.375f;
        sum += (side0 + side1) * 0.3125f;
        hres[m] = sum;
    }
    // vertical pass (pyr_down_fused's): taps at rows floor(2y -+ 1.2) and the row after, centre 2y
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int y = y0 + k;
        if (y >= oh) break;
        const float cy = 2.f * (float)y + 0.5f;
        const float fl = (cy - 1.2f) - 0.5f, fh = (cy + 1.2f) - 0.5f;
        const float al = fl - floorf(fl), ah = fh - floorf(fh);
        const float side0 = hres[2 * k] * (1.f - al) + hres[2 * k + 1] * al;
        const float side1 = hres[2 * k + 3] * (1.f - ah) + hres[2 * k + 4] * ah;
        float sum = hres[2 * k + 2] * 0.375f;
        sum += (side0 + side1) * 0.3125f;
        store_with_apron(out, opitch, ow, oh, oapron, xr, y, sum);
    }
}

// The small end of the pyramid in one launch: from level l0 on (where the horizontal result of level l-1 fits the
// 64 KiB LDS tile) one workgroup per frame walks the remaining levels, the horizontal pass into LDS, the decimating
// vertical pass from it, then the level's apron.  Same pixel functions as the per-level kernels; it only replaces a dozen tiny launches.
constexpr int kTailPixels = 16384;
constexpr int kTailAprChunks = 4;   // 64-texel pieces of a padded row of a tail level: 256 texels hold the levels of ordinary shapes
// (slimmer, wider tail levels: the launcher leaves their aprons to pyr_apron_fill)

__global__ __launch_bounds__(1024) void pyr_tail(float *__restrict__ pyr, long pyr_stride, PyramidDesc pd, int l0,
                                                 int with_apron) {
    __shared__ float s_tmp[kTailPixels];
    float *base = pyr + blockIdx.x * pyr_stride;
    for (int l = l0; l < pd.levels; ++l) {
        const int pw = pd.w[l - 1], ph = pd.h[l - 1], ow = pd.w[l], oh = pd.h[l];
        const float *in = base + pd.offset[l - 1];
        // this workgroup is one CU's worth of arithmetic per frame: the taps reach two texels past an edge at most, so one
        // reflection stands for MirroredRepeat on all but one-texel-wide levels, and the row of a flat index comes from a
        // reciprocal (indices stay below 2^14: exact up to the fix-up)
        const bool reflect = pw >= 2 && ph >= 2;
        auto row_of = [](int i, int n, float inv, int &x) {
            int y = (int)((float)i * inv);
            x = i - y * n;
            if (x < 0) { --y; x += n; }
            if (x >= n) { ++y; x -= n; }
            return y;
        };
        const float inv_pw = 1.f / (float)pw, inv_ow = 1.f / (float)ow;
        for (int i = threadIdx.x; i < pw * ph; i += 1024) {
            int x;
            const int y = row_of(i, pw, inv_pw, x);
            s_tmp[i] = reflect ? sep3_pixel<true>(in, pw, ph, pd.pitch[l - 1], x, y, 0.375f, 0.3125f, 1.2f, 0)
                               : sep3_pixel<false>(in, pw, ph, pd.pitch[l - 1], x, y, 0.375f, 0.3125f, 1.2f, 0);
        }
        __syncthreads();
        float *out = base + pd.offset[l];
        for (int i = threadIdx.x; i < ow * oh; i += 1024) {
            int x;
            const int y = row_of(i, ow, inv_ow, x);
            out[(size_t)y * pd.pitch[l] + x] = reflect ? down_v_pixel<true>(s_tmp, pw, ph, x, y) : down_v_pixel<false>(s_tmp, pw, ph, x, y);
        }
        __threadfence_block();   // level l is the input of level l + 1, read by other threads of this workgroup
        __syncthreads();
        // its mirrored apron (any number of reflections: these levels are smaller than the apron is wide); the next level's
        // horizontal pass reads the interior only, so no barrier is needed behind this.  A wave takes four rows of the padded
        // level at a time, all their loads before the first store (one round trip per four rows, not per texel).
        if (!with_apron) continue;
        const int a = pd.apron[l], pitch = pd.pitch[l];
        const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
        for (int r0 = 4 * wave - a; r0 < oh + a; r0 += 64) {
            float v[4][kTailAprChunks];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float *src = out + (long)mirror_idx(min(r0 + j, oh + a - 1), oh) * pitch;
#pragma unroll
                for (int m = 0; m < kTailAprChunks; ++m) v[j][m] = src[mirror_idx(min(lane + 64 * m - a, ow + a - 1), ow)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int py = r0 + j;
                if (py >= oh + a) break;
                float *dst = out + (long)py * pitch;
                const bool inside = py >= 0 && py < oh;
#pragma unroll
                for (int m = 0; m < kTailAprChunks; ++m) {
                    const int px = lane + 64 * m - a;
                    if (px < ow + a && (!inside || px < 0 || px >= ow)) dst[px] = v[j][m];
                }
            }
        }
    }
}

// The apron of the pyramid's levels (mkd_device.h): every texel outside the level, up to kPyrApron away, takes the value
// MirroredRepeat addressing would have fetched for it.  The kernels that produce a level write its apron with it
// (store_with_apron, pyr_tail); this one is for what they leave: levels narrower than the apron that are too large for
// pyr_tail (frames of extreme aspect ratio).  blockIdx.y = level - first, blockIdx.z = frame.
__global__ __launch_bounds__(256) void pyr_apron_fill(float *__restrict__ pyr, long pyr_stride, PyramidDesc pd, int first) {
    const int l = first + (int)blockIdx.y;
    const int w = pd.w[l], h = pd.h[l], a = pd.apron[l], pitch = pd.pitch[l];
    float *lvl0 = pyr + blockIdx.z * pyr_stride + pd.offset[l];
    const int task = (int)blockIdx.x;
    if (task < 2 * a) {
        // a row of the band above or below the level, over the whole padded width: the mirrored source row is uniform
        const int y = task < a ? task - a : h + (task - a);
        const float *src = lvl0 + (long)mirror_idx(y, h) * pitch;
        float *dst = lvl0 + (long)y * pitch;
        for (int t = (int)threadIdx.x; t < pitch; t += 256) dst[t - a] = src[mirror_idx(t - a, w)];
    } else {
        // the bands left and right of eight of the level's own rows
        const int y0 = 8 * (task - 2 * a);
        if (y0 >= h) return;
        for (int i = (int)threadIdx.x; i < 8 * 2 * a; i += 256) {
            const int rr = i / (2 * a), c = i - rr * (2 * a), y = y0 + rr;
            if (y >= h) break;
            const int x = c < a ? c - a : w + (c - a);
            lvl0[(long)y * pitch + x] = lvl0[(long)y * pitch + mirror_idx(x, w)];
        }
    }
}

// patch_gradients.glsl:42-70 as a launch of its own: the verification tap (lf_mkd_sample_patches_device) and the sampling
// stage of the f32 verification mode; the product path samples inside the describe kernel (mkd_describe.hip) with the same
// arithmetic (mkd_sample.h), bit for bit.  One wave per keypoint (4 per block): the per-keypoint scale/level/rotation math
// is done once per wave instruction, each lane then samples 16 pixels.  A load instruction covers an 8 x 8 block of the
// patch -- a compact footprint of ~11 texel rows -- rather than two 32-pixel patch rows along a rotated line (~30 lines:
// the texture path's cost is per cache line touched, tools/micro/gather_patterns.hip); the patch is transposed through
// LDS so that it still leaves in 256-byte row stores.
// frame_of_kp (optional) selects the keypoint's pyramid among the frames of the batch (pyr_stride floats apart).
__global__ __launch_bounds__(256) void sample_patches(const float *__restrict__ pyr, long pyr_stride, PyramidDesc pd,
                                                      const float *__restrict__ kps /*[n][5]*/,
                                                      const unsigned *__restrict__ frame_of_kp, unsigned n_frames,
                                                      long n_host, const unsigned long long *__restrict__ n_dev, float psf,
                                                      float *__restrict__ patches) {
    __shared__ float s_patch[4][1024];
    const long n = n_dev ? (long)*n_dev : n_host;
    const long k = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    if (frame_of_kp) pyr += (long)min(frame_of_kp[k], n_frames - 1u) * pyr_stride;
    const float *kp = kps + k * 5;
    const KpGeom g = keypoint_geometry(kp[0], kp[1], kp[2], kp[3], psf, level_table(pd));
    const float *lvl0 = pyr + pd.offset[g.level];
    const int w = pd.w[g.level], h = pd.h[g.level], pitch = pd.pitch[g.level], apron = pd.apron[g.level];
    float *tile = s_patch[threadIdx.x >> 6];
    // pixel of this lane in step i: block (i & 3, i >> 2) of 8 x 8 pixels, lane = 8 * row + column inside it
    const int lx0 = lane & 7, ly0 = lane >> 3;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int lx = 8 * (i & 3) + lx0, ly = 8 * (i >> 2) + ly0;
        const SamplePos p = sample_position(g.ca, g.sa, g.rem, g.cx, g.cy, lx, ly);
        tile[ly * 32 + lx] = sample_level(lvl0, w, h, pitch, apron, g.covered, p);
    }
    __builtin_amdgcn_wave_barrier();
    float *dst = patches + k * 1024 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j * 64] = tile[j * 64 + lane];
}

void launch_sample_patches(const float *pyr, long pyr_stride, const PyramidDesc &pd, const float *kps,
                           const unsigned *frame_of_kp, unsigned n_frames, long n, const unsigned long long *n_dev, float psf,
                           float *patches, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(sample_patches, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, pyr, pyr_stride, pd, kps,
                       frame_of_kp, n_frames ? n_frames : 1u, n, n_dev, psf, patches);
}

// Builds the pyramids of `frames` frames (image_stride floats apart) into pyr (pyr_stride apart); tmp_a and tmp_b
// hold frames x w x h floats each.
// one a-trous layer (both passes) for `frames` frames
// returns whether the launch also wrote `blit` (pyramid level 1, from the dilation-1 layer)
// (row_lo, row_hi: the launch covers output rows [row_lo, row_hi) only -- multiples of d, or the frame's height / -1 for row_hi:
//  the same lattice rows of every residue class, in tiles of kSwtRows of them from row_lo / d on (the last tile of a range may
//  be partial); the default is the whole frame)
static bool launch_swt(const float *in, long in_stride, int in_pitch, float *out, long out_stride, int w, int h, int d,
                       int frames, hipStream_t stream, SwtBlit blit = SwtBlit{nullptr, 0, 0, 0}, int row_lo = 0, int row_hi = -1) {
    const int classes = d < h ? d : h;                                   // residue classes that hold rows
    const int lattice = (h + d - 1) / d;                                 // rows of the longest class
    const int k_first = row_lo / d;
    const int k_end = row_hi < 0 || row_hi >= h ? lattice : row_hi / d;
    const int per_class = (k_end - k_first + kSwtRows - 1) / kSwtRows;   // tiles of a class
    if (k_end <= k_first) return false;
    // the staged form where its 16-byte requests are aligned and its halo fits
    // (d = 32 leaves 128 of the segment's 256 columns to write and is still ahead: 17 against 21 us on a 4K frame)
    if (d <= 32 && w % 4 == 0 && in_pitch % 4 == 0 && in_stride % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const bool with_blit = blit.out && d == 1 && w % 2 == 0 && h % 2 == 0;
        if (!with_blit) blit.out = nullptr;
        auto go = [&](auto kernel, int h4) {
            const int oc = kSwtCols - 2 * h4;
            hipLaunchKernelGGL(kernel, dim3((w + oc - 1) / oc, classes * per_class, frames), dim3(256), 0, stream, in, out,
                               in_stride, out_stride, w, h, in_pitch, d, per_class, blit, k_first, k_end);
        };
        if (d <= 2) go(pyr_swt_staged<4>, 4);
        else if (d == 4) go(pyr_swt_staged<8>, 8);
        else if (d == 8) go(pyr_swt_staged<16>, 16);
        else if (d == 16) go(pyr_swt_staged<32>, 32);
        else go(pyr_swt_staged<64>, 64);
        return with_blit;
    }
    hipLaunchKernelGGL(pyr_swt_fused, dim3((w + kSwtCols - 1) / kSwtCols, classes * per_class, frames), dim3(256), 0, stream,
                       in, out, in_stride, out_stride, w, h, in_pitch, d, per_class, k_first, k_end);
    return false;
}

// (blockIdx.x = a row of the upper / lower band or a group of eight rows of the side bands; sized for level `first`, the
// smaller levels' surplus workgroups leave at once)
static void launch_apron_fill(float *pyr, long pyr_stride, const PyramidDesc &pd, int first, int end, int frames,
                              hipStream_t stream) {
    if (first >= end) return;
    const unsigned gx = (unsigned)(2 * pd.apron[first] + (pd.h[first] + 7) / 8);
    hipLaunchKernelGGL(pyr_apron_fill, dim3(gx, (unsigned)(end - first), (unsigned)frames), dim3(256), 0, stream, pyr,
                       pyr_stride, pd, first);
}

// With layer1 != nullptr the a-trous layer 1 the pyramid needs anyway is written there (frames layer1_stride apart) instead
// of tmp_b: it is layer 1 of the stack orientation and the detector read, so they need not build it again.
void launch_build_pyramid(const float *image, long image_stride, float *pyr, long pyr_stride, float *tmp_a,
                          float *tmp_b, const PyramidDesc &pd, int frames, float *layer1, long layer1_stride,
                          hipStream_t stream, hipStream_t rest_stream, hipEvent_t fork, hipEvent_t join,
                          const std::function<void()> &main_next, const unsigned char *image_u8, const RowBands *bands) {
    const int w = pd.w[0], h = pd.h[0];
    // (bands: this launch sequence is one piece of a banded frame -- it produces the rows [lo, hi) of level 0 and of a-trous
    //  layer 1 its piece allows; only the last piece builds the levels below level 1 and everything else)
    const bool head_only = bands && !bands->last;
    const int l0_first = bands ? bands->level0_lo / 12 : 0;
    const int l0_tiles = (bands && !bands->last ? bands->level0_hi / 12 : (h + 11) / 12) - l0_first;
    const long ts = (long)w * h;
    const dim3 blk(32, 8);
    auto grid = [&](int gw, int gh) { return dim3((gw + 31) / 32, (gh + 7) / 8, frames); };
    // levels >= 2: binomial H at the resolution of level l-1, then V with 2x decimation; the small levels in one launch
    int l0 = pd.levels;
    while (l0 > 2 && pd.w[l0 - 2] * pd.h[l0 - 2] <= kTailPixels) --l0;
    // the kernels of levels 0 .. l0-1 write the level's apron with it where one reflection covers it; for a batch of frames
    // pyr_tail writes the aprons of its levels (one workgroup per frame: for a single frame the many workgroups of
    // pyr_apron_fill are quicker); pyr_apron_fill does what is left, levels fill_from .. fill_end-1
    const int n_direct = std::min(l0, pd.levels);
    const bool tail_aprons = frames >= 8 && l0 < pd.levels && pd.w[l0] + 2 * pd.apron[l0] <= 64 * kTailAprChunks;
    const int fill_end = tail_aprons ? n_direct : pd.levels;
    int fill_from = 0;
    while (fill_from < n_direct && std::min(pd.w[fill_from], pd.h[fill_from]) >= pd.apron[fill_from]) ++fill_from;
    auto apron_of = [&](int l) { return l < fill_from ? pd.apron[l] : 0; };
    // level 0: sigma-0.6 blur, H then V (tasks_detect.rs:150-161, mod.rs:1043-1067)
    // (image_u8 != nullptr: the frames are 8-bit, image_stride bytes apart; converted as they are read: px_f32)
    if (l0_tiles <= 0) {
    } else if (image_u8) {
        if (w % 4 == 0 && image_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(image_u8) & 3) == 0)
            hipLaunchKernelGGL(pyr_sep3_staged<unsigned char>, dim3((w + 247) / 248, l0_tiles, frames), dim3(256), 0, stream,
                               image_u8, pyr + pd.offset[0], image_stride, pyr_stride, w, h, pd.pitch[0], apron_of(0),
                               0.66381836f, 0.16809084f, 1.015267163f, l0_first);
        else
            hipLaunchKernelGGL(pyr_sep3_fused<unsigned char>, dim3((w + 255) / 256, l0_tiles, frames), dim3(256), 0, stream,
                               image_u8, pyr + pd.offset[0], image_stride, pyr_stride, w, h, pd.pitch[0], apron_of(0),
                               0.66381836f, 0.16809084f, 1.015267163f, l0_first);
    } else if (w % 4 == 0 && image_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(image) & 15) == 0)
        hipLaunchKernelGGL(pyr_sep3_staged<float>, dim3((w + 247) / 248, l0_tiles, frames), dim3(256), 0, stream, image,
                           pyr + pd.offset[0], image_stride, pyr_stride, w, h, pd.pitch[0], apron_of(0), 0.66381836f,
                           0.16809084f, 1.015267163f, l0_first);
    else
        hipLaunchKernelGGL(pyr_sep3_fused<float>, dim3((w + 255) / 256, l0_tiles, frames), dim3(256), 0, stream, image,
                           pyr + pd.offset[0], image_stride, pyr_stride, w, h, pd.pitch[0], apron_of(0), 0.66381836f,
                           0.16809084f, 1.015267163f, l0_first);
    if (pd.levels < 2) {
        launch_apron_fill(pyr, pyr_stride, pd, fill_from, fill_end, frames, stream);
        if (main_next) main_next();
        return;
    }
    // level 1: one a-trous pass over level 0, nearest-decimated.  Without a taker for the a-trous layer itself (layer1 ==
    // nullptr: no detector / orientation stage has been used on this handle) only the texels the blit picks are computed.
    const bool need_layer1 = layer1 != nullptr;
    float *l1 = layer1 ? layer1 : tmp_b;
    const long l1s = layer1 ? layer1_stride : ts;
    bool level1_done = false;
    if (need_layer1)
        level1_done = launch_swt(pyr + pd.offset[0], pyr_stride, pd.pitch[0], l1, l1s, w, h, 1, frames, stream,
                                 SwtBlit{pyr + pd.offset[1], pyr_stride, pd.pitch[1], apron_of(1)},
                                 bands ? bands->layer_lo[0] : 0, head_only ? bands->layer_hi[0] : -1);
    if (head_only) {          // a piece's share ends with the a-trous layers its rows allow (queued by the caller)
        if (main_next) main_next();
        return;
    }
    // Levels >= 1 are only read by the patch sampler: a caller whose next steps need level 0 and layer 1 alone (the
    // detector) can have them built on `rest_stream` beside those steps and wait for `join` before it samples.
    // `main_next` queues the caller's next steps on `stream` BEFORE the branch is queued: of two launches that wait for the
    // same event the one queued first starts ~6 us earlier, and the caller's are the critical path.
    if (rest_stream) {
        (void)hipEventRecord(fork, stream);
        (void)hipStreamWaitEvent(rest_stream, fork, 0);
        stream = rest_stream;
    }
    if (main_next) main_next();
    if (level1_done) {
        // level 1 came with layer 1
    } else if (need_layer1)
        hipLaunchKernelGGL(pyr_decimate, grid(pd.w[1], pd.h[1]), blk, 0, stream, (const float *)l1, pyr + pd.offset[1],
                           l1s, pyr_stride, w, h, pd.w[1], pd.h[1], pd.pitch[1], apron_of(1));
    else if (w % 4 == 0 && h % 2 == 0 && pd.pitch[0] % 4 == 0 && pyr_stride % 4 == 0 &&
             (reinterpret_cast<uintptr_t>(pyr + pd.offset[0]) & 15) == 0) {
        if (frames >= 8)
            hipLaunchKernelGGL(pyr_level1_staged<16>, dim3((pd.w[1] + 123) / 124, (pd.h[1] + 15) / 16, frames), dim3(128), 0,
                               stream, (const float *)(pyr + pd.offset[0]), pyr + pd.offset[1], pyr_stride, pyr_stride, w, h,
                               pd.pitch[0], pd.w[1], pd.h[1], pd.pitch[1], apron_of(1));
        else
            hipLaunchKernelGGL(pyr_level1_staged<8>, dim3((pd.w[1] + 123) / 124, (pd.h[1] + 7) / 8, frames), dim3(128), 0,
                               stream, (const float *)(pyr + pd.offset[0]), pyr + pd.offset[1], pyr_stride, pyr_stride, w, h,
                               pd.pitch[0], pd.w[1], pd.h[1], pd.pitch[1], apron_of(1));
    } else
        if (frames >= 8)
            hipLaunchKernelGGL(pyr_level1_fused<16>, dim3((pd.w[1] + kL1Cols - 1) / kL1Cols, (pd.h[1] + 15) / 16, frames),
                               dim3(kL1Cols), 0, stream, (const float *)(pyr + pd.offset[0]), pyr + pd.offset[1], pyr_stride,
                               pyr_stride, w, h, pd.pitch[0], pd.w[1], pd.h[1], pd.pitch[1], apron_of(1));
        else
            hipLaunchKernelGGL(pyr_level1_fused<8>, dim3((pd.w[1] + kL1Cols - 1) / kL1Cols, (pd.h[1] + 7) / 8, frames),
                               dim3(kL1Cols), 0, stream, (const float *)(pyr + pd.offset[0]), pyr + pd.offset[1], pyr_stride,
                               pyr_stride, w, h, pd.pitch[0], pd.w[1], pd.h[1], pd.pitch[1], apron_of(1));
    for (int l = 2; l < l0; ++l) {
        // the staged form where its 16-byte requests are aligned (level l-1's texel (0, 0) of every frame and its pitch)
        const bool staged = pd.w[l - 1] % 4 == 0 && pd.pitch[l - 1] % 4 == 0 && pyr_stride % 4 == 0 &&
                            (reinterpret_cast<uintptr_t>(pyr + pd.offset[l - 1]) & 15) == 0 && getenv("LF_MKD_NO_DOWN_STAGED") == nullptr;
        if (staged && frames >= 8)
            hipLaunchKernelGGL(pyr_down_staged<12>, dim3((pd.w[l] + 123) / 124, (pd.h[l] + 11) / 12, frames), dim3(128), 0, stream,
                               (const float *)(pyr + pd.offset[l - 1]), pyr + pd.offset[l], pyr_stride, pyr_stride, pd.w[l - 1],
                               pd.h[l - 1], pd.pitch[l - 1], pd.w[l], pd.h[l], pd.pitch[l], apron_of(l));
        else if (staged)
            hipLaunchKernelGGL(pyr_down_staged<6>, dim3((pd.w[l] + 123) / 124, (pd.h[l] + 5) / 6, frames), dim3(128), 0, stream,
                               (const float *)(pyr + pd.offset[l - 1]), pyr + pd.offset[l], pyr_stride, pyr_stride, pd.w[l - 1],
                               pd.h[l - 1], pd.pitch[l - 1], pd.w[l], pd.h[l], pd.pitch[l], apron_of(l));
        else
            hipLaunchKernelGGL(pyr_down_fused, dim3((pd.w[l] + 255) / 256, (pd.h[l] + kDownRows - 1) / kDownRows, frames),
                               dim3(256), 0, stream, (const float *)(pyr + pd.offset[l - 1]), pyr + pd.offset[l], pyr_stride,
                               pyr_stride, pd.w[l - 1], pd.h[l - 1], pd.pitch[l - 1], pd.w[l], pd.h[l], pd.pitch[l], apron_of(l));
    }
    if (l0 < pd.levels)
        hipLaunchKernelGGL(pyr_tail, dim3(frames), dim3(1024), 0, stream, pyr, pyr_stride, pd, l0, tail_aprons ? 1 : 0);
    launch_apron_fill(pyr, pyr_stride, pd, fill_from, fill_end, frames, stream);
    if (rest_stream) (void)hipEventRecord(join, rest_stream);
}

// Layers 1 .. n_layers-1 of the a-trous stack (mod.rs:1093-1130): layer l+1 = [1 4 6 4 1]/16 H then V over layer l
// with taps 2^l apart.  Layer 0 is pyramid level 0 (the sigma-0.6 blur), so it is read in place.
void launch_build_coarse_stack(const float *layer0, long layer0_stride, int layer0_pitch, float *coarse, long coarse_stride,
                               long layer_stride, float *tmp, int n_layers, int first_layer, int w, int h, int frames,
                               hipStream_t stream, const RowBands *bands) {
    (void)tmp;
    const bool head_only = bands && !bands->last;
    for (int l = first_layer; l + 1 < n_layers; ++l) {   // first_layer = 1: layer 1 came with the pyramid
        const float *in = l == 0 ? layer0 : coarse + (long)(l - 1) * layer_stride;
        const long in_stride = l == 0 ? layer0_stride : coarse_stride;
        launch_swt(in, in_stride, l == 0 ? layer0_pitch : w, coarse + (long)l * layer_stride, coarse_stride, w, h, 1 << l, frames,
                   stream, SwtBlit{nullptr, 0, 0, 0}, bands ? bands->layer_lo[l] : 0, head_only ? bands->layer_hi[l] : -1);
    }
}

// Which rows each stage of the pipeline's front can produce from the frame's first `raw_rows` rows (see RowBands): level 0
// needs two raw rows below an output row, a-trous layer l + 1 (dilation d = 2^l) needs 2 d rows of layer l, the extremum scan one
// row of every layer below a tile's candidates; every split falls on a boundary its kernel can start from (12 rows for
// level 0, d rows for a layer, a tile row for the scan).  Fills the `hi` fields.  False: the frame does not take row bands.
bool plan_row_bands(int raw_rows, int w, int h, int n_layers, int border, RowBands &b) {
    if (w % 4 || h % 2 || n_layers - 1 > 8 || raw_rows >= h) return false;
    b.level0_hi = raw_rows > 2 ? (raw_rows - 2) / 12 * 12 : 0;
    int prev = b.level0_hi;
    for (int l = 0; l < 8; ++l) b.layer_hi[l] = 0;
    for (int l = 0; l + 1 < n_layers; ++l) {
        const int d = 1 << l;
        b.layer_hi[l] = prev - 2 * d > 0 ? (prev - 2 * d) / d * d : 0;
        prev = b.layer_hi[l];
    }
    b.scan_hi = prev - border - 1 > 0 ? (prev - border - 1) / 8 : 0;     // tile rows of 8 candidates from row `border`
    return true;
}

}  // namespace lfmkd
