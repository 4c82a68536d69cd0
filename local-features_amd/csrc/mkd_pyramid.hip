// MKD descriptor path, keypoint mode: patch pyramid, a-trous stack and patch sampling for gfx950.
//
//   pyr_*             blur.glsl, swt.glsl (all levels of the a-trous stack), blur_pyramid.glsl, patch_pyramid.rs blits
//   sample_patches    mkd/patch_gradients.glsl:42-70
// (paths under local_features/src/vulkan/)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mkd_device.h"

namespace lfmkd {

// ---------------------------------------------------------------------------------------------
// Keypoint mode: pyramid and sampling.  Sampler = linear filter, MirroredRepeat (mod.rs:940-943),
// restated with exact f32 weights; texel centres at i + 0.5.
// ---------------------------------------------------------------------------------------------
namespace {

// MirroredRepeat: t = i mod 2n, index = min(t, 2n-1-t).  Branch-free, float reciprocal instead of an integer
// division (|i| stays far below 2^23, so the float arithmetic is exact up to the +-1 fix-ups).
__device__ __forceinline__ int mirror_idx(int i, int n) {
    const int pp = 2 * n;
    const float q = floorf((float)i * (1.f / (float)pp));
    int t = i - (int)q * pp;
    t = t < 0 ? t + pp : t;
    t = t >= pp ? t - pp : t;
    const int r = t < n ? t : pp - 1 - t;
    return r < 0 ? 0 : (r > n - 1 ? n - 1 : r);   // only binding for absurd |i| (non-finite caller data): never out of range
}

__device__ __forceinline__ float tex_bilinear(const float *__restrict__ img, int w, int h, float u, float v) {
    const float fu = u - 0.5f, fv = v - 0.5f;
    const float x0f = floorf(fu), y0f = floorf(fv);
    const float ax = fu - x0f, ay = fv - y0f;
    const int x0 = mirror_idx((int)x0f, w), x1 = mirror_idx((int)x0f + 1, w);
    const int y0 = mirror_idx((int)y0f, h), y1 = mirror_idx((int)y0f + 1, h);
    const float *r0 = img + y0 * w, *r1 = img + y1 * w;   // a level holds < 2^31 texels
    const float t00 = r0[x0], t10 = r0[x1];
    const float t01 = r1[x0], t11 = r1[x1];
    const float top = t00 * (1.f - ax) + t10 * ax;
    const float bot = t01 * (1.f - ax) + t11 * ax;
    return top * (1.f - ay) + bot * ay;
}

}  // namespace

// All pyramid kernels work on a batch of frames of one size: blockIdx.z = frame, consecutive frames are
// in_stride / out_stride floats apart.
// blur.glsl:34-65 (sigma 0.6) and blur_pyramid.glsl horizontal pass share this shape:
// out = w0 * tex(c) + w1 * (tex(c - off) + tex(c + off)) along one axis.  The bilinear fetch is evaluated exactly as
// tex_bilinear does, minus the terms that are multiplied by a weight of exactly 0: the centre tap sits on a texel
// centre (both fractions 0), the side taps have fraction 0 across the pass direction.
__device__ __forceinline__ float sep3_pixel(const float *__restrict__ in, int w, int h, int x, int y, float w0, float w1,
                                            float off, int vertical) {
#pragma clang fp contract(off)   // the detector's decisions sit on these values: round like the restatement they are tested against
    const float c = (float)(vertical ? y : x) + 0.5f;
    const int n = vertical ? h : w;
    const long stride = vertical ? w : 1;
    const float *line = vertical ? in + x : in + (size_t)y * w;
    float side[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float u = k == 0 ? c - off : c + off;
        const float fu = u - 0.5f;
        const float f0 = floorf(fu);
        const float a = fu - f0;
        const int i0 = mirror_idx((int)f0, n), i1 = mirror_idx((int)f0 + 1, n);
        side[k] = line[i0 * stride] * (1.f - a) + line[i1 * stride] * a;
    }
    float s = in[(size_t)y * w + x] * w0;
    s += (side[0] + side[1]) * w1;
    return s;
}

// blur.glsl's two passes (horizontal, then vertical) in one launch, for tap offsets in (1, 2): the same LDS tiling as
// pyr_swt_fused with dilation 1 -- a workgroup computes the horizontal pass of kSwtRows + 4 rows of a 256-column strip
// (slot m = virtual row y0 - 2 + m, holding the row that index mirrors to) and the vertical pass of the kSwtRows rows
// in the middle from them.  The vertical taps of row y blend rows floor(y - off) .. +1 and floor(y + off) .. +1, i.e.
// rows y-2 .. y+2.  Same arithmetic as the two pyr_sep3 dispatches: bit-identical.
__global__ __launch_bounds__(256) void pyr_sep3_fused(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                      long out_stride, int w, int h, float w0, float w1, float off) {
#pragma clang fp contract(off)
    __shared__ float s_h[16][256];   // kSwtRows + 4 rows
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int y0 = (int)blockIdx.y * 12;
    const int xr = (int)blockIdx.x * 256 + (int)threadIdx.x, x = xr < w ? xr : w - 1;
#pragma unroll 4
    for (int m = 0; m < 16; ++m) s_h[m][threadIdx.x] = sep3_pixel(in, w, h, x, mirror_idx(y0 - 2 + m, h), w0, w1, off, 0);
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
        if (xr < w) out[(size_t)y * w + xr] = sum;
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
                                                     long out_stride, int w, int h, int d, int blocks_per_class) {
#pragma clang fp contract(off)
    __shared__ float s_h[kSwtRows + 4][kSwtCols];
    const float k0 = 6.f / 16.f, k1 = 4.f / 16.f, k2 = 1.f / 16.f;
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    const int r = blockIdx.y / blocks_per_class;                    // residue class of the rows
    const int kb = (blockIdx.y - r * blocks_per_class) * kSwtRows;  // first lattice index of this workgroup
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
        const float *row = in + (size_t)mirror_idx(v, h) * w;
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
        if (y >= h) break;
        float sum = s_h[k + 2][threadIdx.x] * k0;
        sum += s_h[k + 1][threadIdx.x] * k1;
        sum += s_h[k][threadIdx.x] * k2;
        sum += s_h[k + 4][threadIdx.x] * k2;
        sum += s_h[k + 3][threadIdx.x] * k1;
        if (xr < w) out[(size_t)y * w + xr] = sum;
    }
}

// Nearest blit [0,w)x[0,h) -> [0,w/2)x[0,h/2): patch_pyramid.rs:251-285.
__global__ void pyr_decimate(const float *__restrict__ in, float *__restrict__ out, long in_stride, long out_stride,
                             int w, int h, int ow, int oh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= ow || y >= oh) return;
    in += blockIdx.z * in_stride;
    out += blockIdx.z * out_stride;
    int sx = (int)floorf(((float)x + 0.5f) * (float)w / (float)(w / 2));
    int sy = (int)floorf(((float)y + 0.5f) * (float)h / (float)(h / 2));
    sx = sx > w - 1 ? w - 1 : sx;
    sy = sy > h - 1 ? h - 1 : sy;
    out[(size_t)y * ow + x] = in[(size_t)sy * w + sx];
}

// blur_pyramid.glsl:36-49 vertical pass: binomial taps centred on texel (2x, 2y) of the H result.
__device__ __forceinline__ float down_v_pixel(const float *__restrict__ in, int w, int h, int x, int y) {
#pragma clang fp contract(off)
    // taps centred on texel (2x, 2y): same arithmetic as tex_bilinear, zero-weight terms left out (see sep3_pixel)
    const int sx = mirror_idx(2 * x, w);
    const float cy = 2.f * (float)y + 0.5f;
    float side[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float u = k == 0 ? cy - 1.2f : cy + 1.2f;
        const float fu = u - 0.5f;
        const float f0 = floorf(fu);
        const float a = fu - f0;
        const int i0 = mirror_idx((int)f0, h), i1 = mirror_idx((int)f0 + 1, h);
        side[k] = in[(size_t)i0 * w + sx] * (1.f - a) + in[(size_t)i1 * w + sx] * a;
    }
    float s = in[(size_t)mirror_idx(2 * y, h) * w + sx] * 0.375f;
    s += (side[0] + side[1]) * 0.3125f;
    return s;
}

// blur_pyramid.glsl's two passes for one level in one launch: the horizontal pass is only needed at the even columns and
// at the 2 kDownRows + 3 rows around the output rows (taps at 2y -+ 1.2 blend rows 2y-2 .. 2y+2), kept in LDS; the
// two-dispatch form computes it for every texel of level l-1 and writes it out.  Same pixel arithmetic: bit-identical.
constexpr int kDownRows = 6;

__global__ __launch_bounds__(256) void pyr_down_fused(const float *__restrict__ in, float *__restrict__ out, long in_stride,
                                                      long out_stride, int pw, int ph, int ow, int oh) {
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
        s_h[m][threadIdx.x] = sep3_pixel(in, pw, ph, sx, mirror_idx(v0 + m, ph), 0.375f, 0.3125f, 1.2f, 0);
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
        if (xr < ow) out[(size_t)y * ow + xr] = sum;
    }
}

// The small end of the pyramid in one launch: from level l0 on (where the horizontal result of level l-1 fits the
// 64 KiB LDS tile) one workgroup per frame walks the remaining levels, the horizontal pass into LDS, the decimating
// vertical pass from it.  Same pixel functions as the per-level kernels; it only replaces a dozen tiny launches.
constexpr int kTailPixels = 16384;

__global__ __launch_bounds__(1024) void pyr_tail(float *__restrict__ pyr, long pyr_stride, PyramidDesc pd, int l0) {
    __shared__ float s_tmp[kTailPixels];
    float *base = pyr + blockIdx.x * pyr_stride;
    for (int l = l0; l < pd.levels; ++l) {
        const int pw = pd.w[l - 1], ph = pd.h[l - 1], ow = pd.w[l], oh = pd.h[l];
        const float *in = base + pd.offset[l - 1];
        for (int i = threadIdx.x; i < pw * ph; i += 1024) {
            const int y = i / pw, x = i - y * pw;
            s_tmp[i] = sep3_pixel(in, pw, ph, x, y, 0.375f, 0.3125f, 1.2f, 0);
        }
        __syncthreads();
        float *out = base + pd.offset[l];
        for (int i = threadIdx.x; i < ow * oh; i += 1024) {
            const int y = i / ow, x = i - y * ow;
            out[i] = down_v_pixel(s_tmp, pw, ph, x, y);
        }
        __threadfence_block();   // level l is the input of level l + 1, read by other threads of this workgroup
        __syncthreads();
    }
}

constexpr int kSampleBox = 96;   // >= 32 * 2 * sqrt2 + 4: the bounding box of every footprint with rem < 2

// patch_gradients.glsl:42-70.  One wave per keypoint (4 per block): the per-keypoint scale/level/rotation math is
// done once per wave instruction, each lane then samples 16 pixels.  The kernel is bound by the number of cache lines
// its gathers touch (counters: 67 % of wave time waiting on loads, 10 % issuing), so a load instruction covers an
// 8 x 8 block of the patch -- a compact footprint of ~11 texel rows -- rather than two 32-pixel patch rows along a
// rotated line (~30 lines); the patch is transposed through LDS so that it still leaves in 256-byte row stores.
// frame_of_kp (optional) selects the keypoint's pyramid among the frames of the batch (pyr_stride floats apart).
__global__ __launch_bounds__(256) void sample_patches(const float *__restrict__ pyr, long pyr_stride, PyramidDesc pd,
                                                      const float *__restrict__ kps /*[n][5]*/,
                                                      const unsigned *__restrict__ frame_of_kp, long n_host,
                                                      const unsigned long long *__restrict__ n_dev, float psf,
                                                      float *__restrict__ patches) {
    __shared__ int s_mirror[4][2 * kSampleBox];
    __shared__ float s_patch[4][1024];
    const long n = n_dev ? (long)*n_dev : n_host;
    const long k = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    if (frame_of_kp) pyr += (long)frame_of_kp[k] * pyr_stride;
    const float *kp = kps + k * 5;
    const float scale = kp[2] * psf / 32.f;
    const float l2 = log2f(scale);
    float lvl = floorf(l2);
    lvl = lvl < 0.f ? 0.f : (lvl > (float)(pd.levels - 1) ? (float)(pd.levels - 1) : lvl);
    const float rem = exp2f(l2 - lvl);
    int l = (int)lvl;   // a non-finite size (caller-supplied keypoints) must not index outside the pyramid
    l = l < 0 ? 0 : (l > pd.levels - 1 ? pd.levels - 1 : l);
    const float ang = kp[3] * (3.14159265358979323846f / 180.f);
    const float ca = cosf(ang), sa = sinf(ang);
    const float inv = 1.f / exp2f(lvl);
    const float *img = pyr + pd.offset[l];
    const int w = pd.w[l], h = pd.h[l];
    float *tile = s_patch[threadIdx.x >> 6];
    // pixel of this lane in step i: block (i & 3, i >> 2) of 8 x 8 pixels, lane = 8 * row + column inside it
    const int lx0 = lane & 7, ly0 = lane >> 3;
    // When the rotated patch footprint (half diagonal 16 sqrt2 rem, plus the bilinear neighbour) stays inside the level,
    // MirroredRepeat is the identity and its index arithmetic is skipped: same texels, same weights.
    const float reach = 22.7f * rem + 2.f, pcx = kp[0] * inv, pcy = kp[1] * inv;
    const bool interior = pcx - reach >= 0.f && pcx + reach <= (float)(w - 1) && pcy - reach >= 0.f &&
                          pcy + reach <= (float)(h - 1);   // uniform over the wave
    // Otherwise MirroredRepeat is needed, but only for the <= 96 texel columns and rows of the footprint's bounding
    // box: computed once per keypoint into a small LDS table instead of four times per sample.
    bool boxed = false;
    int bx0 = 0, by0 = 0, bw = 2, bh = 2;
    int *tab = s_mirror[threadIdx.x >> 6];   // [0, 96): columns, [96, 192): row offsets y * w
    if (!interior) {
        float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float cdx = (c & 1) ? 15.f : -16.f, cdy = (c & 2) ? 15.f : -16.f;
            const float sx = (cdx * ca - cdy * sa) * rem + pcx, sy = (cdx * sa + cdy * ca) * rem + pcy;
            xlo = fminf(xlo, sx); xhi = fmaxf(xhi, sx);
            ylo = fminf(ylo, sy); yhi = fmaxf(yhi, sy);
        }
        const float bxf = floorf(xlo) - 1.f, byf = floorf(ylo) - 1.f;   // one texel of margin: +1 neighbour, rounding
        const float bwf = floorf(xhi) + 3.f - bxf, bhf = floorf(yhi) + 3.f - byf;
        // (comparisons are false for NaN: a non-finite keypoint takes the general path)
        boxed = bwf >= 1.f && bwf <= (float)kSampleBox && bhf >= 1.f && bhf <= (float)kSampleBox &&
                fabsf(bxf) < 1e9f && fabsf(byf) < 1e9f;
        if (boxed) {
            bx0 = (int)bxf; by0 = (int)byf; bw = (int)bwf; bh = (int)bhf;
            for (int i = lane; i < bw; i += 64) tab[i] = mirror_idx(bx0 + i, w);
            for (int i = lane; i < bh; i += 64) tab[kSampleBox + i] = mirror_idx(by0 + i, h) * w;
            __builtin_amdgcn_wave_barrier();   // written and read by this wave only; LDS keeps a wave's accesses in order
        }
    }
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int lx = 8 * (i & 3) + lx0, ly = 8 * (i >> 2) + ly0;
        const float dx = (float)lx - 16.f, dy = (float)ly - 16.f;
        const float xx = dx * ca - dy * sa, yy = dx * sa + dy * ca;
        const float sx = xx * rem + kp[0] * inv, sy = yy * rem + kp[1] * inv;
        float v;
        if (interior || boxed) {   // tex_bilinear(img, w, h, sx + 0.5f, sy + 0.5f), texel indices without / from the table
            const float fu = (sx + 0.5f) - 0.5f, fv = (sy + 0.5f) - 0.5f;
            const float x0f = floorf(fu), y0f = floorf(fv);
            const float ax = fu - x0f, ay = fv - y0f;
            int x0, x1, r0, r1;
            if (interior) {
                x0 = (int)x0f; x1 = x0 + 1; r0 = (int)y0f * w; r1 = r0 + w;
            } else {
                int ix = (int)x0f - bx0, iy = (int)y0f - by0;
                ix = ix < 0 ? 0 : (ix > bw - 2 ? bw - 2 : ix);   // never binding (margin); keeps the table reads in range
                iy = iy < 0 ? 0 : (iy > bh - 2 ? bh - 2 : iy);
                x0 = tab[ix]; x1 = tab[ix + 1]; r0 = tab[kSampleBox + iy]; r1 = tab[kSampleBox + iy + 1];
            }
            const float top = img[r0 + x0] * (1.f - ax) + img[r0 + x1] * ax;
            const float bot = img[r1 + x0] * (1.f - ax) + img[r1 + x1] * ax;
            v = top * (1.f - ay) + bot * ay;
        } else {
            v = tex_bilinear(img, w, h, sx + 0.5f, sy + 0.5f);
        }
        tile[ly * 32 + lx] = v;
    }
    __builtin_amdgcn_wave_barrier();
    float *dst = patches + k * 1024 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j * 64] = tile[j * 64 + lane];
}

// The same stage with the texels of INTERIOR footprints staged through LDS.  The gather form above is bound by the
// texture-address path: 4096 scattered lane-loads per keypoint, ~3500 L1 accesses.  Here the patch is handled in four
// 16 x 16 pixel quadrants; where the bounding box of a quadrant's footprint (at most 48 x 48 texels for every scale
// remainder < 2 and every angle) lies inside the pyramid level with room to spare, it is copied into LDS by LDS-DMA --
// 16 bytes per lane over consecutive addresses, five box rows per request, no registers -- and the four bilinear taps of
// a sample are two ds_read2_b32.  A quadrant whose box touches the level's border (MirroredRepeat) takes the gather path
// for its 256 samples.  This form is bound by the instructions it issues, so everything uniform over the wave lives in
// scalar registers, the copy is five scalar instructions per request, per-lane constants are hoisted out of the sample loop.
// Arithmetic of a sample (coordinates, floor / fraction, blend) follows the gather form term by term.
constexpr int kQuadBox = 48;    // texels per box row: >= 15 * 2 * sqrt2 + 5
constexpr int kQuadRows = 50;   // box rows: ten requests of five

// (uniform base) + (32-bit lane offset): the SGPR-base addressing form, no 64-bit VALU add per request (see the
// describe kernel's lds_dma16_sv for the two empty asm statements)
__device__ __forceinline__ void lds_dma16_sv(const unsigned char *uniform_base, unsigned lane_off, void *ldst) {
    asm("" : "+s"(uniform_base));
    asm volatile("" : "+v"(lane_off));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uniform_base + lane_off),
                                     (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
}

__global__ __launch_bounds__(256) void sample_patches_lds(const float *__restrict__ pyr, long pyr_stride, PyramidDesc pd,
                                                          const float *__restrict__ kps /*[n][5]*/,
                                                          const unsigned *__restrict__ frame_of_kp, long n_host,
                                                          const unsigned long long *__restrict__ n_dev, float psf,
                                                          float *__restrict__ patches) {
    __shared__ __attribute__((aligned(16))) float s_box[4][kQuadRows * kQuadBox];
    const long n = n_dev ? (long)*n_dev : n_host;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long k = (long)blockIdx.x * 4 + wave;
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    if (frame_of_kp) pyr += (long)__builtin_amdgcn_readfirstlane((int)frame_of_kp[k]) * pyr_stride;
    const float *kp = kps + k * 5;
    const float scale = kp[2] * psf / 32.f;
    const float l2 = log2f(scale);
    float lvl = floorf(l2);
    lvl = lvl < 0.f ? 0.f : (lvl > (float)(pd.levels - 1) ? (float)(pd.levels - 1) : lvl);
    const float rem = exp2f(l2 - lvl);
    int l = (int)lvl;   // a non-finite size (caller-supplied keypoints) must not index outside the pyramid
    l = l < 0 ? 0 : (l > pd.levels - 1 ? pd.levels - 1 : l);
    l = __builtin_amdgcn_readfirstlane(l);   // uniform, but computed on the vector side: level geometry into scalar registers
    const float ang = kp[3] * (3.14159265358979323846f / 180.f);
    const float ca = cosf(ang), sa = sinf(ang);
    const float inv = 1.f / exp2f(lvl);
    const float *img = pyr + pd.offset[l];
    const int w = pd.w[l], h = pd.h[l];
    const float cx = kp[0] * inv, cy = kp[1] * inv;
    float *box = s_box[wave];
    float *dst = patches + k * 1024;
    // a step samples 4 patch rows x 16 columns of the quadrant
    const int col = lane & 15, row4 = lane >> 4;
    const float colf = (float)col - 16.f, row4f = (float)row4 - 16.f;
    float *dst_lane = dst + row4 * 32 + col;
    // copy: lane = (row within a group of five, 16-byte chunk within the row)
    const int sub = lane / 12, chunk = lane - 12 * sub;
    const unsigned copy_off = ((unsigned)sub * (unsigned)w + 4u * (unsigned)chunk) * 4u;
    // A quadrant's samples lie within +-ext of its centre in x and in y (half-size 7.5 pixels, rotated, scaled); with the
    // +1 neighbour and margin either side (rounding of the centre against the samples' own arithmetic) the box is `bsz`
    // texels wide and high, the same for the four quadrants.
    const float ext = 7.5f * rem * (fabsf(ca) + fabsf(sa));
    const float bszf = floorf(2.f * ext) + 6.f;
    // (comparisons are false for NaN: a non-finite keypoint takes the gather path)
    const bool finite = bszf >= 6.f && bszf <= (float)kQuadBox && fabsf(cx) < 1e9f && fabsf(cy) < 1e9f;
    const int bsz = __builtin_amdgcn_readfirstlane(finite ? (int)bszf : kQuadBox);
    const int n_req = (bsz + 4) / 5, rows = 5 * n_req;   // rows <= kQuadRows
    const int n_chunk = (bsz + 3) / 4;                   // 16-byte chunks of a box row that are needed
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int px0 = 16 * (q & 1), py0 = 16 * (q >> 1);
        // centre of the quadrant (pixel offset 7.5 into it), by the samples' expression
        const float qdx = (float)px0 - 8.5f, qdy = (float)py0 - 8.5f;
        const float qx = (qdx * ca - qdy * sa) * rem + cx, qy = (qdx * sa + qdy * ca) * rem + cy;
        const int bx0 = __builtin_amdgcn_readfirstlane(finite ? (int)floorf(qx - ext) : -1) - 2;
        const int by0 = __builtin_amdgcn_readfirstlane(finite ? (int)floorf(qy - ext) : -1) - 2;
        if (!(bx0 >= 0 && bx0 + 4 * n_chunk <= w && by0 >= 0 && by0 + rows <= h)) {   // uniform: the gather path
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int lx = px0 + col, ly = py0 + 4 * i + row4;
                const float dx = (float)lx - 16.f, dy = (float)ly - 16.f;
                const float xx = dx * ca - dy * sa, yy = dx * sa + dy * ca;
                const float sx = xx * rem + kp[0] * inv, sy = yy * rem + kp[1] * inv;
                dst[ly * 32 + lx] = tex_bilinear(img, w, h, sx + 0.5f, sy + 0.5f);
            }
            continue;
        }
        // the previous quadrant's LDS reads have returned before its texels are overwritten (LDS-DMA writes do not queue
        // behind ds_read instructions)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // Inside the level (MirroredRepeat is the identity) with room for full-width rows and whole groups of five:
        // 12 lanes cover a box row (the LDS pitch is exactly their 192 bytes), one request moves five rows.  Global
        // addresses are only 4-byte aligned (gfx950 runs in unaligned access mode).
        if (sub < 5 && chunk < n_chunk) {
            // (the offset goes through readfirstlane: left to itself hipcc forms part of this address on the vector side)
            const int first = __builtin_amdgcn_readfirstlane(by0 * w + bx0);   // a level holds < 2^31 texels
            const unsigned char *src = reinterpret_cast<const unsigned char *>(img + first);   // uniform
            for (int r = 0; r < n_req; ++r)
                lds_dma16_sv(src + (long)r * 5 * w * 4, copy_off, box + r * 5 * kQuadBox);
        }
        const float a_q = ((float)px0 + colf) * ca, b_q = ((float)px0 + colf) * sa;   // dx ca, dx sa of this lane's column
        const float bxf = (float)bx0, byf = (float)by0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dy = row4f + (float)(py0 + 4 * i);
            const float xx = __builtin_fmaf(-dy, sa, a_q), yy = __builtin_fmaf(dy, ca, b_q);
            const float sx = __builtin_fmaf(xx, rem, cx), sy = __builtin_fmaf(yy, rem, cy);
            // tex_bilinear(img, w, h, sx + 0.5f, sy + 0.5f) with the texels read from the box
            const float fu = (sx + 0.5f) - 0.5f, fv = (sy + 0.5f) - 0.5f;
            const float x0f = floorf(fu), y0f = floorf(fv);
            const float ax = fu - x0f, ay = fv - y0f;
            int ix = (int)(x0f - bxf), iy = (int)(y0f - byf);
            ix = min(max(ix, 0), bsz - 2);   // never binding (margins); keeps the LDS reads inside what was copied
            iy = min(max(iy, 0), rows - 2);
            const float *t = box + __umul24(iy, kQuadBox) + ix;   // 24-bit multiply: full rate
            const float t00 = t[0], t10 = t[1], t01 = t[kQuadBox], t11 = t[kQuadBox + 1];
            const float top = t00 * (1.f - ax) + t10 * ax;
            const float bot = t01 * (1.f - ax) + t11 * ax;
            dst_lane[(py0 + 4 * i) * 32 + px0] = top * (1.f - ay) + bot * ay;
        }
    }
}

void launch_sample_patches(const float *pyr, long pyr_stride, const PyramidDesc &pd, const float *kps,
                           const unsigned *frame_of_kp, long n, const unsigned long long *n_dev, float psf,
                           float *patches, hipStream_t stream, bool beside_describe) {
    if (n <= 0) return;
    // The LDS-staged form is 10-30 % faster on its own.  Beside the describe kernel (large batches, lf_mkd.cpp) the
    // all-gather form is used: it leaves the vector ALUs and the LDS to the describe kernel (19 KiB of LDS per workgroup
    // against 38, so that two of its workgroups fit next to a describe workgroup on a CU), and the pair is 1.5 % faster.
    // LF_MKD_SAMPLER=gather / lds forces one form everywhere (A/B timing, and the cross-check in the tests).
    static const int forced = [] { const char *e = getenv("LF_MKD_SAMPLER"); return e ? (e[0] == 'g' ? 1 : (e[0] == 'l' ? 2 : 0)) : 0; }();
    const bool gather = forced == 1 || (forced == 0 && beside_describe);
    if (gather)
        hipLaunchKernelGGL(sample_patches, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, pyr, pyr_stride, pd, kps,
                           frame_of_kp, n, n_dev, psf, patches);
    else
        hipLaunchKernelGGL(sample_patches_lds, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, pyr, pyr_stride, pd,
                           kps, frame_of_kp, n, n_dev, psf, patches);
}

// Builds the pyramids of `frames` frames (image_stride floats apart) into pyr (pyr_stride apart); tmp_a and tmp_b
// hold frames x w x h floats each.
// one a-trous layer (both passes) for `frames` frames
static void launch_swt(const float *in, long in_stride, float *out, long out_stride, int w, int h, int d, int frames,
                       hipStream_t stream) {
    const int classes = d < h ? d : h;                                   // residue classes that hold rows
    const int lattice = (h + d - 1) / d;                                 // rows of the longest class
    const int per_class = (lattice + kSwtRows - 1) / kSwtRows;
    hipLaunchKernelGGL(pyr_swt_fused, dim3((w + kSwtCols - 1) / kSwtCols, classes * per_class, frames), dim3(256), 0, stream,
                       in, out, in_stride, out_stride, w, h, d, per_class);
}

// With layer1 != nullptr the a-trous layer 1 the pyramid needs anyway is written there (frames layer1_stride apart) instead
// of tmp_b: it is layer 1 of the stack orientation and the detector read, so they need not build it again.
void launch_build_pyramid(const float *image, long image_stride, float *pyr, long pyr_stride, float *tmp_a,
                          float *tmp_b, const PyramidDesc &pd, int frames, float *layer1, long layer1_stride,
                          hipStream_t stream, hipStream_t rest_stream, hipEvent_t fork, hipEvent_t join) {
    const int w = pd.w[0], h = pd.h[0];
    const long ts = (long)w * h;
    const dim3 blk(32, 8);
    auto grid = [&](int gw, int gh) { return dim3((gw + 31) / 32, (gh + 7) / 8, frames); };
    // level 0: sigma-0.6 blur, H then V (tasks_detect.rs:150-161, mod.rs:1043-1067)
    hipLaunchKernelGGL(pyr_sep3_fused, dim3((w + 255) / 256, (h + 11) / 12, frames), dim3(256), 0, stream, image,
                       pyr + pd.offset[0], image_stride, pyr_stride, w, h, 0.66381836f, 0.16809084f, 1.015267163f);
    if (pd.levels < 2) return;
    // level 1: one a-trous pass over level 0, nearest-decimated
    float *l1 = layer1 ? layer1 : tmp_b;
    const long l1s = layer1 ? layer1_stride : ts;
    launch_swt(pyr + pd.offset[0], pyr_stride, l1, l1s, w, h, 1, frames, stream);
    // Levels >= 1 are only read by the patch sampler: a caller whose next steps need level 0 and layer 1 alone (the
    // detector) can have them built on `rest_stream` beside those steps and wait for `join` before it samples.
    if (rest_stream) {
        (void)hipEventRecord(fork, stream);
        (void)hipStreamWaitEvent(rest_stream, fork, 0);
        stream = rest_stream;
    }
    hipLaunchKernelGGL(pyr_decimate, grid(pd.w[1], pd.h[1]), blk, 0, stream, (const float *)l1, pyr + pd.offset[1],
                       l1s, pyr_stride, w, h, pd.w[1], pd.h[1]);
    // levels >= 2: binomial H at the resolution of level l-1, then V with 2x decimation; the small levels in one launch
    int l0 = pd.levels;
    while (l0 > 2 && pd.w[l0 - 2] * pd.h[l0 - 2] <= kTailPixels) --l0;
    for (int l = 2; l < l0; ++l)
        hipLaunchKernelGGL(pyr_down_fused, dim3((pd.w[l] + 255) / 256, (pd.h[l] + kDownRows - 1) / kDownRows, frames),
                           dim3(256), 0, stream, (const float *)(pyr + pd.offset[l - 1]), pyr + pd.offset[l], pyr_stride,
                           pyr_stride, pd.w[l - 1], pd.h[l - 1], pd.w[l], pd.h[l]);
    if (l0 < pd.levels) hipLaunchKernelGGL(pyr_tail, dim3(frames), dim3(1024), 0, stream, pyr, pyr_stride, pd, l0);
    if (rest_stream) (void)hipEventRecord(join, rest_stream);
}

// Layers 1 .. n_layers-1 of the a-trous stack (mod.rs:1093-1130): layer l+1 = [1 4 6 4 1]/16 H then V over layer l
// with taps 2^l apart.  Layer 0 is pyramid level 0 (the sigma-0.6 blur), so it is read in place.
void launch_build_coarse_stack(const float *layer0, long layer0_stride, float *coarse, long coarse_stride,
                               long layer_stride, float *tmp, int n_layers, int first_layer, int w, int h, int frames,
                               hipStream_t stream) {
    (void)tmp;
    for (int l = first_layer; l + 1 < n_layers; ++l) {   // first_layer = 1: layer 1 came with the pyramid
        const float *in = l == 0 ? layer0 : coarse + (long)(l - 1) * layer_stride;
        const long in_stride = l == 0 ? layer0_stride : coarse_stride;
        launch_swt(in, in_stride, coarse + (long)l * layer_stride, coarse_stride, w, h, 1 << l, frames, stream);
    }
}

}  // namespace lfmkd
