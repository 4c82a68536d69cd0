// Patch sampling arithmetic (mkd/patch_gradients.glsl:42-70), shared by the stand-alone sampler (mkd_pyramid.hip) and by the
// producer waves of the fused keypoint-mode describe kernel (mkd_describe.hip).  ONE definition, every fused multiply-add
// written as one and no implicit contraction: the two kernels sample the same bits, so a patch read through
// lf_mkd_sample_patches_device is exactly what the fused kernel described.
//
// Sampler = linear filter, MirroredRepeat (mod.rs:940-943), restated with exact f32 weights; texel centres at i + 0.5.
#pragma once
#include <hip/hip_runtime.h>

#include "mkd_device.h"

namespace lfmkd {

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

// What a keypoint (x, y, size, angle in degrees) fixes for its 1024 samples.
struct KpGeom {
    float ca, sa, rem, cx, cy;   // rotation, scale remainder in [1, 2), centre in texels of its level
    int level;
    bool covered;                // every tap of the footprint lies inside the level plus its apron
};

// Per-level geometry as plain arrays: the members of a PyramidDesc, or a copy of them in LDS (a kernel whose lanes look up
// different levels must not index its kernel argument dynamically: hipcc would move the struct to scratch memory).
struct LevelTable {
    int levels;
    const int *w, *h, *pitch, *apron;
};
__device__ __forceinline__ LevelTable level_table(const PyramidDesc &pd) {
    return LevelTable{pd.levels, pd.w, pd.h, pd.pitch, pd.apron};
}

__device__ __forceinline__ KpGeom keypoint_geometry(float x, float y, float size, float angle_deg, float psf,
                                                    const LevelTable &pd) {
#pragma clang fp contract(off)
    KpGeom g;
    const float scale = size * psf / 32.f;
    const float l2 = log2f(scale);
    float lvl = floorf(l2);
    lvl = lvl < 0.f ? 0.f : (lvl > (float)(pd.levels - 1) ? (float)(pd.levels - 1) : lvl);
    g.rem = exp2f(l2 - lvl);
    int l = (int)lvl;   // a non-finite size (caller-supplied keypoints) must not index outside the pyramid
    g.level = l < 0 ? 0 : (l > pd.levels - 1 ? pd.levels - 1 : l);
    const float ang = angle_deg * (3.14159265358979323846f / 180.f);
    g.ca = cosf(ang);
    g.sa = sinf(ang);
    const float inv = 1.f / exp2f(lvl);
    g.cx = x * inv;
    g.cy = y * inv;
    // footprint: |offset| <= 16 sqrt2 rem, plus the +1 bilinear neighbour and a texel for rounding (comparisons are false
    // for NaN: a non-finite keypoint is "not covered" and takes the mirror path, whose indices are clamped)
    const float reach = 22.7f * g.rem + 2.f, a = (float)pd.apron[g.level];
    g.covered = g.cx - reach >= -a && g.cx + reach <= (float)(pd.w[g.level] - 1) + a && g.cy - reach >= -a &&
                g.cy + reach <= (float)(pd.h[g.level] - 1) + a;
    return g;
}

// position of patch pixel (lx, ly) in texels of the keypoint's level: floor parts and fractions of the bilinear fetch at
// normalised coordinate (s + 0.5) / size
struct SamplePos { float ax, ay; int ix, iy; float fx0, fy0; };   // (fx0, fy0) = (ix, iy) as floats: integers below 2^23

__device__ __forceinline__ SamplePos sample_position(float ca, float sa, float rem, float cx, float cy, int lx, int ly) {
#pragma clang fp contract(off)
    // x and y halves of each step as one two-wide operation (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 round element by
    // element like the scalar instructions; the producer waves of the describe kernel are bound by instruction issue):
    //   rotated = (dx ca - dy sa, dx sa + dy ca), position = rotated rem + centre, then the +0.5 -0.5 of the normalised
    //   coordinate, floor and fraction
    typedef float v2 __attribute__((ext_vector_type(2)));
    const float dx = (float)lx - 16.f, dy = (float)ly - 16.f;
    const v2 cross = v2{-dy, dy} * v2{sa, ca};
    const v2 rot = __builtin_elementwise_fma(v2{dx, dx}, v2{ca, sa}, cross);
    const v2 s = __builtin_elementwise_fma(rot, v2{rem, rem}, v2{cx, cy});
    const v2 f = (s + 0.5f) - 0.5f;
    const v2 f0 = v2{floorf(f.x), floorf(f.y)};
    const v2 a = f - f0;
    SamplePos p;
    p.ax = a.x;
    p.ay = a.y;
    p.ix = (int)f0.x;
    p.iy = (int)f0.y;
    p.fx0 = f0.x;
    p.fy0 = f0.y;
    return p;
}

__device__ __forceinline__ float bilinear_blend(float t00, float t10, float t01, float t11, float ax, float ay) {
#pragma clang fp contract(off)
    const float wx = 1.f - ax, wy = 1.f - ay;
    const float top = __builtin_fmaf(t10, ax, t00 * wx), bot = __builtin_fmaf(t11, ax, t01 * wx);
    return __builtin_fmaf(bot, ay, top * wy);
}

// The four taps of a sample.  `lvl0` points at texel (0, 0) of the level.  A covered footprint is read through the apron
// (indices clamped into it: never binding for a covered footprint, and what keeps every other address inside the level's
// allocation); the others evaluate MirroredRepeat per tap.
struct Taps { float t00, t10, t01, t11; };

__device__ __forceinline__ Taps fetch_covered(const float *__restrict__ lvl0, int w, int h, int pitch, int apron,
                                              const SamplePos &p) {
    const int ix = min(max(p.ix, -apron), w + apron - 2), iy = min(max(p.iy, -apron), h + apron - 2);
    const float *t = lvl0 + (long)iy * pitch + ix;
    return Taps{t[0], t[1], t[pitch], t[pitch + 1]};
}

__device__ __forceinline__ Taps fetch_mirrored(const float *__restrict__ lvl0, int w, int h, int pitch, const SamplePos &p) {
    const int x0 = mirror_idx(p.ix, w), x1 = mirror_idx(p.ix + 1, w);
    const long r0 = (long)mirror_idx(p.iy, h) * pitch, r1 = (long)mirror_idx(p.iy + 1, h) * pitch;
    return Taps{lvl0[r0 + x0], lvl0[r0 + x1], lvl0[r1 + x0], lvl0[r1 + x1]};
}

__device__ __forceinline__ float sample_level(const float *__restrict__ lvl0, int w, int h, int pitch, int apron,
                                              bool covered, const SamplePos &p) {
    const Taps t = covered ? fetch_covered(lvl0, w, h, pitch, apron, p) : fetch_mirrored(lvl0, w, h, pitch, p);
    return bilinear_blend(t.t00, t.t10, t.t01, t.t11, p.ax, p.ay);
}

}  // namespace lfmkd
