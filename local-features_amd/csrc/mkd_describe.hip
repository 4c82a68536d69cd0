// MKD descriptor path, the describe kernel: device code for gfx950 (CDNA4, wave64).
//
//   mkd_pool   mkd/patch_gradients.glsl:72-104 + mkd/embedding.glsl:53-121 (both variants)
//              + mkd/normalize.glsl:22-142 + mkd/whitening.glsl:22-77 + mkd/normalize_final.glsl
// (paths under local_features/src/vulkan/shaders/; the other stages live in mkd_pyramid.hip -- pyramid and patch
//  sampling --, mkd_orient.hip, mkd_detect.hip and mkd_match.hip)
//
// Pooling is a GEMM with M = patches, K = pixels, N = (stream, spatial kernel) columns.  One wave
// owns 16 patches; lane l = (patch p = l & 15, segment q = l >> 4) holds the 8 pixels
// x in [8q, 8q+8) of the current patch row, which is exactly the A-operand lane map of the
// 16x16 MFMAs (row = l & 15, k-group = l >> 4).  So blur, gradients and the von-Mises
// embedding are computed in the registers that feed the matrix cores; nothing but the final
// sums leaves the wave.  Horizontal neighbours come from lanes l -/+ 16 via ds_bpermute; vertical
// neighbours from a ring of raw patch rows that LDS-DMA keeps filled.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mkd_ablate.h"
#include "mkd_device.h"
#include "mkd_sample.h"

namespace lfmkd {


namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Two adjacent pixels per register pair: arithmetic on f32x2 compiles to v_pk_{mul,add,fma}_f32, which issue
// in the time of one scalar-f32 VALU instruction (tools/micro/valu_rate.hip) -- the kernel is VALU-issue-bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_set(float v) { return f32x2{v, v}; }

// LF_POOL_F16X3 and LF_POOL_F16_FP6 share everything but the arithmetic of the harmonics' cross terms (row images by
// LDS-DMA, counted vmcnt waits, the f16x3 whitening epilogue, the m stream's three tiles)
__device__ __host__ constexpr bool f16_family(int pool) { return pool == LF_POOL_F16X3 || pool == LF_POOL_F16_FP6; }

// Seven per-pixel streams -- m, and m cos / sin (k theta) for k = 1..3 -- are pooled; the rotation by gradient_angle(px)
// that the polar kernels' streams carry (embedding.glsl:70-72) is folded into the LUT (mkd_consts.hpp):
//   LUT tiles of a patch row: 0-2 m | per harmonic h = k-1, 3 + 4h + {0: P0 = EPc[0:16], 1: Q0 = EPs[0:16],
//                                      2: R = EPc[16:25] | EC[0:7], 3: S = EPs[16:25] | EC[7:9]}
//   accumulator tiles of the row loop: 0-2 m | base 3 + 7h: cos x P0 | sin x Q0 | sin x P0 + cos x Q0 (one tile: both
//                                      products of relsin[0:16]) | cos x R | sin x R | cos x S | sin x S
//   packed output tiles after the epilogue's combine step (colmap, whitening fragments): 21, see combine_tiles.
constexpr int kAccTiles = 24;
constexpr int kTiles = 21;
constexpr int kUniqueTiles = 15;
constexpr int kLutPieces = 2 * kUniqueTiles;   // 1 KiB pieces per LUT row

// 5-tap sigma=0.7 kernel, patch_gradients.glsl:22-28
constexpr float kB0 = 0.0096f, kB1 = 0.2054f, kB2 = 0.5699f;

// LDS map of the pooling kernel (bytes)
constexpr int kRowBytes = kLutPieces * 1024;          // 30720: one LUT row image, 2 x 1 KiB pieces per tile
constexpr int kRingOff = 2 * kRowBytes;              // raw patch rows: [wave 8][slot 6][2 KiB]
constexpr int kRingSlots = 6;
// total: kRingOff + waves * kRingSlots * 2048 = 159744 B for 8 waves, 110592 B for 4
// Keypoint mode (SRC = kSrcKeypoints): the ring of a describe wave is filled by a producer wave of the same workgroup, four
// patch rows at a time (a group), so it holds 12 rows: the 8 a describe wave reads while a group is being written + that
// group.  Raw row r of the workgroup's it-th batch lives in slot (8 it + r) mod 12 (32 mod 12 = 8: the numbering simply
// runs on across batches).  4 describe waves: kRingOff + 4 * 12 * 2048 = 159744 B, plus the level table.
// SRC = kSrcKeypointsSplit (round 6; requests of at most 4096 keypoints): the 32 rows of a batch's patches are divided among
// R = 2 or 4 WORKGROUPS (row-split form, see mkd_pool) whose partial pooled sums meet in global memory.
[[maybe_unused]] constexpr int kSrcPatches = 0, kSrcKeypoints = 1, kSrcKeypointsSplit = 2;
constexpr int kRingSlotsKp = 12;
constexpr int kLevelTableBytes = 5 * kMaxPyrLevels * 4;

__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

__device__ __forceinline__ void lds_dma16(const void *gsrc, void *ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
}
// (uniform base) + (32-bit lane offset).  The base goes through an opaque SGPR constraint so that hipcc cannot
// re-associate it with the lane offset into a per-lane 64-bit base plus a uniform offset (which it then hoists).
__device__ __forceinline__ void lds_dma16_sv(const unsigned char *uniform_base, unsigned lane_off, void *ldst) {
    asm("" : "+s"(uniform_base));
    // volatile: keeps the 32-bit offset's zero-extension in the basic block of its use, where instruction selection can
    // fold it into the addressing mode (hoisted out of the row loop it becomes a 64-bit VGPR pair and an add per request)
    asm volatile("" : "+v"(lane_off));
    lds_dma16(uniform_base + lane_off, ldst);
}

// Global addresses of the DMA requests are formed as (wave-uniform 64-bit base) + (32-bit lane offset), in that order:
// hipcc then selects the SGPR-base form `global_load_lds_dwordx4 v_off, s[base:base+1]`, which costs no 64-bit VALU add and
// no VGPR pair per request (written as per-lane pointer + uniform offset, every request carried its own v_lshl_add_u64
// and the hoisted address pairs were spilled).
// one LUT row (30 pieces) into an LDS row buffer, piece p by wave p mod W (the first waves take one piece more)
template <int W>
__device__ __forceinline__ void issue_lut_row(const unsigned char *__restrict__ lut_rows, int row,
                                              unsigned char *lds_row, int wave, int lane) {
    const unsigned lane16 = (unsigned)lane * 16u;
#pragma unroll
    for (int j = 0; j < (kLutPieces + W - 1) / W; ++j) {
        if (wave + W * j < kLutPieces) {   // uniform
            const unsigned char *g = lut_rows + ((size_t)row * kRowBytes + (size_t)(wave + W * j) * 1024);   // uniform
            lds_dma16_sv(g, lane16, lds_row + (wave + W * j) * 1024);
        }
    }
}

// Whitening fragments are staged through the two LUT row buffers (idle during the epilogue) one 16 KiB step at a
// time, shared by the workgroup's waves: three slots, placed so that step 0 lies in row buffer 0 (it is requested during
// patch row 31, when only that buffer is free) and the last step in row buffer 1 (so that LUT row 0 of the next batch
// can be requested into buffer 0 while the last step is still being consumed).
constexpr int kWStepBytes = 16384;
__device__ __forceinline__ constexpr int wstage_slot(int step) { return step % 3 == 0 ? 0 : (step % 3 == 1 ? 32768 : 16384); }
static_assert(wstage_slot(0) + kWStepBytes <= kRowBytes && wstage_slot(10) >= kRowBytes && wstage_slot(10) + kWStepBytes <= 2 * kRowBytes, "");

template <int W>
__device__ __forceinline__ void issue_w_step(const unsigned char *__restrict__ wfrag, int step, unsigned char *lds,
                                             int wave, int lane) {
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned char *d = lds + wstage_slot(step);
#pragma unroll
    for (int j = 0; j < 16 / W; ++j) {
        const unsigned char *g = wfrag + ((size_t)step * kWStepBytes + (size_t)(wave + W * j) * 1024);   // uniform
        lds_dma16_sv(g, lane16, d + (wave + W * j) * 1024);
    }
}

// Counted wait on the vector-memory counter.  INVARIANT: between the DMA requests this counts and the wait itself, the
// wave must issue no other vector-memory instruction -- in particular no scratch spill: spill stores share vmcnt and retire
// out of order with respect to loads, so one in flight lets the wait pass before the counted DMA has landed.  hipcc
// inserts spills on its own when registers run out; the Makefile therefore rejects an f16 build of this kernel whose
// ScratchSize is not 0 (tools/check_scratch.py), and the raw_out verification tap, which stores, waits with vmcnt(0).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one raw patch row (16 patches x 128 B) of this wave into ring slot `slot`; rows outside [0,31] replicate.
// src = the wave's (uniform) first patch, lane_off = byte offset of the lane's patch and 16-byte chunk from it.
struct RawSrc { const unsigned char *base; unsigned lane_off; };
__device__ __forceinline__ void issue_raw_row(const RawSrc &src, int row, unsigned char *ring, int slot) {
    const int y = row < 0 ? 0 : (row > 31 ? 31 : row);
    const unsigned char *g = src.base + y * 128;   // uniform
    lds_dma16_sv(g, src.lane_off, ring + slot * 2048);
    lds_dma16_sv(g + 64, src.lane_off, ring + slot * 2048 + 1024);
}

// No implicit contraction in the describe kernel (down to the end of mkd_pool): every fused multiply-add in it is written
// as one.  The 4-wave and 8-wave forms are separate instantiations, and left to itself hipcc may contract an
// expression in one and not in the other -- a descriptor must not depend on the size of the request it was part of
// (tests/test_gpu_parity.py::test_full_size_properties compares the two forms bit for bit).
#pragma clang fp contract(off)

// max(a, b, 1e-30) in one instruction (a, b >= 0): the larger gradient component as a divisor that is never 0
__device__ __forceinline__ float max3_tiny(float a, float b) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), 1e-30f);   // v_max3_f32, with the |.| of its operands folded in
}

// cos/sin of the gradient angle theta = -atan2(gy over gx), for a pair of pixels.
template <int ANGLE>
__device__ __forceinline__ void gradient_direction(f32x2 gx, f32x2 gy, f32x2 r2, f32x2 &ct, f32x2 &st) {
    if (ANGLE == LF_ANGLE_EXACT || ANGLE == LF_ANGLE_EXACT_ZERO) {
        // r2 = gx^2 + gy^2 comes from the caller (it is the magnitude's radicand before its epsilon)
        const f32x2 inv = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
        const f32x2 c = gx * inv, s = -gy * inv;
        // EXACT: angle 0 only for the null gradient (the CPU twin's atan2(0, 0)); EXACT_ZERO: wherever gx == 0, which is
        // the shader's convention (atan2.glsl:33-38) -- the one place where the two angle definitions are far apart
        const bool z0 = ANGLE == LF_ANGLE_EXACT ? r2.x == 0.f : gx.x == 0.f;
        const bool z1 = ANGLE == LF_ANGLE_EXACT ? r2.y == 0.f : gx.y == 0.f;
        ct = f32x2{z0 ? 1.f : c.x, z1 ? 1.f : c.y};
        st = f32x2{z0 ? 0.f : s.x, z1 ? 0.f : s.y};
    } else {
        // atan2.glsl:19-46 called as atan2(x = gx, y = gy): a = (smaller / larger component), p = poly(a), then
        //   swap (|x| < |y|):  res = sign(a) pi/2 - p;   x < 0: res += +-pi.
        // poly is odd, so with a' = |a| and p' = poly(a') = |p| these branches are the octant symmetries
        //   (|cos res|, |sin res|) = swap ? (sin p', cos p') : (cos p', sin p'),  sign(cos res) = sign(x),  sign(sin res) = sign(y)
        // (case by case from the three lines above), and the first octant needs no signs and no select of num / den:
        const float ax0 = fabsf(gx.x), ay0 = fabsf(gy.x), ax1 = fabsf(gx.y), ay1 = fabsf(gy.y);
        // gx == 0: the shader returns angle 0 both for atan2(0, 0) and (its quirk) for atan2(0, y != 0), i.e.
        // (cos, sin) = (1, 0).  That is the un-swapped branch at a = 0 (p = 0: cos p = 1, sin p = 0), so such a pixel simply
        // does not swap, and the denominator is kept away from 0 (0 x rcp(0) would be NaN) -- no selects afterwards
        const bool sw0 = ax0 < ay0 && gx.x != 0.f, sw1 = ax1 < ay1 && gx.y != 0.f;
        const f32x2 mn = {__builtin_fminf(ax0, ay0), __builtin_fminf(ax1, ay1)};
        const f32x2 mx = {max3_tiny(ax0, ay0), max3_tiny(ax1, ay1)};
        const f32x2 a = mn * f32x2{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
        const f32x2 s = a * a;
        f32x2 p = pk_fma(s, pk_set(-0.0117212f), pk_set(0.05265332f));
        p = pk_fma(s, p, pk_set(-0.11643287f));
        p = pk_fma(s, p, pk_set(0.19354346f));
        p = pk_fma(s, p, pk_set(-0.33262347f));
        p = pk_fma(s, p, pk_set(0.99997726f));
        p = a * p;
        // 0 <= p <= 0.7854: minimax fits in p^2 (Remez on [0, (pi/4)^2], float64, rounded to f32) -- 2.4e-9 / 2.8e-8
        // before rounding, one term shorter than the Taylor series of the same accuracy
        const f32x2 p2 = p * p;
        f32x2 sn = pk_fma(p2, pk_set(-0.000195038549f), pk_set(0.0083320355f));
        sn = pk_fma(p2, sn, pk_set(-0.166666508f));
        sn = pk_fma(p2, sn, pk_set(1.f));
        sn = p * sn;
        f32x2 cs = pk_fma(p2, pk_set(-0.00135857589f), pk_set(0.0416550152f));
        cs = pk_fma(p2, cs, pk_set(-0.499998569f));
        cs = pk_fma(p2, cs, pk_set(1.f));
        // cos theta = cos res: sign of gx; sin theta = -sin res: opposite sign of gy (sn, cs >= 0: OR the sign bit in)
        const unsigned sgn = 0x80000000u;
        float cr0 = __uint_as_float(__float_as_uint(sw0 ? sn.x : cs.x) | (__float_as_uint(gx.x) & sgn));
        float cr1 = __uint_as_float(__float_as_uint(sw1 ? sn.y : cs.y) | (__float_as_uint(gx.y) & sgn));
        float sr0 = __uint_as_float(__float_as_uint(sw0 ? cs.x : sn.x) | (~__float_as_uint(gy.x) & sgn));
        float sr1 = __uint_as_float(__float_as_uint(sw1 ? cs.y : sn.y) | (~__float_as_uint(gy.y) & sgn));
        ct = f32x2{cr0, cr1};
        st = f32x2{sr0, sr1};
    }
}

// The same for the lane's four pixel pairs at once, written step by step across the pairs: the polynomial chains of a
// pair are serial (each packed fma waits for the one before it, and hipcc pads back-to-back dependent packed operations
// with s_nop), the four pairs are independent -- in this order every instruction has three others between it and the one
// it depends on.  Same operations on the same values as gradient_direction: bit-identical.
template <int ANGLE>
__device__ __forceinline__ void gradient_direction4(const f32x2 (&gx)[4], const f32x2 (&gy)[4], const f32x2 (&r2)[4],
                                                    f32x2 (&ct)[4], f32x2 (&st)[4]) {
    if (ANGLE != LF_ANGLE_SHADER) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gradient_direction<ANGLE>(gx[e], gy[e], r2[e], ct[e], st[e]);
        return;
    }
    f32x2 a[4], s[4], p[4], p2[4], sn[4], cs[4];
    bool sw0[4], sw1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ax0 = fabsf(gx[e].x), ay0 = fabsf(gy[e].x), ax1 = fabsf(gx[e].y), ay1 = fabsf(gy[e].y);
        sw0[e] = ax0 < ay0 && gx[e].x != 0.f;   // gx == 0 -> (1, 0): the un-swapped branch at a = 0 (see gradient_direction)
        sw1[e] = ax1 < ay1 && gx[e].y != 0.f;
        const f32x2 mn = {__builtin_fminf(ax0, ay0), __builtin_fminf(ax1, ay1)};
        const f32x2 mx = {max3_tiny(ax0, ay0), max3_tiny(ax1, ay1)};
        a[e] = mn * f32x2{__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = a[e] * a[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = pk_fma(s[e], pk_set(-0.0117212f), pk_set(0.05265332f));
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = pk_fma(s[e], p[e], pk_set(-0.11643287f));
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = pk_fma(s[e], p[e], pk_set(0.19354346f));
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = pk_fma(s[e], p[e], pk_set(-0.33262347f));
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = pk_fma(s[e], p[e], pk_set(0.99997726f));
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = a[e] * p[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) p2[e] = p[e] * p[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sn[e] = pk_fma(p2[e], pk_set(-0.000195038549f), pk_set(0.0083320355f));
        cs[e] = pk_fma(p2[e], pk_set(-0.00135857589f), pk_set(0.0416550152f));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sn[e] = pk_fma(p2[e], sn[e], pk_set(-0.166666508f));
        cs[e] = pk_fma(p2[e], cs[e], pk_set(-0.499998569f));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sn[e] = pk_fma(p2[e], sn[e], pk_set(1.f));
        cs[e] = pk_fma(p2[e], cs[e], pk_set(1.f));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sn[e] = p[e] * sn[e];
    const unsigned sgn = 0x80000000u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float cr0 = __uint_as_float(__float_as_uint(sw0[e] ? sn[e].x : cs[e].x) | (__float_as_uint(gx[e].x) & sgn));
        float cr1 = __uint_as_float(__float_as_uint(sw1[e] ? sn[e].y : cs[e].y) | (__float_as_uint(gx[e].y) & sgn));
        float sr0 = __uint_as_float(__float_as_uint(sw0[e] ? cs[e].x : sn[e].x) | (~__float_as_uint(gy[e].x) & sgn));
        float sr1 = __uint_as_float(__float_as_uint(sw1[e] ? cs[e].y : sn[e].y) | (~__float_as_uint(gy[e].y) & sgn));
        ct[e] = f32x2{cr0, cr1};
        st[e] = f32x2{sr0, sr1};
    }
}

// ---- B fragments (LUT) and A fragments (streams) -------------------------------------------------
// One unique LUT tile = two 1 KiB pieces: f32: pixels 0-3 / 4-7 of the lane's segment (K = 4 MFMAs);
// f16: hi / lo halves of all 8 pixels (K = 32 MFMAs).  16 B per lane either way.
struct BFrag { u32x4 p0, p1; };

__device__ __forceinline__ BFrag load_b(const unsigned char *brow, int ut) {   // ut: a constant after unrolling
    BFrag b;
    if constexpr (ablate::kNoFragmentReads) {
        b.p0 = u32x4{(unsigned)ut, 1u, 2u, 3u};
        b.p1 = b.p0;
        return b;
    }
    b.p0 = *reinterpret_cast<const u32x4 *>(brow + (ut * 2 + 0) * 1024);
    b.p1 = *reinterpret_cast<const u32x4 *>(brow + (ut * 2 + 1) * 1024);
    return b;
}

__device__ __forceinline__ unsigned pack_rtz(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
}

// A operand of one stream: f32: the 8 values themselves; f16: hi = f16 truncation, lo = f16(a - hi)
template <int POOL> struct AFrag;
template <> struct AFrag<LF_POOL_F32> {
    float v[8];
    __device__ __forceinline__ void set(const f32x2 (&a)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = a[e].x; v[2 * e + 1] = a[e].y; }
    }
};
template <> struct AFrag<LF_POOL_F16X3> {
    u32x4 hi, lo;
    __device__ __forceinline__ void set(const f32x2 (&a)[4]) {
        // residual a - hi as ONE v_fma_mix_f32 (f32 x f32 - f16); the factor 1 is hidden from hipcc, which otherwise
        // folds the fma into an unpack plus a subtract
        if constexpr (ablate::kNoSplit) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { hi[e] = __float_as_uint(a[e].x); lo[e] = __float_as_uint(a[e].y); }
            return;
        }
        // (written step by step across the four pairs: conversion, residuals, conversion are each one dependent on the
        // other; across the pairs they are not)
        float one = 1.f;
        asm("" : "+v"(one));
        float r0[4], r1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = pack_rtz(a[e].x, a[e].y);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi[e] & 0xffffu));
            const float h1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(hi[e] >> 16));
            r0[e] = __builtin_fmaf(a[e].x, one, -h0);
            r1[e] = __builtin_fmaf(a[e].y, one, -h1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) lo[e] = pack_rtz(r0[e], r1[e]);
    }
    // two streams at once (the cos and sin streams of a harmonic): twice as many independent operations per step
    __device__ __forceinline__ static void set2(AFrag &x, const f32x2 (&a)[4], AFrag &y, const f32x2 (&b)[4]) {
        if constexpr (ablate::kNoSplit) {
            x.set(a);
            y.set(b);
            return;
        }
        float one = 1.f;
        asm("" : "+v"(one));
        float ra0[4], ra1[4], rb0[4], rb1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x.hi[e] = pack_rtz(a[e].x, a[e].y); y.hi[e] = pack_rtz(b[e].x, b[e].y); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ra0[e] = __builtin_fmaf(a[e].x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(x.hi[e] & 0xffffu)));
            ra1[e] = __builtin_fmaf(a[e].y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(x.hi[e] >> 16)));
            rb0[e] = __builtin_fmaf(b[e].x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(y.hi[e] & 0xffffu)));
            rb1[e] = __builtin_fmaf(b[e].y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(y.hi[e] >> 16)));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { x.lo[e] = pack_rtz(ra0[e], ra1[e]); y.lo[e] = pack_rtz(rb0[e], rb1[e]); }
    }
};

// acc += LUT^T x stream for one (stream, tile): the LUT fragment is the MFMA's A operand (rows = packed
// columns), the stream its B operand (columns = patches), so a lane ends up holding packed columns
// 16t + 4(lane >> 4) + i of ITS OWN patch (lane & 15) -- which is what the fused epilogue needs.
// `part` selects one third of the f16 split so callers can interleave independent accumulators between the
// dependent MFMAs of one tile.
template <int POOL, int PART>
__device__ __forceinline__ void mma_part(const AFrag<POOL> &a, const BFrag &b, f32x4 &acc) {
    static_assert(POOL == LF_POOL_F32 || POOL == LF_POOL_F16X3, "the three-term / f32 forms");
    if constexpr (ablate::kNoMma && POOL == LF_POOL_F16X3) {   // (operands kept alive)
        asm volatile("" ::"v"(a.hi), "v"(a.lo), "v"(b.p0), "v"(b.p1));
        return;
    }
    if constexpr (POOL == LF_POOL_F32) {
        const f32x4 b0 = __builtin_bit_cast(f32x4, b.p0), b1 = __builtin_bit_cast(f32x4, b.p1);
        if (PART == 0) {
#pragma unroll
            for (int e = 0; e < 3; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[e], a.v[e], acc, 0, 0, 0);
        } else if (PART == 1) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[3], a.v[3], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[0], a.v[4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[1], a.v[5], acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[2], a.v[6], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[3], a.v[7], acc, 0, 0, 0);
        }
    } else {
        const f16x8 bh = __builtin_bit_cast(f16x8, b.p0), bl = __builtin_bit_cast(f16x8, b.p1);
        const f16x8 ah = __builtin_bit_cast(f16x8, a.hi), al = __builtin_bit_cast(f16x8, a.lo);
        if (PART == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc, 0, 0, 0);
        else if (PART == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc, 0, 0, 0);
    }
}

// Blurred row of this lane's segment from the raw-row ring (patch_gradients.glsl:72-92): vertical 5 taps over
// ring slots s0..s0+4, then horizontal 5 taps with the neighbours fetched from lanes -/+16.
// Returns the row plus its x-1 / x+8 neighbours.
// slot_of_tap(i): ring slot of the i-th of the five raw rows.
template <class SlotOfTap>
__device__ __forceinline__ void blur_row_impl(const unsigned char *ring_lane, SlotOfTap slot_of_tap, int addr_l, int addr_r,
                                              bool has_l, bool has_r, float (&out)[8], float &out_l, float &out_r) {
    if constexpr (ablate::kNoFrontEnd) {   // no blur: the first tap's raw row as it is
        const int sl = slot_of_tap(0);
        const f32x4 a_ = *reinterpret_cast<const f32x4 *>(ring_lane + sl * 2048);
        const f32x4 b_ = *reinterpret_cast<const f32x4 *>(ring_lane + sl * 2048 + 256);
#pragma unroll
        for (int x = 0; x < 4; ++x) { out[x] = a_[x]; out[4 + x] = b_[x]; }
        out_l = out[0];
        out_r = out[7];
        return;
    }
    float vb[8];
    {
        const float kk[5] = {kB0, kB1, kB2, kB1, kB0};
        f32x2 v2[4];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int sl = slot_of_tap(i);
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(ring_lane + sl * 2048);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(ring_lane + sl * 2048 + 256);
            const f32x2 r[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
#pragma unroll
            for (int e = 0; e < 4; ++e) v2[e] = i == 0 ? pk_set(kk[0]) * r[e] : pk_fma(pk_set(kk[i]), r[e], v2[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { vb[2 * e] = v2[e].x; vb[2 * e + 1] = v2[e].y; }
    }
    float ext[12];
    const float l0 = lane_fetch(addr_l, vb[6]), l1 = lane_fetch(addr_l, vb[7]);
    const float r0 = lane_fetch(addr_r, vb[0]), r1 = lane_fetch(addr_r, vb[1]);
    ext[0] = has_l ? l0 : vb[0];
    ext[1] = has_l ? l1 : vb[0];
    ext[10] = has_r ? r0 : vb[7];
    ext[11] = has_r ? r1 : vb[7];
#pragma unroll
    for (int x = 0; x < 8; ++x) ext[2 + x] = vb[x];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        // (the first tap as an fma onto +0.0: the same bits as the product for every non-zero result, and +0.0 where the
        //  factor is -0.0.  A product of a negative denormal that underflows still rounds to -0.0 -- the caller normalises
        //  gx itself, see "left - right" in the row loop)
        float s = ablate::kMulFirstTap ? kB0 * ext[x] : fmaf(kB0, ext[x], 0.0f);
        s = fmaf(kB1, ext[x + 1], s);
        s = fmaf(kB2, ext[x + 2], s);
        s = fmaf(kB1, ext[x + 3], s);
        s = fmaf(kB0, ext[x + 4], s);
        out[x] = s;
    }
    const float hl = lane_fetch(addr_l, out[7]), hr = lane_fetch(addr_r, out[0]);
    out_l = has_l ? hl : out[0];
    out_r = has_r ? hr : out[7];
}

__device__ __forceinline__ void blur_row(const unsigned char *ring_lane, int s0, int addr_l, int addr_r, bool has_l,
                                         bool has_r, float (&out)[8], float &out_l, float &out_r) {
    blur_row_impl(ring_lane, [s0](int i) { const int sl = s0 + i; return sl >= kRingSlots ? sl - kRingSlots : sl; }, addr_l,
                  addr_r, has_l, has_r, out, out_l, out_r);
}

// Harmonics k = 1..3 of the gradient angle by angle addition, each stream pair against its four LUT tiles.  The 21 matrix
// instructions of a harmonic (7 accumulators x 3 terms) are issued term by term across the accumulators, so dependent
// ones are seven apart; `g` holds the fragments of P0 on entry (prefetched by the caller) and P0 of the next harmonic on
// exit.
template <int POOL>
__device__ __forceinline__ void pool_harmonics(const f32x2 (&m)[4], const f32x2 (&c1)[4], const f32x2 (&s1)[4],
                                               const unsigned char *brow, BFrag &g, f32x4 (&acc)[kAccTiles]) {
    // The recurrence runs on the products themselves, (pk, qk) = m (cos, sin)(k theta), as a three-term (Chebyshev)
    // recurrence x_{k+1} = 2 c1 x_k - x_{k-1} with x_0 = (m, 0): one instruction per stream and harmonic instead of a
    // rotation of the unit vector (two) plus a product.
    f32x2 pk[4], qk[4], pp[4], qp[4], tc[4];
    AFrag<POOL> ac, as;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pk[e] = m[e] * c1[e]; qk[e] = m[e] * s1[e]; tc[e] = c1[e] + c1[e]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if constexpr (POOL == LF_POOL_F16X3) {
            AFrag<POOL>::set2(ac, pk, as, qk);
        } else {
            ac.set(pk);
            as.set(qk);
        }
        if (k < 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 pn = pk_fma(tc[e], pk[e], k == 0 ? -m[e] : -pp[e]);
                const f32x2 qn = k == 0 ? tc[e] * qk[e] : pk_fma(tc[e], qk[e], -qp[e]);
                pp[e] = pk[e]; qp[e] = qk[e];
                pk[e] = pn; qk[e] = qn;
            }
        }
        const int a0 = 3 + 7 * k, u0 = 3 + 4 * k;
        const BFrag p0 = g, q0 = load_b(brow, u0 + 1), r = load_b(brow, u0 + 2), sf = load_b(brow, u0 + 3);
        if (k < 2) g = load_b(brow, u0 + 4);
        // cos x P0 -> a0 | sin x P0 -> a0+2 | sin x Q0 -> a0+1 | cos x R, sin x R -> a0+3, a0+4 | cos x S, sin x S -> a0+5, a0+6
        mma_part<POOL, 0>(ac, p0, acc[a0 + 0]); mma_part<POOL, 0>(as, p0, acc[a0 + 2]); mma_part<POOL, 0>(as, q0, acc[a0 + 1]);
        mma_part<POOL, 0>(ac, r, acc[a0 + 3]); mma_part<POOL, 0>(as, r, acc[a0 + 4]);
        mma_part<POOL, 0>(ac, sf, acc[a0 + 5]); mma_part<POOL, 0>(as, sf, acc[a0 + 6]);
        mma_part<POOL, 1>(ac, p0, acc[a0 + 0]); mma_part<POOL, 1>(as, p0, acc[a0 + 2]); mma_part<POOL, 1>(as, q0, acc[a0 + 1]);
        mma_part<POOL, 1>(ac, r, acc[a0 + 3]); mma_part<POOL, 1>(as, r, acc[a0 + 4]);
        mma_part<POOL, 1>(ac, sf, acc[a0 + 5]); mma_part<POOL, 1>(as, sf, acc[a0 + 6]);
        mma_part<POOL, 2>(ac, p0, acc[a0 + 0]); mma_part<POOL, 2>(as, p0, acc[a0 + 2]); mma_part<POOL, 2>(as, q0, acc[a0 + 1]);
        mma_part<POOL, 2>(ac, r, acc[a0 + 3]); mma_part<POOL, 2>(as, r, acc[a0 + 4]);
        mma_part<POOL, 2>(ac, sf, acc[a0 + 5]); mma_part<POOL, 2>(as, sf, acc[a0 + 6]);
        // cos x Q0: the second product of relsin[0:16], into the accumulator of the first
        mma_part<POOL, 0>(ac, q0, acc[a0 + 2]); mma_part<POOL, 1>(ac, q0, acc[a0 + 2]); mma_part<POOL, 2>(ac, q0, acc[a0 + 2]);
    }
}

// LF_POOL_F16_FP6 (an experiment kept as a mode; NOTEBOOK.md section 11): the harmonics with hi x hi in f16 (8 instructions per
// harmonic) and BOTH cross terms of every accumulator tile in one v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands (7):
// a lane's 32 K slots hold (a_e, 2048 r_e) of its 8 pixels for the cos stream (fields 0-15 = registers 0-2) and for the
// sin stream (16-31 = registers 3-5), a = the stream value, r = a - f16(a), all divided by the lane's block scale S (from
// the largest m of its 8 pixels: |a| <= m); the LUT operand holds (2048 lo, hi) / T the same way (mkd_consts.hpp).  The
// two streams' values go through ONE v_cvt_scalef32_2xpk16_fp6_f32, which interleaves its two 16-value sources field by
// field (tools/micro/fp6_cross.hip).  A single-stream product reads the stream's three registers beside three zeros (the
// LUT registers opposite them may hold anything: e2m3 has no NaN or infinity); the merged tile reads all six.
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void pool_harmonics_fp6(const f32x2 (&m)[4], const f32x2 (&c1)[4], const f32x2 (&s1)[4],
                                                   const unsigned char *brow, BFrag &g, f32x4 (&acc)[kAccTiles]) {
    // the harmonic's two streams live in the conversion's first source, aq = (pk[0..3], qk[0..3]): the recurrence writes the
    // next harmonic into a fresh one (the k loop is unrolled: renaming, no copies)
    v16f aq, ap;
    f32x2 tc[4];
    auto pair = [](const v16f &v, int i) { return f32x2{v[2 * i], v[2 * i + 1]}; };
    auto put = [](v16f &v, int i, f32x2 x) { v[2 * i] = x.x; v[2 * i + 1] = x.y; };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        put(aq, e, m[e] * c1[e]);
        put(aq, 4 + e, m[e] * s1[e]);
        tc[e] = c1[e] + c1[e];
    }
    // block scale S = 2^(exponent of the largest m) / 2: every |a| / S < 4 and every 2048 r / S < 4 (r < one f16 ulp of a)
    float mx = fmaxf(fmaxf(m[0].x, m[0].y), fmaxf(m[1].x, m[1].y));
    mx = fmaxf(mx, fmaxf(fmaxf(m[2].x, m[2].y), fmaxf(m[3].x, m[3].y)));
    const unsigned s_bits = (__float_as_uint(mx) & 0x7F800000u) - 0x00800000u;
    const float s_scale = __uint_as_float(s_bits);
    const int scale_b = (int)(s_bits >> 23) - 11;        // E8M0 of S / 2048
    float one = 1.f;
    asm("" : "+v"(one));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        u32x4 hc, hs;
        v16f r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 p = pair(aq, e), q = pair(aq, 4 + e);
            hc[e] = pack_rtz(p.x, p.y);
            hs[e] = pack_rtz(q.x, q.y);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 p = pair(aq, e), q = pair(aq, 4 + e);
            const f32x2 rc = {__builtin_fmaf(p.x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(hc[e] & 0xffffu))),
                              __builtin_fmaf(p.y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(hc[e] >> 16)))};
            const f32x2 rs = {__builtin_fmaf(q.x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(hs[e] & 0xffffu))),
                              __builtin_fmaf(q.y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(hs[e] >> 16)))};
            put(r16, e, rc * pk_set(2048.f));
            put(r16, 4 + e, rs * pk_set(2048.f));
        }
        // (written as inline assembly for its early-clobber destination: through the builtin, hipcc of ROCm 7.2 lets the six
        //  destination registers overlap a source that dies here -- v[36:41] <- v[16:31], v[32:47] -- and the instruction
        //  writes them while it still reads its thirty-two inputs: harmonics 2 and 3 came out with garbage cross terms)
        v6u d;
        asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(aq), "v"(r16), "v"(s_scale));
        if (k < 2) {   // x_{k+1} = 2 c1 x_k - x_{k-1}, written over x_{k-1}; then the two change roles
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 p = pair(aq, e), q = pair(aq, 4 + e);
                put(ap, e, pk_fma(tc[e], p, k == 0 ? -m[e] : -pair(ap, e)));
                put(ap, 4 + e, k == 0 ? tc[e] * q : pk_fma(tc[e], q, -pair(ap, 4 + e)));
            }
            const v16f t = aq;
            aq = ap;
            ap = t;
        }
        const int a0 = 3 + 7 * k, u0 = 3 + 4 * k;
        const BFrag p0 = g, q0 = load_b(brow, u0 + 1), r = load_b(brow, u0 + 2), sf = load_b(brow, u0 + 3);
        if (k < 2) g = load_b(brow, u0 + 4);
        const f16x8 ch = __builtin_bit_cast(f16x8, hc), sh = __builtin_bit_cast(f16x8, hs);
        auto hh = [&](const BFrag &b, const f16x8 &st, f32x4 &c) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b.p0), st, c, 0, 0, 0);
        };
        // hi x hi: cos x P0 -> a0 | sin x P0, cos x Q0 -> a0+2 | sin x Q0 -> a0+1 | cos, sin x R -> a0+3, a0+4 | x S -> a0+5, a0+6
        hh(p0, ch, acc[a0 + 0]); hh(p0, sh, acc[a0 + 2]); hh(q0, sh, acc[a0 + 1]); hh(r, ch, acc[a0 + 3]);
        hh(r, sh, acc[a0 + 4]); hh(sf, ch, acc[a0 + 5]); hh(sf, sh, acc[a0 + 6]); hh(q0, ch, acc[a0 + 2]);
        // the cross terms: one instruction per accumulator; the LUT fragment's fourth word is its block scale
        v8i bc, bs, bm;      // (registers 6 and 7 of an fp6 operand are not read)
        bc[0] = (int)d[0]; bc[1] = (int)d[1]; bc[2] = (int)d[2]; bc[3] = 0; bc[4] = 0; bc[5] = 0;
        bs[0] = (int)d[3]; bs[1] = (int)d[4]; bs[2] = (int)d[5]; bs[3] = 0; bs[4] = 0; bs[5] = 0;
        bm[0] = (int)d[0]; bm[1] = (int)d[1]; bm[2] = (int)d[2]; bm[3] = (int)d[3]; bm[4] = (int)d[4]; bm[5] = (int)d[5];
        auto single = [&](const u32x4 &f, const v8i &b, f32x4 &c) {
            v8i a;   // the fragment's own four registers; what follows them meets the zeros of b
            a[0] = (int)f[0]; a[1] = (int)f[1]; a[2] = (int)f[2]; a[3] = (int)f[3];
            c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, (int)f[3], 0, scale_b);
        };
        single(p0.p1, bc, acc[a0 + 0]);
        single(q0.p1, bs, acc[a0 + 1]);
        {   // cos x Q0 + sin x P0 (P0 and Q0 carry the same scale)
            v8i a;
            a[0] = (int)q0.p1[0]; a[1] = (int)q0.p1[1]; a[2] = (int)q0.p1[2];
            a[3] = (int)p0.p1[0]; a[4] = (int)p0.p1[1]; a[5] = (int)p0.p1[2];
            acc[a0 + 2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, bm, acc[a0 + 2], 2, 2, 0, (int)p0.p1[3], 0, scale_b);
        }
        single(r.p1, bc, acc[a0 + 3]);
        single(r.p1, bs, acc[a0 + 4]);
        single(sf.p1, bc, acc[a0 + 5]);
        single(sf.p1, bs, acc[a0 + 6]);
    }
}

// The 24 accumulator tiles of the row loop -> the 21 tiles of packed output columns (mkd_consts.hpp, packed_desc):
//   relcos_k = cos x EPc - sin x EPs, relsin_k = sin x EPc + cos x EPs (columns 0-15 in P0 / Q0, 16-24 in slots 0-8 of R / S),
//   abscos_k / abssin_k = the EC columns: 0-6 in slots 9-15 of R, 7-8 in slots 9-10 of S.
// Slot of (lane, i) in a tile: 4 (lane >> 4) + i.  Unused slots are set to zero (the norms run over whole tiles).
__device__ __forceinline__ void combine_tiles(const f32x4 (&a)[kAccTiles], int q, f32x4 (&o)[kTiles]) {
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const int b = 3 + 7 * h, l = 3 + 6 * h;
        o[l + 0] = a[b + 0] - a[b + 1];
        o[l + 1] = a[b + 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = 4 * q + i;
            const bool lo9 = slot < 9, s910 = slot == 9 || slot == 10;
            o[l + 2][i] = a[b + 3][i] - (lo9 ? a[b + 6][i] : 0.f);   // cos x R - sin x S
            o[l + 3][i] = a[b + 4][i] + (lo9 ? a[b + 5][i] : 0.f);   // sin x R + cos x S
            o[l + 4][i] = s910 ? a[b + 5][i] : 0.f;
            o[l + 5][i] = s910 ? a[b + 6][i] : 0.f;
        }
    }
}

// which block of the raw descriptor a packed tile's slots belong to (normalize.glsl: polar | cartesian):
// 0 = all polar, 1 = all cartesian (or unused = 0), 2 = slots 0-8 polar, 9-15 cartesian
__device__ __forceinline__ constexpr int tile_block(int t) {
    if (t < 3) return t == 0 ? 0 : (t == 1 ? 2 : 1);
    const int part = (t - 3) % 6;
    return part < 2 ? 0 : (part < 4 ? 2 : 1);
}

// Epilogue, per wave and batch: acc[t][i] holds the pooled sum of packed column 16t + 4q + i for the lane's
// own patch.  normalize.glsl:22-142 (polar | cartesian | all), whitening.glsl:22-77 as an MFMA with the
// accumulators as B operand (out^T = W_T x raw, the mean folded into a bias), normalize_final.glsl.
// Whitening fragments (host: mkd_consts.cpp): f16: [step 11][row tile 8][hi|lo][lane][8], step s covers
// accumulator tiles 2s, 2s+1; f32: [tile 21][i 4][row tile 8][lane].
// f16 path: on entry step 0 of the fragments is on its way into its slot (requested by the caller during patch row 31);
// on exit LUT row 0 of the next batch is on its way into row buffer 0 if `more`.  Every wave of the workgroup must call.
template <int POOL, int W>
__device__ __forceinline__ void finish_descriptors(const f32x4 (&acc_row)[kAccTiles], int lane, int wave, bool valid, long patch,
                                                   const short *__restrict__ colmap,
                                                   const unsigned char *__restrict__ wfrag,
                                                   const float *__restrict__ bias, float *__restrict__ out,
                                                   float *__restrict__ raw_out, unsigned char *s_mem,
                                                   const unsigned char *__restrict__ lut_rows, bool more) {
    const int q = lane >> 4;
    f32x4 acc[kTiles];
    combine_tiles(acc_row, q, acc);
    if constexpr (f16_family(POOL)) {
        __syncthreads();   // every wave has left patch row 31: row buffer 1 is free too
        issue_w_step<W>(wfrag, 1, s_mem, wave, lane);
        issue_w_step<W>(wfrag, 2, s_mem, wave, lane);
    }
    // mixed tiles hold polar kernels in packed columns 0-8 and cartesian ones in 9-15
    bool lo9[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) lo9[i] = 4 * q + i < 9;
    float sp = 0.f, sc = 0.f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (tile_block(t) == 2) {
                const float v2 = acc[t][i] * acc[t][i];
                sp += lo9[i] ? v2 : 0.f;
                sc += lo9[i] ? 0.f : v2;
            } else if (tile_block(t) == 0) {
                sp = fmaf(acc[t][i], acc[t][i], sp);
            } else {
                sc = fmaf(acc[t][i], acc[t][i], sc);
            }
        }
    sp += __shfl_xor(sp, 16); sp += __shfl_xor(sp, 32);
    sc += __shfl_xor(sc, 16); sc += __shfl_xor(sc, 32);
    const float inv_p = 1.f / __builtin_amdgcn_sqrtf(sp), inv_c = 1.f / __builtin_amdgcn_sqrtf(sc);
    // normalize.glsl then L2-normalises the concatenation of the two unit blocks.  Its squared norm follows from the
    // block sums (sp inv_p^2 + sc inv_c^2, within 2 ulp of the shader's second pass over the 238 values), so the two
    // scalings fold into one pass
    const float sa = fmaf(sp * inv_p, inv_p, (sc * inv_c) * inv_c);
    const float inv_a = 1.f / __builtin_amdgcn_sqrtf(sa);
    const float k_p = inv_p * inv_a, k_c = inv_c * inv_a;
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool polar = tile_block(t) == 0 || (tile_block(t) == 2 && lo9[i]);
            acc[t][i] *= polar ? k_p : k_c;
        }
    if (raw_out) {  // verification tap: the 238-D descriptor before whitening
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = colmap[t * 16 + 4 * q + i];
                if (valid && d >= 0) raw_out[patch * 238 + d] = acc[t][i];
            }
    }
    f32x4 o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (f16_family(POOL)) {
        // The whitening is bound by operand delivery (each wave needs all 176 KiB of fragments per batch): from LDS, where
        // the 8 waves share one copy, they arrive at twice the rate the vector L1 gives each wave its own.
        constexpr int kDma = 16 / W;   // DMA instructions per lane and step
        if (raw_out) wait_vmcnt<0>();  // stores and loads retire out of order with respect to each other
        // Within a wave the LDS reads run one unit (4 row tiles, hi + lo = 8 KiB) ahead of the MFMAs, in two register
        // buffers: the waves of a workgroup move in lock step here, so without that the LDS and the matrix pipe would
        // take turns idling.
        u32x4 wbuf[2][8];   // [buffer][row tile rr][hi|lo]
        auto read_unit = [&](int u, u32x4 (&dst)[8]) {
            const u32x4 *w = reinterpret_cast<const u32x4 *>(s_mem + wstage_slot(u >> 1)) + (u & 1) * 8 * 64 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = w[j * 64];
        };
        f16x8 yh, yl;
        auto mma_unit = [&](int u, const u32x4 (&src)[8]) {
            const int r0 = (u & 1) * 4;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                o[r0 + rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, src[2 * rr + 1]), yh, o[r0 + rr], 0, 0, 0);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                o[r0 + rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, src[2 * rr]), yl, o[r0 + rr], 0, 0, 0);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                o[r0 + rr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, src[2 * rr]), yh, o[r0 + rr], 0, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < 11; ++s) {
            // own share of step s has landed (steps s+1, and s+2 for s = 0, may still be in flight) and own reads of
            // step s-1 have returned; after the barrier that holds for every wave, so the slot of step s-1 can take
            // step s+2
            if (s == 0) wait_vmcnt<2 * kDma>();
            else if (s < 10) wait_vmcnt<kDma>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (s >= 1 && s + 2 < 11) issue_w_step<W>(wfrag, s + 2, s_mem, wave, lane);
            if (s == 10 && more) issue_lut_row<W>(lut_rows, 0, s_mem, wave, lane);
            read_unit(2 * s, wbuf[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (s > 0) mma_unit(2 * s - 1, wbuf[1]);
            {
                const f32x4 y0 = acc[2 * s];
                const f32x4 y1 = 2 * s + 1 < kTiles ? acc[(2 * s + 1 < kTiles) ? 2 * s + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x2 y[4] = {{y0[0], y0[1]}, {y0[2], y0[3]}, {y1[0], y1[1]}, {y1[2], y1[3]}};
                AFrag<LF_POOL_F16X3> yb;
                yb.set(y);
                yh = __builtin_bit_cast(f16x8, yb.hi);
                yl = __builtin_bit_cast(f16x8, yb.lo);
            }
            __builtin_amdgcn_sched_barrier(0);
            read_unit(2 * s + 1, wbuf[1]);
            __builtin_amdgcn_sched_barrier(0);
            mma_unit(2 * s, wbuf[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        mma_unit(21, wbuf[1]);
    } else {
        const float *w = reinterpret_cast<const float *>(wfrag);
        float wbuf[2][16];  // unit = two (tile, i) steps x 8 row tiles
        auto load_unit = [&](int u, float (&dst)[16]) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 8; ++r) dst[k * 8 + r] = w[((2 * u + k) * 8 + r) * 64 + lane];
        };
        load_unit(0, wbuf[0]);
#pragma unroll
        for (int u = 0; u < 42; ++u) {
            if (u + 1 < 42) load_unit(u + 1, wbuf[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ti = 2 * u + k;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    o[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wbuf[u & 1][k * 8 + r], acc[ti >> 2][ti & 3], o[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // lane holds outputs 16r + 4q + i of its patch: add the bias (-W mean), L2-normalise, store
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + 16 * r + 4 * q);
        o[r] += b;
#pragma unroll
        for (int i = 0; i < 4; ++i) ss = fmaf(o[r][i], o[r][i], ss);
    }
    ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
    const float inv = 1.f / __builtin_amdgcn_sqrtf(ss);   // one division, 32 multiplications: 1 ulp from 32 divisions
    if (valid) {
#pragma unroll
        for (int r = 0; r < 8; ++r) *reinterpret_cast<f32x4 *>(out + patch * 128 + 16 * r + 4 * q) = o[r] * inv;
    }
}

// ---- keypoint mode (SRC = kSrcKeypoints) --------------------------------------------------------------------------
// What the describe kernel is given instead of patches: the pyramids (mkd_pyramid.hip) and the keypoint list.
struct KpSource {
    const float *pyr;            // pyramids of the frames, pyr_stride floats apart
    long pyr_stride;
    const float *kps;            // [n][5] x, y, size, angle (degrees), response
    const unsigned *frame_of;    // [n] frame of each keypoint, or null: all on frame 0
    unsigned n_frames;           // frames in the store: larger indices (caller's data) are clamped
    float psf;                   // patch_scale_factor
    PyramidDesc pd;
    // row-split form only: workgroups per batch (2 or 4), the partial sums' exchange buffer
    // [batch][role R-1][wave W][tile 24][half 2][lane 64] 2 x f32, its arrival counters [batch][wave] and an error word
    int split;
    float *xchg;
    unsigned *xchg_cnt;
};

// Batches of a workgroup.  Keypoints arrive ordered by frame, and every XCD has its own L2: workgroups that share an XCD
// (b, b + 8, ...: dispatch is round-robin over the 8 XCDs -- a matter of speed only, nothing depends on it) walk ONE
// contiguous eighth of the batches, so that a frame's pyramid is fetched into one L2 instead of eight.
struct BatchWalk { long cur, end, step; };
__device__ __forceinline__ BatchWalk kp_batches(long nbatch) {
    const long g = gridDim.x, b = blockIdx.x;
    if ((g & 7) == 0 && nbatch > g) {
        const long chunk = (nbatch + 7) / 8, lo = (b & 7) * chunk, hi = lo + chunk < nbatch ? lo + chunk : nbatch;
        return BatchWalk{lo + (b >> 3), hi, g >> 3};
    }
    return BatchWalk{b, nbatch, g};
}

__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// A producer wave: samples the patches of ITS describe wave (wave pw of the workgroup's W describe waves) into that wave's
// row ring, patch_gradients.glsl:42-70 with the arithmetic of mkd_sample.h.
// Work unit = a quarter of a row group: rows 4G..4G+3 of patches 4q..4q+3, as eight blocks of 4 rows x 16 pixels -- one
// block per load instruction, lane = (row r = lane >> 4, column c = lane & 15): a compact footprint of a few cache lines
// (the texture path's cost is per line touched per instruction: a whole patch row per instruction, which the ring's row
// order would suggest, costs 7x as much -- tools/micro/gather_patterns.hip).  One quarter per row step of the describe
// waves: quarter Q = g + 7 is written during step g, i.e. a group is complete one barrier before its first row is read and
// is written into the slots of rows read for the last time one barrier earlier (ring of 12, see kRingSlotsKp).  Quarters
// 32..38 are the first seven of the workgroup's NEXT batch; the epilogue of the describe waves is a pause.
// The taps of a quarter are requested one step BEFORE it is written (two register sets, the step loop unrolled by two):
// their latency passes beside the previous quarter's arithmetic instead of in front of the barrier.
// Every barrier of the describe waves has its twin here: 32 row steps + 12 in finish_descriptors per batch.
// The producers also request the LUT rows of the row loop: an LDS-DMA request stalls its issuer for 60-180 cycles, which a
// producer can afford, so a describe wave issues no memory instruction between its barriers.
struct KpTaps { f32x2 top[8], bot[8]; float ax[8], ay[8]; };

// min(max(x, 0), hi) for hi >= 0 as ONE instruction (the median of x, 0, hi); hipcc cannot fuse the pair itself because it
// does not know hi >= 0
__device__ __forceinline__ int clamp_med3(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

// (uniform base) + (unsigned 32-bit lane offset) -> global_load_dwordx2 v, v_off, s[base:base+1]: no 64-bit address arithmetic
// (the two empty asm statements: see lds_dma16_sv).  Texel addresses are 4-byte aligned only.
__device__ __forceinline__ f32x2 load2_sv(const unsigned char *uniform_base, unsigned lane_off) {
    asm("" : "+s"(uniform_base));
    asm volatile("" : "+v"(lane_off));
    typedef f32x2 f32x2_a4 __attribute__((aligned(4)));
    return *reinterpret_cast<const __attribute__((address_space(1))) f32x2_a4 *>(
        (const __attribute__((address_space(1))) unsigned char *)uniform_base + lane_off);
}

// The sampling machinery of one row ring: the geometry of the ring's keypoints and the two halves of a quarter's work.
template <int W>
struct KpSampler {
    const KpSource &ks;
    const LevelTable &lt;
    const long *lvl_offset;
    long n;
    unsigned char *ring;
    int pw, lane, r, c;
    // geometry of a batch's 16 keypoints, one per lane: lanes 0-15 hold set 0, lanes 16-31 set 1 (this batch / the next).
    // g_xmax / g_ymax: largest first-tap index in the level's allocation (apron included); the row pitch is g_xmax + 2
    // texels and the apron kPyrApron on every level, so neither needs a register of its own.
    float g_ca = 0.f, g_sa = 0.f, g_rem = 0.f, g_cx = 0.f, g_cy = 0.f;
    int g_xmax = 0, g_ymax = 0, g_cov = 0;
    unsigned g_lo = 0, g_hi = 0;

    __device__ __forceinline__ KpSampler(const KpSource &ks_, const LevelTable &lt_, const long *lvl_offset_, long n_,
                                         unsigned char *s_mem, int pw_, int lane_)
        : ks(ks_), lt(lt_), lvl_offset(lvl_offset_), n(n_), ring(s_mem + kRingOff + pw_ * (kRingSlotsKp * 2048)), pw(pw_),
          lane(lane_), r(lane_ >> 4), c(lane_ & 15) {}

    __device__ __forceinline__ void load_geometry(long batch, int set) {
        long k = batch * (16 * W) + pw * 16 + (lane & 15);
        k = k < n ? k : n - 1;
        const float *kp = ks.kps + k * 5;
        const KpGeom g = keypoint_geometry(kp[0], kp[1], kp[2], kp[3], ks.psf, lt);
        // first texel of the level's allocation (apron included): offsets from it are never negative
        const int a = lt.apron[g.level], pitch = lt.pitch[g.level];
        const float *a0 = ks.pyr + (ks.frame_of ? (long)min(ks.frame_of[k], ks.n_frames - 1u) * ks.pyr_stride : 0L) +
                          lvl_offset[g.level] -
                          (long)a * pitch - a;
        if ((lane >> 4) == set) {
            g_ca = g.ca; g_sa = g.sa; g_rem = g.rem; g_cx = g.cx; g_cy = g.cy;
            g_xmax = lt.w[g.level] + 2 * a - 2; g_ymax = lt.h[g.level] + 2 * a - 2;   // (pitch = w + 2a, a = kPyrApron)
            g_cov = g.covered ? 1 : 0;
            g_lo = (unsigned)(uintptr_t)a0; g_hi = (unsigned)((uintptr_t)a0 >> 32);
        }
    }
    __device__ __forceinline__ const unsigned char *base_of(int src) const {
        return reinterpret_cast<const unsigned char *>((uintptr_t)(unsigned)readlane_i((int)g_lo, src) |
                                                       ((uintptr_t)(unsigned)readlane_i((int)g_hi, src) << 32));
    }
    // requests the taps of a quarter (fetch_covered of mkd_sample.h: indices clamped into the level's allocation)
    __device__ __forceinline__ void request(int quarter, int set, KpTaps &t) const {
        if constexpr (ablate::kNoProducer) return;
        const int q = quarter & 3, ly = 4 * (quarter >> 2) + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = 16 * set + 4 * q + i;   // uniform
            const float ca = readlane_f(g_ca, src), sa = readlane_f(g_sa, src), rem = readlane_f(g_rem, src);
            const float cx = readlane_f(g_cx, src), cy = readlane_f(g_cy, src);
            const int xmax = readlane_i(g_xmax, src), ymax = readlane_i(g_ymax, src), pitch4 = 4 * (xmax + 2);
            constexpr int apron = kPyrApron;
            const unsigned char *a0 = base_of(src), *a1 = a0 + pitch4;   // the two tap rows: two uniform bases, one lane offset
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const SamplePos p = sample_position(ca, sa, rem, cx, cy, 16 * hh + c, ly);
                t.ax[2 * i + hh] = p.ax;
                t.ay[2 * i + hh] = p.ay;
                // the apron goes onto both coordinates as ONE two-wide float addition (exact: integers far below 2^23)
                typedef float v2 __attribute__((ext_vector_type(2)));
                const v2 fa = v2{p.fx0, p.fy0} + v2{(float)apron, (float)apron};
                const int ix = clamp_med3((int)fa.x, xmax), iy = clamp_med3((int)fa.y, ymax);
                const unsigned off = __umul24((unsigned)iy, (unsigned)pitch4) + 4u * (unsigned)ix;   // both < 2^24: full-rate multiply
                if constexpr (ablate::kNoTaps) {
                    t.top[2 * i + hh] = f32x2{__uint_as_float(off), p.ax};
                    t.bot[2 * i + hh] = f32x2{p.ay, (float)pitch4};
                    continue;
                }
                t.top[2 * i + hh] = load2_sv(a0, off);
                t.bot[2 * i + hh] = load2_sv(a1, off);
            }
        }
    }
    // blends them and writes the quarter's rows into the ring
    __device__ __forceinline__ void finish(int quarter, int set, int slot0, const KpTaps &t) const {
        if constexpr (ablate::kNoProducer) return;
        const int G = quarter >> 2, q = quarter & 3;
        // (slot0 + 4G) mod 12 is a multiple of 4: the group's four rows sit in consecutive slots.  Ring slot layout:
        // [half 2][16-byte chunk of the half-row 4][patch 16][4 pixels]; this lane's pixel c (+16) of patch 4q + i
        unsigned char *dst = ring + ((slot0 + 4 * G) % kRingSlotsKp + r) * 2048 + (c & 3) * 4 + (c >> 2) * 256 + q * 64;
        int any_uncovered = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            any_uncovered |= 1 - readlane_i(g_cov, 16 * set + 4 * q + i);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int k = 2 * i + hh;
                *reinterpret_cast<float *>(dst + hh * 1024 + i * 16) =
                    bilinear_blend(t.top[k].x, t.top[k].y, t.bot[k].x, t.bot[k].y, t.ax[k], t.ay[k]);
            }
        }
        // keypoints whose footprint leaves the apron (centre outside the frame, sizes beyond the pyramid, non-finite data):
        // rare -- their samples are redone with MirroredRepeat evaluated per tap
        if (any_uncovered) {
            const int ly = 4 * G + r;
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int src = 16 * set + 4 * q + i;
                if (readlane_i(g_cov, src)) continue;
                constexpr int apron = kPyrApron;
                const int pitch = readlane_i(g_xmax, src) + 2;
                const int w = readlane_i(g_xmax, src) + 2 - 2 * apron, h = readlane_i(g_ymax, src) + 2 - 2 * apron;
                const float *lvl0 = reinterpret_cast<const float *>(base_of(src)) + (long)apron * pitch + apron;
#pragma unroll 1
                for (int hh = 0; hh < 2; ++hh) {
                    const SamplePos p = sample_position(readlane_f(g_ca, src), readlane_f(g_sa, src), readlane_f(g_rem, src),
                                                        readlane_f(g_cx, src), readlane_f(g_cy, src), 16 * hh + c, ly);
                    const Taps tm = fetch_mirrored(lvl0, w, h, pitch, p);
                    *reinterpret_cast<float *>(dst + hh * 1024 + i * 16) =
                        bilinear_blend(tm.t00, tm.t10, tm.t01, tm.t11, p.ax, p.ay);
                }
            }
        }
    }
};

template <int W>
__device__ __forceinline__ void kp_produce(const KpSource &ks, const LevelTable &lt, const long *lvl_offset, long n,
                                           BatchWalk walk, unsigned char *s_mem, const unsigned char *__restrict__ lut_rows,
                                           int pw, int lane) {
    // The producer is the shorter instruction stream of the two waves on its SIMD, but its step ends in the barrier the
    // describe waves of all four SIMDs wait at: served first, it keeps out of their way (same-box A/B: +4 %)
    __builtin_amdgcn_s_setprio(ablate::kProducerPrio);
    KpSampler<W> sm(ks, lt, lvl_offset, n, s_mem, pw, lane);
    KpTaps ta, tb;
    int set = 0, slot0 = 0;
    // the first seven quarters of the first batch, before the describe waves' first row barrier.  (Two tap sets in flight
    // here, as in kp_produce_split, was measured in round 6: 76.5 / 79.2 / 93.0 us against 76.9 / 78.7 / 93.6 at 6000 / 8192 /
    // 10 000 keypoints -- nothing.  Letting the describe
    // waves, idle until then, sample every other one of them was tried: nothing to gain -- same-box, a one-round launch of
    // 10 000 keypoints took 96 us either way.)
    sm.load_geometry(walk.cur, 0);
#pragma unroll 1
    for (int quarter = 0; quarter < 7; ++quarter) {
        sm.request(quarter, 0, ta);
        sm.finish(quarter, 0, 0, ta);
    }
    sm.request(7, 0, ta);
#pragma unroll 1
    for (long batch = walk.cur; batch < walk.end; batch += walk.step) {
        const bool more = batch + walk.step < walk.end;
        if (more) sm.load_geometry(batch + walk.step, set ^ 1);
        const int slot1 = (slot0 + 8) % kRingSlotsKp;
        // step g: LUT row g + 1 is requested into the row buffer the describe waves have just left (row g uses buffer g & 1),
        // then the taps of the quarter step g + 1 will write (vector-memory results return in issue order: the taps, which
        // have a whole step, queue behind the LUT pieces, not the other way round); then the quarter of THIS step, whose
        // taps landed during the previous one, is blended and written
        auto step = [&](int g, KpTaps &cur, KpTaps &nxt) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // ring writes, LUT pieces and taps have landed
            __syncthreads();
            if (g < 31) issue_lut_row<W>(lut_rows, g + 1, s_mem + ((g + 1) & 1) * kRowBytes, pw, lane);
            const int qn = g + 8;    // the quarter of step g + 1
            if (qn < 32) sm.request(qn, set, nxt);
            else if (more) sm.request(qn - 32, set ^ 1, nxt);   // (g = 31: quarter 7 of the next batch)
            __builtin_amdgcn_sched_barrier(0);
            const int qc = g + 7;
            if (qc < 32) sm.finish(qc, set, slot0, cur);
            else if (more) sm.finish(qc - 32, set ^ 1, slot1, cur);
        };
#pragma unroll 1
        for (int g = 0; g < ablate::kRows; g += 2) {
            step(g, ta, tb);
            step(g + 1, tb, ta);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int e = 0; e < 12; ++e) __syncthreads();   // finish_descriptors: one on entry, one per whitening step
        set ^= 1;
        slot0 = slot1;
    }
}

// The row-split form's share of a batch: patch rows [lo, hi) of its 32 keypoints.  The describe waves need blurred rows
// lo-1 .. hi, i.e. raw rows lo-3 .. hi+2 (clamped to the patch): the row groups of four that hold them are the quarters
// [q_first, q_end) of the producers' numbering (quarter = 4 x group + patch quad).  n_pro of them are sampled before the
// first row barrier: the describe waves' first step reads raw rows up to lo+3 and the second up to lo+4, so from the group
// of row lo+4 on a group has exactly the four steps before its first use -- everything before it, and all but one quarter
// of that group when lo > 0 (its first use comes one step after the start), belongs to the prologue: 7 quarters for the
// rows from 0 (as in the whole-patch form), 11 otherwise.  Ring slot of raw row r: r mod 12, as in the whole-patch form.
struct RowSpan {
    int lo, hi, q_first, q_end, n_pro;
    bool consumer;      // the workgroup of the LAST rows: it collects the others' partial sums and finishes the descriptors
};
__device__ __forceinline__ RowSpan row_span(int role, int n_roles) {
    RowSpan sp;
    sp.lo = 32 * role / n_roles;
    sp.hi = 32 * (role + 1) / n_roles;
    const int r_first = sp.lo - 3 > 0 ? sp.lo - 3 : 0, r_last = sp.hi + 2 < 31 ? sp.hi + 2 : 31;
    sp.q_first = 4 * (r_first >> 2);
    sp.q_end = 4 * ((r_last >> 2) + 1);
    sp.n_pro = sp.lo == 0 ? 7 : 11;
    sp.consumer = role == n_roles - 1;
    return sp;
}

// The producer wave of a row-split workgroup: kp_produce for rows [lo, hi) of ONE batch.  One quarter per row step of the
// describe waves, requested a step ahead like there; every barrier of the describe waves has its twin here: hi - lo row steps,
// and the 12 of finish_descriptors in the consumer workgroup.
template <int W>
__device__ __forceinline__ void kp_produce_split(const KpSource &ks, const LevelTable &lt, const long *lvl_offset, long n,
                                                 long batch, const RowSpan &sp, unsigned char *s_mem,
                                                 const unsigned char *__restrict__ lut_rows, int pw, int lane) {
    __builtin_amdgcn_s_setprio(ablate::kProducerPrio);
    KpSampler<W> sm(ks, lt, lvl_offset, n, s_mem, pw, lane);
    KpTaps ta, tb;
    sm.load_geometry(batch, 0);
    // the prologue, two tap sets in flight: the taps of quarter i + 1 are on their way while quarter i is blended
    sm.request(sp.q_first, 0, ta);
#pragma unroll 1
    for (int i = 0; i + 1 < sp.n_pro; i += 2) {
        sm.request(sp.q_first + i + 1, 0, tb);
        sm.finish(sp.q_first + i, 0, 0, ta);
        sm.request(sp.q_first + i + 2, 0, ta);     // (n_pro is odd: i + 2 <= n_pro - 1 is a prologue quarter)
        sm.finish(sp.q_first + i + 1, 0, 0, tb);
    }
    sm.finish(sp.q_first + sp.n_pro - 1, 0, 0, ta);
    const int q_step0 = sp.q_first + sp.n_pro;      // the quarter row step 0 writes
    if (q_step0 < sp.q_end) sm.request(q_step0, 0, ta);
    auto step = [&](int st, KpTaps &cur, KpTaps &nxt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // ring writes, LUT pieces and taps have landed
        __syncthreads();
        // LUT row lo + st + 1 into the buffer the describe waves have just left (never beyond the span: during the last
        // step of the consumer workgroup buffer 0 takes the epilogue's first whitening step)
        if (sp.lo + st + 1 < sp.hi) issue_lut_row<W>(lut_rows, sp.lo + st + 1, s_mem + ((st + 1) & 1) * kRowBytes, pw, lane);
        if (q_step0 + st + 1 < sp.q_end) sm.request(q_step0 + st + 1, 0, nxt);
        __builtin_amdgcn_sched_barrier(0);
        if (q_step0 + st < sp.q_end) sm.finish(q_step0 + st, 0, 0, cur);
    };
#pragma unroll 1
    for (int st = 0; st < sp.hi - sp.lo; st += 2) {
        step(st, ta, tb);
        step(st + 1, tb, ta);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (sp.consumer) {
#pragma unroll 1
        for (int e = 0; e < 12; ++e) __syncthreads();   // finish_descriptors: one on entry, one per whitening step
    }
}

constexpr int kSplitSpinMax = 1 << 21;   // polls of a consumer wave for its partners' partial sums before it gives up (~1 s)

}  // namespace

// grid = min(#batches, #CUs) persistent workgroups of 8 waves; a batch is 128 patches (16 per wave).
// Per patch row g (32 per batch): one barrier, after which the LUT row g is in LDS (issued a whole row
// earlier, double-buffered) and every wave has left row g-1.  Raw patch rows arrive by LDS-DMA into a
// 6-slot ring private to each wave, one row per step, so the main loop holds no patch data in VGPRs beyond
// the three blurred rows of the gradient stencil.
// Algorithmic HBM bytes per patch: 4096 read + 512 written; the kernel moves nothing else.
// W = waves per workgroup: 8 (128 patches, two waves per SIMD) for throughput; 4 (64 patches) when the whole request
// fits one round of workgroups anyway, so that it spreads over twice as many CUs with a SIMD to each wave.
// SRC = kSrcKeypoints (keypoint mode, f16x3 pooling): the workgroup has 2 W waves.  Waves 0 .. W-1 are the describe waves
// of the text above; waves W .. 2W-1 are their producers (kp_produce): wave W + i samples the patches of describe wave i from
// the pyramid straight into its row ring, so a sampled patch never leaves the CU.  `patches` is unused, `ks` says what to
// sample.  The two kinds of wave sit side by side on every SIMD -- one waits on the texture path while the other computes.
// SRC = kSrcKeypointsSplit (W = 2; requests small enough that every workgroup below is resident at once): a launch of the
// form above is a wave walking 32 rows for its 16 patches whatever the request's size -- ~70 us of latency at the reference's
// own 2000-3000 keypoints, on a quarter of the chip.  Here R = 2 or 4 workgroups share a batch of 32 keypoints by ROWS:
// workgroup `role` samples and pools rows [32 role / R, 32 (role + 1) / R) only (pooling is a sum over pixels), the first
// R - 1 leave their 24 accumulator tiles in global memory and count themselves in, the last one -- the highest workgroup id of
// the batch, so that those it waits for were dispatched before it -- adds them to its own in a fixed order and runs the
// epilogue.  A descriptor's sum is then R partial chains instead of one: bit-identical within the form, ~1e-7 relative
// from the whole-patch forms.  Same sampling, blur, gradient and MFMA arithmetic per row.
template <int ANGLE, int POOL, int W, int SRC>
__global__ __launch_bounds__(SRC != kSrcPatches ? 128 * W : 64 * W) void mkd_pool(
    const float *__restrict__ patches, long n_host, const unsigned long long *__restrict__ n_dev,
    const unsigned char *__restrict__ lut_rows, const short *__restrict__ colmap, const unsigned char *__restrict__ wfrag,
    const float *__restrict__ bias, float *__restrict__ out, float *__restrict__ raw_out,
    std::conditional_t<SRC != kSrcPatches, KpSource, int> ks, unsigned long long *__restrict__ clk) {
    constexpr bool kKp = SRC != kSrcPatches, kSplit = SRC == kSrcKeypointsSplit;
    static_assert(!kSplit || W == 2, "the row-split form is the narrow (2 + 2 wave) form");
    // clk (LF_MKD_FLAG_KERNEL_TIMING, else null): workgroup 0 stamps the shader clock (s_memtime) and the constant 100 MHz
    // clock (s_memrealtime) on entry and on exit -- the clock the chip sustained under THIS launch is their quotient
    // (MI355X_MICROARCH.md, DVFS give-back); the values go to a buffer nothing else reads
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter();
        clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    constexpr int kSlots = kKp ? kRingSlotsKp : kRingSlots;
    static_assert(!kKp || POOL == LF_POOL_F16X3, "keypoint mode pools in f16x3");
    constexpr int kMPool = POOL == LF_POOL_F16_FP6 ? LF_POOL_F16X3 : POOL;   // the m stream keeps the three-term form
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[kRingOff + W * kSlots * 2048 +
                                                                (kKp ? kLevelTableBytes + kMaxPyrLevels * 8 : 0)];
    // number of patches: given by the host, or (graph-captured pipelines) left on the device by the previous stage
    const long n = n_dev ? (long)*n_dev : n_host;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int addr_l = ((lane - 16) & 63) * 4, addr_r = ((lane + 16) & 63) * 4;
    const bool has_l = q > 0, has_r = q < 3;
    const long nbatch = (n + 16 * W - 1) / (16 * W);
    unsigned char *ring = s_mem + kRingOff + wave * (kSlots * 2048);
    BatchWalk walk{(long)blockIdx.x, nbatch, (long)gridDim.x};
    [[maybe_unused]] RowSpan span{0, 32, 0, 32, 7, true};
    [[maybe_unused]] int role = 0, n_roles = 1;
    if constexpr (kKp) {
        if constexpr (kSplit) {
            // workgroup id -> (batch, role): ids 8 R k + 8 role + x serve batch 8 k + x -- the R workgroups of a batch sit 8 ids
            // apart (dispatch deals ids to the 8 XCDs round-robin: they share an L2, a matter of speed only) and the consumer
            // (role R - 1) has the highest id of its batch (its partners were dispatched before it)
            n_roles = ks.split;
            const long id = blockIdx.x, grp = id / (8 * n_roles), in = id - grp * (8 * n_roles);
            role = (int)(in >> 3);
            const long b = grp * 8 + (in & 7);
            walk = BatchWalk{b, b < nbatch ? b + 1 : b, 1};
            span = row_span(role, n_roles);
        } else {
            walk = kp_batches(nbatch);
        }
        if (walk.cur >= walk.end) return;
        // per-level geometry into LDS (lanes look up different levels: see LevelTable)
        int *lv = reinterpret_cast<int *>(s_mem + kRingOff + W * kSlots * 2048);
        long *lv_off = reinterpret_cast<long *>(lv + 5 * kMaxPyrLevels);
        if (threadIdx.x < kMaxPyrLevels) {
            const int l = threadIdx.x;
            int w_ = 0, h_ = 0, pi_ = 0, ap_ = 0;
            long of_ = 0;
#pragma unroll
            for (int j = 0; j < kMaxPyrLevels; ++j)
                if (j == l) { w_ = ks.pd.w[j]; h_ = ks.pd.h[j]; pi_ = ks.pd.pitch[j]; ap_ = ks.pd.apron[j]; of_ = ks.pd.offset[j]; }
            lv[l] = w_; lv[kMaxPyrLevels + l] = h_; lv[2 * kMaxPyrLevels + l] = pi_; lv[3 * kMaxPyrLevels + l] = ap_;
            lv_off[l] = of_;
        }
        __syncthreads();
        const LevelTable lt{ks.pd.levels, lv, lv + kMaxPyrLevels, lv + 2 * kMaxPyrLevels, lv + 3 * kMaxPyrLevels};
        if (wave >= W) {
            if constexpr (kSplit) kp_produce_split<W>(ks, lt, lv_off, n, walk.cur, span, s_mem, lut_rows, wave - W, lane);
            else kp_produce<W>(ks, lt, lv_off, n, walk, s_mem, lut_rows, wave - W, lane);
            return;
        }

        if constexpr (ablate::kConsumerPrio >= 0) __builtin_amdgcn_s_setprio(ablate::kConsumerPrio);
    }
    // DMA writes are lane-linear (lane l -> bytes [16l, 16l+16) of a 1 KiB piece): lane (p, q) moves the 16-B
    // chunk q of its patch's half-row; the reader (p, q) needs chunks 2(q&1), 2(q&1)+1 of half q>>1.
    const unsigned char *ring_lane = ring + (q >> 1) * 1024 + ((2 * (q & 1)) * 16 + p) * 16;

    long batch = walk.cur;
    if (batch >= walk.end) return;
    // lanes beyond the last patch recompute it: the wave's base is clamped the same way, so lane offsets stay >= 0
    auto raw_src = [&](const float *pt, long b) {
        const long b0 = b * (16 * W) + wave * 16;
        const long wave0 = b0 < n ? b0 : n - 1;                 // uniform
        const long pidx = b0 + p < n ? b0 + p : n - 1;
        return RawSrc{reinterpret_cast<const unsigned char *>(pt + wave0 * 1024), (unsigned)(pidx - wave0) * 4096u + 16u * q};
    };
    {
        if constexpr (!kKp) {
            const RawSrc src = raw_src(patches, batch);
#pragma unroll
            for (int r = -2; r <= 3; ++r) issue_raw_row(src, r, ring, r + 2);
        }
        issue_lut_row<W>(lut_rows, kSplit ? span.lo : 0, s_mem, wave, lane);
    }
    int slot0 = 0;   // keypoint mode: ring slot of raw row 0 of the current batch
    unsigned par = 0;  // LUT row buffer holding the row about to be consumed
    ablate::PhaseClock phase;   // (instrument builds only: tools/phase_timing.py)
    phase.start();

    for (; batch < walk.end; batch += walk.step) {
        // Launder the (uniform) table and buffer pointers once per batch.  Otherwise hipcc hoists one 64-bit per-lane VGPR
        // address per DMA / fragment load / store out of the batch loop and spills them (1.4 KB per lane for the
        // whitening fragments alone).  Spills matter beyond their cost here: scratch loads and stores count on vmcnt
        // like the LDS-DMA requests, stores retire out of order with respect to loads, and the f16 epilogue waits for its
        // DMA steps with COUNTED waits (wait_vmcnt<kDma>) -- a scratch store in flight there would let a wait pass
        // before its DMA has landed.  The Makefile fails the build if an f16 instantiation of this kernel uses scratch.
        const unsigned char *wf = wfrag, *lr = lut_rows;
        const float *bs = bias, *pt = patches;
        float *o = out, *ro = raw_out;
        const short *cm = colmap;
        asm volatile("" : "+s"(wf), "+s"(bs), "+s"(lr), "+s"(o), "+s"(ro), "+s"(cm), "+s"(pt));
        const long base = batch * (16 * W) + wave * 16;
        const bool more = batch + walk.step < walk.end;
        // (keypoint mode: pt is null and these stay unused)
        const RawSrc src = kKp ? RawSrc{nullptr, 0u} : raw_src(pt, batch);
        const RawSrc src_next = !kKp && more ? raw_src(pt, batch + walk.step) : src;

        f32x4 acc[kAccTiles];
#pragma unroll
        for (int t = 0; t < kAccTiles; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float cur[8], prv[8], cur_l = 0.f, cur_r = 0.f;  // blurred rows g and g-1 (row -1 replicates row 0)
        int s0 = 1;                                      // ring slot of raw row g-1 (rows g-1..g+3 feed hb(g+1))

        // The first and the last row of a batch differ from the 30 between them (two blurs and a later ring request /
        // no blur and the next batch's first rows): they are separate copies of the row body, so that the loop over the
        // middle rows carries none of their branches -- hipcc speculated the last row's "row 32 = row 31" copy into
        // every row (ten v_mov).
        auto patch_row = [&](auto kind, const int g) __attribute__((always_inline)) {
            constexpr bool kFirst = decltype(kind)::value == 0, kLast = decltype(kind)::value == 2;
            constexpr bool kFirstInner = decltype(kind)::value == 3;   // row-split form: a first row that is not row 0

            // LUT row g and ring row g+3 have landed (own DMA: vmcnt; everyone's: barrier); row g-1 is done
            phase.mark(7);
            if constexpr (!ablate::kNoRowSync) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            phase.mark(0);
            const unsigned char *brow = s_mem + par * kRowBytes + lane * 16;
            // row buffer par ^ 1 is free: next LUT row; during row 31 the f16 epilogue's first whitening step instead
            // (keypoint mode: the producer waves request LUT rows 1..31 -- an LDS-DMA request stalls its issuer for 60-180
            // cycles, which a producer can afford -- so a describe wave issues no memory instruction in the row loop)
            if (f16_family(POOL) ? !kLast : (!kLast || more)) {
                if constexpr (!kKp) issue_lut_row<W>(lr, (g + 1) & 31, s_mem + (par ^ 1) * kRowBytes, wave, lane);
            } else if (f16_family(POOL)) {
                issue_w_step<W>(wf, 0, s_mem, wave, lane);
            }
            par ^= 1;
            BFrag bm[3] = {load_b(brow, 0), load_b(brow, 1), load_b(brow, 2)};   // m-stream fragments

            // Raw row g+4 goes into the slot of row g-2, whose last reader was the blur of the previous iteration: for
            // g >= 1 it is requested here, a whole row before the vmcnt(0) that waits for it (counters: the waves spend
            // 29 % of their time in s_waitcnt and only 2 % of that on LDS), for g == 0 after the first blur below.
            if constexpr (!kKp)
                if (!kFirst && !kLast && g <= 29) issue_raw_row(src, g + 4, ring, s0 == 0 ? kRingSlots - 1 : s0 - 1);
            // keypoint mode: the five raw rows of hb(y) are rows y-2 .. y+2 clamped to the patch (replicate border), each in
            // its slot of the producer's ring
            auto kp_slot = [&](int y, int i) {
                int r_ = y - 2 + i;
                r_ = r_ < 0 ? 0 : (r_ > 31 ? 31 : r_);
                return (slot0 + r_) % kRingSlotsKp;
            };
            if (kFirst) {  // first blurred row of the batch: rows -2..2 sit in slots 0..4
                if constexpr (kKp)
                    blur_row_impl(ring_lane, [&](int i) { return kp_slot(0, i); }, addr_l, addr_r, has_l, has_r, cur, cur_l, cur_r);
                else
                    blur_row(ring_lane, 0, addr_l, addr_r, has_l, has_r, cur, cur_l, cur_r);
#pragma unroll
                for (int x = 0; x < 8; ++x) prv[x] = cur[x];
            }
            if constexpr (kFirstInner) {   // blurred rows g - 1 and g themselves: nothing replicates
                float edge_l, edge_r;
                blur_row_impl(ring_lane, [&](int i) { return kp_slot(g - 1, i); }, addr_l, addr_r, has_l, has_r, prv, edge_l, edge_r);
                blur_row_impl(ring_lane, [&](int i) { return kp_slot(g, i); }, addr_l, addr_r, has_l, has_r, cur, cur_l, cur_r);
            }
            float nxt[8], nxt_l, nxt_r;
            if (!kLast) {  // hb(g+1) from raw rows g-1..g+3 = slots s0..s0+4
                if constexpr (kKp)
                    blur_row_impl(ring_lane, [&](int i) { return kp_slot(g + 1, i); }, addr_l, addr_r, has_l, has_r, nxt, nxt_l, nxt_r);
                else
                    blur_row(ring_lane, s0, addr_l, addr_r, has_l, has_r, nxt, nxt_l, nxt_r);
            } else {  // row 32 replicates row 31
#pragma unroll
                for (int x = 0; x < 8; ++x) nxt[x] = cur[x];
                nxt_l = cur_l;
                nxt_r = cur_r;
            }
            // the slot of raw row g-2 is free now (its last reader was the blur above when g == 0)
            asm volatile("" ::: "memory");
            if constexpr (!kKp) {
                if (kFirst) {
                    issue_raw_row(src, g + 4, ring, s0 == 0 ? kRingSlots - 1 : s0 - 1);
                } else if (kLast && more) {
#pragma unroll
                    for (int r = -2; r <= 3; ++r) issue_raw_row(src_next, r, ring, r + 2);  // next batch's first rows
                }
            }
            s0 = s0 == kRingSlots - 1 ? 0 : s0 + 1;
            phase.mark(1);

            f32x2 m[4], c1[4], s1[4];
            {
                f32x2 gx[4], gy[4], r2n[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {  // patch_gradients.glsl:94-100, two pixels at a time
                    const int x = 2 * e;
                    const f32x2 left = {x == 0 ? cur_l : cur[x - 1], cur[x]};
                    const f32x2 right = {cur[x + 1], x == 6 ? cur_r : cur[x + 2]};
                    // left - right, plus +0.0: the value unchanged, but -0.0 (a blurred -0.0 on the left of a +0.0 -- the blur of
                    // negative denormals underflows to it, fma or not) becomes +0.0, so that the sign bit gradient_direction
                    // takes from gx is never the sign of a zero: the shader's (cos, sin) = (1, 0) at gx == -0.0, atan2.glsl:29-45
                    gx[e] = (left - right) + pk_set(0.0f);
                    gy[e] = f32x2{nxt[x], nxt[x + 1]} - f32x2{prv[x], prv[x + 1]};      // down - up
                    r2n[e] = pk_fma(gy[e], gy[e], gx[e] * gx[e]);
                    const f32x2 r2 = r2n[e] + pk_set(1e-8f);
                    m[e] = f32x2{__builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(r2.x)),
                                 __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(r2.y))};
                }
                if constexpr (ablate::kNoFrontEnd) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { c1[e] = gx[e]; s1[e] = gy[e]; }
                } else {
                    gradient_direction4<ANGLE>(gx, gy, r2n, c1, s1);
                }
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                prv[x] = cur[x];
                cur[x] = nxt[x];
            }
            cur_l = nxt_l;
            cur_r = nxt_r;
            phase.mark(2);

            // m stream x (polar | cartesian) kernels: accumulator tiles 0-2
            BFrag gfrag = load_b(brow, 3);   // first harmonic: P0
            {
                AFrag<kMPool> am;
                am.set(m);
#pragma unroll
                for (int t = 0; t < 3; ++t) mma_part<kMPool, 0>(am, bm[t], acc[t]);
#pragma unroll
                for (int t = 0; t < 3; ++t) mma_part<kMPool, 1>(am, bm[t], acc[t]);
#pragma unroll
                for (int t = 0; t < 3; ++t) mma_part<kMPool, 2>(am, bm[t], acc[t]);
            }
            phase.mark(3);
            // cos / sin streams of the three harmonics x their four LUT tiles: accumulator tiles 3-23
            if constexpr (POOL == LF_POOL_F16_FP6) pool_harmonics_fp6(m, c1, s1, brow, gfrag, acc);
            else pool_harmonics<POOL>(m, c1, s1, brow, gfrag, acc);
            phase.mark(4);
            phase.mark(5);
        };
        if constexpr (kSplit) {
            if (span.lo == 0) patch_row(std::integral_constant<int, 0>(), 0);
            else patch_row(std::integral_constant<int, 3>(), span.lo);
#pragma unroll 1
            for (int g = span.lo + 1; g < span.hi - 1; ++g) patch_row(std::integral_constant<int, 1>(), g);
            if (span.consumer) patch_row(std::integral_constant<int, 2>(), span.hi - 1);
            else patch_row(std::integral_constant<int, 1>(), span.hi - 1);
            // the partial sums meet: [batch][role][wave][tile][lane] f32x4, one counter per (batch, wave)
            unsigned *cnt = ks.xchg_cnt + batch * W + wave;
            // Every access to the exchange buffer and its counters is a relaxed atomic of agent scope: such a store is
            // complete (vmcnt) once it has reached the level of the memory system all XCDs see, such a load is served from
            // there -- no release / acquire fence, whose cost on this chip is a write-back and an invalidation of the whole L2
            // (the LUT rows and the pyramid every other workgroup is reading: measured, profiles/r06_row_split.md).
            typedef unsigned long long u64;
            if (!span.consumer) {
                u64 *dst = reinterpret_cast<u64 *>(ks.xchg + ((((long)batch * (n_roles - 1) + role) * W + wave) * kAccTiles) * 256) + lane;
#pragma unroll
                for (int t = 0; t < kAccTiles; ++t) {
                    const u64 v0 = ((u64)__float_as_uint(acc[t][1]) << 32) | __float_as_uint(acc[t][0]);
                    const u64 v1 = ((u64)__float_as_uint(acc[t][3]) << 32) | __float_as_uint(acc[t][2]);
                    __hip_atomic_store(dst + t * 128, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + t * 128 + 64, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's partial sums are where everyone sees them ...
                if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before it counts itself in
                if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
                    clk[2] = __builtin_readcyclecounter();
                    clk[3] = __builtin_amdgcn_s_memrealtime();
                }
                return;
            }
            int polls = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(n_roles - 1) &&
                   polls < kSplitSpinMax) {
                __builtin_amdgcn_s_sleep(1);
                ++polls;
            }
            asm volatile("" ::: "memory");
            if (polls >= kSplitSpinMax && lane == 0) atomicAdd(ks.xchg_cnt - 1, 1u);   // the error word: a partner never arrived
#pragma unroll 1
            for (int r = 0; r + 1 < n_roles; ++r) {
                const u64 *srcp = reinterpret_cast<const u64 *>(ks.xchg + ((((long)batch * (n_roles - 1) + r) * W + wave) * kAccTiles) * 256) + lane;
                // a partner's 24 tiles are requested in one go (48 loads in flight: one trip to memory per partner, not per
                // tile -- a workgroup of this form has a SIMD's 512 registers per wave) and added in tile order
                u64 v[2 * kAccTiles];
#pragma unroll
                for (int t = 0; t < 2 * kAccTiles; ++t)
                    v[t] = __hip_atomic_load(srcp + t * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < kAccTiles; ++t)
                    acc[t] += f32x4{__uint_as_float((unsigned)v[2 * t]), __uint_as_float((unsigned)(v[2 * t] >> 32)),
                                    __uint_as_float((unsigned)v[2 * t + 1]), __uint_as_float((unsigned)(v[2 * t + 1] >> 32))};
            }
            if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            // (the epilogue counts its own LDS-DMA requests on vmcnt: nothing of the above may still be in flight)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            patch_row(std::integral_constant<int, 0>(), 0);
#pragma unroll 1
            for (int g = 1; g < ablate::kRows - 1; ++g) patch_row(std::integral_constant<int, 1>(), g);
            patch_row(std::integral_constant<int, 2>(), ablate::kRows - 1);
        }
        if constexpr (ablate::kNoEpilogue) {
            f32x4 sum = acc[0];
            for (int t = 1; t < kAccTiles; ++t) sum += acc[t];
            if (base + p < n) *reinterpret_cast<f32x4 *>(out + (base + p) * 128 + 4 * q) = sum;
            if (f16_family(POOL)) {
                __syncthreads();
                if (more) issue_lut_row<W>(lr, 0, s_mem, wave, lane);
            }
        } else {
            finish_descriptors<POOL, W>(acc, lane, wave, base + p < n, base + p, cm, wf, bs, o, ro, s_mem, lr, more);
        }
        phase.mark(6);
        if constexpr (kKp) slot0 = (slot0 + 8) % kRingSlotsKp;
    }
    phase.dump(out, wave, lane);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[2] = __builtin_readcyclecounter();
        clk[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// This source is compiled twice (Makefile): as is -- the patch-mode launcher, scheduled with max-ILP -- and with
// -DLF_DESCRIBE_KP -- the keypoint-mode launcher alone, under hipcc's default scheduling strategy: with max-ILP the two exact
// angle modes of the keypoint kernel need 259 registers (three spilled, which the counted vmcnt waits of the epilogue
// forbid), with the default strategy 203-209.
#ifndef LF_DESCRIBE_KP
// n_dev != nullptr: the patch count is read on the device (<= n, which then only sizes the grid)
void launch_describe(const float *patches, long n, const unsigned long long *n_dev, const DeviceConsts &dc,
                     int angle_mode, int pool_mode, float *out, float *raw_out, int num_cus, hipStream_t stream, int waves,
                     unsigned long long *clk) {
    if (n <= 0) return;
    // one 100-152 KiB-LDS workgroup per CU; requests of at most one round of 64-patch workgroups take the 4-wave form
    const bool small = ablate::kForceFourWaves || waves == 4 || (waves != 8 && n <= 64L * num_cus);
    const long nbatch = small ? (n + 63) / 64 : (n + 127) / 128;
    const unsigned grid = (unsigned)(nbatch < num_cus ? nbatch : num_cus);
    const bool f16 = f16_family(pool_mode);
    const unsigned char *lut = pool_mode == LF_POOL_F16_FP6 ? reinterpret_cast<const unsigned char *>(dc.pool_b_fp6)
                               : f16                        ? reinterpret_cast<const unsigned char *>(dc.pool_b_f16)
                                                            : reinterpret_cast<const unsigned char *>(dc.pool_b_f32);
    const unsigned char *wf = f16 ? reinterpret_cast<const unsigned char *>(dc.white_a_f16)
                                  : reinterpret_cast<const unsigned char *>(dc.white_a_f32);
#define LF_LAUNCH_W(A, P, WV)                                                                                          \
    hipLaunchKernelGGL((mkd_pool<A, P, WV, kSrcPatches>), dim3(grid), dim3(64 * WV), 0, stream, patches, n, n_dev, lut, \
                       dc.colmap, wf, dc.white_bias, out, raw_out, 0, clk)
#define LF_LAUNCH(A, P)            \
    do {                           \
        if (small) LF_LAUNCH_W(A, P, 4); \
        else LF_LAUNCH_W(A, P, 8); \
    } while (0)
    if (pool_mode == LF_POOL_F16_FP6) {
        if (angle_mode == LF_ANGLE_EXACT) LF_LAUNCH(LF_ANGLE_EXACT, LF_POOL_F16_FP6);
        else if (angle_mode == LF_ANGLE_EXACT_ZERO) LF_LAUNCH(LF_ANGLE_EXACT_ZERO, LF_POOL_F16_FP6);
        else LF_LAUNCH(LF_ANGLE_SHADER, LF_POOL_F16_FP6);
    } else if (f16) {
        if (angle_mode == LF_ANGLE_EXACT) LF_LAUNCH(LF_ANGLE_EXACT, LF_POOL_F16X3);
        else if (angle_mode == LF_ANGLE_EXACT_ZERO) LF_LAUNCH(LF_ANGLE_EXACT_ZERO, LF_POOL_F16X3);
        else LF_LAUNCH(LF_ANGLE_SHADER, LF_POOL_F16X3);
    } else {
        if (angle_mode == LF_ANGLE_EXACT) LF_LAUNCH(LF_ANGLE_EXACT, LF_POOL_F32);
        else if (angle_mode == LF_ANGLE_EXACT_ZERO) LF_LAUNCH(LF_ANGLE_EXACT_ZERO, LF_POOL_F32);
        else LF_LAUNCH(LF_ANGLE_SHADER, LF_POOL_F32);
    }
#undef LF_LAUNCH_W
#undef LF_LAUNCH
}

#else   // LF_DESCRIBE_KP
// Keypoint mode in one launch: 4 describe waves + 4 producer waves per workgroup, 64 keypoints per batch, one workgroup
// per CU (160 KB of LDS), persistent; requests of at most 8192 keypoints: 2 + 2 waves, 32 keypoints per workgroup (below).
static long nbatch32(long n) { return (n + 31) / 32; }
size_t kp_split_exchange_bytes(int num_cus) {
    // R = 4: num_cus / 4 batches x 3 partial roles; R = 2: num_cus / 2 x 1 -- the larger of the two, x 2 waves x 24 tiles x 1 KiB
    return size_t(num_cus / 4 + 8) * 3 * 2 * kAccTiles * 1024;
}
size_t kp_split_counter_words(int num_cus) { return 1 + size_t(num_cus / 2 + 8) * 2; }   // the error word, then [batch][wave]

void launch_describe_keypoints(const float *pyr, long pyr_stride, const PyramidDesc &pd, const float *kps,
                               const unsigned *frame_of_kp, unsigned n_frames, long n, const unsigned long long *n_dev,
                               float psf, const DeviceConsts &dc, int angle_mode, float *out, int num_cus, hipStream_t stream,
                               unsigned long long *clk, float *xchg, unsigned *xchg_words, long form_n) {
    if (n <= 0) return;
    if (form_n < n) form_n = n;
    // A request of at most one round of 32-keypoint workgroups (<= 32 x CUs = 8192 keypoints: a frame at the reference's own
    // settings, top_n 2000 / max_features 3000) takes the 2 + 2-wave form: every wave has a SIMD to itself -- the describe
    // waves do not share theirs with a producer -- and the launch spreads over twice as many CUs.  It is a launch's latency
    // that this shortens; on full batches the 4 + 4 form's eight waves per CU are the faster arrangement.
    const bool narrow = n <= 32L * num_cus;      // (with n_dev, n is the capacity: the count on the device is no larger)
    const long per_wg = narrow ? 32 : 64;
    const long nbatch = (n + per_wg - 1) / per_wg;
    const unsigned grid = (unsigned)(nbatch < num_cus ? nbatch : num_cus);
    const unsigned char *lut = reinterpret_cast<const unsigned char *>(dc.pool_b_f16);
    const unsigned char *wf = reinterpret_cast<const unsigned char *>(dc.white_a_f16);
    KpSource ks;
    ks.pyr = pyr; ks.pyr_stride = pyr_stride; ks.kps = kps; ks.frame_of = frame_of_kp; ks.n_frames = n_frames ? n_frames : 1u;
    ks.psf = psf; ks.pd = pd;
    ks.split = 1; ks.xchg = nullptr; ks.xchg_cnt = nullptr;
    // The row-split form where R workgroups per batch of 32 are all resident at once (one per CU): R = 4 up to 8 x CUs
    // keypoints (2048), R = 2 up to 16 x CUs (4096) -- the reference's own operating point (top_n 2000, max_features 3000).
    // LF_MKD_KP_SPLIT=1 / 2 / 4 forces a form where it fits (tests, A/B runs).
    if (xchg && xchg_words) {
        // (the form follows form_n -- the capacity of the request the launch belongs to: lf_mkd_detect stage by stage, which
        //  knows its keypoint count, takes the form its recorded twin, which does not, takes for max_out: the same bits)
        const long nf8 = (nbatch32(form_n) + 7) / 8 * 8, nb8 = (nbatch32(n) + 7) / 8 * 8;
        int r = nf8 * 4 <= num_cus ? 4 : (nf8 * 2 <= num_cus ? 2 : 1);
        if (const char *e = getenv("LF_MKD_KP_SPLIT")) {
            const int want = atoi(e);
            if (want == 1 || ((want == 2 || want == 4) && nf8 * want <= num_cus)) r = want;
        }
        if (r > 1) {
            ks.split = r; ks.xchg = xchg; ks.xchg_cnt = xchg_words + 1;
            const unsigned sgrid = (unsigned)(nb8 * r);
#define LF_LAUNCH_KP_SPLIT(A)                                                                                                  \
    hipLaunchKernelGGL((mkd_pool<A, LF_POOL_F16X3, 2, kSrcKeypointsSplit>), dim3(sgrid), dim3(256), 0, stream,                    \
                       (const float *)nullptr, n, n_dev, lut, dc.colmap, wf, dc.white_bias, out, (float *)nullptr, ks, clk)
            if (angle_mode == LF_ANGLE_EXACT) LF_LAUNCH_KP_SPLIT(LF_ANGLE_EXACT);
            else if (angle_mode == LF_ANGLE_EXACT_ZERO) LF_LAUNCH_KP_SPLIT(LF_ANGLE_EXACT_ZERO);
            else LF_LAUNCH_KP_SPLIT(LF_ANGLE_SHADER);
#undef LF_LAUNCH_KP_SPLIT
            return;
        }
    }
#define LF_LAUNCH_KP_W(A, WV)                                                                                          \
    hipLaunchKernelGGL((mkd_pool<A, LF_POOL_F16X3, WV, kSrcKeypoints>), dim3(grid), dim3(128 * WV), 0, stream,          \
                       (const float *)nullptr, n, n_dev, lut, dc.colmap, wf, dc.white_bias, out, (float *)nullptr, ks, clk)
#define LF_LAUNCH_KP(A)                 \
    do {                                \
        if (narrow) LF_LAUNCH_KP_W(A, 2); \
        else LF_LAUNCH_KP_W(A, 4);      \
    } while (0)
    if (angle_mode == LF_ANGLE_EXACT) LF_LAUNCH_KP(LF_ANGLE_EXACT);
    else if (angle_mode == LF_ANGLE_EXACT_ZERO) LF_LAUNCH_KP(LF_ANGLE_EXACT_ZERO);
    else LF_LAUNCH_KP(LF_ANGLE_SHADER);
#undef LF_LAUNCH_KP_W
#undef LF_LAUNCH_KP
}
#endif  // LF_DESCRIBE_KP

}  // namespace lfmkd
