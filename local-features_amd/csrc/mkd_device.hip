// MKD descriptor path: device code for gfx950 (CDNA4, wave64).
//
// Kernels (reference stage each one replaces; paths under local_features/src/vulkan/shaders/):
//   mkd_pool          mkd/patch_gradients.glsl:72-104 + mkd/embedding.glsl:53-121 (both variants)
//                     + mkd/normalize.glsl:22-142 + mkd/whitening.glsl:22-77 + mkd/normalize_final.glsl
//   sample_patches    mkd/patch_gradients.glsl:42-70
//   pyr_*             blur.glsl, swt.glsl (all levels of the a-trous stack), blur_pyramid.glsl, patch_pyramid.rs blits
//   orient_*          keypoint_orientation.glsl:36-171 (+ the ordered compaction that replaces its atomic append)
//   scan_extrema      swt_sub.glsl:17-30 + scan_extrema.glsl:36-241; cubes_* = its ordered compaction
//   topk_filter       the host blob filter of detect_top_n (vulkan/mod.rs:1753-1786) on the device; segments_* batch it
//   (the matcher lives in mkd_match.hip)
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

#include "mkd_device.h"

namespace lfmkd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Two adjacent pixels per register pair: arithmetic on f32x2 compiles to v_pk_{mul,add,fma}_f32, which issue
// in the time of one scalar-f32 VALU instruction (tools/micro/valu_rate.hip) -- the kernel is VALU-issue-bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_set(float v) { return f32x2{v, v}; }

constexpr int kTiles = 21;   // accumulator tiles (16 packed columns each), see mkd_consts.hpp
// Accumulator tile numbering: m 0-2 | rel cos k 3+2(k-1).. | rel sin k 9+2(k-1).. | abs cos k 14+k | abs sin k 17+k.
// The cos and sin streams of one harmonic meet the same LUT columns, so a LUT row holds 12 unique tiles:
//   0-2 m | 3-5 abs k=1..3 | 6-11 rel k=1..3 (two tiles each).
constexpr int kUniqueTiles = 12;

// 5-tap sigma=0.7 kernel, patch_gradients.glsl:22-28
constexpr float kB0 = 0.0096f, kB1 = 0.2054f, kB2 = 0.5699f;

// LDS map of the pooling kernel (bytes)
constexpr int kRowBytes = kUniqueTiles * 2 * 1024;   // 24576: one LUT row image, 2 x 1 KiB pieces per tile
constexpr int kPhiOff = 2 * kRowBytes;               // cos/sin(phi) table, 8 KiB
constexpr int kRingOff = kPhiOff + 8192;             // raw patch rows: [wave 8][slot 6][2 KiB]
constexpr int kRingSlots = 6;
// total: kRingOff + waves * kRingSlots * 2048 = 155648 B for 8 waves, 106496 B for 4

__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

__device__ __forceinline__ void lds_dma16(const void *gsrc, void *ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
}

// one LUT row (24 pieces) into an LDS row buffer, 24 / W pieces per wave
template <int W>
__device__ __forceinline__ void issue_lut_row(const unsigned char *__restrict__ lut_rows, int row,
                                              unsigned char *lds_row, int wave, int lane) {
    const unsigned char *g = lut_rows + (size_t)row * kRowBytes + lane * 16;
#pragma unroll
    for (int j = 0; j < 24 / W; ++j) lds_dma16(g + (wave + W * j) * 1024, lds_row + (wave + W * j) * 1024);
}

// Whitening fragments are staged through the two LUT row buffers (idle during the epilogue) one 16 KiB step at a
// time, shared by the workgroup's waves: three slots, placed so that step 0 lies in row buffer 0 (it is requested during
// patch row 31, when only that buffer is free) and the last step in row buffer 1 (so that LUT row 0 of the next batch
// can be requested into buffer 0 while the last step is still being consumed).
constexpr int kWStepBytes = 16384;
__device__ __forceinline__ constexpr int wstage_slot(int step) { return step % 3 == 0 ? 0 : (step % 3 == 1 ? 32768 : 16384); }
static_assert(wstage_slot(0) + kWStepBytes <= kRowBytes && wstage_slot(10) >= kRowBytes && 3 * kWStepBytes <= 2 * kRowBytes, "");

template <int W>
__device__ __forceinline__ void issue_w_step(const unsigned char *__restrict__ wfrag, int step, unsigned char *lds,
                                             int wave, int lane) {
    const unsigned char *g = wfrag + (size_t)step * kWStepBytes + lane * 16;
    unsigned char *d = lds + wstage_slot(step);
#pragma unroll
    for (int j = 0; j < 16 / W; ++j) lds_dma16(g + (wave + W * j) * 1024, d + (wave + W * j) * 1024);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one raw patch row (16 patches x 128 B) of this wave into ring slot `slot`; rows outside [0,31] replicate
__device__ __forceinline__ void issue_raw_row(const float *__restrict__ src_lane, int row, unsigned char *ring,
                                              int slot) {
    const int y = row < 0 ? 0 : (row > 31 ? 31 : row);
    lds_dma16(src_lane + y * 32, ring + slot * 2048);
    lds_dma16(src_lane + y * 32 + 16, ring + slot * 2048 + 1024);
}

// No implicit contraction in the describe kernel (down to the end of mkd_pool): every fused multiply-add in it is written
// as one.  The 4-wave and 8-wave forms are separate instantiations, and left to itself hipcc may contract an
// expression in one and not in the other -- a descriptor must not depend on the size of the request it was part of
// (tests/test_gpu_parity.py::test_full_size_properties compares the two forms bit for bit).
#pragma clang fp contract(off)

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
        const bool sw0 = ax0 < ay0, sw1 = ax1 < ay1;
        const f32x2 mn = {__builtin_fminf(ax0, ay0), __builtin_fminf(ax1, ay1)};
        const f32x2 mx = {__builtin_fmaxf(ax0, ay0), __builtin_fmaxf(ax1, ay1)};
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
        // gx == 0: the shader returns 0 both for atan2(0, 0) and (its quirk) for atan2(0, y != 0)
        if (gx.x == 0.f) { cr0 = 1.f; sr0 = 0.f; }
        if (gx.y == 0.f) { cr1 = 1.f; sr1 = 0.f; }
        ct = f32x2{cr0, cr1};
        st = f32x2{sr0, sr1};
    }
}

// ---- B fragments (LUT) and A fragments (streams) -------------------------------------------------
// One unique LUT tile = two 1 KiB pieces: f32: pixels 0-3 / 4-7 of the lane's segment (K = 4 MFMAs);
// f16: hi / lo halves of all 8 pixels (K = 32 MFMAs).  16 B per lane either way.
struct BFrag { u32x4 p0, p1; };

template <int UT>
__device__ __forceinline__ BFrag load_b(const unsigned char *brow) {
    BFrag b;
#ifdef LF_ABLATE_BLOAD  // timing-only build: no LUT fragment reads
    b.p0 = u32x4{(unsigned)UT, 1u, 2u, 3u}; b.p1 = b.p0; return b;
#endif
    b.p0 = *reinterpret_cast<const u32x4 *>(brow + (UT * 2 + 0) * 1024);
    b.p1 = *reinterpret_cast<const u32x4 *>(brow + (UT * 2 + 1) * 1024);
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
#ifdef LF_ABLATE_SPLIT  // timing-only build: no hi-lo split
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = __float_as_uint(a[e].x); lo[e] = __float_as_uint(a[e].y); }
        return;
#endif
        float one = 1.f;
        asm("" : "+v"(one));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned h = pack_rtz(a[e].x, a[e].y);
            const float h0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu));
            const float h1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h >> 16));
            hi[e] = h;
            lo[e] = pack_rtz(__builtin_fmaf(a[e].x, one, -h0), __builtin_fmaf(a[e].y, one, -h1));
        }
    }
};

// acc += LUT^T x stream for one (stream, tile): the LUT fragment is the MFMA's A operand (rows = packed
// columns), the stream its B operand (columns = patches), so a lane ends up holding packed columns
// 16t + 4(lane >> 4) + i of ITS OWN patch (lane & 15) -- which is what the fused epilogue needs.
// `part` selects one third of the f16 split so callers can interleave independent accumulators between the
// dependent MFMAs of one tile.
template <int POOL, int PART>
__device__ __forceinline__ void mma_part(const AFrag<POOL> &a, const BFrag &b, f32x4 &acc) {
#ifdef LF_ABLATE_MMA   // timing-only build: no matrix instructions, operands kept alive
    if constexpr (POOL == LF_POOL_F16X3) { asm volatile("" ::"v"(a.hi), "v"(a.lo), "v"(b.p0), "v"(b.p1)); return; }
#endif
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

// cos and sin streams of one harmonic against the NT tiles they share
template <int POOL, int NT>
__device__ __forceinline__ void mma_pair(const AFrag<POOL> &ac, const AFrag<POOL> &as, const BFrag (&b)[NT],
                                         f32x4 *acc_c, f32x4 *acc_s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) { mma_part<POOL, 0>(ac, b[t], acc_c[t]); mma_part<POOL, 0>(as, b[t], acc_s[t]); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { mma_part<POOL, 1>(ac, b[t], acc_c[t]); mma_part<POOL, 1>(as, b[t], acc_s[t]); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { mma_part<POOL, 2>(ac, b[t], acc_c[t]); mma_part<POOL, 2>(as, b[t], acc_s[t]); }
}

// Blurred row of this lane's segment from the raw-row ring (patch_gradients.glsl:72-92): vertical 5 taps over
// ring slots s0..s0+4, then horizontal 5 taps with the neighbours fetched from lanes -/+16.
// Returns the row plus its x-1 / x+8 neighbours.
__device__ __forceinline__ void blur_row(const unsigned char *ring_lane, int s0, int addr_l, int addr_r, bool has_l,
                                         bool has_r, float (&out)[8], float &out_l, float &out_r) {
    float vb[8];
    {
        const float kk[5] = {kB0, kB1, kB2, kB1, kB0};
        f32x2 v2[4];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int sl = s0 + i;
            sl = sl >= kRingSlots ? sl - kRingSlots : sl;
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
        float s = kB0 * ext[x];
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

// One orientation family (absolute or relative angle): harmonics k = 1..3 by angle addition; harmonic k
// uses unique LUT tiles [U0 + NT*(k-1), +NT) and accumulator tiles C0 + NT*(k-1).. (cos), S0 + NT*(k-1).. (sin).
// `bnext` holds the fragments of harmonic 1 on entry (prefetched by the caller).
template <int POOL, int NT, int U0, int C0, int S0>
__device__ __forceinline__ void pool_family(const f32x2 (&m)[4], const f32x2 (&c1)[4], const f32x2 (&s1)[4],
                                            const unsigned char *brow, BFrag (&b)[NT], f32x4 (&acc)[kTiles]) {
    // The recurrence runs on the products themselves, (pk, qk) = m (cos, sin)(k theta), as a three-term (Chebyshev)
    // recurrence x_{k+1} = 2 c1 x_k - x_{k-1} with x_0 = (m, 0): one instruction per stream and harmonic instead of a
    // rotation of the unit vector (two) plus a product.
    f32x2 pk[4], qk[4], pp[4], qp[4], tc[4];
    AFrag<POOL> ac, as;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pk[e] = m[e] * c1[e]; qk[e] = m[e] * s1[e]; tc[e] = c1[e] + c1[e]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ac.set(pk);
        as.set(qk);
        BFrag bcur[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bcur[t] = b[t];
        if (k < 2) {  // fragments of the next harmonic: in flight behind this harmonic's matrix work
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (k == 0) b[t] = load_b<U0 + NT>(brow + t * 2048);
                else b[t] = load_b<U0 + 2 * NT>(brow + t * 2048);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 pn = pk_fma(tc[e], pk[e], k == 0 ? -m[e] : -pp[e]);
                const f32x2 qn = k == 0 ? tc[e] * qk[e] : pk_fma(tc[e], qk[e], -qp[e]);
                pp[e] = pk[e]; qp[e] = qk[e];
                pk[e] = pn; qk[e] = qn;
            }
        }
        mma_pair<POOL, NT>(ac, as, bcur, &acc[C0 + NT * k], &acc[S0 + NT * k]);
    }
}

// Epilogue, per wave and batch: acc[t][i] holds the pooled sum of packed column 16t + 4q + i for the lane's
// own patch.  normalize.glsl:22-142 (polar | cartesian | all), whitening.glsl:22-77 as an MFMA with the
// accumulators as B operand (out^T = W_T x raw, the mean folded into a bias), normalize_final.glsl.
// Whitening fragments (host: mkd_consts.cpp): f16: [step 11][row tile 8][hi|lo][lane][8], step s covers
// accumulator tiles 2s, 2s+1; f32: [tile 21][i 4][row tile 8][lane].
// f16 path: on entry step 0 of the fragments is on its way into its slot (requested by the caller during patch row 31);
// on exit LUT row 0 of the next batch is on its way into row buffer 0 if `more`.  Every wave of the workgroup must call.
template <int POOL, int W>
__device__ __forceinline__ void finish_descriptors(f32x4 (&acc)[kTiles], int lane, int wave, bool valid, long patch,
                                                   const short *__restrict__ colmap,
                                                   const unsigned char *__restrict__ wfrag,
                                                   const float *__restrict__ bias, float *__restrict__ out,
                                                   float *__restrict__ raw_out, unsigned char *s_mem,
                                                   const unsigned char *__restrict__ lut_rows, bool more) {
    const int q = lane >> 4;
    if constexpr (POOL == LF_POOL_F16X3) {
        __syncthreads();   // every wave has left patch row 31: row buffer 1 is free too
        issue_w_step<W>(wfrag, 1, s_mem, wave, lane);
        issue_w_step<W>(wfrag, 2, s_mem, wave, lane);
    }
    // tile 1 mixes polar (packed columns 0-8 of the tile) and cartesian (9-15) kernels of the m stream
    bool t1_polar[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t1_polar[i] = 4 * q + i < 9;
    float sp = 0.f, sc = 0.f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool polar = t == 0 || (t >= 3 && t <= 14) || (t == 1 && t1_polar[i]);
            if (t == 1) {
                const float v2 = acc[t][i] * acc[t][i];
                sp += polar ? v2 : 0.f;
                sc += polar ? 0.f : v2;
            } else if (polar) {
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
            const bool polar = t == 0 || (t >= 3 && t <= 14) || (t == 1 && t1_polar[i]);
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
    if constexpr (POOL == LF_POOL_F16X3) {
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

}  // namespace

// grid = min(#batches, #CUs) persistent workgroups of 8 waves; a batch is 128 patches (16 per wave).
// Per patch row g (32 per batch): one barrier, after which the LUT row g is in LDS (issued a whole row
// earlier, double-buffered) and every wave has left row g-1.  Raw patch rows arrive by LDS-DMA into a
// 6-slot ring private to each wave, one row per step, so the main loop holds no patch data in VGPRs beyond
// the three blurred rows of the gradient stencil.
// Algorithmic HBM bytes per patch: 4096 read + 512 written; the kernel moves nothing else.
// W = waves per workgroup: 8 (128 patches, two waves per SIMD) for throughput; 4 (64 patches) when the whole request
// fits one round of workgroups anyway, so that it spreads over twice as many CUs with a SIMD to each wave.
#ifdef LF_PHASE_TIMING   // timing-only build (tools/phase_timing.py): per-wave wall clock of the phases of a patch row,
                         // left by workgroup 0 in out[wave * 128 + phase]
#define LF_PT(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                      pt[i] += t_ - pt_prev; pt_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LF_PT(i) do { } while (0)
#endif

template <int ANGLE, int POOL, int W>
__global__ __launch_bounds__(64 * W) void mkd_pool(const float *__restrict__ patches, long n_host,
                                                const unsigned long long *__restrict__ n_dev,
                                                const unsigned char *__restrict__ lut_rows,
                                                const float *__restrict__ phi_cs,
                                                const short *__restrict__ colmap,
                                                const unsigned char *__restrict__ wfrag,
                                                const float *__restrict__ bias,
                                                float *__restrict__ out, float *__restrict__ raw_out) {
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[kRingOff + W * kRingSlots * 2048];
    // number of patches: given by the host, or (graph-captured pipelines) left on the device by the previous stage
    const long n = n_dev ? (long)*n_dev : n_host;
    float *s_phi = reinterpret_cast<float *>(s_mem + kPhiOff);
    // (cos, sin) per pixel in the table; in LDS per pixel pair as (cos, cos, sin, sin), so that a 16-byte read lands as
    // two aligned register pairs (interleaved, every pair cost three v_mov to take apart)
    for (int i = threadIdx.x; i < 2048; i += 64 * W) s_phi[(i >> 2) * 4 + (i & 1) * 2 + ((i >> 1) & 1)] = phi_cs[i];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int addr_l = ((lane - 16) & 63) * 4, addr_r = ((lane + 16) & 63) * 4;
    const bool has_l = q > 0, has_r = q < 3;
    const long nbatch = (n + 16 * W - 1) / (16 * W);
    unsigned char *ring = s_mem + kRingOff + wave * (kRingSlots * 2048);
    // DMA writes are lane-linear (lane l -> bytes [16l, 16l+16) of a 1 KiB piece): lane (p, q) moves the 16-B
    // chunk q of its patch's half-row; the reader (p, q) needs chunks 2(q&1), 2(q&1)+1 of half q>>1.
    const unsigned char *ring_lane = ring + (q >> 1) * 1024 + ((2 * (q & 1)) * 16 + p) * 16;

    auto lane_src = [&](long batch) {
        const long b0 = batch * (16 * W) + wave * 16;
        const long pidx = (b0 + p < n) ? b0 + p : n - 1;  // tail lanes recompute the last patch
        return patches + pidx * 1024 + 4 * q;
    };

    long batch = blockIdx.x;
    if (batch >= nbatch) return;
    {
        const float *src = lane_src(batch);
#pragma unroll
        for (int r = -2; r <= 3; ++r) issue_raw_row(src, r, ring, r + 2);
        issue_lut_row<W>(lut_rows, 0, s_mem, wave, lane);
    }
    unsigned par = 0;  // LUT row buffer holding the row about to be consumed
#ifdef LF_PHASE_TIMING
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_prev = __builtin_readcyclecounter();
#endif

    for (; batch < nbatch; batch += gridDim.x) {
        const long base = batch * (16 * W) + wave * 16;
        const float *src = lane_src(batch);
        const bool more = batch + gridDim.x < nbatch;
        const float *src_next = more ? lane_src(batch + gridDim.x) : src;

        f32x4 acc[kTiles];
#pragma unroll
        for (int t = 0; t < kTiles; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float cur[8], prv[8], cur_l = 0.f, cur_r = 0.f;  // blurred rows g and g-1 (row -1 replicates row 0)
        int s0 = 1;                                      // ring slot of raw row g-1 (rows g-1..g+3 feed hb(g+1))

        // The first and the last row of a batch differ from the 30 between them (two blurs and a later ring request /
        // no blur and the next batch's first rows): they are separate copies of the row body, so that the loop over the
        // middle rows carries none of their branches -- hipcc speculated the last row's "row 32 = row 31" copy into
        // every row (ten v_mov).
        auto patch_row = [&](auto kind, const int g) __attribute__((always_inline)) {
            constexpr bool kFirst = decltype(kind)::value == 0, kLast = decltype(kind)::value == 2;

            // LUT row g and ring row g+3 have landed (own DMA: vmcnt; everyone's: barrier); row g-1 is done
            LF_PT(7);
#ifndef LF_ABLATE_SYNC
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#endif
            LF_PT(0);
            const unsigned char *brow = s_mem + par * kRowBytes + lane * 16;
            // row buffer par ^ 1 is free: next LUT row; during row 31 the f16 epilogue's first whitening step instead
            if (POOL == LF_POOL_F16X3 ? !kLast : (!kLast || more))
                issue_lut_row<W>(lut_rows, (g + 1) & 31, s_mem + (par ^ 1) * kRowBytes, wave, lane);
            else if (POOL == LF_POOL_F16X3)
                issue_w_step<W>(wfrag, 0, s_mem, wave, lane);
            par ^= 1;
            BFrag bm[3] = {load_b<0>(brow), load_b<1>(brow), load_b<2>(brow)};   // m-stream fragments

#ifdef LF_ABLATE_FRONT  // timing-only build: no blur, no gradient direction
#define blur_row(rl, s, al, ar, hl, hr, o, ol, or_) do { const f32x4 a_ = *reinterpret_cast<const f32x4 *>(rl + (s) * 2048); const f32x4 b_ = *reinterpret_cast<const f32x4 *>(rl + (s) * 2048 + 256); for (int x_ = 0; x_ < 4; ++x_) { o[x_] = a_[x_]; o[4 + x_] = b_[x_]; } ol = o[0]; or_ = o[7]; } while (0)
#endif
            // Raw row g+4 goes into the slot of row g-2, whose last reader was the blur of the previous iteration: for
            // g >= 1 it is requested here, a whole row before the vmcnt(0) that waits for it (counters: the waves spend
            // 29 % of their time in s_waitcnt and only 2 % of that on LDS), for g == 0 after the first blur below.
            if (!kFirst && !kLast && g <= 29) issue_raw_row(src, g + 4, ring, s0 == 0 ? kRingSlots - 1 : s0 - 1);
            if (kFirst) {  // first blurred row of the batch: rows -2..2 sit in slots 0..4
                blur_row(ring_lane, 0, addr_l, addr_r, has_l, has_r, cur, cur_l, cur_r);
#pragma unroll
                for (int x = 0; x < 8; ++x) prv[x] = cur[x];
            }
            float nxt[8], nxt_l, nxt_r;
            if (!kLast) {  // hb(g+1) from raw rows g-1..g+3 = slots s0..s0+4
                blur_row(ring_lane, s0, addr_l, addr_r, has_l, has_r, nxt, nxt_l, nxt_r);
            } else {  // row 32 replicates row 31
#pragma unroll
                for (int x = 0; x < 8; ++x) nxt[x] = cur[x];
                nxt_l = cur_l;
                nxt_r = cur_r;
            }
            // the slot of raw row g-2 is free now (its last reader was the blur above when g == 0)
            asm volatile("" ::: "memory");
            if (kFirst) {
                issue_raw_row(src, g + 4, ring, s0 == 0 ? kRingSlots - 1 : s0 - 1);
            } else if (kLast && more) {
#pragma unroll
                for (int r = -2; r <= 3; ++r) issue_raw_row(src_next, r, ring, r + 2);  // next batch's first rows
            }
            s0 = s0 == kRingSlots - 1 ? 0 : s0 + 1;
            LF_PT(1);

            f32x2 m[4], c1[4], s1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // patch_gradients.glsl:94-100, two pixels at a time
                const int x = 2 * e;
                const f32x2 left = {x == 0 ? cur_l : cur[x - 1], cur[x]};
                const f32x2 right = {cur[x + 1], x == 6 ? cur_r : cur[x + 2]};
                const f32x2 gx = left - right;                                   // left - right
                const f32x2 gy = f32x2{nxt[x], nxt[x + 1]} - f32x2{prv[x], prv[x + 1]};   // down - up
                const f32x2 r2n = pk_fma(gy, gy, gx * gx);
                const f32x2 r2 = r2n + pk_set(1e-8f);
                m[e] = f32x2{__builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(r2.x)),
                             __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(r2.y))};
#ifdef LF_ABLATE_FRONT
                c1[e] = gx; s1[e] = gy;
#else
                gradient_direction<ANGLE>(gx, gy, r2n, c1[e], s1[e]);
#endif
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                prv[x] = cur[x];
                cur[x] = nxt[x];
            }
            cur_l = nxt_l;
            cur_r = nxt_r;
            LF_PT(2);

            // m stream x (polar | cartesian) kernels: accumulator tiles 0-2
            BFrag babs[1] = {load_b<3>(brow)};
            {
                AFrag<POOL> am;
                am.set(m);
#pragma unroll
                for (int t = 0; t < 3; ++t) mma_part<POOL, 0>(am, bm[t], acc[t]);
#pragma unroll
                for (int t = 0; t < 3; ++t) mma_part<POOL, 1>(am, bm[t], acc[t]);
#pragma unroll
                for (int t = 0; t < 3; ++t) mma_part<POOL, 2>(am, bm[t], acc[t]);
            }
            LF_PT(3);
            // absolute angle x cartesian kernels: unique tiles 3-5, accumulators 15-17 (cos), 18-20 (sin)
            BFrag brel[2] = {load_b<6>(brow), load_b<7>(brow)};
            pool_family<POOL, 1, 3, 15, 18>(m, c1, s1, brow, babs, acc);
            LF_PT(4);
            // angle + gradient_angle(px) (embedding.glsl:70-72) x polar kernels: unique tiles 6-11
            f32x2 d1[4], e1[4];
            {
                const f32x4 *pp = reinterpret_cast<const f32x4 *>(&s_phi[(g * 32 + 8 * q) * 2]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 t4 = pp[e];   // cos of phi for pixels 2e, 2e+1, then their sines
                    const f32x2 cp = {t4[0], t4[1]}, sp = {t4[2], t4[3]};
                    d1[e] = pk_fma(c1[e], cp, -(s1[e] * sp));
                    e1[e] = pk_fma(s1[e], cp, c1[e] * sp);
                }
            }
            pool_family<POOL, 2, 6, 3, 9>(m, d1, e1, brow, brel, acc);
            LF_PT(5);
        };
        patch_row(std::integral_constant<int, 0>(), 0);
#pragma unroll 1
        for (int g = 1; g < 31; ++g) patch_row(std::integral_constant<int, 1>(), g);
        patch_row(std::integral_constant<int, 2>(), 31);
        // launder the (uniform) table pointers once per batch: otherwise hipcc hoists one 64-bit VGPR address per
        // whitening-fragment load out of the batch loop and spills 1.4 KB of them per lane
        const unsigned char *wf = wfrag;
        const float *bs = bias;
        asm volatile("" : "+s"(wf), "+s"(bs));
#ifdef LF_ABLATE_EPILOGUE  // timing-only build
        { f32x4 sum = acc[0]; for (int t = 1; t < kTiles; ++t) sum += acc[t];
          if (base + p < n) *reinterpret_cast<f32x4 *>(out + (base + p) * 128 + 4 * q) = sum;
          if (POOL == LF_POOL_F16X3) { __syncthreads(); if (more) issue_lut_row<W>(lut_rows, 0, s_mem, wave, lane); } }
#else
        finish_descriptors<POOL, W>(acc, lane, wave, base + p < n, base + p, colmap, wf, bs, out, raw_out, s_mem, lut_rows,
                                    more);
#endif
        LF_PT(6);
    }
#ifdef LF_PHASE_TIMING
    __syncthreads();
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) out[wave * 128 + i] = (float)pt[i];
#endif
}

#pragma clang fp contract(fast)   // hipcc's default, for what follows (the detector kernels set their own)

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

// ---------------------------------------------------------------------------------------------
// Keypoint orientation (keypoint_orientation.glsl:36-171).  One wave per extremum, 4 per block.
// The 15x15 window (taps `step` texels apart on a-trous layer `level`) is staged in LDS; every lane owns up to
// four of its 225 texels.  The reference sums the histogram with ONE thread walking the window row-major
// (lines 114-124); here lane b sums bin b in that same order, so every bin sees the same addition sequence.
// Contraction is off in this kernel: peak decisions compare sums, and the restatement they are tested against
// (oracle/mkd_oracle.c) is built without fma contraction.
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float glsl_sign(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// shaders/atan2.glsl:19-46, the angle of point (x, y); sign(0) = 0 makes atan2(0, y != 0) return 0.
__device__ __forceinline__ float atan2_shader(float x, float y) {
#pragma clang fp contract(off)
    if (x == 0.f && y == 0.f) return 0.f;
    const bool swap = fabsf(x) < fabsf(y);
    const float a = swap ? (x / y) : (y / x);
    const float s = a * a;
    const float p = a * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f +
                    s * (0.05265332f + s * -0.0117212f)))));
    const float res = swap ? (1.5707964f * glsl_sign(a) - p) : p;
    if (x < 0.f) return 3.1415927f * (y < 0.f ? -1.f : 1.f) + res;
    return res;
}

constexpr int kOriWin = 15, kOriPx = kOriWin * kOriWin, kOriBins = 36, kOriMaxPeaks = 18;

}  // namespace

__global__ __launch_bounds__(256) void orient_peaks(const float *__restrict__ layer0, long layer0_stride,
                                                    const float *__restrict__ coarse, long coarse_stride,
                                                    long layer_stride, int n_layers, int w, int h,
                                                    const float *__restrict__ extrema /*[n][4]*/,
                                                    const unsigned *__restrict__ frame_of, long n_host,
                                                    const unsigned long long *__restrict__ n_dev,
                                                    float *__restrict__ angles /*[n][18]*/,
                                                    unsigned *__restrict__ counts /*[n]*/) {
#pragma clang fp contract(off)
    const long n = n_dev ? (long)*n_dev : n_host;
    if (n <= 0) return;   // uniform: no barrier is skipped by part of a block
    __shared__ float s_patch[4][kOriPx];
    __shared__ float s_weight[4][kOriPx];
    __shared__ int s_bin[4][kOriPx];
    __shared__ float s_hist[4][kOriBins];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long kk = (long)blockIdx.x * 4 + wave;
    const bool live = kk < n;
    const long k = live ? kk : n - 1;   // idle waves redo the last extremum so that the barriers stay uniform
    const float *ex = extrema + k * 4;
    const int kx = (int)ex[0], ky = (int)ex[1];
    const float size = ex[2];
    const float kSigmaRadius = 1.41421356237f;   // DOG_SIGMA_RADIUS_FACTOR = sqrt(2), DOG_FIRST_SCALE_SIGMA = 0.82
    int level = (int)roundf(log2f(size / (0.82f * kSigmaRadius)));
    level = level < 0 ? 0 : (level > n_layers - 1 ? n_layers - 1 : level);
    const int step = 1 << level;
    const int radius = (int)roundf(3.f * 1.5f * size / kSigmaRadius);
    const float sigma = 1.5f * size / kSigmaRadius;
    const unsigned f = frame_of ? frame_of[k] : 0u;
    const float *img = level == 0 ? layer0 + f * layer0_stride : coarse + f * coarse_stride + (level - 1) * layer_stride;

    float *patch = s_patch[wave], *weight = s_weight[wave];
    int *bin = s_bin[wave];
    float *hist = s_hist[wave];
    unsigned ingrad = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (i < kOriPx) {
            const int ly = i / kOriWin, lx = i - ly * kOriWin;
            const int xd = (lx - 7) * step, yd = (ly - 7) * step;
            const int xi = kx + xd, yi = ky + yd;
            // valid_px admits y == height (line 73); that row reads as 0 like every out-of-image load
            const bool valid = 0 <= xi && xi < w && 0 <= yi && yi <= h;
            patch[i] = (valid && yi < h) ? img[(size_t)yi * w + xi] : 0.f;
            const bool inner = lx > 0 && lx < kOriWin - 1 && ly > 0 && ly < kOriWin - 1;
            if (valid && inner && abs(xd) <= radius && abs(yd) <= radius) ingrad |= 1u << j;
        }
    }
    __syncthreads();
    int voters = 0;   // uniform over the wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        {
            int b = kOriBins + 1;
            float wgt = 0.f;
            if (i < kOriPx && (ingrad & (1u << j))) {
                const float gx = patch[i + 1] - patch[i - 1];
                const float gy = patch[i - kOriWin] - patch[i + kOriWin];
                if (gx != 0.f || gy != 0.f) {
                    const int ly = i / kOriWin, lx = i - ly * kOriWin;
                    const float fx = (float)(lx - 7) * (float)step, fy = (float)(ly - 7) * (float)step;
                    const float dist = fx * fx + fy * fy;
                    wgt = expf(-dist / (2.f * sigma * sigma)) * sqrtf(gx * gx + gy * gy);
                    const int rb = (int)roundf(atan2_shader(gx, gy) * ((float)kOriBins / (2.f * 3.1415927f)));
                    b = rb < 0 ? rb + kOriBins : (rb >= kOriBins ? rb - kOriBins : rb);
                }
            }
            // texels that vote (bin < 36) are appended in texel order: chunk j holds texels 64 j .. 64 j + 63, one per
            // lane, so a ballot gives each its rank -- the histogram walk below then skips the texels that do not vote
            const unsigned long long vm = __ballot(b < kOriBins);
            if (b < kOriBins) {
                const int pos = voters + __popcll(vm & ((1ull << lane) - 1ull));
                bin[pos] = b;
                weight[pos] = wgt;
            }
            voters += __popcll(vm);
        }
    }
    __syncthreads();
    float raw = 0.f;
    if (lane < kOriBins)
        for (int i = 0; i < voters; ++i)   // still the reference's single-thread, row-major addition order per bin
            if (bin[i] == lane) raw += weight[i];
    __syncthreads();
    if (lane < kOriBins) hist[lane] = raw;   // raw histogram, circular
    __syncthreads();
    float hv = 0.f;
    if (lane < kOriBins) {
        auto at = [&](int b) { return hist[b < 0 ? b + kOriBins : (b >= kOriBins ? b - kOriBins : b)]; };
        hv = (at(lane - 2) + at(lane + 2)) * (1.0f / 16.0f) + (at(lane - 1) + at(lane + 1)) * (4.0f / 16.0f) +
             at(lane) * (6.0f / 16.0f);
    }
    __syncthreads();
    if (lane < kOriBins) hist[lane] = hv;    // smoothed histogram
    float mx = hv;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();
    bool peak = false;
    float angle = 0.f;
    if (lane < kOriBins) {
        const float left = hist[lane > 0 ? lane - 1 : kOriBins - 1], right = hist[lane < kOriBins - 1 ? lane + 1 : 0];
        peak = left < hv && right < hv && mx * 0.8f <= hv;
        const float interp = (left - right) / (left - 2.f * hv + right);
        const float rbin = (float)lane + interp / 2.0f;
        const float b = rbin < 0.f ? rbin + kOriBins : (rbin > kOriBins ? rbin - kOriBins : rbin);
        angle = 360.0f - (360.0f / (float)kOriBins) * b;
    }
    const unsigned long long m = __ballot(peak);
    if (live) {
        if (peak) angles[k * kOriMaxPeaks + __popcll(m & ((1ull << lane) - 1ull))] = angle;
        if (lane == 0) counts[k] = (unsigned)__popcll(m);
    }
}

// Ordered compaction: keypoint list = for each extremum in index order, its peaks in bin order.  One workgroup
// walks the extrema 1024 at a time with a running offset (n is a few thousand per frame; the scan is not the
// cost).  totals[0] = keypoints written, totals[1] = keypoints dropped because max_out was reached.
__global__ __launch_bounds__(1024) void orient_compact(const float *__restrict__ extrema,
                                                       const unsigned *__restrict__ frame_of,
                                                       const float *__restrict__ angles,
                                                       const unsigned *__restrict__ counts, long n_host,
                                                       const unsigned long long *__restrict__ n_dev,
                                                       float *__restrict__ kps /*[max_out][5]*/,
                                                       unsigned *__restrict__ frame_of_kp, unsigned long long max_out,
                                                       unsigned long long *__restrict__ totals) {
    __shared__ unsigned wave_sum[16];
    const long n = n_dev ? (long)*n_dev : n_host;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long base = 0;
    for (long chunk = 0; chunk < n; chunk += 1024) {
        const long i = chunk + threadIdx.x;
        const unsigned c = i < n ? counts[i] : 0u;
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = wave_sum[v];
            before += v < wave ? t : 0u;
            all += t;
        }
        const unsigned long long first = base + before + incl - c;
        for (unsigned j = 0; j < c; ++j) {
            const unsigned long long o = first + j;
            if (o < max_out) {
                kps[o * 5 + 0] = extrema[i * 4 + 0];
                kps[o * 5 + 1] = extrema[i * 4 + 1];
                kps[o * 5 + 2] = extrema[i * 4 + 2];
                kps[o * 5 + 3] = angles[i * kOriMaxPeaks + j];
                kps[o * 5 + 4] = extrema[i * 4 + 3];
                if (frame_of_kp) frame_of_kp[o] = frame_of ? frame_of[i] : 0u;
            }
        }
        base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = base < max_out ? base : max_out;
        totals[1] = base < max_out ? 0ull : base - max_out;
    }
}

// The same ordered compaction as orient_compact for long lists (batched frames: hundreds of thousands of extrema),
// in the three-launch form of cubes_*: sums of 1024 counts, one workgroup scans the sums, every workgroup rescans its
// counts and writes its keypoints.
__global__ __launch_bounds__(1024) void orient_block_sums(const unsigned *__restrict__ counts, long n_host,
                                                          const unsigned long long *__restrict__ n_dev,
                                                          unsigned *__restrict__ sums) {
    __shared__ unsigned ws[16];
    const long n = n_dev ? (long)*n_dev : n_host;
    const long i = (long)blockIdx.x * 1024 + threadIdx.x;
    unsigned v = i < n ? counts[i] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int k = 0; k < 16; ++k) t += ws[k];
        sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void orient_scatter(const float *__restrict__ extrema,
                                                       const unsigned *__restrict__ frame_of,
                                                       const float *__restrict__ angles,
                                                       const unsigned *__restrict__ counts,
                                                       const unsigned *__restrict__ block_offsets, long n_host,
                                                       const unsigned long long *__restrict__ n_dev,
                                                       float *__restrict__ kps, unsigned *__restrict__ frame_of_kp,
                                                       unsigned long long max_out) {
    __shared__ unsigned ws[16];
    const long n = n_dev ? (long)*n_dev : n_host;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 1024 + threadIdx.x;
    const unsigned c = i < n ? counts[i] : 0u;
    unsigned incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    unsigned before = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) before += v < wave ? ws[v] : 0u;
    const unsigned long long first = (unsigned long long)block_offsets[blockIdx.x] + before + incl - c;
    for (unsigned j = 0; j < c; ++j) {
        const unsigned long long o = first + j;
        if (o < max_out) {
            kps[o * 5 + 0] = extrema[i * 4 + 0];
            kps[o * 5 + 1] = extrema[i * 4 + 1];
            kps[o * 5 + 2] = extrema[i * 4 + 2];
            kps[o * 5 + 3] = angles[i * kOriMaxPeaks + j];
            kps[o * 5 + 4] = extrema[i * 4 + 3];
            if (frame_of_kp) frame_of_kp[o] = frame_of ? frame_of[i] : 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Detector: DoG + 3-D extremum scan + quadratic refinement + edge test (swt_sub.glsl:17-30,
// scan_extrema.glsl:36-241).  The reference works in 4x4x4 cubes with at most 8 candidates each; a cube is
// exactly one wavefront here (lane = x + 4 y + 16 z), so "which 8" and the order of the survivors are settled
// by ballots in lane order instead of atomics.  The DoG is never written to HBM: a workgroup differences the
// a-trous layers into an LDS tile (the subtraction is the same single f32 operation either way).
// Output per cube: count + up to 8 slots {x, y, size, contrast}; cubes_compact_* turn that into the ordered list.
// ---------------------------------------------------------------------------------------------
// Workgroup = a 32 x 8 pixel tile (8 x 2 cubes per cube layer): the tile's DoG volume (plus a one-texel rim) is
// differenced into LDS once with row-contiguous loads, then each of the 4 waves walks its share of the cubes.
constexpr int kScanTX = 32, kScanTY = 8, kScanMaxFine = 8;
constexpr int kScanRowLen = kScanTX + 2, kScanPlane = (kScanTY + 2) * kScanRowLen;

__global__ __launch_bounds__(256) void scan_extrema(const float *__restrict__ layer0, long layer0_stride,
                                                    const float *__restrict__ coarse, long coarse_stride,
                                                    long layer_stride, int n_fine, int w, int h, int border,
                                                    int skip_layers, float contrast_threshold, int gx, int gy, int gz,
                                                    float *__restrict__ slots /*[frames*cubes][8][4]*/,
                                                    unsigned *__restrict__ counts /*[frames*cubes]*/) {
#pragma clang fp contract(off)
    __shared__ float s_dog[kScanMaxFine * kScanPlane];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned f = blockIdx.z;
    const int tx0 = blockIdx.x * kScanTX + border, ty0 = blockIdx.y * kScanTY + border;   // first candidate texel
    const float *l0 = layer0 + f * layer0_stride, *cs = coarse + f * coarse_stride;
    // fine[z] = coarse[z] - coarse[z+1] (swt_sub.glsl:24-29) for the tile and its rim; outside the frame: 0
    for (int i = threadIdx.x; i < kScanPlane; i += 256) {
        const int yy = i / kScanRowLen, xx = i - yy * kScanRowLen;
        const int x = tx0 - 1 + xx, y = ty0 - 1 + yy;
        const bool in = x >= 0 && x < w && y >= 0 && y < h;
        const size_t o = in ? (size_t)y * w + x : 0;
        float c[kScanMaxFine + 1];   // all layers of this texel requested at once
        c[0] = in ? l0[o] : 0.f;
#pragma unroll
        for (int z = 0; z < kScanMaxFine; ++z) c[z + 1] = (in && z < n_fine) ? cs[(size_t)z * layer_stride + o] : 0.f;
#pragma unroll
        for (int z = 0; z < kScanMaxFine; ++z)
            if (z < n_fine) s_dog[z * kScanPlane + i] = c[z] - c[z + 1];
    }
    __syncthreads();
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    const int b1 = border > 1 ? border : 1;
    const int ncubes = gx * gy * gz;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int q = wave; q < 16 * gz; q += 4) {   // cube q of this tile: (cube layer, cube row, cube column)
        const int qz = q >> 4, qy = (q >> 3) & 1, qx = q & 7;
        const int cx = blockIdx.x * 8 + qx, cy = blockIdx.y * 2 + qy;
        if (cx >= gx || cy >= gy) continue;      // uniform per wave
        const int x = tx0 + qx * 4 + lx, y = ty0 + qy * 4 + ly, z = qz * 4 + lz + 1 + skip_layers;
        const bool inside = !(x < b1 || x >= w - b1 || y < b1 || y >= h - b1 || z <= 0 || z >= n_fine - 1);
        // LDS index of this voxel; out-of-range lanes are parked on a valid interior voxel and never become candidates
        const int c = (inside ? z : 1) * kScanPlane + (qy * 4 + ly + 1) * kScanRowLen + (qx * 4 + lx + 1);
        auto at = [&](int dz, int dy, int dx) { return s_dog[c + dz * kScanPlane + dy * kScanRowLen + dx]; };
        const float val = s_dog[c];
        // sign(val) val >= sign(val) neighbour for all 26 neighbours (lines 97-124)  <=>  val >= their maximum when
        // val > 0, val <= their minimum when val < 0 (multiplying by +-1 is exact): 13 max3 + 13 min3 instead of 26
        // multiply-compare-and chains, and no divergence
        float nmax = -INFINITY, nmin = INFINITY;
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
                    if (dz || dy || dx) {
                        const float v = at(dz, dy, dx);
                        nmax = fmaxf(nmax, v);
                        nmin = fminf(nmin, v);
                    }
        const bool cand = inside && fabsf(val) > contrast_threshold && (val > 0.f ? val >= nmax : val <= nmin);
        const unsigned long long cm = __ballot(cand);
        bool emit = false;
        float ox = 0.f, oy = 0.f, size = 0.f, contrast = 0.f;
        if (cand && __popcll(cm & below) < 8) {   // max_wg_extrema = 8 (scan_extrema.glsl:28)
            const float dds = (at(1, 0, 0) - at(-1, 0, 0)) / 2.0f;
            const float ddy = (at(0, 1, 0) - at(0, -1, 0)) / 2.0f;
            const float ddx = (at(0, 0, 1) - at(0, 0, -1)) / 2.0f;
            const float value2x = val * 2.0f;
            const float h11 = at(1, 0, 0) + at(-1, 0, 0) - value2x;
            const float h22 = at(0, 1, 0) + at(0, -1, 0) - value2x;
            const float h33 = at(0, 0, 1) + at(0, 0, -1) - value2x;
            const float h12 = (at(1, 1, 0) - at(-1, 1, 0) - at(1, -1, 0) + at(-1, -1, 0)) / 4.0f;
            const float h13 = (at(1, 0, 1) - at(-1, 0, 1) - at(1, 0, -1) + at(-1, 0, -1)) / 4.0f;
            const float h23 = (at(0, 1, 1) - at(0, 1, -1) - at(0, -1, 1) + at(0, -1, -1)) / 4.0f;
            const float det =
                h11 * h22 * h33 - h11 * h23 * h23 - h12 * h12 * h33 + 2.f * h12 * h13 * h23 - h13 * h13 * h22;
            const float hinv11 = (h22 * h33 - h23 * h23) / det;
            const float hinv12 = (h13 * h23 - h12 * h33) / det;
            const float hinv13 = (h12 * h23 - h13 * h22) / det;
            const float hinv22 = (h11 * h33 - h13 * h13) / det;
            const float hinv23 = (h12 * h13 - h11 * h23) / det;
            const float hinv33 = (h11 * h22 - h12 * h12) / det;
            const float os = -(hinv11 * dds + hinv12 * ddy + hinv13 * ddx);
            oy = -(hinv12 * dds + hinv22 * ddy + hinv23 * ddx);
            ox = -(hinv13 * dds + hinv23 * ddy + hinv33 * ddx);
            // |offset| > 0.5 in any direction: the shader moves x, y, z and emits nothing (lines 200-203).
            // A singular hessian gives NaN offsets; the shader would emit NaN coordinates, here it is dropped.
            const bool within = fabsf(ox) <= 0.5f && fabsf(oy) <= 0.5f && fabsf(os) <= 0.5f;
            const float interp = os * dds + oy * ddy + ox * ddx;
            contrast = fabsf(val + interp / 2.0f);
            const float denom = (h22 + h33) * (h22 + h33);
            const float cmv = 1.f - 4.f * (h22 * h33 - h23 * h23) / denom;
            emit = within && denom != 0.f && !(0.7f <= cmv && cmv <= 1.5f);
            size = 0.82f * 1.41421356237f * exp2f((float)z + os);
        }
        const unsigned long long em = __ballot(emit);
        const size_t g = (size_t)f * ncubes + ((size_t)qz * gy + cy) * gx + cx;
        if (emit) {
            float *o = slots + (g * 8 + __popcll(em & below)) * 4;
            o[0] = (float)x + ox;
            o[1] = (float)y + oy;
            o[2] = size;
            o[3] = contrast;
        }
        if (lane == 0) counts[g] = (unsigned)__popcll(em);
    }
}

// Ordered compaction of per-cube slots, three small launches: (1) sums of 1024 counts, (2) one workgroup scans the
// sums, (3) every workgroup rescans its 1024 counts from its base and copies the slots.  Item i belongs to frame
// i / items_per_frame; frame_start[f] (optional) receives the offset of the frame's first extremum.
__global__ __launch_bounds__(1024) void cubes_block_sums(const unsigned *__restrict__ counts, long n,
                                                         unsigned *__restrict__ sums) {
    __shared__ unsigned ws[16];
    const long i = (long)blockIdx.x * 1024 + threadIdx.x;
    unsigned v = i < n ? counts[i] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int k = 0; k < 16; ++k) t += ws[k];
        sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void cubes_scan_sums(unsigned *__restrict__ sums, long nb, unsigned long long max_out,
                                                        unsigned long long *__restrict__ totals) {
    __shared__ unsigned ws[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long base = 0;
    for (long chunk = 0; chunk < nb; chunk += 1024) {
        const long i = chunk + threadIdx.x;
        const unsigned c = i < nb ? sums[i] : 0u;
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = ws[v];
            before += v < wave ? t : 0u;
            all += t;
        }
        if (i < nb) sums[i] = (unsigned)(base + before + incl - c);   // exclusive offset of the block
        base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = base < max_out ? base : max_out;
        totals[1] = base < max_out ? 0ull : base - max_out;
    }
}

__global__ __launch_bounds__(1024) void cubes_scatter(const unsigned *__restrict__ counts,
                                                      const float *__restrict__ slots,
                                                      const unsigned *__restrict__ block_offsets, long n,
                                                      long items_per_frame, float *__restrict__ out /*[max_out][4]*/,
                                                      unsigned *__restrict__ frame_of, unsigned *__restrict__ frame_start,
                                                      unsigned long long max_out) {
    __shared__ unsigned ws[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 1024 + threadIdx.x;
    const unsigned c = i < n ? counts[i] : 0u;
    unsigned incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    unsigned before = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) before += v < wave ? ws[v] : 0u;
    const unsigned long long first = (unsigned long long)block_offsets[blockIdx.x] + before + incl - c;
    if (i >= n) return;
    const unsigned f = (unsigned)(i / items_per_frame);
    if (frame_start && i == (long)f * items_per_frame) frame_start[f] = (unsigned)(first < max_out ? first : max_out);
    for (unsigned j = 0; j < c; ++j) {
        const unsigned long long o = first + j;
        if (o < max_out) {
            *reinterpret_cast<f32x4 *>(out + o * 4) = *reinterpret_cast<const f32x4 *>(slots + ((size_t)i * 8 + j) * 4);
            if (frame_of) frame_of[o] = f;
        }
    }
}

// TopKContrastFilter::filter (vulkan/mod.rs:1753-1786) for one frame's extrema [n][4], one workgroup per frame:
// keep blobs with size >= min_size; if more than n_keep remain, find the (n_keep+1)-th largest contrast (radix
// select on the float bits, contrast >= 0) and keep, in index order, the first n_keep blobs that reach it.
// seg_start[f], seg_start[f+1] delimit frame f (seg_start == nullptr: one segment [0, *n_in)).
// out: gathered extrema from out_base(f) = f * n_keep; out_count[f].
__global__ __launch_bounds__(1024) void topk_filter(const float *__restrict__ extrema, const unsigned *__restrict__ seg_start,
                                                    const unsigned long long *__restrict__ n_in,
                                                    unsigned long long n_host, unsigned n_frames, unsigned seg_cap,
                                                    unsigned n_keep, float min_size, float *__restrict__ out,
                                                    unsigned *__restrict__ out_index, unsigned *__restrict__ out_count,
                                                    unsigned long long *__restrict__ out_count64) {
    __shared__ unsigned hist[256];
    __shared__ unsigned ws[16];
    __shared__ unsigned sh_prefix, sh_rank, sh_m;
    const unsigned f = blockIdx.x;
    const unsigned total = (unsigned)(n_in ? n_in[0] : n_host);   // count on the device, or given by the host
    unsigned lo = seg_start ? seg_start[f] : 0u;
    unsigned hi = seg_start ? (f + 1 < n_frames ? seg_start[f + 1] : total) : total;
    lo = lo < total ? lo : total;
    hi = hi < total ? hi : total;
    hi = hi - lo > seg_cap ? lo + seg_cap : hi;   // a frame's extrema beyond max_extrema are dropped (mod.rs:627)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // One sweep over the segment, four extrema per thread and step so that their loads are in flight together (the
    // segment is a few thousand to a few ten thousand entries: the sweeps are latency-, not bandwidth-bound).
    // fn(index, passes min_size, key = float bits of |contrast|)
    auto sweep = [&](auto fn) {
        for (unsigned i0 = lo + threadIdx.x; i0 < hi; i0 += 4096) {
            float sz[4], ct[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned i = i0 + 1024u * j;
                const bool in = i < hi;
                sz[j] = in ? extrema[(size_t)i * 4 + 2] : 0.f;
                ct[j] = in ? extrema[(size_t)i * 4 + 3] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned i = i0 + 1024u * j;
                if (i < hi) fn(i, sz[j] >= min_size, __float_as_uint(fabsf(ct[j])));
            }
        }
    };
    // how many pass min_size
    if (threadIdx.x == 0) sh_m = 0;
    __syncthreads();
    unsigned mine = 0;
    sweep([&](unsigned, bool pass, unsigned) { mine += pass ? 1u : 0u; });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) atomicAdd(&sh_m, mine);
    __syncthreads();
    const unsigned m = sh_m;
    unsigned cutoff = 0;   // key threshold; 0 keeps everything that passes
    if (m > n_keep) {
        // radix select, most significant byte first: rank n_keep (0-based) in descending key order
        unsigned prefix = 0, rank = n_keep;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int b = threadIdx.x; b < 256; b += 1024) hist[b] = 0;
            __syncthreads();
            const unsigned mask = shift == 24 ? 0u : 0xFFFFFFFFu << (shift + 8);
            sweep([&](unsigned, bool pass, unsigned k) {
                if (pass && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            });
            __syncthreads();
            // the bin holding the wanted rank, walking down from byte value 255: threads 0..255 take bins 255..0,
            // prefix sums over them locate it in parallel
            {
                const unsigned c = threadIdx.x < 256 ? hist[255 - threadIdx.x] : 0u;   // waves 4..15 carry zeros
                unsigned incl = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                if (lane == 63) ws[wave] = incl;
                __syncthreads();
                unsigned before = 0;
#pragma unroll
                for (int v = 0; v < 16; ++v) before += v < wave ? ws[v] : 0u;
                const unsigned excl = before + incl - c;
                if (c != 0 && rank >= excl && rank < excl + c) {
                    sh_prefix = prefix | ((255u - threadIdx.x) << shift);
                    sh_rank = rank - excl;
                }
            }
            __syncthreads();
            prefix = sh_prefix;
            rank = sh_rank;
            __syncthreads();
        }
        cutoff = prefix;
    }
    // ordered compaction of {passes && key >= cutoff}, first n_keep
    unsigned base = 0;
    for (unsigned chunk = lo; chunk < hi && base < n_keep; chunk += 1024) {
        const unsigned i = chunk + threadIdx.x;
        const bool take = i < hi && extrema[(size_t)i * 4 + 2] >= min_size &&
                          __float_as_uint(fabsf(extrema[(size_t)i * 4 + 3])) >= cutoff;
        const unsigned long long bm = __ballot(take);
        if (lane == 0) ws[wave] = (unsigned)__popcll(bm);
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = ws[v];
            before += v < wave ? t : 0u;
            all += t;
        }
        const unsigned o = base + before + (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
        if (take && o < n_keep) {
            const size_t dst = (size_t)f * n_keep + o;
            *reinterpret_cast<f32x4 *>(out + dst * 4) = *reinterpret_cast<const f32x4 *>(extrema + (size_t)i * 4);
            if (out_index) out_index[dst] = i - lo;
        }
        base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_count[f] = base < n_keep ? base : n_keep;
        if (out_count64 && f == 0) out_count64[0] = base < n_keep ? base : n_keep;
    }
}

// The same selection for ONE long list (a 4K frame yields tens of thousands of extrema), spread over the chip: per
// radix byte one histogram launch over all workgroups and one 256-thread launch that picks the bin, then the ordered
// compaction in the three-launch form.  work (u32): [0,256) histogram, [256] prefix, [257] rank, [258] done,
// [259] cutoff key, [264,268) two u64 totals, [272...) sums of the compaction.
constexpr int kTopkState = 256, kTopkTotals = 264, kTopkSums = 272;

__global__ void topk_init(unsigned *__restrict__ work, unsigned n_keep) {
    work[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        work[kTopkState + 0] = 0;        // prefix
        work[kTopkState + 1] = n_keep;   // rank wanted (0-based, descending)
        work[kTopkState + 2] = 0;        // done: every blob that passes min_size is kept
        work[kTopkState + 3] = 0;        // cutoff key
    }
}

__global__ __launch_bounds__(1024) void topk_hist(const float *__restrict__ extrema, const unsigned long long *__restrict__ n_in,
                                                  unsigned long long n_host, float min_size, int shift,
                                                  unsigned *__restrict__ work) {
    __shared__ unsigned lh[16][256];   // one histogram per wave: lanes of different waves never collide
    if (work[kTopkState + 2]) return;
    const unsigned n = (unsigned)(n_in ? n_in[0] : n_host);
    const unsigned prefix = work[kTopkState + 0];
    const unsigned mask = shift == 24 ? 0u : 0xFFFFFFFFu << (shift + 8);
    const int wave = threadIdx.x >> 6;
    for (int b = threadIdx.x; b < 16 * 256; b += 1024) (&lh[0][0])[b] = 0;
    __syncthreads();
    float sz[4], ct[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned i = blockIdx.x * 4096u + threadIdx.x + 1024u * j;
        sz[j] = i < n ? extrema[(size_t)i * 4 + 2] : 0.f;
        ct[j] = i < n ? extrema[(size_t)i * 4 + 3] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned i = blockIdx.x * 4096u + threadIdx.x + 1024u * j;
        const unsigned k = __float_as_uint(fabsf(ct[j]));
        if (i < n && sz[j] >= min_size && (k & mask) == prefix) atomicAdd(&lh[wave][(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        unsigned t = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) t += lh[v][threadIdx.x];
        if (t) atomicAdd(&work[threadIdx.x], t);
    }
}

__global__ __launch_bounds__(256) void topk_pick(unsigned *__restrict__ work, unsigned n_keep, int shift) {
    __shared__ unsigned ws[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool done = work[kTopkState + 2] != 0;
    const unsigned prefix = work[kTopkState + 0], rank = work[kTopkState + 1];
    const unsigned c = work[255 - threadIdx.x];   // thread t takes bin 255 - t: prefix sums walk down from the top
    unsigned incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        before += v < wave ? ws[v] : 0u;
        all += ws[v];
    }
    const unsigned excl = before + incl - c;
    work[threadIdx.x] = 0;   // histogram ready for the next byte
    if (done) return;
    if (shift == 24 && all <= n_keep) {   // the first histogram counts everything that passes min_size
        if (threadIdx.x == 0) { work[kTopkState + 2] = 1; work[kTopkState + 3] = 0; }
        return;
    }
    if (c != 0 && rank >= excl && rank < excl + c) {
        const unsigned np = prefix | ((255u - threadIdx.x) << shift);
        work[kTopkState + 0] = np;
        work[kTopkState + 1] = rank - excl;
        if (shift == 0) work[kTopkState + 3] = np;
    }
}

// take flags of the compaction: passes min_size and reaches the cutoff key
__device__ __forceinline__ bool topk_take(const float *__restrict__ extrema, unsigned i, unsigned n, float min_size,
                                          unsigned cutoff) {
    return i < n && extrema[(size_t)i * 4 + 2] >= min_size && __float_as_uint(fabsf(extrema[(size_t)i * 4 + 3])) >= cutoff;
}

__global__ __launch_bounds__(1024) void topk_sums(const float *__restrict__ extrema, const unsigned long long *__restrict__ n_in,
                                                  unsigned long long n_host, float min_size, unsigned *__restrict__ work) {
    __shared__ unsigned ws[16];
    const unsigned n = (unsigned)(n_in ? n_in[0] : n_host);
    const bool take = topk_take(extrema, blockIdx.x * 1024u + threadIdx.x, n, min_size, work[kTopkState + 3]);
    const unsigned long long bm = __ballot(take);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = (unsigned)__popcll(bm);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int v = 0; v < 16; ++v) t += ws[v];
        work[kTopkSums + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void topk_scatter(const float *__restrict__ extrema, const unsigned long long *__restrict__ n_in,
                                                     unsigned long long n_host, float min_size, unsigned n_keep,
                                                     const unsigned *__restrict__ work, float *__restrict__ out,
                                                     unsigned *__restrict__ out_index, unsigned *__restrict__ out_count,
                                                     unsigned long long *__restrict__ out_count64) {
    __shared__ unsigned ws[16];
    const unsigned n = (unsigned)(n_in ? n_in[0] : n_host);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned i = blockIdx.x * 1024u + threadIdx.x;
    const bool take = topk_take(extrema, i, n, min_size, work[kTopkState + 3]);
    const unsigned long long bm = __ballot(take);
    if (lane == 0) ws[wave] = (unsigned)__popcll(bm);
    __syncthreads();
    unsigned before = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) before += v < wave ? ws[v] : 0u;
    const unsigned o = work[kTopkSums + blockIdx.x] + before + (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
    if (take && o < n_keep) {
        *reinterpret_cast<f32x4 *>(out + (size_t)o * 4) = *reinterpret_cast<const f32x4 *>(extrema + (size_t)i * 4);
        if (out_index) out_index[o] = i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long kept = *reinterpret_cast<const unsigned long long *>(work + kTopkTotals);
        out_count[0] = (unsigned)kept;
        if (out_count64) out_count64[0] = kept;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// n_dev != nullptr: the patch count is read on the device (<= n, which then only sizes the grid)
void launch_describe(const float *patches, long n, const unsigned long long *n_dev, const DeviceConsts &dc,
                     int angle_mode, int pool_mode, float *out, float *raw_out, int num_cus, hipStream_t stream) {
    if (n <= 0) return;
    // one 100-152 KiB-LDS workgroup per CU; requests of at most one round of 64-patch workgroups take the 4-wave form
#ifdef LF_ABLATE_FORCE_W4   // timing-only build: the 4-wave form at every size
    const bool small = true;
#else
    const bool small = n <= 64L * num_cus;
#endif
    const long nbatch = small ? (n + 63) / 64 : (n + 127) / 128;
    const unsigned grid = (unsigned)(nbatch < num_cus ? nbatch : num_cus);
    const bool f16 = pool_mode == LF_POOL_F16X3;
    const unsigned char *lut = f16 ? reinterpret_cast<const unsigned char *>(dc.pool_b_f16)
                                   : reinterpret_cast<const unsigned char *>(dc.pool_b_f32);
    const unsigned char *wf = f16 ? reinterpret_cast<const unsigned char *>(dc.white_a_f16)
                                  : reinterpret_cast<const unsigned char *>(dc.white_a_f32);
#define LF_LAUNCH_W(A, P, WV)                                                                                          \
    hipLaunchKernelGGL((mkd_pool<A, P, WV>), dim3(grid), dim3(64 * WV), 0, stream, patches, n, n_dev, lut, dc.phi_cs, \
                       dc.colmap, wf, dc.white_bias, out, raw_out)
#define LF_LAUNCH(A, P)            \
    do {                           \
        if (small) LF_LAUNCH_W(A, P, 4); \
        else LF_LAUNCH_W(A, P, 8); \
    } while (0)
    if (f16) {
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

void launch_sample_patches(const float *pyr, long pyr_stride, const PyramidDesc &pd, const float *kps,
                           const unsigned *frame_of_kp, long n, const unsigned long long *n_dev, float psf,
                           float *patches, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(sample_patches, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, pyr, pyr_stride, pd, kps,
                       frame_of_kp, n, n_dev, psf, patches);
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
                          hipStream_t stream) {
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

void launch_orient(const float *layer0, long layer0_stride, const float *coarse, long coarse_stride, long layer_stride,
                   int n_layers, int w, int h, const float *extrema, const unsigned *frame_of, long n,
                   const unsigned long long *n_dev, float *angles, unsigned *counts, unsigned *sums, float *kps,
                   unsigned *frame_of_kp, unsigned long long max_out, unsigned long long *totals, hipStream_t stream) {
    if (n > 0)
        hipLaunchKernelGGL(orient_peaks, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, layer0, layer0_stride,
                           coarse, coarse_stride, layer_stride, n_layers, w, h, extrema, frame_of, n, n_dev, angles,
                           counts);
    if (n <= 8192 || !sums) {   // a few thousand extrema (one frame): one workgroup walks them
        hipLaunchKernelGGL(orient_compact, dim3(1), dim3(1024), 0, stream, extrema, frame_of, (const float *)angles,
                           (const unsigned *)counts, n, n_dev, kps, frame_of_kp, max_out, totals);
        return;
    }
    const long nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(orient_block_sums, dim3((unsigned)nb), dim3(1024), 0, stream, (const unsigned *)counts, n, n_dev,
                       sums);
    hipLaunchKernelGGL(cubes_scan_sums, dim3(1), dim3(1024), 0, stream, sums, nb, max_out, totals);
    hipLaunchKernelGGL(orient_scatter, dim3((unsigned)nb), dim3(1024), 0, stream, extrema, frame_of, (const float *)angles,
                       (const unsigned *)counts, (const unsigned *)sums, n, n_dev, kps, frame_of_kp, max_out);
}

void scan_grid(int w, int h, int n_fine, int border, int skip_layers, int &gx, int &gy, int &gz) {
    gx = w - 2 * border > 0 ? (w - 2 * border + 3) / 4 : 0;
    gy = h - 2 * border > 0 ? (h - 2 * border + 3) / 4 : 0;
    gz = n_fine - 2 - skip_layers > 0 ? (n_fine - 2 - skip_layers + 3) / 4 : 0;
}

// a-trous stack -> ordered extrema of `frames` frames.  slots/counts/sums are scratch sized for frames x cubes.
void launch_detect_extrema(const float *layer0, long layer0_stride, const float *coarse, long coarse_stride,
                           long layer_stride, int n_layers, int w, int h, int frames, int border, int skip_layers,
                           float contrast_threshold, float *slots, unsigned *counts, unsigned *sums, float *extrema,
                           unsigned *frame_of, unsigned *frame_start, unsigned long long max_out,
                           unsigned long long *totals, hipStream_t stream) {
    int gx, gy, gz;
    scan_grid(w, h, n_layers - 1, border, skip_layers, gx, gy, gz);
    const long ncubes = (long)gx * gy * gz, n = ncubes * frames;
    const long nb = (n + 1023) / 1024;
    if (ncubes > 0) {
        hipLaunchKernelGGL(scan_extrema, dim3((gx + 7) / 8, (gy + 1) / 2, frames), dim3(256), 0, stream, layer0,
                           layer0_stride, coarse, coarse_stride, layer_stride, n_layers - 1, w, h, border, skip_layers,
                           contrast_threshold, gx, gy, gz, slots, counts);
        hipLaunchKernelGGL(cubes_block_sums, dim3((unsigned)nb), dim3(1024), 0, stream, (const unsigned *)counts, n, sums);
    }
    hipLaunchKernelGGL(cubes_scan_sums, dim3(1), dim3(1024), 0, stream, sums, nb, max_out, totals);
    if (ncubes > 0)
        hipLaunchKernelGGL(cubes_scatter, dim3((unsigned)nb), dim3(1024), 0, stream, (const unsigned *)counts,
                           (const float *)slots, (const unsigned *)sums, n, ncubes, extrema, frame_of, frame_start,
                           max_out);
}

void launch_topk_filter(const float *extrema, const unsigned *seg_start, const unsigned long long *n_in,
                        unsigned long long n_host, unsigned n_frames, unsigned seg_cap, unsigned n_keep, float min_size,
                        float *out, unsigned *out_index, unsigned *out_count, unsigned long long *out_count64,
                        unsigned long long n_cap, unsigned *work, hipStream_t stream) {
    // one frame with a long list and scratch to work in: the multi-workgroup form; otherwise one workgroup per frame
    if (n_frames == 1 && !seg_start && work && n_cap > 8192 && seg_cap >= n_cap) {
        const unsigned nb4 = (unsigned)((n_cap + 4095) / 4096), nb1 = (unsigned)((n_cap + 1023) / 1024);
        hipLaunchKernelGGL(topk_init, dim3(1), dim3(256), 0, stream, work, n_keep);
        for (int shift = 24; shift >= 0; shift -= 8) {
            hipLaunchKernelGGL(topk_hist, dim3(nb4), dim3(1024), 0, stream, extrema, n_in, n_host, min_size, shift, work);
            hipLaunchKernelGGL(topk_pick, dim3(1), dim3(256), 0, stream, work, n_keep, shift);
        }
        hipLaunchKernelGGL(topk_sums, dim3(nb1), dim3(1024), 0, stream, extrema, n_in, n_host, min_size, work);
        hipLaunchKernelGGL(cubes_scan_sums, dim3(1), dim3(1024), 0, stream, work + kTopkSums, (long)nb1,
                           (unsigned long long)n_keep, reinterpret_cast<unsigned long long *>(work + kTopkTotals));
        hipLaunchKernelGGL(topk_scatter, dim3(nb1), dim3(1024), 0, stream, extrema, n_in, n_host, min_size, n_keep,
                           (const unsigned *)work, out, out_index, out_count, out_count64);
        return;
    }
    hipLaunchKernelGGL(topk_filter, dim3(n_frames), dim3(1024), 0, stream, extrema, seg_start, n_in, n_host, n_frames,
                       seg_cap, n_keep, min_size, out, out_index, out_count, out_count64);
}

size_t topk_work_words(unsigned long long n_cap) { return kTopkSums + (size_t)((n_cap + 1023) / 1024) + 2; }

// [frames][n_keep] padded per-frame selections + counts -> one contiguous list with the frame of every entry.
// totals[0] = entries, totals[1] = extrema the per-frame cap dropped (dropped_blobs summed over the frames).
__global__ __launch_bounds__(1024) void segments_offsets(const unsigned *__restrict__ counts, unsigned n_frames,
                                                         const unsigned *__restrict__ seg_start,
                                                         const unsigned long long *__restrict__ n_total, unsigned seg_cap,
                                                         unsigned *__restrict__ offsets,
                                                         unsigned long long *__restrict__ totals) {
    __shared__ unsigned ws[16];
    __shared__ unsigned long long dropped;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) dropped = 0;
    __syncthreads();
    unsigned long long base = 0;
    for (unsigned chunk = 0; chunk < n_frames; chunk += 1024) {
        const unsigned f = chunk + threadIdx.x;
        const unsigned c = f < n_frames ? counts[f] : 0u;
        if (f < n_frames) {
            const unsigned total = (unsigned)n_total[0];
            const unsigned lo = seg_start[f], hi = f + 1 < n_frames ? seg_start[f + 1] : total;
            if (hi - lo > seg_cap) atomicAdd(&dropped, (unsigned long long)(hi - lo - seg_cap));
        }
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = ws[v];
            before += v < wave ? t : 0u;
            all += t;
        }
        if (f < n_frames) offsets[f] = (unsigned)(base + before + incl - c);
        base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = base;
        totals[1] = dropped;
    }
}

__global__ __launch_bounds__(256) void segments_gather(const float *__restrict__ padded, const unsigned *__restrict__ counts,
                                                       const unsigned *__restrict__ offsets, unsigned n_keep,
                                                       float *__restrict__ out, unsigned *__restrict__ frame_of) {
    const unsigned f = blockIdx.x, c = counts[f], o = offsets[f];
    for (unsigned i = threadIdx.x; i < c; i += 256) {
        *reinterpret_cast<f32x4 *>(out + (size_t)(o + i) * 4) =
            *reinterpret_cast<const f32x4 *>(padded + ((size_t)f * n_keep + i) * 4);
        frame_of[o + i] = f;
    }
}

void launch_segments_compact(const float *padded, const unsigned *counts, const unsigned *seg_start,
                             const unsigned long long *n_total, unsigned n_frames, unsigned seg_cap, unsigned n_keep,
                             unsigned *offsets, float *out, unsigned *frame_of, unsigned long long *totals,
                             hipStream_t stream) {
    hipLaunchKernelGGL(segments_offsets, dim3(1), dim3(1024), 0, stream, counts, n_frames, seg_start, n_total, seg_cap,
                       offsets, totals);
    hipLaunchKernelGGL(segments_gather, dim3(n_frames), dim3(256), 0, stream, padded, counts, (const unsigned *)offsets,
                       n_keep, out, frame_of);
}

}  // namespace lfmkd
