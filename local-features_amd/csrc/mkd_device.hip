// MKD descriptor path: device code for gfx950 (CDNA4, wave64).
//
// Kernels (reference stage each one replaces; paths under local_features/src/vulkan/shaders/):
//   mkd_pool_f32      mkd/patch_gradients.glsl:72-104 + mkd/embedding.glsl:53-121 (both variants)
//   mkd_whiten_f32    mkd/normalize.glsl:22-142 + mkd/whitening.glsl:22-77 + mkd/normalize_final.glsl
//   sample_patches    mkd/patch_gradients.glsl:42-70
//   pyr_*             blur.glsl, swt.glsl (level 0), blur_pyramid.glsl, patch_pyramid.rs blits
//
// Pooling is a GEMM with M = patches, K = pixels, N = (stream, spatial kernel) columns.  One wave
// owns 16 patches; lane l = (patch p = l & 15, segment q = l >> 4) holds the 8 pixels
// x in [8q, 8q+8) of the current patch row, which is exactly the A-operand lane map of the
// 16x16 MFMAs (row = l & 15, k-group = l >> 4).  So blur, gradients and the von-Mises
// embedding are computed in the registers that feed the matrix cores; nothing but the final
// sums leaves the wave.  The vertical blur runs as a transposed FIR (the window shifts through
// the FMA destinations), horizontal neighbours come from lanes l -/+ 16 via ds_bpermute.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mkd_device.h"

namespace lfmkd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTiles = 21;
// tile -> A stream (host twin: mkd_consts.hpp kTileStream)
__device__ constexpr int kTileStreamDev[kTiles] = {0, 0, 0, 7, 7, 8, 8, 9, 9, 10, 10,
                                                   11, 11, 12, 12, 1, 2, 3, 4, 5, 6};

// 5-tap sigma=0.7 kernel, patch_gradients.glsl:22-28
constexpr float kB0 = 0.0096f, kB1 = 0.2054f, kB2 = 0.5699f;

__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// cos/sin of the gradient angle theta = -atan2(gy over gx).
template <int ANGLE>
__device__ __forceinline__ void gradient_direction(float gx, float gy, float &ct, float &st) {
    if (ANGLE == LF_ANGLE_EXACT) {
        const float r2 = fmaf(gx, gx, gy * gy);
        const float inv = __builtin_amdgcn_rsqf(r2);
        const bool zero = r2 == 0.f;
        ct = zero ? 1.f : gx * inv;
        st = zero ? 0.f : -gy * inv;
    } else {
        // atan2.glsl:19-46 called as atan2(x = gx, y = gy).  With p = poly(a), |a| <= 1, the
        // branches of the shader are quadrant symmetries of (cos p, sin p):
        //   swap:  res = sign(a) pi/2 - p  -> (cos, sin) = sign(a) (sin p, cos p); a == 0 -> res = 0
        //   x < 0: res += +-pi              -> both negated
        const float ax = fabsf(gx), ay = fabsf(gy);
        const bool swap = ax < ay;
        const float num = swap ? gx : gy, den = swap ? gy : gx;
        const float a = num * __builtin_amdgcn_rcpf(den);
        const float s = a * a;
        const float p = a * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f +
                             s * (0.05265332f + s * -0.0117212f)))));
        const float p2 = p * p;  // |p| <= 0.7854: Taylor to p^9 / p^8 is below 1e-8
        const float sn = p * (1.f + p2 * (-1.6666667e-1f + p2 * (8.3333333e-3f + p2 * (-1.9841270e-4f +
                              p2 * 2.7557319e-6f))));
        const float cs = 1.f + p2 * (-0.5f + p2 * (4.1666667e-2f + p2 * (-1.3888889e-3f + p2 * 2.4801587e-5f)));
        const float sa = a > 0.f ? 1.f : -1.f;
        float cr = swap ? sa * sn : cs;
        float sr = swap ? sa * cs : sn;
        if (swap && a == 0.f) { cr = 1.f; sr = 0.f; }  // the atan2(0, y != 0) == 0 quirk
        if (gx < 0.f) { cr = -cr; sr = -sr; }
        if (ax == 0.f && ay == 0.f) { cr = 1.f; sr = 0.f; }
        ct = cr;   // theta = -res
        st = -sr;
    }
}

// The 13 A-operand values of one pixel (streams: see mkd_consts.hpp).
template <int ANGLE>
__device__ __forceinline__ void pixel_streams(float gx, float gy, float cphi, float sphi, float (&a)[13]) {
    // patch_gradients.glsl:98-100: mag = sqrt(sqrt(gx^2 + gy^2 + eps))
    const float m = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(gx * gx + gy * gy + 1e-8f));
    float c1, s1;
    gradient_direction<ANGLE>(gx, gy, c1, s1);
    const float c2 = c1 * c1 - s1 * s1, s2 = 2.f * c1 * s1;
    const float c3 = c2 * c1 - s2 * s1, s3 = s2 * c1 + c2 * s1;
    // polar variant: angle + gradient_angle(px), embedding.glsl:70-72
    const float d1 = c1 * cphi - s1 * sphi, e1 = s1 * cphi + c1 * sphi;
    const float d2 = d1 * d1 - e1 * e1, e2 = 2.f * d1 * e1;
    const float d3 = d2 * d1 - e2 * e1, e3 = e2 * d1 + d2 * e1;
    a[0] = m;
    a[1] = m * c1; a[2] = m * c2; a[3] = m * c3;
    a[4] = m * s1; a[5] = m * s2; a[6] = m * s3;
    a[7] = m * d1; a[8] = m * d2; a[9] = m * d3;
    a[10] = m * e1; a[11] = m * e2; a[12] = m * e3;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Pooling, f32 MFMA.  grid = ceil(n / 64) blocks of 4 independent waves, 16 patches per wave.
// Algorithmic HBM bytes per patch: 4096 read; 952 written (pooled sums, read back by whitening).
// ---------------------------------------------------------------------------------------------
template <int ANGLE>
__global__ __launch_bounds__(256) void mkd_pool_f32(const float *__restrict__ patches, long n,
                                                    const f32x4 *__restrict__ lut,
                                                    const float *__restrict__ phi_cs,
                                                    const short *__restrict__ colmap,
                                                    float *__restrict__ pooled) {
    __shared__ float s_phi[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s_phi[i] = phi_cs[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (base >= n) return;
    const int p = lane & 15, q = lane >> 4;
    const long pidx = (base + p < n) ? base + p : n - 1;  // tail lanes recompute the last patch
    const float *src = patches + pidx * 1024 + 8 * q;
    const int addr_l = ((lane - 16) & 63) * 4, addr_r = ((lane + 16) & 63) * 4;
    const bool has_l = q > 0, has_r = q < 3;

    f32x4 acc[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a0[8], a1[8], a2[8], a3[8];  // transposed-FIR state of the vertical blur
    float cur[8], prv[8];              // blurred rows g and g-1
    float cur_l = 0.f, cur_r = 0.f;    // blurred row g at x = 8q-1 and 8q+8
#pragma unroll
    for (int x = 0; x < 8; ++x) a0[x] = a1[x] = a2[x] = a3[x] = cur[x] = prv[x] = 0.f;

    // r: raw row fed to the vertical blur (rows -2,-1 and 32,33 replicate the border);
    // v = r - 2: blurred row produced; g = r - 3: gradient row consumed by the matrix cores.
#pragma unroll 2
    for (int r = -2; r <= 34; ++r) {
        float nxt[8], nxt_l = 0.f, nxt_r = 0.f;
        if (r <= 33) {
            const int rr = r < 0 ? 0 : (r > 31 ? 31 : r);
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(src + rr * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(src + rr * 32 + 4);
            const float raw[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            float vb[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {  // patch_gradients.glsl:72-81, k0..k4 in order
                vb[x] = fmaf(kB0, raw[x], a0[x]);
                a0[x] = fmaf(kB1, raw[x], a1[x]);
                a1[x] = fmaf(kB2, raw[x], a2[x]);
                a2[x] = fmaf(kB1, raw[x], a3[x]);
                a3[x] = kB0 * raw[x];
            }
            if (r >= 2) {  // horizontal pass on row v, patch_gradients.glsl:83-92
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
                    nxt[x] = s;
                }
                const float hl = lane_fetch(addr_l, nxt[7]), hr = lane_fetch(addr_r, nxt[0]);
                nxt_l = has_l ? hl : nxt[0];
                nxt_r = has_r ? hr : nxt[7];
            }
        } else {
#pragma unroll
            for (int x = 0; x < 8; ++x) nxt[x] = cur[x];  // row 32 replicates row 31
        }
        if (r < 2) continue;
        if (r == 2) {  // row 0: also stands in for row -1
#pragma unroll
            for (int x = 0; x < 8; ++x) cur[x] = prv[x] = nxt[x];
            cur_l = nxt_l;
            cur_r = nxt_r;
            continue;
        }
        const int g = r - 3;
        // patch_gradients.glsl:94-96: gx = left - right, gy = down - up (replicated border)
        float gx[8], gy[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const float left = x == 0 ? cur_l : cur[x - 1];
            const float right = x == 7 ? cur_r : cur[x + 1];
            gx[x] = left - right;
            gy[x] = nxt[x] - prv[x];
        }
        const f32x4 *ph = reinterpret_cast<const f32x4 *>(&s_phi[(g * 32 + 8 * q) * 2]);
        const f32x4 *lrow = lut + (size_t)g * kTiles * 2 * 64 + lane;
#pragma unroll
        for (int jg = 0; jg < 2; ++jg) {
            float av[4][13];
            const f32x4 pa = ph[2 * jg], pb = ph[2 * jg + 1];
            pixel_streams<ANGLE>(gx[4 * jg + 0], gy[4 * jg + 0], pa[0], pa[1], av[0]);
            pixel_streams<ANGLE>(gx[4 * jg + 1], gy[4 * jg + 1], pa[2], pa[3], av[1]);
            pixel_streams<ANGLE>(gx[4 * jg + 2], gy[4 * jg + 2], pb[0], pb[1], av[2]);
            pixel_streams<ANGLE>(gx[4 * jg + 3], gy[4 * jg + 3], pb[2], pb[3], av[3]);
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                const f32x4 b = lrow[(t * 2 + jg) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e][kTileStreamDev[t]], b[e], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            prv[x] = cur[x];
            cur[x] = nxt[x];
        }
        cur_l = nxt_l;
        cur_r = nxt_r;
    }

    // C layout of the 16x16 MFMA: lane holds column (lane & 15) of rows 4*(lane >> 4) + i.
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        const int d = colmap[t * 16 + p];
        if (d < 0) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = base + 4 * q + i;
            if (row < n) pooled[row * 238 + d] = acc[t][i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Normalise (polar | cartesian | all), subtract mean, whiten 238 -> 128 on f32 MFMA, L2.
// One wave per 16 patches.  normalize.glsl:22-142, whitening.glsl:22-77, normalize_final.glsl.
// raw_out (optional): the 238-D un-whitened descriptor.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mkd_whiten_f32(const float *__restrict__ pooled, long n,
                                                     const float *__restrict__ wfrag,
                                                     const float *__restrict__ mean,
                                                     float *__restrict__ out,
                                                     float *__restrict__ raw_out) {
    __shared__ float s_v[16 * 241];
    const int lane = threadIdx.x;
    const long base = (long)blockIdx.x * 16;
    const int rows = (n - base) < 16 ? int(n - base) : 16;
    for (int i = lane; i < 16 * 238; i += 64) {
        const int pr = i / 238, c = i - pr * 238;
        s_v[pr * 241 + c] = pr < rows ? pooled[base * 238 + i] : 1.f;
    }
    __syncthreads();
    const int p = lane & 15, q = lane >> 4;
    // each lane owns columns q, q+4, ... of patch p: the A-operand map of the 16x16x4 MFMA
    float v[60];
    float sp = 0.f, sc = 0.f;
#pragma unroll
    for (int ks = 0; ks < 60; ++ks) {
        const int c = 4 * ks + q;
        const float x = c < 238 ? s_v[p * 241 + c] : 0.f;
        v[ks] = x;
        if (c < 175) sp = fmaf(x, x, sp);
        else sc = fmaf(x, x, sc);
    }
    sp += __shfl_xor(sp, 16); sp += __shfl_xor(sp, 32);
    sc += __shfl_xor(sc, 16); sc += __shfl_xor(sc, 32);
    const float np = __builtin_amdgcn_sqrtf(sp), nc = __builtin_amdgcn_sqrtf(sc);
    float sa = 0.f;
#pragma unroll
    for (int ks = 0; ks < 60; ++ks) {
        const int c = 4 * ks + q;
        v[ks] = c < 175 ? v[ks] / np : v[ks] / nc;
        sa = fmaf(v[ks], v[ks], sa);
    }
    sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
    const float na = __builtin_amdgcn_sqrtf(sa);
#pragma unroll
    for (int ks = 0; ks < 60; ++ks) {
        const int c = 4 * ks + q;
        const float raw = v[ks] / na;
        if (raw_out && c < 238 && p < rows) raw_out[(base + p) * 238 + c] = raw;
        v[ks] = c < 238 ? raw - mean[c] : 0.f;
    }
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 60; ++ks)
#pragma unroll
        for (int t = 0; t < 8; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[ks], wfrag[(ks * 8 + t) * 64 + lane], acc[t], 0, 0, 0);
    // lane holds out[patch 4q+i][16t + p]; norm over the 128 columns of each patch
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) ss[i] = fmaf(acc[t][i], acc[t][i], ss[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ss[i] += __shfl_xor(ss[i], 1); ss[i] += __shfl_xor(ss[i], 2);
        ss[i] += __shfl_xor(ss[i], 4); ss[i] += __shfl_xor(ss[i], 8);
        ss[i] = __builtin_amdgcn_sqrtf(ss[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pr = 4 * q + i;
        if (pr >= rows) continue;
#pragma unroll
        for (int t = 0; t < 8; ++t) out[(base + pr) * 128 + 16 * t + p] = acc[t][i] / ss[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Keypoint mode: pyramid and sampling.  Sampler = linear filter, MirroredRepeat (mod.rs:940-943),
// restated with exact f32 weights; texel centres at i + 0.5.
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int mirror_idx(int i, int n) {
    const int pp = 2 * n;
    int m = i % pp;
    if (m < 0) m += pp;
    return m < n ? m : pp - 1 - m;
}

__device__ __forceinline__ float tex_bilinear(const float *__restrict__ img, int w, int h, float u, float v) {
    const float fu = u - 0.5f, fv = v - 0.5f;
    const float x0f = floorf(fu), y0f = floorf(fv);
    const float ax = fu - x0f, ay = fv - y0f;
    const int x0 = mirror_idx((int)x0f, w), x1 = mirror_idx((int)x0f + 1, w);
    const int y0 = mirror_idx((int)y0f, h), y1 = mirror_idx((int)y0f + 1, h);
    const float t00 = img[(size_t)y0 * w + x0], t10 = img[(size_t)y0 * w + x1];
    const float t01 = img[(size_t)y1 * w + x0], t11 = img[(size_t)y1 * w + x1];
    const float top = t00 * (1.f - ax) + t10 * ax;
    const float bot = t01 * (1.f - ax) + t11 * ax;
    return top * (1.f - ay) + bot * ay;
}

}  // namespace

// blur.glsl:34-65 (sigma 0.6) and blur_pyramid.glsl horizontal pass share this shape:
// out = w0 * tex(c) + w1 * (tex(c - off) + tex(c + off)) along one axis.
__global__ void pyr_sep3(const float *__restrict__ in, float *__restrict__ out, int w, int h, float w0,
                         float w1, float off, int vertical) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
    const float dx = vertical ? 0.f : off, dy = vertical ? off : 0.f;
    float s = tex_bilinear(in, w, h, cx, cy) * w0;
    s += (tex_bilinear(in, w, h, cx - dx, cy - dy) + tex_bilinear(in, w, h, cx + dx, cy + dy)) * w1;
    out[(size_t)y * w + x] = s;
}

// swt.glsl:24-58 with in_level = 0: [1 4 6 4 1]/16 at texel centres, mirrored.
__global__ void pyr_swt0(const float *__restrict__ in, float *__restrict__ out, int w, int h, int vertical) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float k0 = 6.f / 16.f, k1 = 4.f / 16.f, k2 = 1.f / 16.f;
    float s;
    if (!vertical) {
        const float *row = in + (size_t)y * w;
        s = row[x] * k0;
        s += row[mirror_idx(x - 2, w)] * k2;
        s += row[mirror_idx(x - 1, w)] * k1;
        s += row[mirror_idx(x + 1, w)] * k1;
        s += row[mirror_idx(x + 2, w)] * k2;
    } else {
        s = in[(size_t)y * w + x] * k0;
        s += in[(size_t)mirror_idx(y - 1, h) * w + x] * k1;
        s += in[(size_t)mirror_idx(y - 2, h) * w + x] * k2;
        s += in[(size_t)mirror_idx(y + 2, h) * w + x] * k2;
        s += in[(size_t)mirror_idx(y + 1, h) * w + x] * k1;
    }
    out[(size_t)y * w + x] = s;
}

// Nearest blit [0,w)x[0,h) -> [0,w/2)x[0,h/2): patch_pyramid.rs:251-285.
__global__ void pyr_decimate(const float *__restrict__ in, float *__restrict__ out, int w, int h, int ow, int oh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= ow || y >= oh) return;
    int sx = (int)floorf(((float)x + 0.5f) * (float)w / (float)(w / 2));
    int sy = (int)floorf(((float)y + 0.5f) * (float)h / (float)(h / 2));
    sx = sx > w - 1 ? w - 1 : sx;
    sy = sy > h - 1 ? h - 1 : sy;
    out[(size_t)y * ow + x] = in[(size_t)sy * w + sx];
}

// blur_pyramid.glsl:36-49 vertical pass: binomial taps centred on texel (2x, 2y) of the H result.
__global__ void pyr_down_v(const float *__restrict__ in, float *__restrict__ out, int w, int h, int ow, int oh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= ow || y >= oh) return;
    const float cx = 2.f * (float)x + 0.5f, cy = 2.f * (float)y + 0.5f;
    float s = tex_bilinear(in, w, h, cx, cy) * 0.375f;
    s += (tex_bilinear(in, w, h, cx, cy - 1.2f) + tex_bilinear(in, w, h, cx, cy + 1.2f)) * 0.3125f;
    out[(size_t)y * ow + x] = s;
}

// patch_gradients.glsl:42-70.  One 1024-thread block per keypoint; thread = patch pixel.
__global__ __launch_bounds__(1024) void sample_patches(const float *__restrict__ pyr, PyramidDesc pd,
                                                       const float *__restrict__ kps /*[n][5]*/, long n,
                                                       float psf, float *__restrict__ patches) {
    const long k = blockIdx.x;
    if (k >= n) return;
    const float *kp = kps + k * 5;
    const float scale = kp[2] * psf / 32.f;
    const float l2 = log2f(scale);
    float lvl = floorf(l2);
    lvl = lvl < 0.f ? 0.f : (lvl > (float)(pd.levels - 1) ? (float)(pd.levels - 1) : lvl);
    const float rem = exp2f(l2 - lvl);
    const int l = (int)lvl;
    const float ang = kp[3] * (3.14159265358979323846f / 180.f);
    const float ca = cosf(ang), sa = sinf(ang);
    const float inv = 1.f / exp2f(lvl);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const float dx = (float)lx - 16.f, dy = (float)ly - 16.f;
    const float xx = dx * ca - dy * sa, yy = dx * sa + dy * ca;
    const float sx = xx * rem + kp[0] * inv, sy = yy * rem + kp[1] * inv;
    patches[k * 1024 + threadIdx.x] = tex_bilinear(pyr + pd.offset[l], pd.w[l], pd.h[l], sx + 0.5f, sy + 0.5f);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void launch_pool_f32(const float *patches, long n, const DeviceConsts &dc, int angle_mode, float *pooled,
                     hipStream_t stream) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)((n + 63) / 64);
    const f32x4 *lut = reinterpret_cast<const f32x4 *>(dc.pool_b_f32);
    if (angle_mode == LF_ANGLE_EXACT)
        hipLaunchKernelGGL(mkd_pool_f32<LF_ANGLE_EXACT>, dim3(grid), dim3(256), 0, stream, patches, n, lut,
                           dc.phi_cs, dc.colmap, pooled);
    else
        hipLaunchKernelGGL(mkd_pool_f32<LF_ANGLE_SHADER>, dim3(grid), dim3(256), 0, stream, patches, n, lut,
                           dc.phi_cs, dc.colmap, pooled);
}

void launch_whiten_f32(const float *pooled, long n, const DeviceConsts &dc, float *out, float *raw_out,
                       hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mkd_whiten_f32, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, pooled, n,
                       dc.white_b_f32, dc.mean_pad, out, raw_out);
}

void launch_sample_patches(const float *pyr, const PyramidDesc &pd, const float *kps, long n, float psf,
                           float *patches, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(sample_patches, dim3((unsigned)n), dim3(1024), 0, stream, pyr, pd, kps, n, psf, patches);
}

void launch_build_pyramid(const float *image, float *pyr, float *tmp_a, float *tmp_b, const PyramidDesc &pd,
                          hipStream_t stream) {
    const int w = pd.w[0], h = pd.h[0];
    const dim3 blk(32, 8);
    auto grid = [&](int gw, int gh) { return dim3((gw + 31) / 32, (gh + 7) / 8); };
    // level 0: sigma-0.6 blur, H then V (tasks_detect.rs:150-161, mod.rs:1043-1067)
    hipLaunchKernelGGL(pyr_sep3, grid(w, h), blk, 0, stream, image, tmp_a, w, h, 0.66381836f, 0.16809084f,
                       1.015267163f, 0);
    hipLaunchKernelGGL(pyr_sep3, grid(w, h), blk, 0, stream, (const float *)tmp_a, pyr + pd.offset[0], w, h,
                       0.66381836f, 0.16809084f, 1.015267163f, 1);
    if (pd.levels < 2) return;
    // level 1: one a-trous pass over level 0, nearest-decimated
    hipLaunchKernelGGL(pyr_swt0, grid(w, h), blk, 0, stream, (const float *)(pyr + pd.offset[0]), tmp_a, w, h, 0);
    hipLaunchKernelGGL(pyr_swt0, grid(w, h), blk, 0, stream, (const float *)tmp_a, tmp_b, w, h, 1);
    hipLaunchKernelGGL(pyr_decimate, grid(pd.w[1], pd.h[1]), blk, 0, stream, (const float *)tmp_b,
                       pyr + pd.offset[1], w, h, pd.w[1], pd.h[1]);
    // levels >= 2: binomial H at the resolution of level l-1, then V with 2x decimation
    for (int l = 2; l < pd.levels; ++l) {
        const int pw = pd.w[l - 1], ph = pd.h[l - 1];
        hipLaunchKernelGGL(pyr_sep3, grid(pw, ph), blk, 0, stream, (const float *)(pyr + pd.offset[l - 1]), tmp_a,
                           pw, ph, 0.375f, 0.3125f, 1.2f, 0);
        hipLaunchKernelGGL(pyr_down_v, grid(pd.w[l], pd.h[l]), blk, 0, stream, (const float *)tmp_a,
                           pyr + pd.offset[l], pw, ph, pd.w[l], pd.h[l]);
    }
}

}  // namespace lfmkd
