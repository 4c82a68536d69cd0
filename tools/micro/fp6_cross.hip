// fp6 cross terms for the pooling contraction (NOTEBOOK.md section 8.1): can the two cross terms of the
// f16 hi/lo split -- hi_s x lo_L and lo_s x hi_L, 54 of the 81 v_mfma_f32_16x16x32_f16 per wave-row -- ride on
// v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands (4x the K at the cycles of the f16 form)?
//   1. layout:  which input of v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_fp6_f16 lands in which 6-bit field, and
//               which field the matrix instruction reads as which K index (checked against a host dot product, scales included)
//   2. prices:  cycles per scaled MFMA (fp6, fp8) beside the f16 form, per conversion instruction, one and two waves per SIMD
//   3. a harmonic of the row loop both ways: split of the cos / sin streams + its matrix instructions
//      old: 2 x (4 pkrtz + 8 fma_mix + 4 pkrtz) + 24 f16 MFMAs;  new: 8 pkrtz + 16 fma_mix + 8 pk_mul + 1 cvt + 8 f16 + 7 fp6 MFMAs
// build: hipcc -O3 --offload-arch=gfx950 -o fp6_cross fp6_cross.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// e2m3 magnitude of code c (5 bits): exponent c >> 3, mantissa c & 7
static float e2m3(int c) {
    const int e = (c >> 3) & 3, m = c & 7;
    return e == 0 ? m / 8.f : (1.f + m / 8.f) * (float)(1 << (e - 1));
}

__global__ void probe_cvt(const float *in, unsigned *out_f32, unsigned *out_f16, float scale) {
    v16f a, b;
    v32h h;
    for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
    for (int i = 0; i < 32; ++i) h[i] = (_Float16)in[i];
    const v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    const v6u s = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scale);
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) { out_f32[i] = r[i]; out_f16[i] = s[i]; }
}

// one scaled MFMA on host-packed operands: lane (r = l & 15, q = l >> 4) brings K = 32 q .. 32 q + 31 of row r
__global__ void probe_mfma(const unsigned *a6, const unsigned *b6, float *d, int sa, int sb) {
    v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = A;
    for (int i = 0; i < 6; ++i) { A[i] = a6[threadIdx.x * 6 + i]; B[i] = b6[threadIdx.x * 6 + i]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 2, 2, 0, sa, 0, sb);
    for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = c[i];
}

// ---- prices ---------------------------------------------------------------------------------------------------------
template <int FMT>   // 2: fp6 e2m3, 0: fp8 e4m3, 4: fp4
__global__ void t_scaled(float *out, int iters) {
    v8i A, B;
    for (int i = 0; i < 8; ++i) { A[i] = threadIdx.x * 2654435761u + i * 40503u; B[i] = threadIdx.x * 97u + i * 7919u; }
    if (FMT == 0) for (int i = 0; i < 8; ++i) { A[i] &= 0x77777777; B[i] &= 0x77777777; }   // no NaN encodings
    f4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c[j], FMT, FMT, 0, 127, 0, 127);
    }
    f4 s = c[0];
    for (int j = 1; j < 8; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void t_f16(float *out, int iters) {
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(0.001f * (threadIdx.x + i)); B[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c[j], 0, 0, 0);
    }
    f4 s = c[0];
    for (int j = 1; j < 8; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// conversions: independent instructions (the inputs are laundered so that nothing is hoisted)
template <int SRC>   // 0: 2xpk16 from f32, 1: pk32 from f16
__global__ void t_cvt(float *out, int iters, float scale) {
    v16f a, b;
    v32h h;
    for (int i = 0; i < 16; ++i) { a[i] = 0.01f * (threadIdx.x + i); b[i] = 0.02f * (threadIdx.x + i); }
    for (int i = 0; i < 32; ++i) h[i] = (_Float16)(0.01f * (threadIdx.x + i));
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v6u r;
            if (SRC == 0) { asm volatile("" : "+v"(a), "+v"(b)); r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale); }
            else { asm volatile("" : "+v"(h)); r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scale); }
            asm volatile("" ::"v"(r));
            acc ^= r[0];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

// ---- a harmonic of the row loop, both ways ------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_rtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// old: the product kernel's AFrag::set2 + 24 MFMAs (7 accumulators; LUT fragments held in registers here: the LDS reads
// are the same in both forms apart from their width)
__global__ void harm_old(float *out, int iters, float seed) {
    f2 pa[4], pb[4];
    for (int e = 0; e < 4; ++e) { pa[e] = f2{seed * (threadIdx.x + e), seed * (threadIdx.x + 2 * e)}; pb[e] = pa[e] * 0.7f; }
    h8 lut_h[4], lut_l[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 8; ++i) { lut_h[t][i] = (_Float16)(0.01f * (t + i + (threadIdx.x & 7))); lut_l[t][i] = (_Float16)(1e-5f * (t + i)); }
    f4 acc[7];
    for (int j = 0; j < 7; ++j) acc[j] = f4{0, 0, 0, 0};
    float one = 1.f;
    asm("" : "+v"(one));
    for (int it = 0; it < iters; ++it) {
        for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(pa[e]), "+v"(pb[e])); }
        u4 xh, xl, yh, yl;
        float ra0[4], ra1[4], rb0[4], rb1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { xh[e] = pack_rtz(pa[e].x, pa[e].y); yh[e] = pack_rtz(pb[e].x, pb[e].y); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ra0[e] = __builtin_fmaf(pa[e].x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(xh[e] & 0xffffu)));
            ra1[e] = __builtin_fmaf(pa[e].y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(xh[e] >> 16)));
            rb0[e] = __builtin_fmaf(pb[e].x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(yh[e] & 0xffffu)));
            rb1[e] = __builtin_fmaf(pb[e].y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(yh[e] >> 16)));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { xl[e] = pack_rtz(ra0[e], ra1[e]); yl[e] = pack_rtz(rb0[e], rb1[e]); }
        const h8 ch = __builtin_bit_cast(h8, xh), cl = __builtin_bit_cast(h8, xl), sh = __builtin_bit_cast(h8, yh), sl = __builtin_bit_cast(h8, yl);
        // products: cos x P0 -> 0 | sin x P0 -> 2 | sin x Q0 -> 1 | cos x R -> 3 | sin x R -> 4 | cos x S -> 5 | sin x S -> 6 | cos x Q0 -> 2
        const int tile[8] = {0, 0, 1, 2, 2, 3, 3, 1}, dst[8] = {0, 2, 1, 3, 4, 5, 6, 2};
        const bool is_cos[8] = {true, false, false, true, false, true, false, true};
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const h8 sH = is_cos[p] ? ch : sh, sL = is_cos[p] ? cl : sl;
                if (part == 0) acc[dst[p]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut_h[tile[p]], sL, acc[dst[p]], 0, 0, 0);
                else if (part == 1) acc[dst[p]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut_l[tile[p]], sH, acc[dst[p]], 0, 0, 0);
                else acc[dst[p]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut_h[tile[p]], sH, acc[dst[p]], 0, 0, 0);
            }
    }
    f4 s = acc[0];
    for (int j = 1; j < 7; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// new: hi parts as f16 for the hi x hi term; (a, 1024 r) of both streams through ONE conversion into 32 fp6 slots
// [cos: a, r | sin: a, r]; single-stream tiles read the stream's three registers beside three zeros, the merged tile all six
template <int ZERO_ON>   // 0: zeros on the stream side (B = [stream | 0]), 1: on the LUT side (A = [frag | 0], B = all six)
__global__ void harm_new(float *out, int iters, float seed, int scale_bits) {
    f2 pa[4], pb[4];
    for (int e = 0; e < 4; ++e) { pa[e] = f2{seed * (threadIdx.x + e), seed * (threadIdx.x + 2 * e)}; pb[e] = pa[e] * 0.7f; }
    h8 lut_h[4];
    unsigned lut_x[4][3];   // fp6 cross fragments: 16 slots = 3 registers per tile
    for (int t = 0; t < 4; ++t) {
        for (int i = 0; i < 8; ++i) lut_h[t][i] = (_Float16)(0.01f * (t + i + (threadIdx.x & 7)));
        for (int i = 0; i < 3; ++i) lut_x[t][i] = (threadIdx.x * 2654435761u) ^ (t * 40503u + i);
    }
    f4 acc[7];
    for (int j = 0; j < 7; ++j) acc[j] = f4{0, 0, 0, 0};
    float one = 1.f;
    asm("" : "+v"(one));
    const f2 k1024 = {1024.f, 1024.f};
    for (int it = 0; it < iters; ++it) {
        for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(pa[e]), "+v"(pb[e])); }
        u4 xh, yh;
        f2 ra[4], rb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { xh[e] = pack_rtz(pa[e].x, pa[e].y); yh[e] = pack_rtz(pb[e].x, pb[e].y); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ra[e].x = __builtin_fmaf(pa[e].x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(xh[e] & 0xffffu)));
            ra[e].y = __builtin_fmaf(pa[e].y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(xh[e] >> 16)));
            rb[e].x = __builtin_fmaf(pb[e].x, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(yh[e] & 0xffffu)));
            rb[e].y = __builtin_fmaf(pb[e].y, one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(yh[e] >> 16)));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { ra[e] *= k1024; rb[e] *= k1024; }
        v16f c16, s16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c16[2 * e] = pa[e].x; c16[2 * e + 1] = pa[e].y; c16[8 + 2 * e] = ra[e].x; c16[8 + 2 * e + 1] = ra[e].y;
            s16[2 * e] = pb[e].x; s16[2 * e + 1] = pb[e].y; s16[8 + 2 * e] = rb[e].x; s16[8 + 2 * e + 1] = rb[e].y;
        }
        const v6u d = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(c16, s16, __builtin_bit_cast(float, scale_bits));
        const h8 ch = __builtin_bit_cast(h8, xh), sh = __builtin_bit_cast(h8, yh);
        const int tile[8] = {0, 0, 1, 2, 2, 3, 3, 1}, dst[8] = {0, 2, 1, 3, 4, 5, 6, 2};
        const bool is_cos[8] = {true, false, false, true, false, true, false, true};
        // hi x hi
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[dst[p]] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut_h[tile[p]], is_cos[p] ? ch : sh, acc[dst[p]], 0, 0, 0);
        // cross terms: one instruction per accumulator
        const int sb = scale_bits >> 23;
        auto frag = [&](int t, int u) { v8i A; A[0] = lut_x[t][0]; A[1] = lut_x[t][1]; A[2] = lut_x[t][2];
                                          if (u >= 0) { A[3] = lut_x[u][0]; A[4] = lut_x[u][1]; A[5] = lut_x[u][2]; }
                                          else if (ZERO_ON == 1) { A[3] = 0; A[4] = 0; A[5] = 0; }
                                          return A; };
        auto frag_hi = [&](int t) { v8i A; A[0] = 0; A[1] = 0; A[2] = 0; A[3] = lut_x[t][0]; A[4] = lut_x[t][1]; A[5] = lut_x[t][2]; return A; };
        v8i Bc, Bs, Bm;
        Bm[0] = d[0]; Bm[1] = d[1]; Bm[2] = d[2]; Bm[3] = d[3]; Bm[4] = d[4]; Bm[5] = d[5];
        if (ZERO_ON == 0) {
            Bc[0] = d[0]; Bc[1] = d[1]; Bc[2] = d[2]; Bc[3] = 0; Bc[4] = 0; Bc[5] = 0;
            Bs[0] = d[3]; Bs[1] = d[4]; Bs[2] = d[5]; Bs[3] = 0; Bs[4] = 0; Bs[5] = 0;
#define XMMA(A_, B_, j_) acc[j_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A_, B_, acc[j_], 2, 2, 0, 127, 0, sb)
            XMMA(frag(0, -1), Bc, 0); XMMA(frag(1, -1), Bs, 1); XMMA(frag(1, 0), Bm, 2);   // merged: cos x Q0 + sin x P0
            XMMA(frag(2, -1), Bc, 3); XMMA(frag(2, -1), Bs, 4); XMMA(frag(3, -1), Bc, 5); XMMA(frag(3, -1), Bs, 6);
        } else {
            XMMA(frag(0, -1), Bm, 0); XMMA(frag_hi(1), Bm, 1); XMMA(frag(1, 0), Bm, 2);
            XMMA(frag(2, -1), Bm, 3); XMMA(frag_hi(2), Bm, 4); XMMA(frag(3, -1), Bm, 5); XMMA(frag_hi(3), Bm, 6);
        }
    }
    f4 s = acc[0];
    for (int j = 1; j < 7; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <class F>
static void time_it(const char *name, F launch, double per_iter_units, const char *unit) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int thr : {256, 512}) {
        const int iters = 20000;
        launch(thr, 50);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch(thr, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // SIMD time per unit: a SIMD hosts thr / 256 waves, each runs iters x per_iter_units units
        const double ns = ms * 1e6 / ((double)iters * per_iter_units * (thr / 256));
        printf("%-44s %d wave(s)/SIMD: %8.2f ns per %s per SIMD\n", name, thr / 256, ns, unit);
    }
}

int main() {
    float *d_out;
    CK(hipMalloc(&d_out, 1 << 22));
    // ---- 1a. conversion layout
    {
        float h_in[32];
        for (int i = 0; i < 32; ++i) h_in[i] = e2m3(i);
        float *d_in; unsigned *d_a, *d_b;
        CK(hipMalloc(&d_in, 128)); CK(hipMalloc(&d_a, 24)); CK(hipMalloc(&d_b, 24));
        CK(hipMemcpy(d_in, h_in, 128, hipMemcpyHostToDevice));
        for (float scale : {1.f, 2.f}) {
            hipLaunchKernelGGL(probe_cvt, dim3(1), dim3(64), 0, 0, d_in, d_a, d_b, scale);
            unsigned a[6], b[6];
            CK(hipMemcpy(a, d_a, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(b, d_b, 24, hipMemcpyDeviceToHost));
            for (int which = 0; which < 2; ++which) {
                const unsigned *r = which ? b : a;
                printf("%s scale %.0f: field -> code of the input it came from (input i holds e2m3 code i):\n  ",
                       which ? "v_cvt_scalef32_pk32_fp6_f16   " : "v_cvt_scalef32_2xpk16_fp6_f32 ", scale);
                for (int f = 0; f < 32; ++f) {
                    const int bit = 6 * f;
                    unsigned long long w = r[bit >> 5] | ((unsigned long long)(bit / 32 + 1 < 6 ? r[bit / 32 + 1] : 0) << 32);
                    printf("%d ", (int)((w >> (bit & 31)) & 63));
                }
                printf("\n");
            }
        }
    }
    // ---- 1b. matrix instruction layout + scales
    {
        std::vector<int> ca(16 * 128), cb(16 * 128);
        srand(1);
        for (auto &c : ca) c = rand() & 63;
        for (auto &c : cb) c = rand() & 63;
        auto val = [](int c) { return (c & 32 ? -1.f : 1.f) * e2m3(c & 31); };
        auto pack = [&](const std::vector<int> &c) {
            std::vector<unsigned> p(64 * 6, 0u);
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 32; ++j) {
                    const unsigned long long v = (unsigned long long)c[(l & 15) * 128 + 32 * (l >> 4) + j] << ((6 * j) & 31);
                    p[l * 6 + (6 * j) / 32] |= (unsigned)v;
                    if ((6 * j) / 32 + 1 < 6) p[l * 6 + (6 * j) / 32 + 1] |= (unsigned)(v >> 32);
                }
            return p;
        };
        const auto pa = pack(ca), pb = pack(cb);
        unsigned *d_a, *d_b; float *d_d;
        CK(hipMalloc(&d_a, pa.size() * 4)); CK(hipMalloc(&d_b, pb.size() * 4)); CK(hipMalloc(&d_d, 1024));
        CK(hipMemcpy(d_a, pa.data(), pa.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_b, pb.data(), pb.size() * 4, hipMemcpyHostToDevice));
        for (int sa : {127, 125})
            for (int sb : {127, 130}) {
                hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d, sa, sb);
                float d[256];
                CK(hipMemcpy(d, d_d, 1024, hipMemcpyDeviceToHost));
                double worst = 0;
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 4; ++i) {
                        const int m = 4 * (l >> 4) + i, n = l & 15;
                        double ref = 0;
                        for (int k = 0; k < 128; ++k) ref += (double)val(ca[m * 128 + k]) * val(cb[n * 128 + k]);
                        ref *= std::ldexp(1.0, sa - 127 + sb - 127);
                        worst = std::fmax(worst, std::fabs(ref - d[l * 4 + i]));
                    }
                printf("scaled fp6 MFMA, K = 32 (lane >> 4) + field, D[4 (lane >> 4) + i][lane & 15], scales 2^%d x 2^%d: max |diff| vs host = %g\n",
                       sa - 127, sb - 127, worst);
            }
    }
    // ---- 2. prices
    time_it("v_mfma_f32_16x16x32_f16", [&](int thr, int it) { hipLaunchKernelGGL(t_f16, dim3(256), dim3(thr), 0, 0, d_out, it); }, 8, "MFMA");
    time_it("v_mfma_scale_f32_16x16x128_f8f6f4 fp6", [&](int thr, int it) { hipLaunchKernelGGL(t_scaled<2>, dim3(256), dim3(thr), 0, 0, d_out, it); }, 8, "MFMA");
    time_it("v_mfma_scale_f32_16x16x128_f8f6f4 fp4", [&](int thr, int it) { hipLaunchKernelGGL(t_scaled<4>, dim3(256), dim3(thr), 0, 0, d_out, it); }, 8, "MFMA");
    time_it("v_mfma_scale_f32_16x16x128_f8f6f4 fp8", [&](int thr, int it) { hipLaunchKernelGGL(t_scaled<0>, dim3(256), dim3(thr), 0, 0, d_out, it); }, 8, "MFMA");
    time_it("v_cvt_scalef32_2xpk16_fp6_f32", [&](int thr, int it) { hipLaunchKernelGGL(t_cvt<0>, dim3(256), dim3(thr), 0, 0, d_out, it, 1.f); }, 4, "cvt");
    time_it("v_cvt_scalef32_pk32_fp6_f16", [&](int thr, int it) { hipLaunchKernelGGL(t_cvt<1>, dim3(256), dim3(thr), 0, 0, d_out, it, 1.f); }, 4, "cvt");
    // ---- 3. a harmonic both ways
    time_it("harmonic, f16x3 (24 MFMA + split)", [&](int thr, int it) { hipLaunchKernelGGL(harm_old, dim3(256), dim3(thr), 0, 0, d_out, it, 1e-3f); }, 1, "harmonic");
    time_it("harmonic, f16 + fp6 cross, zeros in B", [&](int thr, int it) { hipLaunchKernelGGL(harm_new<0>, dim3(256), dim3(thr), 0, 0, d_out, it, 1e-3f, 0x3f800000); }, 1, "harmonic");
    time_it("harmonic, f16 + fp6 cross, zeros in A", [&](int thr, int it) { hipLaunchKernelGGL(harm_new<1>, dim3(256), dim3(thr), 0, 0, d_out, it, 1e-3f, 0x3f800000); }, 1, "harmonic");
    return 0;
}
