// Which structure runs the describe kernel's instruction mix faster on gfx950?
//   A: two waves per SIMD, 16 patches per wave, v_mfma_f32_16x16x32_f16 (what mkd_pool does): per patch row of a wave
//      NMA matrix instructions on 24 four-register accumulators, NVA vector instructions, NLA 16-byte LDS reads
//   B: one wave per SIMD, 32 patches per wave, v_mfma_f32_32x32x16_f16: per patch row NMA/2 + a few matrix instructions on
//      14 sixteen-register accumulators (AccVGPRs), 2 NVA vector instructions, ~1.3 NLA LDS reads
// Both do the same flops per patch; the question is how much of the matrix time hides behind the vector work.
// Vector mix per 8 instructions: 3 v_pk_fma_f32, 2 v_cvt_pkrtz_f16_f32, 2 v_fma_mix_f32, 1 v_fma_f32; the cvt results feed
// the matrix instructions' B operand (so the matrix work depends on the vector work as in the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void valu8(f2 (&p)[6], float (&x)[4], u4 &frag, float a) {
    // 3 pk_fma
    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[0]) : "v"(p[3]));
    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[1]) : "v"(p[4]));
    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[2]) : "v"(p[5]));
    // 2 cvt_pkrtz into the fragment, 2 fma_mix
    unsigned h0, h1;
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h0) : "v"(p[0].x), "v"(p[0].y));
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(p[1].x), "v"(p[1].y));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(x[0]) : "v"(p[0].x), "v"(a), "v"(h0));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(x[1]) : "v"(p[1].x), "v"(a), "v"(h1));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[2]) : "v"(a), "v"(x[3]));
    frag[0] ^= h0; frag[1] ^= h1;   // (two more VALU ops: keeps the fragment data-dependent)
}

template <int NM, int NV8, int NL>
__global__ __launch_bounds__(512) void kernA(float *out, int rows, float a) {
    __shared__ u4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = u4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    f2 p[6]; float x[4];
    for (int i = 0; i < 6; ++i) p[i] = f2{(float)threadIdx.x + i, 1.0f + i * 1e-3f};
    for (int i = 0; i < 4; ++i) x[i] = i;
    u4 frag = {1, 2, 3, 4}, lut = {5, 6, 7, 8};
    f4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f4{0, 0, 0, 0};
    const u4 *lp = lds + (threadIdx.x & 63);
    for (int r = 0; r < rows; ++r) {
        int vi = 0, li = 0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            // interleave: per matrix instruction its share of vector work and LDS reads
            for (; vi * NM < (m + 1) * NV8; ++vi) valu8(p, x, frag, a);
            for (; li * NM < (m + 1) * NL; ++li) { const u4 t = lp[((r + li) & 63) * 64]; lut[0] ^= t[0]; lut[1] ^= t[1]; lut[2] ^= t[2]; lut[3] ^= t[3]; }
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m % 24]) : "v"(lut), "v"(frag));
        }
        __syncthreads();
    }
    float s = x[0] + x[1] + x[2] + p[0].x + p[1].y + p[2].x;
    for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV8, int NL, int THREADS>
__global__ __launch_bounds__(THREADS) void kernB(float *out, int rows, float a) {
    __shared__ u4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = u4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    f2 p[6]; float x[4];
    for (int i = 0; i < 6; ++i) p[i] = f2{(float)threadIdx.x + i, 1.0f + i * 1e-3f};
    for (int i = 0; i < 4; ++i) x[i] = i;
    u4 frag = {1, 2, 3, 4}, lut = {5, 6, 7, 8};
    f16v acc[14];
    for (int i = 0; i < 14; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    const u4 *lp = lds + (threadIdx.x & 63);
    for (int r = 0; r < rows; ++r) {
        int vi = 0, li = 0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            for (; vi * NM < (m + 1) * NV8; ++vi) valu8(p, x, frag, a);
            for (; li * NM < (m + 1) * NL; ++li) { const u4 t = lp[((r + li) & 63) * 64]; lut[0] ^= t[0]; lut[1] ^= t[1]; lut[2] ^= t[2]; lut[3] ^= t[3]; }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m % 14]) : "v"(lut), "v"(frag));
        }
        __syncthreads();
    }
    float s = x[0] + x[1] + x[2] + p[0].x + p[1].y + p[2].x;
    for (int i = 0; i < 14; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(8);
    (void)hipEventRecord(e0);
    launch(512);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    const int rows = 512;
    // A: per wave row 81 matrix, ~425 + 16 vector (valu8 = 10 instructions -> 44 calls), 46 LDS reads; 8 waves x 16 patches per CU
    {
        double ms = time_ms([&](int r) { hipLaunchKernelGGL((kernA<81, 44, 46>), dim3(256), dim3(512), 0, 0, d, r, 1.0001f); });
        printf("A  16x16x32, 2 waves/SIMD, 16 patches/wave: %.3f ms for %d rows -> %.2f ns per patch row per CU\n", ms, rows, ms * 1e6 / rows / 128);
    }
    {   // matrix only / vector only
        double mm = time_ms([&](int r) { hipLaunchKernelGGL((kernA<81, 0, 0>), dim3(256), dim3(512), 0, 0, d, r, 1.0001f); });
        double mv = time_ms([&](int r) { hipLaunchKernelGGL((kernA<1, 44, 46>), dim3(256), dim3(512), 0, 0, d, r, 1.0001f); });
        printf("   matrix alone %.3f ms, vector + LDS alone %.3f ms\n", mm, mv);
    }
    // B: per wave row 84 matrix (32x32x16), 2 x 44 valu8 calls, 60 LDS reads; 4 waves x 32 patches per CU
    {
        double ms = time_ms([&](int r) { hipLaunchKernelGGL((kernB<84, 88, 60, 256>), dim3(256), dim3(256), 0, 0, d, r, 1.0001f); });
        printf("B  32x32x16, 1 wave/SIMD, 32 patches/wave:  %.3f ms for %d rows -> %.2f ns per patch row per CU\n", ms, rows, ms * 1e6 / rows / 128);
    }
    {
        double mm = time_ms([&](int r) { hipLaunchKernelGGL((kernB<84, 0, 0, 256>), dim3(256), dim3(256), 0, 0, d, r, 1.0001f); });
        double mv = time_ms([&](int r) { hipLaunchKernelGGL((kernB<1, 88, 60, 256>), dim3(256), dim3(256), 0, 0, d, r, 1.0001f); });
        printf("   matrix alone %.3f ms, vector + LDS alone %.3f ms\n", mm, mv);
    }
    // B2: the same with two waves per SIMD (spills: 224 accumulator registers do not fit 256; an upper bound on nothing, shown for curiosity)
    {
        double ms = time_ms([&](int r) { hipLaunchKernelGGL((kernB<84, 88, 60, 512>), dim3(256), dim3(512), 0, 0, d, r, 1.0001f); });
        printf("B2 32x32x16, 2 waves/SIMD (hypothetical):   %.3f ms for %d rows -> %.2f ns per patch row per CU\n", ms, rows, ms * 1e6 / rows / 256);
    }
    return 0;
}
