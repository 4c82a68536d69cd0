// Does the matrix pipe run beside the VALU on gfx950?  Exact instruction sequences via inline asm.
// Per loop iteration: G groups of [NM x v_mfma_f32_16x16x32_f16 (4 rotating accumulators)] + [NV x VALU op].
// VALU op selected by KIND: 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_mul_f32 (2-operand), 3 = v_cvt_pkrtz_f16_f32,
// 4 = v_fma_f32 with all-distinct banks.  Reports ns per iteration per SIMD and the sum of the parts.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu(float &x, f2 &p, float a, float b) {
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
    if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    if (KIND == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
}

template <int NM, int NV, int KIND, int MSHAPE>
__global__ void k(float *out, int iters, float a, float b) {
    float x[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; p[i] = f2{x[i], x[i] + 1}; }
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(threadIdx.x & 3); B[i] = (_Float16)1; }
    f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v C32[2];
    for (int i = 0; i < 16; ++i) { C32[0][i] = 0; C32[1][i] = 0; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (MSHAPE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[(g * NM + m) & 3]) : "v"(A), "v"(B));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C32[(g * NM + m) & 1]) : "v"(A), "v"(B));
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) valu<KIND>(x[j & 7], p[j & 7], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    for (int i = 0; i < 16; ++i) s += C32[0][i] + C32[1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

template <int NM, int NV, int KIND, int MSHAPE>
double run(int threads, float *d) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, KIND, MSHAPE>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, KIND, MSHAPE>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)iters * 8 * (threads / 256.0));   // ns per group per wave-slot of a SIMD
}

template <int NM, int NV, int KIND, int MSHAPE>
void report(float *d, const char *name) {
    for (int thr : {256, 512, 1024}) {
        const double both = run<NM, NV, KIND, MSHAPE>(thr, d), m = run<NM, 0, KIND, MSHAPE>(thr, d), v = run<0, NV, KIND, MSHAPE>(thr, d);
        printf("%-14s shape%d NM=%d NV=%2d waves/SIMD=%d: both %.2f ns  mfma %.2f  valu %.2f  sum %.2f  max %.2f  -> overlap %.0f%%\n", name,
               MSHAPE, NM, NV, thr / 256, both, m, v, m + v, m > v ? m : v, 100.0 * (m + v - both) / (m < v ? m : v));
    }
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    report<1, 2, 0, 0>(d, "v_fma_f32"); report<1, 4, 0, 0>(d, "v_fma_f32"); report<1, 8, 0, 0>(d, "v_fma_f32");
    report<4, 16, 0, 0>(d, "v_fma_f32"); report<8, 32, 0, 0>(d, "v_fma_f32");
    report<1, 4, 1, 0>(d, "v_pk_fma_f32"); report<1, 4, 2, 0>(d, "v_mul_f32"); report<1, 4, 3, 0>(d, "v_cvt_pkrtz");
    report<1, 4, 0, 1>(d, "v_fma_f32"); report<1, 8, 0, 1>(d, "v_fma_f32"); report<2, 16, 0, 1>(d, "v_fma_f32");
    return 0;
}
