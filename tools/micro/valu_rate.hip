// VALU issue rate on gfx950: cycles per wave64 instruction per SIMD for scalar-f32 and packed-f32 ops
// at 1, 2 and 4 waves per SIMD.  (Decides whether the pooling kernel's per-pixel math should be packed.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x2}, p5 = {x3, x4}, p6 = {x5, x6}, p7 = {x7, x0};
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {  // 8 independent v_fma_f32
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            } else if (MODE == 1) {  // 8 independent v_pk_fma_f32
                p0 = __builtin_elementwise_fma(p0, a2, b2); p1 = __builtin_elementwise_fma(p1, a2, b2);
                p2 = __builtin_elementwise_fma(p2, a2, b2); p3 = __builtin_elementwise_fma(p3, a2, b2);
                p4 = __builtin_elementwise_fma(p4, a2, b2); p5 = __builtin_elementwise_fma(p5, a2, b2);
                p6 = __builtin_elementwise_fma(p6, a2, b2); p7 = __builtin_elementwise_fma(p7, a2, b2);
            } else {  // 8 independent v_mul_f32
                x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

template <int MODE>
void run(const char *name, int threads, float *d) {
    const int iters = 4000, blocks = 256;  // one block per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * 8;
    const double waves_per_simd = threads / 64.0 / 4.0;
    // cycles at a nominal 2.4 GHz per wave-instruction per SIMD
    printf("%-14s %4d thr/CU (%.0f waves/SIMD): %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, threads,
           waves_per_simd, ms, ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4);
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512, 1024}) { run<0>("v_fma_f32", thr, d); run<1>("v_pk_fma_f32", thr, d); run<2>("v_mul_f32", thr, d); }
    return 0;
}
