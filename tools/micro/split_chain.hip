// Cost of the f16 hi/lo split of one stream (4 pixel pairs: pk_mul, cvt_pkrtz, 2 fma_mix, cvt_pkrtz each) on gfx950
// when the five instructions of a pair follow each other (what hipcc emits: every instruction depends on the one
// before it) against the same 20 instructions ordered stage by stage across the four pairs, at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(unsigned *out, int iters, float a, float b) {
    f32x2 m[4], t[4];
    for (int e = 0; e < 4; ++e) { m[e] = f32x2{a + threadIdx.x + e, a * 2 + e}; t[e] = f32x2{b + e, b * 3 + threadIdx.x}; }
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            f32x2 p[4];
            unsigned h[4], l[4];
            float l0[4], l1[4];
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 0" : "=v"(p[e]) : "v"(m[e]), "v"(t[e]));
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[e]) : "v"(p[e].x), "v"(p[e].y));
                    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0[e]) : "v"(m[e].x), "v"(t[e].x), "v"(h[e]));
                    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1[e]) : "v"(m[e].y), "v"(t[e].y), "v"(h[e]));
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(l[e]) : "v"(l0[e]), "v"(l1[e]));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[e]) : "v"(m[e]), "v"(t[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[e]) : "v"(p[e].x), "v"(p[e].y));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0[e]) : "v"(m[e].x), "v"(t[e].x), "v"(h[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1[e]) : "v"(m[e].y), "v"(t[e].y), "v"(h[e]));
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(l[e]) : "v"(l0[e]), "v"(l1[e]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc ^= h[e] ^ l[e];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(int threads, unsigned *d) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double streams = (double)iters * 4;   // one "stream split" = 4 pairs = 20 instructions (+ 8 xor here)
    printf("%s  %4d thr/CU: %.1f ns per stream split per wave, %.1f ns per SIMD\n", MODE == 0 ? "pair by pair  " : "stage by stage",
           threads, ms * 1e6 / streams, ms * 1e6 / streams / (threads / 256.0));
}

int main() {
    unsigned *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512}) { run<0>(thr, d); run<1>(thr, d); }
    return 0;
}
