// Do transcendental VALU ops (v_sqrt_f32: quarter rate) overlap with plain VALU ops on gfx950 when interleaved?
// Per iteration 16 v_sqrt_f32 and 48 v_fma_f32, all independent: clustered (16 sqrt, then 48 fma) against interleaved
// (sqrt, fma, fma, fma) x 16, at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(float *out, int iters, float a, float b) {
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x + i + 1.f; y[i] = threadIdx.x * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[j]));
#pragma unroll
            for (int j = 0; j < 48; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[j & 15]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[j]));
#pragma unroll
                for (int q = 0; q < 3; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[(3 * j + q) & 15]) : "v"(a), "v"(b));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 48; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[j & 15]) : "v"(a), "v"(b));
        }
    }
    float t = 0;
    for (int i = 0; i < 16; ++i) t += x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE>
void run(int threads, float *d, const char *what) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-24s %4d thr/CU: %.1f ns per iteration per SIMD\n", what, threads, ms * 1e6 / iters / (threads / 256.0));
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512}) {
        run<2>(thr, d, "16 sqrt"); run<3>(thr, d, "48 fma"); run<0>(thr, d, "clustered"); run<1>(thr, d, "interleaved");
    }
    return 0;
}
