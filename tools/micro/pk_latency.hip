// Dependent-issue cost of PACKED f32 VALU ops on gfx950 (v_pk_fma_f32): ns per instruction per SIMD for chains with
// ILP = 1, 2, 3, 4, 6, 8 independent accumulators, at 1 and 2 waves per SIMD.  (valu_latency.hip is the scalar twin.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <int ILP>
__global__ void k(float *out, int iters, float a, float b) {
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)threadIdx.x + i, (float)i};
    const f32x2 va = {a, a}, vb = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 48 / ILP; ++u)
#pragma unroll
            for (int j = 0; j < ILP; ++j) x[j] = pk_fma(x[j], va, vb);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
void run(int threads, float *d) {
    const int iters = 8000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * (48 / ILP) * ILP;
    printf("v_pk_fma_f32 ILP=%d  %4d thr/CU: %.2f ns per instruction per wave  (%.2f ns per instr per SIMD)\n", ILP, threads,
           ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave / (threads / 256.0));
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512}) { run<1>(thr, d); run<2>(thr, d); run<3>(thr, d); run<4>(thr, d); run<6>(thr, d); run<8>(thr, d); }
    return 0;
}
