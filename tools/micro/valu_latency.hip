// Dependent-issue latency of scalar f32 VALU ops on gfx950: time per instruction for chains with
// ILP = 1, 2, 4, 8 independent accumulators at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP>
__global__ void k(float *out, int iters, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64 / ILP; ++u)
#pragma unroll
            for (int j = 0; j < ILP; ++j) x[j] = fmaf(x[j], a, b);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
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
    const double instr_per_wave = (double)iters * 64;
    printf("ILP=%d  %4d thr/CU: %.2f ns per instruction per wave  (%.2f ns per instr per SIMD)\n", ILP, threads,
           ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave / (threads / 256.0));
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512}) { run<1>(thr, d); run<2>(thr, d); run<4>(thr, d); run<8>(thr, d); }
    return 0;
}
