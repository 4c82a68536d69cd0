// How well do VALU and MFMA instructions of the same wave(s) overlap on gfx950?
// Loop body: 1 x v_mfma_f32_16x16x32_f16 (independent accumulators rotate) + NV independent v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, int NM>
__global__ void k(float *out, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(threadIdx.x & 3); B[i] = (_Float16)1; }
    f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int m = 0; m < NM; ++m) c[(u + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c[(u + m) & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) x[j & 15] = fmaf(x[j & 15], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

template <int NV, int NM>
void run(int threads, float *d) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NM>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NM>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 8 * (threads / 256.0);   // (NM mfma + NV valu) groups per SIMD
    printf("NM=%d MFMA + NV=%2d VALU, %4d thr/CU: %.2f ns per group per SIMD (VALU alone would be %.1f, MFMA alone %.1f)\n", NM, NV,
           threads, ms * 1e6 / groups, NV * 1.1, NM * 16 / 2.1);
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512, 1024}) {
        run<0, 1>(thr, d); run<4, 1>(thr, d); run<8, 1>(thr, d); run<13, 1>(thr, d); run<16, 1>(thr, d); run<24, 1>(thr, d);
        run<26, 2>(thr, d); run<39, 3>(thr, d); run<104, 0>(thr, d); run<208, 0>(thr, d); run<0, 16>(thr, d); run<104, 8>(thr, d); run<208, 16>(thr, d); run<416, 32>(thr, d);
    }
    return 0;
}
