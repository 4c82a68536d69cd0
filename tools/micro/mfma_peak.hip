// What does a bare v_mfma_f32_32x32x16_f16 stream sustain on this part, and does it depend on the operand data?
// (The matcher's scan loop reaches 65 % of the nominal 2.5 PF with everything but the MFMAs removed.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned *seed, float *out, int iters, int mode) {
    h8 A[8], B[8];
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 8; ++i) {
            unsigned r = seed[(threadIdx.x * 64 + s * 8 + i) & 4095];
            float a = mode == 0 ? 0.f : (mode == 1 ? (float)(r & 0xFFFF) / 65536.f - 0.5f : 1.0f);
            float b = mode == 0 ? 0.f : (mode == 1 ? (float)(r >> 16) / 65536.f - 0.5f : 1.0f);
            A[s][i] = (_Float16)a; B[s][i] = (_Float16)b;
        }
    f16v c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], B[s], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(B[s], A[s], c1, 0, 0, 0);
        }
        if (mode == 1) { c0 *= 0.5f; }   // keep values finite without changing the instruction stream much
    }
    float t = 0;
    for (int i = 0; i < 16; ++i) t += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
    unsigned *h = (unsigned *)malloc(4096 * 4), *d; float *o;
    srand(1); for (int i = 0; i < 4096; ++i) h[i] = (unsigned)rand() * 2654435761u;
    (void)hipMalloc(&d, 4096 * 4); (void)hipMalloc(&o, 1 << 22); (void)hipMemcpy(d, h, 4096 * 4, hipMemcpyHostToDevice);
    const char *names[] = {"zeros", "random in [-0.5,0.5)", "ones"};
    for (int thr : {256, 512})
        for (int mode = 0; mode < 3; ++mode)
            for (int iters : {2000, 200000}) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                hipLaunchKernelGGL(k, dim3(256), dim3(thr), 0, 0, d, o, 10, mode);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(thr), 0, 0, d, o, iters, mode);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                const double flops = 256.0 * (thr / 64) * iters * 16.0 * 32 * 32 * 16 * 2;
                printf("%d waves/SIMD, %-22s %7d iters: %8.3f ms  %.2f PFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", thr / 256,
                       names[mode], iters, ms, flops / ms / 1e12, ms * 1e-3 * 2.4e9 / (iters * 16.0 * (thr / 256)));
            }
    return 0;
}
