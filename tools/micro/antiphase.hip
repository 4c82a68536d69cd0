// Two waves per SIMD, each alternating a VALU phase (NV v_fma_f32) and an MFMA burst (NM MFMAs with LDS-fed B
// fragments, like the pooling kernel).  IN-PHASE: both waves do the same phase at the same time (barrier per
// period); ANTI-PHASE: waves 4-7 run half a period shifted, so a SIMD always has one wave in each phase.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, int NM>
__device__ __forceinline__ void valu_phase(float (&x)[16], float a, float b) {
#pragma unroll
    for (int j = 0; j < NV; ++j) x[j & 15] = fmaf(x[j & 15], a, b);
}
template <int NV, int NM>
__device__ __forceinline__ void mfma_phase(f4 (&c)[8], h8 A, const unsigned char *brow, int it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const h8 B = *reinterpret_cast<const h8 *>(brow + ((m + it) % 24) * 1024);
        c[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(B, A, c[m & 7], 0, 0, 0);
    }
}

template <int NV, int NM, int MODE>   // MODE 0: in phase, 1: anti-phase, 2: VALU only, 3: MFMA only
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[24576];
    for (int i = threadIdx.x; i < 24576 / 4; i += blockDim.x) reinterpret_cast<unsigned *>(lds)[i] = 0x3C003C00u;
    __syncthreads();
    const unsigned char *brow = lds + (threadIdx.x & 63) * 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    h8 A;
    for (int i = 0; i < 8; ++i) A[i] = (_Float16)1;
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0, 0, 0, 0};
    const bool shifted = MODE == 1 && wave >= 4;
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        if (MODE == 2) { valu_phase<NV, NM>(x, a, b); continue; }
        if (MODE == 3) { mfma_phase<NV, NM>(c, A, brow, i); continue; }
        if (!shifted) { valu_phase<NV, NM>(x, a, b); mfma_phase<NV, NM>(c, A, brow, i); }
        else { mfma_phase<NV, NM>(c, A, brow, i); valu_phase<NV, NM>(x, a, b); }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += c[i][i & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NM, int MODE>
void run(float *d) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NM, MODE>), dim3(blocks), dim3(512), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NM, MODE>), dim3(blocks), dim3(512), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const char *names[] = {"in-phase  ", "anti-phase", "VALU only ", "MFMA only "};
    printf("NV=%d NM=%d %s: %.1f ns per period per SIMD (2 waves)\n", NV, NM, names[MODE], ms * 1e6 / iters);
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    run<600, 63, 2>(d); run<600, 63, 3>(d); run<600, 63, 0>(d); run<600, 63, 1>(d);
    run<300, 32, 2>(d); run<300, 32, 3>(d); run<300, 32, 0>(d); run<300, 32, 1>(d);
    return 0;
}
