// Does it matter how long before an MFMA its A operand was written by VALU?  (gfx950, v_mfma_f32_16x16x32_f16)
// Per group: 4 x v_cvt_pkrtz_f16_f32 build an A fragment, 12 independent v_fma_f32, 1 MFMA that consumes the
// fragment built DIST groups earlier (DIST = 0: just written; 1: one group old).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }

template <int DIST, int NV, bool MFMA, int LDSB = 0, int NACC = 4>
__global__ void k(float *out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[24576];
    for (int i = threadIdx.x; i < 24576 / 4; i += blockDim.x) reinterpret_cast<unsigned *>(lds)[i] = 0x3C003C00u;
    __syncthreads();
    const unsigned char *brow = lds + (threadIdx.x & 63) * 16;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    h8 B;
    for (int i = 0; i < 8; ++i) B[i] = (_Float16)1;
    f4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f4{0, 0, 0, 0};
    u4 fr[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    h8 Bnext = B, Bnext2 = B;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u4 f = {pk(x[0], x[1]), pk(x[2], x[3]), pk(x[4], x[5]), pk(x[6], x[7])};
            fr[u & 1] = f;
            const u4 use = DIST == 0 ? f : fr[(u + 1) & 1];
            h8 Bv = B;
            if (LDSB == 1) Bv = *reinterpret_cast<const h8 *>(brow + ((u * 3 + i) % 24) * 1024);
            if (LDSB == 2) { Bv = Bnext; Bnext = *reinterpret_cast<const h8 *>(brow + ((u * 3 + i + 1) % 24) * 1024); }
            if (LDSB == 3) { Bv = Bnext; Bnext = Bnext2; Bnext2 = *reinterpret_cast<const h8 *>(brow + ((u * 3 + i + 2) % 24) * 1024); }
            if (MFMA) c[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, use), Bv, c[u % NACC], 0, 0, 0);
            else if (LDSB) asm volatile("" ::"v"(Bv));
            else asm volatile("" ::"v"(use));
#pragma unroll
            for (int j = 0; j < NV; ++j) x[j & 15] = fmaf(x[j & 15], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < NACC; ++i) s += c[i][i & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int DIST, int NV, bool MFMA, int LDSB = 0, int NACC = 4>
void run(int threads, float *d) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<DIST, NV, MFMA, LDSB, NACC>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<DIST, NV, MFMA, LDSB, NACC>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s dist=%d NV=%2d ldsB=%d nacc=%d %4d thr/CU: %.2f ns per group per SIMD\n", MFMA ? "MFMA   " : "no-MFMA", DIST, NV, LDSB, NACC, threads,
           ms * 1e6 / ((double)iters * 8 * (threads / 256.0)));
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512}) {
        run<0, 12, false>(thr, d); run<0, 12, true>(thr, d); run<1, 12, true>(thr, d);
        run<0, 24, false>(thr, d); run<0, 24, true>(thr, d); run<1, 24, true>(thr, d);
        run<0, 12, false, 1>(thr, d); run<0, 12, true, 1>(thr, d); run<0, 12, false, 2>(thr, d); run<0, 12, true, 2>(thr, d); run<0, 12, false, 3>(thr, d); run<0, 12, true, 3>(thr, d);
    }
    return 0;
}
