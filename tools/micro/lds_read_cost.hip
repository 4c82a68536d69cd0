// Issue cost of ds_read_b128 (lane-linear, conflict-free: the LUT fragment reads of the describe kernel) on gfx950:
// NR reads per group, each group followed by a wait and NV independent v_fma_f32; 8 or 4 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NR, int NV>
__global__ void k(float *out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) float s[16384];   // 64 KiB
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = i;
    __syncthreads();
    const f4 *base = reinterpret_cast<const f4 *>(s) + (threadIdx.x & 63);
    f4 acc = {0, 0, 0, 0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
        f4 r[NR > 0 ? NR : 1];
#pragma unroll
        for (int j = 0; j < NR; ++j) r[j] = base[((it + j) & 63) * 64];
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NR; ++j) acc += r[j];
    }
    float t = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 8; ++i) t += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NR, int NV>
void run(int threads, float *d) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NR, NV>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR, NV>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("NR=%2d NV=%2d %4d thr/CU: %.1f ns per iteration per wave, %.1f ns per SIMD; LDS %.0f B/clk/CU at 2.1 GHz\n", NR, NV, threads,
           ms * 1e6 / iters, ms * 1e6 / iters / (threads / 256.0), NR * 1024.0 * (threads / 64) / (ms * 1e6 / iters) / 2.1);
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512}) {
        run<0, 16>(thr, d); run<4, 0>(thr, d); run<8, 0>(thr, d); run<4, 16>(thr, d); run<8, 16>(thr, d); run<8, 64>(thr, d); run<0, 64>(thr, d);
    }
    return 0;
}
