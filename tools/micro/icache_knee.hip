// VALU issue rate vs loop-body size on gfx950: same total FMA count, loop bodies of 0.5 .. 64 KiB of code.
// (If instruction fetch limits large bodies, the pooling kernel's row loop should be kept small.)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BODY>   // FMAs per loop iteration (8 bytes each)
__global__ void k(float *out, int total, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int i = 0; i < total / BODY; ++i) {
#pragma unroll
        for (int j = 0; j < BODY; ++j) x[j & 15] = fmaf(x[j & 15], a, b);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BODY>
void run(int threads, float *d) {
    const int total = 1 << 21, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<BODY>, dim3(blocks), dim3(threads), 0, 0, d, BODY * 4, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<BODY>, dim3(blocks), dim3(threads), 0, 0, d, total, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("body %5d FMAs (%5.1f KiB)  %4d thr/CU: %.2f ns per instruction per SIMD\n", BODY, BODY * 8 / 1024.0, threads,
           ms * 1e6 / ((double)total * (threads / 256.0)));
}

int main() {
    float *d; (void)hipMalloc(&d, 1 << 22);
    for (int thr : {256, 512}) {
        run<64>(thr, d); run<256>(thr, d); run<512>(thr, d); run<1024>(thr, d); run<2048>(thr, d); run<4096>(thr, d); run<8192>(thr, d);
    }
    return 0;
}
