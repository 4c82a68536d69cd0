// HBM bandwidth of the pooling kernel's patch-read pattern alone: per wave, per step, one 128-B row of each of
// its 16 patches by LDS-DMA (2 x global_load_lds_dwordx4), 32+ steps per batch, persistent 8-wave workgroups.
// MODE 0: row-by-row as the kernel does; MODE 1: the same bytes as whole 4 KiB patches (4 rows per DMA pair x 8).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, bool BARRIER>
__global__ __launch_bounds__(512) void k(const float *__restrict__ patches, long n, float *out) {
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[8 * 8 * 2048];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, q = lane >> 4;
    unsigned char *ring = s_mem + wave * 8 * 2048;
    const long nbatch = n / 128;
    float acc = 0.f;
    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const float *src = patches + (batch * 128 + wave * 16 + p) * 1024 + 4 * q;
        for (int g = 0; g < 32; ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (BARRIER) __syncthreads();
            const float *rowp = MODE == 0 ? src + g * 32 : patches + (batch * 128 + wave * 16 + (g >> 1)) * 1024 + (g & 1) * 512 + lane * 4;
            // MODE 1: lane-linear 1 KiB pieces of one patch (2 KiB per step = same bytes per step)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowp),
                                             (__attribute__((address_space(3))) void *)(ring + (g & 7) * 2048), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowp + (MODE == 0 ? 16 : 256)),
                                             (__attribute__((address_space(3))) void *)(ring + (g & 7) * 2048 + 1024), 16, 0, 0);
            acc += *reinterpret_cast<const float *>(ring + ((g + 4) & 7) * 2048 + lane * 4);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, bool BARRIER>
void run(const float *d_p, long n, float *d_o) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, BARRIER>), dim3(256), dim3(512), 0, 0, d_p, n, d_o);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, BARRIER>), dim3(256), dim3(512), 0, 0, d_p, n, d_o);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("mode %d (%s) barrier=%d: %.3f ms for %.2f GB -> %.2f TB/s\n", MODE, MODE == 0 ? "128-B rows of 16 patches" : "2 KiB of one patch", (int)BARRIER, ms,
           n * 4096.0 / 1e9, n * 4096.0 / ms / 1e9);
}

int main() {
    const long n = 1 << 20;
    float *d_p, *d_o;
    (void)hipMalloc(&d_p, n * 4096);
    (void)hipMemset(d_p, 0, n * 4096);
    (void)hipMalloc(&d_o, 256 * 512 * 4);
    run<0, true>(d_p, n, d_o); run<0, false>(d_p, n, d_o); run<1, true>(d_p, n, d_o); run<1, false>(d_p, n, d_o);
    return 0;
}
