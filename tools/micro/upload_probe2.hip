// Second upload probe (round 5, NOTEBOOK.md section 12): can a kernel pull the caller's image over PCIe itself?
// upload_probe showed a copy kernel reading pinned host memory at the DMA engines' rate (56 GB/s) with 64 workgroups.  The
// caller's array is pageable, so it has to be registered first: what does that cost on FRESH memory (upload_probe reused one
// array the runtime had already pinned for its own staging), and does the pull run at full rate from registered memory?
// Then the shape the library would use: the pull cut into B bands on a side branch of a hipGraph, a stand-in compute kernel
// per band on the main branch (waits for its band), against one whole pull followed by the same compute.
// build: hipcc --offload-arch=gfx950 -O3 upload_probe2.hip -o upload_probe2
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// pull n16 16-byte words from (host) memory; `src_slot` holds the source pointer (a mailbox in pinned host memory, so that a
// recorded graph can be pointed at a new image without touching its nodes)
__global__ __launch_bounds__(256) void k_pull(const u32x4 *const *src_slot, long first, long n16, u32x4 *__restrict__ out) {
    const u32x4 *in = *src_slot + first;
    out += first;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) out[i] = in[i];
}
// u8 -> f32 / 255 (true division) while pulling: 16 pixels per lane
__global__ __launch_bounds__(256) void k_pull_u8(const u32x4 *const *src_slot, long first, long n16, float *__restrict__ out) {
    const u32x4 *in = *src_slot + first;
    out += first * 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const u32x4 v = in[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 f;
            f.x = (float)(w[j] & 255u) / 255.0f;
            f.y = (float)((w[j] >> 8) & 255u) / 255.0f;
            f.z = (float)((w[j] >> 16) & 255u) / 255.0f;
            f.w = (float)(w[j] >> 24) / 255.0f;
            *reinterpret_cast<f32x4 *>(out + i * 16 + 4 * j) = f;
        }
    }
}
// stand-in for a band's level-0 work: two passes over the band's rows in HBM
__global__ __launch_bounds__(256) void k_work(const float *__restrict__ in, float *__restrict__ out, long first, long n) {
    const long i = first + (long)blockIdx.x * 256 + threadIdx.x;
    if (i < first + n) out[i] = in[i] * 0.5f + 1.f;
}

template <typename F>
static double median_of(std::vector<double> &t) { std::sort(t.begin(), t.end()); return t[t.size() / 2]; }

int main() {
    const int W = 4096, H = 3072;
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const u32x4 **slot = nullptr;      // the mailbox
    CK(hipHostMalloc((void **)&slot, 64, hipHostMallocDefault));
    const u32x4 *const *d_slot = nullptr;
    CK(hipHostGetDevicePointer((void **)&d_slot, slot, 0));
    for (int bpp : {4, 1}) {
        const size_t bytes = (size_t)W * H * bpp;
        printf("---- %d x %d, %d B/px = %.1f MB, FRESH pageable memory for every repetition\n", W, H, bpp, bytes / 1e6);
        char *dev = nullptr;
        float *dev_f = nullptr, *dev_g = nullptr;
        CK(hipMalloc((void **)&dev, (size_t)W * H * 4));
        CK(hipMalloc((void **)&dev_f, (size_t)W * H * 4));
        CK(hipMalloc((void **)&dev_g, (size_t)W * H * 4));
        std::vector<double> t_reg, t_pull, t_unreg, t_all, t_memcpy, t_memcpy2;
        for (int rep = 0; rep < 7; ++rep) {
            char *host = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (host == MAP_FAILED) { printf("mmap failed\n"); return 1; }
            for (size_t i = 0; i < bytes; i += 64) host[i] = (char)(i >> 6);     // touched: the pages exist
            const double a = now_ms();
            CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
            const double b = now_ms();
            void *dp = nullptr;
            CK(hipHostGetDevicePointer(&dp, host, 0));
            *slot = (const u32x4 *)dp;
            if (bpp == 4) hipLaunchKernelGGL(k_pull, dim3(64), dim3(256), 0, s, d_slot, 0L, (long)(bytes / 16), (u32x4 *)dev);
            else hipLaunchKernelGGL(k_pull_u8, dim3(64), dim3(256), 0, s, d_slot, 0L, (long)(bytes / 16), dev_f);
            CK(hipStreamSynchronize(s));
            const double c = now_ms();
            CK(hipHostUnregister(host));
            const double d = now_ms();
            if (rep >= 2) { t_reg.push_back(b - a); t_pull.push_back(c - b); t_unreg.push_back(d - c); t_all.push_back(d - a); }
            munmap(host, bytes);
            // the runtime's own path on fresh memory
            host = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            for (size_t i = 0; i < bytes; i += 64) host[i] = (char)(i >> 6);
            const double e = now_ms();
            CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            const double f = now_ms();
            CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            const double g = now_ms();
            if (rep >= 2) { t_memcpy.push_back(f - e); t_memcpy2.push_back(g - f); }
            munmap(host, bytes);
        }
        printf("register            %8.3f ms\n", median_of<void>(t_reg));
        printf("pull (64 wgs%s) %8.3f ms  %6.1f GB/s\n", bpp == 1 ? ", u8->f32" : "        ", median_of<void>(t_pull), bytes / median_of<void>(t_pull) / 1e6);
        printf("unregister          %8.3f ms\n", median_of<void>(t_unreg));
        printf("register+pull+unreg %8.3f ms  %6.1f GB/s\n", median_of<void>(t_all), bytes / median_of<void>(t_all) / 1e6);
        printf("hipMemcpyAsync, first time on this memory %8.3f ms  %6.1f GB/s; second time %8.3f ms\n", median_of<void>(t_memcpy),
               bytes / median_of<void>(t_memcpy) / 1e6, median_of<void>(t_memcpy2));
        fflush(stdout);

        // banded pull in a graph beside per-band compute
        char *host = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        for (size_t i = 0; i < bytes; i += 64) host[i] = (char)(i >> 6);
        CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
        void *dp = nullptr;
        CK(hipHostGetDevicePointer(&dp, host, 0));
        *slot = (const u32x4 *)dp;
        const long px = (long)W * H;
        for (int B : {1, 4, 8, 16, 32}) {
            for (int wgs : {32, 64, 128}) {
                std::vector<hipEvent_t> ev(B + 1);
                for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                hipGraph_t graph;
                hipGraphExec_t exec;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                CK(hipEventRecord(ev[B], s));
                CK(hipStreamWaitEvent(s2, ev[B], 0));
                const long n16 = (long)(bytes / 16), per = (n16 / B + 63) / 64 * 64;
                for (int b = 0; b < B; ++b) {
                    const long first = per * b, cnt = std::min(per, n16 - first);
                    if (bpp == 4) hipLaunchKernelGGL(k_pull, dim3(wgs), dim3(256), 0, s2, d_slot, first, cnt, (u32x4 *)dev_f);
                    else hipLaunchKernelGGL(k_pull_u8, dim3(wgs), dim3(256), 0, s2, d_slot, first, cnt, dev_f);
                    CK(hipEventRecord(ev[b], s2));
                    CK(hipStreamWaitEvent(s, ev[b], 0));
                    const long pfirst = first * (bpp == 4 ? 4 : 16), pcnt = cnt * (bpp == 4 ? 4 : 16);
                    for (int pass = 0; pass < 2; ++pass)
                        hipLaunchKernelGGL(k_work, dim3((unsigned)((pcnt + 255) / 256)), dim3(256), 0, s, pass ? dev_g : dev_f, pass ? dev_f : dev_g, pfirst, pcnt);
                }
                CK(hipStreamEndCapture(s, &graph));
                CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                std::vector<double> t;
                for (int rep = 0; rep < 9; ++rep) {
                    const double a = now_ms();
                    CK(hipGraphLaunch(exec, s));
                    CK(hipStreamSynchronize(s));
                    if (rep >= 2) t.push_back(now_ms() - a);
                }
                printf("graph: %2d bands, %3d pull wgs, 2 passes of work per band: %8.3f ms  %6.1f GB/s\n", B, wgs, median_of<void>(t),
                       bytes / median_of<void>(t) / 1e6);
                fflush(stdout);
                CK(hipGraphExecDestroy(exec));
                CK(hipGraphDestroy(graph));
                for (auto &e : ev) CK(hipEventDestroy(e));
            }
        }
        (void)px;
        CK(hipHostUnregister(host));
        munmap(host, bytes);
        CK(hipFree(dev)); CK(hipFree(dev_f)); CK(hipFree(dev_g));
    }
    return 0;
}
