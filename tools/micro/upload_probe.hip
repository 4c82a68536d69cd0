// How a host image reaches the device fastest (round 5, NOTEBOOK.md section 12): the reference's callers hand detect_top_n a
// pageable host array (examples/match_images/src/main.rs:44-76), 4096 x 3072 f32 = 50 MB in its own benchmark
// (benches/bench.rs:41-112).  Measures, per strategy, wall time from the call to "data usable on the device":
//   pageable-whole     one hipMemcpyAsync from the caller's (pageable) array + sync            (what lf_mkd_detect did)
//   pageable-bands     the same in B row bands, back to back on one stream
//   pinned-whole/bands the same from hipHostMalloc memory (the DMA rate without the runtime's pinning / staging)
//   stage-T            T host threads copy bands into a pinned double buffer, each band DMA'd as it lands
//   register           hipHostRegister the caller's array, copy, unregister
//   zero-copy          a kernel reads the registered / pinned host array itself (one pass, 16 B per lane)
// for f32 (4 B/px) and u8 (1 B/px).  build: hipcc --offload-arch=gfx950 -O3 -pthread upload_probe.hip -o upload_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_pull(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_touch(const float *__restrict__ in, float *__restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 0.5f;
}

// a pool of T workers that split one memcpy
struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    const char *src = nullptr;
    char *dst = nullptr;
    size_t bytes = 0;
    int gen = 0, left = 0;
    bool quit = false;
    explicit Pool(int t) {
        for (int i = 0; i < t; ++i) th.emplace_back([this, i, t] {
            int seen = 0;
            for (;;) {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                const char *s = src; char *d = dst; size_t b = bytes;
                lk.unlock();
                const size_t per = (b / t + 63) / 64 * 64, lo = std::min(b, per * i), hi = std::min(b, lo + per);
                if (hi > lo) memcpy(d + lo, s + lo, hi - lo);
                lk.lock();
                if (--left == 0) done_cv.notify_one();
            }
        });
    }
    void copy(char *d, const char *s, size_t b) {
        std::unique_lock<std::mutex> lk(m);
        src = s; dst = d; bytes = b; left = (int)th.size(); ++gen;
        cv.notify_all();
        done_cv.wait(lk, [&] { return left == 0; });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
};

template <typename F>
static double median_ms(int reps, F f) {
    std::vector<double> t;
    for (int i = 0; i < reps + 2; ++i) {
        const double a = now_ms();
        f();
        const double b = now_ms();
        if (i >= 2) t.push_back(b - a);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const int W = 4096, H = 3072;
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int bpp : {4, 1}) {
        const size_t bytes = (size_t)W * H * bpp;
        printf("---- %d x %d, %d B/px = %.1f MB\n", W, H, bpp, bytes / 1e6);
        std::vector<char> pageable(bytes + 64);
        char *host = pageable.data();
        for (size_t i = 0; i < bytes; i += 4096) host[i] = (char)i;
        char *pinned = nullptr, *dev = nullptr, *dev2 = nullptr;
        CK(hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault));
        memset(pinned, 1, bytes);
        CK(hipMalloc((void **)&dev, bytes));
        CK(hipMalloc((void **)&dev2, bytes));
        auto report = [&](const char *what, double ms) { printf("%-34s %8.3f ms  %7.1f GB/s\n", what, ms, bytes / ms / 1e6); fflush(stdout); };
        report("pageable-whole", median_ms(10, [&] { CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }));
        report("pageable-whole hipMemcpy(sync)", median_ms(10, [&] { CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice)); }));
        for (int B : {4, 8, 16, 32}) {
            char name[64];
            snprintf(name, sizeof name, "pageable-bands %d", B);
            double t_call = 0;
            const double ms = median_ms(10, [&] {
                const size_t per = bytes / B;
                const double a = now_ms();
                for (int b = 0; b < B; ++b) CK(hipMemcpyAsync(dev + b * per, host + b * per, per, hipMemcpyHostToDevice, s));
                t_call = now_ms() - a;
                CK(hipStreamSynchronize(s));
            });
            report(name, ms);
            printf("    (host time inside the %d calls: %.3f ms)\n", B, t_call);
        }
        report("pinned-whole", median_ms(10, [&] { CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }));
        for (int B : {8, 32}) {
            char name[64];
            snprintf(name, sizeof name, "pinned-bands %d", B);
            report(name, median_ms(10, [&] {
                const size_t per = bytes / B;
                for (int b = 0; b < B; ++b) CK(hipMemcpyAsync(dev + b * per, pinned + b * per, per, hipMemcpyHostToDevice, s));
                CK(hipStreamSynchronize(s));
            }));
        }
        // two streams, alternating bands (two DMA engines?)
        report("pinned-bands 8 on 2 streams", median_ms(10, [&] {
            const size_t per = bytes / 8;
            for (int b = 0; b < 8; ++b) CK(hipMemcpyAsync(dev + b * per, pinned + b * per, per, hipMemcpyHostToDevice, b & 1 ? s2 : s));
            CK(hipStreamSynchronize(s));
            CK(hipStreamSynchronize(s2));
        }));
        report("host memcpy 1 thread -> pinned", median_ms(5, [&] { memcpy(pinned, host, bytes); }));
        for (int T : {2, 4, 8, 12, 16}) {
            Pool pool(T);
            char name[64];
            snprintf(name, sizeof name, "host memcpy %d threads -> pinned", T);
            report(name, median_ms(5, [&] { pool.copy(pinned, host, bytes); }));
            for (int B : {8, 16}) {
                snprintf(name, sizeof name, "stage-%d: %d bands memcpy+DMA", T, B);
                report(name, median_ms(8, [&] {
                    const size_t per = bytes / B;
                    for (int b = 0; b < B; ++b) {
                        pool.copy(pinned + b * per, host + b * per, per);
                        CK(hipMemcpyAsync(dev + b * per, pinned + b * per, per, hipMemcpyHostToDevice, s));
                    }
                    CK(hipStreamSynchronize(s));
                }));
            }
        }
        report("register + copy + unregister", median_ms(5, [&] {
            CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
            CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            CK(hipHostUnregister(host));
        }));
        report("register only (+unregister)", median_ms(5, [&] {
            CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
            CK(hipHostUnregister(host));
        }));
        {
            void *dp = nullptr;
            CK(hipHostGetDevicePointer(&dp, pinned, 0));
            for (int blocks : {64, 256, 1024}) {
                char name[64];
                snprintf(name, sizeof name, "zero-copy kernel pull, %d wgs", blocks);
                report(name, median_ms(8, [&] {
                    hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, s, (const f32x4 *)dp, (f32x4 *)dev, (long)(bytes / 16));
                    CK(hipStreamSynchronize(s));
                }));
            }
        }
        // a kernel on the other stream while the pageable copy runs: does the copy block the device?
        {
            const long n = (long)bytes / 4;
            const double alone = median_ms(8, [&] {
                for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s2, (const float *)dev2, (float *)dev2, n);
                CK(hipStreamSynchronize(s2));
            });
            printf("%-34s %8.3f ms\n", "8 device passes alone", alone);
            report("pageable-whole beside 8 passes", median_ms(8, [&] {
                for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s2, (const float *)dev2, (float *)dev2, n);
                CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
                CK(hipStreamSynchronize(s));
                CK(hipStreamSynchronize(s2));
            }));
        }
        CK(hipFree(dev)); CK(hipFree(dev2)); CK(hipHostFree(pinned));
    }
    return 0;
}
