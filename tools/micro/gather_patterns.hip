// Texture-path cost of bilinear patch sampling on gfx950, by lane mapping and load width.  A wave of the describe kernel
// owns 16 patches and consumes one patch row (16 x 32 pixels = 512 samples, 8 per lane) per step; this times the gathers
// such a step needs when the taps come straight from a (mirror-padded) pyramid:
//   ROWSEG   lane (p, q) samples pixels 8q..8q+7 of patch p's row (the describe kernel's own lane map)
//   ROWLIN   lane l samples pixel l & 31 of patch 2j + (l >> 5) in step j = 0..7 (neighbouring lanes = neighbouring pixels)
//   BLOCK8   the stand-alone sampler's map: an 8 x 8 pixel block of ONE patch per load (not row-synchronous: for reference)
// each with four dword loads per sample (t00, t10, t01, t11) or two dwordx2 loads (t00|t10, t01|t11).
// Frames: 640 x 480 or 1920 x 1080 pyramids, keypoint sizes log-uniform in [1.64, 52] like the detector's.
//   hipcc --offload-arch=gfx950 -O3 gather_patterns.hip -o gather_patterns
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Kp { float cx, cy, ax, ay; int base, pitch, pad0, pad1; };   // centre at its level, step vector (rem cos, rem sin)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WIDE>
__device__ __forceinline__ float tap(const float *__restrict__ img, const Kp &k, float dx, float dy) {
    const float sx = k.cx + dx * k.ax - dy * k.ay, sy = k.cy + dx * k.ay + dy * k.ax;
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float fx = sx - x0f, fy = sy - y0f;
    const float *t = img + k.base + (int)y0f * k.pitch + (int)x0f;
    float t00, t10, t01, t11;
    if (WIDE) {
        f32x2 a, b;
        __builtin_memcpy(&a, t, 8);
        __builtin_memcpy(&b, t + k.pitch, 8);
        t00 = a.x; t10 = a.y; t01 = b.x; t11 = b.y;
    } else {
        t00 = t[0]; t10 = t[1]; t01 = t[k.pitch]; t11 = t[k.pitch + 1];
    }
    const float top = t00 * (1.f - fx) + t10 * fx, bot = t01 * (1.f - fx) + t11 * fx;
    return top * (1.f - fy) + bot * fy;
}

// mode 0 ROWSEG, 1 ROWLIN, 2 BLOCK8
template <int MODE, int WIDE>
__global__ __launch_bounds__(512) void k_gather(const float *__restrict__ img, const Kp *__restrict__ kps, int batches,
                                                 float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    float acc = 0.f;
    for (int b = blockIdx.x; b < batches; b += gridDim.x) {
        const Kp *kw = kps + ((long)b * waves + wave) * 16;
        if (MODE == 0) {
            const Kp k = kw[lane & 15];
            const int q = lane >> 4;
#pragma unroll 1
            for (int row = 0; row < 32; ++row) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += tap<WIDE>(img, k, (float)(8 * q + i - 16), (float)(row - 16));
            }
        } else if (MODE == 1) {
#pragma unroll 1
            for (int row = 0; row < 32; ++row) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const Kp k = kw[2 * j + (lane >> 5)];
                    acc += tap<WIDE>(img, k, (float)((lane & 31) - 16), (float)(row - 16));
                }
            }
        } else {
#pragma unroll 1
            for (int p = 0; p < 16; ++p) {
                const Kp k = kw[p];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc += tap<WIDE>(img, k, (float)(8 * (i & 3) + (lane & 7) - 16), (float)(8 * (i >> 2) + (lane >> 3) - 16));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int kApron = 48;
    for (int frame = 0; frame < 2; ++frame) {
        const int W = frame ? 1920 : 640, H = frame ? 1080 : 480;
        // pyramid with a 48-texel apron around every level
        std::vector<int> lw, lh, lbase, lpitch;
        long total = 0;
        for (int l = 0, w = W, h = H; w >= 2 && h >= 2; ++l, w >>= 1, h >>= 1) {
            lw.push_back(w); lh.push_back(h); lpitch.push_back(w + 2 * kApron);
            lbase.push_back((int)(total + (long)kApron * (w + 2 * kApron) + kApron));
            total += (long)(w + 2 * kApron) * (h + 2 * kApron);
        }
        const int frames = frame ? 16 : 64;   // several frames, so that the working set is not one L2-resident pyramid
        std::vector<float> img((size_t)total * frames);
        srand(1);
        for (auto &v : img) v = (float)rand() / RAND_MAX;
        const int waves = 8, batches = 2048;
        std::vector<Kp> kps((size_t)batches * waves * 16);
        for (auto &k : kps) {
            const double u = (double)rand() / RAND_MAX;
            const float size = 1.64f * powf(52.f / 1.64f, (float)u), scale = size * 24.f / 32.f;
            int l = (int)floorf(log2f(scale));
            l = l < 0 ? 0 : (l > (int)lw.size() - 1 ? (int)lw.size() - 1 : l);
            const float rem = scale / exp2f((float)l), ang = 6.2831853f * rand() / RAND_MAX;
            k.cx = (float)rand() / RAND_MAX * (lw[l] - 1);
            k.cy = (float)rand() / RAND_MAX * (lh[l] - 1);
            k.ax = rem * cosf(ang); k.ay = rem * sinf(ang);
            k.base = lbase[l] + (int)((rand() % frames) * total);
            k.pitch = lpitch[l];
            if (rem >= 2.f) { k.ax *= 1.9f / rem; k.ay *= 1.9f / rem; }   // keep the footprint inside the apron
        }
        float *d_img, *d_out; Kp *d_k;
        (void)hipMalloc(&d_img, img.size() * 4); (void)hipMalloc(&d_out, 256 * 512 * 4); (void)hipMalloc(&d_k, kps.size() * sizeof(Kp));
        (void)hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_k, kps.data(), kps.size() * sizeof(Kp), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto time = [&](auto kern, const char *name, int wv) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(256), dim3(64 * wv), 0, 0, d_img, d_k, batches * 8 / wv, d_out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double nk = (double)batches * 8 * 16;
            printf("%4dx%-4d %-14s %d waves/CU: %7.3f ms  %6.1f M keypoints/s  %6.1f ns per wave-row (512 samples) per CU-wave\n", W, H,
                   name, wv, best, nk / best / 1e3, best * 1e6 / (batches * 8.0 / 256 / wv * 32) / wv);
        };
        for (int wv : {8, 4}) {
            time(k_gather<0, 0>, "ROWSEG dword", wv);
            time(k_gather<0, 1>, "ROWSEG dwordx2", wv);
            time(k_gather<1, 0>, "ROWLIN dword", wv);
            time(k_gather<1, 1>, "ROWLIN dwordx2", wv);
            time(k_gather<2, 0>, "BLOCK8 dword", wv);
            time(k_gather<2, 1>, "BLOCK8 dwordx2", wv);
        }
        (void)hipFree(d_img); (void)hipFree(d_out); (void)hipFree(d_k);
    }
    return 0;
}
