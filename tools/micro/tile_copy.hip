// How fast a batch of small frames (256 x 640 x 480 f32) streams through the chip by workgroup-to-texel mapping: what the
// pyramid / a-trous / scan kernels of the detector rows can expect from HBM (NOTEBOOK.md section 10).
//   linear   one dword per lane, consecutive workgroups consecutive kilobytes
//   tile     256 columns x R rows per workgroup (the kernels' mapping: grid (ceil(w/256), ceil(h/R), frames)), each row a
//            1 KB segment, `taps` overlapping loads per output (columns x-2 .. x+2 clamped) summed
//   rows     whole rows: a workgroup owns R consecutive rows of a frame (contiguous in memory), threads stride the row
//   tile4    like tile, 16 bytes per lane (64 lanes = one 1 KB row segment), R rows per wave in flight
// build: hipcc --offload-arch=gfx950 -O3 tile_copy.hip -o tile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_linear(const float *__restrict__ in, float *__restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

template <int R, int TAPS>
__global__ __launch_bounds__(256) void k_tile(const float *__restrict__ in, float *__restrict__ out, int w, int h) {
    const long f = (long)blockIdx.z * w * h;
    const int xr = blockIdx.x * 256 + threadIdx.x, x = xr < w ? xr : w - 1, y0 = blockIdx.y * R;
    int xi[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) { int v = x + t - TAPS / 2; xi[t] = v < 0 ? 0 : (v > w - 1 ? w - 1 : v); }
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
        const float *row = in + f + (long)y * w;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) s += row[xi[t]];
        if (xr < w) out[f + (long)y * w + xr] = s;
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_rows(const float *__restrict__ in, float *__restrict__ out, int w, int h) {
    const long f = (long)blockIdx.z * w * h;
    const int y0 = blockIdx.y * R;
    const int n = min(R, h - y0) * w;   // contiguous block
    const float *src = in + f + (long)y0 * w;
    float *dst = out + f + (long)y0 * w;
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

template <int R>
__global__ __launch_bounds__(256) void k_rows4(const float *__restrict__ in, float *__restrict__ out, int w, int h) {
    const long f = (long)blockIdx.z * w * h;
    const int y0 = blockIdx.y * R;
    const int n4 = min(R, h - y0) * w / 4;
    const float4 *src = reinterpret_cast<const float4 *>(in + f + (long)y0 * w);
    float4 *dst = reinterpret_cast<float4 *>(out + f + (long)y0 * w);
    for (int i = threadIdx.x; i < n4; i += 256) dst[i] = src[i];
}

// 64 lanes x 16 bytes = a 256-column row segment; a wave takes rows wave, wave + 4, ... of the tile
template <int R>
__global__ __launch_bounds__(256) void k_tile4(const float *__restrict__ in, float *__restrict__ out, int w, int h) {
    const long f = (long)blockIdx.z * w * h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 256 + 4 * lane, y0 = blockIdx.y * R;
    if (x >= w) return;
    float4 v[R / 4];
#pragma unroll
    for (int j = 0; j < R / 4; ++j) {
        const int y = min(y0 + wave + 4 * j, h - 1);
        v[j] = *reinterpret_cast<const float4 *>(in + f + (long)y * w + x);
    }
#pragma unroll
    for (int j = 0; j < R / 4; ++j) {
        const int y = y0 + wave + 4 * j;
        if (y < h) *reinterpret_cast<float4 *>(out + f + (long)y * w + x) = v[j];
    }
}

int main() {
    const int w = 640, h = 480, frames = 256;
    const long n = (long)w * h * frames;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) launch();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %8.1f us  %6.2f TB/s (read + write of %.0f MB each)\n", name, ms * 1e3, 2.0 * n * 4 / ms / 1e9, n * 4 / 1e6);
    };
    run("linear dword", [&] { hipLaunchKernelGGL(k_linear, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); });
    run("tile 256x12, 1 tap", [&] { hipLaunchKernelGGL((k_tile<12, 1>), dim3(3, 40, frames), dim3(256), 0, 0, a, b, w, h); });
    run("tile 256x12, 5 taps", [&] { hipLaunchKernelGGL((k_tile<12, 5>), dim3(3, 40, frames), dim3(256), 0, 0, a, b, w, h); });
    run("tile 256x24, 5 taps", [&] { hipLaunchKernelGGL((k_tile<24, 5>), dim3(3, 20, frames), dim3(256), 0, 0, a, b, w, h); });
    run("tile 256x48, 1 tap", [&] { hipLaunchKernelGGL((k_tile<48, 1>), dim3(3, 10, frames), dim3(256), 0, 0, a, b, w, h); });
    run("rows x12 dword", [&] { hipLaunchKernelGGL((k_rows<12>), dim3(1, 40, frames), dim3(256), 0, 0, a, b, w, h); });
    run("rows x12 dwordx4", [&] { hipLaunchKernelGGL((k_rows4<12>), dim3(1, 40, frames), dim3(256), 0, 0, a, b, w, h); });
    run("rows x24 dwordx4", [&] { hipLaunchKernelGGL((k_rows4<24>), dim3(1, 20, frames), dim3(256), 0, 0, a, b, w, h); });
    run("tile4 256x16 dwordx4", [&] { hipLaunchKernelGGL((k_tile4<16>), dim3(3, 30, frames), dim3(256), 0, 0, a, b, w, h); });
    run("tile4 256x32 dwordx4", [&] { hipLaunchKernelGGL((k_tile4<32>), dim3(3, 15, frames), dim3(256), 0, 0, a, b, w, h); });
    return 0;
}
