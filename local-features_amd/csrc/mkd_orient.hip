// MKD descriptor path: keypoint orientation for gfx950.
//
//   orient_*   keypoint_orientation.glsl:36-171 (+ the ordered compaction that replaces its atomic append)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mkd_device.h"

namespace lfmkd {

// defined in mkd_detect.hip: exclusive scan of per-block sums, capped at max_out (the ordered compactions share it)
__global__ void cubes_scan_sums(unsigned *__restrict__ sums, long nb, unsigned long long max_out,
                                unsigned long long *__restrict__ totals);

// ---------------------------------------------------------------------------------------------
// Keypoint orientation (keypoint_orientation.glsl:36-171).  One wave per extremum, 4 per block.
// The 15x15 window (taps `step` texels apart on a-trous layer `level`) is staged in LDS; every lane owns up to
// four of its 225 texels.  The reference sums the histogram with ONE thread walking the window row-major
// (lines 114-124); here lane b sums bin b in that same order, so every bin sees the same addition sequence.
// Contraction is off in this kernel: peak decisions compare sums, and the restatement they are tested against
// (oracle/mkd_oracle.c) is built without fma contraction.
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float glsl_sign(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// shaders/atan2.glsl:19-46, the angle of point (x, y); sign(0) = 0 makes atan2(0, y != 0) return 0.
__device__ __forceinline__ float atan2_shader(float x, float y) {
#pragma clang fp contract(off)
    if (x == 0.f && y == 0.f) return 0.f;
    const bool swap = fabsf(x) < fabsf(y);
    const float a = swap ? (x / y) : (y / x);
    const float s = a * a;
    const float p = a * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f +
                    s * (0.05265332f + s * -0.0117212f)))));
    const float res = swap ? (1.5707964f * glsl_sign(a) - p) : p;
    if (x < 0.f) return 3.1415927f * (y < 0.f ? -1.f : 1.f) + res;
    return res;
}

constexpr int kOriWin = 15, kOriPx = kOriWin * kOriWin, kOriBins = 36, kOriMaxPeaks = 18;

}  // namespace

// A wave's LDS arrays are its own here (one extremum per wave): what its lanes exchange through them needs program order
// and nothing else -- the LDS executes a wave's operations in the order they were issued -- so the workgroup barrier is
// replaced by a fence at wavefront scope, which only keeps the compiler from moving LDS accesses across it.  The four
// waves of a workgroup (four extrema with different numbers of voters) then no longer wait for each other six times.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void orient_peaks(const float *__restrict__ layer0, long layer0_stride, int layer0_pitch,
                                                    const float *__restrict__ coarse, long coarse_stride,
                                                    long layer_stride, int n_layers, int w, int h,
                                                    const float *__restrict__ extrema /*[n][4]*/,
                                                    const unsigned *__restrict__ frame_of, long n_host,
                                                    const unsigned long long *__restrict__ n_dev,
                                                    float *__restrict__ angles /*[n][18]*/,
                                                    unsigned *__restrict__ counts /*[n]*/) {
#pragma clang fp contract(off)
    const long n = n_dev ? (long)*n_dev : n_host;
    if (n <= 0) return;
    __shared__ float s_patch[4][kOriPx];
    __shared__ float s_weight[4][kOriPx];
    __shared__ int s_bin[4][kOriPx];
    __shared__ float s_hist[4][kOriBins];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long kk = (long)blockIdx.x * 4 + wave;
    if (kk >= n) return;   // (whole waves leave: the kernel has no workgroup barrier)
    const bool live = true;
    const long k = kk;
    const float *ex = extrema + k * 4;
    const int kx = (int)ex[0], ky = (int)ex[1];
    const float size = ex[2];
    const float kSigmaRadius = 1.41421356237f;   // DOG_SIGMA_RADIUS_FACTOR = sqrt(2), DOG_FIRST_SCALE_SIGMA = 0.82
    int level = (int)roundf(log2f(size / (0.82f * kSigmaRadius)));
    level = level < 0 ? 0 : (level > n_layers - 1 ? n_layers - 1 : level);
    const int step = 1 << level;
    const int radius = (int)roundf(3.f * 1.5f * size / kSigmaRadius);
    const float sigma = 1.5f * size / kSigmaRadius;
    const unsigned f = frame_of ? frame_of[k] : 0u;
    const float *img = level == 0 ? layer0 + f * layer0_stride : coarse + f * coarse_stride + (level - 1) * layer_stride;
    const int ipitch = level == 0 ? layer0_pitch : w;   // layer 0 is pyramid level 0, stored with its apron

    float *patch = s_patch[wave], *weight = s_weight[wave];
    int *bin = s_bin[wave];
    float *hist = s_hist[wave];
    unsigned ingrad = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (i < kOriPx) {
            const int ly = i / kOriWin, lx = i - ly * kOriWin;
            const int xd = (lx - 7) * step, yd = (ly - 7) * step;
            const int xi = kx + xd, yi = ky + yd;
            // valid_px admits y == height (line 73); that row reads as 0 like every out-of-image load
            const bool valid = 0 <= xi && xi < w && 0 <= yi && yi <= h;
            patch[i] = (valid && yi < h) ? img[(size_t)yi * ipitch + xi] : 0.f;
            const bool inner = lx > 0 && lx < kOriWin - 1 && ly > 0 && ly < kOriWin - 1;
            if (valid && inner && abs(xd) <= radius && abs(yd) <= radius) ingrad |= 1u << j;
        }
    }
    wave_sync();
    int voters = 0;   // uniform over the wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        {
            int b = kOriBins + 1;
            float wgt = 0.f;
            if (i < kOriPx && (ingrad & (1u << j))) {
                const float gx = patch[i + 1] - patch[i - 1];
                const float gy = patch[i - kOriWin] - patch[i + kOriWin];
                if (gx != 0.f || gy != 0.f) {
                    const int ly = i / kOriWin, lx = i - ly * kOriWin;
                    const float fx = (float)(lx - 7) * (float)step, fy = (float)(ly - 7) * (float)step;
                    const float dist = fx * fx + fy * fy;
                    wgt = expf(-dist / (2.f * sigma * sigma)) * sqrtf(gx * gx + gy * gy);
                    const int rb = (int)roundf(atan2_shader(gx, gy) * ((float)kOriBins / (2.f * 3.1415927f)));
                    b = rb < 0 ? rb + kOriBins : (rb >= kOriBins ? rb - kOriBins : rb);
                }
            }
            // texels that vote (bin < 36) are appended in texel order: chunk j holds texels 64 j .. 64 j + 63, one per
            // lane, so a ballot gives each its rank -- the histogram walk below then skips the texels that do not vote
            const unsigned long long vm = __ballot(b < kOriBins);
            if (b < kOriBins) {
                const int pos = voters + __popcll(vm & ((1ull << lane) - 1ull));
                bin[pos] = b;
                weight[pos] = wgt;
            }
            voters += __popcll(vm);
        }
    }
    wave_sync();
    // still the reference's single-thread, row-major addition order per bin.  Branch-free, eight voters' records read
    // together: a voter of another bin adds +0 (the sums are of non-negative weights: x + 0 == x bit for bit), and the
    // LDS latency is paid once per eight voters instead of twice per voter.
    float raw = 0.f;
    if (lane < kOriBins) {
        int i = 0;
        for (; i + 8 <= voters; i += 8) {
            int bb[8];
            float ww[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                bb[u] = bin[i + u];
                ww[u] = weight[i + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) raw += bb[u] == lane ? ww[u] : 0.f;
        }
        for (; i < voters; ++i) raw += bin[i] == lane ? weight[i] : 0.f;
    }
    wave_sync();
    if (lane < kOriBins) hist[lane] = raw;   // raw histogram, circular
    wave_sync();
    float hv = 0.f;
    if (lane < kOriBins) {
        auto at = [&](int b) { return hist[b < 0 ? b + kOriBins : (b >= kOriBins ? b - kOriBins : b)]; };
        hv = (at(lane - 2) + at(lane + 2)) * (1.0f / 16.0f) + (at(lane - 1) + at(lane + 1)) * (4.0f / 16.0f) +
             at(lane) * (6.0f / 16.0f);
    }
    wave_sync();
    if (lane < kOriBins) hist[lane] = hv;    // smoothed histogram
    float mx = hv;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    wave_sync();
    bool peak = false;
    float angle = 0.f;
    if (lane < kOriBins) {
        const float left = hist[lane > 0 ? lane - 1 : kOriBins - 1], right = hist[lane < kOriBins - 1 ? lane + 1 : 0];
        peak = left < hv && right < hv && mx * 0.8f <= hv;
        const float interp = (left - right) / (left - 2.f * hv + right);
        const float rbin = (float)lane + interp / 2.0f;
        const float b = rbin < 0.f ? rbin + kOriBins : (rbin > kOriBins ? rbin - kOriBins : rbin);
        angle = 360.0f - (360.0f / (float)kOriBins) * b;
    }
    const unsigned long long m = __ballot(peak);
    if (live) {
        if (peak) angles[k * kOriMaxPeaks + __popcll(m & ((1ull << lane) - 1ull))] = angle;
        if (lane == 0) counts[k] = (unsigned)__popcll(m);
    }
}

// Ordered compaction: keypoint list = for each extremum in index order, its peaks in bin order.  A workgroup takes 1024
// extrema; its base is the sum of the counts before them, which it adds up itself (n is a few thousand per frame: a few
// loads per thread, cheaper than a launch that scans).  The last workgroup leaves totals[0] = keypoints written,
// totals[1] = keypoints dropped because max_out was reached.
__global__ __launch_bounds__(1024) void orient_compact(const float *__restrict__ extrema,
                                                       const unsigned *__restrict__ frame_of,
                                                       const float *__restrict__ angles,
                                                       const unsigned *__restrict__ counts, long n_host,
                                                       const unsigned long long *__restrict__ n_dev,
                                                       float *__restrict__ kps /*[max_out][5]*/,
                                                       unsigned *__restrict__ frame_of_kp, unsigned long long max_out,
                                                       unsigned long long *__restrict__ totals) {
    __shared__ unsigned wave_sum[16], wave_base[16];
    const long n = n_dev ? (long)*n_dev : n_host;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long chunk = (long)blockIdx.x * 1024;
    unsigned mine = 0;
    for (long k = threadIdx.x; k < chunk && k < n; k += 1024) mine += counts[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) wave_base[wave] = mine;
    const long i = chunk + threadIdx.x;
    const unsigned c = i < n ? counts[i] : 0u;
    unsigned incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    unsigned before = 0, all = 0;
    unsigned long long base = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const unsigned t = wave_sum[v];
        before += v < wave ? t : 0u;
        all += t;
        base += wave_base[v];
    }
    const unsigned long long first = base + before + incl - c;
    for (unsigned j = 0; j < c; ++j) {
        const unsigned long long o = first + j;
        if (o < max_out) {
            kps[o * 5 + 0] = extrema[i * 4 + 0];
            kps[o * 5 + 1] = extrema[i * 4 + 1];
            kps[o * 5 + 2] = extrema[i * 4 + 2];
            kps[o * 5 + 3] = angles[i * kOriMaxPeaks + j];
            kps[o * 5 + 4] = extrema[i * 4 + 3];
            if (frame_of_kp) frame_of_kp[o] = frame_of ? frame_of[i] : 0u;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const unsigned long long total = base + all;
        totals[0] = total < max_out ? total : max_out;
        totals[1] = total < max_out ? 0ull : total - max_out;
    }
}

// The same ordered compaction as orient_compact for long lists (batched frames: hundreds of thousands of extrema),
// in the three-launch form of cubes_*: sums of 1024 counts, one workgroup scans the sums, every workgroup rescans its
// counts and writes its keypoints.
__global__ __launch_bounds__(1024) void orient_block_sums(const unsigned *__restrict__ counts, long n_host,
                                                          const unsigned long long *__restrict__ n_dev,
                                                          unsigned *__restrict__ sums) {
    __shared__ unsigned ws[16];
    const long n = n_dev ? (long)*n_dev : n_host;
    const long i = (long)blockIdx.x * 1024 + threadIdx.x;
    unsigned v = i < n ? counts[i] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int k = 0; k < 16; ++k) t += ws[k];
        sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void orient_scatter(const float *__restrict__ extrema,
                                                       const unsigned *__restrict__ frame_of,
                                                       const float *__restrict__ angles,
                                                       const unsigned *__restrict__ counts,
                                                       const unsigned *__restrict__ block_offsets, long n_host,
                                                       const unsigned long long *__restrict__ n_dev,
                                                       float *__restrict__ kps, unsigned *__restrict__ frame_of_kp,
                                                       unsigned long long max_out) {
    __shared__ unsigned ws[16];
    const long n = n_dev ? (long)*n_dev : n_host;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 1024 + threadIdx.x;
    const unsigned c = i < n ? counts[i] : 0u;
    unsigned incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    unsigned before = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) before += v < wave ? ws[v] : 0u;
    const unsigned long long first = (unsigned long long)block_offsets[blockIdx.x] + before + incl - c;
    for (unsigned j = 0; j < c; ++j) {
        const unsigned long long o = first + j;
        if (o < max_out) {
            kps[o * 5 + 0] = extrema[i * 4 + 0];
            kps[o * 5 + 1] = extrema[i * 4 + 1];
            kps[o * 5 + 2] = extrema[i * 4 + 2];
            kps[o * 5 + 3] = angles[i * kOriMaxPeaks + j];
            kps[o * 5 + 4] = extrema[i * 4 + 3];
            if (frame_of_kp) frame_of_kp[o] = frame_of ? frame_of[i] : 0u;
        }
    }
}

void launch_orient(const float *layer0, long layer0_stride, int layer0_pitch, const float *coarse, long coarse_stride, long layer_stride,
                   int n_layers, int w, int h, const float *extrema, const unsigned *frame_of, long n,
                   const unsigned long long *n_dev, float *angles, unsigned *counts, unsigned *sums, float *kps,
                   unsigned *frame_of_kp, unsigned long long max_out, unsigned long long *totals, hipStream_t stream) {
    if (n > 0)
        hipLaunchKernelGGL(orient_peaks, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, layer0, layer0_stride, layer0_pitch,
                           coarse, coarse_stride, layer_stride, n_layers, w, h, extrema, frame_of, n, n_dev, angles,
                           counts);
    if (n <= 8192 || !sums) {   // a few thousand extrema (one frame): every workgroup adds up the counts before its own
        const long nbc = n > 0 ? (n + 1023) / 1024 : 1;
        hipLaunchKernelGGL(orient_compact, dim3((unsigned)nbc), dim3(1024), 0, stream, extrema, frame_of,
                           (const float *)angles, (const unsigned *)counts, n, n_dev, kps, frame_of_kp, max_out, totals);
        return;
    }
    const long nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(orient_block_sums, dim3((unsigned)nb), dim3(1024), 0, stream, (const unsigned *)counts, n, n_dev,
                       sums);
    hipLaunchKernelGGL(cubes_scan_sums, dim3(1), dim3(1024), 0, stream, sums, nb, max_out, totals);
    hipLaunchKernelGGL(orient_scatter, dim3((unsigned)nb), dim3(1024), 0, stream, extrema, frame_of, (const float *)angles,
                       (const unsigned *)counts, (const unsigned *)sums, n, n_dev, kps, frame_of_kp, max_out);
}

}  // namespace lfmkd
