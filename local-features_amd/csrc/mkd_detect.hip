// MKD descriptor path: scale-space detector and blob filter for gfx950.
//
//   scan_extrema      swt_sub.glsl:17-30 + scan_extrema.glsl:36-241; cubes_* = its ordered compaction
//   topk_*            the host blob filter of detect_top_n (vulkan/mod.rs:1753-1786) on the device; segments_* batch it
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>
#include <stdlib.h>

#include "mkd_device.h"

namespace lfmkd {

// ---------------------------------------------------------------------------------------------
// Detector: DoG + 3-D extremum scan + quadratic refinement + edge test (swt_sub.glsl:17-30,
// scan_extrema.glsl:36-241).  The reference works in 4x4x4 cubes with at most 8 candidates each; a cube is
// exactly one wavefront here (lane = x + 4 y + 16 z), so "which 8" and the order of the survivors are settled
// by ballots in lane order instead of atomics.  The DoG is never written to HBM: a workgroup differences the
// a-trous layers into an LDS tile (the subtraction is the same single f32 operation either way).
// Output per cube: count + up to 8 slots {x, y, size, contrast}; cubes_compact_* turn that into the ordered list.
// ---------------------------------------------------------------------------------------------
// Workgroup = a TX x 8 pixel tile (TX / 4 x 2 cubes per cube layer): the tile's DoG volume (plus a one-texel rim) is
// differenced into LDS once with row-contiguous loads, then each of the 4 waves walks its share of the cubes.  TX = 64 from
// two megapixels per launch on (a 1080p frame, a 4K frame, a batch of small frames: longer row segments, 6-8 % faster there),
// 32 below (a 640 x 480 frame: more workgroups, 15 % faster than 64).
constexpr int kScanTY = 8, kScanMaxFine = 8;

template <int kScanTX>
__global__ __launch_bounds__(256) void scan_extrema(const float *__restrict__ layer0, long layer0_stride, int layer0_pitch,
                                                    const float *__restrict__ coarse, long coarse_stride,
                                                    long layer_stride, int n_fine, int w, int h, int border,
                                                    int skip_layers, float contrast_threshold, int gx, int gy, int gz,
                                                    int aligned,
                                                    float *__restrict__ slots /*[frames*cubes][8][4]*/,
                                                    unsigned *__restrict__ counts /*[frames*cubes]*/, int by_base) {
#pragma clang fp contract(off)
    // a row of the LDS tile: the TX + 2 texels the cubes reach, inside a window of TX + 8 that starts on a multiple of four
    // texels when the layers can be read 16 bytes at a time (`aligned`: widths and strides multiples of 4)
    constexpr int kScanRowLen = kScanTX + 8, kScanPlane = (kScanTY + 2) * kScanRowLen, kRowF4 = kScanRowLen / 4;
    __shared__ __attribute__((aligned(16))) float s_dog[kScanMaxFine * kScanPlane];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned f = blockIdx.z;
    const int by = (int)blockIdx.y + by_base;   // (by_base: a launch over part of the frame's tile rows, see RowBands)
    const int tx0 = blockIdx.x * kScanTX + border, ty0 = by * kScanTY + border;   // first candidate texel
    const float *l0 = layer0 + f * layer0_stride, *cs = coarse + f * coarse_stride;
    const int seg0 = aligned ? (tx0 - 1) & ~3 : tx0 - 1;   // frame column of the window's first texel
    const int shift = tx0 - 1 - seg0;
    // fine[z] = coarse[z] - coarse[z+1] (swt_sub.glsl:24-29) for the tile and its rim; outside the frame: 0
    if (aligned) {
        // 16 bytes per lane and layer, all layers of a lane's four texels requested at once (whole row segments per request:
        // tools/micro/tile_copy.hip)
        for (int i = threadIdx.x; i < (kScanTY + 2) * kRowF4; i += 256) {
            const int yy = i / kRowF4, q4 = i - yy * kRowF4;
            const int x = seg0 + 4 * q4, y = ty0 - 1 + yy;
            const bool row_in = y >= 0 && y < h;
            f32x4 c[kScanMaxFine + 1];
            if (row_in && x >= 0 && x + 3 < w) {
                const size_t o = (size_t)y * w + x;
                c[0] = *reinterpret_cast<const f32x4 *>(l0 + (size_t)y * layer0_pitch + x);
#pragma unroll
                for (int z = 0; z < kScanMaxFine; ++z)
                    c[z + 1] = z < n_fine ? *reinterpret_cast<const f32x4 *>(cs + (size_t)z * layer_stride + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = row_in && x + e >= 0 && x + e < w;
                    const size_t o = in ? (size_t)y * w + x + e : 0;
                    c[0][e] = in ? l0[(size_t)y * layer0_pitch + x + e] : 0.f;
#pragma unroll
                    for (int z = 0; z < kScanMaxFine; ++z) c[z + 1][e] = (in && z < n_fine) ? cs[(size_t)z * layer_stride + o] : 0.f;
                }
            }
#pragma unroll
            for (int z = 0; z < kScanMaxFine; ++z)
                if (z < n_fine) *reinterpret_cast<f32x4 *>(&s_dog[z * kScanPlane + yy * kScanRowLen + 4 * q4]) = c[z] - c[z + 1];
        }
    } else {
        for (int i = threadIdx.x; i < (kScanTY + 2) * (kScanTX + 2); i += 256) {
            const int yy = i / (kScanTX + 2), xx = i - yy * (kScanTX + 2);
            const int x = tx0 - 1 + xx, y = ty0 - 1 + yy;
            const bool in = x >= 0 && x < w && y >= 0 && y < h;
            const size_t o = in ? (size_t)y * w + x : 0;
            float c[kScanMaxFine + 1];   // all layers of this texel requested at once
            c[0] = in ? l0[(size_t)y * layer0_pitch + x] : 0.f;   // layer 0 = pyramid level 0, stored with its apron
#pragma unroll
            for (int z = 0; z < kScanMaxFine; ++z) c[z + 1] = (in && z < n_fine) ? cs[(size_t)z * layer_stride + o] : 0.f;
#pragma unroll
            for (int z = 0; z < kScanMaxFine; ++z)
                if (z < n_fine) s_dog[z * kScanPlane + yy * kScanRowLen + xx] = c[z] - c[z + 1];
        }
    }
    __syncthreads();
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    const int b1 = border > 1 ? border : 1;
    const int ncubes = gx * gy * gz;
    const unsigned long long below = (1ull << lane) - 1ull;
    constexpr int kCX = kScanTX / 4, kCY = kScanTY / 4;   // cubes of a tile per cube layer
    for (int q = wave; q < kCX * kCY * gz; q += 4) {   // cube q of this tile: (cube layer, cube row, cube column)
        const int qz = q / (kCX * kCY), qy = (q / kCX) % kCY, qx = q % kCX;
        const int cx = blockIdx.x * kCX + qx, cy = by * kCY + qy;
        if (cx >= gx || cy >= gy) continue;      // uniform per wave
        const int x = tx0 + qx * 4 + lx, y = ty0 + qy * 4 + ly, z = qz * 4 + lz + 1 + skip_layers;
        const bool inside = !(x < b1 || x >= w - b1 || y < b1 || y >= h - b1 || z <= 0 || z >= n_fine - 1);
        // LDS index of this voxel; out-of-range lanes are parked on a valid interior voxel and never become candidates
        const int c = (inside ? z : 1) * kScanPlane + (qy * 4 + ly + 1) * kScanRowLen + (qx * 4 + lx + 1) + shift;
        auto at = [&](int dz, int dy, int dx) { return s_dog[c + dz * kScanPlane + dy * kScanRowLen + dx]; };
        const float val = s_dog[c];
        // a candidate needs |val| above the contrast threshold (line 95): a cube without such a voxel -- most cubes of a
        // natural image -- is done after this one read per lane.  (Round 5 measured the kernel on a 12.6 MP photograph --
        // 5.97e7 vector wave-instructions, 43 % of its LDS cycles two-way bank conflicts of this cube-shaped read -- and rebuilt
        // the loop on that: the tile padded to a conflict-free pitch, a wave's eight centre voxels read and tested together,
        // only the hot cubes walked.  Same extrema, no faster: 4K frame 74 us against 71, the photograph 135 against 120 in two
        // bands.  Reverted; tools/pmc_detect_host.sh has the counters.)
        const bool hot = inside && fabsf(val) > contrast_threshold;
#ifdef LF_SCAN_ABLATE_HOT   // timing-only build: every cube taken for cold
        if (true) {
#else
        if (__ballot(hot) == 0ull) {   // uniform
#endif
            if (lane == 0) counts[(size_t)f * ncubes + ((size_t)qz * gy + cy) * gx + cx] = 0u;
            continue;
        }
        // sign(val) val >= sign(val) neighbour for all 26 neighbours (lines 97-124)  <=>  val >= their maximum when
        // val > 0, val <= their minimum when val < 0 (multiplying by +-1 is exact): 13 max3 + 13 min3 instead of 26
        // multiply-compare-and chains, and no divergence
        float nmax = -INFINITY, nmin = INFINITY;
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
                    if (dz || dy || dx) {
                        const float v = at(dz, dy, dx);
                        nmax = fmaxf(nmax, v);
                        nmin = fminf(nmin, v);
                    }
        const bool cand = hot && (val > 0.f ? val >= nmax : val <= nmin);
        const unsigned long long cm = __ballot(cand);
        bool emit = false;
        float ox = 0.f, oy = 0.f, size = 0.f, contrast = 0.f;
        if (cand && __popcll(cm & below) < 8) {   // max_wg_extrema = 8 (scan_extrema.glsl:28)
            // Quadratic fit around the voxel (axes: s = scale layer, y, x).  The extremum test downstream decides on the
            // last bits of these values, and the oracle evaluates the reference's formulas in the reference's order, so the
            // ORDER OF OPERATIONS below is load-bearing (fma contraction is off in this file): central differences divide
            // the difference, second differences subtract twice the centre last, a mixed difference runs
            // (+u+v) - (-u+v) - (+u-v) + (-u-v) left to right, the determinant is the five-term cofactor expansion left to
            // right, every adjugate entry is one 2x2 minor divided by the determinant, and each offset negates a
            // three-term dot product summed left to right (scan_extrema.glsl:167-197).
            struct Axis { int s, y, x; };
            constexpr Axis kS{1, 0, 0}, kY{0, 1, 0}, kX{0, 0, 1};
            auto tap = [&](int cu, Axis u, int cv, Axis v) {
                return at(cu * u.s + cv * v.s, cu * u.y + cv * v.y, cu * u.x + cv * v.x);
            };
            auto first = [&](Axis u) { return (tap(1, u, 0, u) - tap(-1, u, 0, u)) / 2.0f; };
            const float twice = val * 2.0f;
            auto second = [&](Axis u) { return tap(1, u, 0, u) + tap(-1, u, 0, u) - twice; };
            auto mixed = [&](Axis u, Axis v) {
                return (tap(1, u, 1, v) - tap(-1, u, 1, v) - tap(1, u, -1, v) + tap(-1, u, -1, v)) / 4.0f;
            };
            const float g_s = first(kS), g_y = first(kY), g_x = first(kX);
            // symmetric hessian: diagonal, then the three mixed terms (the y-x one differences x first)
            const float h_ss = second(kS), h_yy = second(kY), h_xx = second(kX);
            const float h_sy = mixed(kS, kY), h_sx = mixed(kS, kX), h_yx = mixed(kX, kY);
            const float det = h_ss * h_yy * h_xx - h_ss * h_yx * h_yx - h_sy * h_sy * h_xx + 2.f * h_sy * h_sx * h_yx -
                              h_sx * h_sx * h_yy;
            auto minor_over_det = [&](float a, float b, float c, float d) { return (a * b - c * d) / det; };
            const float i_ss = minor_over_det(h_yy, h_xx, h_yx, h_yx);
            const float i_sy = minor_over_det(h_sx, h_yx, h_sy, h_xx);
            const float i_sx = minor_over_det(h_sy, h_yx, h_sx, h_yy);
            const float i_yy = minor_over_det(h_ss, h_xx, h_sx, h_sx);
            const float i_yx = minor_over_det(h_sy, h_sx, h_ss, h_yx);
            const float i_xx = minor_over_det(h_ss, h_yy, h_sy, h_sy);
            const float os = -(i_ss * g_s + i_sy * g_y + i_sx * g_x);
            oy = -(i_sy * g_s + i_yy * g_y + i_yx * g_x);
            ox = -(i_sx * g_s + i_yx * g_y + i_xx * g_x);
            // |offset| > 0.5 in any direction: the shader moves x, y, z and emits nothing (lines 200-203).
            // A singular hessian gives NaN offsets; the shader would emit NaN coordinates, here it is dropped.
            const bool within = fabsf(ox) <= 0.5f && fabsf(oy) <= 0.5f && fabsf(os) <= 0.5f;
            const float interp = os * g_s + oy * g_y + ox * g_x;
            contrast = fabsf(val + interp / 2.0f);
            // edge response from the spatial 2x2 block of the hessian (lines 217-224)
            const float trace2 = (h_yy + h_xx) * (h_yy + h_xx);
            const float cmv = 1.f - 4.f * (h_yy * h_xx - h_yx * h_yx) / trace2;
            emit = within && trace2 != 0.f && !(0.7f <= cmv && cmv <= 1.5f);
            size = 0.82f * 1.41421356237f * exp2f((float)z + os);
        }
        const unsigned long long em = __ballot(emit);
        const size_t g = (size_t)f * ncubes + ((size_t)qz * gy + cy) * gx + cx;
        if (emit) {
            float *o = slots + (g * 8 + __popcll(em & below)) * 4;
            o[0] = (float)x + ox;
            o[1] = (float)y + oy;
            o[2] = size;
            o[3] = contrast;
        }
        if (lane == 0) counts[g] = (unsigned)__popcll(em);
    }
}

// (Round 4 tried to drop the first of the two launches below, twice, and measured both on a 4K frame: an atomic per
//  extremum-holding cube inside the scan -- scan 71 -> 109 us; a decoupled look-back inside cubes_scatter with device-scope
//  flags -- 15 -> 31 us, the flag reads go past the XCDs' L2s.  Two launches it stays.)
// Ordered compaction of per-cube slots, two small launches: (1) sums of 1024 counts, (2) every workgroup adds up the sums
// before its own (a few hundred to a few thousand words from L2: cheaper than a third launch in between, which costs
// ~5 us of a frame's critical path), rescans its 1024 counts from that base and copies the slots; the last workgroup
// also leaves the totals.  Item i belongs to frame i / items_per_frame; frame_start[f] (optional) receives the offset of
// the frame's first extremum.  (cubes_scan_sums, the separate scan, still serves the orientation's batched form.)
__global__ __launch_bounds__(1024) void cubes_block_sums(const unsigned *__restrict__ counts, long n,
                                                         unsigned *__restrict__ sums) {
    __shared__ unsigned ws[16];
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

__global__ __launch_bounds__(1024) void cubes_scan_sums(unsigned *__restrict__ sums, long nb, unsigned long long max_out,
                                                        unsigned long long *__restrict__ totals) {
    __shared__ unsigned ws[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long base = 0;
    for (long chunk = 0; chunk < nb; chunk += 1024) {
        const long i = chunk + threadIdx.x;
        const unsigned c = i < nb ? sums[i] : 0u;
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = ws[v];
            before += v < wave ? t : 0u;
            all += t;
        }
        if (i < nb) sums[i] = (unsigned)(base + before + incl - c);   // exclusive offset of the block
        base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = base < max_out ? base : max_out;
        totals[1] = base < max_out ? 0ull : base - max_out;
    }
}

__global__ __launch_bounds__(1024) void cubes_scatter(const unsigned *__restrict__ counts,
                                                      const float *__restrict__ slots,
                                                      const unsigned *__restrict__ block_sums, long n,
                                                      long items_per_frame, float *__restrict__ out /*[max_out][4]*/,
                                                      unsigned *__restrict__ frame_of, unsigned *__restrict__ frame_start,
                                                      unsigned long long max_out, unsigned long long *__restrict__ totals) {
    __shared__ unsigned ws[16];
    __shared__ unsigned long long wbase[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // extrema in the workgroups before this one
    unsigned long long mine = 0;
    for (long b = threadIdx.x; b < (long)blockIdx.x; b += 1024) mine += block_sums[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) wbase[wave] = mine;
    __syncthreads();
    unsigned long long block_base = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) block_base += wbase[v];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const unsigned long long all = block_base + block_sums[blockIdx.x];
        totals[0] = all < max_out ? all : max_out;
        totals[1] = all < max_out ? 0ull : all - max_out;
    }
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
    const unsigned long long first = block_base + before + incl - c;
    if (i >= n) return;
    const unsigned f = (unsigned)(i / items_per_frame);
    if (frame_start && i == (long)f * items_per_frame) frame_start[f] = (unsigned)(first < max_out ? first : max_out);
    for (unsigned j = 0; j < c; ++j) {
        const unsigned long long o = first + j;
        if (o < max_out) {
            *reinterpret_cast<f32x4 *>(out + o * 4) = *reinterpret_cast<const f32x4 *>(slots + ((size_t)i * 8 + j) * 4);
            if (frame_of) frame_of[o] = f;
        }
    }
}

// TopKContrastFilter::filter (vulkan/mod.rs:1753-1786) for one frame's extrema [n][4], one workgroup per frame:
// keep blobs with size >= min_size; if more than n_keep remain, find the (n_keep+1)-th largest contrast (radix
// select on the float bits, contrast >= 0) and keep, in index order, the first n_keep blobs that reach it.
// seg_start[f], seg_start[f+1] delimit frame f (seg_start == nullptr: one segment [0, *n_in)).
// out: gathered extrema from out_base(f) = f * n_keep; out_count[f].
__global__ __launch_bounds__(1024) void topk_filter(const float *__restrict__ extrema, const unsigned *__restrict__ seg_start,
                                                    const unsigned long long *__restrict__ n_in,
                                                    unsigned long long n_host, unsigned n_frames, unsigned seg_cap,
                                                    unsigned n_keep, float min_size, float *__restrict__ out,
                                                    unsigned *__restrict__ out_index, unsigned *__restrict__ out_count,
                                                    unsigned long long *__restrict__ out_count64) {
    __shared__ unsigned hist[256];
    __shared__ unsigned ws[16];
    __shared__ unsigned sh_prefix, sh_rank, sh_m;
    const unsigned f = blockIdx.x;
    const unsigned total = (unsigned)(n_in ? n_in[0] : n_host);   // count on the device, or given by the host
    unsigned lo = seg_start ? seg_start[f] : 0u;
    unsigned hi = seg_start ? (f + 1 < n_frames ? seg_start[f + 1] : total) : total;
    lo = lo < total ? lo : total;
    hi = hi < total ? hi : total;
    hi = hi - lo > seg_cap ? lo + seg_cap : hi;   // a frame's extrema beyond max_extrema are dropped (mod.rs:627)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // One sweep over the segment, four extrema per thread and step so that their loads are in flight together (the
    // segment is a few thousand to a few ten thousand entries: the sweeps are latency-, not bandwidth-bound).
    // fn(index, passes min_size, key = float bits of |contrast|)
    auto sweep = [&](auto fn) {
        for (unsigned i0 = lo + threadIdx.x; i0 < hi; i0 += 4096) {
            float sz[4], ct[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned i = i0 + 1024u * j;
                const bool in = i < hi;
                sz[j] = in ? extrema[(size_t)i * 4 + 2] : 0.f;
                ct[j] = in ? extrema[(size_t)i * 4 + 3] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned i = i0 + 1024u * j;
                if (i < hi) fn(i, sz[j] >= min_size, __float_as_uint(fabsf(ct[j])));
            }
        }
    };
    // how many pass min_size
    if (threadIdx.x == 0) sh_m = 0;
    __syncthreads();
    unsigned mine = 0;
    sweep([&](unsigned, bool pass, unsigned) { mine += pass ? 1u : 0u; });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) atomicAdd(&sh_m, mine);
    __syncthreads();
    const unsigned m = sh_m;
    unsigned cutoff = 0;   // key threshold; 0 keeps everything that passes
    if (m > n_keep) {
        // radix select, most significant byte first: rank n_keep (0-based) in descending key order
        unsigned prefix = 0, rank = n_keep;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int b = threadIdx.x; b < 256; b += 1024) hist[b] = 0;
            __syncthreads();
            const unsigned mask = shift == 24 ? 0u : 0xFFFFFFFFu << (shift + 8);
            sweep([&](unsigned, bool pass, unsigned k) {
                if (pass && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            });
            __syncthreads();
            // the bin holding the wanted rank, walking down from byte value 255: threads 0..255 take bins 255..0,
            // prefix sums over them locate it in parallel
            {
                const unsigned c = threadIdx.x < 256 ? hist[255 - threadIdx.x] : 0u;   // waves 4..15 carry zeros
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
                const unsigned excl = before + incl - c;
                if (c != 0 && rank >= excl && rank < excl + c) {
                    sh_prefix = prefix | ((255u - threadIdx.x) << shift);
                    sh_rank = rank - excl;
                }
            }
            __syncthreads();
            prefix = sh_prefix;
            rank = sh_rank;
            __syncthreads();
        }
        cutoff = prefix;
    }
    // ordered compaction of {passes && key >= cutoff}, first n_keep
    unsigned base = 0;
    for (unsigned chunk = lo; chunk < hi && base < n_keep; chunk += 1024) {
        const unsigned i = chunk + threadIdx.x;
        const bool take = i < hi && extrema[(size_t)i * 4 + 2] >= min_size &&
                          __float_as_uint(fabsf(extrema[(size_t)i * 4 + 3])) >= cutoff;
        const unsigned long long bm = __ballot(take);
        if (lane == 0) ws[wave] = (unsigned)__popcll(bm);
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = ws[v];
            before += v < wave ? t : 0u;
            all += t;
        }
        const unsigned o = base + before + (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
        if (take && o < n_keep) {
            const size_t dst = (size_t)f * n_keep + o;
            *reinterpret_cast<f32x4 *>(out + dst * 4) = *reinterpret_cast<const f32x4 *>(extrema + (size_t)i * 4);
            if (out_index) out_index[dst] = i - lo;
        }
        base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_count[f] = base < n_keep ? base : n_keep;
        if (out_count64 && f == 0) out_count64[0] = base < n_keep ? base : n_keep;
    }
}

// The same selection for ONE long list (a 4K frame yields tens of thousands of extrema), spread over the chip, in five
// launches: one histogram launch per radix digit, then the ordered compaction in two.  The key (float bits of a
// non-negative contrast: bit 31 clear) is cut into digits of 11, 11 and 9 bits -- three histogram launches instead of one per
// byte (a dependent launch costs ~5 us of the frame's critical path; the top BYTE of such keys takes three values and
// selects next to nothing).  There is no launch that "picks the bin" between them: each digit has its own histogram, and
// every workgroup of a later launch replays the picks of the digits before (a scan of <= 2048 bins each).
// work (u32): [0, 3 x 2048) the three histograms (zeroed by the scatter launch when everybody is done with them, and by
// the host when the buffer is allocated), [kTopkState] cutoff key, [kTopkSums...) block sums of the compaction.
constexpr int kTopkDigits = 3, kTopkBins = 2048;
constexpr int kTopkState = kTopkDigits * kTopkBins, kTopkSums = kTopkState + 16;
__device__ __forceinline__ constexpr int topk_shift(int q) { return q == 0 ? 20 : (q == 1 ? 9 : 0); }
__device__ __forceinline__ constexpr unsigned topk_digit_mask(int q) { return q == 2 ? 511u : 2047u; }
// bits above digit q (they must equal the prefix found so far)
__device__ __forceinline__ constexpr unsigned topk_above(int q) { return q == 0 ? 0u : (q == 1 ? 0xFFF00000u : 0xFFFFFE00u); }

// the state after `passes` digits: prefix of the wanted key, its rank inside that prefix; done = everything that passes
// min_size is kept (first digit only).  All 1024 threads of the workgroup call it; thread t owns bins 2047 - 2t and
// 2046 - 2t (prefix sums walk down from the top bin).
__device__ void topk_replay(const unsigned *__restrict__ hist, int passes, unsigned n_keep, unsigned &prefix,
                            unsigned &rank, bool &done) {
    __shared__ unsigned ws[16], s_prefix, s_rank, s_done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    prefix = 0;
    rank = n_keep;          // rank wanted (0-based, descending)
    done = false;
    for (int q = 0; q < passes; ++q) {
        const int shift = topk_shift(q);
        const int hi_bin = kTopkBins - 1 - 2 * (int)threadIdx.x;
        const unsigned c_hi = hist[q * kTopkBins + hi_bin], c_lo = hist[q * kTopkBins + hi_bin - 1];
        const unsigned c = c_hi + c_lo;
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) ws[wave] = incl;
        if (threadIdx.x == 0) s_done = 0;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            before += v < wave ? ws[v] : 0u;
            all += ws[v];
        }
        const unsigned excl = before + incl - c;
        if (q == 0 && all <= n_keep) {               // the first histogram counts everything that passes min_size
            if (threadIdx.x == 0) s_done = 1;
        } else if (c_hi != 0 && rank >= excl && rank < excl + c_hi) {
            s_prefix = prefix | ((unsigned)hi_bin << shift);
            s_rank = rank - excl;
        } else if (c_lo != 0 && rank >= excl + c_hi && rank < excl + c) {
            s_prefix = prefix | ((unsigned)(hi_bin - 1) << shift);
            s_rank = rank - excl - c_hi;
        }
        __syncthreads();
        done = s_done != 0;
        const unsigned np = s_prefix, nr = s_rank;
        __syncthreads();
        if (done) return;
        prefix = np;
        rank = nr;
    }
}

__global__ __launch_bounds__(1024) void topk_hist(const float *__restrict__ extrema, const unsigned long long *__restrict__ n_in,
                                                  unsigned long long n_host, float min_size, int pass, unsigned n_keep,
                                                  unsigned *__restrict__ work) {
    __shared__ unsigned lh[4][kTopkBins];   // one histogram per four waves (32 KiB)
    unsigned prefix, rank;
    bool done;
    topk_replay(work, pass, n_keep, prefix, rank, done);
    if (done) return;
    const int shift = topk_shift(pass);
    const unsigned n = (unsigned)(n_in ? n_in[0] : n_host);
    const unsigned above = topk_above(pass), dmask = topk_digit_mask(pass);
    const int copy = threadIdx.x >> 8;
    for (int b = threadIdx.x; b < 4 * kTopkBins; b += 1024) (&lh[0][0])[b] = 0;
    __syncthreads();
    float sz[4], ct[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned i = blockIdx.x * 4096u + threadIdx.x + 1024u * j;
        sz[j] = i < n ? extrema[(size_t)i * 4 + 2] : 0.f;
        ct[j] = i < n ? extrema[(size_t)i * 4 + 3] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned i = blockIdx.x * 4096u + threadIdx.x + 1024u * j;
        const unsigned k = __float_as_uint(fabsf(ct[j]));
        if (i < n && sz[j] >= min_size && (k & above) == prefix) atomicAdd(&lh[copy][(k >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kTopkBins; b += 1024) {
        const unsigned t = lh[0][b] + lh[1][b] + lh[2][b] + lh[3][b];
        if (t) atomicAdd(&work[pass * kTopkBins + b], t);
    }
}

// take flags of the compaction: passes min_size and reaches the cutoff key
__device__ __forceinline__ bool topk_take(const float *__restrict__ extrema, unsigned i, unsigned n, float min_size,
                                          unsigned cutoff) {
    return i < n && extrema[(size_t)i * 4 + 2] >= min_size && __float_as_uint(fabsf(extrema[(size_t)i * 4 + 3])) >= cutoff;
}

__global__ __launch_bounds__(1024) void topk_sums(const float *__restrict__ extrema, const unsigned long long *__restrict__ n_in,
                                                  unsigned long long n_host, float min_size, unsigned n_keep,
                                                  unsigned *__restrict__ work) {
    __shared__ unsigned ws[16];
    unsigned prefix, rank;
    bool done;
    topk_replay(work, kTopkDigits, n_keep, prefix, rank, done);
    const unsigned cutoff = done ? 0u : prefix;      // key threshold; 0 keeps everything that passes min_size
    if (blockIdx.x == 0 && threadIdx.x == 0) work[kTopkState] = cutoff;
    const unsigned n = (unsigned)(n_in ? n_in[0] : n_host);
    const bool take = topk_take(extrema, blockIdx.x * 1024u + threadIdx.x, n, min_size, cutoff);
    const unsigned long long bm = __ballot(take);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = (unsigned)__popcll(bm);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int v = 0; v < 16; ++v) t += ws[v];
        work[kTopkSums + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void topk_scatter(const float *__restrict__ extrema, const unsigned long long *__restrict__ n_in,
                                                     unsigned long long n_host, float min_size, unsigned n_keep,
                                                     unsigned *__restrict__ work, float *__restrict__ out,
                                                     unsigned *__restrict__ out_index, unsigned *__restrict__ out_count,
                                                     unsigned long long *__restrict__ out_count64) {
    __shared__ unsigned ws[16], wb[16];
    const unsigned n = (unsigned)(n_in ? n_in[0] : n_host);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // kept extrema in the workgroups before this one (workgroup 0: in all of them, for the count)
    const unsigned upto = blockIdx.x == 0 ? gridDim.x : blockIdx.x;
    unsigned mine = 0;
    for (unsigned b = threadIdx.x; b < upto; b += 1024) mine += work[kTopkSums + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) wb[wave] = mine;
    const unsigned i = blockIdx.x * 1024u + threadIdx.x;
    const bool take = topk_take(extrema, i, n, min_size, work[kTopkState]);
    const unsigned long long bm = __ballot(take);
    if (lane == 0) ws[wave] = (unsigned)__popcll(bm);
    __syncthreads();
    unsigned before = 0, sum_b = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        before += v < wave ? ws[v] : 0u;
        sum_b += wb[v];
    }
    const unsigned block_base = blockIdx.x == 0 ? 0u : sum_b;
    const unsigned o = block_base + before + (unsigned)__popcll(bm & ((1ull << lane) - 1ull));
    if (take && o < n_keep) {
        *reinterpret_cast<f32x4 *>(out + (size_t)o * 4) = *reinterpret_cast<const f32x4 *>(extrema + (size_t)i * 4);
        if (out_index) out_index[o] = i;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            const unsigned kept = sum_b < n_keep ? sum_b : n_keep;
            out_count[0] = kept;
            if (out_count64) out_count64[0] = kept;
        }
    }
    // the histograms were last read by the launch before this one: ready for the next list
    if (blockIdx.x == 0)
        for (int b = threadIdx.x; b < kTopkDigits * kTopkBins; b += 1024) work[b] = 0;
}

// The same selection for one list of at most 32 768 extrema (a 1080p frame yields about ten thousand) in ONE launch of ONE
// workgroup: the five launches above are ~5 us of launch cadence each on a frame's critical path, and such a list is small
// enough to live in the registers of 1024 threads -- wave w owns the contiguous range [64 K w, 64 K (w + 1)) of the list,
// K = ceil(n / 1024) <= 32, lane l its items 64 K w + 64 j + l (coalesced loads, read once).  (Beyond that one CU is the
// wrong place: measured with 64 items per thread, a 4K frame's 41 k extrema took 36 us against the five launches' 29;
// tools/bench_topk.py: level at 32 k, 2-6 us ahead below 24 k.)  Key = float bits of the
// non-negative contrast with bit 31 set if the blob passes min_size, 0 otherwise.  Radix select with LDS histograms over
// 11-bit digits of the bits in which the keys differ, then the ordered compaction: ballots inside a wave, one scan over the 16 waves.
// Same result as topk_hist / topk_sums / topk_scatter (and as one workgroup of topk_filter), decision for decision.
constexpr int kTopkOneMaxK = 32;
template <int KMAX>
__global__ __launch_bounds__(1024) void topk_one(const float *__restrict__ extrema, const unsigned long long *__restrict__ n_in,
                                                 unsigned long long n_host, float min_size, unsigned n_keep,
                                                 float *__restrict__ out, unsigned *__restrict__ out_index,
                                                 unsigned *__restrict__ out_count, unsigned long long *__restrict__ out_count64) {
    __shared__ unsigned hist[kTopkBins];
    __shared__ unsigned ws[16], s_prefix, s_rank, s_m, s_min, s_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned n = (unsigned)(n_in ? n_in[0] : n_host);
    const unsigned K = (n + 1023u) / 1024u;                      // <= KMAX (the launcher's contract)
    const unsigned first = (unsigned)wave * 64u * K + (unsigned)lane;
    unsigned key[KMAX];
    // sixteen items per lane at a time: the loads of a group are all requested before the first is looked at (the values
    // pass through an opaque statement: hipcc otherwise makes the contrast's load conditional on the size's test and waits
    // for every item in turn -- 41 dependent round trips on a 4K frame's list)
#pragma unroll
    for (int j0 = 0; j0 < KMAX; j0 += 16) {
        if ((unsigned)j0 < K) {   // uniform
            unsigned long long raw[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned i = first + 64u * (j0 + j);
                const unsigned ii = i < n ? i : n - 1u;      // lanes past the end read the last item and drop it
                // (uniform base + 32-bit byte offset: i < 65 536, so no 64-bit address arithmetic per item)
                raw[j] = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(extrema) + (ii * 16u + 8u));
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(raw[j]));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned i = first + 64u * (j0 + j);
                const float size = __uint_as_float((unsigned)raw[j]), contrast = __uint_as_float((unsigned)(raw[j] >> 32));
                key[j0 + j] = ((unsigned)(j0 + j) < K && i < n && size >= min_size) ? (__float_as_uint(fabsf(contrast)) | 0x80000000u) : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) key[j0 + j] = 0u;
        }
    }
    // how many pass min_size, and the range of their keys
    unsigned mine = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        const unsigned k = key[j];
        mine += k >> 31;
        kmax = k > kmax ? k : kmax;
        kmin = (k >> 31) && k < kmin ? k : kmin;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mine += __shfl_xor(mine, o);
        const unsigned a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if (threadIdx.x == 0) { s_m = 0; s_min = 0xFFFFFFFFu; s_max = 0u; }
    __syncthreads();
    if (lane == 0) { atomicAdd(&s_m, mine); atomicMin(&s_min, kmin); atomicMax(&s_max, kmax); }
    __syncthreads();
    unsigned cutoff = 0x80000000u;    // keep every blob that passes min_size
    if (s_m > n_keep) {
        // Radix select of the key of rank n_keep (0-based, descending) over the bits in which the keys DIFFER: the bits
        // above the highest differing one are common to all (contrasts of one frame span a few binades: digits taken from
        // bit 30 down would put every key into a handful of bins, and same-address LDS atomics serialise), so the first
        // 11-bit digit starts there and the keys spread over its 2048 bins.
        const unsigned diff = (s_min ^ s_max) & 0x7FFFFFFFu;
        unsigned prefix = s_max, rank = n_keep;
        if (diff != 0u) {
            int hi = 32 - __builtin_clz(diff);            // undecided bits: [hi - 1 : 0]
            prefix = s_max & ~((1u << hi) - 1u);
#pragma unroll 1
            while (hi > 0) {
                const int bits = hi < 11 ? hi : 11, shift = hi - bits;
                const unsigned above = ~((1u << hi) - 1u), dmask = (1u << bits) - 1u;
                for (int b = threadIdx.x; b < kTopkBins; b += 1024) hist[b] = 0;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < KMAX; ++j) {
                    const unsigned k = key[j];
                    if ((k >> 31) && (k & above) == prefix) atomicAdd(&hist[(k >> shift) & dmask], 1u);
                }
                __syncthreads();
                // thread t owns bins 2047 - 2t and 2046 - 2t: prefix sums walk down from the top bin (as topk_replay)
                const int hi_bin = kTopkBins - 1 - 2 * (int)threadIdx.x;
                const unsigned c_hi = hist[hi_bin], c_lo = hist[hi_bin - 1];
                const unsigned c = c_hi + c_lo;
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
                const unsigned excl = before + incl - c;
                if (c_hi != 0 && rank >= excl && rank < excl + c_hi) {
                    s_prefix = prefix | ((unsigned)hi_bin << shift);
                    s_rank = rank - excl;
                } else if (c_lo != 0 && rank >= excl + c_hi && rank < excl + c) {
                    s_prefix = prefix | ((unsigned)(hi_bin - 1) << shift);
                    s_rank = rank - excl - c_hi;
                }
                __syncthreads();
                prefix = s_prefix;
                rank = s_rank;
                __syncthreads();
                hi = shift;
            }
        }
        cutoff = prefix;
    }
    // ordered compaction of {key >= cutoff}: first the waves' totals, then every wave walks its range again
    unsigned total = 0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) total += (unsigned)__popcll(__ballot(key[j] >= cutoff));
    if (lane == 0) ws[wave] = total;
    __syncthreads();
    unsigned at = 0, all = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        at += v < wave ? ws[v] : 0u;
        all += ws[v];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned first2 = first;
    asm volatile("" : "+v"(first2));   // (the item addresses are formed again here, not kept alive from the loads above)
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        const bool take = key[j] >= cutoff;
        const unsigned long long bm = __ballot(take);
        const unsigned o = at + (unsigned)__popcll(bm & below);
        if (take && o < n_keep) {
            const unsigned i = first2 + 64u * j;
            *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(out) + o * 16u) =
                *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(extrema) + i * 16u);
            if (out_index) out_index[o] = i;
        }
        at += (unsigned)__popcll(bm);
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // (keeps the copies of later items from being hoisted: registers)
    }
    if (threadIdx.x == 0) {
        const unsigned kept = all < n_keep ? all : n_keep;
        out_count[0] = kept;
        if (out_count64) out_count64[0] = kept;
    }
}

void scan_grid(int w, int h, int n_fine, int border, int skip_layers, int &gx, int &gy, int &gz) {
    gx = w - 2 * border > 0 ? (w - 2 * border + 3) / 4 : 0;
    gy = h - 2 * border > 0 ? (h - 2 * border + 3) / 4 : 0;
    gz = n_fine - 2 - skip_layers > 0 ? (n_fine - 2 - skip_layers + 3) / 4 : 0;
}

// a-trous stack -> ordered extrema of `frames` frames.  slots/counts/sums are scratch sized for frames x cubes.
void launch_detect_extrema(const float *layer0, long layer0_stride, int layer0_pitch, const float *coarse, long coarse_stride,
                           long layer_stride, int n_layers, int w, int h, int frames, int border, int skip_layers,
                           float contrast_threshold, float *slots, unsigned *counts, unsigned *sums, float *extrema,
                           unsigned *frame_of, unsigned *frame_start, unsigned long long max_out,
                           unsigned long long *totals, hipStream_t stream, const RowBands *bands) {
    int gx, gy, gz;
    scan_grid(w, h, n_layers - 1, border, skip_layers, gx, gy, gz);
    const long ncubes = (long)gx * gy * gz, n = ncubes * frames;
    const long nb = (n + 1023) / 1024;
    if (ncubes > 0) {
        // layers readable 16 bytes at a time: rows start on multiples of four texels in every layer of every frame
        const int aligned = w % 4 == 0 && layer0_pitch % 4 == 0 && layer0_stride % 4 == 0 && coarse_stride % 4 == 0 &&
                            layer_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(layer0) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(coarse) & 15) == 0;
        // (bands: a piece scans the tile rows [lo, hi) its rows allow and stops; the last piece scans the rest and compacts)
        const bool head_only = bands && !bands->last;
        const int all_rows = (gy + kScanTY / 4 - 1) / (kScanTY / 4);
        const int by_first = bands ? std::min(bands->scan_lo, all_rows) : 0;
        const int by_rows = (head_only ? std::min(bands->scan_hi, all_rows) : all_rows) - by_first;
        auto scan = [&](auto kernel, int tx) {
            if (by_rows > 0)
                hipLaunchKernelGGL(kernel, dim3((gx + tx / 4 - 1) / (tx / 4), by_rows, frames), dim3(256), 0, stream, layer0,
                                   layer0_stride, layer0_pitch, coarse, coarse_stride, layer_stride, n_layers - 1, w, h, border,
                                   skip_layers, contrast_threshold, gx, gy, gz, aligned, slots, counts, by_first);
        };
        if ((long)frames * w * h >= 2000000L) scan(scan_extrema<64>, 64);
        else scan(scan_extrema<32>, 32);
        if (head_only) return;
        hipLaunchKernelGGL(cubes_block_sums, dim3((unsigned)nb), dim3(1024), 0, stream, (const unsigned *)counts, n, sums);
    } else if (bands && !bands->last) {
        return;
    }
    if (ncubes > 0)
        hipLaunchKernelGGL(cubes_scatter, dim3((unsigned)nb), dim3(1024), 0, stream, (const unsigned *)counts,
                           (const float *)slots, (const unsigned *)sums, n, ncubes, extrema, frame_of, frame_start,
                           max_out, totals);
    else
        (void)hipMemsetAsync(totals, 0, 2 * sizeof(unsigned long long), stream);
}

void launch_topk_filter(const float *extrema, const unsigned *seg_start, const unsigned long long *n_in,
                        unsigned long long n_host, unsigned n_frames, unsigned seg_cap, unsigned n_keep, float min_size,
                        float *out, unsigned *out_index, unsigned *out_count, unsigned long long *out_count64,
                        unsigned long long n_cap, unsigned *work, hipStream_t stream) {
    // one frame with a long list and scratch to work in: the multi-workgroup form; otherwise one workgroup per frame
    const char *form = getenv("LF_MKD_TOPK");          // "multi": the five-launch form at every size (A/B runs, tests)
    const bool multi = form && form[0] == 'm';
    if (n_frames == 1 && !seg_start && n_cap > 8192 && n_cap <= 1024ull * kTopkOneMaxK && seg_cap >= n_cap && !multi) {
        // one list that fits the registers of one workgroup (K = ceil(n / 1024) <= 32 items per thread): one launch
        hipLaunchKernelGGL(topk_one<kTopkOneMaxK>, dim3(1), dim3(1024), 0, stream, extrema, n_in, n_host, min_size, n_keep, out,
                           out_index, out_count, out_count64);
        return;
    }
    if (n_frames == 1 && !seg_start && work && n_cap > 8192 && seg_cap >= n_cap) {
        const unsigned nb4 = (unsigned)((n_cap + 4095) / 4096), nb1 = (unsigned)((n_cap + 1023) / 1024);
        for (int pass = 0; pass < kTopkDigits; ++pass)
            hipLaunchKernelGGL(topk_hist, dim3(nb4), dim3(1024), 0, stream, extrema, n_in, n_host, min_size, pass, n_keep,
                               work);
        hipLaunchKernelGGL(topk_sums, dim3(nb1), dim3(1024), 0, stream, extrema, n_in, n_host, min_size, n_keep, work);
        hipLaunchKernelGGL(topk_scatter, dim3(nb1), dim3(1024), 0, stream, extrema, n_in, n_host, min_size, n_keep, work,
                           out, out_index, out_count, out_count64);
        return;
    }
    hipLaunchKernelGGL(topk_filter, dim3(n_frames), dim3(1024), 0, stream, extrema, seg_start, n_in, n_host, n_frames,
                       seg_cap, n_keep, min_size, out, out_index, out_count, out_count64);
}

size_t topk_work_words(unsigned long long n_cap) { return kTopkSums + (size_t)((n_cap + 1023) / 1024) + 2; }

// [frames][n_keep] padded per-frame selections + counts -> one contiguous list with the frame of every entry.
// totals[0] = entries, totals[1] = extrema the per-frame cap dropped (dropped_blobs summed over the frames).
__global__ __launch_bounds__(1024) void segments_offsets(const unsigned *__restrict__ counts, unsigned n_frames,
                                                         const unsigned *__restrict__ seg_start,
                                                         const unsigned long long *__restrict__ n_total, unsigned seg_cap,
                                                         unsigned *__restrict__ offsets,
                                                         unsigned long long *__restrict__ totals) {
    __shared__ unsigned ws[16];
    __shared__ unsigned long long dropped;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) dropped = 0;
    __syncthreads();
    unsigned long long base = 0;
    for (unsigned chunk = 0; chunk < n_frames; chunk += 1024) {
        const unsigned f = chunk + threadIdx.x;
        const unsigned c = f < n_frames ? counts[f] : 0u;
        if (f < n_frames) {
            const unsigned total = (unsigned)n_total[0];
            const unsigned lo = seg_start[f], hi = f + 1 < n_frames ? seg_start[f + 1] : total;
            if (hi - lo > seg_cap) atomicAdd(&dropped, (unsigned long long)(hi - lo - seg_cap));
        }
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = ws[v];
            before += v < wave ? t : 0u;
            all += t;
        }
        if (f < n_frames) offsets[f] = (unsigned)(base + before + incl - c);
        base += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = base;
        totals[1] = dropped;
    }
}

__global__ __launch_bounds__(256) void segments_gather(const float *__restrict__ padded, const unsigned *__restrict__ counts,
                                                       const unsigned *__restrict__ offsets, unsigned n_keep,
                                                       float *__restrict__ out, unsigned *__restrict__ frame_of) {
    const unsigned f = blockIdx.x, c = counts[f], o = offsets[f];
    for (unsigned i = threadIdx.x; i < c; i += 256) {
        *reinterpret_cast<f32x4 *>(out + (size_t)(o + i) * 4) =
            *reinterpret_cast<const f32x4 *>(padded + ((size_t)f * n_keep + i) * 4);
        frame_of[o + i] = f;
    }
}

void launch_segments_compact(const float *padded, const unsigned *counts, const unsigned *seg_start,
                             const unsigned long long *n_total, unsigned n_frames, unsigned seg_cap, unsigned n_keep,
                             unsigned *offsets, float *out, unsigned *frame_of, unsigned long long *totals,
                             hipStream_t stream) {
    hipLaunchKernelGGL(segments_offsets, dim3(1), dim3(1024), 0, stream, counts, n_frames, seg_start, n_total, seg_cap,
                       offsets, totals);
    hipLaunchKernelGGL(segments_gather, dim3(n_frames), dim3(256), 0, stream, padded, counts, (const unsigned *)offsets,
                       n_keep, out, frame_of);
}

}  // namespace lfmkd
