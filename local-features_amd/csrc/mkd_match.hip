// gfx950 brute-force descriptor matcher.
// Reference semantics: examples/match_images/src/main.rs:8-27 -- for every row of a: similarity = dot product
// with every row of b, best and second best, accept if best * ratio > second.
//
// The similarity matrix is a GEMM (K = 128) and lives on the matrix cores: S^T tile = b tile (32 rows) x a tile
// (32 columns) by v_mfma_f32_32x32x16_f16.  With b on the M side, a lane of the accumulator tile holds ONE a
// column and 16 b rows, so the running best / second best of an a row is an in-register reduction; lanes never
// exchange anything until the very end.  f32 accuracy comes from the same three-term f16 split as the pooling
// kernel (hi*hi + lo*hi + hi*lo, f32 accumulate, ~2^-21 relative): both sides are split once by `match_split`
// into MFMA operand order, so the hot loop only moves fragments.
//
//   workgroup = 8 waves x 64 a rows = 512 a rows, resident in registers (2 x 64 VGPRs of fragments per wave);
//   b streams through LDS in 32-row tiles (16 KiB, LDS-DMA, double buffered), each tile read once per workgroup;
//   grid = (a blocks, b splits): a split scans one contiguous range of b tiles and writes partial
//   (best, index, second) per a row; `match_merge` folds the partials and applies the ratio test.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mkd_device.h"

namespace lfmkd {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileRows = 32;                 // rows of a / b per MFMA tile
constexpr int kTileBytes = 2 * 8 * 2 * 32 * 16;  // [hi|lo][k-step 8][k-half 2][row 32][8 f16] = 16 KiB
constexpr int kWaves = 8, kATiles = 2;        // per wave: 2 a tiles

__device__ __forceinline__ void lds_dma16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds(g, reinterpret_cast<__attribute__((address_space(3))) void *>(
                                            reinterpret_cast<uintptr_t>(l)), 16, 0, 0);
}

}  // namespace

// x [n][128] f32 -> tiles of 32 rows in MFMA operand order: [tile][part: hi, lo][s 8][h 2][r 32][8 f16], where
// element j of (s, h, r) is x[32 tile + r][16 s + 8 h + j]; rows beyond n are zero.
__global__ __launch_bounds__(256) void match_split(const float *__restrict__ x, long n, unsigned char *__restrict__ out) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);   // 16 threads per row, 8 floats each
    const int c8 = threadIdx.x & 15;                               // k = 8 c8 .. 8 c8 + 7  ->  s = c8 >> 1, h = c8 & 1
    const long tiles = (n + kTileRows - 1) / kTileRows;
    if (row >= tiles * kTileRows) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = row < n ? x[row * 128 + c8 * 8 + j] : 0.f;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (_Float16)v[j];
        lo[j] = (_Float16)(v[j] - (float)hi[j]);
    }
    const long tile = row / kTileRows;
    const int r = (int)(row - tile * kTileRows), s = c8 >> 1, h = c8 & 1;
    unsigned char *base = out + tile * kTileBytes + ((s * 2 + h) * 32 + r) * 16;
    *reinterpret_cast<h8 *>(base) = hi;
    *reinterpret_cast<h8 *>(base + kTileBytes / 2) = lo;
}

// partial results: [split][a row]: best value, best index, second value
__global__ __launch_bounds__(512) void match_scan(const unsigned char *__restrict__ a_tiles, long na,
                                                  const unsigned char *__restrict__ b_tiles, long nb,
                                                  long tiles_per_split, const unsigned *__restrict__ excl_lo,
                                                  const unsigned *__restrict__ excl_hi, float *__restrict__ p_best,
                                                  int *__restrict__ p_index, float *__restrict__ p_second) {
    __shared__ __attribute__((aligned(16))) unsigned char s_b[2][kTileBytes];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const long a_tile0 = ((long)blockIdx.x * kWaves + wave) * kATiles;
    const long a_tiles_total = (na + kTileRows - 1) / kTileRows;
    const long b_tiles_total = (nb + kTileRows - 1) / kTileRows;
    const long t_begin = (long)blockIdx.y * tiles_per_split;
    long t_end = t_begin + tiles_per_split;
    t_end = t_end < b_tiles_total ? t_end : b_tiles_total;

    // a fragments: B operand of the MFMA, lane (r, h) holds a[col r][16 s + 8 h + j]
    h8 ah[kATiles][8], al[kATiles][8];
    unsigned lo_x[kATiles], hi_x[kATiles];
#pragma unroll
    for (int t = 0; t < kATiles; ++t) {
        const long at = a_tile0 + t < a_tiles_total ? a_tile0 + t : a_tiles_total - 1;   // idle tiles redo the last one
        const unsigned char *src = a_tiles + at * kTileBytes + (h * 32 + r) * 16;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            ah[t][s] = *reinterpret_cast<const h8 *>(src + s * 1024);
            al[t][s] = *reinterpret_cast<const h8 *>(src + s * 1024 + kTileBytes / 2);
        }
        const long arow = at * kTileRows + r;
        lo_x[t] = excl_lo && arow < na ? excl_lo[arow] : 0u;
        hi_x[t] = excl_lo && arow < na ? excl_hi[arow] : 0u;
    }
    float best[kATiles], second[kATiles];
    int best_i[kATiles];
#pragma unroll
    for (int t = 0; t < kATiles; ++t) { best[t] = -INFINITY; second[t] = -INFINITY; best_i[t] = -1; }

    // tile `t` of b -> LDS buffer `buf`: 16 KiB = 512 threads x 2 x 16 B
    auto issue = [&](long t, int buf) {
        const unsigned char *g = b_tiles + t * kTileBytes + threadIdx.x * 16;
        lds_dma16(g, &s_b[buf][0] + wave * 1024);
        lds_dma16(g + 8192, &s_b[buf][0] + 8192 + wave * 1024);
    };
    if (t_begin < t_end) issue(t_begin, 0);
    for (long t = t_begin; t < t_end; ++t) {
        const int buf = (int)((t - t_begin) & 1);
#ifndef LF_MATCH_ABLATE_SYNC
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's pieces of tile t have landed
        __syncthreads();                      // ... and everybody's; everybody is also done with the other buffer
        if (t + 1 < t_end) issue(t + 1, buf ^ 1);
#endif
        const unsigned char *bb = &s_b[buf][0] + (h * 32 + r) * 16;
        f32x16 acc[kATiles];
#pragma unroll
        for (int q = 0; q < kATiles; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {   // (requesting the fragments one k-step ahead was measured: no gain, the
                                        // partner wave on the SIMD already covers the LDS latency)
            const h8 bh = *reinterpret_cast<const h8 *>(bb + s * 1024);
            const h8 bl = *reinterpret_cast<const h8 *>(bb + s * 1024 + kTileBytes / 2);
#pragma unroll
            for (int q = 0; q < kATiles; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah[q][s], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al[q][s], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[q][s], acc[q], 0, 0, 0);
            }
        }
#ifdef LF_MATCH_ABLATE_EPILOGUE
#pragma unroll
        for (int q = 0; q < kATiles; ++q) best[q] += acc[q][0] + acc[q][7] + acc[q][15];
        continue;
#endif
        const int row0 = (int)(t * kTileRows) + 4 * h;
        const bool tail = (t + 1) * kTileRows > nb;
#pragma unroll
        for (int q = 0; q < kATiles; ++q) {
            // rows masked for this a: beyond nb, or inside the a row's own excluded range
            const bool touch = tail || ((unsigned)(t * kTileRows) < hi_x[q] && (unsigned)((t + 1) * kTileRows) > lo_x[q]);
            if (__builtin_amdgcn_ballot_w64(touch)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned row = (unsigned)(row0 + (i & 3) + 8 * (i >> 2));
                    if (row >= (unsigned)nb || (row >= lo_x[q] && row < hi_x[q])) acc[q][i] = -INFINITY;
                }
            }
            float m = acc[q][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[q][i]);
            if (__builtin_amdgcn_ballot_w64(m > second[q] || m >= best[q])) {   // rare once the scan is under way
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = acc[q][i];
                    const int row = row0 + (i & 3) + 8 * (i >> 2);
                    const bool nb_ = v >= best[q] && v > -INFINITY;   // later index wins among equals (stable sort, last)
                    const bool ns = !nb_ && v > second[q];
                    second[q] = nb_ ? best[q] : (ns ? v : second[q]);
                    best_i[q] = nb_ ? row : best_i[q];
                    best[q] = nb_ ? v : best[q];
                }
            }
        }
    }
    // fold the two k-halves' row sets (lanes l and l ^ 32 hold the same a column)
#pragma unroll
    for (int q = 0; q < kATiles; ++q) {
        const float ob = __shfl_xor(best[q], 32), os = __shfl_xor(second[q], 32);
        const int oi = __shfl_xor(best_i[q], 32);
        const bool other = ob > best[q] || (ob == best[q] && oi > best_i[q]);
        const float nbest = other ? ob : best[q];
        const float nsecond = other ? fmaxf(best[q], os) : fmaxf(second[q], ob);
        const int nidx = other ? oi : best_i[q];
        const long arow = (a_tile0 + q) * kTileRows + r;
        if (h == 0 && a_tile0 + q < a_tiles_total && arow < na) {
            const long o = (long)blockIdx.y * na + arow;
            p_best[o] = nbest;
            p_index[o] = nidx;
            p_second[o] = nsecond;
        }
    }
}

// folds the splits of one a row (ascending b ranges) and applies Lowe's ratio test (main.rs:22)
__global__ __launch_bounds__(256) void match_merge(const float *__restrict__ p_best, const int *__restrict__ p_index,
                                                   const float *__restrict__ p_second, long na, int splits, float ratio,
                                                   int *__restrict__ match, float *__restrict__ best_out,
                                                   float *__restrict__ second_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= na) return;
    float b = -INFINITY, s = -INFINITY;
    int bi = -1;
    for (int k = 0; k < splits; ++k) {
        const float ob = p_best[(long)k * na + i], os = p_second[(long)k * na + i];
        const int oi = p_index[(long)k * na + i];
        const bool other = oi >= 0 && ob >= b;   // later range = higher indices: wins among equals
        s = other ? fmaxf(b, os) : fmaxf(s, ob);
        bi = other ? oi : bi;
        b = other ? ob : b;
    }
    match[i] = (bi >= 0 && (ratio <= 0.f || b * ratio > s)) ? bi : -1;   // ratio <= 0: no test, the best index as is
    if (best_out) best_out[i] = b;
    if (second_out) second_out[i] = s;
}

size_t match_tiles_bytes(long n) { return (size_t)((n + kTileRows - 1) / kTileRows) * kTileBytes; }

int match_splits(long na, long nb, int num_cus) {
    const long a_blocks = (na + kWaves * kATiles * kTileRows - 1) / (kWaves * kATiles * kTileRows);
    const long b_tiles = (nb + kTileRows - 1) / kTileRows;
    long want = (2L * num_cus + a_blocks - 1) / a_blocks;   // about two workgroups per CU in flight
    want = want < 1 ? 1 : want;
    want = want > b_tiles ? b_tiles : want;
    want = want > 1024 ? 1024 : want;
    return (int)(want < 1 ? 1 : want);
}

void launch_match_split(const float *x, long n, unsigned char *tiles, hipStream_t stream) {
    if (n <= 0) return;
    const long rows = (n + kTileRows - 1) / kTileRows * kTileRows;
    hipLaunchKernelGGL(match_split, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, x, n, tiles);
}

void launch_match(const unsigned char *a_tiles, long na, const unsigned char *b_tiles, long nb, const unsigned *excl_lo,
                  const unsigned *excl_hi, float ratio, int splits, float *p_best, int *p_index, float *p_second,
                  int *match, float *best, float *second, hipStream_t stream) {
    if (na <= 0) return;
    const long a_blocks = (na + kWaves * kATiles * kTileRows - 1) / (kWaves * kATiles * kTileRows);
    const long n_b_tiles = (nb + kTileRows - 1) / kTileRows;
    const long per = (n_b_tiles + splits - 1) / splits;
    hipLaunchKernelGGL(match_scan, dim3((unsigned)a_blocks, (unsigned)splits), dim3(512), 0, stream, a_tiles, na, b_tiles,
                       nb, per, excl_lo, excl_hi, p_best, p_index, p_second);
    hipLaunchKernelGGL(match_merge, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, stream, (const float *)p_best,
                       (const int *)p_index, (const float *)p_second, na, splits, ratio, match, best, second);
}

}  // namespace lfmkd
