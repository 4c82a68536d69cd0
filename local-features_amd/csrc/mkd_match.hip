// gfx950 brute-force descriptor matcher.
// Reference semantics: examples/match_images/src/main.rs:8-27 -- for every row of a: similarity = dot product
// with every row of b, best and second best, accept if best * ratio > second.
//
// The similarity matrix is a GEMM (K = 128) and lives on the matrix cores, b on the M side: a lane of an accumulator
// tile then holds ONE a column and a few b rows, so the running best / second best of an a row is an in-register
// reduction.  Both sides are split once by `match_split` into f16 hi / lo MFMA operand tiles, so the hot loops only move
// fragments.  Two forms (lf_mkd.h; NOTEBOOK.md 4d), chosen by problem size in lf_mkd_match_device:
//
//   SCAN (`match_scan` + `match_merge`): every pair from three terms, hi*hi + lo*hi + hi*lo (f32 accumulate, ~2^-21
//   relative), v_mfma_f32_32x32x16_f16.  Workgroup = 8 waves x 64 a rows resident in registers (2 x 64 VGPRs of fragments
//   per wave); b streams through LDS in 32-row tiles (16 KiB, LDS-DMA, double buffered), each tile read once per
//   workgroup; grid = (a blocks, b splits): a split scans one contiguous range of b tiles and writes partial (best,
//   index, second) per a row; `match_merge` folds the partials and applies the ratio test.
//
//   SCREEN + VERIFY (`match_screen`, `match_verify`): every pair from the hi*hi term alone (a third of the matrix work,
//   v_mfma_f32_16x16x32_f16), which is within a margin derived from the rows' norms of the true similarity; every
//   candidate that could still be among an a row's two best is recorded, the few survivors are re-scored in f32 and the
//   decision taken on those values -- the result of an exhaustive f32 scan (see `screen_margin`).  An a row whose record
//   ring lost something verify needed is redone by the scan, alone (`match_split_rows`) or with everybody, gated on the
//   device (`MatchGate`: no host round trip).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mkd_device.h"

namespace lfmkd {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileRows = 32;                 // rows of a / b per MFMA tile
constexpr int kTileBytes = 2 * 8 * 2 * 32 * 16;  // [hi|lo][k-step 8][k-half 2][row 32][8 f16] = 16 KiB
constexpr int kWaves = 8, kATiles = 2;        // per wave: 2 a tiles

}  // namespace

// When the three-term scan runs as the fallback of the two-pass form it is enqueued unconditionally and decides on the
// device whether it has anything to do: `count` = a rows whose screening records overflowed.
//   rows == nullptr ("all"):   active when *count > above; scans every a row
//   rows != nullptr ("few"):   active when 0 < *count <= upto; the a tiles hold only the overflowed rows, rows[j] = the
//                              a row of compact row j, and *count replaces na
struct MatchGate {
    const int *count;
    const int *rows;
    int above, upto;
};

namespace {

__device__ __forceinline__ bool gate_open(const MatchGate &g, long &na) {
    if (!g.count) return true;
    const int n = *g.count;
    if (g.rows) {
        if (n <= 0 || n > g.upto) return false;
        na = n;
        return true;
    }
    return n > g.above;
}

__device__ __forceinline__ void lds_dma16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds(g, reinterpret_cast<__attribute__((address_space(3))) void *>(
                                            reinterpret_cast<uintptr_t>(l)), 16, 0, 0);
}

}  // namespace

// x [n][128] f32 -> tiles of 32 rows in MFMA operand order: [tile][part: hi, lo][s 8][h 2][r 32][8 f16], where
// element j of (s, h, r) is x[32 tile + r][16 s + 8 h + j]; rows beyond n are zero.
// norms (may be null) receives |x_row| rounded up; max_norm_bits (may be null) the largest of them, as float bits.
__global__ __launch_bounds__(256) void match_split(const float *__restrict__ x, long n, unsigned char *__restrict__ out,
                                                   float *__restrict__ norms, unsigned *__restrict__ max_norm_bits) {
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
    if (norms || max_norm_bits) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(v[j], v[j], ss);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) ss += __shfl_xor(ss, m, 16);
        if (c8 == 0 && row < n) {
            const float nr = sqrtf(ss) * 1.00001f;
            if (norms) norms[row] = nr;
            // one address for everybody: ask first, the maximum settles after a few rows
            if (max_norm_bits && __float_as_uint(nr) > __atomic_load_n(max_norm_bits, __ATOMIC_RELAXED))
                atomicMax(max_norm_bits, __float_as_uint(nr));
        }
    }
}

// The overflowed rows of a (rows[0 .. *count)), gathered and split into tiles of their own, with their exclusion ranges.
__global__ __launch_bounds__(256) void match_split_rows(const float *__restrict__ x, const int *__restrict__ rows,
                                                        const int *__restrict__ count, int upto,
                                                        const unsigned *__restrict__ excl_lo,
                                                        const unsigned *__restrict__ excl_hi,
                                                        unsigned char *__restrict__ out, unsigned *__restrict__ lo_out,
                                                        unsigned *__restrict__ hi_out) {
    const int n = *count;
    if (n <= 0 || n > upto) return;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c8 = threadIdx.x & 15;
    const long tiles = ((long)n + kTileRows - 1) / kTileRows;
    if (row >= tiles * kTileRows) return;
    const long src = row < n ? rows[row] : -1;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = src >= 0 ? x[src * 128 + c8 * 8 + j] : 0.f;
        hi[j] = (_Float16)v;
        lo[j] = (_Float16)(v - (float)hi[j]);
    }
    const long tile = row / kTileRows;
    const int r = (int)(row - tile * kTileRows), s = c8 >> 1, h = c8 & 1;
    unsigned char *base = out + tile * kTileBytes + ((s * 2 + h) * 32 + r) * 16;
    *reinterpret_cast<h8 *>(base) = hi;
    *reinterpret_cast<h8 *>(base + kTileBytes / 2) = lo;
    if (c8 == 0 && src >= 0 && excl_lo) { lo_out[row] = excl_lo[src]; hi_out[row] = excl_hi[src]; }
}

// partial results: [split][a row]: best value, best index, second value
__global__ __launch_bounds__(512) void match_scan(const unsigned char *__restrict__ a_tiles, long na,
                                                  const unsigned char *__restrict__ b_tiles, long nb,
                                                  long tiles_per_split, const unsigned *__restrict__ excl_lo,
                                                  const unsigned *__restrict__ excl_hi, float *__restrict__ p_best,
                                                  int *__restrict__ p_index, float *__restrict__ p_second,
                                                  MatchGate gate) {
    __shared__ __attribute__((aligned(16))) unsigned char s_b[2][kTileBytes];
    if (!gate_open(gate, na)) return;
    if ((long)blockIdx.x * kWaves * kATiles * kTileRows >= na) return;   // (the grid of the "few" form is sized for its capacity)
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
                                                   float *__restrict__ second_out, MatchGate gate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (!gate_open(gate, na) || i >= na) return;
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
    const long o = gate.rows ? gate.rows[i] : i;
    match[o] = (bi >= 0 && (ratio <= 0.f || b * ratio > s)) ? bi : -1;   // ratio <= 0: no test, the best index as is
    if (best_out) best_out[o] = b;
    if (second_out) second_out[o] = s;
}


// ---- small problems in ONE launch ------------------------------------------------------------------------------------
// The reference's own use is 2000 x 2000 (examples/match_images/src/main.rs:62-76): there the split / scan / merge chain is
// four dependent launches of a few microseconds each, i.e. launch cadence, not arithmetic.  `match_small` does the whole
// match in one launch with no scratch: a workgroup owns 16 a rows (the B operand of v_mfma_f32_16x16x32_f16: column = a row)
// and reads ALL of b straight from the caller's f32 rows, its 16 waves taking the 16-row b tiles round-robin (the next
// tile's rows are requested before the current one is processed); both sides are split into f16 hi / lo in registers
// (hi = truncation, lo = f16 of the residual, as the describe kernel splits its streams) and every pair gets the scan's
// three terms in the scan's order (lo*hi, hi*lo, hi*hi, f32 accumulate); the waves' (best, index, second) meet in LDS.
// Ties as everywhere: the higher b index wins.
// Cost: every workgroup converts all of b (64 vector instructions per tile and lane beside 12 matrix instructions), which
// is why this form is for small problems only (match_small_fits).  (Cutting b into ranges per a block, the ranges' results
// folded by whichever workgroup of the block finishes last, was built twice in round 4 and measured: with the partial
// results in memory behind release / acquire fences 4000 x 4000 took 300 us instead of 62 -- the fences write back and
// invalidate the XCDs' L2s; with the results meeting in two atomically maintained 64-bit words per a row (largest key,
// largest key that ever lost; device-scope atomics only, no fence) 2000 x 2000 took 50 us instead of 35 -- 30 k
// device-scope atomics, five workgroups per word.  One range it stays.)
constexpr int kSmallWaves = 16;

// (the body of a workgroup: `block` = which 16 rows of a it owns)
__device__ __forceinline__ void match_small_block(const float *__restrict__ a, long na, const float *__restrict__ b, long nb,
                                                  const unsigned *__restrict__ excl_lo, const unsigned *__restrict__ excl_hi,
                                                  float ratio, int *__restrict__ match, float *__restrict__ best_out,
                                                  float *__restrict__ second_out, long block) {
    __shared__ float s_best[kSmallWaves][16], s_second[kSmallWaves][16];
    __shared__ int s_idx[kSmallWaves][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const long arow = block * 16 + n;
    const long arow_c = arow < na ? arow : na - 1;
    struct Raw { f32x4 v[8]; };   // a row's share of the four k-steps: k = 32 s + 8 g .. + 7
    auto load_row = [&](const float *row) {
        Raw r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            r.v[2 * s] = *reinterpret_cast<const f32x4 *>(row + 32 * s + 8 * g);
            r.v[2 * s + 1] = *reinterpret_cast<const f32x4 *>(row + 32 * s + 8 * g + 4);
        }
        return r;
    };
    float one = 1.f;
    asm("" : "+v"(one));   // (keeps the residual a single v_fma_mix_f32: see AFrag in mkd_describe.hip)
    auto split8 = [&](const f32x4 &v0, const f32x4 &v1, h8 &hi, h8 &lo) {
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        u32x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2 * e], v[2 * e + 1]));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r0 = __builtin_fmaf(v[2 * e], one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(h[e] & 0xffffu)));
            const float r1 = __builtin_fmaf(v[2 * e + 1], one, -(float)__builtin_bit_cast(_Float16, (unsigned short)(h[e] >> 16)));
            l[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
        }
        hi = __builtin_bit_cast(h8, h);
        lo = __builtin_bit_cast(h8, l);
    };
    const long b_tiles = (nb + 15) / 16;
    auto b_row = [&](long t) {   // as A operand: lane (n, g) brings b row n of tile t (clamped: masked below)
        const long r = t * 16 + n;
        return b + (r < nb ? r : nb - 1) * 128;
    };
    // the wave's first b tile is requested together with its a rows: one round trip to memory in front of the loop, not two
    const Raw ra = load_row(a + arow_c * 128);
    Raw cur = load_row(b_row(wave < b_tiles ? wave : 0));
    // a fragments: lane (n, g) holds a[row n][32 s + 8 g + j] of k-step s
    h8 ah[4], al[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) split8(ra.v[2 * s], ra.v[2 * s + 1], ah[s], al[s]);
    const unsigned lo_x = excl_lo && arow < na ? excl_lo[arow] : 0u, hi_x = excl_lo && arow < na ? excl_hi[arow] : 0u;
    float best = -INFINITY, second = -INFINITY;
    int best_i = -1;
    for (long t = wave; t < b_tiles; t += kSmallWaves) {
        const long tn = t + kSmallWaves < b_tiles ? t + kSmallWaves : t;
        const Raw nxt = load_row(b_row(tn));                // in flight while this tile is split and multiplied
        // (the scheduler sinks these eight requests into the tile's arithmetic -- 60 VGPRs of the 128 the launch bounds allow;
        //  fencing them here, all in flight before the tile's first instruction at 127 VGPRs, was measured in round 5 and is
        //  SLOWER: 2000 x 2000 38 us instead of 35, 500 x 500 17 instead of 14)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            h8 bh, bl;
            split8(cur.v[2 * s], cur.v[2 * s + 1], bh, bl);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[s], acc, 0, 0, 0);
        }
        // the lane holds (a row n) x (b rows 16 t + 4 g + i), ascending: the later index wins among equals.  (The scan's
        // "does this tile hold a new best at all" short cut was measured here and lost 3 us at 2000 x 2000: with eight tiles
        // per wave the running best is still settling in most of them.)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned row = (unsigned)(t * 16 + 4 * g + i);
            const bool masked = row >= (unsigned)nb || (row >= lo_x && row < hi_x);
            const float v = masked ? -INFINITY : acc[i];
            const bool nb_ = v >= best && v > -INFINITY;
            const bool ns = !nb_ && v > second;
            second = nb_ ? best : (ns ? v : second);
            best_i = nb_ ? (int)row : best_i;
            best = nb_ ? v : best;
        }
        cur = nxt;
    }
    // fold: the four row groups of a column (lanes n, n + 16, n + 32, n + 48), then the waves
    auto fold = [](float &b0, int &i0, float &s0, float ob, int oi, float os) {
        const bool other = ob > b0 || (ob == b0 && oi > i0);
        const float ns = other ? fmaxf(b0, os) : fmaxf(s0, ob);
        i0 = other ? oi : i0;
        b0 = other ? ob : b0;
        s0 = ns;
    };
#pragma unroll
    for (int m = 16; m <= 32; m <<= 1) {
        const float ob = __shfl_xor(best, m), os = __shfl_xor(second, m);
        const int oi = __shfl_xor(best_i, m);
        fold(best, best_i, second, ob, oi, os);
    }
    if (g == 0) { s_best[wave][n] = best; s_second[wave][n] = second; s_idx[wave][n] = best_i; }
    __syncthreads();
    float bb = -INFINITY, ss = -INFINITY;
    int bi = -1;
    if (threadIdx.x < 16) {
        bb = s_best[0][n];
        ss = s_second[0][n];
        bi = s_idx[0][n];
#pragma unroll
        for (int w = 1; w < kSmallWaves; ++w) fold(bb, bi, ss, s_best[w][n], s_idx[w][n], s_second[w][n]);
    }
    if (threadIdx.x < 16 && arow < na) {
        match[arow] = (bi >= 0 && (ratio <= 0.f || bb * ratio > ss)) ? bi : -1;
        if (best_out) best_out[arow] = bb;
        if (second_out) second_out[arow] = ss;
    }
}

__global__ __launch_bounds__(64 * kSmallWaves) void match_small(const float *__restrict__ a, long na,
                                                                const float *__restrict__ b, long nb,
                                                                const unsigned *__restrict__ excl_lo,
                                                                const unsigned *__restrict__ excl_hi, float ratio,
                                                                int *__restrict__ match, float *__restrict__ best_out,
                                                                float *__restrict__ second_out,
                                                                unsigned *__restrict__ overflowed_word) {
    if (overflowed_word && blockIdx.x == 0 && threadIdx.x == 0) *overflowed_word = 0u;   // this form redoes nothing
    match_small_block(a, na, b, nb, excl_lo, excl_hi, ratio, match, best_out, second_out, (long)blockIdx.x);
}

// BOTH directions of the reference's example in one launch (examples/match_images/src/main.rs:113-116 matches 1 -> 2 and
// 2 -> 1): the first ceil(na / 16) workgroups match a's rows against b, the others b's rows against a -- the same body with the
// operands' roles exchanged, so each direction decides exactly as a launch of match_small in that direction would.  At
// 2000 x 2000 the two directions are 250 workgroups, one round of the chip: the second direction costs no second launch and
// next to no time.  (Keeping a column top-2 beside the row top-2 and merging it across the a blocks would save the second
// direction's arithmetic, but needs workgroups to meet through memory -- measured twice in round 4 at +40 % and +400 %.)
__global__ __launch_bounds__(64 * kSmallWaves) void match_small_both(const float *__restrict__ a, long na,
                                                                     const float *__restrict__ b, long nb, float ratio,
                                                                     int *__restrict__ match_ab, int *__restrict__ match_ba,
                                                                     unsigned *__restrict__ overflowed_word) {
    if (overflowed_word && blockIdx.x == 0 && threadIdx.x == 0) *overflowed_word = 0u;
    const long blocks_ab = (na + 15) / 16;
    if ((long)blockIdx.x < blocks_ab)
        match_small_block(a, na, b, nb, nullptr, nullptr, ratio, match_ab, nullptr, nullptr, (long)blockIdx.x);
    else
        match_small_block(b, nb, a, na, nullptr, nullptr, ratio, match_ba, nullptr, nullptr, (long)blockIdx.x - blocks_ab);
}

// ---- two-pass form ----------------------------------------------------------------------------------------------
// Screen: s~ = <hi(a), hi(b)> on the matrix cores.  With a = hi(a) + da, |da_k| <= 2^-11 |a_k| (f16 round to nearest;
// below 2^-14 the f16 grid is 2^-24 wide, the MFMA does not flush) and the same for b,
//   |s - s~| = |<da, b> + <a, db> - <da, db>| <= (2^-10 + 2^-22) |a||b| + 2^-25 sqrt(128) (|a| + |b|),
// plus 128 f32 roundings of the accumulation (<= 7.7e-6 |a||b|): eps(a, b) <= 1.01e-3 |a||b| + 2.5e-7 (|a| + |b|).
// `screen_margin` is 2 eps with |b| replaced by the largest candidate norm.  If u is the second-largest s~ of a query,
// two candidates have s >= u - eps, so the true best and second best have s >= u - eps and s~ >= u - 2 eps: every
// candidate with s~ >= u - margin is re-scored exactly, nothing else can be among the two best.  During the scan u is not
// known yet; a lane's running second-largest only grows towards it, so recording against the running value keeps a
// superset.
constexpr int kOverRows = 16384;   // overflowed rows the "few" form of the fallback takes; beyond that every row is redone
constexpr int kOverSplits = 64;
constexpr int kRecCap = 64;   // records per stream (one a row x one of its 4 lanes x one b split); ~2.3 ln(rows scanned) expected

__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ float screen_margin(float a_norm, float b_norm_max) {
    return 2.02e-3f * a_norm * b_norm_max + 5e-7f * (a_norm + b_norm_max);
}

constexpr int kStageTiles = 2;   // b tiles per LDS stage of the screen (one barrier per stage): 2 x 8 KiB, double buffered
constexpr int kSub = 4;          // a sub-tiles of 16 rows per wave (the same 64 a rows as the scan's two tiles of 32)
constexpr int kLanesPerRow = 4;  // lanes that hold values of one a row: streams per (a row, b split)


// The screen uses v_mfma_f32_16x16x32_f16 (the chip holds a higher clock under this shape than under 32x32x16 on
// non-trivial operands, MI355X_MICROARCH.md "DVFS give-back" (7)): b rows on the M side again, so lane (n = lane & 15,
// g = lane >> 4) of an accumulator holds a column n and b rows 4 g .. 4 g + 3.  The operand tiles are the scan's:
// chunk c8 = 4 s + g of a row (k = 8 c8 .. 8 c8 + 7) is the lane's share of k-step s for either operand, and a 32-row
// tile is two 16-row halves.
//
// Registers: 64 hold the wave's a rows, 16 the accumulators of one half tile; to stay within 128 (four waves per SIMD,
// nothing spilled in the loop: a spill reload would wait on vmcnt, i.e. on the LDS-DMA in flight) the running (best,
// second, count) of a lane's stream live in LDS and only the threshold second - margin in a register.  Common path per
// half tile: 16 MFMA, then two max3 and a compare per sub-tile; the rare path (a candidate near the threshold) fetches
// its state from LDS and writes the record.
//
// The splits of a query share their bounds (round 6).  Every b split of a query starts with nothing -- until its running
// second best has grown, every value is a candidate and the rare path is the common one: at 65 536 x 65 536 (4 splits of
// 16 384 rows) the candidate path is 45 % of the kernel's time (timing-only build without it: 0.81 of 1.48 ms), at
// 2^20 x 2^20 (one split) nothing; profiles/r06_matcher.md.  Any split's running second-largest is a lower bound of the
// query's u, so the splits tell each other theirs as they go (shared_floor below): -10 % of the call at 65 536^2, identical
// decisions.  (A separate first pass over a sample of b, leaving a floor for every split, was built too: it takes 17 % off
// the main pass and costs as much itself -- the start of a scan is its most expensive part whoever runs it.  Removed.)
__global__ __launch_bounds__(512, 4) void match_screen(const unsigned char *__restrict__ a_tiles, long na,
                                                       const unsigned char *__restrict__ b_tiles, long nb,
                                                       long tiles_per_split, const unsigned *__restrict__ excl_lo,
                                                       const unsigned *__restrict__ excl_hi,
                                                       const float *__restrict__ a_norms,
                                                       const unsigned *__restrict__ b_max_norm_bits,
                                                       uint2 *rec, uint2 *rec_info,
                                                       int *__restrict__ shared_floor, int share_every) {
    // shared_floor [na] (nullable; float bits as int, -inf before the launch): the splits of a query tell each other their
    // bounds as they go -- every share_every stages a lane publishes the bound of one of the wave's 64 rows (atomic max;
    // positive floats order like their bits) and takes the others' back.  Any split's bound is a lower bound of the query's
    // u (the argument at the top): the splits then warm each other instead of each starting cold.
    __shared__ __attribute__((aligned(16))) unsigned char s_b[2][kStageTiles][kTileBytes / 2];
    __shared__ float2 s_top[kSub][512];                     // a stream's running (best, second)
    __shared__ float s_margin[kSub][512];
    __shared__ float s_floor[kSub][512];                    // what the query's other splits have reached (-inf until they say)
    __shared__ int s_cnt[kSub][512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const long a_tile0 = ((long)blockIdx.x * kWaves + wave) * kATiles;
    const long a_tiles_total = (na + kTileRows - 1) / kTileRows;
    const long b_tiles_total = (nb + kTileRows - 1) / kTileRows;
    const long t_begin = (long)blockIdx.y * tiles_per_split;
    long t_end = t_begin + tiles_per_split;
    t_end = t_end < b_tiles_total ? t_end : b_tiles_total;
    const float b_max = __uint_as_float(*b_max_norm_bits);
    const long arow0 = a_tile0 * kTileRows + n;          // sub-tile j holds a row arow0 + 16 j

    h8 ah[kSub][4];
    float thr[kSub];                                      // second best so far - margin: what a candidate must reach
    unsigned any_lo = 0xffffffffu, any_hi = 0u;           // the union of the wave's excluded ranges
#pragma unroll
    for (int j = 0; j < kSub; ++j) {
        const long arow = arow0 + 16 * j;
        const long at = a_tile0 + (j >> 1) < a_tiles_total ? a_tile0 + (j >> 1) : a_tiles_total - 1;   // idle: redo the last
        const unsigned char *src = a_tiles + at * kTileBytes + (g * 32 + 16 * (j & 1) + n) * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) ah[j][s] = *reinterpret_cast<const h8 *>(src + s * 2048);
        const bool live = arow < na;
        const float fl = -INFINITY;
        thr[j] = live ? fl : INFINITY;                    // a dead lane never has a candidate
        s_top[j][threadIdx.x] = make_float2(-INFINITY, -INFINITY);
        s_cnt[j][threadIdx.x] = 0;
        s_floor[j][threadIdx.x] = fl;
        s_margin[j][threadIdx.x] = live ? screen_margin(a_norms[arow], b_max) : 0.f;
        if (live) {
            rec_info[((long)blockIdx.y * na + arow) * kLanesPerRow + g] = make_uint2(0u, 0xff800000u);   // (count, lost)
            if (excl_lo) {
                any_lo = min(any_lo, excl_lo[arow]);
                any_hi = max(any_hi, excl_hi[arow]);
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        any_lo = min(any_lo, (unsigned)__shfl_xor((int)any_lo, m));
        any_hi = max(any_hi, (unsigned)__shfl_xor((int)any_hi, m));
    }
    any_lo = __builtin_amdgcn_readfirstlane(any_lo);
    any_hi = __builtin_amdgcn_readfirstlane(any_hi);

    auto issue = [&](long t, int buf) {   // the hi halves of the stage's tiles: 8 KiB each = 512 threads x 16 B
#pragma unroll
        for (int u = 0; u < kStageTiles; ++u)
            if (t + u < t_end)
                lds_dma16(b_tiles + (t + u) * kTileBytes + threadIdx.x * 16, &s_b[buf][u][0] + wave * 1024);
    };
    // a candidate near the threshold in sub-tile j, rows 16 hf .. of tile t: record, update the stream's state
    auto near = [&](int j, long t, int hf, f32x4 v4) {
        const long arow = arow0 + 16 * j;
        if (!(arow < na)) return;
        const long sidx = ((long)blockIdx.y * na + arow) * kLanesPerRow + g;
        // The four lanes of an a row share what they know: the row's second best is at least every lane's second and at
        // least the second largest of the lanes' bests.  Values only grow, so whatever a lane reads of the others (they
        // may be in here too) is a valid bound; a lane that comes with a stale threshold leaves again at once.
        const float margin = s_margin[j][threadIdx.x];
        const float2 *row_top = &s_top[j][threadIdx.x & ~48];
        auto row_bound = [&]() {
            float b1 = -INFINITY, b2 = -INFINITY, s2 = -INFINITY;
#pragma unroll
            for (int k = 0; k < kLanesPerRow; ++k) {
                const float2 o = row_top[16 * k];
                b2 = o.x > b1 ? b1 : fmaxf(b2, o.x);
                b1 = fmaxf(b1, o.x);
                s2 = fmaxf(s2, o.y);
            }
            return fmaxf(fmaxf(b2, s2) - margin, s_floor[j][threadIdx.x]);
        };
        float from = row_bound();
        thr[j] = from;
        if (!(fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3])) >= from)) return;
        int cnt = s_cnt[j][threadIdx.x];
        float best = row_top[16 * g].x, second = row_top[16 * g].y;
        const int row0 = (int)(t * kTileRows) + 16 * hf + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = v4[e];
            if (v > -INFINITY && v >= fmaxf(from, second - margin)) {
                // a ring: beyond the capacity the oldest record goes, and the largest value that went is kept --
                // verify needs only to know that nothing it would have re-scored was lost
#ifndef LF_SCREEN_ABLATE_STORE   // (timing-only build: the candidate path without its record stores)
                uint2 *slot = rec + sidx * kRecCap + (cnt & (kRecCap - 1));
                if (cnt >= kRecCap) rec_info[sidx].y = __float_as_uint(fmaxf(__uint_as_float(rec_info[sidx].y),
                                                                             __uint_as_float(slot->x)));
                *slot = make_uint2(__float_as_uint(v), (unsigned)(row0 + e));
#endif
                ++cnt;
            }
            const bool nb_ = v >= best && v > -INFINITY;
            const bool ns = !nb_ && v > second;
            second = nb_ ? best : (ns ? v : second);
            best = nb_ ? v : best;
        }
        s_cnt[j][threadIdx.x] = cnt;
        s_top[j][threadIdx.x] = make_float2(best, second);
        thr[j] = row_bound();
    };
    auto tile = [&](long t, const unsigned char *bb) {   // bb: the lane's chunk of row n of the tile, k-step 0
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                  // the tile's two halves of 16 rows, one after the other
            h8 bh[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) bh[s] = *reinterpret_cast<const h8 *>(bb + hf * 256 + s * 2048);
            f32x4 acc[kSub];
#pragma unroll
            for (int j = 0; j < kSub; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < kSub; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[s], ah[j][s], acc[j], 0, 0, 0);
            // rows masked for an a row: beyond nb, or inside its own excluded range (scalar test on the wave's union first)
            const unsigned tr0 = (unsigned)(t * kTileRows) + 16 * hf;
            if (tr0 + 16 > (unsigned)nb || (tr0 < any_hi && tr0 + 16 > any_lo)) {
#pragma unroll
                for (int j = 0; j < kSub; ++j) {
                    const long arow = arow0 + 16 * j;
                    const unsigned lo = excl_lo && arow < na ? excl_lo[arow] : 0u, hi = excl_lo && arow < na ? excl_hi[arow] : 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned row = tr0 + 4 * g + e;
                        if (row >= (unsigned)nb || (row >= lo && row < hi)) acc[j][e] = -INFINITY;
                    }
                }
            }
            // (fmaxf would first canonicalise each input: more instructions than the maximum itself)
            float m[kSub];
            bool hit[kSub];
#pragma unroll
            for (int j = 0; j < kSub; ++j) {
                m[j] = max3(max3(acc[j][0], acc[j][1], acc[j][2]), acc[j][3], acc[j][3]);
                hit[j] = m[j] >= thr[j];
            }
#if defined(LF_SCREEN_ABLATE_EPILOGUE) || defined(LF_SCREEN_ABLATE_NEAR)
#pragma unroll
            for (int j = 0; j < kSub; ++j) thr[j] = hit[j] ? m[j] : thr[j];
            continue;
#endif
            if (__builtin_amdgcn_ballot_w64(hit[0] | hit[1] | hit[2] | hit[3])) {   // one branch; rare once the scan is under way
#pragma unroll
                for (int j = 0; j < kSub; ++j)
                    if (hit[j]) near(j, t, hf, acc[j]);
            }
        }
    };
    if (t_begin < t_end) issue(t_begin, 0);
    int shared_seen = (int)0xff800000;        // what the other splits said last time (applied a stage later: no stall)
    const int sj = lane >> 4, sn = lane & 15; // the (sub-tile, row) this lane speaks for when bounds are shared
    const long srow = a_tile0 * kTileRows + sn + 16 * sj;
    for (long t = t_begin; t < t_end; t += kStageTiles) {
        const int buf = (int)(((t - t_begin) / kStageTiles) & 1);
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's pieces of the stage have landed
#ifndef LF_SCREEN_ABLATE_BARRIER
        __syncthreads();                      // ... and everybody's; everybody is also done with the other buffer
#endif
        if (shared_floor && share_every) {
            const long stage = (t - t_begin) / kStageTiles;
            if (stage % share_every == 1 && shared_seen > 0) {
                // a stage after the exchange: the answer has landed with the stage's data; the row's four lanes take it
                const float fl = __int_as_float(shared_seen);
#pragma unroll
                for (int k = 0; k < kLanesPerRow; ++k) {
                    float *p = &s_floor[sj][(threadIdx.x & ~63) + 16 * k + sn];
                    *p = fmaxf(*p, fl);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < kSub; ++j) thr[j] = fmaxf(thr[j], s_floor[j][threadIdx.x]);
            }
            if (stage % share_every == 0 && stage > 0 && srow < na) {
                const float2 *row_top = &s_top[sj][(threadIdx.x & ~63) + sn];
                float b1 = -INFINITY, b2 = -INFINITY, s2 = -INFINITY;
#pragma unroll
                for (int k = 0; k < kLanesPerRow; ++k) {
                    const float2 o = row_top[16 * k];
                    b2 = o.x > b1 ? b1 : fmaxf(b2, o.x);
                    b1 = fmaxf(b1, o.x);
                    s2 = fmaxf(s2, o.y);
                }
                const float mine = fmaxf(fmaxf(b2, s2) - s_margin[sj][(threadIdx.x & ~63) + sn], s_floor[sj][(threadIdx.x & ~63) + sn]);
                shared_seen = mine > 0.f ? atomicMax(shared_floor + srow, __float_as_int(mine))
                                         : __hip_atomic_load(shared_floor + srow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (t + kStageTiles < t_end) issue(t + kStageTiles, buf ^ 1);
#pragma unroll
        for (int u = 0; u < kStageTiles; ++u)
            if (t + u < t_end) tile(t + u, &s_b[buf][u][0] + (g * 32 + n) * 16);
    }
#pragma unroll
    for (int j = 0; j < kSub; ++j)
        if (arow0 + 16 * j < na)
            rec_info[((long)blockIdx.y * na + arow0 + 16 * j) * kLanesPerRow + g].x = (unsigned)s_cnt[j][threadIdx.x];
#if defined(LF_SCREEN_ABLATE_EPILOGUE) || defined(LF_SCREEN_ABLATE_NEAR)
    if (thr[0] + thr[1] + thr[2] + thr[3] == 12345.f) rec_info[0].x = 1u;
#endif
}

// 16 lanes per a row: the survivors of the screen are re-scored as f32 dot products (each lane 8 elements as an fma
// chain, then a fixed shuffle tree), best / second / index decided with the scan's tie rule (the later index wins).
__global__ __launch_bounds__(256) void match_verify(const float *__restrict__ a, long na, const float *__restrict__ b,
                                                    const float *__restrict__ a_norms,
                                                    const unsigned *__restrict__ b_max_norm_bits,
                                                    const uint2 *__restrict__ rec, const uint2 *__restrict__ rec_info,
                                                    int splits, float ratio, int *__restrict__ match,
                                                    float *__restrict__ best_out, float *__restrict__ second_out,
                                                    int *__restrict__ n_over, int *__restrict__ over_rows) {
    const int l = threadIdx.x & 15, gw = (threadIdx.x >> 4) & 3;
    const long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= na) return;                               // whole 16-lane groups leave together
    float av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = a[i * 128 + l * 8 + j];
    const float margin = screen_margin(a_norms[i], __uint_as_float(*b_max_norm_bits));

    // the query's second-largest screened similarity
    float b1 = -INFINITY, b2 = -INFINITY, lost = -INFINITY;
    for (int k = 0; k < kLanesPerRow * splits; ++k) {
        const long stream = ((long)(k / kLanesPerRow) * na + i) * kLanesPerRow + (k % kLanesPerRow);
        const uint2 info = rec_info[stream];
        int c = (int)info.x;
        if (c > kRecCap) lost = fmaxf(lost, __uint_as_float(info.y));
        c = c < kRecCap ? c : kRecCap;
        for (int j = l; j < c; j += 16) {
            const float v = __uint_as_float(rec[stream * kRecCap + j].x);
            b2 = v > b1 ? b1 : fmaxf(b2, v);
            b1 = fmaxf(b1, v);
        }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const float o1 = __shfl_xor(b1, m, 16), o2 = __shfl_xor(b2, m, 16);
        b2 = fmaxf(fminf(b1, o1), fmaxf(b2, o2));
        b1 = fmaxf(b1, o1);
    }
    // (b2 is the second largest of the records that survive; had a lost one been larger, `over` says so)
    const float keep_from = b2 - margin;
    const bool over = lost > -INFINITY && lost >= keep_from;

    float e1 = -INFINITY, e2 = -INFINITY;
    int ei = -1;
    for (int k = 0; k < kLanesPerRow * splits; ++k) {
        const long stream = ((long)(k / kLanesPerRow) * na + i) * kLanesPerRow + (k % kLanesPerRow);
        int c = (int)rec_info[stream].x;
        c = c < kRecCap ? c : kRecCap;
        for (int j0 = 0; j0 < c; j0 += 16) {
            const int j = j0 + l;
            const uint2 rc = j < c ? rec[stream * kRecCap + j] : make_uint2(0xff800000u, 0u);
            const bool pass = j < c && __uint_as_float(rc.x) >= keep_from;
            unsigned todo = (unsigned)(__builtin_amdgcn_ballot_w64(pass) >> (16 * gw)) & 0xffffu;
            while (todo) {
                const int src = __builtin_ctz(todo);
                todo &= todo - 1;
                const int row = __shfl((int)rc.y, src, 16);
                const float4 *bp = reinterpret_cast<const float4 *>(b + (long)row * 128 + l * 8);
                const float4 u0 = bp[0], u1 = bp[1];
                float p = av[0] * u0.x;
                p = fmaf(av[1], u0.y, p); p = fmaf(av[2], u0.z, p); p = fmaf(av[3], u0.w, p);
                p = fmaf(av[4], u1.x, p); p = fmaf(av[5], u1.y, p); p = fmaf(av[6], u1.z, p); p = fmaf(av[7], u1.w, p);
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) p += __shfl_xor(p, m, 16);
                const bool nb_ = p > e1 || (p == e1 && row > ei);
                e2 = nb_ ? e1 : fmaxf(e2, p);
                ei = nb_ ? row : ei;
                e1 = nb_ ? p : e1;
            }
        }
    }
    if (l == 0) {
        if (over) {                                   // the fallback scan redoes this row (and overwrites what follows)
            const int j = atomicAdd(n_over, 1);
            atomicAdd(n_over + 1, 1);                     // (the call's total, for lf_mkd_match_overflowed)
            if (j < kOverRows) over_rows[j] = (int)i;
        }
        match[i] = (ei >= 0 && (ratio <= 0.f || e1 * ratio > e2)) ? ei : -1;
        if (best_out) best_out[i] = e1;
        if (second_out) second_out[i] = e2;
    }
}

size_t match_tiles_bytes(long n) { return (size_t)((n + kTileRows - 1) / kTileRows) * kTileBytes; }

int match_splits(long na, long nb, int num_cus) {
    const long a_blocks = (na + kWaves * kATiles * kTileRows - 1) / (kWaves * kATiles * kTileRows);
    const long b_tiles = (nb + kTileRows - 1) / kTileRows;
    long want = (2L * num_cus + a_blocks - 1) / a_blocks;   // about two workgroups per CU in flight
    want = want < 1 ? 1 : want;
    want = want > b_tiles ? b_tiles : want;
    want = want > 1024 ? 1024 : want;
    if (const char *e = getenv("LF_MKD_MATCH_SPLITS")) want = atol(e) < b_tiles ? atol(e) : b_tiles;   // (experiments)
    return (int)(want < 1 ? 1 : want);
}

void launch_match_split(const float *x, long n, unsigned char *tiles, float *norms, unsigned *max_norm_bits,
                        hipStream_t stream) {
    if (n <= 0) return;
    const long rows = (n + kTileRows - 1) / kTileRows * kTileRows;
    hipLaunchKernelGGL(match_split, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, x, n, tiles, norms,
                       max_norm_bits);
}

static void launch_scan_merge(const unsigned char *a_tiles, long na, const unsigned char *b_tiles, long nb,
                              const unsigned *excl_lo, const unsigned *excl_hi, float ratio, int splits, float *p_best,
                              int *p_index, float *p_second, int *match, float *best, float *second, MatchGate gate,
                              hipStream_t stream) {
    const long a_blocks = (na + kWaves * kATiles * kTileRows - 1) / (kWaves * kATiles * kTileRows);
    const long n_b_tiles = (nb + kTileRows - 1) / kTileRows;
    const long per = (n_b_tiles + splits - 1) / splits;
    hipLaunchKernelGGL(match_scan, dim3((unsigned)a_blocks, (unsigned)splits), dim3(512), 0, stream, a_tiles, na, b_tiles,
                       nb, per, excl_lo, excl_hi, p_best, p_index, p_second, gate);
    hipLaunchKernelGGL(match_merge, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, stream, (const float *)p_best,
                       (const int *)p_index, (const float *)p_second, na, splits, ratio, match, best, second, gate);
}

void launch_match(const unsigned char *a_tiles, long na, const unsigned char *b_tiles, long nb, const unsigned *excl_lo,
                  const unsigned *excl_hi, float ratio, int splits, float *p_best, int *p_index, float *p_second,
                  int *match, float *best, float *second, const int *n_over, hipStream_t stream) {
    if (na <= 0) return;
    launch_scan_merge(a_tiles, na, b_tiles, nb, excl_lo, excl_hi, ratio, splits, p_best, p_index, p_second, match, best,
                      second, MatchGate{n_over, nullptr, kOverRows, 0}, stream);
}

// every workgroup walks all of b: beyond a few thousand candidates the scan's shared operand tiles win (measured: 8192 x 2048
// and 1000 x 16000 are faster through the scan, 4000 x 4000 is level: the limit is half of that)
// (a workgroup's 16 waves take b's 16-row tiles round-robin: with fewer than 8 tiles most of them idle, which only a request of few
// a rows -- launch-bound whatever the form -- can afford: a million a rows against eight candidates go through the scan)
bool match_small_fits(long na, long nb) {
    return na > 0 && nb >= 2 && nb <= 4096 && (double)na * (double)nb <= 8388608.0 && (nb >= 128 || na <= 4096);
}

void launch_match_small(const float *a, long na, const float *b, long nb, const unsigned *excl_lo, const unsigned *excl_hi,
                        float ratio, int *match, float *best, float *second, unsigned *overflowed_word,
                        hipStream_t stream) {
    if (na <= 0) return;
    hipLaunchKernelGGL(match_small, dim3((unsigned)((na + 15) / 16)), dim3(64 * kSmallWaves), 0, stream, a, na, b, nb,
                       excl_lo, excl_hi, ratio, match, best, second, overflowed_word);
}

void launch_match_small_both(const float *a, long na, const float *b, long nb, float ratio, int *match_ab, int *match_ba,
                             unsigned *overflowed_word, hipStream_t stream) {
    if (na <= 0 || nb <= 0) return;
    hipLaunchKernelGGL(match_small_both, dim3((unsigned)((na + 15) / 16 + (nb + 15) / 16)), dim3(64 * kSmallWaves), 0, stream, a,
                       na, b, nb, ratio, match_ab, match_ba, overflowed_word);
}

size_t match_few_tiles_bytes() { return match_tiles_bytes(kOverRows); }
size_t match_few_words() { return (size_t)kOverRows * (3 + 3 * kOverSplits); }   // rows, lo, hi + partials

// the overflowed rows alone (0 < *n_over <= kOverRows): gathered, scanned with the three terms, scattered back
void launch_match_few(const float *a, const unsigned char *b_tiles, long nb, const unsigned *excl_lo,
                      const unsigned *excl_hi, float ratio, const int *n_over, unsigned char *few_tiles, unsigned *few_words,
                      int *match, float *best, float *second, hipStream_t stream) {
    int *rows = reinterpret_cast<int *>(few_words);
    unsigned *lo = few_words + kOverRows, *hi = few_words + 2 * kOverRows;
    float *p_best = reinterpret_cast<float *>(few_words + 3 * kOverRows);
    float *p_second = p_best + (size_t)kOverSplits * kOverRows;
    int *p_index = reinterpret_cast<int *>(p_second + (size_t)kOverSplits * kOverRows);
    hipLaunchKernelGGL(match_split_rows, dim3(kOverRows / 16), dim3(256), 0, stream, a, (const int *)rows, n_over,
                       kOverRows, excl_lo, excl_hi, few_tiles, lo, hi);
    const long n_b_tiles = (nb + kTileRows - 1) / kTileRows;
    const int splits = (int)(n_b_tiles < kOverSplits ? n_b_tiles : kOverSplits);
    launch_scan_merge(few_tiles, kOverRows, b_tiles, nb, excl_lo ? lo : nullptr, excl_lo ? hi : nullptr, ratio, splits,
                      p_best, p_index, p_second, match, best, second, MatchGate{n_over, rows, 0, kOverRows}, stream);
}

size_t match_record_bytes(long na, int splits) { return (size_t)splits * na * kLanesPerRow * kRecCap * sizeof(uint2); }
size_t match_count_bytes(long na, int splits) { return (size_t)splits * na * kLanesPerRow * sizeof(uint2); }

void launch_match_screen(const unsigned char *a_tiles, long na, const unsigned char *b_tiles, long nb,
                         const unsigned *excl_lo, const unsigned *excl_hi, int splits, const float *a_norms,
                         const unsigned *b_max_norm_bits, void *rec, void *rec_info, hipStream_t stream, int *shared_floor) {
    if (na <= 0) return;
    const long a_blocks = (na + kWaves * kATiles * kTileRows - 1) / (kWaves * kATiles * kTileRows);
    const long n_b_tiles = (nb + kTileRows - 1) / kTileRows;
    const long per = (n_b_tiles + splits - 1) / splits;
    // the splits share their bounds where there are several of moderate length (LF_MKD_MATCH_SHARE=stages, 0 = off)
    int share_every = shared_floor && splits >= 2 && per <= 4096 ? 8 : 0;
    if (const char *e = getenv("LF_MKD_MATCH_SHARE")) share_every = shared_floor && splits >= 2 ? atoi(e) : 0;
    if (share_every > 0)
        (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(shared_floor), (int)0xff800000, (size_t)na, stream);
    hipLaunchKernelGGL(match_screen, dim3((unsigned)a_blocks, (unsigned)splits), dim3(512), 0, stream, a_tiles, na, b_tiles,
                       nb, per, excl_lo, excl_hi, a_norms, b_max_norm_bits, static_cast<uint2 *>(rec),
                       static_cast<uint2 *>(rec_info), share_every > 0 ? shared_floor : nullptr, share_every);
}

void launch_match_verify(const float *a, long na, const float *b, const float *a_norms, const unsigned *b_max_norm_bits,
                         const void *rec, const void *rec_info, int splits, float ratio, int *match, float *best,
                         float *second, int *n_over, int *over_rows, hipStream_t stream) {
    if (na <= 0) return;
    hipLaunchKernelGGL(match_verify, dim3((unsigned)((na + 15) / 16)), dim3(256), 0, stream, a, na, b, a_norms,
                       b_max_norm_bits, static_cast<const uint2 *>(rec), static_cast<const uint2 *>(rec_info), splits, ratio,
                       match, best, second,
                       n_over, over_rows);
}

}  // namespace lfmkd
