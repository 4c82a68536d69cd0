// Host-side constants of the MKD path and their device layouts.
//
// What the reference uploads once in upload_constant_data (vulkan/mod.rs:1587-1713) as
// ConstantData (shaders/common.glsl:34-40) is rebuilt here and re-laid-out for the MFMA
// kernels: the spatial kernels become B-operand fragments, the PCA matrix too.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lfmkd {

constexpr int kPatch = 32;
constexpr int kPx = kPatch * kPatch;
constexpr int kDimsIn = 7;      // shaders/common.glsl:22
constexpr int kCart = 9;        // shaders/common.glsl:23
constexpr int kPolar = 25;      // shaders/common.glsl:24
constexpr int kRaw = 238;       // shaders/common.glsl:33
constexpr int kOut = 128;       // shaders/common.glsl:20

// Pooling as a GEMM: rows = patches, K = pixels, columns = (stream, kernel) pairs.
// A "stream" is one per-pixel operand value; there are SEVEN:
//   0: m | 1..3: m cos(k t) | 4..6: m sin(k t)          (t = gradient angle, m = sqrt(|grad|); embedding.glsl:34-51)
// The reference also embeds the RELATIVE angle t + phi(px) (embedding.glsl:70-77, polar kernels).  phi depends on the
// pixel only, so its rotation is folded into the LUT instead of into six more streams:
//   sum_px m cos(k(t+phi)) EP_j = sum_px [m cos kt] (EP_j cos k phi) - [m sin kt] (EP_j sin k phi)
//   sum_px m sin(k(t+phi)) EP_j = sum_px [m sin kt] (EP_j cos k phi) + [m cos kt] (EP_j sin k phi)
// Per harmonic k both streams meet the same four LUT tiles of 16 columns:
//   P0 = EPc[0:16] | Q0 = EPs[0:16] | R = EPc[16:25], EC[0:7] | S = EPs[16:25], EC[7:9], 5 unused
// (EPc_j = c_k EP_j cos k phi, EPs_j = c_k EP_j sin k phi, EC_j scaled by c_k; c_k the von-Mises coefficient), into 8
// products; the m stream keeps its 3 tiles.  24 accumulator tiles in the row loop (the two products of relsin[0:16] share
// one); the epilogue adds / subtracts
// them into the 21 tiles of packed output columns the whitening consumes.
constexpr int kStreams = 7;
constexpr int kAccTiles = 24;     // row loop: 0-2 m | per harmonic h = k-1, base 3 + 7h: cos x P0, sin x Q0, sin x P0 + cos x Q0,
                                  //           cos x R, sin x R, cos x S, sin x S
constexpr int kTiles = 21;        // after the epilogue's combine step (what `colmap` and the whitening fragments index):
                                  // 0-2 m | per harmonic, base 3 + 6h: relcos[0:16] | relsin[0:16] | relcos[16:25],abscos[0:7]
                                  //        | relsin[16:25],abssin[0:7] | abscos[7:9] in slots 9,10 | abssin[7:9] in slots 9,10
constexpr int kTileCols = 16;
constexpr int kPackedCols = kTiles * kTileCols;  // 336
constexpr int kUniqueTiles = 15;  // LUT tiles per patch row: 0-2 m | 3 + 4h + {0: P0, 1: Q0, 2: R, 3: S}

struct PcaModel {
    std::vector<float> mean, eigvals, eigvecs;  // [238], [238], [238*238] row-major
};

// Reads the reference's safetensors PCA model (mkd_ref.rs:352-391). Returns "" or an error text.
std::string load_pca_safetensors(const std::string &path, PcaModel &out);

struct HostConsts {
    // ConstantData, reference layout (kept for tests / debugging)
    std::vector<float> gradient_angle;       // [1024]      phi
    std::vector<float> embedding_polar;      // [25][1024]  EP
    std::vector<float> embedding_cartesian;  // [9][1024]   EC
    std::vector<float> mean;                 // [238]
    std::vector<float> w_t;                  // [128][238]  scaled eigenvectors, transposed

    // device layouts
    std::vector<int16_t> colmap;    // [336] packed column (after the combine step) -> descriptor index (0..237) or -1
    // f32 pooling fragments for v_mfma_f32_16x16x4_f32:
    //   [row y 32][unique tile 15][jg 2][lane 64][e 4] = LUT_col(lane&15)[y][8*(lane>>4) + 4*jg + e]
    std::vector<float> pool_b_f32;
    // f16 hi/lo pooling fragments for v_mfma_f32_16x16x32_f16 (B[k][col], k = 8*(lane>>4)+e):
    //   [row y 32][unique tile 15][hi|lo 2][lane 64][e 8] (uint16 bit patterns); v = hi + lo, both f16
    std::vector<uint16_t> pool_b_f16;
    // The same row images for LF_MKD_POOL_F16_FP6: the hi pieces (and the m stream's three tiles) as above; for the tiles of
    // the harmonics the lo piece is replaced by the lane's operand of v_mfma_scale_f32_16x16x128_f8f6f4 that carries BOTH
    // cross terms of the split: 16 e2m3 fields (12 bytes) -- field 2e = 2048 (v - hi) / T, field 2e + 1 = hi / T for the
    // lane's pixel e -- and a word with the lane's block scale T as an E8M0 byte (tiles P0 and Q0 of a harmonic share T:
    // they meet in one instruction).
    std::vector<uint16_t> pool_b_fp6;
    // Whitening as out^T = W_T x raw with the pooling accumulators as B operand: a lane (patch p, q) holds
    // packed columns 16t + 4q + i in accumulator (t, i).  A fragments = rows of W_T, K ordered to match:
    //   f16 (16x16x32): [step 11][row tile 8][hi|lo 2][lane 64][j 8], lane = (row n = lane&15, q = lane>>4),
    //        value W_T[16r + n][desc(16*(2s + (j>>2)) + 4q + (j&3))], 0 for padding columns
    //   f32 (16x16x4):  [tile 21][i 4][row tile 8][lane 64] = W_T[16r + n][desc(16t + 4q + i)]
    std::vector<uint16_t> white_a_f16;
    std::vector<float> white_a_f32;
    std::vector<float> white_bias;  // [128] = -sum_d W_T[n][d] mean[d]  (whitening.glsl subtracts the mean first)
};

// Returns "" or why the model cannot be used (never aborts: include/lf_mkd.h promises a status for every failure).
std::string build_host_consts(const PcaModel &pca, HostConsts &out);

}  // namespace lfmkd
