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
// A "stream" is one per-pixel A-operand value:
//   0: m | 1..3: m cos(k t) | 4..6: m sin(k t) | 7..9: m cos(k(t+phi)) | 10..12: m sin(k(t+phi))
// (t = gradient angle, m = sqrt(|grad|); embedding.glsl:34-51,70-77).  A 16-column tile
// belongs to exactly one stream; 21 tiles cover the 238 outputs (71 % column efficiency).
constexpr int kStreams = 13;
constexpr int kTiles = 21;
constexpr int kTileCols = 16;
constexpr int kPackedCols = kTiles * kTileCols;  // 336

// tile -> stream (device code keeps an identical constexpr table)
constexpr int kTileStream[kTiles] = {0, 0, 0, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12,
                                     1, 2, 3, 4, 5, 6};

// The cos and the sin stream of one harmonic meet identical LUT columns (same kernels, same von-Mises
// coefficient), so a LUT row stores 12 unique tiles: 0-2 m | 3-5 abs k=1..3 | 6-11 rel k=1..3 (2 each).
constexpr int kUniqueTiles = 12;
constexpr int unique_tile_repr(int ut) { return ut < 3 ? ut : (ut < 6 ? 15 + (ut - 3) : 3 + (ut - 6)); }   // a tile holding it
constexpr int unique_tile_twin(int ut) { return ut < 3 ? ut : (ut < 6 ? 18 + (ut - 3) : 9 + (ut - 6)); }   // its sin twin

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
    std::vector<float> phi_cs;      // [1024][2] cos(phi), sin(phi)
    std::vector<int16_t> colmap;    // [336] packed column -> descriptor index (0..237) or -1
    // f32 pooling fragments for v_mfma_f32_16x16x4_f32:
    //   [row y 32][unique tile 12][jg 2][lane 64][e 4] = coef * E_col(lane&15)[y][8*(lane>>4) + 4*jg + e]
    std::vector<float> pool_b_f32;
    // f16 hi/lo pooling fragments for v_mfma_f32_16x16x32_f16 (B[k][col], k = 8*(lane>>4)+e):
    //   [row y 32][unique tile 12][hi|lo 2][lane 64][e 8] (uint16 bit patterns); v = hi + lo, both f16
    std::vector<uint16_t> pool_b_f16;
    // Whitening as out^T = W_T x raw with the pooling accumulators as B operand: a lane (patch p, q) holds
    // packed columns 16t + 4q + i in accumulator (t, i).  A fragments = rows of W_T, K ordered to match:
    //   f16 (16x16x32): [step 11][row tile 8][hi|lo 2][lane 64][j 8], lane = (row n = lane&15, q = lane>>4),
    //        value W_T[16r + n][desc(16*(2s + (j>>2)) + 4q + (j&3))], 0 for padding columns
    //   f32 (16x16x4):  [tile 21][i 4][row tile 8][lane 64] = W_T[16r + n][desc(16t + 4q + i)]
    std::vector<uint16_t> white_a_f16;
    std::vector<float> white_a_f32;
    std::vector<float> white_bias;  // [128] = -sum_d W_T[n][d] mean[d]  (whitening.glsl subtracts the mean first)
};

void build_host_consts(const PcaModel &pca, HostConsts &out);

}  // namespace lfmkd
