// Builds the constants of the MKD path on the host (once per handle).
// Formulas follow the reference's LUT builders, local_features/src/mkd_ref.rs:146-267, and its
// upload step, local_features/src/vulkan/mod.rs:1594-1619; the layouts are this library's own.
#include "mkd_consts.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lfmkd {

namespace {

// von-Mises Fourier coefficients, mkd_ref.rs:7-9
const float kVmN3K8[4] = {0.37872374f, 0.51796234f, 0.46882015f, 0.39798096f};
const float kVmN1K1[2] = {0.618176f, 0.6934725f};
const float kVmN2K8[3] = {0.37872374f, 0.51796234f, 0.46882015f};

struct Grid {
    float x[kPx], y[kPx];
    Grid() {  // mkd_ref.rs:173-185
        for (int r = 0; r < kPatch; ++r)
            for (int c = 0; c < kPatch; ++c) {
                x[r * kPatch + c] = 2.f * float(c) / float(kPatch - 1) - 1.f;
                y[r * kPatch + c] = 2.f * float(r) / float(kPatch - 1) - 1.f;
            }
    }
};

// component a of the (2n+1)-vector [c0, c_k cos(k t), c_k sin(k t)], mkd_ref.rs:146-171
float vm_component(const float *coef, int n, int a, float t) {
    if (a == 0) return coef[0];
    if (a <= n) return std::cos(t * float(a)) * coef[a];
    return std::sin(t * float(a - n)) * coef[a - n];
}

uint16_t f16_bits(float v) {
    _Float16 h = static_cast<_Float16>(v);  // round to nearest even
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
float f16_value(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return static_cast<float>(h);
}

// nearest e2m3 code of |v| <= 7.5 (6 bits: sign, 2 exponent, 3 mantissa; no infinities, no NaNs), ties to the even code
uint32_t e2m3_bits(float v) {
    const float a = std::fabs(v);
    int best = 0;
    float best_d = 1e30f;
    for (int c = 0; c < 32; ++c) {
        const int e = c >> 3, m = c & 7;
        const float val = e == 0 ? float(m) / 8.f : (1.f + float(m) / 8.f) * float(1 << (e - 1));
        const float d = std::fabs(val - a);
        if (d < best_d || (d == best_d && (c & 1) == 0)) {
            best_d = d;
            best = c;
        }
    }
    return uint32_t(best) | (v < 0.f ? 32u : 0u);
}

// One column of a LUT tile: which spatial kernel, rotated by which harmonic of phi.
struct LutColumn {
    bool used;
    bool polar;     // EP or EC
    int j;          // kernel index within EP / EC
    int k;          // harmonic (von-Mises coefficient c_k; 0 for the m stream)
    int rot;        // 0: as is, 1: times cos(k phi), 2: times sin(k phi)
};

LutColumn lut_column(int ut, int c) {
    LutColumn z{false, false, 0, 0, 0};
    if (ut == 0) return {true, true, c, 0, 0};                                        // m: EP 0..15
    if (ut == 1) return c < 9 ? LutColumn{true, true, 16 + c, 0, 0} : LutColumn{true, false, c - 9, 0, 0};   // EP 16..24 | EC 0..6
    if (ut == 2) return c < 2 ? LutColumn{true, false, 7 + c, 0, 0} : z;               // EC 7, 8
    const int k = (ut - 3) / 4 + 1, part = (ut - 3) % 4;
    if (part == 0) return {true, true, c, k, 1};                                       // P0
    if (part == 1) return {true, true, c, k, 2};                                       // Q0
    if (part == 2) return c < 9 ? LutColumn{true, true, 16 + c, k, 1} : LutColumn{true, false, c - 9, k, 0};   // R
    if (c < 9) return {true, true, 16 + c, k, 2};                                      // S
    return c < 11 ? LutColumn{true, false, 7 + (c - 9), k, 0} : z;
}

// Packed output column (tile after the epilogue's combine step, slot) -> descriptor index, or -1.
// Descriptor order (shaders/common.glsl:114-139): polar block [in-dim i][kernel j] (175), then cartesian [i][j] (63);
// in-dims: 0 = m, k = cos k, k + 3 = sin k.
int packed_desc(int tile, int c) {
    const int cart0 = kDimsIn * kPolar;
    if (tile == 0) return c;
    if (tile == 1) return c < 9 ? 16 + c : cart0 + (c - 9);
    if (tile == 2) return c < 2 ? cart0 + 7 + c : -1;
    const int k = (tile - 3) / 6 + 1, part = (tile - 3) % 6;
    const int ic = k, is = k + 3;
    switch (part) {
        case 0: return ic * kPolar + c;
        case 1: return is * kPolar + c;
        case 2: return c < 9 ? ic * kPolar + 16 + c : cart0 + ic * kCart + (c - 9);
        case 3: return c < 9 ? is * kPolar + 16 + c : cart0 + is * kCart + (c - 9);
        case 4: return (c == 9 || c == 10) ? cart0 + ic * kCart + 7 + (c - 9) : -1;
        default: return (c == 9 || c == 10) ? cart0 + is * kCart + 7 + (c - 9) : -1;
    }
}

}  // namespace

std::string load_pca_safetensors(const std::string &path, PcaModel &out) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    std::string err;
    uint64_t hlen = 0;
    std::vector<char> hdr;
    if (std::fread(&hlen, 8, 1, f) != 1 || hlen == 0 || hlen > (1u << 20)) err = "bad header length";
    if (err.empty()) {
        hdr.assign(hlen + 1, 0);
        if (std::fread(hdr.data(), 1, hlen, f) != hlen) err = "truncated header";
    }
    auto tensor = [&](const char *name, size_t count, std::vector<float> &dst) {
        if (!err.empty()) return;
        const std::string key = std::string("\"") + name + "\"";
        const char *p = std::strstr(hdr.data(), key.c_str());
        if (p) p = std::strstr(p, "\"data_offsets\"");
        if (p) p = std::strchr(p, '[');
        long b = 0, e = 0;
        if (!p || std::sscanf(p, "[%ld,%ld]", &b, &e) != 2 || size_t(e - b) != count * 4) {
            err = std::string("tensor '") + name + "' missing or of unexpected size";
            return;
        }
        dst.resize(count);
        std::fseek(f, long(8 + hlen) + b, SEEK_SET);
        if (std::fread(dst.data(), 4, count, f) != count) err = "truncated tensor data";
    };
    tensor("mean", kRaw, out.mean);
    tensor("eigvals", kRaw, out.eigvals);
    tensor("eigvecs", size_t(kRaw) * kRaw, out.eigvecs);
    std::fclose(f);
    return err.empty() ? "" : path + ": " + err;
}

std::string build_host_consts(const PcaModel &pca, HostConsts &hc) {
    // The whitening scale is eigvals^-0.35 (mod.rs:1604-1612): a non-positive or non-finite eigenvalue among the 128 it uses,
    // or a non-finite mean / eigenvector entry, would put NaNs into every descriptor.  The reference unwraps its embedded
    // models (mkd_ref.rs:352-391); a caller-supplied model is checked and refused instead.
    if (pca.mean.size() != size_t(kRaw) || pca.eigvals.size() != size_t(kRaw) || pca.eigvecs.size() != size_t(kRaw) * kRaw)
        return "PCA model: mean[238], eigvals[238] and eigvecs[238*238] are required";
    for (int r = 0; r < kOut; ++r)
        if (!(pca.eigvals[r] > 0.f) || !std::isfinite(pca.eigvals[r]))
            return "PCA model: eigvals[" + std::to_string(r) + "] must be positive and finite (the whitening scales by eigvals^-0.35)";
    for (float v : pca.mean)
        if (!std::isfinite(v)) return "PCA model: mean holds a non-finite value";
    for (float v : pca.eigvecs)
        if (!std::isfinite(v)) return "PCA model: eigvecs holds a non-finite value";
    static const Grid grid;
    const float kPi = 3.14159265358979323846f, kSqrt2 = 1.41421356237309504880f;

    // gradient_angle = -atan2(y, x) of the pixel grid; rho with its epsilon (mkd_ref.rs:133-144)
    hc.gradient_angle.resize(kPx);
    std::vector<float> rho(kPx), gauss(kPx);
    float max_norm = 0.f;
    for (int p = 0; p < kPx; ++p) {
        hc.gradient_angle[p] = -std::atan2(grid.y[p], grid.x[p]);
        rho[p] = std::sqrt(grid.x[p] * grid.x[p] + grid.y[p] * grid.y[p] + 1e-8f);
        max_norm = std::fmax(max_norm, std::sqrt(grid.x[p] * grid.x[p] + grid.y[p] * grid.y[p]));
    }
    for (int p = 0; p < kPx; ++p) {  // mkd_ref.rs:259-267, sigma = 1
        const float nn = std::sqrt(grid.x[p] * grid.x[p] + grid.y[p] * grid.y[p]) / max_norm;
        gauss[p] = std::exp((nn * nn) / -1.f);
    }
    // EC[a*3+b] = vM1(x pi/2)[a] vM1(y pi/2)[b] G   (mkd_ref.rs:210-231, mod.rs:1614-1615)
    hc.embedding_cartesian.resize(size_t(kCart) * kPx);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            for (int p = 0; p < kPx; ++p) {
                const float ea = vm_component(kVmN1K1, 1, a, grid.x[p] * (kPi / 2.f));
                const float eb = vm_component(kVmN1K1, 1, b, grid.y[p] * (kPi / 2.f));
                hc.embedding_cartesian[size_t(a * 3 + b) * kPx + p] = (ea * eb) * gauss[p];
            }
    // EP[a*5+b] = vM2(+atan2)[a] vM2(rho pi/sqrt2)[b] G   (mkd_ref.rs:233-257, mod.rs:1616-1617)
    hc.embedding_polar.resize(size_t(kPolar) * kPx);
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            for (int p = 0; p < kPx; ++p) {
                const float ea = vm_component(kVmN2K8, 2, a, hc.gradient_angle[p] * -1.f);
                const float eb = vm_component(kVmN2K8, 2, b, rho[p] * kPi / kSqrt2);
                hc.embedding_polar[size_t(a * 5 + b) * kPx + p] = (ea * eb) * gauss[p];
            }
    // W_T[r][c] = eigvecs[c][r] * eigvals[r]^(-0.35)   (mod.rs:1604-1612)
    hc.mean = pca.mean;
    hc.w_t.resize(size_t(kOut) * kRaw);
    const float expo = -0.5f * 0.7f;
    for (int r = 0; r < kOut; ++r) {
        const float s = std::pow(pca.eigvals[r], expo);
        for (int c = 0; c < kRaw; ++c) hc.w_t[size_t(r) * kRaw + c] = pca.eigvecs[size_t(c) * kRaw + r] * s;
    }

    // ---- device layouts ----
    hc.colmap.assign(kPackedCols, -1);
    {
        std::vector<int> seen(kRaw, 0);
        for (int t = 0; t < kTiles; ++t)
            for (int c = 0; c < kTileCols; ++c) {
                const int d = packed_desc(t, c);
                hc.colmap[t * kTileCols + c] = int16_t(d);
                if (d >= 0) ++seen[d];
            }
        for (int d = 0; d < kRaw; ++d)
            if (seen[d] != 1)   // every descriptor entry has exactly one packed column (a property of packed_desc alone)
                return "internal: descriptor entry " + std::to_string(d) + " has " + std::to_string(seen[d]) + " packed columns";
    }
    auto lut_value = [&](const LutColumn &col, int px) -> float {
        if (!col.used) return 0.f;
        const double e = col.polar ? hc.embedding_polar[size_t(col.j) * kPx + px]
                                   : hc.embedding_cartesian[size_t(col.j) * kPx + px];
        const double ph = double(col.k) * double(hc.gradient_angle[px]);
        const double rot = col.rot == 1 ? std::cos(ph) : (col.rot == 2 ? std::sin(ph) : 1.0);
        return float(double(kVmN3K8[col.k]) * e * rot);
    };
    hc.pool_b_f32.assign(size_t(kPatch) * kUniqueTiles * 2 * 64 * 4, 0.f);
    hc.pool_b_f16.assign(size_t(kPatch) * kUniqueTiles * 2 * 64 * 8, 0);
    for (int y = 0; y < kPatch; ++y)
        for (int ut = 0; ut < kUniqueTiles; ++ut)
            for (int lane = 0; lane < 64; ++lane) {
                const LutColumn col = lut_column(ut, lane & 15);
                const int q = lane >> 4;
                for (int e = 0; e < 8; ++e) {
                    const int px = y * kPatch + 8 * q + e;
                    const float v = lut_value(col, px);
                    hc.pool_b_f32[(((size_t(y) * kUniqueTiles + ut) * 2 + (e >> 2)) * 64 + lane) * 4 + (e & 3)] = v;
                    const uint16_t hi = f16_bits(v);
                    const uint16_t lo = f16_bits(v - f16_value(hi));  // f16 subnormals survive the MFMA (tools/micro)
                    hc.pool_b_f16[(((size_t(y) * kUniqueTiles + ut) * 2 + 0) * 64 + lane) * 8 + e] = hi;
                    hc.pool_b_f16[(((size_t(y) * kUniqueTiles + ut) * 2 + 1) * 64 + lane) * 8 + e] = lo;
                }
            }
    // LF_MKD_POOL_F16_FP6: the cross-term operands of the harmonics' tiles (see mkd_consts.hpp)
    hc.pool_b_fp6 = hc.pool_b_f16;
    {
        const float kX = 2048.f;      // the residuals' common factor (the kernel scales its stream residuals by the same)
        auto slot_values = [&](int y, int ut, int lane, float (&sv)[16]) {
            const LutColumn col = lut_column(ut, lane & 15);
            const int q = lane >> 4;
            for (int e = 0; e < 8; ++e) {
                const float v = lut_value(col, y * kPatch + 8 * q + e);
                const float hi = f16_value(f16_bits(v));
                sv[2 * e] = kX * (v - hi);
                sv[2 * e + 1] = hi;
            }
        };
        for (int y = 0; y < kPatch; ++y)
            for (int ut = 3; ut < kUniqueTiles; ++ut)
                for (int lane = 0; lane < 64; ++lane) {
                    float sv[16], other[16];
                    slot_values(y, ut, lane, sv);
                    float mx = 0.f;
                    for (float v : sv) mx = std::fmax(mx, std::fabs(v));
                    const int part = (ut - 3) % 4;
                    if (part < 2) {   // P0 and Q0 of a harmonic share the scale
                        slot_values(y, part == 0 ? ut + 1 : ut - 1, lane, other);
                        for (float v : other) mx = std::fmax(mx, std::fabs(v));
                    }
                    int ex = 0;                                   // T = 2^ex with mx / T in [2, 4)
                    if (mx > 0.f) ex = int(std::floor(std::log2(mx))) - 1;
                    const float inv_t = std::ldexp(1.f, -ex);
                    uint32_t w[4] = {0u, 0u, 0u, uint32_t(127 + ex) & 255u};
                    for (int f = 0; f < 16; ++f) {
                        const uint64_t bits = uint64_t(e2m3_bits(sv[f] * inv_t)) << ((6 * f) & 31);
                        w[(6 * f) / 32] |= uint32_t(bits);
                        if ((6 * f) / 32 + 1 < 3) w[(6 * f) / 32 + 1] |= uint32_t(bits >> 32);
                    }
                    uint16_t *dst = &hc.pool_b_fp6[(((size_t(y) * kUniqueTiles + ut) * 2 + 1) * 64 + lane) * 8];
                    std::memcpy(dst, w, 16);
                }
    }
    auto w_packed = [&](int n, int packed_col) -> float {
        const int d = packed_col < kPackedCols ? hc.colmap[packed_col] : -1;
        return d < 0 ? 0.f : hc.w_t[size_t(n) * kRaw + d];
    };
    hc.white_a_f16.assign(size_t(11) * 8 * 2 * 64 * 8, 0);
    hc.white_a_f32.assign(size_t(kTiles) * 4 * 8 * 64, 0.f);
    for (int r = 0; r < 8; ++r)
        for (int lane = 0; lane < 64; ++lane) {
            const int n = 16 * r + (lane & 15), q = lane >> 4;
            for (int s = 0; s < 11; ++s)
                for (int j = 0; j < 8; ++j) {
                    const float v = w_packed(n, 16 * (2 * s + (j >> 2)) + 4 * q + (j & 3));
                    const uint16_t hi = f16_bits(v);
                    hc.white_a_f16[(((size_t(s) * 8 + r) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    hc.white_a_f16[(((size_t(s) * 8 + r) * 2 + 1) * 64 + lane) * 8 + j] = f16_bits(v - f16_value(hi));
                }
            for (int t = 0; t < kTiles; ++t)
                for (int i = 0; i < 4; ++i)
                    hc.white_a_f32[((size_t(t) * 4 + i) * 8 + r) * 64 + lane] = w_packed(n, 16 * t + 4 * q + i);
        }
    hc.white_bias.assign(kOut, 0.f);
    for (int n = 0; n < kOut; ++n) {
        double b = 0.0;
        for (int d = 0; d < kRaw; ++d) b -= double(hc.w_t[size_t(n) * kRaw + d]) * double(pca.mean[d]);
        hc.white_bias[n] = float(b);
    }
    return "";
}

}  // namespace lfmkd
