// local_features.hpp -- C++ face of liblf_mkd.so mirroring the reference crate's public API.
//
// The reference is a Rust crate (no Rust toolchain exists in the build image), so the host side above the C ABI
// (include/lf_mkd.h) is given in C++ with the crate's own names, argument meaning and error behaviour:
//
//   reference (local_features/src/)                         here
//   -------------------------------------------------------------------------------------------------
//   lib.rs:12-15   RAW_DESCRIPTOR_LEN, DESCRIPTOR_LEN, PATCH_SIZE   same constants
//   lib.rs:17-24   struct Keypoint {x, y, size, angle, response}     local_features::Keypoint (= lf_mkd_keypoint)
//   lib.rs:26-32   enum MKDPCA                                       enum class MKDPCA
//   lib.rs:34-52   FeatureDetectParams {patch_scale_factor = 24}     same
//   lib.rs:54-75   BuildTimeParams {n_scales = 4, max_image_*, max_features = 2000, max_blobs = 8000, pca}   same
//   lib.rs:77-83   FeaturesResult {keypoints, descriptors, dropped_blobs, dropped_features}   same (descriptors row-major [n][128])
//   lib.rs:85-92   LocalFeaturesError::{InvalidParameters, VulkanError}   LocalFeaturesError (kind InvalidParameters / Backend)
//   lib.rs:94-100  new_vulkan(fixed, params)                         new_hip(fixed, params, model_dir)
//   vulkan/mod.rs:346-367  detect_extract_all / detect_top_n / detect(img, Option<&mut dyn FilterBlobs>)   same three
//   vulkan/mod.rs:1740-1751  trait FilterBlobs                       struct FilterBlobs (virtual filter())
//
// Differences a caller can observe: keypoints come in a defined order (by extremum, then histogram bin; the
// reference's is unordered), extrema too (frame raster order); the PCA models are read from a directory instead of
// being compiled in (mkd_ref.rs:26-31 embeds them).  Header-only; link with -llf_mkd.
#ifndef LOCAL_FEATURES_HPP
#define LOCAL_FEATURES_HPP

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lf_mkd.h"

namespace local_features {

constexpr std::size_t RAW_DESCRIPTOR_LEN = LF_MKD_RAW_LEN;   // lib.rs:12
constexpr std::size_t DESCRIPTOR_LEN = LF_MKD_DESC_LEN;      // lib.rs:13
constexpr std::size_t PATCH_SIZE = LF_MKD_PATCH_SIZE;        // lib.rs:15

using Keypoint = lf_mkd_keypoint;   // lib.rs:17-24: x, y, size, angle (degrees), response

enum class MKDPCA { LIBERTY, NOTREDAME, YOSEMITE };   // lib.rs:26-32

struct FeatureDetectParams {   // lib.rs:34-52
    float patch_scale_factor = 24.f;
};

struct BuildTimeParams {   // lib.rs:54-75
    std::uint32_t n_scales = 4;
    std::uint32_t max_image_width = 0;
    std::uint32_t max_image_height = 0;
    std::uint32_t max_features = 2000;
    std::uint32_t max_blobs = 8000;
    MKDPCA pca = MKDPCA::LIBERTY;
};

struct FeaturesResult {   // lib.rs:77-83
    std::vector<Keypoint> keypoints;
    std::vector<float> descriptors;   // [keypoints.size()][DESCRIPTOR_LEN], row-major (Array2<f32>)
    std::uint32_t dropped_blobs = 0;
    std::uint32_t dropped_features = 0;
};

class LocalFeaturesError : public std::runtime_error {   // lib.rs:85-92
public:
    enum class Kind { InvalidParameters, Backend };
    LocalFeaturesError(Kind k, const std::string &msg)
        : std::runtime_error((k == Kind::InvalidParameters ? "Illegal parameter: " : "") + msg), kind(k) {}
    Kind kind;
};

// ArrayView2<f32> of the reference: a borrowed, contiguous, row-major image in [0, 1] (mod.rs:368)
struct ImageView {
    const float *data;
    std::size_t nrows, ncols;
};

// The 8-bit luma frame the reference's callers convert from (examples/match_images/src/main.rs:44-60): same results as the
// f32 frame v / 255, a quarter of the upload (lf_mkd_detect_u8).
struct ImageViewU8 {
    const std::uint8_t *data;
    std::size_t nrows, ncols;
};

// One candidate blob as the host filter sees it (BlobLocationsView, vulkan/shaders.rs:257-319): refined position,
// size and contrast of an extremum.
using Blob = lf_mkd_extremum;

// trait FilterBlobs (vulkan/mod.rs:1740-1751): choose which candidate blobs go on to orientation + description by
// pushing their indices (into `blobs`) to `indices`.
struct FilterBlobs {
    virtual ~FilterBlobs() = default;
    virtual void filter(const std::vector<Blob> &blobs, std::vector<std::uint32_t> &indices) = 0;
};

class LocalFeaturesHip {   // the role of LocalFeaturesVulkan (vulkan/mod.rs:98-131)
public:
    LocalFeaturesHip(const LocalFeaturesHip &) = delete;
    LocalFeaturesHip &operator=(const LocalFeaturesHip &) = delete;
    LocalFeaturesHip(LocalFeaturesHip &&o) noexcept : h_(o.h_), fixed_(o.fixed_) { o.h_ = nullptr; }
    ~LocalFeaturesHip() { lf_mkd_destroy(h_); }

    // detect_extract_all (mod.rs:346-351): every extremum the detector finds, at most max_blobs
    FeaturesResult detect_extract_all(const ImageView &img) { return run(img, 0, 0.f); }

    // detect_top_n (mod.rs:353-361): the n blobs of largest contrast among those with size >= min_size
    FeaturesResult detect_top_n(const ImageView &img, std::uint32_t n, float min_size) { return run(img, n, min_size); }
    // the same from the 8-bit frame (not in the reference: its callers convert to f32 first)
    FeaturesResult detect_top_n(const ImageViewU8 &img, std::uint32_t n, float min_size) {
        if (!img.data || img.nrows == 0 || img.ncols == 0)
            throw LocalFeaturesError(LocalFeaturesError::Kind::InvalidParameters, "empty image");
        FeaturesResult r;
        r.keypoints.resize(fixed_.max_features);
        r.descriptors.resize(std::size_t(fixed_.max_features) * DESCRIPTOR_LEN);
        std::uint64_t m = 0, dropped_blobs = 0, dropped_features = 0;
        check(lf_mkd_detect_u8(h_, img.data, std::uint32_t(img.ncols), std::uint32_t(img.nrows), n, min_size, r.keypoints.data(),
                               r.descriptors.data(), fixed_.max_features, &m, &dropped_blobs, &dropped_features));
        r.keypoints.resize(m);
        r.descriptors.resize(m * DESCRIPTOR_LEN);
        r.dropped_blobs = std::uint32_t(dropped_blobs);
        r.dropped_features = std::uint32_t(dropped_features);
        return r;
    }

    // detect (mod.rs:363-593) with a caller-supplied blob filter: the detect graph, the filter on the host, the
    // extract graph on the blobs it kept.  A null filter keeps everything (= detect_extract_all).
    FeaturesResult detect(const ImageView &img, FilterBlobs *filter_keypoints) {
        if (!filter_keypoints) return detect_extract_all(img);
        check_image(img);
        check(lf_mkd_set_image(h_, img.data, std::uint32_t(img.ncols), std::uint32_t(img.nrows)));
        const std::uint64_t max_extrema = 256ull * ((std::uint64_t(fixed_.max_blobs) + 255) / 256);   // mod.rs:279-286
        std::vector<Blob> blobs(max_extrema);
        std::uint64_t n = 0, dropped_blobs = 0;
        check(lf_mkd_detect_extrema(h_, blobs.data(), max_extrema, &n, &dropped_blobs));
        blobs.resize(n);
        std::vector<std::uint32_t> indices;
        filter_keypoints->filter(blobs, indices);
        std::vector<Blob> kept;
        kept.reserve(indices.size());
        for (std::uint32_t i : indices) {
            if (i >= blobs.size())
                throw LocalFeaturesError(LocalFeaturesError::Kind::InvalidParameters, "FilterBlobs returned an index out of range");
            kept.push_back(blobs[i]);
        }
        FeaturesResult r;
        r.dropped_blobs = std::uint32_t(dropped_blobs);
        r.keypoints.resize(fixed_.max_features);
        std::uint64_t n_kp = 0, dropped_features = 0;
        check(lf_mkd_orient_keypoints(h_, kept.data(), kept.size(), r.keypoints.data(), fixed_.max_features, &n_kp,
                                      &dropped_features));
        r.keypoints.resize(n_kp);
        r.dropped_features = std::uint32_t(dropped_features);
        r.descriptors.resize(n_kp * DESCRIPTOR_LEN);
        check(lf_mkd_describe_keypoints(h_, r.keypoints.data(), n_kp, r.descriptors.data()));
        return r;
    }

    // match_features of the reference's example (examples/match_images/src/main.rs:8-27)
    std::vector<std::pair<std::size_t, std::size_t>> match_features(const std::vector<float> &a, const std::vector<float> &b) {
        const std::size_t na = a.size() / DESCRIPTOR_LEN, nb = b.size() / DESCRIPTOR_LEN;
        std::vector<std::int32_t> m(na, -1);
        check(lf_mkd_match(h_, a.data(), na, b.data(), nb, 0.8f, m.data()));
        std::vector<std::pair<std::size_t, std::size_t>> res;
        for (std::size_t i = 0; i < na; ++i)
            if (m[i] >= 0) res.emplace_back(i, std::size_t(m[i]));
        return res;
    }

    lf_mkd *handle() { return h_; }

private:
    friend LocalFeaturesHip new_hip(const BuildTimeParams &, const FeatureDetectParams &, const std::string &);
    LocalFeaturesHip(lf_mkd *h, const BuildTimeParams &fixed) : h_(h), fixed_(fixed) {}

    void check(int rc) const {
        if (rc != LF_MKD_OK)
            throw LocalFeaturesError(rc == LF_MKD_ERR_BAD_ARG ? LocalFeaturesError::Kind::InvalidParameters
                                                               : LocalFeaturesError::Kind::Backend,
                                     lf_mkd_last_error(h_));
    }
    void check_image(const ImageView &img) const {
        if (!img.data || img.nrows == 0 || img.ncols == 0)
            throw LocalFeaturesError(LocalFeaturesError::Kind::InvalidParameters, "empty image");
    }
    FeaturesResult run(const ImageView &img, std::uint32_t top_n, float min_size) {
        check_image(img);
        FeaturesResult r;
        r.keypoints.resize(fixed_.max_features);
        r.descriptors.resize(std::size_t(fixed_.max_features) * DESCRIPTOR_LEN);
        std::uint64_t n = 0, dropped_blobs = 0, dropped_features = 0;
        check(lf_mkd_detect(h_, img.data, std::uint32_t(img.ncols), std::uint32_t(img.nrows), top_n, min_size,
                            r.keypoints.data(), r.descriptors.data(), fixed_.max_features, &n, &dropped_blobs,
                            &dropped_features));
        r.keypoints.resize(n);
        r.descriptors.resize(n * DESCRIPTOR_LEN);
        r.dropped_blobs = std::uint32_t(dropped_blobs);
        r.dropped_features = std::uint32_t(dropped_features);
        return r;
    }

    lf_mkd *h_;
    BuildTimeParams fixed_;
};

// new_vulkan (lib.rs:94-100).  model_dir holds concat-pca-{liberty,notredame,yosemite}.safetensors (the files the
// reference embeds, local_features/models/mkd/); empty = $LF_MKD_MODEL_DIR.
inline LocalFeaturesHip new_hip(const BuildTimeParams &fixed_params, const FeatureDetectParams &params,
                                const std::string &model_dir = std::string()) {
    if (fixed_params.max_image_width == 0 || fixed_params.max_image_height == 0)
        throw LocalFeaturesError(LocalFeaturesError::Kind::InvalidParameters, "max_image_width/height must be set");
    std::string dir = model_dir;
    if (dir.empty())
        if (const char *e = std::getenv("LF_MKD_MODEL_DIR")) dir = e;
    if (dir.empty())
        throw LocalFeaturesError(LocalFeaturesError::Kind::InvalidParameters, "no model directory (argument or LF_MKD_MODEL_DIR)");
    static const char *names[] = {"liberty", "notredame", "yosemite"};
    const std::string path = dir + "/concat-pca-" + names[int(fixed_params.pca)] + ".safetensors";
    lf_mkd_params p{};
    p.max_image_width = fixed_params.max_image_width;
    p.max_image_height = fixed_params.max_image_height;
    p.max_features = fixed_params.max_features;
    p.patch_scale_factor = params.patch_scale_factor;
    p.pool_mode = LF_MKD_POOL_F16X3;
    p.n_scales = fixed_params.n_scales;
    p.max_blobs = fixed_params.max_blobs;
    lf_mkd *h = nullptr;
    const int rc = lf_mkd_create_from_file(&p, path.c_str(), &h);
    if (rc != LF_MKD_OK)
        throw LocalFeaturesError(rc == LF_MKD_ERR_BAD_ARG ? LocalFeaturesError::Kind::InvalidParameters
                                                           : LocalFeaturesError::Kind::Backend,
                                 lf_mkd_last_error(nullptr));
    return LocalFeaturesHip(h, fixed_params);
}

}  // namespace local_features
#endif  // LOCAL_FEATURES_HPP
