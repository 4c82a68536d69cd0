// Exercises include/local_features.hpp (the C++ mirror of the reference crate's API) against liblf_mkd.so.
// usage: demo_local_features MODEL_DIR IMAGE.f32 WIDTH HEIGHT OUT_PREFIX
// Writes OUT_PREFIX.{all,top,filt}.{kps,desc} (raw f32) and prints the counts; tests/test_gpu_cpp_api.py compares them
// with what the Python binding returns for the same image.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "local_features.hpp"

namespace lf = local_features;

// keeps every other blob of size >= 3: a host-side policy only the FilterBlobs hook can express
struct EveryOtherBigBlob : lf::FilterBlobs {
    void filter(const std::vector<lf::Blob> &blobs, std::vector<std::uint32_t> &indices) override {
        bool take = true;
        for (std::uint32_t i = 0; i < blobs.size(); ++i)
            if (blobs[i].size >= 3.f) {
                if (take) indices.push_back(i);
                take = !take;
            }
    }
};

static void dump(const std::string &prefix, const lf::FeaturesResult &r) {
    std::ofstream(prefix + ".kps", std::ios::binary)
        .write(reinterpret_cast<const char *>(r.keypoints.data()), r.keypoints.size() * sizeof(lf::Keypoint));
    std::ofstream(prefix + ".desc", std::ios::binary)
        .write(reinterpret_cast<const char *>(r.descriptors.data()), r.descriptors.size() * sizeof(float));
}

int main(int argc, char **argv) {
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s MODEL_DIR IMAGE.f32 WIDTH HEIGHT OUT_PREFIX\n", argv[0]);
        return 2;
    }
    const std::string model_dir = argv[1], out = argv[5];
    const std::size_t w = std::strtoul(argv[3], nullptr, 10), h = std::strtoul(argv[4], nullptr, 10);
    std::vector<float> img(w * h);
    std::ifstream(argv[2], std::ios::binary).read(reinterpret_cast<char *>(img.data()), img.size() * sizeof(float));
    try {
        lf::BuildTimeParams fixed;
        fixed.max_image_width = std::uint32_t(w);
        fixed.max_image_height = std::uint32_t(h);
        fixed.max_features = 1500;
        fixed.max_blobs = 1000;
        lf::LocalFeaturesHip feats = lf::new_hip(fixed, lf::FeatureDetectParams{}, model_dir);
        const lf::ImageView view{img.data(), h, w};
        const lf::FeaturesResult all = feats.detect_extract_all(view);
        const lf::FeaturesResult top = feats.detect_top_n(view, 100, 0.f);
        EveryOtherBigBlob policy;
        const lf::FeaturesResult filt = feats.detect(view, &policy);
        dump(out + ".all", all);
        dump(out + ".top", top);
        dump(out + ".filt", filt);
        const auto matches = feats.match_features(top.descriptors, all.descriptors);
        std::printf("all %zu top %zu filt %zu matches %zu dropped %u %u\n", all.keypoints.size(), top.keypoints.size(),
                    filt.keypoints.size(), matches.size(), all.dropped_blobs, all.dropped_features);
        // the 8-bit frame and the f32 frame made from it (u8 as f32 / 255., examples/webcam/src/main.rs:136) give the same bits
        std::vector<std::uint8_t> img8(w * h);
        std::vector<float> img8f(w * h);
        for (std::size_t i = 0; i < w * h; ++i) {
            const float v = img[i] < 0.f ? 0.f : (img[i] > 1.f ? 1.f : img[i]);
            img8[i] = std::uint8_t(v * 255.f + 0.5f);
            img8f[i] = float(img8[i]) / 255.f;
        }
        const lf::FeaturesResult from8 = feats.detect_top_n(lf::ImageViewU8{img8.data(), h, w}, 100, 0.f);
        const lf::FeaturesResult from32 = feats.detect_top_n(lf::ImageView{img8f.data(), h, w}, 100, 0.f);
        const bool same8 = from8.keypoints.size() == from32.keypoints.size() && from8.descriptors == from32.descriptors &&
                           std::memcmp(from8.keypoints.data(), from32.keypoints.data(), from8.keypoints.size() * sizeof(from8.keypoints[0])) == 0;
        std::printf("u8 %zu keypoints, same as the f32 frame: %s\n", from8.keypoints.size(), same8 ? "yes" : "NO");
        if (!same8 || from8.keypoints.empty()) return 1;
        // error behaviour: an image larger than max_image_* is an InvalidParameters error, not a crash
        std::vector<float> big((w + 8) * h, 0.5f);
        try {
            feats.detect_extract_all(lf::ImageView{big.data(), h, w + 8});
            std::printf("oversize: no error\n");
            return 1;
        } catch (const lf::LocalFeaturesError &e) {
            std::printf("oversize: %s (%s)\n", e.kind == lf::LocalFeaturesError::Kind::InvalidParameters ? "InvalidParameters" : "Backend", e.what());
        }
    } catch (const lf::LocalFeaturesError &e) {
        std::fprintf(stderr, "LocalFeaturesError: %s\n", e.what());
        return 1;
    }
    return 0;
}
