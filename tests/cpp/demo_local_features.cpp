// Exercises include/local_features.hpp (the C++ mirror of the reference crate's API) against liblf_mkd.so.
// usage: demo_local_features MODEL_DIR IMAGE.f32 WIDTH HEIGHT OUT_PREFIX
// Writes OUT_PREFIX.{all,top,filt}.{kps,desc} (raw f32) and prints the counts; tests/test_gpu_cpp_api.py compares them
// with what the Python binding returns for the same image.
#include <cstdio>
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
