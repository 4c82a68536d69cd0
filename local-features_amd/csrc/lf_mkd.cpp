// C ABI of the MKD path (include/lf_mkd.h): handle management, batching, host<->device staging.
// Product code: it never touches oracle/; without a HIP device every entry point fails loudly.
#include "../../include/lf_mkd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lf_mkd_internal.h"
#include "mkd_consts.hpp"
#include "mkd_device.h"

using namespace lfmkd;

struct lf_mkd {
    lf_mkd_params params{};
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // lf_mkd_detect / the recorded pipeline: pyramid levels >= 1 are built here beside the detector
    std::vector<hipEvent_t> side_events;
    DeviceConsts dc;
    uint64_t batch = 0;         // descriptors per internal batch (multiple of 64)
    float *d_patches = nullptr; // [batch][1024] staging: host patches / sampled patches
    float *d_out = nullptr;     // [batch][128]  staging for host output
    float *d_kps = nullptr;     // [batch][5]
    // keypoint mode
    PyramidDesc pd{};
    float *d_image = nullptr, *d_pyr = nullptr, *d_tmp_a = nullptr, *d_tmp_b = nullptr;
    bool have_image = false;
    uint32_t max_frames = 1, n_frames = 0;  // frames held by the pyramid store / currently loaded
    long pyr_stride = 0;                    // floats between the pyramids of consecutive frames
    // keypoint orientation: a-trous layers 1 .. n_layers-1 per frame (layer 0 = pyramid level 0), allocated on first use
    int n_layers = 7;
    float *d_coarse = nullptr;
    long layer_stride = 0, coarse_stride = 0;  // floats between layers / between frames
    bool coarse_valid = false, coarse_l1_valid = false;
    uint64_t orient_cap = 0;                   // extrema the scratch arrays below hold
    float *d_extrema = nullptr, *d_angles = nullptr, *d_kps_out = nullptr;
    unsigned *d_counts = nullptr, *d_orient_sums = nullptr;
    uint64_t kps_out_cap = 0;
    unsigned long long *d_totals = nullptr;
    unsigned long long *d_clk = nullptr;       // LF_MKD_FLAG_KERNEL_TIMING: clock stamps of the latest describe launch
    // the row-split form of the keypoint kernel (requests of at most 16 x CUs keypoints): partial sums and their counters
    float *d_kp_xchg = nullptr;
    unsigned *d_kp_words = nullptr;
    // detector scratch (allocated on first use): per-cube slots and counts for max_frames frames of the maximum size
    uint64_t max_extrema = 8192;
    float *d_slots = nullptr, *d_det_extrema = nullptr, *d_det_selected = nullptr, *d_det_desc = nullptr;
    unsigned *d_cube_counts = nullptr, *d_cube_sums = nullptr, *d_sel_count = nullptr, *d_topk_work = nullptr;
    uint64_t topk_work_cap = 0;
    uint64_t det_out_cap = 0, det_sel_cap = 0;
    // multi-frame detect (lf_mkd_detect_frames_device)
    float *d_mf_padded = nullptr, *d_mf_list = nullptr;
    unsigned *d_mf_frame_start = nullptr, *d_mf_offsets = nullptr, *d_mf_frame_of = nullptr;
    uint64_t mf_padded_cap = 0, mf_list_cap = 0, mf_start_cap = 0, mf_off_cap = 0, mf_fo_cap = 0;
    // graph-captured per-frame pipeline (lf_mkd_stream_*)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    PyramidDesc graph_pd{};     // frame geometry the recorded pipeline was captured for
    float *d_stream_patches = nullptr;
    uint64_t stream_patch_cap = 0;
    // lf_mkd_detect / lf_mkd_detect_u8: the same launch sequence recorded once per (frame size, top_n, min_size, max_out,
    // pixel type) and kept -- a call is one upload, one graph launch, one wait, the result copies
    struct DetectPlan {
        uint32_t w = 0, h = 0, top_n = 0, min_size_bits = 0;
        uint64_t max_out = 0, stamp = 0;
        bool u8 = false;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        // banded form (frames large enough to be worth it): the frame is uploaded in pieces, rows [0, cuts[0]) first, then
        // [cuts[0], cuts[1]) ...; heads[p] is the share of the pipeline's front the frame's first cuts[p] rows allow beyond
        // heads[p - 1]'s (RowBands) and runs while piece p + 1 is on its way, `graph` is then the last piece's share +
        // everything else
        std::vector<uint32_t> cuts;
        std::vector<hipGraph_t> head_graphs;
        std::vector<hipGraphExec_t> heads;
        PyramidDesc pd{};
    };
    std::vector<DetectPlan> plans;
    uint64_t plan_clock = 0;
    // Requests seen but not recorded yet.  Recording costs a capture and an instantiation (more when the upload is banded) --
    // several times the call itself -- so a request is served by the pipeline's plain launches until it has been seen record_after times
    // (default 1: the reference's match_images detects each image once, at its own size, and never pays a recording; a
    // camera loop records on its second frame and replays from the third).  LF_MKD_DETECT_RECORD_AFTER in the environment,
    // read at creation: 0 records on the first sighting.
    struct Sighting {
        uint32_t w, h, top_n, min_size_bits;
        uint64_t max_out, count, stamp;
        bool u8;
    };
    std::vector<Sighting> sightings;
    uint64_t record_after = 1;
    unsigned char *d_image_u8 = nullptr;              // 8-bit frame(s) on their way to level 0, allocated on first use
    unsigned long long *d_det_counts = nullptr;       // [8] the recorded detect pipeline's counts (as lf_mkd_stream_create's d_counts)
    unsigned long long *h_det_counts = nullptr;       // the same in pinned host memory: the last node of a plan copies them here
    // ... and the keypoints, copied there by the pipeline's last node.
    // (The descriptors stay in device memory until the count is known: having the describe kernel store them straight into
    // pinned host memory was measured -- +20 us of kernel time at 100 keypoints, +45 us at 3000, PCIe writes stall its
    // epilogue -- and so was a copy node of all max_out rows followed by a host memcpy of n: reading memory the device has
    // just written runs at 25 GB/s on one core, 60 us for 3000 rows, where the runtime's own copy into the caller's array takes 15.)
    lf_mkd_keypoint *h_res_kps = nullptr;
    uint64_t h_res_cap = 0;
    hipStream_t copy_stream = nullptr;                    // the banded upload: pieces are copied here, the pipeline's parts wait for them
    std::vector<hipEvent_t> copy_ev;
    hipEvent_t det_ev[3] = {nullptr, nullptr, nullptr};   // LF_MKD_FLAG_KERNEL_TIMING: before the upload, after it, after the pipeline
    double det_upload_ms = 0, det_pipeline_ms = 0, det_readback_ms = 0;
    // matcher scratch
    unsigned char *d_match_a = nullptr, *d_match_b = nullptr;
    float *d_match_part = nullptr, *d_match_in = nullptr;
    int *d_match_out = nullptr;
    uint64_t match_a_cap = 0, match_b_cap = 0, match_part_cap = 0, match_in_cap = 0, match_out_cap = 0;
    unsigned char *d_match_rec = nullptr;      // two-pass form: candidate records, their counts, |a| per row, two words
    unsigned char *d_match_cnt = nullptr;      // (largest |b| as float bits, number of overflowed rows)
    float *d_match_norm = nullptr, *d_match_floor = nullptr;   // (floor: the bounds the screen's b splits share, per a row)
    unsigned *d_match_misc = nullptr;
    unsigned char *d_match_few_tiles = nullptr;   // the overflowed rows' own tiles; their indices, exclusion ranges, partials
    unsigned *d_match_few = nullptr;
    uint64_t match_rec_cap = 0, match_cnt_cap = 0, match_norm_cap = 0, match_misc_cap = 0, match_few_tiles_cap = 0,
             match_few_cap = 0, match_floor_cap = 0;
    int num_cus = 256;
    // LF_MKD_FLAG_KERNEL_TIMING: (start, end) of the describe kernel per batch
    std::vector<hipEvent_t> ev_pending, ev_free;
    std::string err;
};

namespace {

thread_local std::string g_create_error;

#define LF_HIP(h, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return LF_MKD_ERR_HIP;                                                          \
        }                                                                                   \
    } while (0)

// Every entry point makes the handle's device current for its own duration and puts the caller's device back on EVERY return
// path (a one-process / one-handle-per-GPU caller -- INTEGRATION.md section 3 -- keeps the current device it had).
#define LF_ENTER(h)              \
    lfmkd::DeviceScope scope_;   \
    LF_HIP(h, scope_.enter((h)->params.device))

int fail(lf_mkd *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    return code;
}

template <typename T>
hipError_t upload(T **dst, const void *src, size_t bytes) {
    hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
}

long pyramid_levels(uint32_t w, uint32_t h) {  // mod.rs:271-277,372-373
    return std::lround(std::ceil(std::log2(float(std::min(w, h)))));
}

// returns the floats one frame's pyramid occupies (every level with its mirrored apron: mkd_device.h)
long describe_pyramid(uint32_t w, uint32_t h, PyramidDesc &pd) {
    pd.levels = int(std::min<long>(std::max<long>(pyramid_levels(w, h), 1), kMaxPyrLevels));
    long off = 0;
    for (int l = 0; l < pd.levels; ++l) {
        pd.w[l] = std::max<int>(int(w >> l), 1);
        pd.h[l] = std::max<int>(int(h >> l), 1);
        pd.apron[l] = kPyrApron;
        pd.pitch[l] = pd.w[l] + 2 * pd.apron[l];
        pd.offset[l] = off + long(pd.apron[l]) * pd.pitch[l] + pd.apron[l];
        off += long(pd.pitch[l]) * (pd.h[l] + 2 * pd.apron[l]);
    }
    return (off + 63) / 64 * 64;   // frames 256 bytes apart at least: level 0 of every frame starts 16-byte aligned (pyr_swt_staged)
}

long pyramid_floats(uint32_t w, uint32_t h) {
    PyramidDesc pd{};
    return describe_pyramid(w, h, pd);
}

int create_impl(const lf_mkd_params *params, const PcaModel &pca, lf_mkd **out) {
    if (!params || !out) return LF_MKD_ERR_BAD_ARG;
    *out = nullptr;
    // the model first: a bad model is the caller's error whatever the machine (and is reported without a device)
    HostConsts hc;
    if (const std::string e = build_host_consts(pca, hc); !e.empty()) {
        g_create_error = e;
        return LF_MKD_ERR_BAD_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || params->device < 0 || params->device >= ndev) {
        g_create_error = "no HIP device " + std::to_string(params->device) + " (devices visible: " +
                         std::to_string(ndev) + ")";
        return LF_MKD_ERR_NO_DEVICE;
    }
    lf_mkd *h = new (std::nothrow) lf_mkd;
    if (!h) return LF_MKD_ERR_BAD_ARG;
    h->params = *params;
    if (h->params.patch_scale_factor == 0.f) h->params.patch_scale_factor = 24.f;  // lib.rs:46
    if (h->params.pool_mode == LF_MKD_POOL_DEFAULT) h->params.pool_mode = LF_MKD_POOL_F16X3;
    if (h->params.pool_mode != LF_MKD_POOL_F16X3 && h->params.pool_mode != LF_MKD_POOL_F32 &&
        h->params.pool_mode != LF_MKD_POOL_F16_FP6) {
        g_create_error = "pool_mode must be LF_MKD_POOL_DEFAULT, LF_MKD_POOL_F16X3, LF_MKD_POOL_F32 or LF_MKD_POOL_F16_FP6";
        delete h;
        return LF_MKD_ERR_BAD_ARG;
    }
    if (h->params.angle_mode < LF_MKD_ANGLE_SHADER || h->params.angle_mode > LF_MKD_ANGLE_EXACT_ZERO) {
        g_create_error = "angle_mode must be one of lf_mkd_angle_mode";
        delete h;
        return LF_MKD_ERR_BAD_ARG;
    }
    const uint64_t mf = params->max_features ? params->max_features : 2000;        // lib.rs:69
    h->n_layers = int(params->n_scales ? params->n_scales : 4) + 3;                // lib.rs:70, mod.rs:1093
    if (h->n_layers - 1 > 8) {   // the extremum scan keeps at most 8 DoG layers of a tile in LDS
        g_create_error = "n_scales must not exceed 6";
        delete h;
        return LF_MKD_ERR_BAD_ARG;
    }
    h->max_extrema = 256 * ((uint64_t(params->max_blobs ? params->max_blobs : 8000) + 255) / 256);  // mod.rs:279-286
    if (const char *e = getenv("LF_MKD_DETECT_RECORD_AFTER")) h->record_after = uint64_t(std::max(0, atoi(e)));
    h->batch = (mf + 63) / 64 * 64;
    auto bail = [&](int code) {
        g_create_error = h->err;
        lf_mkd_destroy(h);
        return code;
    };
#define LF_CREATE_HIP(call)                                                   \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);       \
            return bail(LF_MKD_ERR_HIP);                                      \
        }                                                                     \
    } while (0)
    lfmkd::DeviceScope scope_;
    LF_CREATE_HIP(scope_.enter(params->device));
    LF_CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {
        hipDeviceProp_t prop;
        LF_CREATE_HIP(hipGetDeviceProperties(&prop, params->device));
        h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    LF_CREATE_HIP(upload(&h->dc.colmap, hc.colmap.data(), hc.colmap.size() * 2));
    LF_CREATE_HIP(upload(&h->dc.pool_b_f32, hc.pool_b_f32.data(), hc.pool_b_f32.size() * 4));
    LF_CREATE_HIP(upload(&h->dc.pool_b_f16, hc.pool_b_f16.data(), hc.pool_b_f16.size() * 2));
    if (h->params.pool_mode == LF_MKD_POOL_F16_FP6)
        LF_CREATE_HIP(upload(&h->dc.pool_b_fp6, hc.pool_b_fp6.data(), hc.pool_b_fp6.size() * 2));
    LF_CREATE_HIP(upload(&h->dc.white_a_f16, hc.white_a_f16.data(), hc.white_a_f16.size() * 2));
    LF_CREATE_HIP(upload(&h->dc.white_a_f32, hc.white_a_f32.data(), hc.white_a_f32.size() * 4));
    LF_CREATE_HIP(upload(&h->dc.white_bias, hc.white_bias.data(), hc.white_bias.size() * 4));
    // (the staging buffers of the host-pointer and keypoint entry points -- 4.6 KiB per descriptor of the internal batch --
    // are allocated on first use: a caller of the device-pointer patch API never needs them)
    LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_totals), 8 * sizeof(unsigned long long)));
    if (h->params.flags & LF_MKD_FLAG_KERNEL_TIMING) {
        LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_clk), 4 * sizeof(unsigned long long)));
        LF_CREATE_HIP(hipMemset(h->d_clk, 0, 4 * sizeof(unsigned long long)));
    }
    if (params->max_image_width && params->max_image_height) {
        // the sampler addresses a pyramid level with 32-bit byte offsets from its first texel
        if ((uint64_t(params->max_image_width) + 2 * kPyrApron) * (uint64_t(params->max_image_height) + 2 * kPyrApron) >= (1ull << 30) ||
            params->max_image_width >= (1u << 20) || params->max_image_height >= (1u << 20)) {
            h->err = "max_image_width x max_image_height (with the pyramid's apron of 48 texels a side) must stay below 2^30 pixels";
            return bail(LF_MKD_ERR_BAD_ARG);
        }
        h->max_frames = params->max_frames ? params->max_frames : 1;
        const size_t px = size_t(params->max_image_width) * params->max_image_height * h->max_frames;
        h->pyr_stride = pyramid_floats(params->max_image_width, params->max_image_height);
        LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_image), px * 4));
        LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_tmp_a), px * 4));
        LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_tmp_b), px * 4));
        LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_pyr), size_t(h->pyr_stride) * h->max_frames * 4));
        if (h->params.pool_mode == LF_MKD_POOL_F16X3 && !(h->params.flags & LF_MKD_FLAG_UNFUSED_KEYPOINTS)) {
            LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_kp_xchg), kp_split_exchange_bytes(h->num_cus)));
            LF_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&h->d_kp_words), kp_split_counter_words(h->num_cus) * 4));
            LF_CREATE_HIP(hipMemset(h->d_kp_words, 0, kp_split_counter_words(h->num_cus) * 4));
        }
    }
#undef LF_CREATE_HIP
    *out = h;
    return LF_MKD_OK;
}

// staging for one internal batch: sampled / uploaded patches; descriptors on their way to the host and uploaded keypoints
int ensure_patch_staging(lf_mkd *h) {
    if (h->d_patches) return LF_MKD_OK;
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_patches), h->batch * kPx * 4));
    return LF_MKD_OK;
}
int ensure_io_staging(lf_mkd *h) {
    if (h->d_out) return LF_MKD_OK;
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_out), h->batch * kOut * 4));
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_kps), h->batch * 5 * 4));
    return LF_MKD_OK;
}
int ensure_staging(lf_mkd *h) {
    if (int rc = ensure_patch_staging(h)) return rc;
    return ensure_io_staging(h);
}

// LF_MKD_FLAG_KERNEL_TIMING: an event on the launch stream before and after every describe launch
int mark(lf_mkd *h, hipStream_t s) {
    if (!(h->params.flags & LF_MKD_FLAG_KERNEL_TIMING)) return LF_MKD_OK;
    hipEvent_t e;
    if (!h->ev_free.empty()) {
        e = h->ev_free.back();
        h->ev_free.pop_back();
    } else {
        LF_HIP(h, hipEventCreate(&e));
    }
    h->ev_pending.push_back(e);
    LF_HIP(h, hipEventRecord(e, s));
    return LF_MKD_OK;
}

constexpr uint64_t kMatchChunk = 1 << 20;   // a rows per pass of the two-pass matcher (2 GiB of records at one b split)

int ensure_side_stream(lf_mkd *h, size_t events) {
    if (!h->side_stream) LF_HIP(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    while (h->side_events.size() < events) {
        hipEvent_t e;
        LF_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->side_events.push_back(e);
    }
    return LF_MKD_OK;
}

// patches of one batch, already resident on the device -> descriptors (or the un-whitened 238-D vectors)
int run_batch(lf_mkd *h, const float *d_patches, uint64_t n, float *d_out, float *d_raw, hipStream_t s, int waves = 0) {
    if (int rc = mark(h, s)) return rc;
    launch_describe(d_patches, long(n), nullptr, h->dc, h->params.angle_mode, h->params.pool_mode, d_out ? d_out : h->d_out,
                    d_raw, h->num_cus, s, waves, h->d_clk);
    LF_HIP(h, hipGetLastError());
    if (int rc = mark(h, s)) return rc;
    return LF_MKD_OK;
}

// Keypoints resident on the device -> descriptors.  The product form is ONE launch: producer waves of the describe kernel
// sample the patches into its LDS ring (f16x3 pooling).  The f32 verification mode and LF_MKD_FLAG_UNFUSED_KEYPOINTS take
// the two-launch form through the staging patches in HBM, an internal batch at a time; same sampling arithmetic, same bits.
bool fused_keypoints(const lf_mkd *h) {
    return h->params.pool_mode == LF_MKD_POOL_F16X3 && !(h->params.flags & LF_MKD_FLAG_UNFUSED_KEYPOINTS);
}
int describe_keypoints_on_device(lf_mkd *h, const float *d_kps, const uint32_t *d_frame_of, uint64_t n, float *d_out,
                                 hipStream_t s, uint64_t form_n = 0) {
    if (fused_keypoints(h)) {
        if (int rc = mark(h, s)) return rc;
        launch_describe_keypoints(h->d_pyr, h->pyr_stride, h->pd, d_kps, d_frame_of, h->n_frames, long(n), nullptr,
                                  h->params.patch_scale_factor, h->dc, h->params.angle_mode, d_out, h->num_cus, s, h->d_clk,
                                  h->d_kp_xchg, h->d_kp_words, long(form_n));
        LF_HIP(h, hipGetLastError());
        return mark(h, s);
    }
    if (int rc = ensure_patch_staging(h)) return rc;
    for (uint64_t off = 0; off < n; off += h->batch) {
        const uint64_t m = std::min<uint64_t>(h->batch, n - off);
        launch_sample_patches(h->d_pyr, h->pyr_stride, h->pd, d_kps + off * 5, d_frame_of ? d_frame_of + off : nullptr,
                              h->n_frames, long(m), nullptr, h->params.patch_scale_factor, h->d_patches, s);
        LF_HIP(h, hipGetLastError());
        if (int rc = run_batch(h, h->d_patches, m, d_out + off * kOut, nullptr, s)) return rc;
    }
    return LF_MKD_OK;
}

void destroy_plan(lf_mkd::DetectPlan &p) {
    if (p.exec) (void)hipGraphExecDestroy(p.exec);
    if (p.graph) (void)hipGraphDestroy(p.graph);
    for (hipGraphExec_t e : p.heads) (void)hipGraphExecDestroy(e);
    for (hipGraph_t g : p.head_graphs) (void)hipGraphDestroy(g);
    p.exec = nullptr;
    p.graph = nullptr;
    p.heads.clear();
    p.head_graphs.clear();
}

// A recorded stream pipeline holds raw pointers into the scratch buffers: re-allocating one of them retires the graph
// (lf_mkd_stream_frame then asks for a new lf_mkd_stream_create instead of touching freed memory).
void retire_graph(lf_mkd *h) {
    if (h->graph_exec || !h->plans.empty()) (void)hipDeviceSynchronize();   // a launch may still be running (on any stream)
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->graph_exec = nullptr;
    h->graph = nullptr;
    for (auto &p : h->plans) destroy_plan(p);   // lf_mkd_detect's recordings hold the same raw pointers: the next call records anew
    h->plans.clear();
}

// Extends the loaded frames' level 0 into the a-trous stack (once per set_image*), allocating it on first use.
int ensure_coarse_stack(lf_mkd *h, hipStream_t s) {
    if (h->coarse_valid) return LF_MKD_OK;
    if (!h->d_coarse) {
        h->layer_stride = long(h->params.max_image_width) * h->params.max_image_height;
        h->coarse_stride = h->layer_stride * (h->n_layers - 1);
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_coarse), size_t(h->coarse_stride) * h->max_frames * 4));
    }
    launch_build_coarse_stack(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse, h->coarse_stride, h->layer_stride,
                              h->d_tmp_a, h->n_layers, h->coarse_l1_valid ? 1 : 0, h->pd.w[0], h->pd.h[0],
                              int(h->n_frames), s);
    LF_HIP(h, hipGetLastError());
    h->coarse_valid = true;
    return LF_MKD_OK;
}

int ensure_orient_scratch(lf_mkd *h, uint64_t n, bool staging, uint64_t max_out) {
    if (n > h->orient_cap) {
        retire_graph(h);
        for (void *p : {static_cast<void *>(h->d_extrema), static_cast<void *>(h->d_angles),
                        static_cast<void *>(h->d_counts), static_cast<void *>(h->d_orient_sums)})
            if (p) (void)hipFree(p);
        h->d_extrema = h->d_angles = nullptr;
        h->d_counts = h->d_orient_sums = nullptr;
        h->orient_cap = 0;
        const uint64_t cap = std::max<uint64_t>(n, h->batch);
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_extrema), cap * sizeof(lf_mkd_extremum)));
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_angles), cap * LF_MKD_MAX_ANGLES_PER_EXTREMUM * 4));
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_counts), cap * 4));
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_orient_sums), (cap / 1024 + 2) * 4));
        h->orient_cap = cap;
    }
    if (staging && max_out > h->kps_out_cap) {
        retire_graph(h);
        if (h->d_kps_out) (void)hipFree(h->d_kps_out);
        h->d_kps_out = nullptr;
        h->kps_out_cap = 0;
        if (h->d_det_desc) (void)hipFree(h->d_det_desc);
        h->d_det_desc = nullptr;
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_kps_out), max_out * sizeof(lf_mkd_keypoint)));
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_det_desc), max_out * kOut * 4));   // lf_mkd_detect's staging
        h->kps_out_cap = max_out;
    }
    return LF_MKD_OK;
}

constexpr int kBorder = 5;                  // mod.rs:405
constexpr float kContrastThreshold = 0.035f; // mod.rs:76
constexpr int kSkipLayers = 0;              // mod.rs:398

int ensure_detect_scratch(lf_mkd *h) {
    if (h->d_slots) return LF_MKD_OK;
    int gx, gy, gz;
    scan_grid(int(h->params.max_image_width), int(h->params.max_image_height), h->n_layers - 1, kBorder, kSkipLayers, gx,
              gy, gz);
    const size_t cubes = std::max<size_t>(size_t(gx) * gy * gz, 1) * h->max_frames;
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_slots), cubes * 8 * 4 * sizeof(float)));
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_cube_counts), cubes * 4));
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_cube_sums), ((cubes + 1023) / 1024 + 1) * 4));
    if (!h->d_sel_count) LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_sel_count), 4 * h->max_frames));
    return LF_MKD_OK;
}

// ordered extrema of all loaded frames into d_out (device), counts to the host
int detect_extrema_device(lf_mkd *h, float *d_out, uint32_t *d_frame_of, uint64_t max_out, uint64_t *n_out,
                          uint64_t *n_dropped, hipStream_t s) {
    if (int rc = ensure_coarse_stack(h, s)) return rc;
    if (int rc = ensure_detect_scratch(h)) return rc;
    launch_detect_extrema(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse, h->coarse_stride, h->layer_stride,
                          h->n_layers, h->pd.w[0], h->pd.h[0], int(h->n_frames), kBorder, kSkipLayers, kContrastThreshold,
                          h->d_slots, h->d_cube_counts, h->d_cube_sums, d_out, d_frame_of, nullptr, max_out, h->d_totals,
                          s);
    LF_HIP(h, hipGetLastError());
    unsigned long long totals[2] = {0, 0};
    LF_HIP(h, hipMemcpyAsync(totals, h->d_totals, sizeof(totals), hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipStreamSynchronize(s));
    *n_out = totals[0];
    if (n_dropped) *n_dropped = totals[1];
    return LF_MKD_OK;
}

// recorded: the buffer is one a recorded pipeline names by address (record_pipeline: the detector's extrema and selection, the
// top-n scratch, the staging patches; the orientation scratch, d_kps_out, d_det_desc and h_res_kps have their own sites) --
// moving it retires every recording.  The matcher's and the multi-frame detect's scratch is named by no recording: growing it
// leaves the recordings alone (and does not wait for the device).
enum Recorded { kNotRecorded = 0, kRecorded = 1 };
template <typename T>
int grow(lf_mkd *h, T **p, uint64_t *cap, uint64_t want, size_t elem_bytes, Recorded recorded = kNotRecorded) {
    if (want <= *cap && *p) return LF_MKD_OK;
    if (recorded == kRecorded) retire_graph(h);
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(p), std::max<uint64_t>(want, 1) * elem_bytes));
    *cap = want;
    return LF_MKD_OK;
}

// scratch of the long-list top-n selection; its histograms are zero between uses (the selection's last launch leaves them
// so), which a fresh allocation has to establish once
// (zeroed on the stream the selection will run on, outside any capture: the consumers' streams are non-blocking, i.e. not
// ordered with the null stream)
int grow_topk_work(lf_mkd *h, uint64_t n_cap, hipStream_t s) {
    const uint64_t before = h->topk_work_cap;
    if (int rc = grow(h, &h->d_topk_work, &h->topk_work_cap, topk_work_words(n_cap), 4, kRecorded)) return rc;
    if (h->topk_work_cap != before) LF_HIP(h, hipMemsetAsync(h->d_topk_work, 0, topk_work_words(n_cap) * 4, s));
    return LF_MKD_OK;
}

// the matcher's three words -- [0] largest |b| (float bits), [1] rows of the current chunk whose candidate records
// overflowed, [2] the same over the call (lf_mkd_match_overflowed) -- allocated on first use and zeroed once, on the stream
// the match will run on (so that a form that does not reset [2] adds to a defined value)
int ensure_match_misc(lf_mkd *h, hipStream_t s) {
    if (h->d_match_misc) return LF_MKD_OK;
    if (int rc = grow(h, &h->d_match_misc, &h->match_misc_cap, 3, sizeof(unsigned))) return rc;
    LF_HIP(h, hipMemsetAsync(h->d_match_misc, 0, 3 * sizeof(unsigned), s));
    return LF_MKD_OK;
}

int orient_device(lf_mkd *h, const float *d_extrema, const uint32_t *d_frame_of, uint64_t n, float *d_out,
                  uint32_t *d_frame_of_kp, uint64_t max_out, uint64_t *n_out, uint64_t *n_dropped, hipStream_t s) {
    if (int rc = ensure_coarse_stack(h, s)) return rc;
    launch_orient(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse, h->coarse_stride, h->layer_stride, h->n_layers,
                  h->pd.w[0], h->pd.h[0], d_extrema, d_frame_of, long(n), nullptr, h->d_angles, h->d_counts,
                  h->d_orient_sums, d_out, d_frame_of_kp, max_out, h->d_totals, s);
    LF_HIP(h, hipGetLastError());
    unsigned long long totals[2] = {0, 0};
    LF_HIP(h, hipMemcpyAsync(totals, h->d_totals, sizeof(totals), hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipStreamSynchronize(s));
    *n_out = totals[0];
    if (n_dropped) *n_dropped = totals[1];
    return LF_MKD_OK;
}

// ---- the banded upload's plan (lf_mkd_detect / lf_mkd_detect_u8 on large frames) --------------------------------------
// Stage rows [lo, hi) of piece p of a frame cut at raw rows cuts[0] < cuts[1] < ... < h (the last piece ends at h).
bool piece_bands(const std::vector<uint32_t> &cuts, size_t p, int w, int hgt, int n_layers, RowBands &b) {
    RowBands lo{}, hi{};
    if (p > 0 && !plan_row_bands(int(cuts[p - 1]), w, hgt, n_layers, kBorder, lo)) return false;
    const bool last = p == cuts.size();
    if (!last && !plan_row_bands(int(cuts[p]), w, hgt, n_layers, kBorder, hi)) return false;
    b = RowBands{};
    b.last = last ? 1 : 0;
    b.level0_lo = lo.level0_hi;
    b.level0_hi = last ? hgt : hi.level0_hi;
    for (int l = 0; l < 8; ++l) {
        b.layer_lo[l] = lo.layer_hi[l];
        b.layer_hi[l] = last ? hgt : hi.layer_hi[l];
    }
    b.scan_lo = lo.scan_hi;
    b.scan_hi = last ? (hgt - 2 * kBorder + 7) / 8 : hi.scan_hi;
    return true;
}

// What the pieces cost, from this chip's timelines of the reference benchmark's frame (4096 x 3072; profiles/r06_bands.md):
// picoseconds per pixel of a stage's rows plus a floor per launch; the link's rate for a copy from pageable memory and what
// an extra piece costs the link: ~11 us per copy call (r05_upload_probe.txt) + ~15 us in which the host, blocked in the
// copy until then, records the event and launches the head (LF_MKD_BAND_TRACE prints the calls' host times).
struct FrontModel {
    double sep3 = 2.3, swt = 1.7, swt_deep = 2.3, floor_us = 5.0, link_gb_s = 56.0, gap_us = 26.0;
    double scan(int n_fine) const { return std::max(3.25 * n_fine - 6.85, 2.0); }
};
// Modelled time (us) at which the device is through with the front's share of pieces 0 .. n_pieces-1 of a frame cut at
// `cuts` (n_pieces = cuts.size() + 1: the whole front); *link_us: when the last of these pieces has arrived.
double front_finish_us(const std::vector<uint32_t> &cuts, size_t n_pieces, int w, int hgt, int bpp, int n_layers,
                       const FrontModel &m, double *link_us = nullptr) {
    const double row_us = double(w) * bpp / (m.link_gb_s * 1e3);
    double t_link = 0, t_dev = 0;
    uint32_t from = 0;
    for (size_t p = 0; p < n_pieces; ++p) {
        const uint32_t to = p < cuts.size() ? cuts[p] : uint32_t(hgt);
        t_link += (p ? m.gap_us : 0.0) + double(to - from) * row_us;
        from = to;
        RowBands b;
        if (!piece_bands(cuts, p, w, hgt, n_layers, b)) return 1e30;
        double head = 0;
        auto add = [&](int rows, double ps) {
            if (rows > 0) head += m.floor_us + double(rows) * w * ps * 1e-6;
        };
        add(b.level0_hi - b.level0_lo, m.sep3);
        for (int l = 0; l + 1 < n_layers; ++l) add(b.layer_hi[l] - b.layer_lo[l], (1 << l) > 32 ? m.swt_deep : m.swt);
        add((b.scan_hi - b.scan_lo) * 8, m.scan(n_layers - 1));
        t_dev = std::max(t_dev, t_link) + head;
    }
    if (link_us) *link_us = t_link;
    return t_dev;
}

// Where to cut a frame of `bpp` bytes per pixel.  Candidates: k equal pieces, and plans in which every piece after the first
// is sized so that its upload ends when the device is through with the pieces before it (the link and the device both stay
// busy) -- an 8-bit frame, whose front takes longer than its upload, comes out as a small first piece and growing ones after
// it; an f32 frame, four times the bytes, as a large first piece and shrinking ones, so that little is left to do when the
// last byte lands.  The candidate with the earliest modelled finish wins; no cuts (one piece) unless it is ahead by 4 %.
// LF_MKD_DETECT_BANDS=0: one piece.  LF_MKD_BAND_SPLIT=f1[,f2..]: cuts at these fractions of the height;
// LF_MKD_BAND_PIECES=k: k equal pieces -- both whatever the frame's size (tests, A/B runs).
std::vector<uint32_t> plan_cuts(int n_layers, uint32_t width, uint32_t height, bool u8) {
    const std::vector<uint32_t> none;
    const int w = int(width), hgt = int(height), bpp = u8 ? 1 : 4;
    RowBands probe;
    // (a frame with a pyramid of one level -- fewer than 3 rows or columns -- has nothing to band)
    if (pyramid_levels(width, height) < 2 || height < 64 || !plan_row_bands(int(height) / 2 / 4 * 4, w, hgt, n_layers, kBorder, probe)) return none;
    // cuts on multiples of four rows, at least 16 rows from either end, strictly increasing
    auto valid = [&](std::vector<uint32_t> &c) {
        for (auto &r : c) r = std::min<uint32_t>(std::max<uint32_t>(r, 16), height - 16) / 4 * 4;
        c.erase(std::unique(c.begin(), c.end()), c.end());
        for (size_t i = 0; i < c.size(); ++i)
            if (c[i] >= height || (i && c[i] <= c[i - 1])) return false;
        return !c.empty();
    };
    const char *env_b = getenv("LF_MKD_DETECT_BANDS"), *env_f = getenv("LF_MKD_BAND_SPLIT"), *env_k = getenv("LF_MKD_BAND_PIECES");
    auto equal_pieces = [&](int k) {
        std::vector<uint32_t> c;
        for (int i = 1; i < k; ++i) c.push_back(uint32_t(uint64_t(height) * i / k));
        return c;
    };
    if (env_f) {
        std::vector<uint32_t> c;
        for (const char *q = env_f; *q;) {
            char *end = nullptr;
            const double f = strtod(q, &end);
            if (end == q) break;
            c.push_back(uint32_t(double(height) * (f >= 0.0 ? std::min(f, 1.0) : 0.0)));     // (a NaN counts as 0)
            q = *end == ',' ? end + 1 : end;
        }
        std::sort(c.begin(), c.end());
        return valid(c) ? c : none;
    }
    if (env_k) {
        std::vector<uint32_t> c = equal_pieces(std::min(std::max(atoi(env_k), 1), 16));
        return valid(c) ? c : none;
    }
    if ((env_b && env_b[0] == '0') || uint64_t(width) * height * bpp < 6000000ull) return none;
    const FrontModel m;
    const double row_us = double(w) * bpp / (m.link_gb_s * 1e3);
    std::vector<uint32_t> best;
    double best_t = front_finish_us(none, 1, w, hgt, bpp, n_layers, m) * 0.96;
    auto consider = [&](std::vector<uint32_t> c) {
        if (!valid(c)) return;
        const double t = front_finish_us(c, c.size() + 1, w, hgt, bpp, n_layers, m);
        if (t < best_t) {
            best_t = t;
            best = c;
        }
    };
    for (int k = 2; k <= 6; ++k) consider(equal_pieces(k));
    for (int i = 1; i <= 14; ++i) {             // first piece = i / 16 of the frame, the others by the recurrence
        std::vector<uint32_t> c{uint32_t(uint64_t(height) * i / 16)};
        while (c.size() < 6 && valid(c)) {
            consider(c);
            // the next piece: as many rows as the link delivers while the device works off the pieces so far
            double t_link = 0;
            const double t_dev = front_finish_us(c, c.size(), w, hgt, bpp, n_layers, m, &t_link);
            const uint32_t rows = uint32_t(std::max((t_dev - t_link - m.gap_us) / row_us, double(height) / 16));
            if (c.back() + rows + height / 16 >= height) break;
            c.push_back(c.back() + rows);
        }
    }
    return best;
}

}  // namespace

int lf_mkd_internal_device(const lf_mkd *h) { return h->params.device; }
hipStream_t lf_mkd_internal_stream(lf_mkd *h) { return h->stream; }
int lf_mkd_internal_fail(lf_mkd *h, int code, const std::string &msg) { return fail(h, code, msg); }

extern "C" {

const char *lf_mkd_version(void) { return "lf_mkd 0.1.0 gfx950"; }

int lf_mkd_create(const lf_mkd_params *params, const float *mean, const float *eigvals, const float *eigvecs,
                  lf_mkd **out) {
    if (!mean || !eigvals || !eigvecs) return LF_MKD_ERR_BAD_ARG;
    PcaModel pca;
    pca.mean.assign(mean, mean + kRaw);
    pca.eigvals.assign(eigvals, eigvals + kRaw);
    pca.eigvecs.assign(eigvecs, eigvecs + size_t(kRaw) * kRaw);
    return create_impl(params, pca, out);
}

int lf_mkd_build_constants(const float *mean, const float *eigvals, const float *eigvecs, float *gradient_angle,
                           float *embedding_polar, float *embedding_cartesian, float *w_t) {
    if (!mean || !eigvals || !eigvecs) return LF_MKD_ERR_BAD_ARG;
    PcaModel pca;
    pca.mean.assign(mean, mean + kRaw);
    pca.eigvals.assign(eigvals, eigvals + kRaw);
    pca.eigvecs.assign(eigvecs, eigvecs + size_t(kRaw) * kRaw);
    HostConsts hc;
    if (const std::string e = build_host_consts(pca, hc); !e.empty()) {
        g_create_error = e;
        return LF_MKD_ERR_BAD_ARG;
    }
    if (gradient_angle) std::memcpy(gradient_angle, hc.gradient_angle.data(), hc.gradient_angle.size() * 4);
    if (embedding_polar) std::memcpy(embedding_polar, hc.embedding_polar.data(), hc.embedding_polar.size() * 4);
    if (embedding_cartesian)
        std::memcpy(embedding_cartesian, hc.embedding_cartesian.data(), hc.embedding_cartesian.size() * 4);
    if (w_t) std::memcpy(w_t, hc.w_t.data(), hc.w_t.size() * 4);
    return LF_MKD_OK;
}

int lf_mkd_plan_upload(uint32_t width, uint32_t height, uint32_t bytes_per_pixel, uint32_t n_scales, uint32_t *cuts,
                       uint32_t max_cuts, uint32_t *n_cuts, double *modelled_us, double *one_piece_us) {
    if (!n_cuts || (bytes_per_pixel != 1 && bytes_per_pixel != 4) || width < 2 || height < 2 || (max_cuts && !cuts))
        return LF_MKD_ERR_BAD_ARG;
    const int n_layers = int(n_scales ? n_scales : 4) + 3;
    if (n_layers - 1 > 8) return LF_MKD_ERR_BAD_ARG;
    const std::vector<uint32_t> c = plan_cuts(n_layers, width, height, bytes_per_pixel == 1);
    *n_cuts = uint32_t(c.size());
    for (size_t i = 0; i < c.size() && i < max_cuts; ++i) cuts[i] = c[i];
    const FrontModel m;
    if (modelled_us) *modelled_us = front_finish_us(c, c.size() + 1, int(width), int(height), int(bytes_per_pixel), n_layers, m);
    if (one_piece_us) *one_piece_us = front_finish_us({}, 1, int(width), int(height), int(bytes_per_pixel), n_layers, m);
    return LF_MKD_OK;
}

int lf_mkd_create_from_file(const lf_mkd_params *params, const char *path, lf_mkd **out) {
    if (!path) return LF_MKD_ERR_BAD_ARG;
    PcaModel pca;
    const std::string e = load_pca_safetensors(path, pca);
    if (!e.empty()) {
        g_create_error = e;
        return LF_MKD_ERR_IO;
    }
    return create_impl(params, pca, out);
}

void lf_mkd_destroy(lf_mkd *h) {
    if (!h) return;
    lfmkd::DeviceScope scope_;
    (void)scope_.enter(h->params.device);
    (void)hipDeviceSynchronize();   // work of this handle may be in flight on the caller's streams too
    void *ptrs[] = {h->dc.colmap,     h->dc.pool_b_f32, h->dc.pool_b_f16, h->dc.pool_b_fp6, h->dc.white_a_f16,
                    h->dc.white_a_f32, h->dc.white_bias, h->d_patches,     h->d_out,         h->d_kps,
                    h->d_image,        h->d_pyr,         h->d_tmp_a,       h->d_tmp_b,       h->d_coarse,
                    h->d_extrema,      h->d_angles,      h->d_counts,      h->d_kps_out,     h->d_totals,      h->d_clk,
                    h->d_slots,        h->d_det_extrema, h->d_det_selected, h->d_det_desc,
                    h->d_cube_counts,  h->d_cube_sums,   h->d_sel_count,   h->d_match_a,     h->d_match_b,
                    h->d_orient_sums,  h->d_topk_work,
                    h->d_match_part,   h->d_match_in,    h->d_match_out,   h->d_mf_padded,   h->d_mf_list,
                    h->d_mf_frame_start, h->d_mf_offsets, h->d_mf_frame_of,
                    h->d_match_rec,    h->d_match_cnt,   h->d_match_norm,  h->d_match_misc,
                    h->d_match_few_tiles, h->d_match_few, h->d_kp_xchg, h->d_kp_words, h->d_match_floor};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    for (auto &p : h->plans) destroy_plan(p);
    for (hipEvent_t e : h->copy_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->d_image_u8) (void)hipFree(h->d_image_u8);
    for (hipEvent_t e : h->det_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->d_det_counts) (void)hipFree(h->d_det_counts);
    if (h->h_det_counts) (void)hipHostFree(h->h_det_counts);
    if (h->h_res_kps) (void)hipHostFree(h->h_res_kps);
    if (h->d_stream_patches) (void)hipFree(h->d_stream_patches);
    for (hipEvent_t e : h->ev_pending) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_free) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->side_events) (void)hipEventDestroy(e);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *lf_mkd_last_error(const lf_mkd *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int lf_mkd_kernel_times(lf_mkd *h, double *pool_ms, double *whiten_ms, uint64_t *launches) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    LF_ENTER(h);
    double tp = 0, tw = 0;
    const size_t nb = h->ev_pending.size() / 2;
    for (size_t i = 0; i < nb; ++i) {
        float a = 0;
        LF_HIP(h, hipEventSynchronize(h->ev_pending[2 * i + 1]));
        LF_HIP(h, hipEventElapsedTime(&a, h->ev_pending[2 * i], h->ev_pending[2 * i + 1]));
        tp += a;
    }
    h->ev_free.insert(h->ev_free.end(), h->ev_pending.begin(), h->ev_pending.end());
    h->ev_pending.clear();
    if (pool_ms) *pool_ms = tp;
    if (whiten_ms) *whiten_ms = tw;
    if (launches) *launches = nb;
    return LF_MKD_OK;
}

int lf_mkd_kernel_clock(lf_mkd *h, void *stream, double *shader_mhz, double *kernel_ms) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (shader_mhz) *shader_mhz = 0;
    if (kernel_ms) *kernel_ms = 0;
    if (!h->d_clk) return fail(h, LF_MKD_ERR_BAD_ARG, "kernel_clock: the handle was not created with LF_MKD_FLAG_KERNEL_TIMING");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    unsigned long long c[4] = {0, 0, 0, 0};
    LF_HIP(h, hipMemcpyAsync(c, h->d_clk, sizeof(c), hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipStreamSynchronize(s));
    if (c[3] > c[1] && c[2] > c[0]) {
        const double ticks = double(c[3] - c[1]);            // s_memrealtime: 100 MHz
        if (shader_mhz) *shader_mhz = double(c[2] - c[0]) / ticks * 100.0;
        if (kernel_ms) *kernel_ms = ticks / 1e5;
    }
    return LF_MKD_OK;
}

int lf_mkd_synchronize(lf_mkd *h) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    LF_ENTER(h);
    LF_HIP(h, hipStreamSynchronize(h->stream));
    return LF_MKD_OK;
}

int lf_mkd_describe_patches_device(lf_mkd *h, const float *d_patches, uint64_t n, float *d_out, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (n == 0) return LF_MKD_OK;
    if (!d_patches || !d_out) return fail(h, LF_MKD_ERR_BAD_ARG, "describe_patches_device: null pointer");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    for (uint64_t off = 0; off < n; off += h->batch) {
        const uint64_t m = std::min<uint64_t>(h->batch, n - off);
        const int rc = run_batch(h, d_patches + off * kPx, m, d_out + off * kOut, nullptr, s);
        if (rc) return rc;
    }
    return LF_MKD_OK;
}

int lf_mkd_raw_descriptors_device(lf_mkd *h, const float *d_patches, uint64_t n, float *d_raw, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (n == 0) return LF_MKD_OK;
    if (!d_patches || !d_raw) return fail(h, LF_MKD_ERR_BAD_ARG, "raw_descriptors_device: null pointer");
    LF_ENTER(h);
    if (int rc = ensure_staging(h)) return rc;   // the kernel also writes the whitened descriptors: into the staging buffer
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    for (uint64_t off = 0; off < n; off += h->batch) {
        const uint64_t m = std::min<uint64_t>(h->batch, n - off);
        const int rc = run_batch(h, d_patches + off * kPx, m, nullptr, d_raw + off * kRaw, s);
        if (rc) return rc;
    }
    return LF_MKD_OK;
}

int lf_mkd_describe_patches(lf_mkd *h, const float *patches, uint64_t n, float *out) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (n == 0) return LF_MKD_OK;
    if (!patches || !out) return fail(h, LF_MKD_ERR_BAD_ARG, "describe_patches: null pointer");
    LF_ENTER(h);
    if (int rc = ensure_staging(h)) return rc;
    for (uint64_t off = 0; off < n; off += h->batch) {
        const uint64_t m = std::min<uint64_t>(h->batch, n - off);
        LF_HIP(h, hipMemcpyAsync(h->d_patches, patches + off * kPx, m * kPx * 4, hipMemcpyHostToDevice, h->stream));
        const int rc = run_batch(h, h->d_patches, m, h->d_out, nullptr, h->stream);
        if (rc) return rc;
        LF_HIP(h, hipMemcpyAsync(out + off * kOut, h->d_out, m * kOut * 4, hipMemcpyDeviceToHost, h->stream));
        LF_HIP(h, hipStreamSynchronize(h->stream));
    }
    return LF_MKD_OK;
}

// lf_mkd_set_images_device and its 8-bit twin: exactly one of d_images / d_images_u8 is given
static int set_images_impl(lf_mkd *h, const float *d_images, const unsigned char *d_images_u8, uint32_t n_frames,
                           uint32_t width, uint32_t height, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if ((!d_images && !d_images_u8) || width < 2 || height < 2 || n_frames == 0)
        return fail(h, LF_MKD_ERR_BAD_ARG, "set_image: bad image");
    if (!h->d_pyr || width > h->params.max_image_width || height > h->params.max_image_height)
        return fail(h, LF_MKD_ERR_BAD_ARG,
                    "set_image: image " + std::to_string(width) + "x" + std::to_string(height) +
                        " exceeds max_image_width/height given at creation");
    if (n_frames > h->max_frames)
        return fail(h, LF_MKD_ERR_BAD_ARG, "set_images: " + std::to_string(n_frames) + " frames exceed max_frames = " +
                                               std::to_string(h->max_frames) + " given at creation");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    describe_pyramid(width, height, h->pd);
    // once the a-trous stack exists (orientation or the detector have been used on this handle) the pyramid's
    // a-trous layer 1 goes straight into it
    const bool share = h->d_coarse != nullptr && h->pd.levels >= 2;
    launch_build_pyramid(d_images, long(width) * height, h->d_pyr, h->pyr_stride, h->d_tmp_a, h->d_tmp_b, h->pd,
                         int(n_frames), share ? h->d_coarse : nullptr, h->coarse_stride, s, nullptr, nullptr, nullptr, {},
                         d_images_u8);
    h->coarse_l1_valid = share;
    LF_HIP(h, hipGetLastError());
    h->have_image = true;
    h->coarse_valid = false;
    h->n_frames = n_frames;
    return LF_MKD_OK;
}

int lf_mkd_set_images_device(lf_mkd *h, const float *d_images, uint32_t n_frames, uint32_t width, uint32_t height,
                             void *stream) {
    if (h && !d_images) return fail(h, LF_MKD_ERR_BAD_ARG, "set_image: bad image");
    return set_images_impl(h, d_images, nullptr, n_frames, width, height, stream);
}

int lf_mkd_set_images_u8_device(lf_mkd *h, const uint8_t *d_images, uint32_t n_frames, uint32_t width, uint32_t height,
                                void *stream) {
    if (h && !d_images) return fail(h, LF_MKD_ERR_BAD_ARG, "set_image_u8: bad image");
    return set_images_impl(h, nullptr, d_images, n_frames, width, height, stream);
}

int lf_mkd_set_image_device(lf_mkd *h, const float *d_image, uint32_t width, uint32_t height, void *stream) {
    return lf_mkd_set_images_device(h, d_image, 1, width, height, stream);
}

// the 8-bit staging frame (1 B/px; lf_mkd_set_image_u8, lf_mkd_detect_u8), allocated on first use
static int ensure_u8_staging(lf_mkd *h) {
    if (h->d_image_u8) return LF_MKD_OK;
    // (a recorded pipeline reads it by address: allocated once at the maximum frame size, it never moves)
    LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_image_u8),
                        (size_t(h->params.max_image_width) * h->params.max_image_height + 3) / 4 * 4));
    return LF_MKD_OK;
}

int lf_mkd_set_image(lf_mkd *h, const float *image, uint32_t width, uint32_t height) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!image) return fail(h, LF_MKD_ERR_BAD_ARG, "set_image: null image");
    if (!h->d_image || width > h->params.max_image_width || height > h->params.max_image_height)
        return fail(h, LF_MKD_ERR_BAD_ARG, "set_image: image exceeds max_image_width/height given at creation");
    LF_ENTER(h);
    LF_HIP(h, hipMemcpyAsync(h->d_image, image, size_t(width) * height * 4, hipMemcpyHostToDevice, h->stream));
    const int rc = lf_mkd_set_image_device(h, h->d_image, width, height, h->stream);
    if (rc) return rc;
    LF_HIP(h, hipStreamSynchronize(h->stream));
    return LF_MKD_OK;
}

int lf_mkd_set_image_u8(lf_mkd *h, const uint8_t *image, uint32_t width, uint32_t height) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!image) return fail(h, LF_MKD_ERR_BAD_ARG, "set_image_u8: null image");
    if (!h->d_image || width > h->params.max_image_width || height > h->params.max_image_height)
        return fail(h, LF_MKD_ERR_BAD_ARG, "set_image_u8: image exceeds max_image_width/height given at creation");
    LF_ENTER(h);
    if (int rc = ensure_u8_staging(h)) return rc;
    LF_HIP(h, hipMemcpyAsync(h->d_image_u8, image, size_t(width) * height, hipMemcpyHostToDevice, h->stream));
    const int rc = lf_mkd_set_images_u8_device(h, h->d_image_u8, 1, width, height, h->stream);
    if (rc) return rc;
    LF_HIP(h, hipStreamSynchronize(h->stream));
    return LF_MKD_OK;
}

int lf_mkd_sample_patches_device(lf_mkd *h, const lf_mkd_keypoint *d_kps, uint64_t n, float *d_patches,
                                 void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "sample_patches: call lf_mkd_set_image first");
    if (n == 0) return LF_MKD_OK;
    if (!d_kps || !d_patches) return fail(h, LF_MKD_ERR_BAD_ARG, "sample_patches: null pointer");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    launch_sample_patches(h->d_pyr, h->pyr_stride, h->pd, reinterpret_cast<const float *>(d_kps), nullptr, 1, long(n),
                          nullptr, h->params.patch_scale_factor, d_patches, s);
    LF_HIP(h, hipGetLastError());
    return LF_MKD_OK;
}

int lf_mkd_describe_keypoints_frames_device(lf_mkd *h, const lf_mkd_keypoint *d_kps, const uint32_t *d_frame_of_kp,
                                            uint64_t n, float *d_out, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "describe_keypoints: call lf_mkd_set_image first");
    if (n == 0) return LF_MKD_OK;
    if (!d_kps || !d_out) return fail(h, LF_MKD_ERR_BAD_ARG, "describe_keypoints_device: null pointer");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    return describe_keypoints_on_device(h, reinterpret_cast<const float *>(d_kps), d_frame_of_kp, n, d_out, s);
}

int lf_mkd_describe_keypoints_device(lf_mkd *h, const lf_mkd_keypoint *d_kps, uint64_t n, float *d_out,
                                     void *stream) {
    return lf_mkd_describe_keypoints_frames_device(h, d_kps, nullptr, n, d_out, stream);
}

int lf_mkd_describe_keypoints(lf_mkd *h, const lf_mkd_keypoint *kps, uint64_t n, float *out) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "describe_keypoints: call lf_mkd_set_image first");
    if (n == 0) return LF_MKD_OK;
    if (!kps || !out) return fail(h, LF_MKD_ERR_BAD_ARG, "describe_keypoints: null pointer");
    LF_ENTER(h);
    if (int rc = ensure_io_staging(h)) return rc;
    for (uint64_t off = 0; off < n; off += h->batch) {
        const uint64_t m = std::min<uint64_t>(h->batch, n - off);
        LF_HIP(h, hipMemcpyAsync(h->d_kps, kps + off, m * sizeof(lf_mkd_keypoint), hipMemcpyHostToDevice, h->stream));
        if (int rc = describe_keypoints_on_device(h, h->d_kps, nullptr, m, h->d_out, h->stream)) return rc;
        LF_HIP(h, hipMemcpyAsync(out + off * kOut, h->d_out, m * kOut * 4, hipMemcpyDeviceToHost, h->stream));
        LF_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (h->d_kp_words) {   // the row-split form counts partial sums that never reached their consumer (none, ever: a fault if so)
        unsigned lost = 0;
        LF_HIP(h, hipMemcpy(&lost, h->d_kp_words, 4, hipMemcpyDeviceToHost));
        if (lost) {
            (void)hipMemset(h->d_kp_words, 0, kp_split_counter_words(h->num_cus) * 4);
            return fail(h, LF_MKD_ERR_HIP, "describe_keypoints: " + std::to_string(lost) + " partial sums of the row-split form did not arrive");
        }
    }
    return LF_MKD_OK;
}

int lf_mkd_orient_keypoints_device(lf_mkd *h, const lf_mkd_extremum *d_extrema, const uint32_t *d_frame_of_extremum,
                                   uint64_t n, lf_mkd_keypoint *d_out, uint32_t *d_frame_of_kp, uint64_t max_out,
                                   uint64_t *n_out, uint64_t *n_dropped, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "orient_keypoints_device: n_out is null");
    *n_out = 0;
    if (n_dropped) *n_dropped = 0;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "orient_keypoints: call lf_mkd_set_image first");
    if (n == 0) return LF_MKD_OK;
    if (!d_extrema || (!d_out && max_out)) return fail(h, LF_MKD_ERR_BAD_ARG, "orient_keypoints_device: null pointer");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (int rc = ensure_orient_scratch(h, n, false, 0)) return rc;
    return orient_device(h, reinterpret_cast<const float *>(d_extrema), d_frame_of_extremum, n,
                         reinterpret_cast<float *>(d_out), d_frame_of_kp, max_out, n_out, n_dropped, s);
}

int lf_mkd_orient_keypoints(lf_mkd *h, const lf_mkd_extremum *extrema, uint64_t n, lf_mkd_keypoint *out,
                            uint64_t max_out, uint64_t *n_out, uint64_t *n_dropped) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "orient_keypoints: n_out is null");
    *n_out = 0;
    if (n_dropped) *n_dropped = 0;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "orient_keypoints: call lf_mkd_set_image first");
    if (n == 0) return LF_MKD_OK;
    if (!extrema || (!out && max_out)) return fail(h, LF_MKD_ERR_BAD_ARG, "orient_keypoints: null pointer");
    LF_ENTER(h);
    if (int rc = ensure_orient_scratch(h, n, true, std::max<uint64_t>(max_out, 1))) return rc;
    LF_HIP(h, hipMemcpyAsync(h->d_extrema, extrema, n * sizeof(lf_mkd_extremum), hipMemcpyHostToDevice, h->stream));
    if (int rc = orient_device(h, h->d_extrema, nullptr, n, h->d_kps_out, nullptr, max_out, n_out, n_dropped, h->stream))
        return rc;
    if (*n_out) LF_HIP(h, hipMemcpy(out, h->d_kps_out, *n_out * sizeof(lf_mkd_keypoint), hipMemcpyDeviceToHost));
    return LF_MKD_OK;
}

int lf_mkd_detect_extrema_device(lf_mkd *h, lf_mkd_extremum *d_out, uint32_t *d_frame_of, uint64_t max_out,
                                 uint64_t *n_out, uint64_t *n_dropped, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "detect_extrema_device: n_out is null");
    *n_out = 0;
    if (n_dropped) *n_dropped = 0;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "detect_extrema: call lf_mkd_set_image first");
    if (!d_out && max_out) return fail(h, LF_MKD_ERR_BAD_ARG, "detect_extrema_device: null pointer");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    return detect_extrema_device(h, reinterpret_cast<float *>(d_out), d_frame_of, max_out, n_out, n_dropped, s);
}

int lf_mkd_detect_extrema(lf_mkd *h, lf_mkd_extremum *out, uint64_t max_out, uint64_t *n_out, uint64_t *n_dropped) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "detect_extrema: n_out is null");
    *n_out = 0;
    if (n_dropped) *n_dropped = 0;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "detect_extrema: call lf_mkd_set_image first");
    if (!out && max_out) return fail(h, LF_MKD_ERR_BAD_ARG, "detect_extrema: null pointer");
    LF_ENTER(h);
    if (int rc = grow(h, &h->d_det_extrema, &h->det_out_cap, max_out, sizeof(lf_mkd_extremum), kRecorded)) return rc;
    if (int rc = detect_extrema_device(h, h->d_det_extrema, nullptr, max_out, n_out, n_dropped, h->stream)) return rc;
    if (*n_out) LF_HIP(h, hipMemcpy(out, h->d_det_extrema, *n_out * sizeof(lf_mkd_extremum), hipMemcpyDeviceToHost));
    return LF_MKD_OK;
}

int lf_mkd_filter_extrema_device(lf_mkd *h, const lf_mkd_extremum *d_extrema, uint64_t n, uint32_t top_n, float min_size,
                                 lf_mkd_extremum *d_out, uint32_t *d_index, uint64_t *n_out, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "filter_extrema_device: n_out is null");
    *n_out = 0;
    if (n == 0 || top_n == 0) return LF_MKD_OK;
    if (!d_extrema || !d_out) return fail(h, LF_MKD_ERR_BAD_ARG, "filter_extrema_device: null pointer");
    if (n > 0xFFFFFFFFull) return fail(h, LF_MKD_ERR_BAD_ARG, "filter_extrema_device: more than 2^32 extrema");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (!h->d_sel_count) LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_sel_count), 4 * h->max_frames));
    if (int rc = grow_topk_work(h, n, s)) return rc;
    launch_topk_filter(reinterpret_cast<const float *>(d_extrema), nullptr, nullptr, n, 1, 0xFFFFFFFFu, top_n, min_size,
                       reinterpret_cast<float *>(d_out), d_index, h->d_sel_count, nullptr, n, h->d_topk_work, s);
    LF_HIP(h, hipGetLastError());
    unsigned cnt = 0;
    LF_HIP(h, hipMemcpyAsync(&cnt, h->d_sel_count, 4, hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipStreamSynchronize(s));
    *n_out = cnt;
    return LF_MKD_OK;
}

// lf_mkd_detect stage by stage, every count fetched by the host before the next stage is sized (three waits on the stream):
// how the call worked before round 5, kept as the verification form (LF_MKD_FLAG_DETECT_STEPWISE) and for max_out == 0.
// The frame is already on the device (d_image or d_image_u8).
static int detect_stepwise(lf_mkd *h, bool u8, uint32_t width, uint32_t height, uint32_t top_n, float min_size,
                           lf_mkd_keypoint *keypoints, float *descriptors, uint64_t max_out, uint64_t *n_out,
                           uint64_t *dropped_blobs, uint64_t *dropped_features) {
    hipStream_t s = h->stream;
    // lf_mkd_set_image without its synchronisation, and -- once the a-trous stack exists, i.e. from the second call on --
    // with pyramid levels >= 1 (read by the sampler only) built on the side stream beside the a-trous passes and the scan
    describe_pyramid(width, height, h->pd);
    const bool share = h->d_coarse != nullptr && h->pd.levels >= 2;
    if (share)
        if (int rc = ensure_side_stream(h, 2)) return rc;
    // (with the branch, the rest of the a-trous stack is queued from inside, ahead of it: see launch_build_pyramid)
    bool stack_queued = false;
    std::function<void()> stack_first;
    if (share)
        stack_first = [&] {
            launch_build_coarse_stack(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse, h->coarse_stride,
                                      h->layer_stride, h->d_tmp_a, h->n_layers, 1, int(width), int(height), 1, s);
            stack_queued = true;
        };
    launch_build_pyramid(h->d_image, long(width) * height, h->d_pyr, h->pyr_stride, h->d_tmp_a, h->d_tmp_b, h->pd, 1,
                         share ? h->d_coarse : nullptr, h->coarse_stride, s, share ? h->side_stream : nullptr,
                         share ? h->side_events[0] : nullptr, share ? h->side_events[1] : nullptr, stack_first,
                         u8 ? h->d_image_u8 : nullptr);
    LF_HIP(h, hipGetLastError());
    h->coarse_l1_valid = share;
    h->have_image = true;
    h->coarse_valid = stack_queued;
    h->n_frames = 1;
    // detect graph: extrema, at most max_extrema of them (mod.rs:625-633)
    if (int rc = grow(h, &h->d_det_extrema, &h->det_out_cap, h->max_extrema, sizeof(lf_mkd_extremum), kRecorded)) return rc;
    uint64_t n_ext = 0;
    const int rc_ext = detect_extrema_device(h, h->d_det_extrema, nullptr, h->max_extrema, &n_ext, dropped_blobs, s);
    if (share) LF_HIP(h, hipStreamWaitEvent(s, h->side_events[1], 0));   // whatever follows on s sees the whole pyramid
    if (rc_ext) return rc_ext;
    // host blob filter of detect_top_n, on the device
    const float *d_sel = h->d_det_extrema;
    if (top_n && n_ext) {
        if (int rc = grow(h, &h->d_det_selected, &h->det_sel_cap, top_n, sizeof(lf_mkd_extremum), kRecorded)) return rc;
        uint64_t n_sel = 0;
        if (int rc = lf_mkd_filter_extrema_device(h, reinterpret_cast<const lf_mkd_extremum *>(h->d_det_extrema), n_ext,
                                                  top_n, min_size, reinterpret_cast<lf_mkd_extremum *>(h->d_det_selected),
                                                  nullptr, &n_sel, s))
            return rc;
        d_sel = h->d_det_selected;
        n_ext = n_sel;
    }
    if (n_ext == 0 || max_out == 0) return LF_MKD_OK;
    // extract graph: orientation, sampling, description
    if (int rc = ensure_orient_scratch(h, n_ext, true, max_out)) return rc;
    uint64_t n_kp = 0;
    if (int rc = orient_device(h, d_sel, nullptr, n_ext, h->d_kps_out, nullptr, max_out, &n_kp, dropped_features, s))
        return rc;
    if (n_kp == 0) return LF_MKD_OK;
    static_assert(sizeof(lf_mkd_keypoint) == 20, "keypoint layout");
    // (in the form the recorded pipeline takes for this request's max_out: the same bits)
    if (int rc = describe_keypoints_on_device(h, h->d_kps_out, nullptr, n_kp, h->d_det_desc, s, max_out)) return rc;
    LF_HIP(h, hipMemcpyAsync(keypoints, h->d_kps_out, n_kp * sizeof(lf_mkd_keypoint), hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipMemcpyAsync(descriptors, h->d_det_desc, n_kp * kOut * 4, hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipStreamSynchronize(s));
    *n_out = n_kp;
    return LF_MKD_OK;
}


// Everything a recorded pipeline for frames of width x height touches, allocated BEFORE the capture starts (an allocation
// inside a capture is an error, and one that moves a buffer retires every recording: retire_graph).
static int prepare_pipeline(lf_mkd *h, uint32_t top_n, uint64_t cap) {
    if (!h->d_coarse) {
        h->layer_stride = long(h->params.max_image_width) * h->params.max_image_height;
        h->coarse_stride = h->layer_stride * (h->n_layers - 1);
        LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_coarse), size_t(h->coarse_stride) * h->max_frames * 4));
    }
    if (int rc = ensure_detect_scratch(h)) return rc;
    if (int rc = grow(h, &h->d_det_extrema, &h->det_out_cap, h->max_extrema, sizeof(lf_mkd_extremum), kRecorded)) return rc;
    if (top_n) {
        if (int rc = grow(h, &h->d_det_selected, &h->det_sel_cap, top_n, sizeof(lf_mkd_extremum), kRecorded)) return rc;
        if (int rc = grow_topk_work(h, h->max_extrema, h->stream)) return rc;
    }
    return ensure_orient_scratch(h, cap, false, 0);
}

// The launch sequence of lf_mkd_detect for frames of width x height read from d_image (f32) or d_image_u8 -- pyramid,
// a-trous stack, extremum scan, [top_n filter if top_n > 0], orientation, sampling + description -- with every count handed
// from stage to stage in device memory (cnt [8]: see lf_mkd_stream_create in lf_mkd.h).  host_counts (nullable, pinned): a
// last operation copies cnt there.  h->pd must describe the frame; every buffer must exist (prepare_pipeline), and the side
// stream when the frame has levels to build beside the detector.  Enqueued on the handle's stream: inside a capture
// (record_pipeline) or as it is (the first sighting of a request: the same launches, no recording, one wait).
static void enqueue_pipeline(lf_mkd *h, uint32_t width, uint32_t height, uint32_t top_n, float min_size, uint64_t max_out,
                             const float *d_image, const unsigned char *d_image_u8, lf_mkd_keypoint *d_keypoints,
                             float *d_descriptors, unsigned long long *cnt, unsigned long long *host_counts,
                             lf_mkd_keypoint *host_keypoints, const RowBands *bands) {
    hipStream_t s = h->stream;
    const uint64_t cap = top_n ? top_n : h->max_extrema;   // extrema that can reach orientation
    // the detector needs pyramid level 0 and a-trous layer 1 only: the other levels (read by the sampler at the very end)
    // are a branch beside the a-trous stack, the scan, the selection and the orientation
    const bool head_only = bands && !bands->last;           // (a piece's share: the front's kernels on its rows, nothing else)
    const bool fork = h->pd.levels >= 2 && !head_only;
    // (the a-trous stack is queued from inside, ahead of the branch: see launch_build_pyramid)
    launch_build_pyramid(d_image, long(width) * height, h->d_pyr, h->pyr_stride, h->d_tmp_a, h->d_tmp_b, h->pd, 1,
                         h->pd.levels >= 2 ? h->d_coarse : nullptr, h->coarse_stride, s, fork ? h->side_stream : nullptr,
                         fork ? h->side_events[0] : nullptr, fork ? h->side_events[1] : nullptr, [&] {
                             launch_build_coarse_stack(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse,
                                                       h->coarse_stride, h->layer_stride, h->d_tmp_a, h->n_layers,
                                                       h->pd.levels >= 2 ? 1 : 0, int(width), int(height), 1, s, bands);
                         }, d_image_u8, bands);
    launch_detect_extrema(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse, h->coarse_stride, h->layer_stride,
                          h->n_layers, int(width), int(height), 1, kBorder, kSkipLayers, kContrastThreshold, h->d_slots,
                          h->d_cube_counts, h->d_cube_sums, h->d_det_extrema, nullptr, nullptr, h->max_extrema, cnt + 0, s, bands);
    if (head_only) return;
    const float *d_sel = h->d_det_extrema;
    const unsigned long long *n_sel = cnt + 0;
    if (top_n) {
        launch_topk_filter(h->d_det_extrema, nullptr, cnt + 0, 0, 1, 0xFFFFFFFFu, top_n, min_size, h->d_det_selected, nullptr,
                           h->d_sel_count, cnt + 2, h->max_extrema, h->d_topk_work, s);
        d_sel = h->d_det_selected;
        n_sel = cnt + 2;
    }
    launch_orient(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse, h->coarse_stride, h->layer_stride, h->n_layers,
                  int(width), int(height), d_sel, nullptr, long(cap), n_sel, h->d_angles, h->d_counts, h->d_orient_sums,
                  reinterpret_cast<float *>(d_keypoints), nullptr, max_out, cnt + 3, s);
    // (the join costs ~12 us of queue latency wherever it stands, measured; the branch saves ~40)
    if (fork) (void)hipStreamWaitEvent(s, h->side_events[1], 0);
    if (fused_keypoints(h)) {
        launch_describe_keypoints(h->d_pyr, h->pyr_stride, h->pd, reinterpret_cast<const float *>(d_keypoints), nullptr, 1,
                                  long(max_out), cnt + 3, h->params.patch_scale_factor, h->dc, h->params.angle_mode,
                                  d_descriptors, h->num_cus, s, nullptr, h->d_kp_xchg, h->d_kp_words);
    } else {
        launch_sample_patches(h->d_pyr, h->pyr_stride, h->pd, reinterpret_cast<const float *>(d_keypoints), nullptr, 1,
                              long(max_out), cnt + 3, h->params.patch_scale_factor, h->d_stream_patches, s);
        launch_describe(h->d_stream_patches, long(max_out), cnt + 3, h->dc, h->params.angle_mode, h->params.pool_mode,
                        d_descriptors, nullptr, h->num_cus, s);
    }
    // the counts and the keypoints go to pinned host memory as the last operations (on a branch of their own beside the describe
    // launch they were measured 15 us slower: every join of two branches costs ~12 us of queue latency)
    if (host_counts) (void)hipMemcpyAsync(host_counts, cnt, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (host_keypoints)
        (void)hipMemcpyAsync(host_keypoints, d_keypoints, max_out * sizeof(lf_mkd_keypoint), hipMemcpyDeviceToHost, s);
}

// Records that sequence as a hipGraph (the whole pipeline, or one piece's share of its front: bands).
static int record_pipeline(lf_mkd *h, uint32_t width, uint32_t height, uint32_t top_n, float min_size, uint64_t max_out,
                           const float *d_image, const unsigned char *d_image_u8, lf_mkd_keypoint *d_keypoints,
                           float *d_descriptors, unsigned long long *cnt, unsigned long long *host_counts,
                           lf_mkd_keypoint *host_keypoints, hipGraph_t *graph_out, hipGraphExec_t *exec_out,
                           const RowBands *bands = nullptr) {
    hipStream_t s = h->stream;
    if (h->pd.levels >= 2 && !(bands && !bands->last))
        if (int rc = ensure_side_stream(h, 2)) return rc;
    LF_HIP(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue_pipeline(h, width, height, top_n, min_size, max_out, d_image, d_image_u8, d_keypoints, d_descriptors, cnt, host_counts,
                     host_keypoints, bands);
    hipGraph_t graph = nullptr;
    hipError_t e_end = hipStreamEndCapture(s, &graph);
    if (e_end != hipSuccess || !graph) {
        h->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e_end);
        return LF_MKD_ERR_HIP;
    }
    hipError_t e_inst = hipGraphInstantiate(exec_out, graph, nullptr, nullptr, 0);
    if (e_inst != hipSuccess) {
        (void)hipGraphDestroy(graph);
        h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e_inst);
        return LF_MKD_ERR_HIP;
    }
    *graph_out = graph;
    return LF_MKD_OK;
}

// LocalFeaturesVulkan::detect / detect_top_n (mod.rs:346-593) from a HOST frame, f32 or 8-bit.  A request -- (frame size, top_n,
// min_size, max_out, pixel type) -- seen for the first time is served by the pipeline's own launches, not recorded (one
// upload, ~20 launches, one wait: no capture and no instantiation, which cost several times the call); the second time the
// pipeline is recorded (lf_mkd::Sighting), and from then on a call is one upload (in planned pieces when the frame is large:
// plan_cuts), one launch of the recording, one wait, the result copies.  The reference's callers make this call per image
// (examples/match_images/src/main.rs:44-76: once per image, at its own size -- never a recording) or per camera frame
// (examples/webcam/src/main.rs:136-160).
constexpr size_t kMaxPlans = 8;

static int detect_host(lf_mkd *h, const float *image, const unsigned char *image_u8, uint32_t width, uint32_t height,
                       uint32_t top_n, float min_size, lf_mkd_keypoint *keypoints, float *descriptors, uint64_t max_out,
                       uint64_t *n_out, uint64_t *dropped_blobs, uint64_t *dropped_features) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "detect: n_out is null");
    *n_out = 0;
    if (dropped_blobs) *dropped_blobs = 0;
    if (dropped_features) *dropped_features = 0;
    if (max_out && (!keypoints || !descriptors)) return fail(h, LF_MKD_ERR_BAD_ARG, "detect: null output pointer");
    if (!image && !image_u8) return fail(h, LF_MKD_ERR_BAD_ARG, "detect: null image");
    if (!h->d_image || !h->d_pyr || width < 2 || height < 2 || width > h->params.max_image_width ||
        height > h->params.max_image_height)
        return fail(h, LF_MKD_ERR_BAD_ARG, "detect: image exceeds max_image_width/height given at creation");
    LF_ENTER(h);
    hipStream_t s = h->stream;
    const bool u8 = image_u8 != nullptr;
    if (u8)
        if (int rc = ensure_u8_staging(h)) return rc;
    // An extremum yields at most 18 keypoints: a capacity beyond that bound sizes nothing -- not the result staging, not
    // the pinned keypoint rows a recording copies on every call, not the key of a recording.
    max_out = std::min<uint64_t>(max_out, std::min<uint64_t>(top_n ? top_n : h->max_extrema, h->max_extrema) *
                                              LF_MKD_MAX_ANGLES_PER_EXTREMUM);
    uint32_t ms_bits;
    std::memcpy(&ms_bits, &min_size, 4);
    // Stage by stage: when asked for (the verification flag), when only the counts are wanted, and on handles whose keypoint
    // mode takes the two-launch form (POOL_F32, F16_FP6, FLAG_UNFUSED_KEYPOINTS: verification forms whose staging patches
    // are sized by the internal batch there, by max_out x 4 KiB in a recording) ...
    bool stepwise = (h->params.flags & LF_MKD_FLAG_DETECT_STEPWISE) || max_out == 0 || !fused_keypoints(h);
    // ... or, a request not yet due for recording, as the recording's own launches without recording them (`direct`: every
    // count handed on in device memory, one wait -- no capture, no instantiation, and none of the three waits either)
    bool direct = false;
    lf_mkd::DetectPlan *plan = nullptr;
    if (!stepwise) {
        for (auto &p : h->plans)
            if (p.w == width && p.h == height && p.top_n == top_n && p.min_size_bits == ms_bits && p.max_out == max_out && p.u8 == u8)
                plan = &p;
        // ... and while a request has not been seen record_after times (see lf_mkd::Sighting)
        if (!plan && h->record_after) {
            lf_mkd::Sighting *seen = nullptr;
            for (auto &g : h->sightings)
                if (g.w == width && g.h == height && g.top_n == top_n && g.min_size_bits == ms_bits && g.max_out == max_out && g.u8 == u8)
                    seen = &g;
            if (!seen) {
                if (h->sightings.size() >= 64) {     // the least recently seen request is forgotten
                    size_t old = 0;
                    for (size_t i = 1; i < h->sightings.size(); ++i)
                        if (h->sightings[i].stamp < h->sightings[old].stamp) old = i;
                    h->sightings.erase(h->sightings.begin() + long(old));
                }
                h->sightings.push_back(lf_mkd::Sighting{width, height, top_n, ms_bits, max_out, 0, 0, u8});
                seen = &h->sightings.back();
            }
            seen->stamp = ++h->plan_clock;
            if (seen->count < h->record_after) {
                ++seen->count;
                direct = true;
            }
        }
    }
    if (!stepwise) {
        // every buffer the recording names, before the upload is queued (growing one waits for the device)
        const uint64_t cap = top_n ? top_n : h->max_extrema;
        if (int rc = prepare_pipeline(h, top_n, cap)) return rc;
        if (int rc = ensure_orient_scratch(h, cap, true, max_out)) return rc;     // d_kps_out, d_det_desc [max_out]
        if (max_out > h->h_res_cap) {          // the recordings store into it by address
            retire_graph(h);
            if (h->h_res_kps) (void)hipHostFree(h->h_res_kps);
            h->h_res_kps = nullptr;
            h->h_res_cap = 0;
            LF_HIP(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_res_kps), max_out * sizeof(lf_mkd_keypoint), hipHostMallocDefault));
            h->h_res_cap = max_out;
        }
        if (!h->d_det_counts) {
            LF_HIP(h, hipMalloc(reinterpret_cast<void **>(&h->d_det_counts), 8 * sizeof(unsigned long long)));
            LF_HIP(h, hipMemsetAsync(h->d_det_counts, 0, 8 * sizeof(unsigned long long), s));
            LF_HIP(h, hipHostMalloc(reinterpret_cast<void **>(&h->h_det_counts), 8 * sizeof(unsigned long long), hipHostMallocDefault));
            LF_HIP(h, hipStreamSynchronize(s));
        }
    }
    const bool timed = (h->params.flags & LF_MKD_FLAG_KERNEL_TIMING) && !stepwise;
    if (timed) {
        for (auto &e : h->det_ev)
            if (!e) LF_HIP(h, hipEventCreate(&e));
        LF_HIP(h, hipEventRecord(h->det_ev[0], s));
    }
    auto upload_whole = [&]() -> int {
        if (u8) LF_HIP(h, hipMemcpyAsync(h->d_image_u8, image_u8, size_t(width) * height, hipMemcpyHostToDevice, s));
        else LF_HIP(h, hipMemcpyAsync(h->d_image, image, size_t(width) * height * 4, hipMemcpyHostToDevice, s));
        return LF_MKD_OK;
    };
    if (stepwise) {
        h->det_upload_ms = h->det_pipeline_ms = h->det_readback_ms = 0;    // lf_mkd_detect_times covers recorded calls only
        if (int rc = upload_whole()) return rc;
        return detect_stepwise(h, u8, width, height, top_n, min_size, keypoints, descriptors, max_out, n_out, dropped_blobs,
                               dropped_features);
    }
    describe_pyramid(width, height, h->pd);
    if (direct) {
        // the first sighting(s) of a request: upload in one piece, the pipeline's launches as they are, then the common tail
        if (h->pd.levels >= 2)
            if (int rc = ensure_side_stream(h, 2)) return rc;
        if (int rc = upload_whole()) return rc;
        if (timed) LF_HIP(h, hipEventRecord(h->det_ev[1], s));
        enqueue_pipeline(h, width, height, top_n, min_size, max_out, u8 ? nullptr : h->d_image, u8 ? h->d_image_u8 : nullptr,
                         reinterpret_cast<lf_mkd_keypoint *>(h->d_kps_out), h->d_det_desc, h->d_det_counts, h->h_det_counts,
                         h->h_res_kps, nullptr);
        LF_HIP(h, hipGetLastError());
    }
    // (the buffers above may have moved and retired every recording, `plan` among them: look again)
    plan = nullptr;
    for (auto &p : h->plans)
        if (!direct && p.w == width && p.h == height && p.top_n == top_n && p.min_size_bits == ms_bits && p.max_out == max_out && p.u8 == u8)
            plan = &p;
    if (!plan && !direct) {
        if (h->plans.size() >= kMaxPlans) {      // the least recently used recording makes room
            size_t old = 0;
            for (size_t i = 1; i < h->plans.size(); ++i)
                if (h->plans[i].stamp < h->plans[old].stamp) old = i;
            LF_HIP(h, hipStreamSynchronize(s));
            destroy_plan(h->plans[old]);
            h->plans.erase(h->plans.begin() + long(old));
        }
        lf_mkd::DetectPlan p;
        p.w = width; p.h = height; p.top_n = top_n; p.min_size_bits = ms_bits; p.max_out = max_out; p.u8 = u8;
        p.pd = h->pd;
        // Large frames go over PCIe in pieces, and the pipeline's front runs on the rows a piece completes while the next one
        // is on its way (RowBands; plan_cuts chooses the pieces).
        p.cuts = h->pd.levels >= 2 ? plan_cuts(h->n_layers, width, height, u8) : std::vector<uint32_t>();
        if (getenv("LF_MKD_BAND_DEBUG")) {
            std::string line = "lf_mkd: detect " + std::to_string(width) + "x" + std::to_string(height) + (u8 ? " u8" : " f32") +
                               ", " + std::to_string(h->n_layers) + " layers: cuts";
            for (uint32_t c : p.cuts) line += " " + std::to_string(c);
            const FrontModel m;
            line += p.cuts.empty() ? " (one piece)" : "";
            line += "; modelled front done at " +
                    std::to_string(front_finish_us(p.cuts, p.cuts.size() + 1, int(width), int(height), u8 ? 1 : 4, h->n_layers, m)) +
                    " us, one piece " +
                    std::to_string(front_finish_us({}, 1, int(width), int(height), u8 ? 1 : 4, h->n_layers, m)) + " us\n";
            fputs(line.c_str(), stderr);
        }
        if (!p.cuts.empty()) {
            if (!h->copy_stream) LF_HIP(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
            while (h->copy_ev.size() < p.cuts.size() + 1) {
                hipEvent_t e;
                LF_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                h->copy_ev.push_back(e);
            }
        }
        RowBands bands{};
        for (size_t piece = 0; piece <= p.cuts.size(); ++piece) {
            const bool last = piece == p.cuts.size();
            int rc = LF_MKD_OK;
            if (!p.cuts.empty() && !piece_bands(p.cuts, piece, int(width), int(height), h->n_layers, bands))
                rc = fail(h, LF_MKD_ERR_BAD_ARG, "detect: internal error planning the banded upload");
            hipGraph_t g = nullptr;
            hipGraphExec_t e = nullptr;
            if (!rc)
                rc = record_pipeline(h, width, height, top_n, min_size, max_out, u8 ? nullptr : h->d_image,
                                     u8 ? h->d_image_u8 : nullptr, reinterpret_cast<lf_mkd_keypoint *>(h->d_kps_out),
                                     h->d_det_desc, h->d_det_counts, h->h_det_counts, h->h_res_kps, &g, &e,
                                     p.cuts.empty() ? nullptr : &bands);
            if (rc) {
                destroy_plan(p);
                return rc;
            }
            if (last) {
                p.graph = g;
                p.exec = e;
            } else {
                p.head_graphs.push_back(g);
                p.heads.push_back(e);
            }
        }
        h->plans.push_back(p);
        plan = &h->plans.back();
    }
    if (plan) plan->stamp = ++h->plan_clock;
    if (direct) {
        // (already enqueued above)
    } else if (plan->cuts.empty()) {
        if (int rc = upload_whole()) return rc;
        if (timed) LF_HIP(h, hipEventRecord(h->det_ev[1], s));
    } else {
        // piece by piece (the call has waited for the device at its previous return: nothing still reads the staging frame):
        // copy, then the head on the rows it completes, the next piece beside it; after the last piece, the rest.  A copy from
        // pageable memory returns when its bytes are on their way, so the host is in the next copy while the device runs a head.
        const size_t bpp = u8 ? 1 : 4, row = size_t(width) * bpp;
        const unsigned char *src = u8 ? image_u8 : reinterpret_cast<const unsigned char *>(image);
        unsigned char *dst = u8 ? h->d_image_u8 : reinterpret_cast<unsigned char *>(h->d_image);
        size_t from = 0;
        static const bool trace = getenv("LF_MKD_BAND_TRACE") != nullptr;
        std::vector<double> stamps;
        const auto t_origin = std::chrono::steady_clock::now();
        auto stamp = [&] {
            if (trace) stamps.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_origin).count());
        };
        for (size_t piece = 0; piece <= plan->cuts.size(); ++piece) {
            const size_t to = piece < plan->cuts.size() ? plan->cuts[piece] : height;
            stamp();
            LF_HIP(h, hipMemcpyAsync(dst + row * from, src + row * from, row * (to - from), hipMemcpyHostToDevice, h->copy_stream));
            stamp();
            LF_HIP(h, hipEventRecord(h->copy_ev[piece], h->copy_stream));
            LF_HIP(h, hipStreamWaitEvent(s, h->copy_ev[piece], 0));
            stamp();
            if (piece < plan->cuts.size()) LF_HIP(h, hipGraphLaunch(plan->heads[piece], s));
            stamp();
            from = to;
        }
        if (trace) {
            std::string line = "lf_mkd: band trace (us: copy call begin, end, events done, head launched):";
            for (size_t i = 0; i < stamps.size(); ++i) line += (i % 4 ? " " : " | ") + std::to_string(int(stamps[i] + 0.5));
            fputs((line + "\n").c_str(), stderr);
        }
        if (timed) LF_HIP(h, hipEventRecord(h->det_ev[1], s));     // (on s behind the last wait: the moment the whole frame is there)
    }
    if (!direct) LF_HIP(h, hipGraphLaunch(plan->exec, s));
    if (timed) LF_HIP(h, hipEventRecord(h->det_ev[2], s));
    h->n_frames = 1;
    h->have_image = h->coarse_valid = h->coarse_l1_valid = true;   // the handle holds this frame's pyramid and a-trous stack
    LF_HIP(h, hipStreamSynchronize(s));
    const auto t_back = std::chrono::steady_clock::now();
    if (timed) {
        float a = 0, b = 0;
        LF_HIP(h, hipEventElapsedTime(&a, h->det_ev[0], h->det_ev[1]));
        LF_HIP(h, hipEventElapsedTime(&b, h->det_ev[1], h->det_ev[2]));
        h->det_upload_ms = a;
        h->det_pipeline_ms = b;
        h->det_readback_ms = 0;
    }
    const unsigned long long *c = h->h_det_counts;
    if (dropped_blobs) *dropped_blobs = c[1];
    if (dropped_features) *dropped_features = c[4];
    const uint64_t n_kp = std::min<uint64_t>(c[3], max_out);
    if (n_kp == 0) return LF_MKD_OK;
    LF_HIP(h, hipMemcpyAsync(descriptors, h->d_det_desc, n_kp * kOut * 4, hipMemcpyDeviceToHost, s));
    std::memcpy(keypoints, h->h_res_kps, n_kp * sizeof(lf_mkd_keypoint));
    LF_HIP(h, hipStreamSynchronize(s));
    if (timed) h->det_readback_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_back).count();
    *n_out = n_kp;
    return LF_MKD_OK;
}

int lf_mkd_detect_recordings(const lf_mkd *h, uint32_t *n_recordings, uint32_t *n_banded, uint32_t *n_sightings) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    uint32_t banded = 0;
    for (const auto &p : h->plans) banded += p.cuts.empty() ? 0u : 1u;
    if (n_recordings) *n_recordings = uint32_t(h->plans.size());
    if (n_banded) *n_banded = banded;
    if (n_sightings) *n_sightings = uint32_t(h->sightings.size());
    return LF_MKD_OK;
}

int lf_mkd_detect_times(lf_mkd *h, double *upload_ms, double *pipeline_ms, double *readback_ms) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!(h->params.flags & LF_MKD_FLAG_KERNEL_TIMING))
        return fail(h, LF_MKD_ERR_BAD_ARG, "detect_times: the handle was not created with LF_MKD_FLAG_KERNEL_TIMING");
    if (upload_ms) *upload_ms = h->det_upload_ms;
    if (pipeline_ms) *pipeline_ms = h->det_pipeline_ms;
    if (readback_ms) *readback_ms = h->det_readback_ms;
    return LF_MKD_OK;
}

int lf_mkd_detect(lf_mkd *h, const float *image, uint32_t width, uint32_t height, uint32_t top_n, float min_size,
                  lf_mkd_keypoint *keypoints, float *descriptors, uint64_t max_out, uint64_t *n_out,
                  uint64_t *dropped_blobs, uint64_t *dropped_features) {
    if (h && !image) return fail(h, LF_MKD_ERR_BAD_ARG, "detect: null image");
    return detect_host(h, image, nullptr, width, height, top_n, min_size, keypoints, descriptors, max_out, n_out,
                       dropped_blobs, dropped_features);
}

int lf_mkd_detect_u8(lf_mkd *h, const uint8_t *image, uint32_t width, uint32_t height, uint32_t top_n, float min_size,
                     lf_mkd_keypoint *keypoints, float *descriptors, uint64_t max_out, uint64_t *n_out,
                     uint64_t *dropped_blobs, uint64_t *dropped_features) {
    if (h && !image) return fail(h, LF_MKD_ERR_BAD_ARG, "detect_u8: null image");
    return detect_host(h, nullptr, image, width, height, top_n, min_size, keypoints, descriptors, max_out, n_out,
                       dropped_blobs, dropped_features);
}

int lf_mkd_detect_frames_device(lf_mkd *h, const float *d_images, uint32_t n_frames, uint32_t width, uint32_t height,
                                uint32_t top_n, float min_size, lf_mkd_keypoint *d_keypoints, uint32_t *d_frame_of_kp,
                                float *d_descriptors, uint64_t max_out, uint64_t *n_out, uint64_t *dropped_blobs,
                                uint64_t *dropped_features, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "detect_frames: n_out is null");
    *n_out = 0;
    if (dropped_blobs) *dropped_blobs = 0;
    if (dropped_features) *dropped_features = 0;
    if (max_out && (!d_keypoints || !d_descriptors || !d_frame_of_kp))
        return fail(h, LF_MKD_ERR_BAD_ARG, "detect_frames: null output pointer");
    LF_ENTER(h);
    if (int rc = lf_mkd_set_images_device(h, d_images, n_frames, width, height, stream)) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    const uint64_t cap_f = h->max_extrema;                       // extrema kept per frame (mod.rs:625-633)
    const uint64_t keep = top_n ? std::min<uint64_t>(top_n, cap_f) : cap_f;
    const uint64_t all_cap = cap_f * n_frames * 2;               // room for frames that exceed their share
    if (int rc = ensure_coarse_stack(h, s)) return rc;
    if (int rc = ensure_detect_scratch(h)) return rc;
    if (int rc = grow(h, &h->d_det_extrema, &h->det_out_cap, all_cap, sizeof(lf_mkd_extremum), kRecorded)) return rc;
    if (int rc = grow(h, &h->d_mf_frame_start, &h->mf_start_cap, n_frames, 4)) return rc;
    if (int rc = grow(h, &h->d_mf_offsets, &h->mf_off_cap, n_frames, 4)) return rc;
    if (int rc = grow(h, &h->d_mf_padded, &h->mf_padded_cap, keep * n_frames, sizeof(lf_mkd_extremum))) return rc;
    if (int rc = grow(h, &h->d_mf_list, &h->mf_list_cap, keep * n_frames, sizeof(lf_mkd_extremum))) return rc;
    if (int rc = grow(h, &h->d_mf_frame_of, &h->mf_fo_cap, keep * n_frames, 4)) return rc;
    // 1. every frame's extrema, ordered by frame; 2. per-frame selection; 3. one contiguous list + frame ids
    launch_detect_extrema(h->d_pyr + h->pd.offset[0], h->pyr_stride, h->pd.pitch[0], h->d_coarse, h->coarse_stride, h->layer_stride,
                          h->n_layers, int(width), int(height), int(n_frames), kBorder, kSkipLayers, kContrastThreshold,
                          h->d_slots, h->d_cube_counts, h->d_cube_sums, h->d_det_extrema, nullptr, h->d_mf_frame_start,
                          all_cap, h->d_totals + 0, s);
    launch_topk_filter(h->d_det_extrema, h->d_mf_frame_start, h->d_totals + 0, 0, n_frames, unsigned(cap_f),
                       unsigned(keep), top_n ? min_size : -INFINITY, h->d_mf_padded, nullptr, h->d_sel_count, nullptr, 0,
                       nullptr, s);
    launch_segments_compact(h->d_mf_padded, h->d_sel_count, h->d_mf_frame_start, h->d_totals + 0, n_frames,
                            unsigned(cap_f), unsigned(keep), h->d_mf_offsets, h->d_mf_list, h->d_mf_frame_of,
                            h->d_totals + 2, s);
    LF_HIP(h, hipGetLastError());
    unsigned long long totals[4] = {0, 0, 0, 0};
    LF_HIP(h, hipMemcpyAsync(totals, h->d_totals, sizeof(totals), hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipStreamSynchronize(s));
    const uint64_t n_sel = totals[2];
    if (dropped_blobs) *dropped_blobs = totals[1] + totals[3];
    if (n_sel == 0 || max_out == 0) return LF_MKD_OK;
    // 4. orientation over the whole list, 5. sampling + description by frame id
    if (int rc = ensure_orient_scratch(h, n_sel, false, 0)) return rc;
    uint64_t n_kp = 0;
    if (int rc = orient_device(h, h->d_mf_list, h->d_mf_frame_of, n_sel, reinterpret_cast<float *>(d_keypoints),
                               d_frame_of_kp, max_out, &n_kp, dropped_features, s))
        return rc;
    *n_out = n_kp;
    if (n_kp == 0) return LF_MKD_OK;
    return lf_mkd_describe_keypoints_frames_device(h, d_keypoints, d_frame_of_kp, n_kp, d_descriptors, s);
}

int lf_mkd_stream_create(lf_mkd *h, uint32_t width, uint32_t height, uint32_t top_n, float min_size, uint64_t max_out,
                         const float *d_image, lf_mkd_keypoint *d_keypoints, float *d_descriptors, uint64_t *d_counts) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!d_image || !d_keypoints || !d_descriptors || !d_counts || max_out == 0 || width < 2 || height < 2)
        return fail(h, LF_MKD_ERR_BAD_ARG, "stream_create: bad argument");
    if (!h->d_pyr || width > h->params.max_image_width || height > h->params.max_image_height)
        return fail(h, LF_MKD_ERR_BAD_ARG, "stream_create: frame exceeds max_image_width/height given at creation");
    LF_ENTER(h);
    LF_HIP(h, hipStreamSynchronize(h->stream));
    // an earlier recording may still be running on a caller's stream
    if (h->graph_exec) {
        (void)hipDeviceSynchronize();
        (void)hipGraphExecDestroy(h->graph_exec);
        (void)hipGraphDestroy(h->graph);
        h->graph_exec = nullptr;
        h->graph = nullptr;
    }
    // every allocation happens before the capture starts
    describe_pyramid(width, height, h->pd);
    h->n_frames = 1;
    // no frame has gone through the recorded pipeline yet: until lf_mkd_stream_frame runs, the pyramid and the a-trous
    // stack hold nothing the keypoint / orientation / verification entry points could use
    h->have_image = false;
    h->coarse_valid = h->coarse_l1_valid = false;
    const uint64_t cap = top_n ? top_n : h->max_extrema;   // extrema that can reach orientation
    if (int rc = prepare_pipeline(h, top_n, cap)) return rc;
    if (!fused_keypoints(h))
        if (int rc = grow(h, &h->d_stream_patches, &h->stream_patch_cap, max_out * kPx, sizeof(float), kRecorded)) return rc;
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(d_counts);
    LF_HIP(h, hipMemsetAsync(cnt, 0, 8 * sizeof(unsigned long long), h->stream));
    LF_HIP(h, hipStreamSynchronize(h->stream));
    if (int rc = record_pipeline(h, width, height, top_n, min_size, max_out, d_image, nullptr, d_keypoints, d_descriptors, cnt,
                                 nullptr, nullptr, &h->graph, &h->graph_exec))
        return rc;
    h->graph_pd = h->pd;
    return LF_MKD_OK;
}

int lf_mkd_stream_frame(lf_mkd *h, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!h->graph_exec)
        return fail(h, LF_MKD_ERR_BAD_ARG, "stream_frame: no recorded pipeline (call lf_mkd_stream_create; a call that "
                                           "grew the handle's scratch buffers retires an earlier recording)");
    LF_ENTER(h);
    LF_HIP(h, hipGraphLaunch(h->graph_exec, stream ? static_cast<hipStream_t>(stream) : h->stream));
    // every launch rebuilds the pyramid and the a-trous stack of the frame in d_image: work enqueued behind it on the same
    // stream (describe_keypoints, orientation, the verification taps) sees that frame
    h->pd = h->graph_pd;
    h->n_frames = 1;
    h->have_image = h->coarse_valid = h->coarse_l1_valid = true;
    return LF_MKD_OK;
}

// keep_overflow: the word lf_mkd_match_overflowed reads is added to, not reset (the second direction of lf_mkd_match_both_device)
static int match_device_impl(lf_mkd *h, const float *d_a, uint64_t na, const float *d_b, uint64_t nb,
                             const uint32_t *d_exclude_lo, const uint32_t *d_exclude_hi, float ratio, int32_t *d_match,
                             float *d_best, float *d_second, void *stream, bool keep_overflow) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (na == 0) return LF_MKD_OK;
    if (!d_a || !d_b || !d_match) return fail(h, LF_MKD_ERR_BAD_ARG, "match_device: null pointer");
    if (nb < 2) return fail(h, LF_MKD_ERR_BAD_ARG, "match: needs at least two candidates in b (main.rs:20)");
    if ((d_exclude_lo == nullptr) != (d_exclude_hi == nullptr))
        return fail(h, LF_MKD_ERR_BAD_ARG, "match_device: exclude_lo and exclude_hi go together");
    if (na > 0x7FFFFFFFull || nb > 0x7FFFFFFFull) return fail(h, LF_MKD_ERR_BAD_ARG, "match: more than 2^31 rows");
    if ((reinterpret_cast<uintptr_t>(d_a) | reinterpret_cast<uintptr_t>(d_b)) & 15)
        return fail(h, LF_MKD_ERR_BAD_ARG, "match_device: d_a and d_b must be 16-byte aligned");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    // Which form: the two passes win once the scan would take about a millisecond and a has enough rows to fill the chip
    // without splitting b many ways (every b split starts its candidate lists from nothing); below that -- the reference's
    // own 2000 x 2000 included -- the three-term scan alone is faster.  LF_MKD_MATCH=scan / =screen in the environment
    // force one form (A/B runs, tests).
    // Small problems -- the reference's own 2000 x 2000 (examples/match_images/src/main.rs:62-76) -- take ONE launch that
    // reads the f32 rows directly (match_small: no operand tiles, no partials, no merge kernel): below ~10^4 rows the chain
    // of split / scan / merge launches is launch cadence, not arithmetic.  LF_MKD_MATCH=small forces it where it fits.
    const char *form = getenv("LF_MKD_MATCH");
    const bool want_small = form ? (form[0] == 's' && form[1] == 'm') : true;
    if (want_small && match_small_fits(long(na), long(nb))) {
        launch_match_small(d_a, long(na), d_b, long(nb), d_exclude_lo, d_exclude_hi, ratio, d_match, d_best, d_second,
                           h->d_match_misc && !keep_overflow ? h->d_match_misc + 2 : nullptr, s);
        LF_HIP(h, hipGetLastError());
        return LF_MKD_OK;
    }
    // (=small where it does not fit: the scan, as include/lf_mkd.h says -- never silently the two-pass form)
    const bool force_scan = form && ((form[0] == 's' && form[1] == 'c' && form[2] == 'a') || (form[0] == 's' && form[1] == 'm'));
    const bool three_term_only = force_scan ? true
                                            : (form && form[0] == 's' && form[1] == 'c' ? false : !(na >= 16384 && na * nb >= (1ull << 29)));
    // a goes through in chunks, so that the per-row scratch (2 KiB of candidate records per a row and b split) stays bounded
    const uint64_t chunk = three_term_only ? na : std::min<uint64_t>(na, kMatchChunk);
    const int splits = match_splits(long(chunk), long(nb), h->num_cus);
    if (int rc = grow(h, &h->d_match_a, &h->match_a_cap, match_tiles_bytes(long(chunk)), 1)) return rc;
    if (int rc = grow(h, &h->d_match_b, &h->match_b_cap, match_tiles_bytes(long(nb)), 1)) return rc;
    if (int rc = grow(h, &h->d_match_part, &h->match_part_cap, uint64_t(splits) * chunk * 3, sizeof(float))) return rc;
    float *p_best = h->d_match_part, *p_second = p_best + uint64_t(splits) * chunk;
    int *p_index = reinterpret_cast<int *>(p_second + uint64_t(splits) * chunk);
    if (three_term_only) {
        // lf_mkd_match_overflowed reports on the LATEST call: this form redoes nothing
        if (h->d_match_misc && !keep_overflow) LF_HIP(h, hipMemsetAsync(h->d_match_misc + 2, 0, sizeof(unsigned), s));
        launch_match_split(d_a, long(na), h->d_match_a, nullptr, nullptr, s);
        launch_match_split(d_b, long(nb), h->d_match_b, nullptr, nullptr, s);
        launch_match(h->d_match_a, long(na), h->d_match_b, long(nb), d_exclude_lo, d_exclude_hi, ratio, splits, p_best,
                     p_index, p_second, d_match, d_best, d_second, nullptr, s);
        LF_HIP(h, hipGetLastError());
        return LF_MKD_OK;
    }
    if (int rc = grow(h, &h->d_match_rec, &h->match_rec_cap, match_record_bytes(long(chunk), splits), 1)) return rc;
    if (int rc = grow(h, &h->d_match_cnt, &h->match_cnt_cap, match_count_bytes(long(chunk), splits), 1)) return rc;
    if (int rc = grow(h, &h->d_match_norm, &h->match_norm_cap, chunk, sizeof(float))) return rc;
    if (int rc = grow(h, &h->d_match_floor, &h->match_floor_cap, chunk, sizeof(float))) return rc;   // the splits' shared bounds
    if (int rc = ensure_match_misc(h, s)) return rc;
    if (int rc = grow(h, &h->d_match_few_tiles, &h->match_few_tiles_cap, match_few_tiles_bytes(), 1)) return rc;
    if (int rc = grow(h, &h->d_match_few, &h->match_few_cap, match_few_words(), sizeof(unsigned))) return rc;
    // misc: [0] largest |b| (float bits), [1] rows of the current chunk whose records overflowed, [2] the same over the call
    LF_HIP(h, hipMemsetAsync(h->d_match_misc, 0, (keep_overflow ? 2 : 3) * sizeof(unsigned), s));
    unsigned *b_max = h->d_match_misc;
    int *n_over = reinterpret_cast<int *>(h->d_match_misc + 1);
    launch_match_split(d_b, long(nb), h->d_match_b, nullptr, b_max, s);
    for (uint64_t at = 0; at < na; at += chunk) {
        const long n = long(std::min(chunk, na - at));
        const float *a = d_a + at * kOut;
        const uint32_t *lo = d_exclude_lo ? d_exclude_lo + at : nullptr, *hi = d_exclude_hi ? d_exclude_hi + at : nullptr;
        float *best = d_best ? d_best + at : nullptr, *second = d_second ? d_second + at : nullptr;
        if (at) LF_HIP(h, hipMemsetAsync(n_over, 0, sizeof(int), s));
        launch_match_split(a, n, h->d_match_a, h->d_match_norm, nullptr, s);
        launch_match_screen(h->d_match_a, n, h->d_match_b, long(nb), lo, hi, splits, h->d_match_norm, b_max,
                            h->d_match_rec, h->d_match_cnt, s, reinterpret_cast<int *>(h->d_match_floor));
        launch_match_verify(a, n, d_b, h->d_match_norm, b_max, h->d_match_rec, h->d_match_cnt, splits, ratio,
                            d_match + at, best, second, n_over, reinterpret_cast<int *>(h->d_match_few), s);
        // rows whose records overflowed (more than 64 near-best candidates in one lane's share of b) are redone by the
        // three-term scan: on their own when they are few, else the whole chunk; both are enqueued unconditionally and
        // read the count on the device -- no host round trip, and nothing to do in the ordinary case
        launch_match_few(a, h->d_match_b, long(nb), lo, hi, ratio, n_over, h->d_match_few_tiles, h->d_match_few,
                         d_match + at, best, second, s);
        launch_match(h->d_match_a, n, h->d_match_b, long(nb), lo, hi, ratio, splits, p_best, p_index, p_second,
                     d_match + at, best, second, n_over, s);
    }
    LF_HIP(h, hipGetLastError());
    return LF_MKD_OK;
}

int lf_mkd_match_device(lf_mkd *h, const float *d_a, uint64_t na, const float *d_b, uint64_t nb,
                        const uint32_t *d_exclude_lo, const uint32_t *d_exclude_hi, float ratio, int32_t *d_match,
                        float *d_best, float *d_second, void *stream) {
    return match_device_impl(h, d_a, na, d_b, nb, d_exclude_lo, d_exclude_hi, ratio, d_match, d_best, d_second, stream, false);
}

int lf_mkd_match_both_device(lf_mkd *h, const float *d_a, uint64_t na, const float *d_b, uint64_t nb, float ratio,
                             int32_t *d_match_ab, int32_t *d_match_ba, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    // an empty side: nothing to match in either direction, nothing is written (as lf_mkd_match_device with na == 0)
    if (na == 0 || nb == 0) return LF_MKD_OK;
    if (!d_a || !d_b || !d_match_ab || !d_match_ba) return fail(h, LF_MKD_ERR_BAD_ARG, "match_both_device: null pointer");
    if (na < 2 || nb < 2) return fail(h, LF_MKD_ERR_BAD_ARG, "match_both: needs at least two rows on either side (main.rs:20)");
    if ((reinterpret_cast<uintptr_t>(d_a) | reinterpret_cast<uintptr_t>(d_b)) & 15)
        return fail(h, LF_MKD_ERR_BAD_ARG, "match_both_device: d_a and d_b must be 16-byte aligned");
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    const char *form = getenv("LF_MKD_MATCH");
    if (!form && match_small_fits(long(na), long(nb)) && match_small_fits(long(nb), long(na))) {
        launch_match_small_both(d_a, long(na), d_b, long(nb), ratio, d_match_ab, d_match_ba,
                                h->d_match_misc ? h->d_match_misc + 2 : nullptr, s);
        LF_HIP(h, hipGetLastError());
        return LF_MKD_OK;
    }
    // larger problems (or a forced form): one call per direction, each in the form its size takes; the second adds its
    // redone rows to the first's, so that lf_mkd_match_overflowed reports on the whole call
    if (int rc = ensure_match_misc(h, s)) return rc;
    if (int rc = match_device_impl(h, d_a, na, d_b, nb, nullptr, nullptr, ratio, d_match_ab, nullptr, nullptr, s, false)) return rc;
    return match_device_impl(h, d_b, nb, d_a, na, nullptr, nullptr, ratio, d_match_ba, nullptr, nullptr, s, true);
}

int lf_mkd_match_overflowed(lf_mkd *h, void *stream, uint64_t *n_rows) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_rows) return fail(h, LF_MKD_ERR_BAD_ARG, "match_overflowed: null pointer");
    *n_rows = 0;
    if (!h->d_match_misc) return LF_MKD_OK;           // no two-pass match yet
    LF_ENTER(h);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    int n = 0;
    LF_HIP(h, hipMemcpyAsync(&n, h->d_match_misc + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    LF_HIP(h, hipStreamSynchronize(s));
    *n_rows = uint64_t(n);
    return LF_MKD_OK;
}

int lf_mkd_match(lf_mkd *h, const float *a, uint64_t na, const float *b, uint64_t nb, float ratio, int32_t *match) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (na == 0) return LF_MKD_OK;
    if (!a || !b || !match) return fail(h, LF_MKD_ERR_BAD_ARG, "match: null pointer");
    if (nb < 2) return fail(h, LF_MKD_ERR_BAD_ARG, "match: needs at least two candidates in b (main.rs:20)");
    LF_ENTER(h);
    if (int rc = grow(h, &h->d_match_in, &h->match_in_cap, (na + nb) * kOut, sizeof(float))) return rc;
    if (int rc = grow(h, &h->d_match_out, &h->match_out_cap, na, sizeof(int))) return rc;
    LF_HIP(h, hipMemcpyAsync(h->d_match_in, a, na * kOut * 4, hipMemcpyHostToDevice, h->stream));
    LF_HIP(h, hipMemcpyAsync(h->d_match_in + na * kOut, b, nb * kOut * 4, hipMemcpyHostToDevice, h->stream));
    if (int rc = lf_mkd_match_device(h, h->d_match_in, na, h->d_match_in + na * kOut, nb, nullptr, nullptr, ratio,
                                     h->d_match_out, nullptr, nullptr, h->stream))
        return rc;
    LF_HIP(h, hipMemcpyAsync(match, h->d_match_out, na * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    LF_HIP(h, hipStreamSynchronize(h->stream));
    return LF_MKD_OK;
}

int lf_mkd_orient_keypoints_blocked(lf_mkd *h, const float *extremum_data, uint64_t n_extrema, uint32_t block_len,
                                    const uint32_t *indices, uint64_t n_indices, uint32_t *kp_extremum_index,
                                    float *kp_orientation, lf_mkd_keypoint *keypoints, uint64_t max_out, uint64_t *n_out,
                                    uint64_t *n_dropped) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!n_out) return fail(h, LF_MKD_ERR_BAD_ARG, "orient_keypoints_blocked: n_out is null");
    *n_out = 0;
    if (n_dropped) *n_dropped = 0;
    if (n_indices == 0) return LF_MKD_OK;
    if (!extremum_data || !indices || block_len == 0 || (max_out && (!kp_extremum_index || !kp_orientation)))
        return fail(h, LF_MKD_ERR_BAD_ARG, "orient_keypoints_blocked: null pointer");
    // gather the blocked coordinate arrays into records (BlobLocations::get, shaders.rs:307-318)
    std::vector<lf_mkd_extremum> ex(n_indices);
    for (uint64_t i = 0; i < n_indices; ++i) {
        const uint64_t idx = indices[i];
        if (idx >= n_extrema) return fail(h, LF_MKD_ERR_BAD_ARG, "orient_keypoints_blocked: index beyond n_extrema");
        const uint64_t base = idx / block_len * 4 * block_len, off = idx % block_len;
        ex[i] = lf_mkd_extremum{extremum_data[base + off], extremum_data[base + block_len + off],
                                extremum_data[base + 2 * block_len + off], extremum_data[base + 3 * block_len + off]};
    }
    std::vector<lf_mkd_keypoint> kps(max_out);
    if (int rc = lf_mkd_orient_keypoints(h, ex.data(), n_indices, kps.data(), max_out, n_out, n_dropped)) return rc;
    // keypoints come ordered by extremum: walk both lists together to recover each keypoint's extremum
    uint64_t j = 0;
    for (uint64_t i = 0; i < *n_out; ++i) {
        while (j < n_indices && !(ex[j].x == kps[i].x && ex[j].y == kps[i].y && ex[j].size == kps[i].size &&
                                  ex[j].response == kps[i].response))
            ++j;
        if (j == n_indices) return fail(h, LF_MKD_ERR_HIP, "orient_keypoints_blocked: keypoint without an extremum");
        kp_extremum_index[i] = indices[j];
        kp_orientation[i] = kps[i].angle;
        if (keypoints) keypoints[i] = kps[i];
    }
    return LF_MKD_OK;
}

int lf_mkd_get_coarse_layer(lf_mkd *h, uint32_t layer, float *out) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "get_coarse_layer: call lf_mkd_set_image first");
    if (layer >= uint32_t(h->n_layers) || !out) return fail(h, LF_MKD_ERR_BAD_ARG, "get_coarse_layer: bad layer");
    LF_ENTER(h);
    if (int rc = ensure_coarse_stack(h, h->stream)) return rc;
    LF_HIP(h, hipStreamSynchronize(h->stream));
    const float *src = layer == 0 ? h->d_pyr + h->pd.offset[0] : h->d_coarse + long(layer - 1) * h->layer_stride;
    const size_t spitch = size_t(layer == 0 ? h->pd.pitch[0] : h->pd.w[0]) * 4;
    LF_HIP(h, hipMemcpy2D(out, size_t(h->pd.w[0]) * 4, src, spitch, size_t(h->pd.w[0]) * 4, size_t(h->pd.h[0]),
                          hipMemcpyDeviceToHost));
    return LF_MKD_OK;
}

int lf_mkd_get_pyramid_level(lf_mkd *h, uint32_t level, float *out, uint32_t *w, uint32_t *hgt) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "get_pyramid_level: call lf_mkd_set_image first");
    if (level >= uint32_t(h->pd.levels)) return fail(h, LF_MKD_ERR_BAD_ARG, "get_pyramid_level: no such level");
    if (w) *w = uint32_t(h->pd.w[level]);
    if (hgt) *hgt = uint32_t(h->pd.h[level]);
    if (!out) return LF_MKD_OK;
    LF_ENTER(h);
    LF_HIP(h, hipStreamSynchronize(h->stream));
    LF_HIP(h, hipMemcpy2D(out, size_t(h->pd.w[level]) * 4, h->d_pyr + h->pd.offset[level], size_t(h->pd.pitch[level]) * 4,
                          size_t(h->pd.w[level]) * 4, size_t(h->pd.h[level]), hipMemcpyDeviceToHost));
    return LF_MKD_OK;
}

int lf_mkd_get_pyramid_level_apron(lf_mkd *h, uint32_t level, float *out, uint32_t *w, uint32_t *hgt, uint32_t *apron) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!h->have_image) return fail(h, LF_MKD_ERR_NO_IMAGE, "get_pyramid_level_apron: call lf_mkd_set_image first");
    if (level >= uint32_t(h->pd.levels)) return fail(h, LF_MKD_ERR_BAD_ARG, "get_pyramid_level_apron: no such level");
    const int a = h->pd.apron[level], pw = h->pd.w[level] + 2 * a, ph = h->pd.h[level] + 2 * a;
    if (w) *w = uint32_t(h->pd.w[level]);
    if (hgt) *hgt = uint32_t(h->pd.h[level]);
    if (apron) *apron = uint32_t(a);
    if (!out) return LF_MKD_OK;
    LF_ENTER(h);
    LF_HIP(h, hipStreamSynchronize(h->stream));
    const float *src = h->d_pyr + h->pd.offset[level] - long(a) * h->pd.pitch[level] - a;
    LF_HIP(h, hipMemcpy2D(out, size_t(pw) * 4, src, size_t(h->pd.pitch[level]) * 4, size_t(pw) * 4, size_t(ph),
                          hipMemcpyDeviceToHost));
    return LF_MKD_OK;
}

}  // extern "C"
