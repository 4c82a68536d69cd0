//! Raw declarations of `include/lf_mkd.h` (liblf_mkd.so, the MI355X-native MKD path): one `extern "C"` item per entry
//! point of the header, `#[repr(C)]` twins of its structs.  `tests/test_rust_binding.py` checks every item here against
//! the header (name, arity, pointer/scalar class of each argument), and the struct layouts against `offsetof`.
//!
//! New file in the reference tree: `local_features/src/hip/ffi.rs`.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const LF_MKD_OK: c_int = 0;
pub const LF_MKD_ERR_BAD_ARG: c_int = -1;
pub const LF_MKD_ERR_HIP: c_int = -2;
pub const LF_MKD_ERR_IO: c_int = -3;
pub const LF_MKD_ERR_NO_IMAGE: c_int = -4;
pub const LF_MKD_ERR_NO_DEVICE: c_int = -5;
pub const LF_MKD_ERR_COMM: c_int = -6;

pub const LF_MKD_ANGLE_SHADER: i32 = 0;
pub const LF_MKD_ANGLE_EXACT: i32 = 1;
pub const LF_MKD_ANGLE_EXACT_ZERO: i32 = 2;
pub const LF_MKD_POOL_DEFAULT: i32 = 0;
pub const LF_MKD_POOL_F16X3: i32 = 1;
pub const LF_MKD_POOL_F32: i32 = 2;
pub const LF_MKD_POOL_F16_FP6: i32 = 3;
pub const LF_MKD_FLAG_KERNEL_TIMING: u32 = 1;
pub const LF_MKD_FLAG_UNFUSED_KEYPOINTS: u32 = 2;
pub const LF_MKD_FLAG_DETECT_STEPWISE: u32 = 4;
pub const LF_MKD_MAX_ANGLES_PER_EXTREMUM: usize = 18;
pub const LF_MKD_COMM_ID_BYTES: usize = 128;
pub const LF_MKD_GATHER_DIRECT: i32 = 0;
pub const LF_MKD_GATHER_RING: i32 = 1;

/// `lf_mkd_params`: BuildTimeParams (lib.rs:54-75) + FeatureDetectParams (lib.rs:34-52) for this path; 0 = default.
#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct lf_mkd_params {
    pub max_image_width: u32,
    pub max_image_height: u32,
    pub max_features: u32,
    pub patch_scale_factor: f32,
    pub device: i32,
    pub angle_mode: i32,
    pub pool_mode: i32,
    pub flags: u32,
    pub max_frames: u32,
    pub n_scales: u32,
    pub max_blobs: u32,
    pub reserved: [u32; 1],
}

/// `lf_mkd_keypoint`: layout-identical to `crate::Keypoint` (lib.rs:17-24), angle in degrees.
#[repr(C)]
#[derive(Debug, Clone, Copy, Default, PartialEq)]
pub struct lf_mkd_keypoint {
    pub x: f32,
    pub y: f32,
    pub size: f32,
    pub angle: f32,
    pub response: f32,
}

/// `lf_mkd_extremum`: one refined blob {x, y, interpolated scale, contrast} (ExtremumLocations, common.glsl:45-81).
#[repr(C)]
#[derive(Debug, Clone, Copy, Default, PartialEq)]
pub struct lf_mkd_extremum {
    pub x: f32,
    pub y: f32,
    pub size: f32,
    pub response: f32,
}

/// Opaque handle; owns all device memory.
#[repr(C)]
pub struct lf_mkd {
    _private: [u8; 0],
}

/// Opaque RCCL communicator of one rank (`lf_mkd_comm_create`).
#[repr(C)]
pub struct lf_mkd_comm {
    _private: [u8; 0],
}

extern "C" {
    pub fn lf_mkd_create(params: *const lf_mkd_params, mean: *const f32, eigvals: *const f32, eigvecs: *const f32,
                         out: *mut *mut lf_mkd) -> c_int;
    pub fn lf_mkd_create_from_file(params: *const lf_mkd_params, safetensors_path: *const c_char,
                                   out: *mut *mut lf_mkd) -> c_int;
    pub fn lf_mkd_destroy(h: *mut lf_mkd);
    pub fn lf_mkd_last_error(h: *const lf_mkd) -> *const c_char;

    pub fn lf_mkd_describe_patches(h: *mut lf_mkd, patches: *const f32, n: u64, out: *mut f32) -> c_int;
    pub fn lf_mkd_describe_patches_device(h: *mut lf_mkd, d_patches: *const f32, n: u64, d_out: *mut f32,
                                          stream: *mut c_void) -> c_int;
    pub fn lf_mkd_raw_descriptors_device(h: *mut lf_mkd, d_patches: *const f32, n: u64, d_raw: *mut f32,
                                         stream: *mut c_void) -> c_int;

    pub fn lf_mkd_set_image(h: *mut lf_mkd, image: *const f32, width: u32, height: u32) -> c_int;
    pub fn lf_mkd_set_image_device(h: *mut lf_mkd, d_image: *const f32, width: u32, height: u32,
                                   stream: *mut c_void) -> c_int;
    pub fn lf_mkd_set_images_device(h: *mut lf_mkd, d_images: *const f32, n_frames: u32, width: u32, height: u32,
                                    stream: *mut c_void) -> c_int;
    pub fn lf_mkd_set_image_u8(h: *mut lf_mkd, image: *const u8, width: u32, height: u32) -> c_int;
    pub fn lf_mkd_set_images_u8_device(h: *mut lf_mkd, d_images: *const u8, n_frames: u32, width: u32, height: u32,
                                       stream: *mut c_void) -> c_int;

    pub fn lf_mkd_describe_keypoints(h: *mut lf_mkd, kps: *const lf_mkd_keypoint, n: u64, out: *mut f32) -> c_int;
    pub fn lf_mkd_describe_keypoints_device(h: *mut lf_mkd, d_kps: *const lf_mkd_keypoint, n: u64, d_out: *mut f32,
                                            stream: *mut c_void) -> c_int;
    pub fn lf_mkd_describe_keypoints_frames_device(h: *mut lf_mkd, d_kps: *const lf_mkd_keypoint,
                                                   d_frame_of_kp: *const u32, n: u64, d_out: *mut f32,
                                                   stream: *mut c_void) -> c_int;

    pub fn lf_mkd_orient_keypoints(h: *mut lf_mkd, extrema: *const lf_mkd_extremum, n: u64,
                                   out: *mut lf_mkd_keypoint, max_out: u64, n_out: *mut u64,
                                   n_dropped: *mut u64) -> c_int;
    pub fn lf_mkd_orient_keypoints_device(h: *mut lf_mkd, d_extrema: *const lf_mkd_extremum,
                                          d_frame_of_extremum: *const u32, n: u64, d_out: *mut lf_mkd_keypoint,
                                          d_frame_of_kp: *mut u32, max_out: u64, n_out: *mut u64,
                                          n_dropped: *mut u64, stream: *mut c_void) -> c_int;
    pub fn lf_mkd_orient_keypoints_blocked(h: *mut lf_mkd, extremum_data: *const f32, n_extrema: u64, block_len: u32,
                                           indices: *const u32, n_indices: u64, kp_extremum_index: *mut u32,
                                           kp_orientation: *mut f32, keypoints: *mut lf_mkd_keypoint, max_out: u64,
                                           n_out: *mut u64, n_dropped: *mut u64) -> c_int;

    pub fn lf_mkd_detect_extrema_device(h: *mut lf_mkd, d_out: *mut lf_mkd_extremum, d_frame_of: *mut u32,
                                        max_out: u64, n_out: *mut u64, n_dropped: *mut u64,
                                        stream: *mut c_void) -> c_int;
    pub fn lf_mkd_detect_extrema(h: *mut lf_mkd, out: *mut lf_mkd_extremum, max_out: u64, n_out: *mut u64,
                                 n_dropped: *mut u64) -> c_int;
    pub fn lf_mkd_filter_extrema_device(h: *mut lf_mkd, d_extrema: *const lf_mkd_extremum, n: u64, top_n: u32,
                                        min_size: f32, d_out: *mut lf_mkd_extremum, d_index: *mut u32,
                                        n_out: *mut u64, stream: *mut c_void) -> c_int;

    pub fn lf_mkd_detect(h: *mut lf_mkd, image: *const f32, width: u32, height: u32, top_n: u32, min_size: f32,
                         keypoints: *mut lf_mkd_keypoint, descriptors: *mut f32, max_out: u64, n_out: *mut u64,
                         dropped_blobs: *mut u64, dropped_features: *mut u64) -> c_int;
    pub fn lf_mkd_detect_u8(h: *mut lf_mkd, image: *const u8, width: u32, height: u32, top_n: u32, min_size: f32,
                            keypoints: *mut lf_mkd_keypoint, descriptors: *mut f32, max_out: u64, n_out: *mut u64,
                            dropped_blobs: *mut u64, dropped_features: *mut u64) -> c_int;
    pub fn lf_mkd_detect_times(h: *mut lf_mkd, upload_ms: *mut f64, pipeline_ms: *mut f64, readback_ms: *mut f64) -> c_int;
    pub fn lf_mkd_detect_frames_device(h: *mut lf_mkd, d_images: *const f32, n_frames: u32, width: u32, height: u32,
                                       top_n: u32, min_size: f32, d_keypoints: *mut lf_mkd_keypoint,
                                       d_frame_of_kp: *mut u32, d_descriptors: *mut f32, max_out: u64,
                                       n_out: *mut u64, dropped_blobs: *mut u64, dropped_features: *mut u64,
                                       stream: *mut c_void) -> c_int;

    pub fn lf_mkd_stream_create(h: *mut lf_mkd, width: u32, height: u32, top_n: u32, min_size: f32, max_out: u64,
                                d_image: *const f32, d_keypoints: *mut lf_mkd_keypoint, d_descriptors: *mut f32,
                                d_counts: *mut u64) -> c_int;
    pub fn lf_mkd_stream_frame(h: *mut lf_mkd, stream: *mut c_void) -> c_int;

    pub fn lf_mkd_match_device(h: *mut lf_mkd, d_a: *const f32, na: u64, d_b: *const f32, nb: u64,
                               d_exclude_lo: *const u32, d_exclude_hi: *const u32, ratio: f32, d_match: *mut i32,
                               d_best: *mut f32, d_second: *mut f32, stream: *mut c_void) -> c_int;
    pub fn lf_mkd_match_both_device(h: *mut lf_mkd, d_a: *const f32, na: u64, d_b: *const f32, nb: u64, ratio: f32,
                                    d_match_ab: *mut i32, d_match_ba: *mut i32, stream: *mut c_void) -> c_int;
    pub fn lf_mkd_match(h: *mut lf_mkd, a: *const f32, na: u64, b: *const f32, nb: u64, ratio: f32,
                        matches: *mut i32) -> c_int;
    pub fn lf_mkd_match_overflowed(h: *mut lf_mkd, stream: *mut c_void, n_rows: *mut u64) -> c_int;

    // the path's one collective: the all-gather of descriptor shards over RCCL (configs[3])
    pub fn lf_mkd_comm_unique_id(id: *mut u8) -> c_int;
    pub fn lf_mkd_comm_create(h: *mut lf_mkd, id: *const u8, n_ranks: i32, rank: i32, out: *mut *mut lf_mkd_comm) -> c_int;
    pub fn lf_mkd_comm_destroy(c: *mut lf_mkd_comm) -> c_int;
    pub fn lf_mkd_comm_info(c: *const lf_mkd_comm, rccl_version: *mut i32, n_ranks: *mut i32, rank: *mut i32) -> c_int;
    pub fn lf_mkd_allgather_descriptors(h: *mut lf_mkd, c: *mut lf_mkd_comm, counts: *const u64, d_buf: *mut f32,
                                        mode: i32, stream: *mut c_void) -> c_int;
    pub fn lf_mkd_plan_upload(width: u32, height: u32, bytes_per_pixel: u32, n_scales: u32, cuts: *mut u32, max_cuts: u32,
                              n_cuts: *mut u32, modelled_us: *mut f64, one_piece_us: *mut f64) -> c_int;
    pub fn lf_mkd_comm_loopback(h: *mut lf_mkd, c: *mut lf_mkd_comm, d_src: *const f32, d_dst: *mut f32, n_rows: u64,
                                stream: *mut c_void) -> c_int;
    pub fn lf_mkd_comm_last_form(c: *const lf_mkd_comm) -> c_int;

    pub fn lf_mkd_get_coarse_layer(h: *mut lf_mkd, layer: u32, out: *mut f32) -> c_int;
    pub fn lf_mkd_sample_patches_device(h: *mut lf_mkd, d_kps: *const lf_mkd_keypoint, n: u64, d_patches: *mut f32,
                                        stream: *mut c_void) -> c_int;
    pub fn lf_mkd_get_pyramid_level(h: *mut lf_mkd, level: u32, out: *mut f32, w: *mut u32, hgt: *mut u32) -> c_int;
    pub fn lf_mkd_get_pyramid_level_apron(h: *mut lf_mkd, level: u32, out: *mut f32, w: *mut u32, hgt: *mut u32,
                                          apron: *mut u32) -> c_int;
    pub fn lf_mkd_build_constants(mean: *const f32, eigvals: *const f32, eigvecs: *const f32,
                                  gradient_angle: *mut f32, embedding_polar: *mut f32,
                                  embedding_cartesian: *mut f32, w_t: *mut f32) -> c_int;
    pub fn lf_mkd_kernel_times(h: *mut lf_mkd, pool_ms: *mut f64, whiten_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn lf_mkd_kernel_clock(h: *mut lf_mkd, stream: *mut c_void, shader_mhz: *mut f64, kernel_ms: *mut f64) -> c_int;
    pub fn lf_mkd_detect_recordings(h: *const lf_mkd, n_recordings: *mut u32, n_banded: *mut u32, n_sightings: *mut u32) -> c_int;
    pub fn lf_mkd_synchronize(h: *mut lf_mkd) -> c_int;
    pub fn lf_mkd_version() -> *const c_char;
}
