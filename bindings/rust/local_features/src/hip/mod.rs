//! `LocalFeaturesHip`: the role of `LocalFeaturesVulkan` (vulkan/mod.rs:98-131) on the MI355X path.  Same public
//! methods -- `detect_extract_all`, `detect_top_n`, `detect(img, Option<&mut dyn FilterBlobs>)` (vulkan/mod.rs:346-367),
//! same `FeaturesResult` -- over liblf_mkd.so (include/lf_mkd.h).  No Vulkan object is held or needed.
//!
//! New file in the reference tree: `local_features/src/hip/mod.rs` (with `ffi.rs` beside it, `build.rs` in the crate root
//! and the few lines of `lib.rs.diff`).  The build image of this repository has no Rust toolchain, so this file has not
//! been compiled there; `tests/test_rust_binding.py` checks its FFI surface against the header mechanically.
//!
//! What a caller can observe differently from the Vulkan backend:
//!   * keypoints come in a defined order (by blob, then orientation-histogram bin; the reference appends atomically),
//!     blobs too (raster order of the 4x4x4 scan cubes);
//!   * the sampler's bilinear weights are exact f32 (the Vulkan sampler's precision is implementation-defined), the
//!     pyramid mirrors at the content edge;
//!   * the pooling contraction runs on the matrix cores from f16 hi+lo splits (descriptors within 1e-5 relative L2 of the
//!     f32 formulation; `LF_MKD_POOL_F32` selects exact f32 arithmetic at half the speed).
mod device;
pub mod ffi;

use std::ffi::{CStr, CString};

use ndarray::{s, Array2, ArrayView2};

use device::DeviceBuffer;

use crate::vulkan::{BlobLocationsView, FilterBlobs, FilterBlobsOutput};
use crate::{BuildTimeParams, FeatureDetectParams, FeaturesResult, Keypoint, DESCRIPTOR_LEN, MKDPCA};

/// Blobs per block of the reference's extremum buffer (vulkan/mod.rs:279); a custom `FilterBlobs` sees chunks of this size.
const EXTREMUM_BLOCK_LEN: usize = 256;

#[derive(Debug, thiserror::Error)]
#[non_exhaustive]
pub enum Error {
    #[error("illegal argument: {0}")]
    BadArgument(String),
    #[error("HIP runtime: {0}")]
    Hip(String),
    #[error("cannot read PCA model: {0}")]
    Io(String),
    #[error("no usable gfx950 device: {0}")]
    NoDevice(String),
    #[error("RCCL: {0}")]
    Comm(String),
    #[error("liblf_mkd: status {0}: {1}")]
    Other(i32, String),
}

fn last_error(h: *const ffi::lf_mkd) -> String {
    // SAFETY: lf_mkd_last_error returns a NUL-terminated string owned by the handle (or by the thread, for h = null)
    unsafe { CStr::from_ptr(ffi::lf_mkd_last_error(h)) }.to_string_lossy().into_owned()
}

fn check(h: *const ffi::lf_mkd, rc: i32) -> Result<(), Error> {
    match rc {
        ffi::LF_MKD_OK => Ok(()),
        ffi::LF_MKD_ERR_BAD_ARG | ffi::LF_MKD_ERR_NO_IMAGE => Err(Error::BadArgument(last_error(h))),
        ffi::LF_MKD_ERR_HIP => Err(Error::Hip(last_error(h))),
        ffi::LF_MKD_ERR_IO => Err(Error::Io(last_error(h))),
        ffi::LF_MKD_ERR_NO_DEVICE => Err(Error::NoDevice(last_error(h))),
        ffi::LF_MKD_ERR_COMM => Err(Error::Comm(last_error(h))),
        other => Err(Error::Other(other, last_error(h))),
    }
}

// `Keypoint` (lib.rs:17-24) and `lf_mkd_keypoint` are both five f32 in the same order.  `Keypoint` is not `#[repr(C)]`
// in the reference; lib.rs.diff adds the attribute, and these assertions keep the two in step.
const _: () = assert!(std::mem::size_of::<Keypoint>() == std::mem::size_of::<ffi::lf_mkd_keypoint>());
const _: () = assert!(std::mem::align_of::<Keypoint>() == std::mem::align_of::<ffi::lf_mkd_keypoint>());

pub struct LocalFeaturesHip {
    h: *mut ffi::lf_mkd,
    fixed_params: BuildTimeParams,
}

// One handle per thread at a time (`&mut self` on every call, as LocalFeaturesVulkan); it may move between threads.
unsafe impl Send for LocalFeaturesHip {}

impl LocalFeaturesHip {
    /// `LocalFeaturesVulkan::new` + `upload_constant_data` (vulkan/mod.rs:253-323,1587-1713).  The PCA tensors are the
    /// ones `PCAModel::from_safetensors` yields from the embedded model bytes (mkd_ref.rs:26-31,352-391).
    pub fn new(fixed_params: BuildTimeParams, params: FeatureDetectParams, device: i32) -> Result<Self, Error> {
        let bytes = match fixed_params.pca {
            MKDPCA::LIBERTY => crate::mkd_ref::PCA_SAFETENSORS_LIBERTY,
            MKDPCA::NOTREDAME => crate::mkd_ref::PCA_SAFETENSORS_NOTREDAME,
            MKDPCA::YOSEMITE => crate::mkd_ref::PCA_SAFETENSORS_YOSEMITE,
        };
        let pca = crate::mkd_ref::PCAModel::from_safetensors(bytes).map_err(|e| Error::Io(e.to_string()))?;
        let mean = pca.mean.as_standard_layout();
        let eigvals = pca.eigvals.as_standard_layout();
        let eigvecs = pca.eigvecs.as_standard_layout(); // [238][238] row-major, as lf_mkd_create expects
        let p = ffi::lf_mkd_params {
            max_image_width: fixed_params.max_image_width,
            max_image_height: fixed_params.max_image_height,
            max_features: fixed_params.max_features,
            patch_scale_factor: params.patch_scale_factor,
            device,
            angle_mode: ffi::LF_MKD_ANGLE_SHADER,
            pool_mode: ffi::LF_MKD_POOL_DEFAULT,
            n_scales: fixed_params.n_scales,
            max_blobs: fixed_params.max_blobs,
            ..Default::default()
        };
        let mut h: *mut ffi::lf_mkd = std::ptr::null_mut();
        // SAFETY: the three arrays hold 238, 238 and 238*238 contiguous f32 and outlive the call; `h` receives the handle
        let rc = unsafe {
            ffi::lf_mkd_create(&p, mean.as_ptr(), eigvals.as_ptr(), eigvecs.as_ptr(), &mut h)
        };
        check(std::ptr::null(), rc)?;
        Ok(Self { h, fixed_params })
    }

    /// Same, reading `concat-pca-*.safetensors` from a directory (for a build that does not embed the models).
    pub fn from_model_dir(fixed_params: BuildTimeParams, params: FeatureDetectParams, device: i32, dir: &str)
        -> Result<Self, Error> {
        let name = match fixed_params.pca {
            MKDPCA::LIBERTY => "liberty",
            MKDPCA::NOTREDAME => "notredame",
            MKDPCA::YOSEMITE => "yosemite",
        };
        let path = CString::new(format!("{dir}/concat-pca-{name}.safetensors")).map_err(|e| Error::Io(e.to_string()))?;
        let p = ffi::lf_mkd_params {
            max_image_width: fixed_params.max_image_width,
            max_image_height: fixed_params.max_image_height,
            max_features: fixed_params.max_features,
            patch_scale_factor: params.patch_scale_factor,
            device,
            n_scales: fixed_params.n_scales,
            max_blobs: fixed_params.max_blobs,
            ..Default::default()
        };
        let mut h: *mut ffi::lf_mkd = std::ptr::null_mut();
        // SAFETY: `path` is NUL-terminated and outlives the call
        let rc = unsafe { ffi::lf_mkd_create_from_file(&p, path.as_ptr(), &mut h) };
        check(std::ptr::null(), rc)?;
        Ok(Self { h, fixed_params })
    }

    /// vulkan/mod.rs:346-351: every blob the detector finds (at most `max_blobs`).
    pub fn detect_extract_all(&mut self, img: &ArrayView2<f32>) -> Result<FeaturesResult, Error> {
        self.detect_on_device(img, 0, 0.0)
    }

    /// vulkan/mod.rs:353-361: the `n` blobs of largest contrast among those of size >= `min_size`
    /// (`TopKContrastFilter`, vulkan/mod.rs:1753-1786, run on the device).
    pub fn detect_top_n(&mut self, img: &ArrayView2<f32>, n: u32, min_size: f32) -> Result<FeaturesResult, Error> {
        self.detect_on_device(img, n, min_size)
    }

    /// `detect_top_n` on the 8-bit luma the reference's callers convert from (`image::open(..).grayscale()` then `convert()`,
    /// examples/match_images/src/main.rs:44-60; `u8 as f32 / 255.`, examples/webcam/src/main.rs:136): a quarter of the
    /// upload, the same keypoints and descriptors bit for bit (the device divides by 255 with a correctly rounded division).
    pub fn detect_top_n_u8(&mut self, img: &ArrayView2<u8>, n: u32, min_size: f32) -> Result<FeaturesResult, Error> {
        let width: u32 = img.ncols().try_into().map_err(|_| Error::BadArgument("image too wide".into()))?;
        let height: u32 = img.nrows().try_into().map_err(|_| Error::BadArgument("image too tall".into()))?;
        let data = img.as_slice().expect("image must be contiguous row-major"); // vulkan/mod.rs:368
        let cap = self.fixed_params.max_features as usize;
        let mut keypoints = vec![ffi::lf_mkd_keypoint::default(); cap];
        let mut descriptors = Array2::<f32>::zeros((cap, DESCRIPTOR_LEN));
        let (mut m, mut dropped_blobs, mut dropped_features) = (0u64, 0u64, 0u64);
        // SAFETY: `data` holds width*height bytes; the two outputs have room for `cap` rows
        unsafe {
            check(self.h, ffi::lf_mkd_detect_u8(self.h, data.as_ptr(), width, height, n, min_size, keypoints.as_mut_ptr(),
                                                descriptors.as_mut_ptr(), cap as u64, &mut m, &mut dropped_blobs,
                                                &mut dropped_features))?;
        }
        keypoints.truncate(m as usize);
        Ok(FeaturesResult {
            keypoints: keypoints.into_iter().map(to_keypoint).collect(),
            descriptors: descriptors.slice_move(s![..m as usize, ..]),
            dropped_blobs: dropped_blobs as u32,
            dropped_features: dropped_features as u32,
        })
    }

    /// vulkan/mod.rs:363-593 with a caller-supplied blob filter: detect graph, the filter on the host (where the
    /// reference calls it, vulkan/mod.rs:596-691), extract graph on the blobs it kept.  `None` keeps every blob.
    pub fn detect(&mut self, img: &ArrayView2<f32>, filter_keypoints: Option<&mut dyn FilterBlobs>)
        -> Result<FeaturesResult, Error> {
        let Some(filter) = filter_keypoints else {
            return self.detect_extract_all(img);
        };
        let (width, height) = Self::image_dims(img)?;
        let data = img.as_slice().expect("image must be contiguous row-major"); // vulkan/mod.rs:368
        let max_extrema = 256 * ((self.fixed_params.max_blobs as usize + 255) / 256); // vulkan/mod.rs:279-286
        let mut blobs = vec![ffi::lf_mkd_extremum::default(); max_extrema];
        let (mut n_blobs, mut dropped_blobs) = (0u64, 0u64);
        // SAFETY: `data` holds width*height f32; `blobs` has room for max_extrema records
        unsafe {
            check(self.h, ffi::lf_mkd_set_image(self.h, data.as_ptr(), width, height))?;
            check(self.h, ffi::lf_mkd_detect_extrema(self.h, blobs.as_mut_ptr(), max_extrema as u64, &mut n_blobs,
                                                     &mut dropped_blobs))?;
        }
        blobs.truncate(n_blobs as usize);

        // the filter sees the blobs as the reference hands them over: structure of arrays, in blocks of 256
        let xs: Vec<f32> = blobs.iter().map(|b| b.x).collect();
        let ys: Vec<f32> = blobs.iter().map(|b| b.y).collect();
        let scales: Vec<f32> = blobs.iter().map(|b| b.size).collect();
        let contrasts: Vec<f32> = blobs.iter().map(|b| b.response).collect();
        let mut indices: Vec<u32> = Vec::new();
        {
            let views = (0..blobs.len()).step_by(EXTREMUM_BLOCK_LEN).map(|lo| {
                let hi = (lo + EXTREMUM_BLOCK_LEN).min(blobs.len());
                BlobLocationsView { xs: &xs[lo..hi], ys: &ys[lo..hi], scales: &scales[lo..hi], contrasts: &contrasts[lo..hi] }
            });
            filter.filter(Box::new(views), FilterBlobsOutput { indices: &mut indices });
        }
        let mut kept = Vec::with_capacity(indices.len());
        for i in indices {
            let blob = blobs.get(i as usize)
                .ok_or_else(|| Error::BadArgument(format!("FilterBlobs returned index {i} of {} blobs", blobs.len())))?;
            kept.push(*blob);
        }

        let cap = self.fixed_params.max_features as usize;
        let mut keypoints = vec![ffi::lf_mkd_keypoint::default(); cap];
        let (mut n_kp, mut dropped_features) = (0u64, 0u64);
        // SAFETY: `kept` holds kept.len() records, `keypoints` has room for `cap`
        unsafe {
            check(self.h, ffi::lf_mkd_orient_keypoints(self.h, kept.as_ptr(), kept.len() as u64, keypoints.as_mut_ptr(),
                                                       cap as u64, &mut n_kp, &mut dropped_features))?;
        }
        keypoints.truncate(n_kp as usize);
        let mut descriptors = Array2::<f32>::zeros((keypoints.len(), DESCRIPTOR_LEN));
        // SAFETY: `descriptors` is a fresh standard-layout [n_kp][128] array
        unsafe {
            check(self.h, ffi::lf_mkd_describe_keypoints(self.h, keypoints.as_ptr(), n_kp, descriptors.as_mut_ptr()))?;
        }
        Ok(FeaturesResult {
            keypoints: keypoints.into_iter().map(to_keypoint).collect(),
            descriptors,
            dropped_blobs: dropped_blobs as u32,
            dropped_features: dropped_features as u32,
        })
    }

    /// The describe half on its own (the extract graph after orientation, vulkan/mod.rs:1277-1572): descriptors of the
    /// given keypoints on `img`.  The reference has no public call for this; the path's benchmarks use it.
    pub fn describe(&mut self, img: &ArrayView2<f32>, keypoints: &[Keypoint]) -> Result<Array2<f32>, Error> {
        let (width, height) = Self::image_dims(img)?;
        let data = img.as_slice().expect("image must be contiguous row-major");
        let kps: Vec<ffi::lf_mkd_keypoint> = keypoints.iter().map(from_keypoint).collect();
        let mut out = Array2::<f32>::zeros((kps.len(), DESCRIPTOR_LEN));
        // SAFETY: sizes as declared; `out` is standard layout
        unsafe {
            check(self.h, ffi::lf_mkd_set_image(self.h, data.as_ptr(), width, height))?;
            check(self.h, ffi::lf_mkd_describe_keypoints(self.h, kps.as_ptr(), kps.len() as u64, out.as_mut_ptr()))?;
        }
        Ok(out)
    }

    /// `match_features` of examples/match_images/src/main.rs:8-27 (dot-product similarity, Lowe's ratio 0.8).
    pub fn match_features(&mut self, a: &ArrayView2<f32>, b: &ArrayView2<f32>) -> Result<Vec<(usize, usize)>, Error> {
        assert_eq!(a.ncols(), DESCRIPTOR_LEN);
        assert_eq!(b.ncols(), DESCRIPTOR_LEN);
        let (a, b) = (a.as_standard_layout(), b.as_standard_layout());
        let mut m = vec![-1i32; a.nrows()];
        // SAFETY: a, b are contiguous [n][128]; `m` has a.nrows() entries
        unsafe {
            check(self.h, ffi::lf_mkd_match(self.h, a.as_ptr(), a.nrows() as u64, b.as_ptr(), b.nrows() as u64, 0.8,
                                            m.as_mut_ptr()))?;
        }
        Ok(m.iter().enumerate().filter(|(_, j)| **j >= 0).map(|(i, j)| (i, *j as usize)).collect())
    }

    /// The match stage of a job sharded by image over the GPUs of a node (one process and one `LocalFeaturesHip` per GPU;
    /// BASELINE configs[3]): `desc_local` holds this rank's descriptors, image after image (`image_sizes` rows each);
    /// the shards are all-gathered over RCCL -- the path's one collective -- and every local descriptor is matched against
    /// the descriptors of all OTHER images (`match_features`' rule, ratio 0.8).  Returns (local row, global row) pairs;
    /// global rows count through the ranks' shards in rank order.  `comm`: see `HipComm::new`.
    pub fn cross_image_match(&mut self, comm: &HipComm, desc_local: &ArrayView2<f32>, image_sizes: &[usize])
        -> Result<Vec<(usize, usize)>, Error>
    {
        assert_eq!(desc_local.ncols(), DESCRIPTOR_LEN);
        assert_eq!(image_sizes.iter().sum::<usize>(), desc_local.nrows());
        let local = desc_local.as_standard_layout();
        let counts = comm.shard_counts(local.nrows() as u64)?;           // one u64 per rank, exchanged by the application
        let base: u64 = counts[..comm.rank as usize].iter().sum();
        let total: u64 = counts.iter().sum();
        let (mut lo, mut hi) = (Vec::with_capacity(local.nrows()), Vec::with_capacity(local.nrows()));
        let mut start = base as u32;
        for &n in image_sizes {
            for _ in 0..n { lo.push(start); hi.push(start + n as u32); }
            start += n as u32;
        }
        let mut matches = vec![-1i32; local.nrows()];
        // SAFETY: device buffers sized as declared; every rank makes the same collective call
        unsafe {
            let gathered = DeviceBuffer::<f32>::new(total as usize * DESCRIPTOR_LEN)?;
            let own = gathered.as_mut_ptr().add(base as usize * DESCRIPTOR_LEN);
            gathered.upload_at(own, local.as_slice().unwrap())?;
            check(self.h, ffi::lf_mkd_allgather_descriptors(self.h, comm.c, counts.as_ptr(), gathered.as_mut_ptr(),
                                                            ffi::LF_MKD_GATHER_DIRECT, std::ptr::null_mut()))?;
            let (d_lo, d_hi) = (DeviceBuffer::from_slice(&lo)?, DeviceBuffer::from_slice(&hi)?);
            let d_match = DeviceBuffer::<i32>::new(local.nrows())?;
            check(self.h, ffi::lf_mkd_match_device(self.h, own, local.nrows() as u64, gathered.as_mut_ptr(), total,
                                                   d_lo.as_ptr(), d_hi.as_ptr(), 0.8, d_match.as_mut_ptr(),
                                                   std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut()))?;
            check(self.h, ffi::lf_mkd_synchronize(self.h))?;
            d_match.download(&mut matches)?;
        }
        Ok(matches.iter().enumerate().filter(|(_, j)| **j >= 0).map(|(i, j)| (i, *j as usize)).collect())
    }

    fn image_dims(img: &ArrayView2<f32>) -> Result<(u32, u32), Error> {
        let width: u32 = img.ncols().try_into().map_err(|_| Error::BadArgument("image too wide".into()))?;
        let height: u32 = img.nrows().try_into().map_err(|_| Error::BadArgument("image too tall".into()))?;
        Ok((width, height))
    }

    // detect / detect_top_n in one library call: image -> pyramid + a-trous stack -> blobs -> [top n] -> orientation ->
    // sampling -> descriptors, every stage on the device
    fn detect_on_device(&mut self, img: &ArrayView2<f32>, top_n: u32, min_size: f32) -> Result<FeaturesResult, Error> {
        let (width, height) = Self::image_dims(img)?;
        let data = img.as_slice().expect("image must be contiguous row-major"); // vulkan/mod.rs:368
        let cap = self.fixed_params.max_features as usize;
        let mut keypoints = vec![ffi::lf_mkd_keypoint::default(); cap];
        let mut descriptors = Array2::<f32>::zeros((cap, DESCRIPTOR_LEN));
        let (mut n, mut dropped_blobs, mut dropped_features) = (0u64, 0u64, 0u64);
        // SAFETY: `data` holds width*height f32; the two outputs have room for `cap` rows
        unsafe {
            check(self.h, ffi::lf_mkd_detect(self.h, data.as_ptr(), width, height, top_n, min_size, keypoints.as_mut_ptr(),
                                             descriptors.as_mut_ptr(), cap as u64, &mut n, &mut dropped_blobs,
                                             &mut dropped_features))?;
        }
        keypoints.truncate(n as usize);
        Ok(FeaturesResult {
            keypoints: keypoints.into_iter().map(to_keypoint).collect(),
            descriptors: descriptors.slice_move(s![..n as usize, ..]),
            dropped_blobs: dropped_blobs as u32,
            dropped_features: dropped_features as u32,
        })
    }
}

/// One rank's RCCL communicator for `cross_image_match`.  Rank 0 draws the identifier (`HipComm::unique_id`) and hands its
/// 128 bytes to the other ranks by whatever channel the application has; every rank then calls `HipComm::new` with it.
pub struct HipComm {
    c: *mut ffi::lf_mkd_comm,
    pub n_ranks: i32,
    pub rank: i32,
    /// how the application exchanges one u64 per rank (the shard sizes): e.g. over the channel that carried the identifier
    exchange_counts: Box<dyn Fn(u64) -> Vec<u64>>,
}

impl HipComm {
    pub fn unique_id() -> Result<[u8; ffi::LF_MKD_COMM_ID_BYTES], Error> {
        let mut id = [0u8; ffi::LF_MKD_COMM_ID_BYTES];
        // SAFETY: `id` has LF_MKD_COMM_ID_BYTES bytes
        check(std::ptr::null(), unsafe { ffi::lf_mkd_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }

    pub fn new(lf: &mut LocalFeaturesHip, id: &[u8; ffi::LF_MKD_COMM_ID_BYTES], n_ranks: i32, rank: i32,
               exchange_counts: Box<dyn Fn(u64) -> Vec<u64>>) -> Result<Self, Error> {
        let mut c = std::ptr::null_mut();
        // SAFETY: collective over the ranks that share `id`
        check(lf.h, unsafe { ffi::lf_mkd_comm_create(lf.h, id.as_ptr(), n_ranks, rank, &mut c) })?;
        Ok(Self { c, n_ranks, rank, exchange_counts })
    }

    fn shard_counts(&self, mine: u64) -> Result<Vec<u64>, Error> {
        let counts = (self.exchange_counts)(mine);
        if counts.len() != self.n_ranks as usize || counts[self.rank as usize] != mine {
            return Err(Error::BadArgument("exchange_counts must return one entry per rank, this rank's own included".into()));
        }
        Ok(counts)
    }

    /// One group of ncclSend + ncclRecv from this rank to itself (`lf_mkd_comm_loopback`): `rows` descriptors go out of one
    /// device buffer and must arrive in another.  What a job runs once per rank before its first `cross_image_match`, so that
    /// the point-to-point branch of the gather has met the machine's librccl before the collective depends on it.
    pub fn self_test(&self, lf: &mut LocalFeaturesHip, rows: usize) -> Result<(), Error> {
        let src: Vec<f32> = (0..rows * DESCRIPTOR_LEN).map(|i| (i % 8191) as f32).collect();
        let mut back = vec![0f32; src.len()];
        // SAFETY: two distinct device buffers of `rows` rows each
        unsafe {
            let (d_src, d_dst) = (DeviceBuffer::from_slice(&src)?, DeviceBuffer::<f32>::new(src.len().max(1))?);
            check(lf.h, ffi::lf_mkd_comm_loopback(lf.h, self.c, d_src.as_ptr(), d_dst.as_mut_ptr(), rows as u64,
                                                  std::ptr::null_mut()))?;
            check(lf.h, ffi::lf_mkd_synchronize(lf.h))?;
            if rows > 0 { d_dst.download(&mut back)?; }
        }
        if back != src { return Err(Error::BadArgument("RCCL loopback: the rows that arrived are not the rows sent".into())); }
        Ok(())
    }

    /// `LF_MKD_GATHER_DIRECT` / `LF_MKD_GATHER_RING` as the latest gather ran, -1 before the first
    pub fn last_form(&self) -> i32 {
        // SAFETY: plain query
        unsafe { ffi::lf_mkd_comm_last_form(self.c) }
    }

    /// (RCCL version code, ranks, this rank)
    pub fn info(&self) -> Result<(i32, i32, i32), Error> {
        let (mut v, mut n, mut r) = (0, 0, 0);
        // SAFETY: plain out-pointers
        check(std::ptr::null(), unsafe { ffi::lf_mkd_comm_info(self.c, &mut v, &mut n, &mut r) })?;
        Ok((v, n, r))
    }
}

impl Drop for HipComm {
    fn drop(&mut self) {
        // SAFETY: `c` came from lf_mkd_comm_create and is destroyed exactly once
        unsafe { ffi::lf_mkd_comm_destroy(self.c); }
    }
}

impl Drop for LocalFeaturesHip {
    fn drop(&mut self) {
        // SAFETY: `h` came from lf_mkd_create* and is destroyed exactly once
        unsafe { ffi::lf_mkd_destroy(self.h) }
    }
}

fn to_keypoint(k: ffi::lf_mkd_keypoint) -> Keypoint {
    Keypoint { x: k.x, y: k.y, size: k.size, angle: k.angle, response: k.response }
}

fn from_keypoint(k: &Keypoint) -> ffi::lf_mkd_keypoint {
    ffi::lf_mkd_keypoint { x: k.x, y: k.y, size: k.size, angle: k.angle, response: k.response }
}
