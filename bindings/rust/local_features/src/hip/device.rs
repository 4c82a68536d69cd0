//! A device allocation of the HIP runtime, just enough for `LocalFeaturesHip::cross_image_match` to stage host arrays for
//! the device-pointer entry points of liblf_mkd.so.  (`hipMalloc` / `hipMemcpy` / `hipFree` come from libamdhip64, which
//! liblf_mkd.so already depends on; build.rs adds it to the link line.)
//!
//! New file in the reference tree: `local_features/src/hip/device.rs`.
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};

use super::Error;

extern "C" {
    fn hipMalloc(ptr: *mut *mut c_void, bytes: usize) -> c_int;
    fn hipFree(ptr: *mut c_void) -> c_int;
    fn hipMemcpy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: c_int) -> c_int;
}
const HIP_MEMCPY_HOST_TO_DEVICE: c_int = 1;
const HIP_MEMCPY_DEVICE_TO_HOST: c_int = 2;

fn hip(rc: c_int, what: &str) -> Result<(), Error> {
    if rc == 0 { Ok(()) } else { Err(Error::Hip(format!("{what}: hipError {rc}"))) }
}

pub struct DeviceBuffer<T: Copy> {
    ptr: *mut c_void,
    len: usize,
    _t: PhantomData<T>,
}

impl<T: Copy> DeviceBuffer<T> {
    pub fn new(len: usize) -> Result<Self, Error> {
        let mut ptr = std::ptr::null_mut();
        // SAFETY: plain allocation; a zero-length request still gets a valid (unused) pointer
        hip(unsafe { hipMalloc(&mut ptr, std::mem::size_of::<T>() * len.max(1)) }, "hipMalloc")?;
        Ok(Self { ptr, len, _t: PhantomData })
    }

    pub fn from_slice(src: &[T]) -> Result<Self, Error> {
        let b = Self::new(src.len())?;
        // SAFETY: `b` holds src.len() elements
        unsafe { b.upload_at(b.as_mut_ptr(), src)? };
        Ok(b)
    }

    pub fn as_ptr(&self) -> *const T { self.ptr as *const T }
    pub fn as_mut_ptr(&self) -> *mut T { self.ptr as *mut T }

    /// Copies `src` to `dst`, which must point into this buffer with room for `src.len()` elements.
    pub unsafe fn upload_at(&self, dst: *mut T, src: &[T]) -> Result<(), Error> {
        debug_assert!(dst as usize >= self.ptr as usize
                      && (dst as usize - self.ptr as usize) / std::mem::size_of::<T>() + src.len() <= self.len);
        hip(hipMemcpy(dst as *mut c_void, src.as_ptr() as *const c_void, std::mem::size_of_val(src), HIP_MEMCPY_HOST_TO_DEVICE),
            "hipMemcpy")
    }

    pub fn download(&self, dst: &mut [T]) -> Result<(), Error> {
        assert!(dst.len() <= self.len);
        // SAFETY: `dst` is a host slice no longer than the allocation
        hip(unsafe { hipMemcpy(dst.as_mut_ptr() as *mut c_void, self.ptr, std::mem::size_of_val(dst), HIP_MEMCPY_DEVICE_TO_HOST) },
            "hipMemcpy")
    }
}

impl<T: Copy> Drop for DeviceBuffer<T> {
    fn drop(&mut self) {
        // SAFETY: allocated by hipMalloc, freed once
        unsafe { hipFree(self.ptr); }
    }
}
