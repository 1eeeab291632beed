//! MI355X backend of `jpeg_decoder::worker::Worker` (new file `src/worker/hip.rs` of image-rs/jpeg-decoder v0.3.2).
//!
//! Planes live in HBM between `start` / `append_row` / `get_result` and `compute_image`: everything behind the
//! `Worker` boundary (dequantization + IDCT, upsampling, colour conversion) runs in `libjpgpu.so`
//! (C ABI: `include/jpgpu.h` of the jpeg-decoder_amd repository).  The crate-side changes that select this worker
//! are in `rust/worker_mod.patch` and `rust/decoder.patch`; `rust/build.rs` finds the library.
//!
//! NOT compiled in the image this repository is built in (no Rust toolchain there): the call sequence below is the
//! one `tests/capi/worker_roundtrip.c` performs from C against the same library, bit-exact against the scalar path.
#![allow(unsafe_code)] // like src/arch/mod.rs:1
use super::{RowData, Worker};
use crate::decoder::ColorTransform;
use crate::error::{Error, Result, UnsupportedFeature};
use crate::parser::{Component, Dimensions};
use alloc::boxed::Box;
use alloc::string::String;
use alloc::vec::Vec;
use core::ffi::c_void;
use std::os::raw::{c_char, c_int}; // (core::ffi::{c_char, c_int} need Rust 1.64; the crate's rust-version is 1.61)

/// `jpgpu_component` (include/jpgpu.h) == `parser::Component` (src/parser.rs:76-89), field for field.
#[repr(C)]
struct JpgpuComponent {
    identifier: u8,
    horizontal_sampling_factor: u8,
    vertical_sampling_factor: u8,
    quantization_table_index: u8,
    dct_scale: u32,
    size_width: u16,
    size_height: u16,
    block_width: u16,
    block_height: u16,
}

impl From<&Component> for JpgpuComponent {
    fn from(c: &Component) -> Self {
        JpgpuComponent {
            identifier: c.identifier,
            horizontal_sampling_factor: c.horizontal_sampling_factor,
            vertical_sampling_factor: c.vertical_sampling_factor,
            quantization_table_index: c.quantization_table_index as u8,
            dct_scale: c.dct_scale as u32,
            size_width: c.size.width,
            size_height: c.size.height,
            block_width: c.block_size.width,
            block_height: c.block_size.height,
        }
    }
}

/// Status codes of include/jpgpu.h:28-36.
const JPGPU_OK: c_int = 0;
const JPGPU_ERR_FORMAT: c_int = 1;
const JPGPU_ERR_UNSUPPORTED: c_int = 2;
const JPGPU_ERR_IO: c_int = 3;
const JPGPU_ERR_INTERNAL: c_int = 4;
const JPGPU_ERR_NO_DEVICE: c_int = 5;

#[link(name = "jpgpu")]
extern "C" {
    fn jpgpu_worker_create(device: c_int, out: *mut *mut c_void) -> c_int;
    fn jpgpu_worker_destroy(w: *mut c_void);
    fn jpgpu_worker_last_error(w: *const c_void) -> *const c_char;
    fn jpgpu_worker_start(w: *mut c_void, index: u32, c: *const JpgpuComponent, qt: *const u16) -> c_int;
    fn jpgpu_worker_append_row(w: *mut c_void, index: u32, coefs: *const i16, len: usize) -> c_int;
    fn jpgpu_worker_finish_plane(w: *mut c_void, index: u32, plane_slot: u32) -> c_int;
    fn jpgpu_compute_image(
        w: *mut c_void,
        comps: *const JpgpuComponent,
        ncomp: u32,
        host_planes: *const *const u8,
        out_w: u16,
        out_h: u16,
        color_transform: c_int,
        dst: *mut u8,
        cap: usize,
        len: *mut usize,
    ) -> c_int;
}

/// `ColorTransform` in the order of src/decoder.rs:76-98, which include/jpgpu.h copies (the enum is
/// `#[non_exhaustive]` without explicit discriminants, so the mapping is spelled out).
fn color_transform_id(ct: ColorTransform) -> c_int {
    match ct {
        ColorTransform::None => 0,
        ColorTransform::Unknown => 1,
        ColorTransform::Grayscale => 2,
        ColorTransform::RGB => 3,
        ColorTransform::YCbCr => 4,
        ColorTransform::CMYK => 5,
        ColorTransform::YCCK => 6,
        ColorTransform::JcsBgYcc => 7,
        ColorTransform::JcsBgRgb => 8,
    }
}

pub struct HipWorker {
    handle: *mut c_void,
    /// `decode_scan` numbers the components of a scan locally (src/decoder.rs:848-852); the frame slot a scan-local
    /// index belongs to is told through `map_index` before `get_result`.
    slot_of_index: [u32; 4],
    /// colour transform of the image in flight, for `Error::Unsupported(ColorTransform(..))`
    last_color_transform: ColorTransform,
}

// The handle is only ever used through `&mut self`.
unsafe impl Send for HipWorker {}

impl HipWorker {
    pub fn new() -> Result<Self> {
        let device = std::env::var("JPGPU_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut handle = core::ptr::null_mut();
        let status = unsafe { jpgpu_worker_create(device, &mut handle) };
        if status != JPGPU_OK {
            return Err(error_from(status, String::from("jpgpu_worker_create: no usable MI355X"), ColorTransform::Unknown));
        }
        Ok(HipWorker { handle, slot_of_index: [0, 1, 2, 3], last_color_transform: ColorTransform::Unknown })
    }

    fn check(&self, status: c_int) -> Result<()> {
        if status == JPGPU_OK {
            return Ok(());
        }
        let msg = unsafe { std::ffi::CStr::from_ptr(jpgpu_worker_last_error(self.handle)) }.to_string_lossy().into_owned();
        Err(error_from(status, msg, self.last_color_transform))
    }
}

/// include/jpgpu.h:28-36 promises status 1..4 <-> `Error::{Format, Unsupported, Io, Internal}` (src/error.rs:36-48), one
/// to one; 5 (no device) is an I/O condition.  `Unsupported` carries an enum in the crate: the two conditions this
/// backend can meet are told apart by the library's message ("NonIntegerSubsamplingRatio", src/upsampler.rs:97-99, and
/// "ColorTransform(n)", src/decoder.rs:1391-1399).
fn error_from(status: c_int, msg: String, ct: ColorTransform) -> Error {
    match status {
        JPGPU_ERR_FORMAT => Error::Format(msg),
        JPGPU_ERR_UNSUPPORTED => {
            if msg.contains("NonIntegerSubsamplingRatio") {
                Error::Unsupported(UnsupportedFeature::NonIntegerSubsamplingRatio)
            } else if msg.contains("ColorTransform") {
                Error::Unsupported(UnsupportedFeature::ColorTransform(ct))
            } else {
                Error::Unsupported(UnsupportedFeature::SubsamplingRatio)
            }
        }
        JPGPU_ERR_IO | JPGPU_ERR_NO_DEVICE => Error::Io(std::io::Error::new(std::io::ErrorKind::Other, msg)),
        JPGPU_ERR_INTERNAL => Error::Internal(Box::<dyn std::error::Error + Send + Sync>::from(msg)),
        _ => Error::Internal(Box::<dyn std::error::Error + Send + Sync>::from(msg)),
    }
}

impl Drop for HipWorker {
    fn drop(&mut self) {
        unsafe { jpgpu_worker_destroy(self.handle) }
    }
}

impl Worker for HipWorker {
    fn start(&mut self, data: RowData) -> Result<()> {
        let c = JpgpuComponent::from(&data.component);
        self.check(unsafe { jpgpu_worker_start(self.handle, data.index as u32, &c, data.quantization_table.as_ptr()) })
    }

    /// The library copies the row into pinned staging memory before it returns: the `Vec` may be dropped.
    fn append_row(&mut self, (index, data): (usize, Vec<i16>)) -> Result<()> {
        self.check(unsafe { jpgpu_worker_append_row(self.handle, index as u32, data.as_ptr(), data.len()) })
    }

    /// The plane stays in HBM; the decoder only tests `!plane.is_empty()` (src/decoder.rs:465-475, 1306-1308), so a
    /// one-byte placeholder stands for it.
    fn get_result(&mut self, index: usize) -> Result<Vec<u8>> {
        self.check(unsafe { jpgpu_worker_finish_plane(self.handle, index as u32, self.slot_of_index[index]) })?;
        Ok(alloc::vec![0u8; 1])
    }

    fn map_index(&mut self, scan_index: usize, frame_index: usize) {
        self.slot_of_index[scan_index] = frame_index as u32;
    }

    fn compute_image(&mut self, components: &[Component], output_size: Dimensions, ct: ColorTransform) -> Option<Result<Vec<u8>>> {
        self.last_color_transform = ct;
        let comps: Vec<JpgpuComponent> = components.iter().map(Into::into).collect();
        // one component: compute_image's stride compaction, size of the component (src/decoder.rs:1310-1332)
        let n = if comps.len() == 1 {
            comps[0].size_width as usize * comps[0].size_height as usize
        } else {
            output_size.width as usize * output_size.height as usize * comps.len()
        };
        let mut image = alloc::vec![0u8; n];
        let mut len = 0usize;
        let status = unsafe {
            jpgpu_compute_image(
                self.handle,
                comps.as_ptr(),
                comps.len() as u32,
                core::ptr::null(), // planes are the ones finish_plane left on the device
                output_size.width,
                output_size.height,
                color_transform_id(ct),
                image.as_mut_ptr(),
                n,
                &mut len,
            )
        };
        Some(self.check(status).map(|()| {
            image.truncate(len);
            image
        }))
    }
}
