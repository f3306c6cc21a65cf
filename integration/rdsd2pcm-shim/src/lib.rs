//! `rdsd2pcm` over the MI355X engine: same names, argument order and error texts as the call sites in
//! dsd2dxd (`src/main.rs:27-31,275,325-345,361-398,429`; `src/bin/dsd_levels/main.rs:153,214-223,252`).
//! Everything forwards to `libdsd2dxd_amd.so` through `include/rdsd2pcm_c.h`.
//!
//! SOURCE ONLY — never compiled in this repository (no Rust toolchain there).

use std::error::Error;
use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::path::{Path, PathBuf};
use std::sync::atomic::{AtomicBool, Ordering};
use std::sync::mpsc::Sender;

pub const ONE_HUNDRED_PERCENT: f32 = 100.0; // src/main.rs:417-418

#[derive(Clone, Copy, Debug)]
pub struct ProgressUpdate {
    pub percent: f32,
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum DitherType { TPDF, Rectangular, FPD, None, /* extension: */ NoiseShaped } // src/main.rs:172-175
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum FmtType { Interleaved, Planar } // src/main.rs:185-186
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Endianness { LsbFirst, MsbFirst } // src/main.rs:194-196
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum FilterType { Equiripple, XLD, Dsd2Pcm, Chebyshev } // src/main.rs:200-204
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum OutputType { Stdout, Aiff, Aifc, Wav, Flac } // src/main.rs:208-213

impl DitherType { fn code(self) -> u32 { match self { Self::TPDF => b'T', Self::Rectangular => b'R', Self::FPD => b'F', Self::None => b'X', Self::NoiseShaped => b'N' } as u32 } }
impl FmtType { fn code(self) -> u32 { match self { Self::Interleaved => b'I', Self::Planar => b'P' } as u32 } }
impl Endianness { fn code(self) -> u32 { match self { Self::LsbFirst => b'L', Self::MsbFirst => b'M' } as u32 } }
impl FilterType { fn code(self) -> u32 { match self { Self::Equiripple => b'E', Self::XLD => b'X', Self::Dsd2Pcm => b'D', Self::Chebyshev => b'C' } as u32 } }
impl OutputType { fn code(self) -> u32 { match self { Self::Stdout => b'S', Self::Aiff => b'A', Self::Aifc => b'C', Self::Wav => b'W', Self::Flac => b'F' } as u32 } }

/// The DSD rate as a multiple of 2.8224 MHz.  The call sites write `cli.input_rate.try_into()?`
/// (src/main.rs:334,385; src/bin/dsd_levels/main.rs:221,280) inside functions that return
/// `Result<_, Box<dyn Error>>` and `Result<_, String>`: the parameter type is concrete and its
/// `TryFrom<u32>::Error` is `String`, which converts into both.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct DsdRate(u32);
impl TryFrom<u32> for DsdRate {
    type Error = String;
    fn try_from(v: u32) -> Result<Self, String> {
        match v { 1 | 2 | 4 | 8 => Ok(Self(v)), _ => Err("Unsupported DSD input rate; must be 1, 2, 4 or 8".to_string()) }
    }
}

/// `DsdFileFormat::from(&path).is_container()` (src/main.rs:361)
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum DsdFileFormat { Dsf, Dff, Raw }
impl<P: AsRef<Path>> From<P> for DsdFileFormat {
    fn from(p: P) -> Self {
        match p.as_ref().extension().and_then(|e| e.to_str()).map(|e| e.to_ascii_lowercase()).as_deref() {
            Some("dsf") => Self::Dsf,
            Some("dff") => Self::Dff,
            _ => Self::Raw,
        }
    }
}
impl DsdFileFormat { pub fn is_container(&self) -> bool { matches!(self, Self::Dsf | Self::Dff) } }

/// Extension list of the inputs (`FormatExtensions`, src/main.rs:29)
pub struct FormatExtensions;
impl FormatExtensions { pub const ALL: [&'static str; 3] = ["dsf", "dff", "dsd"]; }

mod ffi {
    use super::*;
    #[repr(C)] pub struct Conv { _private: [u8; 0] }
    pub type ProgressFn = unsafe extern "C" fn(user: *mut c_void, percent: f32);
    pub type PathFn = unsafe extern "C" fn(user: *mut c_void, path: *const c_char);
    extern "C" {
        pub fn d2dh_new(bit_depth: u32, output: u32, level_db: f64, output_rate: u32, out_dir: *const c_char, dither: u32, fmt: u32,
                        endian: u32, dsd_rate: u32, block_size: u32, channels: u32, filter: u32, append_rate: c_int,
                        base_dir: *const c_char, in_path: *const c_char, out: *mut *mut Conv) -> c_int;
        pub fn d2dh_from_container(bit_depth: u32, output: u32, level_db: f64, output_rate: u32, out_dir: *const c_char, dither: u32,
                                   filter: u32, append_rate: c_int, base_dir: *const c_char, path: *const c_char, out: *mut *mut Conv) -> c_int;
        pub fn d2dh_new_level_check(output_rate: u32, path: *const c_char, fmt: u32, endian: u32, channels: u32, block_size: u32,
                                    input_rate: u32, out: *mut *mut Conv) -> c_int;
        pub fn d2dh_free(c: *mut Conv);
        pub fn d2dh_do_conversion(c: *mut Conv, cancel: *const c_int, progress: Option<ProgressFn>, user: *mut c_void) -> c_int;
        pub fn d2dh_check_level(c: *mut Conv, cancel: *const c_int, progress: Option<ProgressFn>, user: *mut c_void, db: *mut f32) -> c_int;
        pub fn d2dh_file_name(c: *const Conv) -> *const c_char;
        pub fn d2dh_set_device(c: *mut Conv, device: c_int);
        pub fn d2dh_find_dsd_files(paths: *const *const c_char, n: usize, recurse: c_int, each: Option<PathFn>, user: *mut c_void) -> c_int;
        pub fn d2dh_last_error() -> *const c_char;
    }
}

fn last_error() -> String { unsafe { CStr::from_ptr(ffi::d2dh_last_error()).to_string_lossy().into_owned() } }
fn cpath(p: &Path) -> CString { CString::new(p.to_string_lossy().as_bytes()).expect("path with NUL") }
fn opt_cpath(p: &Option<PathBuf>) -> Option<CString> { p.as_ref().map(|p| cpath(p)) }
fn ptr(o: &Option<CString>) -> *const c_char { o.as_ref().map_or(std::ptr::null(), |c| c.as_ptr()) }

/// `find_dsd_files(&paths, recurse)` (src/main.rs:275)
pub fn find_dsd_files(paths: &[PathBuf], recurse: bool) -> Result<Vec<PathBuf>, Box<dyn Error>> {
    unsafe extern "C" fn each(user: *mut c_void, path: *const c_char) {
        let v = &mut *(user as *mut Vec<PathBuf>);
        v.push(PathBuf::from(CStr::from_ptr(path).to_string_lossy().into_owned()));
    }
    let c: Vec<CString> = paths.iter().map(|p| cpath(p)).collect();
    let raw: Vec<*const c_char> = c.iter().map(|s| s.as_ptr()).collect();
    let mut out: Vec<PathBuf> = Vec::new();
    let rc = unsafe { ffi::d2dh_find_dsd_files(raw.as_ptr(), raw.len(), recurse as c_int, Some(each), &mut out as *mut _ as *mut c_void) };
    if rc != 0 { return Err(last_error().into()); }
    Ok(out)
}

pub struct Rdsd2Pcm {
    h: *mut ffi::Conv,
}
// one object lives on one Rayon worker (src/main.rs:361-394,429); it may be moved there
unsafe impl Send for Rdsd2Pcm {}

impl Drop for Rdsd2Pcm {
    fn drop(&mut self) { unsafe { ffi::d2dh_free(self.h) } }
}

struct Run<'a> {
    cancel: &'a AtomicBool,
    mirror: c_int, // the ABI polls an int: refreshed from the AtomicBool on every progress call
    sender: Option<Sender<ProgressUpdate>>,
}
unsafe extern "C" fn progress_cb(user: *mut c_void, percent: f32) {
    let run = &mut *(user as *mut Run);
    if run.cancel.load(Ordering::Relaxed) { std::ptr::write_volatile(&mut run.mirror, 1); }
    if let Some(s) = &run.sender { let _ = s.send(ProgressUpdate { percent }); }
}

impl Rdsd2Pcm {
    /// src/main.rs:325-342 (argument order kept)
    #[allow(clippy::too_many_arguments)]
    pub fn new(bit_depth: usize, output: OutputType, level_db: f64, output_rate: u32, out_dir: Option<PathBuf>, dither: DitherType,
               fmt: FmtType, endian: Endianness, dsd_rate: DsdRate, block_size: u32, channels: usize, filter: FilterType, append_rate: bool,
               base_dir: PathBuf, in_path: Option<PathBuf>) -> Result<Self, String> {
        let rate: u32 = dsd_rate.0;
        let (od, ip, bd) = (opt_cpath(&out_dir), opt_cpath(&in_path), cpath(&base_dir));
        let mut h = std::ptr::null_mut();
        let rc = unsafe {
            ffi::d2dh_new(bit_depth as u32, output.code(), level_db, output_rate, ptr(&od), dither.code(), fmt.code(), endian.code(), rate,
                          block_size, channels as u32, filter.code(), append_rate as c_int, bd.as_ptr(), ptr(&ip), &mut h)
        };
        if rc != 0 { return Err(last_error()); }
        Ok(Self { h })
    }

    /// src/main.rs:362-373: the container's own layout replaces the flags (README.md:103-105)
    #[allow(clippy::too_many_arguments)]
    pub fn from_container(bit_depth: usize, output: OutputType, level_db: f64, output_rate: u32, out_dir: Option<PathBuf>, dither: DitherType,
                          filter: FilterType, append_rate: bool, base_dir: PathBuf, path: PathBuf) -> Result<Self, String> {
        let (od, bd, p) = (opt_cpath(&out_dir), cpath(&base_dir), cpath(&path));
        let mut h = std::ptr::null_mut();
        let rc = unsafe {
            ffi::d2dh_from_container(bit_depth as u32, output.code(), level_db, output_rate, ptr(&od), dither.code(), filter.code(),
                                     append_rate as c_int, bd.as_ptr(), p.as_ptr(), &mut h)
        };
        if rc != 0 { return Err(last_error()); }
        Ok(Self { h })
    }

    /// src/bin/dsd_levels/main.rs:214-223
    pub fn new_level_check(output_rate: u32, path: Option<PathBuf>, fmt: FmtType, endian: Endianness, channels: usize, block_size: u32,
                           input_rate: DsdRate) -> Result<Self, String> {
        let p = opt_cpath(&path); // None = stdin (src/bin/dsd_levels/main.rs:273-281)
        let mut h = std::ptr::null_mut();
        let rc = unsafe { ffi::d2dh_new_level_check(output_rate, ptr(&p), fmt.code(), endian.code(), channels as u32, block_size, input_rate.0, &mut h) };
        if rc != 0 { return Err(last_error()); }
        Ok(Self { h })
    }

    /// `lib.do_conversion(&CANCEL_FLAG, sender)` (src/main.rs:345,429)
    pub fn do_conversion(&mut self, cancel: &AtomicBool, sender: Option<Sender<ProgressUpdate>>) -> Result<(), Box<dyn Error>> {
        let mut run = Run { cancel, mirror: cancel.load(Ordering::Relaxed) as c_int, sender };
        let rc = unsafe { ffi::d2dh_do_conversion(self.h, &run.mirror as *const c_int, Some(progress_cb), &mut run as *mut _ as *mut c_void) };
        if rc != 0 { return Err(last_error().into()); }
        Ok(())
    }

    /// `check_level(&CANCEL, sender) -> Result<f32, _>` (src/bin/dsd_levels/main.rs:252); may be NaN for silence
    pub fn check_level(&mut self, cancel: &AtomicBool, sender: Option<Sender<ProgressUpdate>>) -> Result<f32, Box<dyn Error>> {
        let mut run = Run { cancel, mirror: cancel.load(Ordering::Relaxed) as c_int, sender };
        let mut db = 0f32;
        let rc = unsafe { ffi::d2dh_check_level(self.h, &run.mirror as *const c_int, Some(progress_cb), &mut run as *mut _ as *mut c_void, &mut db) };
        if rc != 0 { return Err(last_error().into()); }
        Ok(db)
    }

    /// src/main.rs:398
    pub fn file_name(&self) -> String { unsafe { CStr::from_ptr(ffi::d2dh_file_name(self.h)).to_string_lossy().into_owned() } }

    /// extension: pin this object's work to a GPU, e.g. `rayon::current_thread_index().unwrap_or(0) % n_gpus`
    pub fn set_device(&mut self, device: i32) { unsafe { ffi::d2dh_set_device(self.h, device as c_int) } }
}
