"""ctypes binding of include/dsd2dxd_amd.h (the same calls the Rust `extern "C"` block in
INTEGRATION.md makes).  Loading fails loudly when the HIP library has not been built."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
KERNEL_AUTO, KERNEL_LUT, KERNEL_MFMA = 0, 1, 2
# d2d_params.debug_flags (include/dsd2dxd_amd.h: D2D_DBG_*): diagnostic dispatch switches, 0 in production
DBG_NO_MX, DBG_NO_GAINQ, DBG_NO_COOP, DBG_HOST_STAGED, DBG_NO_PIPE, DBG_MFMA_V1, DBG_NO_INTQ, DBG_NS_GENERAL = (1 << i for i in range(8))
DBG_TAPS32_2PASS = 1 << 16


def dbg_waves(n):
    """debug_flags bits 8..15: waves per block of the matrix-core kernels"""
    return (int(n) & 0xFF) << 8



def library_path():
    # D2D_AMD_LIB: development A/B builds of the same library (tools/ab_build.sh); never a different implementation
    return os.environ.get("D2D_AMD_LIB") or os.path.join(HERE, "libdsd2dxd_amd.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(HERE, "csrc"), "-j4"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return library_path()


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("dsd_rate", C.c_uint32), ("output_rate", C.c_uint32),
                ("channels", C.c_uint32), ("fmt", C.c_uint32), ("endianness", C.c_uint32),
                ("block_size", C.c_uint32), ("filter", C.c_uint32), ("bit_depth", C.c_uint32),
                ("dither", C.c_uint32), ("kernel", C.c_uint32), ("device", C.c_int32),
                ("level_db", C.c_double), ("seed", C.c_uint64),
                ("channel_first", C.c_uint32), ("channel_count", C.c_uint32),
                ("tap_bits", C.c_uint32), ("debug_flags", C.c_uint32)]


class FileIO(C.Structure):
    _fields_ = [("dsd", C.c_void_p), ("bytes_per_channel", C.c_size_t), ("pcm", C.c_void_p),
                ("pcm_capacity_bytes", C.c_size_t), ("frames_out", C.c_size_t)]


class Info(C.Structure):
    _fields_ = [("decimation", C.c_uint32), ("ntaps", C.c_uint32), ("scale_bits", C.c_uint32),
                ("resamp_L", C.c_uint32), ("resamp_M", C.c_uint32), ("resamp_P", C.c_uint32),
                ("kernel", C.c_uint32), ("abi_version", C.c_uint32), ("filter_name", C.c_char * 32)]


READ_FN = C.CFUNCTYPE(C.c_long, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_float)

EXPORTS = ["d2d_create", "d2d_create_error", "d2d_destroy", "d2d_reset", "d2d_last_error",
           "d2d_frame_bytes", "d2d_next_frames", "d2d_translate", "d2d_translate_batch_device",
           "d2d_translate_batch_host",
           "d2d_peak", "d2d_peak_dbfs", "d2d_convert_stream", "d2d_tables_bytes",
           "d2d_tables_export_device", "d2d_tables_import_device", "d2d_get_info", "d2d_kernel_name",
           "d2d_profile_enable", "d2d_profile_read", "d2d_profile_read_all"]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build the HIP extension first "
                           "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64.so.1 and asks for them by file name, so
    # it does not reuse an already-loaded /opt/rocm copy; two HIP runtimes in one process leave the
    # second without a GPU.  Loading torch FIRST makes the loader satisfy this library's
    # DT_NEEDED libamdhip64.so.7 with torch's copy (same soname): one runtime, shared device pointers.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.d2d_create.argtypes = [C.POINTER(Params), C.c_uint32, C.POINTER(C.c_void_p)]
    L.d2d_create.restype = C.c_int
    L.d2d_create_error.restype = C.c_char_p
    L.d2d_destroy.argtypes = [C.c_void_p]
    L.d2d_destroy.restype = None
    L.d2d_reset.argtypes = [C.c_void_p]
    L.d2d_last_error.argtypes = [C.c_void_p]
    L.d2d_last_error.restype = C.c_char_p
    L.d2d_frame_bytes.argtypes = [C.c_void_p]
    L.d2d_frame_bytes.restype = C.c_size_t
    L.d2d_next_frames.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t]
    L.d2d_next_frames.restype = C.c_size_t
    L.d2d_translate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.d2d_translate_batch_device.argtypes = [C.c_void_p, C.POINTER(FileIO), C.c_uint32, C.c_void_p]
    L.d2d_translate_batch_host.argtypes = [C.c_void_p, C.POINTER(FileIO), C.c_uint32, C.c_size_t]
    L.d2d_translate_batch_host.restype = C.c_int
    L.d2d_peak.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.d2d_peak_dbfs.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]
    L.d2d_convert_stream.argtypes = [C.c_void_p, READ_FN, C.c_void_p, WRITE_FN, C.c_void_p,
                                     C.POINTER(C.c_int), PROGRESS_FN, C.c_void_p, C.c_uint64, C.c_size_t]
    L.d2d_tables_bytes.argtypes = [C.c_void_p]
    L.d2d_tables_bytes.restype = C.c_size_t
    L.d2d_tables_export_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.d2d_tables_import_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.d2d_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.d2d_kernel_name.argtypes = [C.c_void_p]
    L.d2d_kernel_name.restype = C.c_char_p
    L.d2d_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.d2d_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.d2d_profile_read_all.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    _lib = L
    return L


class D2DError(Exception):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


def make_params(dsd_rate=1, output_rate=352800, channels=2, fmt="I", endianness="M", block_size=4096,
                filter="E", bit_depth=24, dither="X", level_db=0.0, seed=0, kernel=KERNEL_AUTO, device=0,
                channel_first=0, channel_count=0, tap_bits=0, debug=0):
    """Argument names and defaults follow the reference CLI (src/main.rs:40-110); channel_first/count
    select a channel subset (0 = all), tap_bits the tap grid (0 / 24, or 32), see include/dsd2dxd_amd.h."""
    return Params(C.sizeof(Params), dsd_rate, output_rate, channels, 1 if fmt.upper() == "P" else 0,
                  1 if endianness.upper() == "M" else 0, block_size, ord(filter.upper()), bit_depth,
                  ord(dither.upper()), kernel, device, level_db, seed, channel_first, channel_count, tap_bits, debug)


class Engine:
    """One conversion context = one Rdsd2Pcm of the reference (src/main.rs:325-342), or a batch of
    `n_files` of them advanced together."""

    def __init__(self, n_files=1, **kw):
        L = lib()
        self.params = make_params(**kw)
        h = C.c_void_p()
        rc = L.d2d_create(C.byref(self.params), n_files, C.byref(h))
        if rc:
            raise D2DError(rc, L.d2d_create_error().decode())
        self._h = h
        self.n_files = n_files
        self.channels = self.params.channels                      # width of the INPUT
        self.out_channels = self.params.channel_count or (self.params.channels - self.params.channel_first)
        self.frame_bytes = L.d2d_frame_bytes(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().d2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise D2DError(rc, lib().d2d_last_error(self._h).decode())

    def info(self):
        i = Info()
        self._check(lib().d2d_get_info(self._h, C.byref(i)))
        return dict(M=i.decimation, ntaps=i.ntaps, S=i.scale_bits, L=i.resamp_L, P=i.resamp_P,
                    kernel=i.kernel, filter=i.filter_name.decode())

    def kernel_name(self):
        return lib().d2d_kernel_name(self._h).decode()

    def reset(self):
        self._check(lib().d2d_reset(self._h))

    def next_frames(self, bytes_per_channel, file=0):
        return lib().d2d_next_frames(self._h, file, bytes_per_channel)

    def translate(self, dsd):
        """Host buffers: `dsd` holds channels*bytes_per_channel bytes; returns (uint8 ndarray, frames)."""
        import numpy as np
        buf = np.ascontiguousarray(np.frombuffer(dsd, dtype=np.uint8) if not isinstance(dsd, np.ndarray) else dsd)
        assert buf.size % self.channels == 0
        bpc = buf.size // self.channels
        n = self.next_frames(bpc)
        out = np.zeros(max(n * self.frame_bytes, 1), dtype=np.uint8)
        frames = C.c_size_t()
        self._check(lib().d2d_translate(self._h, buf.ctypes.data, bpc, out.ctypes.data, out.size, C.byref(frames)))
        return out[:frames.value * self.frame_bytes], frames.value

    def translate_into(self, dsd_ptr, bytes_per_channel, pcm_ptr, pcm_capacity_bytes):
        """d2d_translate on raw HOST addresses (e.g. pinned tensors: the kernels then work on them directly); returns frames."""
        frames = C.c_size_t()
        self._check(lib().d2d_translate(self._h, C.c_void_p(dsd_ptr), bytes_per_channel, C.c_void_p(pcm_ptr), pcm_capacity_bytes, C.byref(frames)))
        return frames.value

    def translate_batch_device(self, ios, stream=None):
        """ios: ctypes array of FileIO with DEVICE pointers; asynchronous on `stream` (hipStream_t int)."""
        self._check(lib().d2d_translate_batch_device(self._h, ios, len(ios), C.c_void_p(stream or 0)))

    def translate_batch_host(self, ios, slice_bytes_per_channel=0):
        """ios: ctypes array of FileIO with HOST pointers (pinned for overlap); synchronous."""
        self._check(lib().d2d_translate_batch_host(self._h, ios, len(ios), slice_bytes_per_channel))

    def peak(self, channel, file=0):
        v = C.c_double()
        self._check(lib().d2d_peak(self._h, file, channel, C.byref(v)))
        return v.value

    def peak_dbfs(self, file=0):
        v = C.c_float()
        self._check(lib().d2d_peak_dbfs(self._h, file, C.byref(v)))
        return v.value

    def profile_enable(self, on=True):
        self._check(lib().d2d_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, n = C.c_double(), C.c_uint64()
        self._check(lib().d2d_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_read_all(self):
        """(FIR kernel ms, whole-call kernels ms, launches) since the last read"""
        fir, step, n = C.c_double(), C.c_double(), C.c_uint64()
        self._check(lib().d2d_profile_read_all(self._h, C.byref(fir), C.byref(step), C.byref(n)))
        return fir.value, step.value, n.value

    def tables_bytes(self):
        return lib().d2d_tables_bytes(self._h)

    def tables_export_device(self, dev_ptr, cap, stream=None):
        self._check(lib().d2d_tables_export_device(self._h, C.c_void_p(dev_ptr), cap, C.c_void_p(stream or 0)))

    def tables_import_device(self, dev_ptr, nbytes, stream=None):
        self._check(lib().d2d_tables_import_device(self._h, C.c_void_p(dev_ptr), nbytes, C.c_void_p(stream or 0)))

    def convert_stream(self, read, write, total_bytes_per_channel=0, chunk_bytes_per_channel=1 << 22,
                       cancel=None, progress=None):
        """do_conversion for callers that own the I/O: read(cap)->bytes (channels*k bytes in layout, k<=cap),
        write(bytes)."""
        keep = {}

        def _read(_u, dst, cap):
            data = read(cap)
            if not data:
                return 0
            n = len(data)
            C.memmove(dst, data, n)
            return n // self.channels

        def _write(_u, p, n):
            write(C.string_at(p, n))
            return 0

        def _prog(_u, pct):
            if progress:
                progress(pct)

        keep["r"], keep["w"], keep["p"] = READ_FN(_read), WRITE_FN(_write), PROGRESS_FN(_prog)
        cflag = cancel if cancel is not None else C.c_int(0)
        self._check(lib().d2d_convert_stream(self._h, keep["r"], None, keep["w"], None, C.byref(cflag),
                                             keep["p"], None, total_bytes_per_channel, chunk_bytes_per_channel))
