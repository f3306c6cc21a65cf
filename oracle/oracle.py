"""ctypes loader for the CPU oracle (oracle/d2d_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package dsd2dxd_amd never does.  PARITY UNPINNED -- see oracle/d2d_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")


LIB_NATIVE = os.path.join(HERE, "liboracle_native.so")


def build(force=False, native=False):
    """native=True: -O3 -march=native build for the cpu_baseline timing; always rebuilt on the box it
    is timed on (the portable build travels with the repo snapshot, this one must not)."""
    src = os.path.join(HERE, "d2d_oracle.c")
    out = LIB_NATIVE if native else LIB
    deps = [src, os.path.join(HERE, "d2d_oracle.h"), os.path.join(HERE, "..", "filters", "filter_tables.inc")]
    if (not force and not native and os.path.exists(out)
            and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps if os.path.exists(d))):
        return out
    flags = ["-O3", "-march=native"] if native else ["-O2"]
    subprocess.check_call(["gcc"] + flags + ["-ffp-contract=off", "-fPIC", "-shared", "-o", out, src, "-lm"])
    return out


class OrcParams(C.Structure):
    _fields_ = [("dsd_rate", C.c_uint32), ("output_rate", C.c_uint32), ("channels", C.c_uint32),
                ("fmt", C.c_uint32), ("endianness", C.c_uint32), ("block_size", C.c_uint32),
                ("filter", C.c_uint32), ("bit_depth", C.c_uint32), ("dither", C.c_uint32),
                ("fir_mode", C.c_uint32), ("level_db", C.c_double), ("seed", C.c_uint64)]


_lib = None


def use_native():
    """Switch this process to the -march=native build (bench.py's cpu_baseline leg)."""
    global _lib
    build(native=True)
    _lib = None
    _load(LIB_NATIVE)


def lib():
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _load(LIB)
    return _lib


def _load(path):
    global _lib
    L = C.CDLL(path)
    L.orc_create.argtypes = [C.POINTER(OrcParams), C.POINTER(C.c_void_p), C.POINTER(C.c_char_p)]
    L.orc_create.restype = C.c_int
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_max_frames.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_max_frames.restype = C.c_size_t
    L.orc_frame_bytes.argtypes = [C.c_void_p]
    L.orc_frame_bytes.restype = C.c_size_t
    L.orc_translate_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                C.c_void_p, C.POINTER(C.c_size_t)]
    L.orc_translate_f64.restype = C.c_int
    L.orc_translate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orc_translate_stream.restype = C.c_int
    L.orc_peak.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_peak.restype = C.c_double
    L.orc_peak_dbfs.argtypes = [C.c_void_p]
    L.orc_peak_dbfs.restype = C.c_float
    L.orc_filter_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
    L.orc_tap.argtypes = [C.c_void_p, C.c_int]
    L.orc_tap.restype = C.c_double
    L.orc_set_half_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_use_fine_taps.argtypes = [C.c_void_p]
    L.orc_use_f64_resamp_coef.argtypes = [C.c_void_p]
    L.orc_use_cascade.argtypes = [C.c_void_p]
    L.orc_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    L.orc_rng.restype = C.c_uint32
    _lib = L


class OracleError(Exception):
    pass


class Oracle:
    """One conversion context (the CPU counterpart of one Rdsd2Pcm, src/main.rs:325-342)."""

    def __init__(self, dsd_rate=1, output_rate=352800, channels=2, fmt="I", endianness="M",
                 block_size=4096, filter="E", bit_depth=24, dither="X", level_db=0.0, seed=0,
                 fir_mode=1, tap_bits=24):
        L = lib()
        p = OrcParams(dsd_rate, output_rate, channels, 1 if fmt.upper() == "P" else 0,
                      1 if endianness.upper() == "M" else 0, block_size, ord(filter.upper()),
                      bit_depth, ord(dither.upper()), fir_mode, level_db, seed)
        h = C.c_void_p()
        err = C.c_char_p()
        rc = L.orc_create(C.byref(p), C.byref(h), C.byref(err))
        if rc:
            raise OracleError(err.value.decode() if err.value else f"error {rc}")
        self._h = h
        self.channels = channels
        self.bit_depth = bit_depth
        self.frame_bytes = L.orc_frame_bytes(h)
        if tap_bits == 32:
            if L.orc_use_fine_taps(h):
                raise OracleError("32-bit taps: 44.1k-family rates with dither T, R, F or X")
        elif tap_bits not in (0, 24):
            raise OracleError("tap_bits must be 24 or 32")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        v = [C.c_int() for _ in range(5)]
        lib().orc_filter_info(self._h, *[C.byref(x) for x in v])
        return dict(M=v[0].value, ntaps=v[1].value, S=v[2].value, L=v[3].value, P=v[4].value)

    def use_f64_resamp_coef(self):
        """stage B of the 48k cascade with the design's f64 coefficients instead of the 2^-28 grid (a study mode)"""
        if lib().orc_use_f64_resamp_coef(self._h):
            raise OracleError("not a 48k-family context")

    def use_cascade(self):
        """DSD64 / DSD128 -> 48k multiples through the two-stage cascade of the earlier definition (a study mode)"""
        if lib().orc_use_cascade(self._h):
            raise OracleError("not a fresh 48k-family context")

    def set_half_taps(self, half):
        """replace the taps (2nd half, centre outward) -- e.g. by the designs' unquantised f64 taps"""
        a = np.ascontiguousarray(np.asarray(half, dtype=np.float64))
        if lib().orc_set_half_taps(self._h, a.ctypes.data, a.size):
            raise OracleError("tap count does not match the context's filter")

    def taps(self):
        n = self.info()["ntaps"]
        return np.array([lib().orc_tap(self._h, i) for i in range(n)])

    def translate(self, dsd, want_f64=False):
        """dsd: bytes-like holding channels*bytes_per_channel bytes in the context's layout.
        Returns (pcm bytes as uint8 array, frames[, f64 array frames x channels])."""
        buf = np.ascontiguousarray(np.frombuffer(dsd, dtype=np.uint8) if not isinstance(dsd, np.ndarray) else dsd)
        assert buf.size % self.channels == 0
        bpc = buf.size // self.channels
        L = lib()
        nmax = L.orc_max_frames(self._h, bpc)
        out = np.zeros(nmax * self.frame_bytes, dtype=np.uint8)
        f64 = np.zeros((nmax, self.channels), dtype=np.float64) if want_f64 else None
        frames = C.c_size_t()
        rc = L.orc_translate_f64(self._h, buf.ctypes.data, bpc, out.ctypes.data, out.size,
                                 f64.ctypes.data if want_f64 else None, C.byref(frames))
        if rc:
            raise OracleError(f"translate failed: {rc}")
        assert frames.value == nmax
        return (out, frames.value, f64) if want_f64 else (out, frames.value)

    def translate_stream(self, dsd, out=None):
        """the streaming organisation of the same conversion (orc_translate_stream): identical bytes and state; `out`
        may be a preallocated uint8 array that is reused from call to call (the timed CPU baseline does that)"""
        buf = np.ascontiguousarray(np.frombuffer(dsd, dtype=np.uint8) if not isinstance(dsd, np.ndarray) else dsd)
        bpc = buf.size // self.channels
        L = lib()
        nmax = L.orc_max_frames(self._h, bpc)
        if out is None or out.size < nmax * self.frame_bytes:
            out = np.zeros(nmax * self.frame_bytes, dtype=np.uint8)
        frames = C.c_size_t()
        rc = L.orc_translate_stream(self._h, buf.ctypes.data, bpc, out.ctypes.data, out.size, C.byref(frames))
        if rc:
            raise OracleError(f"translate failed: {rc}")
        return out, frames.value

    def peak(self, ch):
        return lib().orc_peak(self._h, ch)

    def peak_dbfs(self):
        return lib().orc_peak_dbfs(self._h)


def rng(seed, channel, n):
    return lib().orc_rng(seed, channel, n)
