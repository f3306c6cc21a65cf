"""Shared test helpers: synthetic DSD (oracle/synth.c), layout packing, PCM decoding."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SYN = None


def _synth_lib():
    global _SYN
    if _SYN is None:
        so = os.path.join(ROOT, "oracle", "libsynth.so")
        src = os.path.join(ROOT, "oracle", "synth.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src, "-lm"])
        L = C.CDLL(so)
        L.synth_dsd.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_size_t, C.c_int, C.c_void_p]
        L.synth_dsd.restype = None
        _SYN = L
    return _SYN


def synth(kind, nbytes, seed=1, amp=0.352, freq=1000.0, phase=0.0, dsd_rate=1, msb_first=False):
    """One channel of synthetic DSD: kind 'sine' | 'pink' | 'silence'."""
    buf = np.zeros(nbytes, dtype=np.uint8)
    k = {"sine": 0, "pink": 1, "silence": 2}[kind]
    _synth_lib().synth_dsd(k, seed, amp, freq, phase, 2822400.0 * dsd_rate, nbytes, int(msb_first), buf.ctypes.data)
    return buf


def random_bytes(nbytes, seed):
    return np.random.default_rng(seed).integers(0, 256, nbytes, dtype=np.uint8)


def pack_layout(chans, fmt, block_size):
    """chans: list of equal-length uint8 arrays (one per channel) -> one call buffer in the reference's
    layout: planar [ch0 blk][ch1 blk]... (a short last block keeps that shape), or byte-interleaved."""
    C_ = len(chans)
    L = len(chans[0])
    if fmt.upper() == "I":
        return np.stack(chans, axis=1).reshape(-1).copy()
    out = np.zeros(L * C_, dtype=np.uint8)
    pos = 0
    for b0 in range(0, L, block_size):
        bl = min(block_size, L - b0)
        for c in range(C_):
            out[pos:pos + bl] = chans[c][b0:b0 + bl]
            pos += bl
    return out


BITREV = np.array([int(f"{i:08b}"[::-1], 2) for i in range(256)], dtype=np.uint8)


def decode_pcm(raw, bit_depth, channels):
    """interleaved little-endian PCM bytes -> int64/float32 array [frames, channels]"""
    raw = np.asarray(raw, dtype=np.uint8)
    if bit_depth == 32:
        return raw.view(np.float32).reshape(-1, channels)
    if bit_depth == 16:
        return raw.view("<i2").astype(np.int64).reshape(-1, channels)
    b = raw.reshape(-1, 3).astype(np.int64)
    v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v)
    return v.reshape(-1, channels)


# ---- container writers (for the host-driver tests): built from the public DSF / DSDIFF layouts ----
import struct


def write_dsf(path, chans, dsd_rate=1, lsb_first=True, block=4096, sample_count=None, id3=b""):
    C_ = len(chans)
    n = len(chans[0])
    nblk = -(-n // block)
    payload = bytearray()
    for b in range(nblk):
        for c in range(C_):
            piece = bytes(chans[c][b * block:(b + 1) * block])
            payload += piece + bytes(block - len(piece))
    sc = sample_count if sample_count is not None else n * 8
    data_sz = 12 + len(payload)
    total = 28 + 52 + data_sz + len(id3)
    meta = 28 + 52 + data_sz if id3 else 0
    with open(path, "wb") as f:
        f.write(b"DSD " + struct.pack("<QQQ", 28, total, meta))
        f.write(b"fmt " + struct.pack("<QIIIIIIQII", 52, 1, 0, 2 if C_ == 2 else 1, C_, 2822400 * dsd_rate,
                                      1 if lsb_first else 8, sc, block, 0))
        f.write(b"data" + struct.pack("<Q", data_sz) + payload + id3)


def write_dff(path, chans, dsd_rate=1, tail=b""):
    C_ = len(chans)
    data = np.stack(chans, axis=1).reshape(-1).tobytes()
    ids = [b"SLFT", b"SRGT", b"C   ", b"LFE ", b"LS  ", b"RS  "][:C_] if C_ != 1 else [b"C   "]
    while len(ids) < C_:
        ids.append(b"C%03d" % len(ids))
    chnl = struct.pack(">H", C_) + b"".join(ids)
    cmpr = b"DSD " + bytes([14]) + b"not compressed" + b"\x00"
    prop_body = b"SND " + b"FS  " + struct.pack(">QI", 4, 2822400 * dsd_rate) + b"CHNL" + struct.pack(">Q", len(chnl)) + chnl \
        + b"CMPR" + struct.pack(">Q", len(cmpr)) + cmpr
    body = b"DSD " + b"FVER" + struct.pack(">QI", 4, 0x01050000) + b"PROP" + struct.pack(">Q", len(prop_body)) + prop_body \
        + b"DSD " + struct.pack(">Q", len(data)) + data + (b"\x00" if len(data) & 1 else b"") + tail
    with open(path, "wb") as f:
        f.write(b"FRM8" + struct.pack(">Q", len(body)) + body)
