"""dsd2dxd_amd -- MI355X-native DSD->PCM decimation engine (drop-in for dsd2dxd's rdsd2pcm path).

The product is the C-ABI shared library `libdsd2dxd_amd.so` (include/dsd2dxd_amd.h), built from
dsd2dxd_amd/csrc for gfx950.  This package is only the thin ctypes binding the tests and bench.py
drive it through; there is no Python or CPU implementation of the conversion behind it.
"""
from ._capi import (D2DError, Engine, FileIO, Params, lib, library_path, build_library,  # noqa: F401
                    KERNEL_AUTO, KERNEL_LUT, KERNEL_MFMA,
                    DBG_NO_MX, DBG_NO_GAINQ, DBG_NO_COOP, DBG_HOST_STAGED, DBG_NO_PIPE, DBG_MFMA_V1, DBG_NO_INTQ, DBG_NS_GENERAL, DBG_TAPS32_2PASS, dbg_waves)
