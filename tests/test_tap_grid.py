"""What this build's 24-bit tap grid costs against the designs' own f64 taps.

Every kernel of this repository is bit-exact against the oracle BECAUSE the taps are dyadic 24-bit numbers
(DESIGN.md section 2).  The reference's rdsd2pcm uses f64 taps (its tables are not available); a future swap to such
taps would be approximated on that grid.  These tests measure the approximation with the oracle's f64-tap mode
(filters/filter_taps_f64.json: the same designs before rounding, summed in dsd2pcm's order):
  * float output: RMS difference -- must stay inside the north star's 1e-6 of full scale;
  * 24-bit output: how many samples differ (they differ by one LSB at most).
The CPU tests compare the two oracle modes; the GPU tests compare the production kernels with the f64-tap oracle."""
import json
import os

import numpy as np
import pytest

from helpers import pack_layout, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(1, 88200, "E"), (1, 352800, "D"), (1, 176400, "X"), (2, 88200, "C"), (2, 352800, "E")]


def f64_half_taps(name):
    with open(os.path.join(ROOT, "filters", "filter_taps_f64.json")) as f:
        return np.array([float.fromhex(x) for x in json.load(f)[name]])


def _inputs(dsd_rate, nbytes=4096 * 24):
    return pack_layout([synth("sine", nbytes, seed=5, dsd_rate=dsd_rate), synth("pink", nbytes, seed=6, amp=0.25, dsd_rate=dsd_rate)], "P", 4096)


def _f64_oracle(O, kw):
    o = O.Oracle(**kw)
    M = o.info()["M"]
    o.set_half_taps(f64_half_taps("%s_M%d" % (kw["filter"], M)))
    return o


def _stats(got32, ref32, got24, ref24):
    """float RMS difference (full scale = 1) and the 24-bit mismatch rate / largest difference"""
    rms = float(np.sqrt(np.mean((got32.astype(np.float64) - ref32.astype(np.float64)) ** 2)))
    a = got24.reshape(-1, 3).astype(np.int32); b = ref24.reshape(-1, 3).astype(np.int32)
    ia = a[:, 0] | (a[:, 1] << 8) | (a[:, 2] << 16); ib = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    ia = np.where(ia >= 1 << 23, ia - (1 << 24), ia); ib = np.where(ib >= 1 << 23, ib - (1 << 24), ib)
    return rms, float(np.mean(ia != ib)), int(np.abs(ia - ib).max())


@pytest.mark.parametrize("dsd_rate,out_rate,filt", CASES)
def test_tap_grid_cost_between_the_two_oracle_modes(oracle_mod, dsd_rate, out_rate, filt):
    O = oracle_mod
    buf = _inputs(dsd_rate)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096, filter=filt, seed=3)
    q32, n = O.Oracle(bit_depth=32, dither="X", **kw).translate(buf)
    r32, _ = _f64_oracle(O, dict(kw, bit_depth=32, dither="X")).translate(buf)
    q24, _ = O.Oracle(bit_depth=24, dither="T", **kw).translate(buf)
    r24, _ = _f64_oracle(O, dict(kw, bit_depth=24, dither="T")).translate(buf)
    rms, rate, worst = _stats(q32.view(np.float32), r32.view(np.float32), q24, r24)
    print(f"{filt} DSD{64 * dsd_rate}->{out_rate}: float RMS diff {rms:.3e}, 24-bit samples that differ {100 * rate:.2f} %, by at most {worst} LSB")
    assert rms < 1e-6                 # the north star's float tolerance holds with room to spare
    assert worst <= 2 and rate < 0.35  # the rates are what DESIGN.md quotes (12-29 %), bounded here; 2 LSB happen on the short M = 8 tables only


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("dsd_rate,out_rate,filt", CASES)
def test_production_kernels_against_the_f64_tap_reference(engine_lib, oracle_mod, kernel, dsd_rate, out_rate, filt):
    O = oracle_mod
    buf = _inputs(dsd_rate)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096, filter=filt, seed=3)
    g32, n = engine_lib.Engine(kernel=kernel, bit_depth=32, dither="X", **kw).translate(buf)
    r32, _ = _f64_oracle(O, dict(kw, bit_depth=32, dither="X")).translate(buf)
    g24, _ = engine_lib.Engine(kernel=kernel, bit_depth=24, dither="T", **kw).translate(buf)
    r24, _ = _f64_oracle(O, dict(kw, bit_depth=24, dither="T")).translate(buf)
    rms, rate, worst = _stats(g32.view(np.float32), r32.view(np.float32), g24, r24)
    print(f"kernel {kernel} {filt} DSD{64 * dsd_rate}->{out_rate}: float RMS diff {rms:.3e}, 24-bit mismatch {100 * rate:.2f} %, max {worst} LSB")
    assert rms < 1e-6
    assert worst <= 2 and rate < 0.35
