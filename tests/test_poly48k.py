"""DSD64 / DSD128 -> 96 / 192 / 384 kHz and DSD256 -> 192 / 384 kHz: the 48k cascade composed into one polyphase filter on the bits (round 4).

Until round 3 these rates ran as two stages that met in HBM (a decimator to 352.8 kHz, a polyphase L/147 resampler).  The two designs
are now composed at table-build time (tools/design_filters.py: compose_polyphase; filters/filter_tables.inc: D2D_POLYS) and the output is

    y[m] = sum_j c[rho][j] s[q + D - j],   Mp m = Lp q + rho,   c = Q 2^-S.

The oracle keeps the earlier two-stage definition as a study mode (use_cascade); these tests pin the new tables and measure what the
change of definition costs against the f64 design of that cascade (float tolerance of the north star: 1e-6 RMS of full scale)."""
import json
import os

import numpy as np
import pytest

from helpers import decode_pcm, pack_layout, random_bytes, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATES = [(1, 96000), (1, 192000), (1, 384000), (2, 96000), (2, 192000), (2, 384000), (4, 192000), (4, 384000)]


def _polys():
    with open(os.path.join(ROOT, "filters", "filter_tables.json")) as f:
        return {(p["dsd_rate"], p["out_rate"]): p for p in json.load(f)["polys"]}


def _f64_half(name):
    with open(os.path.join(ROOT, "filters", "filter_taps_f64.json")) as f:
        return np.array([float.fromhex(x) for x in json.load(f)[name]])


def _digits32(v):
    out, v = [], np.array(v, dtype=np.int64)
    for _ in range(5):
        d = ((v + 16) & 31) - 16
        out.append(d)
        v = (v - d) // 32
    assert np.all(v == 0)
    return out


@pytest.mark.parametrize("dsd_rate,out_rate", RATES)
def test_table_invariants(dsd_rate, out_rate):
    """what the kernels rely on: unity DC gain per phase on the dyadic grid, 24-bit taps in five balanced base-32 digits whose sums
    recombine exactly in f32, |sum Q s| inside an int32, and no output that needs a bit the call has not fed yet"""
    p = _polys()[dsd_rate, out_rate]
    q = np.array(p["q"], dtype=np.int64).reshape(p["Lp"], p["NP"])
    from math import gcd
    g = gcd(out_rate, 2822400 * dsd_rate)
    assert (p["Lp"], p["Mp"]) == (out_rate // g, 2822400 * dsd_rate // g)
    assert np.all(q.sum(1) == 1 << p["S"])
    assert np.abs(q).sum(1).max() < 2 ** 31 - 2 ** 26
    assert p["D"] < 0                                      # the newest bit of output m lies before the cascade's newest bit
    assert np.abs(q[:, 0]).max() > 8 or np.abs(q[:, -1]).max() > 8     # trimmed to the taps that carry something
    for ph in range(p["Lp"]):
        dg = _digits32(2 * q[ph])
        sa = [int(np.abs(x).sum()) for x in dg]
        assert sa[0] + 32 * sa[1] + 1024 * sa[2] < 1 << 24
        assert sa[3] + 32 * (sa[4] + (1 << (p["S"] - 20))) < 1 << 24
    # the composed design is what was rounded: q / 2^S within a grid unit of it (plus the spread DC residual)
    c = np.array([float.fromhex(x) for x in p["coef"]]).reshape(p["Lp"], p["NP"])
    assert np.abs(q - c * 2.0 ** p["S"]).max() < 1.51


@pytest.mark.parametrize("dsd_rate,out_rate", RATES)
@pytest.mark.parametrize("endian", ["L", "M"])
def test_impulse_recovers_the_composed_taps(oracle_mod, dsd_rate, out_rate, endian):
    """one flipped bit in the idle pattern: output m changes by 2 c[rho][q + D - b] -- every tap of every phase, exactly"""
    O = oracle_mod
    p = _polys()[dsd_rate, out_rate]
    Lp, Mp, NP, D, S = p["Lp"], p["Mp"], p["NP"], p["D"], p["S"]
    q = np.array(p["q"], dtype=np.int64).reshape(Lp, NP)
    nbytes = 4096
    idle = 0x69 if endian == "M" else 0x96
    base = np.full(nbytes, idle, np.uint8)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=1, fmt="P", endianness=endian, block_size=4096, filter="E", bit_depth=32, dither="X")
    _, n0, y0 = O.Oracle(**kw).translate(base, want_f64=True)
    seen = np.zeros((Lp, NP), bool)
    for bit in (8 * 1500 + 3, 8 * 1500 + 6, 8 * 1501 + 0):
        x = base.copy()
        byte, k = divmod(bit, 8)
        x[byte] ^= (0x80 >> k) if endian == "M" else (1 << k)
        _, n1, y1 = O.Oracle(**kw).translate(x, want_f64=True)
        assert n1 == n0
        was_one = (idle >> (7 - k if endian == "M" else k)) & 1
        d = (y1 - y0)[:, 0] * (-1.0 if was_one else 1.0)
        for m in range(n0):
            qm, rho = divmod(m * Mp, Lp)
            j = qm + D - bit
            want = 2.0 * q[rho, j] * 2.0 ** -S if 0 <= j < NP else 0.0
            assert d[m] == want, (m, j, d[m], want)
            if 0 <= j < NP:
                seen[rho, j] = True
    assert seen.any(1).all()


@pytest.mark.parametrize("dsd_rate,out_rate", RATES)
def test_frames_follow_the_two_stage_rule(oracle_mod, dsd_rate, out_rate):
    """an output exists as soon as the two-stage form would have produced it: ceil(floor(bytes / Mb) * L / 147), in any call pattern"""
    O = oracle_mod
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=1, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="X")
    a, b = O.Oracle(**kw), O.Oracle(**kw)
    b.use_cascade()
    rng = np.random.default_rng(1)
    data = random_bytes(9000, 5)
    pos, tot = 0, 0
    while pos < data.size:
        n = int(rng.integers(0, 700))
        pa, fa = a.translate(data[pos:pos + n])
        pb, fb = b.translate(data[pos:pos + n])
        assert fa == fb
        pos += n
        tot += fa
    L = {96000: 40, 192000: 80, 384000: 160}[out_rate]
    nx = data.size // dsd_rate
    assert tot == -(-nx * L // 147)


def _band_rms(d, fs, lo, hi):
    n = len(d)
    w = np.hanning(n)[:, None]
    D = np.fft.rfft(d * w, axis=0)
    f = np.fft.rfftfreq(n, 1.0 / fs)
    m = (f >= lo) & (f < hi)
    return float(np.sqrt(2 * np.sum(np.abs(D[m]) ** 2) / (np.sum(w ** 2) * n * d.shape[1])))


@pytest.mark.parametrize("dsd_rate,out_rate", RATES)
def test_composed_filter_against_the_f64_design_of_the_cascade(oracle_mod, dsd_rate, out_rate):
    """The change of definition, measured: the float output of the composed 24-bit tables against the two-stage cascade in f64 (stage A's
    unquantised taps, stage B's f64 coefficients, the definition of rounds 2-3 before its integer grids).

    -> 96 kHz and -> 192 kHz: inside the north star's 1e-6 RMS over the whole band.  -> 384 kHz: inside it below 20 kHz and close to it over
    stage B's pass band; above that the two DIFFER BY DESIGN: the cascade sampled stage A's output at 352.8 kHz, which folds what stage A
    lets through between 176 and 300 kHz (DSD noise, attenuated but not stopped: stage A stops from 300 kHz) onto 53-176 kHz, and at
    384 kHz stage B's transition band (80-194 kHz) passes those images on at about -65 dBFS.  The composed filter has no sampling between
    the stages, hence no images; what it leaves above 80 kHz is the DSD noise itself under both filters' attenuation."""
    O = oracle_mod
    nbytes = 4096 * 24 * dsd_rate
    buf = pack_layout([synth("sine", nbytes, seed=1, dsd_rate=dsd_rate), synth("pink", nbytes, seed=2, amp=0.098, dsd_rate=dsd_rate)], "P", 4096)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=32, dither="X")
    _, fa, ya = O.Oracle(**kw).translate(buf, want_f64=True)
    c = O.Oracle(**kw)
    c.use_cascade(); c.use_f64_resamp_coef(); c.set_half_taps(_f64_half(f"A_M{8 * dsd_rate}"))
    _, fc, yc = c.translate(buf, want_f64=True)
    assert fa == fc
    d = (ya - yc)[1000:]                                   # (past the start-up: the cascade's stage-B history starts from zeros, the composed filter's from the idle pattern)
    full = float(np.sqrt(np.mean(d ** 2)))
    audio = _band_rms(d, out_rate, 0, 20e3)
    fpass = 0.227 * min(out_rate, 352800)
    passband = _band_rms(d, out_rate, 0, fpass)
    print(f"DSD{64 * dsd_rate}->{out_rate}: composed vs f64 cascade: full band {full:.2e}, below 20 kHz {audio:.2e}, below {fpass / 1e3:.0f} kHz {passband:.2e}")
    assert audio < 1e-7
    if out_rate < 384000:
        assert full < 1e-6
    else:
        assert passband < 2e-6 and 2e-5 < full < 1e-3      # the cascade's own images in its transition band (see above)


@pytest.mark.parametrize("dsd_rate,out_rate", [(1, 96000), (2, 96000), (1, 192000), (2, 192000)])
def test_integer_output_against_the_earlier_two_stage_definition(oracle_mod, dsd_rate, out_rate):
    """regression handle on the change: 24-bit TPDF output of the composed tables against the integer cascade these rates ran until round 3
    (same dither, same frame count): the samples move by a few LSB at most (the float difference above, -145 dBFS, is 0.4 LSB of 24 bits)"""
    O = oracle_mod
    nbytes = 4096 * 12 * dsd_rate
    buf = pack_layout([synth("sine", nbytes, seed=3, dsd_rate=dsd_rate), synth("pink", nbytes, seed=4, amp=0.098, dsd_rate=dsd_rate)], "P", 4096)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=9)
    a = O.Oracle(**kw)
    b = O.Oracle(**kw); b.use_cascade()
    pa, fa = a.translate(buf)
    pb, fb = b.translate(buf)
    assert fa == fb
    da = decode_pcm(pa[:fa * 6], 24, 2)[500:]
    db = decode_pcm(pb[:fb * 6], 24, 2)[500:]
    worst = int(np.abs(da - db).max())
    rate = float(np.mean(da != db))
    print(f"DSD{64 * dsd_rate}->{out_rate}: 24-bit samples that moved {100 * rate:.1f} %, by at most {worst} LSB")
    assert worst <= (3 if out_rate == 96000 else 48)      # (-> 192 kHz: 2e-7 .. 9e-7 RMS of full scale = 2 .. 7 LSB RMS, the cascade's images between 44 and 96 kHz)
