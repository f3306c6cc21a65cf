#!/usr/bin/env python3
"""Design and freeze the decimation-filter tables used by the engine AND the oracle.

The reference keeps its tap tables inside the (absent) rdsd2pcm crate, so none of
them can be read; these are this build's OWN designs, following only the documented
constraints of the reference:
  * even tap count, symmetric, only the 2nd half stored   (/root/reference/README.md:252)
  * flat to ~20-22 kHz, gentle roll-off, transition band edging slightly past the
    output Nyquist, tap count kept small                   (/root/reference/README.md:254)
  * 44.1k multiples: ONE filter; 48k multiples: CASCADED gentle FIRs (README.md:230)
  * filter families E / X / D / C and where each is legal  (src/main.rs:62-67)

Every integer-decimator tap is rounded to a dyadic grid c = q * 2^-S with |q| < 2^23 (24-bit
fixed point: the coefficient-rounding floor sits near -150 dB, 40 dB under the designs' own stop
bands) and the taps sum to exactly 2^S (unity DC gain).  24 bits is what lets the matrix-core kernel
feed stream bits to the int8 MFMA with ONE mask per operand register (the bit keeps its position
2^p inside the byte, the table holds q * 2^(7-p) in four int8 limbs).  Consequence: a +-1 weighted sum of
the taps is exactly representable in f64 whatever the summation order, so the CPU
oracle (f64, like the reference: README.md:230,236), the LDS-LUT kernel (f64) and the
int8-limb MFMA kernel (exact integers) all produce the SAME number.

Run:  python tools/design_filters.py   -> filters/filter_tables.inc + filters/filter_tables.json
"""
import json
import os
import sys

import numpy as np
import scipy.signal as sg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# dsd2pcm's published 96-tap filter (2nd half, centre outward).  Sebastian Gesemann,
# "dsd2pcm" (BSD-style licence, https://code.google.com/archive/p/dsd2pcm/) -- the lineage the
# reference README acknowledges (README.md:240) and offers as filter "D" (src/main.rs:65).
# NOT taken from /root/reference (it is not there); checked here by its response:
# sum == 1 - 3e-6, -158 dB beyond 300 kHz at fs = 2.8224 MHz.
DSD2PCM_HTAPS = [
    0.09950731974056658, 0.09562845727714668, 0.08819647126516944,
    0.07782552527068175, 0.06534876523171299, 0.05172629311427257,
    0.0379429484910187, 0.02490921351762261, 0.0133774746265897,
    0.003883043418804416, -0.003284703416210726, -0.008080250212687497,
    -0.01067241812471033, -0.01139427235000863, -0.0106813877974587,
    -0.009007905078766049, -0.006828859761015335, -0.004535184322001496,
    -0.002425035959059578, -0.0006922187080790708, 0.0005700762133516592,
    0.001353838005269448, 0.001713709169690937, 0.001742046839472948,
    0.001545601648013235, 0.001226696225277855, 0.0008704322683580222,
    0.0005381636200535649, 0.000266446345425276, 7.002968738383528e-05,
    -5.279407053811266e-05, -0.0001140625650874684, -0.0001304796361231895,
    -0.0001189970287491285, -9.396247155265073e-05, -6.577634378272832e-05,
    -4.07492895872535e-05, -2.17407957554587e-05, -9.163058931391722e-06,
    -2.017460145032201e-06, 1.249721855219005e-06, 2.166655190537392e-06,
    1.930520892991082e-06, 1.319400334374195e-06, 7.410039764949091e-07,
    3.423230509967409e-07, 1.244182214744588e-07, 3.130441005359396e-08,
]


def response_db(h, fs, n=1 << 16):
    w, H = sg.freqz(h, worN=n, fs=fs)
    return w, 20 * np.log10(np.abs(H) + 1e-300)


def design_equiripple(M, fp_rel, fs_rel, ripple_db, stop_db, wstop, nmin=32):
    """Smallest N (multiple of 16) whose remez design meets the spec.
    Frequencies relative to the OUTPUT rate Fo; input rate is M*Fo (normalised Fo=1)."""
    fs = float(M)
    N = max(nmin, 16)
    while True:
        try:
            h = sg.remez(N, [0, fp_rel, fs_rel, fs / 2], [1, 0], weight=[1, wstop],
                         fs=fs, maxiter=400, grid_density=24)
        except Exception:
            h = None
        if h is not None:
            w, mag = response_db(h / h.sum(), fs)
            pb = mag[w <= fp_rel]
            if pb.max() - pb.min() <= ripple_db and mag[w >= fs_rel].max() <= stop_db:
                return h / h.sum()
        N += 16
        if N > 8192:
            raise RuntimeError("no design")


def design_kaiser(M, N, fc_rel, beta):
    h = sg.firwin(N, fc_rel, window=("kaiser", beta), fs=float(M))
    return h / h.sum()


def design_cheby(M, N, fc_rel, at):
    h = sg.firwin(N, fc_rel, window=("chebwin", at), fs=float(M))
    return h / h.sum()


def quantise(h):
    """Round symmetric even-length h to q*2^-S, |q| < 2^23, sum(q) == 2^S exactly."""
    h = np.asarray(h, dtype=np.float64)
    h = h / h.sum()                                       # unity DC gain (dsd2pcm's own table sums to 1 - 3e-6)
    N = len(h)
    assert N % 16 == 0
    half = 0.5 * (h[N // 2:] + h[:N // 2][::-1])        # enforce symmetry
    S = int(np.floor(np.log2((2 ** 23 - 2 ** 16) / np.abs(half).max())))
    S = min(S, 40)
    q = np.rint(half * 2.0 ** S).astype(np.int64)
    resid = (1 << (S - 1)) - int(q.sum())                 # half must sum to 2^(S-1)
    # the (tiny) DC residual goes one unit at a time to the taps whose rounding lost most in that direction: no tap
    # then sits more than ~0.6 unit from its design value (folding it all into the centre tap put up to 12 units
    # there and doubled the grid's error against the f64 design, tests/test_tap_grid.py)
    frac = half * 2.0 ** S - q
    order = np.argsort(-frac, kind="stable") if resid > 0 else np.argsort(frac, kind="stable")
    for i in range(abs(resid)):
        q[order[i]] += 1 if resid > 0 else -1
    assert abs(resid) < 4 * N, resid
    assert np.abs(q).max() < 2 ** 23 - 2 ** 15     # q * 2^7 must fit four balanced int8 limbs (MFMA kernel)
    assert 2 * int(q.sum()) == 1 << S
    assert 2 * int(np.abs(q).sum()) < 1 << 52
    return S, q


FINE_BITS = 8      # the optional 32-bit tap grid: q32 * 2^-(S + 8), eight more fraction bits than the 24-bit tables


def quantise_fine(half_f64, S, q24):
    """The same design on the 32-bit grid 2^-(S+8): q32 = 256 q24 + r with a SMALL residual table r (|r| < 512) whose taps sum to
    zero, so that a conversion with 32-bit taps is the shipped 24-bit FIR plus a second FIR pass over the residuals.
    `half_f64`: the design's own taps (2nd half, unity DC gain)."""
    half = np.asarray(half_f64, dtype=np.float64)
    Sf = S + FINE_BITS
    q = np.rint(half * 2.0 ** Sf).astype(np.int64)
    resid = (1 << (Sf - 1)) - int(q.sum())
    frac = half * 2.0 ** Sf - q
    order = np.argsort(-frac, kind="stable") if resid > 0 else np.argsort(frac, kind="stable")
    assert abs(resid) <= len(q), resid
    for i in range(abs(resid)):
        q[order[i]] += 1 if resid > 0 else -1
    assert 2 * int(q.sum()) == 1 << Sf
    r = q - (np.asarray(q24, dtype=np.int64) << FINE_BITS)
    assert int(r.sum()) == 0 and np.abs(r).max() < 512, (int(r.sum()), int(np.abs(r).max()))
    assert np.abs(q).max() < 2 ** 31
    return q


RESAMP_T = 28      # stage-B coefficient grid: G = round(g * 2^28), four balanced int8 limbs in the matrix-core kernel


def quantise_polyphase(poly):
    """Stage-B coefficients on the dyadic grid 2^-RESAMP_T: every phase sums to 2^RESAMP_T EXACTLY (unity DC gain per phase: the
    f64 design's branches sum to 1 +- 7e-7, which is a signal-independent ripple at the output rate's sub-multiples), the residual
    spread one unit at a time over the taps whose rounding lost most, as quantise() does for the decimators."""
    poly = np.asarray(poly, dtype=np.float64)
    out = np.zeros(poly.shape, dtype=np.int64)
    for ph in range(poly.shape[0]):
        g = poly[ph] / poly[ph].sum()
        q = np.rint(g * 2.0 ** RESAMP_T).astype(np.int64)
        resid = (1 << RESAMP_T) - int(q.sum())
        frac = g * 2.0 ** RESAMP_T - q
        order = np.argsort(-frac, kind="stable") if resid > 0 else np.argsort(frac, kind="stable")
        for i in range(abs(resid)):
            q[order[i]] += 1 if resid > 0 else -1
        assert int(q.sum()) == 1 << RESAMP_T and abs(resid) < 4 * len(q)
        out[ph] = q
    assert np.abs(out).max() < 2 ** 31 - 2 ** 24          # four balanced int8 limbs
    assert int(np.abs(out).sum(1).max()) < 2 ** (RESAMP_T + 1)
    return out


POLY_TRIM = 8      # direct 48k tables: leading / trailing taps that stay within this many grid units in every phase are dropped


def stage_b_dense(out_rate, L, P, dens):
    """Stage B's Kaiser design sampled `dens` times as densely: dens*N - (dens-1) taps whose every dens-th sample IS the frozen design
    (same sinc, same window positions), normalised so that those samples sum to L."""
    fi = 352800.0
    fup = fi * L
    fo = float(out_rate)
    fp, fst = 0.227 * min(fo, fi), 0.55 * min(fo, fi)
    beta = 0.1102 * (120.0 - 8.7)
    N = P * L
    g = sg.firwin(dens * N - (dens - 1), 0.5 * (fp + fst), window=("kaiser", beta), fs=fup * dens)
    return g / g[::dens].sum() * L


def compose_polyphase(dsd_rate, r, hA_half):
    """The 48k cascade composed into ONE polyphase filter on the DSD bits (DSD64 / DSD128 input).

    Stage A (hA, at the bit rate) and stage B (its prototype g at 352.8 kHz * L) are two LTI filters; without the sampling at 352.8 kHz
    between them their cascade is the single filter p = g (*) upsample(hA) on the fine grid of Lp ticks per bit
    (Lp / Mp = out_rate / bit rate in lowest terms), and output m is

        y[m] = sum_j c[rho][j] * s[q + D - j],   Mp * m = Lp * q + rho,   c[rho][j] = p[Lp * j + rho] / MA

    (the sampling between the stages only adds what stage A lets through around the multiples of 352.8 kHz -- its stop band, -150 dB --
    folded onto stage B's stop band).  Where the fine grid is denser than stage B's own (DSD128 -> 96 kHz: 2x) the same Kaiser design is
    sampled more densely.  Every phase is normalised to unity DC gain and rounded to the 24-bit dyadic grid of the decimators."""
    MA = 8 * dsd_rate
    out_rate, L, P = r["out_rate"], r["L"], r["P"]
    from math import gcd
    fbit = 2822400 * dsd_rate
    gg = gcd(out_rate, fbit)
    Lp, Mp = out_rate // gg, fbit // gg
    dens = Lp * MA // L
    assert dens * L == Lp * MA and dens >= 1
    g = stage_b_dense(out_rate, L, P, dens)
    frozen = np.array(r["coef"], dtype=np.float64).reshape(L, P).T.reshape(-1)        # g[k * L + phase]
    assert np.abs(g[::dens] - frozen).max() < 1e-15, "the dense design must contain the frozen stage-B design"
    g = np.concatenate([g, np.zeros(dens * L * P - len(g))])
    hA = np.concatenate([hA_half[::-1], hA_half])
    NA = len(hA)
    up = np.zeros(Lp * (NA - 1) + 1)
    up[::Lp] = hA
    p = np.convolve(g, up)
    NP = MA * P + NA - 1
    assert len(p) == Lp * NP
    c = p.reshape(NP, Lp).T.copy()                          # c[rho][j]
    c /= c.sum(1)[:, None]
    S = int(np.floor(np.log2((2 ** 23 - 2 ** 16) / np.abs(c).max())))
    q0 = np.rint(c * 2.0 ** S).astype(np.int64)
    keep = np.nonzero(np.abs(q0).max(0) > POLY_TRIM)[0]
    lo, hi = int(keep[0]), int(keep[-1])
    c = c[:, lo:hi + 1]
    c = c / c.sum(1)[:, None]
    q = np.zeros(c.shape, dtype=np.int64)
    for ph in range(Lp):
        qq = np.rint(c[ph] * 2.0 ** S).astype(np.int64)
        resid = (1 << S) - int(qq.sum())
        frac = c[ph] * 2.0 ** S - qq
        order = np.argsort(-frac, kind="stable") if resid > 0 else np.argsort(frac, kind="stable")
        assert abs(resid) < len(qq)
        for i in range(abs(resid)):
            qq[order[i]] += 1 if resid > 0 else -1
        assert int(qq.sum()) == 1 << S
        q[ph] = qq
    assert np.abs(q).max() < 2 ** 23 - 2 ** 15 and int(np.abs(q).sum(1).max()) < 2 ** 31 - 2 ** 26
    D = MA - 1 - lo                                          # newest bit of output m: q_m + D (never ahead of the cascade's newest bit)
    assert D < 0
    return dict(name=f"P_{dsd_rate}_{out_rate}", dsd_rate=dsd_rate, out_rate=out_rate, Lp=Lp, Mp=Mp, NP=int(c.shape[1]), D=int(D), S=S,
                q=[int(v) for v in q.reshape(-1)], coef=[float(v) for v in c.reshape(-1)],
                method=f"A_M{MA} (*) B_{out_rate} on {Lp} ticks per bit, 24-bit grid, taps within {POLY_TRIM} units dropped at both ends")


def composed_rates(resamplers):
    """(dsd rate, stage B) pairs that get a composed one-pass table: DSD64 and DSD128 to every 48k multiple, DSD256 to 192 and 384 kHz (a window of
    1,100-1,550 bits; DSD256 -> 96 kHz would need 3,400, DSD512 6,800: those keep the two-kernel cascade)"""
    return [(rate, r) for rate in (1, 2, 4) for r in resamplers if rate < 4 or r["out_rate"] >= 192000]


def fmt_i32(q):
    return ", ".join(str(int(v)) for v in q)


def main():
    if "--regrid" in sys.argv:
        # keep every frozen design (filters/filter_tables.json) and only (re)derive what is computed FROM them: the stage-B integer grid
        with open(os.path.join(ROOT, "filters", "filter_tables.json")) as f:
            frozen = json.load(f)
        filters = frozen["filters"]
        resamplers = [dict(r, coef=[float.fromhex(x) for x in r["coef"]]) for r in frozen["resamplers"]]
        for r in resamplers:
            r["q"] = [int(v) for v in quantise_polyphase(np.array(r["coef"]).reshape(r["L"], r["P"])).reshape(-1)]
        with open(os.path.join(ROOT, "filters", "filter_taps_f64.json")) as f:
            f64 = json.load(f)
        for fl in filters:
            fl["q32"] = [int(v) for v in quantise_fine([float.fromhex(x) for x in f64[fl["name"]]], fl["S"], fl["q"])]
        polys = [compose_polyphase(rate, r, np.array([float.fromhex(x) for x in f64[f"A_M{8 * rate}"]])) for rate, r in composed_rates(resamplers)]
        write_tables(filters, resamplers, None, polys)
        return
    filters = []   # dicts: name, type, M, N, S, q(list), method

    unquantised = {}   # name -> the design's own f64 half taps (symmetrised, unity DC gain), before the 24-bit grid

    def add(name, ftype, M, h, method):
        S, q = quantise(h)
        hn = np.asarray(h, dtype=np.float64) / np.sum(h)
        unquantised[name] = [float(v).hex() for v in 0.5 * (hn[len(hn) // 2:] + hn[:len(hn) // 2][::-1])]
        full = np.concatenate([q[::-1], q]).astype(np.float64) * 2.0 ** -S
        w, mag = response_db(full, float(M))
        filters.append(dict(name=name, type=ftype, M=M, N=2 * len(q), S=S,
                            q=[int(v) for v in q], q32=[int(v) for v in quantise_fine(0.5 * (hn[len(hn) // 2:] + hn[:len(hn) // 2][::-1]), S, q)], method=method,
                            db_at_0p227=float(mag[np.argmin(abs(w - 0.227))]),
                            db_at_nyq=float(mag[np.argmin(abs(w - 0.5))]),
                            stop_max_db=float(mag[w >= 0.56].max())))
        print(f"{name:12s} M={M:3d} N={2*len(q):5d} S={S} {method}", file=sys.stderr)

    # E: equiripple, every M of the 44.1k family (src/main.rs:62, test_all_44k_mults.sh)
    for M in (8, 16, 32, 64, 128):
        h = design_equiripple(M, 0.227, 0.55, 0.01, -110.0, 100.0)
        add(f"E_M{M}", "E", M, h, "remez pass<=0.227Fo stop>=0.55Fo 0.01dB/-110dB")
    # X: stand-in for the XLD tables (which are not available): Kaiser-windowed sinc, DSD64 only
    for M in (8, 16, 32):
        add(f"X_M{M}", "X", M, design_kaiser(M, 12 * M, 0.40, 10.0),
            "kaiser beta=10 fc=0.40Fo (stand-in for XLD)")
    # D: the original dsd2pcm filter, DSD64 -> 352.8k only
    add("D_M8", "D", 8, np.concatenate([DSD2PCM_HTAPS[::-1], DSD2PCM_HTAPS]), "dsd2pcm 96-tap (Gesemann)")
    # C: Dolph-Chebyshev windowed sinc, DSD128 -> 352.8/176.4/88.2k
    for M in (16, 32, 64):
        add(f"C_M{M}", "C", M, design_cheby(M, 16 * M, 0.36, 120.0), "chebwin 120dB fc=0.36Fo")
    # A: first stage of the 48k cascade (-> 352.8 kHz whatever the DSD rate): very gentle,
    # only has to protect the bands that alias onto stage B's pass band (around k*352.8k).
    for M in (8, 16, 32, 64):
        h = design_equiripple(M, 0.07, 0.85, 0.001, -150.0, 1.0)
        add(f"A_M{M}", "A", M, h, "remez pass<=0.07Fi stop>=0.85Fi -150dB (48k cascade stage A)")

    # B: rational resamplers 352.8k -> 96k/192k/384k = L/147, prototype at 352.8k*L
    resamplers = []
    for out_rate, L in ((96000, 40), (192000, 80), (384000, 160)):
        fi = 352800.0
        fup = fi * L
        fo = float(out_rate)
        fp, fst = 0.227 * min(fo, fi), 0.55 * min(fo, fi)
        # Kaiser design (robust at these lengths): A = 120 dB
        A = 120.0
        dw = 2 * np.pi * (fst - fp) / fup
        P = int(np.ceil(((A - 7.95) / (2.285 * dw)) / L / 8.0) * 8)
        N = P * L
        beta = 0.1102 * (A - 8.7)
        g = sg.firwin(N, 0.5 * (fp + fst), window=("kaiser", beta), fs=fup)
        g = g / g.sum() * L
        # phase-major layout: coef[phase][k] = g[k*L + phase]
        poly = g.reshape(P, L).T.copy()
        resamplers.append(dict(name=f"B_{out_rate}", out_rate=out_rate, L=L, Mdn=147, P=P,
                               coef=[float(v) for v in poly.reshape(-1)], q=[int(v) for v in quantise_polyphase(poly).reshape(-1)],
                               method=f"kaiser 120dB pass {fp:.0f} stop {fst:.0f} Hz"))
        print(f"B_{out_rate} L={L} P={P} N={N}", file=sys.stderr)

    polys = [compose_polyphase(rate, r, np.array([float.fromhex(x) for x in unquantised[f"A_M{8 * rate}"]])) for rate, r in composed_rates(resamplers)]
    write_tables(filters, resamplers, unquantised, polys)


def write_tables(filters, resamplers, unquantised, polys):
    os.makedirs(os.path.join(ROOT, "filters"), exist_ok=True)
    # test data only (oracle/, tests/): what the 24-bit tap grid costs against the designs' f64 taps
    if unquantised is not None:
        with open(os.path.join(ROOT, "filters", "filter_taps_f64.json"), "w") as f:
            json.dump(unquantised, f, indent=0)
    with open(os.path.join(ROOT, "filters", "filter_tables.json"), "w") as f:
        json.dump(dict(filters=filters,
                       resamplers=[{k: (v if k != "coef" else [x.hex() for x in v])
                                    for k, v in r.items()} for r in resamplers],
                       polys=[{k: (v if k != "coef" else [x.hex() for x in v]) for k, v in pl.items()} for pl in polys]), f, indent=0)

    out = []
    out.append("/* GENERATED by tools/design_filters.py -- do not edit.\n"
               " * This build's own filter designs (the reference's tables live in the absent rdsd2pcm\n"
               " * crate).  Integer decimators: taps = q * 2^-S, 2nd half stored centre-outward\n"
               " * (as /root/reference/README.md:252 describes), sum(all taps) == 1 exactly.\n"
               " * Stage B of the 48k cascade: coefficient [phase][k] = q * 2^-T, every phase sums to 1 exactly\n"
               " * (coef = the f64 design they were rounded from, kept for tests).\n"
               " * Direct 48k tables (DSD64 / DSD128 input): the cascade's two designs composed into one polyphase filter on the bits,\n"
               " * y[m] = sum_j q[rho][j] 2^-S s[qm + D - j], Mp m = Lp qm + rho; every phase sums to 2^S exactly.\n"
               " * Shared DATA for the engine (dsd2dxd_amd/csrc) and the oracle (oracle/). */\n")
    out.append("#ifndef D2D_FILTER_TABLES_INC\n#define D2D_FILTER_TABLES_INC\n#include <stdint.h>\n")
    out.append("typedef struct { const char* name; char type; int M; int ntaps; int S; const int32_t* half; const int32_t* half32; } d2d_filter_def;\n")
    out.append("typedef struct { const char* name; int out_rate; int L; int Mdn; int P; const double* coef; int T; const int32_t* q; } d2d_resamp_def;\n")
    for i, fl in enumerate(filters):
        out.append(f"static const int32_t d2d_ftab_{i}[{len(fl['q'])}] = {{ {fmt_i32(fl['q'])} }};\n")
        out.append(f"static const int32_t d2d_ftab32_{i}[{len(fl['q32'])}] = {{ {fmt_i32(fl['q32'])} }};   /* the 32-bit grid: q32 * 2^-(S+8) */\n")
    out.append(f"static const d2d_filter_def D2D_FILTERS[{len(filters)}] = {{\n")
    for i, fl in enumerate(filters):
        out.append(f"  {{ \"{fl['name']}\", '{fl['type']}', {fl['M']}, {fl['N']}, {fl['S']}, d2d_ftab_{i}, d2d_ftab32_{i} }},\n")
    out.append("};\n")
    out.append(f"enum {{ D2D_NUM_FILTERS = {len(filters)} }};\n")
    for i, r in enumerate(resamplers):
        body = ", ".join(float(v).hex() for v in r["coef"])
        out.append(f"static const double d2d_rtab_{i}[{len(r['coef'])}] = {{ {body} }};\n")
        out.append(f"static const int32_t d2d_rqtab_{i}[{len(r['q'])}] = {{ {fmt_i32(r['q'])} }};\n")
    out.append(f"static const d2d_resamp_def D2D_RESAMPLERS[{len(resamplers)}] = {{\n")
    for i, r in enumerate(resamplers):
        out.append(f"  {{ \"{r['name']}\", {r['out_rate']}, {r['L']}, {r['Mdn']}, {r['P']}, d2d_rtab_{i}, {RESAMP_T}, d2d_rqtab_{i} }},\n")
    out.append("};\n")
    out.append(f"enum {{ D2D_NUM_RESAMPLERS = {len(resamplers)} }};\n")
    out.append("typedef struct { const char* name; int dsd_rate; int out_rate; int Lp; int Mp; int NP; int D; int S; const int32_t* q; } d2d_poly_def;\n")
    for i, pl in enumerate(polys):
        out.append(f"static const int32_t d2d_ptab_{i}[{len(pl['q'])}] = {{ {fmt_i32(pl['q'])} }};\n")
    out.append(f"static const d2d_poly_def D2D_POLYS[{len(polys)}] = {{\n")
    for i, pl in enumerate(polys):
        out.append(f"  {{ \"{pl['name']}\", {pl['dsd_rate']}, {pl['out_rate']}, {pl['Lp']}, {pl['Mp']}, {pl['NP']}, {pl['D']}, {pl['S']}, d2d_ptab_{i} }},\n")
    out.append("};\n")
    out.append(f"enum {{ D2D_NUM_POLYS = {len(polys)} }};\n#endif\n")
    with open(os.path.join(ROOT, "filters", "filter_tables.inc"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
