#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.

PARITY UNPINNED: the reference cannot run here (Rust, and its rdsd2pcm core is an absent submodule)
and holds no vectors of its own for this path, so these vectors come from THIS repository's CPU
oracle.  They pin the oracle and both GPU kernels against regressions and against each other; they
do not pin anything against the reference.  Inputs are stored with the outputs, so the tests do not
depend on the synthetic generator staying the same.

  python tests/golden/make_golden.py        (run from the repo root)

`fixture_*` entries are derived from the reference's own DSD fixtures when /root/reference/test is
present: only a sha256 and the first frames of the oracle's output are kept, not the fixture bytes.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pack_layout, random_bytes, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, engine kwargs, channels' content
    ("c2_dsd64_f32_352k8", dict(dsd_rate=1, output_rate=352800, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=32, dither="X", seed=0)),
    ("c3_dsd128_s24_88k2_tpdf", dict(dsd_rate=2, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=1)),
    ("c3_dsd128_s24_88k2_ns", dict(dsd_rate=2, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="N", seed=1)),
    ("c4_dsd64_s24_88k2_tpdf", dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=206)),
    ("c5_dsd512_8ch_s24_96k", dict(dsd_rate=8, output_rate=96000, channels=8, fmt="I", endianness="M", block_size=1, filter="E", bit_depth=24, dither="T", seed=5)),
    ("dsd64_s16_176k4_rect_xld", dict(dsd_rate=1, output_rate=176400, channels=2, fmt="I", endianness="M", block_size=4096, filter="X", bit_depth=16, dither="R", seed=2, level_db=-4.0)),
    ("dsd64_s20_352k8_dsd2pcm", dict(dsd_rate=1, output_rate=352800, channels=1, fmt="P", endianness="L", block_size=4096, filter="D", bit_depth=20, dither="T", seed=3, level_db=4.0)),
    ("dsd128_f32_176k4_cheby_fpd", dict(dsd_rate=2, output_rate=176400, channels=2, fmt="P", endianness="L", block_size=4096, filter="C", bit_depth=32, dither="F", seed=4)),
    ("dsd256_s24_192k", dict(dsd_rate=4, output_rate=192000, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=6)),
    # DSD64 / DSD128 -> 48k multiples: the composed polyphase tables (round 4)
    ("dsd64_s24_96k_tpdf", dict(dsd_rate=1, output_rate=96000, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=7)),
    ("dsd64_f32_192k_dff", dict(dsd_rate=1, output_rate=192000, channels=2, fmt="I", endianness="M", block_size=1, filter="E", bit_depth=32, dither="F", seed=8, level_db=-3.0)),
    ("dsd128_s16_384k_mono", dict(dsd_rate=2, output_rate=384000, channels=1, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=16, dither="R", seed=9)),
    ("dsd128_s24_96k_ns_3ch", dict(dsd_rate=2, output_rate=96000, channels=3, fmt="P", endianness="M", block_size=512, filter="E", bit_depth=24, dither="N", seed=10)),
    # round 4: a 5.1 stream (whole frames from one wave on the GPU) and the 32-bit tap grid (one pass on the GPU)
    ("dsd64_s24_88k2_6ch", dict(dsd_rate=1, output_rate=88200, channels=6, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="T", seed=11)),
    ("dsd64_s16_88k2_taps32", dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=16, dither="R", seed=12, level_db=-2.0, tap_bits=32)),
    ("dsd256_s24_1411k2", dict(dsd_rate=4, output_rate=1411200, channels=2, fmt="P", endianness="M", block_size=512, filter="E", bit_depth=24, dither="X", seed=0)),
]


def main():
    O.build()
    meta = {}
    arrays = {}
    for name, kw in CASES:
        C = kw["channels"]
        nbytes = 4096 * 3 + 40
        chans = []
        for c in range(C):
            if c % 3 == 0:
                chans.append(synth("sine", nbytes, seed=100 + c, freq=1000.0 + 111 * c, dsd_rate=kw["dsd_rate"], msb_first=kw["endianness"] == "M"))
            elif c % 3 == 1:
                chans.append(synth("pink", nbytes, seed=200 + c, amp=0.098, dsd_rate=kw["dsd_rate"], msb_first=kw["endianness"] == "M"))
            else:
                chans.append(random_bytes(nbytes, 300 + c))
        cuts = [0, 4096, 4096 * 2 + 7, nbytes]
        o = O.Oracle(**kw)
        outs = []
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            buf = pack_layout([ch[a:b] for ch in chans], kw["fmt"], kw["block_size"])
            pcm, fr = o.translate(buf)
            arrays[f"{name}/in{i}"] = buf
            arrays[f"{name}/out{i}"] = pcm[:fr * o.frame_bytes].copy()
            outs.append(fr)
        meta[name] = dict(kw=kw, calls=len(cuts) - 1, frames=outs, peak_dbfs=float(o.peak_dbfs()),
                          peaks=[o.peak(c) for c in range(C)])
    ref = "/root/reference/test"
    if os.path.isdir(ref):
        fx = {}
        for fname, kw, skip in [
            ("1kHz_mono_p.dsd", dict(dsd_rate=1, output_rate=88200, channels=1, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=24, dither="X"), 0),
            ("1kHz_stereo_i.dsd", dict(dsd_rate=1, output_rate=352800, channels=2, fmt="I", endianness="M", block_size=1, filter="D", bit_depth=24, dither="X"), 0),
            ("pinknoise_stereo_128.dsf", dict(dsd_rate=2, output_rate=176400, channels=2, fmt="P", endianness="L", block_size=4096, filter="C", bit_depth=24, dither="X"), 92),
            ("impulse_mono_toggle.dsd", dict(dsd_rate=1, output_rate=352800, channels=1, fmt="P", endianness="M", block_size=4096, filter="E", bit_depth=32, dither="X"), 0),
        ]:
            raw = np.fromfile(os.path.join(ref, fname), dtype=np.uint8)[skip:]
            raw = raw[:kw["channels"] * 4096 * 16]
            o = O.Oracle(**kw)
            pcm, fr = o.translate(raw)
            pcm = pcm[:fr * o.frame_bytes]
            fx[fname] = dict(kw=kw, skip=skip, nbytes=int(raw.size), frames=int(fr), sha256=hashlib.sha256(pcm.tobytes()).hexdigest(),
                             head=pcm[:96].tolist(), peak_dbfs=float(o.peak_dbfs()))
        meta["_fixtures"] = fx
    np.savez_compressed(os.path.join(HERE, "golden_vectors.npz"), **arrays)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays,", len(meta), "cases")


if __name__ == "__main__":
    main()
