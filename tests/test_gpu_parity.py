"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): integer output bit-exact for identical dither seed; float output
within 1e-6 RMS of the f64 CPU path.  The taps are dyadic (q*2^-S), so the FIR sum is exact in both
and the float output is asserted bit-identical as well; the 1e-6 RMS bound is checked alongside.
"""
import os

import numpy as np
import pytest

from helpers import decode_pcm, pack_layout, random_bytes, synth

pytestmark = pytest.mark.gpu

FLOAT_RMS_TOL = 1e-6

# (dsd_rate, out_rate, filter) -- the matrix of test_all_44k_mults.sh / test_all_48k_mults.sh plus
# the filter-availability list of src/main.rs:62-67
RATE_MATRIX = [
    (1, 88200, "E"), (1, 176400, "E"), (1, 352800, "E"),
    (2, 88200, "E"), (2, 176400, "E"), (2, 352800, "E"), (2, 705600, "E"),
    (4, 88200, "E"), (4, 176400, "E"), (4, 352800, "E"), (4, 705600, "E"), (4, 1411200, "E"),
    (8, 352800, "E"),
    (1, 88200, "X"), (1, 176400, "X"), (1, 352800, "X"), (1, 352800, "D"),
    (2, 88200, "C"), (2, 176400, "C"), (2, 352800, "C"),
    (1, 96000, "E"), (1, 192000, "E"), (1, 384000, "E"),
    (2, 96000, "E"), (2, 192000, "E"), (2, 384000, "E"),
    (4, 96000, "E"), (4, 192000, "E"), (4, 384000, "E"),
    (8, 96000, "E"),
]


def run_pair(engine_lib, oracle_mod, bufs, kw, kernel=0, debug=0):
    """bufs: list of call buffers fed in sequence.  Returns (gpu_bytes, oracle_bytes, gpu_engine, oracle).
    debug: d2d_params.debug_flags (engine_lib.DBG_*) -- a diagnostic route that must give the same bytes."""
    e = engine_lib.Engine(kernel=kernel, debug=debug, **kw)
    o = oracle_mod.Oracle(**kw)
    g_all, o_all = [], []
    for b in bufs:
        g, gf = e.translate(b)
        r, rf = o.translate(b)
        assert gf == rf
        g_all.append(g.copy())
        o_all.append(r[:rf * o.frame_bytes].copy())
    return np.concatenate(g_all), np.concatenate(o_all), e, o


KERNELS = [pytest.param(1, id="lut"), pytest.param(2, id="mfma")]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dsd_rate,out_rate,filt", RATE_MATRIX)
def test_rate_matrix_int24_tpdf(engine_lib, oracle_mod, dsd_rate, out_rate, filt, kernel):
    nbytes = 4096 * 6
    chans = [synth("sine", nbytes, seed=1, dsd_rate=dsd_rate), synth("pink", nbytes, seed=2, amp=0.098, dsd_rate=dsd_rate)]
    buf = pack_layout(chans, "P", 4096)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096,
              filter=filt, bit_depth=24, dither="T", seed=7)
    g, r, e, o = run_pair(engine_lib, oracle_mod, [buf[:4096 * 2 * 2], buf[4096 * 2 * 2:]], kw, kernel)
    assert e.info()["kernel"] == kernel
    assert g.size == r.size and g.size > 0
    assert np.array_equal(g, r)
    for c in range(2):
        assert e.peak(c) == o.peak(c)


@pytest.mark.parametrize("bits,dither", [(16, "T"), (16, "R"), (16, "X"), (20, "T"), (20, "X"), (24, "R"), (24, "X"),
                                          (24, "F"), (32, "F"), (32, "X"), (32, "T")])
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("out_rate", [88200, 96000])
def test_depths_and_dithers(engine_lib, oracle_mod, bits, dither, out_rate, kernel):
    nbytes = 4096 * 4
    chans = [synth("sine", nbytes, seed=3), synth("pink", nbytes, seed=4, amp=0.098)]
    buf = pack_layout(chans, "P", 4096)
    kw = dict(dsd_rate=1, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096,
              filter="E", bit_depth=bits, dither=dither, seed=11, level_db=-3.0)
    g, r, e, o = run_pair(engine_lib, oracle_mod, [buf], kw, kernel)
    assert np.array_equal(g, r)
    if bits == 32:
        a, b = decode_pcm(g, 32, 2).astype(np.float64), decode_pcm(r, 32, 2).astype(np.float64)
        assert np.sqrt(np.mean((a - b) ** 2)) <= FLOAT_RMS_TOL


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("out_rate,bits,level", [(88200, 24, 70.0), (88200, 16, 12.0), (96000, 24, 70.0), (352800, 20, 200.0), (192000, 16, 6.0)])
def test_overload_clips_like_the_oracle(engine_lib, oracle_mod, out_rate, bits, level, kernel):
    """full-scale input and absurd gains: the requantiser saturates (the f64 -> i32 conversion first, then
    the clip to the sample range) exactly where the oracle does, on both sides"""
    nbytes = 4096 * 3
    a = np.concatenate([np.full(4096, 0xFF, np.uint8), np.zeros(4096, np.uint8), synth("sine", 4096, seed=1, amp=0.9)])
    b = np.concatenate([synth("sine", 4096, seed=2, amp=0.9), np.full(4096, 0xFF, np.uint8), np.zeros(4096, np.uint8)])
    buf = pack_layout([a[:nbytes], b[:nbytes]], "P", 4096)
    kw = dict(dsd_rate=1, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096,
              filter="E", bit_depth=bits, dither="T", seed=3, level_db=level)
    g, r, e, o = run_pair(engine_lib, oracle_mod, [buf], kw, kernel)
    assert np.array_equal(g, r)
    pcm = decode_pcm(g, bits, 2)
    lim = 1 << ((20 if bits == 20 else bits) - 1)
    if bits == 20:
        pcm = pcm >> 4
    assert pcm.max() == lim - 1 and pcm.min() == -lim          # both rails are reached
    assert e.peak_dbfs() == o.peak_dbfs()


@pytest.mark.parametrize("bits,dither,out_rate", [(16, "T", 88200), (20, "R", 176400), (32, "F", 352800), (24, "T", 96000), (32, "X", 192000)])
@pytest.mark.parametrize("channels,fmt", [(5, "P"), (6, "I")])
def test_multichannel_depths(engine_lib, oracle_mod, bits, dither, out_rate, channels, fmt):
    """every sample format through the per-pair output path of the MFMA kernel (frames of one pair land
    inside the file's wider frames), with a ragged tail"""
    nbytes = 4096 * 3 + 700
    chans = [random_bytes(nbytes, 300 + c) for c in range(channels)]
    endian = "L" if fmt == "P" else "M"
    cuts = [0, 4096 * 2, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], fmt, 4096 if fmt == "P" else 1) for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(dsd_rate=1, output_rate=out_rate, channels=channels, fmt=fmt, endianness=endian, block_size=4096 if fmt == "P" else 1,
              filter="E", bit_depth=bits, dither=dither, seed=9, level_db=-1.5)
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, 2)
    assert np.array_equal(g, r)
    assert e.peak_dbfs() == o.peak_dbfs()


@pytest.mark.parametrize("fmt,endian,block,channels", [
    ("P", "L", 4096, 1), ("P", "L", 4096, 2), ("P", "M", 4096, 2), ("I", "M", 4096, 2), ("I", "L", 1, 2),
    ("P", "L", 512, 2), ("P", "M", 24, 3), ("I", "M", 1, 6), ("P", "L", 4096, 8), ("P", "L", 100, 2),
    # multichannel files run one block column per channel PAIR (odd counts leave a single), and
    # byte-interleaved ones pass through the de-interleave kernel first
    ("I", "M", 1, 3), ("I", "M", 1, 5), ("I", "L", 1, 8), ("P", "L", 4096, 5), ("P", "M", 4096, 6),
])
@pytest.mark.parametrize("kernel", KERNELS)
def test_layouts_and_ragged_calls(engine_lib, oracle_mod, fmt, endian, block, channels, kernel):
    nbytes = 4096 * 3 + 52
    chans = [random_bytes(nbytes, 100 + c) for c in range(channels)]
    # ragged call sizes, including a zero-length call and calls that are not a multiple of M/8
    cuts = [0, 4096, 4096, 4099, 4800, 9000, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], fmt, block) for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(dsd_rate=1, output_rate=88200, channels=channels, fmt=fmt, endianness=endian, block_size=block,
              filter="E", bit_depth=24, dither="T", seed=5)
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, kernel)
    assert np.array_equal(g, r)
    assert e.peak_dbfs() == o.peak_dbfs()


@pytest.mark.parametrize("kernel", KERNELS)
def test_batch_of_files_matches_per_file(engine_lib, oracle_mod, kernel):
    import torch
    n_files, nbytes = 5, 4096 * 8
    kw = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096,
              filter="E", bit_depth=24, dither="T", seed=9)
    e = engine_lib.Engine(n_files=n_files, kernel=kernel, **kw)
    files = [pack_layout([synth("sine", nbytes, seed=10 + f, freq=500.0 * (f + 1)), synth("pink", nbytes, seed=20 + f, amp=0.098)],
                         "P", 4096) for f in range(n_files)]
    d_in = [torch.from_numpy(b).cuda() for b in files]
    nfr = e.next_frames(nbytes)
    d_out = [torch.zeros(nfr * e.frame_bytes, dtype=torch.uint8, device="cuda") for _ in range(n_files)]
    ios = (engine_lib.FileIO * n_files)()
    for f in range(n_files):
        ios[f].dsd = d_in[f].data_ptr(); ios[f].bytes_per_channel = nbytes
        ios[f].pcm = d_out[f].data_ptr(); ios[f].pcm_capacity_bytes = d_out[f].numel()
    e.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for f in range(n_files):
        o = oracle_mod.Oracle(**kw)
        r, rf = o.translate(files[f])
        assert ios[f].frames_out == rf
        assert np.array_equal(d_out[f].cpu().numpy(), r)
        assert e.peak(0, file=f) == o.peak(0) and e.peak(1, file=f) == o.peak(1)


@pytest.mark.parametrize("out_rate,dither,bits", [(88200, "T", 24), (96000, "R", 16), (88200, "N", 24)])
def test_batch_of_interleaved_files(engine_lib, oracle_mod, out_rate, dither, bits):
    """A batch of byte-interleaved files (the DFF layout) of different lengths through the planar-copy pre-pass, two calls in a row
    (the second reuses the planar copy), 44.1k and 48k family, noise-shaped included."""
    import torch
    n_files, chans = 7, 3
    lens = [4096 * 6 + 64 * f for f in range(n_files)]
    kw = dict(dsd_rate=1, output_rate=out_rate, channels=chans, fmt="I", endianness="M", block_size=4096,
              filter="E", bit_depth=bits, dither=dither, seed=31)
    e = engine_lib.Engine(n_files=n_files, kernel=2, **kw)
    streams = [[synth("sine" if c != 1 else "pink", lens[f], seed=40 + 7 * f + c, amp=0.3 if c != 1 else 0.098, msb_first=True) for c in range(chans)]
               for f in range(n_files)]
    oracles = [oracle_mod.Oracle(**kw) for _ in range(n_files)]
    for part in range(2):
        cut = [(0, 4096 * 3 + 128), (4096 * 3 + 128, None)][part]
        bufs = [pack_layout([ch[cut[0]:cut[1]] for ch in streams[f]], "I", 1) for f in range(n_files)]
        d_in = [torch.from_numpy(b).cuda() for b in bufs]
        ios = (engine_lib.FileIO * n_files)()
        d_out = []
        for f in range(n_files):
            bpc = len(bufs[f]) // chans
            nfr = e.next_frames(bpc, file=f)
            d_out.append(torch.zeros(max(nfr * e.frame_bytes, 16), dtype=torch.uint8, device="cuda"))
            ios[f].dsd = d_in[f].data_ptr(); ios[f].bytes_per_channel = bpc
            ios[f].pcm = d_out[f].data_ptr(); ios[f].pcm_capacity_bytes = d_out[f].numel()
        e.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for f in range(n_files):
            r, rf = oracles[f].translate(bufs[f])
            assert ios[f].frames_out == rf
            assert np.array_equal(d_out[f].cpu().numpy()[:rf * e.frame_bytes], r[:rf * e.frame_bytes])
    for f in range(n_files):
        for c in range(chans):
            assert e.peak(c, file=f) == oracles[f].peak(c)


@pytest.mark.parametrize("kernel", KERNELS)
def test_known_answers_on_device(engine_lib, kernel):
    """All-ones DSD -> +full scale (taps sum to exactly 1), all-zeros -> -full scale; first frames carry
    the idle-history transient."""
    e = engine_lib.Engine(dsd_rate=1, output_rate=88200, channels=1, fmt="P", endianness="M", block_size=4096,
                          filter="E", bit_depth=24, dither="X", kernel=kernel)
    ones = np.full(4096 * 4, 0xFF, dtype=np.uint8)
    g, fr = e.translate(ones)
    v = decode_pcm(g, 24, 1)[:, 0]
    assert (v[200:] == 8388607).all()
    e.reset()
    g, fr = e.translate(np.zeros(4096 * 4, dtype=np.uint8))
    assert (decode_pcm(g, 24, 1)[200:, 0] == -8388608).all()


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dsd_rate,out_rate,bits,level", [(1, 88200, 16, 0.0), (2, 88200, 24, -2.0), (2, 88200, 24, 0.0), (1, 176400, 20, 0.0), (1, 352800, 16, 30.0)])
def test_noise_shaped_dither_matches_the_oracle(engine_lib, oracle_mod, dsd_rate, out_rate, bits, level, kernel):
    """the 'N' extension (BASELINE config 3 names a noise-shaped variant): an error-feedback loop per
    channel, carried across calls and files, identical to the oracle's sequential loop, clipping included.
    (2, 88200, 24, 0 dB) is BASELINE config 3 itself: the FIR kernel's scratch flavour + the stereo shaper's int32 recurrence."""
    nbytes = 4096 * 4 + 90
    chans = [synth("sine", nbytes, seed=1, dsd_rate=dsd_rate, amp=0.5), synth("pink", nbytes, seed=2, amp=0.098, dsd_rate=dsd_rate)]
    cuts = [0, 4096, 4096 * 3, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], "P", 4096) for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096,
              filter="E", bit_depth=bits, dither="N", seed=21, level_db=level)
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, kernel)
    assert np.array_equal(g, r)
    assert e.peak_dbfs() == o.peak_dbfs()
    # the same conversion with plain TPDF differs (the loop is really in the path) ...
    kw_t = dict(kw, dither="T")
    g_t, _, _, _ = run_pair(engine_lib, oracle_mod, bufs, kw_t, kernel)
    assert not np.array_equal(g, g_t)
    # ... and float output has nothing to shape
    e32 = engine_lib.Engine(n_files=1, kernel=kernel, **dict(kw, bit_depth=32))
    x32 = engine_lib.Engine(n_files=1, kernel=kernel, **dict(kw, bit_depth=32, dither="X"))
    assert np.array_equal(e32.translate(bufs[0])[0], x32.translate(bufs[0])[0])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dsd_rate,out_rate,bits,level,channels", [(1, 96000, 24, 0.0, 2), (2, 192000, 16, -1.5, 2), (1, 384000, 20, 0.0, 3)])
def test_noise_shaped_dither_on_the_48k_family(engine_lib, oracle_mod, dsd_rate, out_rate, bits, level, channels, kernel):
    """'N' behind the two-stage path (round 3): stage B hands its outputs to the shaper as f64 (y = (double)v * 2^-(S+T), the oracle's
    number), the loop runs on the output index m with its 8192-output segments and its two carried errors: ragged calls, a call that
    ends inside a segment, odd channel count -- all equal to the oracle"""
    nbytes = 4096 * 12 * dsd_rate + 77
    chans = [synth("sine" if c % 2 == 0 else "pink", nbytes, seed=31 + c, dsd_rate=dsd_rate, amp=0.45 if c % 2 == 0 else 0.098) for c in range(channels)]
    cuts = [0, 4096 * 3, 4096 * 3 + 1000, 4096 * 9, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], "P", 4096) for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=channels, fmt="P", endianness="L", block_size=4096,
              filter="E", bit_depth=bits, dither="N", seed=8, level_db=level)
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, kernel)
    assert g.size == r.size and g.size > 0
    assert np.array_equal(g, r)
    assert e.peak_dbfs() == o.peak_dbfs()
    one, _, _, _ = run_pair(engine_lib, oracle_mod, [pack_layout(chans, "P", 4096)], kw, kernel)
    assert np.array_equal(one, g)
    g_t, _, _, _ = run_pair(engine_lib, oracle_mod, bufs, dict(kw, dither="T"), kernel)
    assert not np.array_equal(g, g_t)


@pytest.mark.parametrize("kernel", KERNELS)
def test_noise_shaped_segments_restart_and_carry(engine_lib, oracle_mod, kernel):
    """the shaper restarts at output indices that are multiples of 8192: calls that end inside a segment,
    start exactly on a boundary, or span several boundaries all match the oracle"""
    nbytes = 65536 * 2 + 5000                             # M = 8: one output per byte, 136072 outputs per channel
    chans = [synth("sine", nbytes, seed=3, msb_first=True, amp=0.4), synth("pink", nbytes, seed=4, amp=0.098, msb_first=True)]
    cuts = [0, 8191, 8192, 30000, 32768, 32769, 49151, 65536, 65537, 131071, 135000, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], "I", 1) for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(dsd_rate=1, output_rate=352800, channels=2, fmt="I", endianness="M", block_size=1,
              filter="E", bit_depth=16, dither="N", seed=5)
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, kernel)
    assert np.array_equal(g, r)
    one, r1, _, _ = run_pair(engine_lib, oracle_mod, [pack_layout(chans, "I", 1)], kw, kernel)
    assert np.array_equal(one, g)                          # one call (nine segments side by side) == nine calls


@pytest.mark.parametrize("bits", [24, 16])
def test_noise_shaped_many_segments_many_waves(engine_lib, oracle_mod, bits):
    """171 segments per channel = six waves in two blocks of the stereo noise-shaping kernel (a wave's lanes hand their frames to
    each other through LDS and store them along the segments): odd call sizes, a call that ends inside a group, segments that
    finish at different rounds inside one wave -- all equal to the oracle's sequential loop, and one call == five calls"""
    nbytes = 8192 * 171 - 3001                            # M = 8: one output per byte
    chans = [synth("sine", nbytes, seed=13, msb_first=True, amp=0.45), synth("pink", nbytes, seed=14, amp=0.098, msb_first=True)]
    cuts = [0, 300001, 300004, 8192 * 100 + 5, nbytes - 7, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], "I", 1) for a, b in zip(cuts[:-1], cuts[1:])]
    kw = dict(dsd_rate=1, output_rate=352800, channels=2, fmt="I", endianness="M", block_size=1,
              filter="E", bit_depth=bits, dither="N", seed=77)
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, 2)
    assert g.size == r.size and g.size == nbytes * 2 * (bits // 8)
    assert np.array_equal(g, r)
    assert e.peak_dbfs() == o.peak_dbfs()
    one, _, _, _ = run_pair(engine_lib, oracle_mod, [pack_layout(chans, "I", 1)], kw, 2)
    assert np.array_equal(one, g)


@pytest.mark.parametrize("chain,bits", [("mx", 24), ("mx", 16), ("mx", 32), ("dense", 24), ("dense", 16), ("dense", 32)],
                         ids=["fp6_chain", "fp6_chain_16bit", "fp6_chain_float", "dense_chain", "dense_chain_16bit", "dense_chain_float"])
@pytest.mark.parametrize("dither", ["T", "R", "X"])
@pytest.mark.parametrize("dsd_rate,out_rate,filt", [(1, 88200, "E"), (1, 88200, "X"), (2, 88200, "E"), (2, 176400, "C"), (4, 176400, "E"),
                                                    (1, 176400, "E"), (1, 352800, "E"), (1, 176400, "X"), (2, 352800, "C"), (1, 352800, "D"), (4, 88200, "E")])
def test_pipelined_stereo_kernel_fast_and_careful_tiles(engine_lib, oracle_mod, dsd_rate, out_rate, filt, dither, chain, bits):
    """Stereo 24-bit (and 16-bit, and float without the float dither) at 0 dB runs a software-pipelined kernel -- d2d_fir_mx_kernel (fp6 x fp4
    matrix-core chain, M = 32 and 64) or d2d_fir_mfma3_kernel (int8 chain; every M up to 64 with the D2D_DBG_NO_MX flag): the requantiser rides on the next chain in its branch-free form and a
    tile that could clip, holds an exact rounding tie or is cut short by the end of the call is redone sample by sample.
    Full-scale stretches (all-ones / all-zeros bytes clip at both rails), quiet stretches, ragged call sizes and a short last
    call exercise those paths (an exact tie under triangular dither is a 2^-16 event per sample: likely here, certain in
    tests/test_gpu_fullsize.py)."""
    M = 2822400 * dsd_rate // out_rate
    if chain == "mx" and M < 32:
        pytest.skip("the fp6 chain serves M = 32, 64 and 128 (at M = 8 and 16 the groups of six phases do not share tap fragments)")
    if chain != "mx" and M == 128:
        pytest.skip("M = 128: the fp6 kernel only (the int8 kernels' tap table does not fit next to eight waves)")
    debug = 0 if chain == "mx" else engine_lib.DBG_NO_MX
    rng = np.random.default_rng(5)
    nbytes = 4096 * 40 * dsd_rate
    chans = []
    for c in range(2):
        x = synth("sine", nbytes, seed=20 + c, dsd_rate=dsd_rate).copy()
        for _ in range(6):                                   # full-scale stretches: +1.0 and -1.0 for a few thousand bits
            a = int(rng.integers(0, nbytes - 3000))
            x[a:a + int(rng.integers(200, 3000))] = 0xFF if rng.integers(0, 2) else 0x00
        a = int(rng.integers(0, nbytes - 9000))
        x[a:a + 8192] = 0x69                                 # an idle-like pattern: output near zero
        chans.append(x)
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096,
              filter=filt, bit_depth=bits, dither=dither, seed=99)
    cuts = [0, 4096 * 7, 4096 * 7 + 4096 * 20, nbytes - 4096, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], "P", 4096) for a, b in zip(cuts[:-1], cuts[1:])]
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, 2, debug)
    targs = [t.strip() for t in e.kernel_name().split("<")[1].rstrip(">").split(",")]
    if chain == "mx":
        assert e.kernel_name().startswith("d2d_fir_mx_kernel")                             # <MB, taps, groups, dither kind, bytes per sample>
        assert targs[0] == str(M // 8) and targs[4] == str(bits // 8)
    else:
        assert e.kernel_name().startswith("d2d_fir_mfma3_kernel")                          # <MB, NPG, taps (0 = dense chain), dither kind, bytes per sample>
        assert targs[2] == "0" and targs[4] == str(bits // 8)
    assert g.size == r.size and g.size > 0
    assert np.array_equal(g, r)
    pcm = decode_pcm(g, bits, 2)
    if bits == 32:
        assert pcm.max() >= 1.0 and pcm.min() <= -1.0                                      # full scale both ways (float does not clip)
    else:
        assert pcm.max() == (1 << (bits - 1)) - 1 and pcm.min() == -(1 << (bits - 1))      # both rails were reached
    for c in range(2):
        assert e.peak(c) == o.peak(c)


@pytest.mark.parametrize("channels,dsd_rate,out_rate,dither,endian", [(8, 8, 96000, "T", "M"), (4, 8, 96000, "X", "L"), (8, 4, 96000, "R", "M"),
                                                                      (4, 2, 88200, "N", "M"), (8, 1, 88200, "N", "L")])
def test_multichannel_interleaved_input_is_deinterleaved_inside_the_fir_kernel(engine_lib, oracle_mod, channels, dsd_rate, out_rate, dither, endian):
    """byte-interleaved 4- and 8-channel streams (DFF) into the stage-A / noise-shaper scratch through d2d_fir_mx_kernel (M = 32, 64): the
    kernel's staging de-interleaves -- a block per (file, tile), one wave per channel pair, two block barriers per tile; tiles at the
    call's edges (history in front, the ragged end) are gathered byte by byte.  Ragged calls, two files of different length, equal to
    the oracle and to the pre-pass route (the D2D_DBG_NO_COOP flag)."""
    nbytes = 4096 * 9 * dsd_rate // 2 + 333
    files = []
    for f in range(2):
        n = nbytes - 1500 * f
        files.append([synth("sine" if (c + f) % 3 else "pink", n, seed=50 + 10 * f + c, dsd_rate=dsd_rate, msb_first=endian == "M",
                            amp=0.4 if (c + f) % 3 else 0.098) for c in range(channels)])
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=channels, fmt="I", endianness=endian, block_size=1,
              filter="E", bit_depth=24, dither=dither, seed=17)
    cuts = [0, 1000, 4096 * 2 + 7, nbytes - 1500 - 40, nbytes]
    outs = {}
    for nocoop in ("0", "1"):
        import torch
        e = engine_lib.Engine(n_files=2, kernel=2, debug=engine_lib.DBG_NO_COOP if nocoop == "1" else 0, **kw)
        got = [[], []]
        for a, b in zip(cuts[:-1], cuts[1:]):
            bufs = [pack_layout([ch[min(a, len(ch)):min(b, len(ch))] for ch in files[f]], "I", 1) for f in range(2)]
            lens = [buf.size // channels for buf in bufs]
            d_in = [torch.from_numpy(buf).cuda() if buf.size else torch.zeros(16, dtype=torch.uint8, device="cuda") for buf in bufs]
            d_out = [torch.zeros(e.next_frames(n, file=i) * e.frame_bytes + 16, dtype=torch.uint8, device="cuda") for i, n in enumerate(lens)]
            ios = (engine_lib.FileIO * 2)()
            for i, n in enumerate(lens):
                ios[i].dsd = d_in[i].data_ptr(); ios[i].bytes_per_channel = n
                ios[i].pcm = d_out[i].data_ptr(); ios[i].pcm_capacity_bytes = d_out[i].numel()
            e.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            for f in range(2):
                got[f].append(d_out[f][:ios[f].frames_out * e.frame_bytes].cpu().numpy())
        outs[nocoop] = [np.concatenate(g) for g in got]
        peaks = [[e.peak(c, f) for c in range(channels)] for f in range(2)]
        if nocoop == "0":
            assert "d2d_fir_mx_kernel" in e.kernel_name()
            for f in range(2):
                o = oracle_mod.Oracle(**kw)
                want = []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    w, fr = o.translate(pack_layout([ch[min(a, len(ch)):min(b, len(ch))] for ch in files[f]], "I", 1))
                    want.append(w[:fr * channels * 3])
                assert np.array_equal(outs["0"][f], np.concatenate(want))
                assert peaks[f] == [o.peak(c) for c in range(channels)]
    for f in range(2):
        assert np.array_equal(outs["0"][f], outs["1"][f])


@pytest.mark.parametrize("dsd_rate,out_rate,bits,dither,endian,nbytes", [
    (1, 88200, 24, "T", "M", 4096 * 9 + 333), (1, 88200, 16, "R", "L", 4096 * 9 + 333), (1, 88200, 32, "X", "M", 4096 * 5 + 17),
    (2, 88200, 24, "T", "M", 4096 * 18 + 100), (2, 176400, 16, "X", "L", 4096 * 9 + 333), (4, 176400, 24, "R", "M", 4096 * 30 + 5),
    (1, 88200, 24, "T", "M", 6_000_000 + 123), (2, 88200, 24, "R", "M", 9_000_000 + 77),
    (1, 352800, 24, "T", "M", 4096 * 5 + 333), (1, 352800, 32, "X", "L", 4096 * 3 + 17), (1, 176400, 16, "R", "M", 4096 * 9 + 1), (2, 352800, 24, "X", "M", 4096 * 9 + 100),
    (2, 705600, 16, "T", "L", 4096 * 7 + 5), (1, 352800, 24, "T", "M", 3_000_000 + 123), (1, 176400, 24, "R", "M", 4_000_000 + 9), (1, 88200, 24, "T", "m3", 4096 * 9 + 333),
    (1, 88200, 16, "R", "m3", 5_000_000 + 3),
    (1, 96000, 24, "T", "M", 4096 * 9 + 333), (1, 192000, 16, "R", "L", 3_000_000 + 11), (2, 384000, 24, "X", "M", 4096 * 20 + 7), (1, 352800, 24, "N", "M", 2_000_000 + 5),
    (1, 176400, 16, "N", "L", 4096 * 9 + 1),
    (1, 88200, 24, "N", "M", 4096 * 9 + 333), (2, 88200, 16, "N", "L", 8_000_000 + 9), (4, 192000, 24, "T", "M", 4096 * 30 + 3), (8, 96000, 24, "R", "M", 12_000_000 + 1)])
def test_interleaved_stereo_is_deinterleaved_inside_the_fir_kernel(engine_lib, oracle_mod, dsd_rate, out_rate, bits, dither, endian, nbytes):
    """byte-interleaved STEREO (DFF files, the reference CLI's default -f I) into frames through d2d_fir_mx_kernel (M = 32, 64) and
    d2d_fir_mfma3_kernel (M = 8, 16; M = 32 with the D2D_DBG_NO_MX flag: "m3"), and at DSD64 / DSD128 -> 48k multiples d2d_fir_px_kernel: the wave that converts the pair pulls the channels apart inside its
    staging (pieces fetched once, channel 1's bytes parked in registers for one region); tiles at the call's edges are gathered byte by
    byte.  Ragged calls, two files of different length, waves that walk several tiles (the long cases); equal to the oracle and to the
    pre-pass route (the D2D_DBG_NO_COOP flag)."""
    import torch
    no_mx = 0
    if endian == "m3":
        endian = "M"
        no_mx = engine_lib.DBG_NO_MX
    files = []
    for f in range(2):
        n = nbytes - 1501 * f
        files.append([synth("sine" if (c + f) % 2 else "pink", n, seed=150 + 10 * f + c, dsd_rate=dsd_rate, msb_first=endian == "M",
                            amp=0.45 if (c + f) % 2 else 0.098) for c in range(2)])
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="I", endianness=endian, block_size=1,
              filter="E", bit_depth=bits, dither=dither, seed=23)
    cuts = [0, 1000, 4096 * 2 + 7, nbytes - 1501 - 40, nbytes]
    outs = {}
    for nocoop in ("0", "1"):
        e = engine_lib.Engine(n_files=2, kernel=2, debug=no_mx | (engine_lib.DBG_NO_COOP if nocoop == "1" else 0), **kw)
        fb = e.frame_bytes
        got = [[], []]
        for a, b in zip(cuts[:-1], cuts[1:]):
            bufs = [pack_layout([ch[min(a, len(ch)):min(b, len(ch))] for ch in files[f]], "I", 1) for f in range(2)]
            lens = [buf.size // 2 for buf in bufs]
            d_in = [torch.from_numpy(buf).cuda() if buf.size else torch.zeros(16, dtype=torch.uint8, device="cuda") for buf in bufs]
            d_out = [torch.zeros(e.next_frames(n, file=i) * fb + 16, dtype=torch.uint8, device="cuda") for i, n in enumerate(lens)]
            ios = (engine_lib.FileIO * 2)()
            for i, n in enumerate(lens):
                ios[i].dsd = d_in[i].data_ptr(); ios[i].bytes_per_channel = n
                ios[i].pcm = d_out[i].data_ptr(); ios[i].pcm_capacity_bytes = d_out[i].numel()
            e.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            for f in range(2):
                got[f].append(d_out[f][:ios[f].frames_out * fb].cpu().numpy())
        outs[nocoop] = [np.concatenate(g) for g in got]
        peaks = [[e.peak(c, f) for c in range(2)] for f in range(2)]
        if nocoop == "0":
            M = 2822400 * dsd_rate // out_rate if out_rate % 44100 == 0 else 8 * dsd_rate          # (48k family: stage A decimates to 352.8 kHz)
            M = min(M, 64)
            composed = out_rate % 48000 == 0 and (dsd_rate <= 2 or (dsd_rate == 4 and out_rate >= 192000))   # the composed 48k tables: one polyphase pass (d2d_kernels_px.hip)
            assert ("d2d_fir_px_kernel" if composed else "d2d_fir_mx_kernel" if M >= 32 and not no_mx else "d2d_fir_mfma3_kernel") in e.kernel_name()
            for f in range(2):
                o = oracle_mod.Oracle(**kw)
                want = []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    w, fr = o.translate(pack_layout([ch[min(a, len(ch)):min(b, len(ch))] for ch in files[f]], "I", 1))
                    want.append(w[:fr * fb])
                assert np.array_equal(outs["0"][f], np.concatenate(want)), f
                assert peaks[f] == [o.peak(c) for c in range(2)]
    for f in range(2):
        assert np.array_equal(outs["0"][f], outs["1"][f])


@pytest.mark.parametrize("dsd_rate,out_rate", [(1, 352800), (1, 176400), (2, 352800), (2, 705600), (4, 1411200), (1, 88200), (2, 88200), (2, 176400), (4, 176400)])
@pytest.mark.parametrize("bits,dither,level", [(24, "T", -3.0), (16, "R", 4.0), (24, "X", -0.5), (32, "X", -4.0), (16, "T", 20.0), (24, "R", -60.0),
                                               (20, "T", 0.0), (20, "R", 20.0), (20, "X", -4.0), (32, "F", 0.0), (32, "F", -4.0)])
def test_level_in_db_inside_the_pipelined_kernel(engine_lib, oracle_mod, dsd_rate, out_rate, bits, dither, level):
    """--level other than 0 dB (the reference's own test scripts use +-4 dB, build_test_*.sh) 20-bit frames and the float dither (the CLI's default for -b 32) at any level: stereo frames stay on the pipelined
    kernels (int8 at M = 8, 16; fp6 at M = 32, 64), whose epilogue then follows the f64 definition (KIND + 4); + 20 dB clips both rails, - 60 dB leaves a few LSB.
    Equal to the oracle and to the one-group kernel (the D2D_DBG_NO_GAINQ flag), samples and peaks, over ragged calls."""
    nbytes = 4096 * 6 * dsd_rate
    chans = [synth("sine", nbytes, seed=71, dsd_rate=dsd_rate, amp=0.5), synth("pink", nbytes, seed=72, dsd_rate=dsd_rate, amp=0.2)]
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=bits, dither=dither, seed=5, level_db=level)
    cuts = [0, 4096, 4096 * 2 + 4096 // 2, nbytes]
    outs = {}
    for off in ("0", "1"):
        e = engine_lib.Engine(n_files=1, kernel=2, debug=engine_lib.DBG_NO_GAINQ if off == "1" else 0, **kw)
        o = oracle_mod.Oracle(**kw)
        got = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            # (planar blocks: a ragged call ends inside a block group only at the end of the stream -- cut on whole blocks)
            a4, b4 = a // 4096 * 4096, b // 4096 * 4096 if b != nbytes else b
            if b4 <= a4:
                continue
            buf = pack_layout([ch[a4:b4] for ch in chans], "P", 4096)
            g, gf = e.translate(buf)
            w, wf = o.translate(buf)
            assert gf == wf and np.array_equal(g, w[:wf * e.frame_bytes]), (off, a4, b4)
            got.append(g)
        outs[off] = np.concatenate(got)
        assert [e.peak(c) for c in range(2)] == [o.peak(c) for c in range(2)]
        name = e.kernel_name()
        pipelined = "d2d_fir_mx_kernel" if 2822400 * dsd_rate // out_rate >= 32 else "d2d_fir_mfma3_kernel"
        if off == "0":
            assert pipelined in name and name.split("<")[1].split(",")[3].strip() in ("4", "5", "6", "7"), name
        else:
            assert pipelined not in name, name
    assert np.array_equal(outs["0"], outs["1"])
    if level == 20.0:
        pcm = decode_pcm(outs["0"], bits, 2)
        top = (1 << (bits - 1)) - 1 if bits != 20 else ((1 << 19) - 1) << 4          # (20-bit samples sit in 24 bits as r << 4)
        assert pcm.max() == top and pcm.min() == (-(1 << (bits - 1)) if bits != 20 else -(1 << 23))


@pytest.mark.parametrize("dsd_rate,out_rate,bits,dither,level", [(1, 88200, 24, "T", -3.0), (1, 352800, 16, "R", 4.0), (2, 88200, 32, "F", 0.0), (1, 176400, 20, "T", 0.0),
                                                                 (4, 88200, 24, "R", -1.0), (2, 176400, 32, "X", -6.0)])
def test_f64_flavour_when_waves_walk_several_tiles(engine_lib, oracle_mod, dsd_rate, out_rate, bits, dither, level):
    """the pipelined kernels' f64 flavour (other levels, 20-bit, the float dither) on streams long enough for every wave to take several
    tiles in its fixed-order loop (the short cases of test_level_in_db_inside_the_pipelined_kernel consist of a trip or two)"""
    M = 2822400 * dsd_rate // out_rate
    nbytes = 6_000_000 * (4 if M >= 32 else 1)               # 2048 waves: several thousand tiles of 512 (M = 8, 16) / 576 / 384 / 192 outputs
    chans = [synth("sine", nbytes, seed=81, dsd_rate=dsd_rate, amp=0.5), synth("pink", nbytes, seed=82, dsd_rate=dsd_rate, amp=0.2)]
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=bits, dither=dither, seed=6, level_db=level)
    e = engine_lib.Engine(n_files=1, kernel=2, **kw)
    o = oracle_mod.Oracle(**kw)
    n4 = nbytes // 4096 * 4096
    buf = pack_layout([ch[:n4] for ch in chans], "P", 4096)
    g, gf = e.translate(buf)
    w, wf = o.translate(buf)
    assert e.kernel_name().split("<")[1].split(",")[3].strip() in ("4", "5", "6", "7"), e.kernel_name()
    assert gf == wf and np.array_equal(g, w[:wf * e.frame_bytes])
    assert [e.peak(c) for c in range(2)] == [o.peak(c) for c in range(2)]


@pytest.mark.parametrize("bits,dither", [(24, "T"), (16, "R"), (32, "X")])
def test_m128_on_the_fp6_kernel_with_several_tiles_per_wave(engine_lib, oracle_mod, bits, dither):
    """DSD256 -> 88.2 kHz (M = 128, one group per column, accumulators from -2^S): 8000 tiles over 2048 waves, the all-integer flavours"""
    nbytes = 24_000_000
    chans = [synth("sine", nbytes, seed=83, dsd_rate=4, amp=0.5), synth("pink", nbytes, seed=84, dsd_rate=4, amp=0.2)]
    kw = dict(dsd_rate=4, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E", bit_depth=bits, dither=dither, seed=7)
    e = engine_lib.Engine(n_files=1, kernel=2, **kw)
    o = oracle_mod.Oracle(**kw)
    n4 = nbytes // 4096 * 4096
    buf = pack_layout([ch[:n4] for ch in chans], "P", 4096)
    g, gf = e.translate(buf)
    w, wf = o.translate(buf)
    assert e.kernel_name().startswith("d2d_fir_mx_kernel<16, 2192, 1,"), e.kernel_name()
    assert gf == wf and np.array_equal(g, w[:wf * e.frame_bytes])
    assert [e.peak(c) for c in range(2)] == [o.peak(c) for c in range(2)]


@pytest.mark.parametrize("dsd_rate,out_rate,filt", [(1, 88200, "E"), (1, 88200, "X"), (2, 88200, "E"), (4, 176400, "E"), (4, 88200, "E"),
                                                    (1, 352800, "E"), (1, 176400, "E"), (2, 352800, "C"), (1, 352800, "D"), (4, 1411200, "E")])
@pytest.mark.parametrize("bits,dither,level", [(24, "T", 0.0), (16, "R", 0.0), (32, "X", 0.0), (24, "T", 4.0), (20, "T", 0.0), (32, "F", -3.0), (24, "X", 0.0)])
def test_mono_streams_as_a_planar_pair_on_the_pipelined_kernel(engine_lib, oracle_mod, dsd_rate, out_rate, filt, bits, dither, level):
    """a MONO stream on the pipelined kernels (round 4, FirArgs::mono2; fp6 at M = 32 / 64 / 128, int8 at M = 8 / 16): the two halves of a call are converted side by side as the two
    "channels" of a planar pair -- the second half's history is the end of the first, its dither indices and frames follow the first's -- and
    leave as one mono stream.  Calls that do not split into two equal halves of whole outputs and 16-byte chunks (odd sizes, very short
    ones) take the ordinary mono kernel; the bytes are the oracle's either way, in any sequence of calls, peaks included."""
    nbytes = 4096 * 40 * dsd_rate
    x = synth("sine", nbytes, seed=31, dsd_rate=dsd_rate, amp=0.5).copy()
    x[5000:7000] = 0xFF; x[90000:91500] = 0x00                       # full-scale stretches: both rails at the integer depths
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=1, fmt="P", endianness="L", block_size=4096, filter=filt, bit_depth=bits,
              dither=dither, seed=77, level_db=level)
    cuts = [0, 4096 * 8, 4096 * 8 + 4096 * 16, 4096 * 24 + 1000, 4096 * 24 + 1000 + 8192 * dsd_rate, nbytes - 4096, nbytes]
    e = engine_lib.Engine(kernel=2, **kw)
    o = oracle_mod.Oracle(**kw)
    names = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        g, gf = e.translate(x[a:b])
        r, rf = o.translate(x[a:b])
        assert gf == rf and np.array_equal(g, r[:rf * o.frame_bytes]), (a, b, e.kernel_name())
        names.append(e.kernel_name())
    pair = "d2d_fir_mx_kernel" if 2822400 * dsd_rate // out_rate >= 32 else "d2d_fir_mfma3_kernel"
    assert names[0].startswith(pair) and names[1].startswith(pair)                                      # whole blocks: the pair route
    assert not names[2].startswith(pair)                                                                # 4096 * 16 + 1000 bytes: the ordinary mono kernel
    assert e.peak(0) == o.peak(0)
    # a batch of files of different lengths in one launch takes the pair route only if every file's call fits
    import torch
    lens = [4096 * 6 * dsd_rate, 4096 * 4 * dsd_rate, 4096 * 10 * dsd_rate]
    eb = engine_lib.Engine(n_files=3, kernel=2, **kw)
    d_in = [torch.from_numpy(x[:n].copy()).cuda() for n in lens]
    d_out = [torch.zeros(eb.next_frames(n, file=i) * eb.frame_bytes + 16, dtype=torch.uint8, device="cuda") for i, n in enumerate(lens)]
    ios = (engine_lib.FileIO * 3)()
    for i, n in enumerate(lens):
        ios[i].dsd = d_in[i].data_ptr(); ios[i].bytes_per_channel = n
        ios[i].pcm = d_out[i].data_ptr(); ios[i].pcm_capacity_bytes = d_out[i].numel()
    eb.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert eb.kernel_name().startswith(pair)
    for i, n in enumerate(lens):
        w, fr = oracle_mod.Oracle(**kw).translate(x[:n])
        assert ios[i].frames_out == fr and np.array_equal(d_out[i][:fr * eb.frame_bytes].cpu().numpy(), w[:fr * eb.frame_bytes]), i


@pytest.mark.parametrize("dsd_rate,out_rate", [(1, 88200), (2, 88200), (4, 176400), (4, 88200)])
@pytest.mark.parametrize("channels,bits,dither,fmt,endian", [(6, 24, "T", "P", "L"), (6, 16, "R", "P", "M"), (6, 32, "X", "P", "L"), (6, 24, "X", "I", "M"), (6, 16, "T", "I", "L"),
                                                             (4, 24, "T", "P", "L"), (4, 16, "R", "I", "M"), (8, 24, "T", "P", "M"), (8, 32, "X", "I", "L"), (8, 16, "X", "P", "L")])
def test_six_channel_frames_leave_whole_from_one_wave(engine_lib, oracle_mod, dsd_rate, out_rate, channels, bits, dither, fmt, endian):
    """a 5.1 stream at 0 dB on the fp6 pipelined kernel (round 4, NPR = 3; quad and eight-channel streams likewise: NPR = 2, 4, at M = 32 and 64):
    a wave converts the three channel pairs of a tile one after
    the other and the tile's whole 18- (12-, 24-) byte frames leave together; a byte-interleaved stream reaches the kernel as the engine's
    planar copy.  Rails, quiet stretches, ragged calls (the careful tiles at a call's edges, a short last call) and the peaks of all six
    channels, against the oracle."""
    rng = np.random.default_rng(11)
    nbytes = 4096 * 36 * dsd_rate
    chans = []
    if channels != 6 and 2822400 * dsd_rate // out_rate == 128:
        pytest.skip("M = 128: whole frames are compiled for 5.1 only")
    for c in range(channels):
        x = synth("sine" if c % 2 == 0 else "pink", nbytes, seed=40 + c, dsd_rate=dsd_rate, amp=0.5 if c % 2 == 0 else 0.098).copy()
        for _ in range(3):
            a = int(rng.integers(0, nbytes - 3000))
            x[a:a + int(rng.integers(200, 3000))] = 0xFF if rng.integers(0, 2) else 0x00
        chans.append(x)
    block = 4096 if fmt == "P" else 1
    kw = dict(dsd_rate=dsd_rate, output_rate=out_rate, channels=channels, fmt=fmt, endianness=endian, block_size=block,
              filter="E", bit_depth=bits, dither=dither, seed=123)
    cuts = [0, 4096 * 5, 4096 * 5 + 4096 * 18, nbytes - 4096 - (300 if fmt == "I" else 0), nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], fmt, block) for a, b in zip(cuts[:-1], cuts[1:])]
    g, r, e, o = run_pair(engine_lib, oracle_mod, bufs, kw, 2)
    M = 2822400 * dsd_rate // out_rate
    assert e.kernel_name() == "d2d_fir_mx_kernel<%d, %d, %d, %d, %d, %d, 5>" % (M // 8, e.info()["ntaps"], {4: 3, 8: 2, 16: 1}[M // 8], {"T": 1, "R": 2, "X": 0}[dither], bits // 8, channels // 2)
    assert g.size == r.size and g.size > 0
    assert np.array_equal(g, r)
    assert [e.peak(c) for c in range(channels)] == [o.peak(c) for c in range(channels)]
    # an odd channel subset (a rank's share of a channel split) and another level keep the two-group kernel
    sub = engine_lib.Engine(kernel=2, **dict(kw, channel_first=1, channel_count=3))
    sub.translate(bufs[0])
    assert not sub.kernel_name().startswith("d2d_fir_mx_kernel")
    lev = engine_lib.Engine(kernel=2, **dict(kw, level_db=-3.0))
    lev.translate(bufs[0])
    assert not lev.kernel_name().startswith("d2d_fir_mx_kernel")
