"""Full-size (BASELINE.json sizes) checks through size-independent properties, where the CPU oracle
would take too long to be the comparator: the two kernels against each other, streaming against
one-shot, exact DC levels, shift invariance.  A 1/16 slice of every file is still checked bit-exactly
against the oracle."""
import ctypes as C

import numpy as np
import pytest

from helpers import pack_layout, synth

pytestmark = pytest.mark.gpu

SECONDS = 60
BLOCKS = int(round(SECONDS * 2822400 / 8 / 4096))     # 4096-byte blocks per channel in a 60 s DSD64 file


def _run_batch(engine_lib, files, kw, kernel, chunks=1):
    import torch
    n = len(files)
    e = engine_lib.Engine(n_files=n, kernel=kernel, **kw)
    bpc = files[0].size // kw["channels"]
    d_in = [torch.from_numpy(f).cuda() for f in files]
    outs = [[] for _ in range(n)]
    blocks = bpc // 4096
    per = (blocks + chunks - 1) // chunks
    for k in range(chunks):
        b0, b1 = k * per, min(blocks, (k + 1) * per)
        if b0 >= b1:
            break
        nb = (b1 - b0) * 4096
        frames = e.next_frames(nb)
        d_out = [torch.empty(frames * e.frame_bytes + 16, dtype=torch.uint8, device="cuda") for _ in range(n)]
        ios = (engine_lib.FileIO * n)()
        for f in range(n):
            # planar blocks are contiguous groups of C*4096 bytes: a block range is a byte range
            ios[f].dsd = d_in[f].data_ptr() + b0 * 4096 * kw["channels"]
            ios[f].bytes_per_channel = nb
            ios[f].pcm = d_out[f].data_ptr()
            ios[f].pcm_capacity_bytes = frames * e.frame_bytes
        e.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for f in range(n):
            outs[f].append(d_out[f][:frames * e.frame_bytes].cpu().numpy())
    return [np.concatenate(o) for o in outs], e


def test_c4_full_length_files_kernels_agree_and_stream(engine_lib, oracle_mod):
    """config 4 shape: DSD64 stereo 60 s files -> 24-bit 88.2 kHz TPDF"""
    kw = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=24, dither="T", seed=206)
    nbytes = BLOCKS * 4096
    files = [pack_layout([synth("sine", nbytes, seed=f, freq=1000.0 + f), synth("pink", nbytes, seed=50 + f, amp=0.098)], "P", 4096)
             for f in range(3)]
    mf, e_m = _run_batch(engine_lib, files, kw, 2)
    lu, e_l = _run_batch(engine_lib, files, kw, 1)
    st, e_s = _run_batch(engine_lib, files, kw, 2, chunks=7)
    frames = BLOCKS * 4096 * 8 // 32
    for f in range(3):
        assert mf[f].size == frames * 6
        assert np.array_equal(mf[f], lu[f])            # matrix-core kernel == LUT kernel, every byte
        assert np.array_equal(mf[f], st[f])            # 7 streaming calls == one shot
        for c in range(2):
            assert e_m.peak(c, f) == e_l.peak(c, f) == e_s.peak(c, f)
        # a slice against the oracle (the first 1/16 of the file)
        sl = (BLOCKS // 16) * 4096 * 2
        r, rf = oracle_mod.Oracle(**kw).translate(files[f][:sl])
        assert np.array_equal(mf[f][:rf * 6], r)


def test_c2_full_length_float_and_dc_levels(engine_lib):
    """config 2 shape: DSD64 stereo -> f32 352.8 kHz, no dither; ones/zeros channels give exactly +-1"""
    kw = dict(dsd_rate=1, output_rate=352800, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=32, dither="X")
    nbytes = (BLOCKS // 4) * 4096
    files = [pack_layout([np.full(nbytes, 0xFF, np.uint8), np.zeros(nbytes, np.uint8)], "P", 4096)]
    peaks = []
    for kernel in (1, 2):
        out, e = _run_batch(engine_lib, files, kw, kernel)
        v = out[0].view(np.float32).reshape(-1, 2)
        assert v.shape[0] == nbytes
        assert (v[64:, 0] == 1.0).all() and (v[64:, 1] == -1.0).all()
        peaks.append((e.peak(0), e.peak(1)))            # includes the step from the idle history: > 1
        assert 1.0 <= peaks[-1][0] < 1.2 and 1.0 <= peaks[-1][1] < 1.2
    assert peaks[0] == peaks[1]


def test_c3_dsd128_shift_invariance(engine_lib):
    """config 3 shape: DSD128 -> 24-bit 88.2 kHz.  Without dither the converter is time invariant: the
    same stream delayed by k*M bits gives the same samples k frames later."""
    kw = dict(dsd_rate=2, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=24, dither="X")
    nbytes = 4096 * 512
    a = [synth("sine", nbytes, seed=1, dsd_rate=2), synth("pink", nbytes, seed=2, amp=0.098, dsd_rate=2)]
    k = 4096 // 8                                       # one whole block = 512 frames at M = 64
    idle = np.full(4096, 0x96, np.uint8)                # the LSB-first idle byte the engine starts from
    b = [np.concatenate([idle, ch[:-4096]]) for ch in a]
    oa, _ = _run_batch(engine_lib, [pack_layout(a, "P", 4096)], kw, 2)
    ob, _ = _run_batch(engine_lib, [pack_layout(b, "P", 4096)], kw, 2)
    fa, fb = oa[0].reshape(-1, 6), ob[0].reshape(-1, 6)
    assert np.array_equal(fb[k:], fa[:-k])


def test_c3_full_length_noise_shaped(engine_lib, oracle_mod):
    """config 3 itself at full length: DSD128 stereo 60 s -> 24-bit 88.2 kHz with the noise-shaped dither 'N' at 0 dB (the FIR kernel's
    scratch flavour + d2d_noise_shape_stereo_kernel's int32 recurrence over 646 segments per channel): one call == five calls (the
    shaper's two carried errors and the segment restarts survive the call boundaries), and the first 1/16 equals the oracle."""
    kw = dict(dsd_rate=2, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
              bit_depth=24, dither="N", seed=3)
    blocks = 2 * BLOCKS                                  # DSD128: twice the bytes per second
    nbytes = blocks * 4096
    files = [pack_layout([synth("sine", nbytes, seed=7, dsd_rate=2), synth("pink", nbytes, seed=8, amp=0.098, dsd_rate=2)], "P", 4096)]
    one, e1 = _run_batch(engine_lib, files, kw, 2)
    five, e5 = _run_batch(engine_lib, files, kw, 2, chunks=5)
    frames = nbytes * 8 // 64
    assert one[0].size == frames * 6
    assert np.array_equal(one[0], five[0])
    for c in range(2):
        assert e1.peak(c) == e5.peak(c)
    sl = (blocks // 16) * 4096 * 2
    r, rf = oracle_mod.Oracle(**kw).translate(files[0][:sl])
    assert np.array_equal(one[0][:rf * 6], r)


def test_c5_full_length_channel_split_and_streaming(engine_lib, oracle_mod):
    """config 5 at its full size: ONE DSD512 8-channel byte-interleaved MSB-first stream of 30 s
    (84.7 MB per channel) -> 24-bit 96 kHz through the 48k cascade.  Properties: four engines that take
    two channels each (the multi-GPU split of SURVEY.md 8e) reproduce the whole conversion exactly;
    feeding the stream in three calls equals one call; the first stretch equals the oracle."""
    import torch
    from dsd2dxd_amd.shard import merge_channel_frames, shard_channels
    chn, seconds = 8, 30
    nbytes = seconds * 2822400 * 8 // 8                    # per channel
    rng = np.random.default_rng(5)
    # a cheap 1-bit stream with low-frequency content: bytes drawn from a slowly varying density
    base = (np.sin(np.arange(nbytes // 4096 + 1) * 0.01)[:, None] * 40 + 128).astype(np.int32)
    data = np.empty((nbytes, chn), np.uint8)
    for c in range(chn):
        dens = np.repeat(base, 4096, axis=0).reshape(-1)[:nbytes]
        data[:, c] = (rng.integers(0, 256, nbytes, dtype=np.int32) + (dens - 128) // 8).clip(0, 255).astype(np.uint8)
    buf = data.reshape(-1)                                 # byte-interleaved: c0 c1 ... c7 per byte time
    kw = dict(dsd_rate=8, output_rate=96000, channels=chn, fmt="I", endianness="M", block_size=1, filter="E",
              bit_depth=24, dither="T", seed=206)
    d_in = torch.from_numpy(buf).cuda()

    def run(chunks, **extra):
        e = engine_lib.Engine(n_files=1, kernel=2, **kw, **extra)
        outs, pos = [], 0
        per = (nbytes // chunks + 15) // 16 * 16
        while pos < nbytes:
            L = min(per, nbytes - pos)
            frames = e.next_frames(L)
            d_out = torch.empty(frames * e.frame_bytes + 16, dtype=torch.uint8, device="cuda")
            ios = (engine_lib.FileIO * 1)()
            ios[0].dsd = d_in.data_ptr() + pos * chn
            ios[0].bytes_per_channel = L
            ios[0].pcm = d_out.data_ptr()
            ios[0].pcm_capacity_bytes = frames * e.frame_bytes
            e.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs.append(d_out[:frames * e.frame_bytes].cpu().numpy())
            pos += L
        return np.concatenate(outs), e

    whole, e = run(1)
    frames = whole.size // (3 * chn)
    assert frames == -(-(nbytes * 8 // 64) * 40 // 147)     # ceil(n_x * L / 147), n_x = bits / 64
    streamed, _ = run(3)
    assert np.array_equal(streamed, whole)
    parts = []
    for r in range(4):
        first, count = shard_channels(chn, 4, r)
        out, es = run(1, channel_first=first, channel_count=count)
        parts.append((first, count, out))
        for c in range(count):
            assert es.peak(c) == e.peak(first + c)
    assert np.array_equal(merge_channel_frames(parts, 3), whole)
    # the head of the stream against the oracle (1/64 of it: the oracle's 48k path is slow)
    head = nbytes // 64 // 16 * 16
    r, rf = oracle_mod.Oracle(**kw).translate(buf[:head * chn])
    assert np.array_equal(whole[:(rf - 200) * 3 * chn], r[:(rf - 200) * 3 * chn])
