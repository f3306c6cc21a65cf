"""GPU tests of the rest of the C ABI: ragged batches, table blob export/import (the multi-GPU
broadcast path), the whole-stream driver with cancel and progress, reset, error codes at run time."""
import ctypes as C

import numpy as np
import pytest

from helpers import pack_layout, random_bytes, synth

pytestmark = pytest.mark.gpu

KW = dict(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, filter="E",
          bit_depth=24, dither="T", seed=77)


@pytest.mark.parametrize("kernel", [1, 2])
def test_ragged_batch(engine_lib, oracle_mod, kernel):
    """files of different lengths (one of them empty) advance together; each matches its own oracle"""
    import torch
    lens = [4096 * 5, 0, 4096 * 2 + 300, 4096 * 9, 40]
    files = [pack_layout([random_bytes(n, 10 + i), random_bytes(n, 20 + i)], "P", 4096) for i, n in enumerate(lens)]
    e = engine_lib.Engine(n_files=len(lens), kernel=kernel, **KW)
    oracles = [oracle_mod.Oracle(**KW) for _ in lens]
    for rnd in range(2):                               # two calls: state carries per file
        d_in = [torch.from_numpy(f).cuda() if f.size else torch.zeros(16, dtype=torch.uint8, device="cuda") for f in files]
        d_out = [torch.zeros(e.next_frames(n, file=i) * e.frame_bytes + 16, dtype=torch.uint8, device="cuda") for i, n in enumerate(lens)]
        ios = (engine_lib.FileIO * len(lens))()
        for i, n in enumerate(lens):
            ios[i].dsd = d_in[i].data_ptr(); ios[i].bytes_per_channel = n
            ios[i].pcm = d_out[i].data_ptr(); ios[i].pcm_capacity_bytes = d_out[i].numel()
        e.translate_batch_device(ios, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for i in range(len(lens)):
            r, rf = oracles[i].translate(files[i])
            assert ios[i].frames_out == rf
            assert np.array_equal(d_out[i][:rf * e.frame_bytes].cpu().numpy(), r[:rf * e.frame_bytes])


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("out_rate", [88200, 96000])
def test_table_blob_roundtrip(engine_lib, oracle_mod, kernel, out_rate):
    """rank 0 exports its tables, another engine adopts them and still converts correctly;
    a blob from a different configuration is refused"""
    import torch
    kw = dict(KW, output_rate=out_rate)
    a = engine_lib.Engine(kernel=kernel, **kw)
    b = engine_lib.Engine(kernel=kernel, **kw)
    nb = a.tables_bytes()
    assert nb == b.tables_bytes() and nb > 1024
    blob = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    a.tables_export_device(blob.data_ptr(), nb)
    torch.cuda.synchronize()
    b.tables_import_device(blob.data_ptr(), nb)
    buf = pack_layout([synth("sine", 4096 * 4, seed=1), synth("pink", 4096 * 4, seed=2, amp=0.098)], "P", 4096)
    g, gf = b.translate(buf)
    r, rf = oracle_mod.Oracle(**kw).translate(buf)
    assert gf == rf and np.array_equal(g, r[:rf * 6])
    other = engine_lib.Engine(kernel=kernel, **dict(kw, endianness="M"))
    with pytest.raises(engine_lib.D2DError) as ei:
        other.tables_import_device(blob.data_ptr(), nb)
    assert ei.value.code == -1
    # the same filter under another conversion can mean another table variant of the SAME size (one channel: the two-group kernel, plane 0
    # unmasked; stereo at 0 dB: a pipelined kernel, every plane masked or fp6 digits): the header names the variant and the import is refused
    if kernel == 2 and out_rate == 88200:
        mono = engine_lib.Engine(kernel=kernel, **dict(kw, channels=1))
        with pytest.raises(engine_lib.D2DError):
            mono.tables_import_device(blob.data_ptr(), nb)
        g3, f3 = mono.translate(buf[:buf.size // 2])      # ... and the refused engine still converts with its own tables
        r3, rf3 = oracle_mod.Oracle(**dict(kw, channels=1)).translate(buf[:buf.size // 2])
        assert f3 == rf3 and np.array_equal(g3, r3[:rf3 * 3])
        # another level in dB, 20-bit frames or the float dither run the same pipelined kernel (its f64 flavour) on the same table: the blob is adopted
        for extra in (dict(level_db=-3.0), dict(bit_depth=20), dict(bit_depth=32, dither="F")):
            lvl = engine_lib.Engine(kernel=kernel, **dict(kw, **extra))
            lvl.tables_import_device(blob.data_ptr(), nb)
            g4, f4 = lvl.translate(buf)
            r4, rf4 = oracle_mod.Oracle(**dict(kw, **extra)).translate(buf)
            assert f4 == rf4 and np.array_equal(g4, r4[:rf4 * lvl.frame_bytes])
    # a corrupted table must change the output (i.e. the imported bytes are really what runs)
    blob[256:(3 * nb) // 4] = 0      # (the MFMA table holds four byte-shift variants; wipe most of them)
    b2 = engine_lib.Engine(kernel=kernel, **kw)
    b2.tables_import_device(blob.data_ptr(), nb)
    g2, _ = b2.translate(buf)
    assert not np.array_equal(g2, g)


def test_convert_stream_progress_and_cancel(engine_lib, oracle_mod):
    nbytes = 4096 * 40
    chans = [synth("sine", nbytes, seed=3), synth("pink", nbytes, seed=4, amp=0.098)]
    whole = pack_layout(chans, "P", 4096)
    pos = [0]
    chunk_blocks = 8

    def read(cap):
        assert cap % 4096 == 0
        a = pos[0]
        b = min(nbytes, a + min(cap, chunk_blocks * 4096))
        pos[0] = b
        return pack_layout([c[a:b] for c in chans], "P", 4096).tobytes() if b > a else b""

    out, prog = [], []
    e = engine_lib.Engine(**KW)
    e.convert_stream(read, out.append, total_bytes_per_channel=nbytes, chunk_bytes_per_channel=chunk_blocks * 4096,
                     progress=prog.append)
    r, rf = oracle_mod.Oracle(**KW).translate(whole)
    assert np.array_equal(np.frombuffer(b"".join(out), dtype=np.uint8), r[:rf * 6])
    assert prog[-1] == 100.0 and all(p < 100.0 for p in prog[:-1]) and prog == sorted(prog)   # src/main.rs:417-418
    # cancel: the flag is polled between chunks (src/main.rs:38,429)
    pos[0] = 0
    flag = C.c_int(0)
    seen = []

    def write_then_cancel(b):
        seen.append(len(b))
        flag.value = 1

    e2 = engine_lib.Engine(**KW)
    with pytest.raises(engine_lib.D2DError) as ei:
        e2.convert_stream(read, write_then_cancel, total_bytes_per_channel=nbytes,
                          chunk_bytes_per_channel=chunk_blocks * 4096, cancel=flag)
    assert ei.value.code == -30 and len(seen) == 1


def test_runtime_errors(engine_lib):
    e = engine_lib.Engine(**KW)
    L = engine_lib.lib()
    buf = np.zeros(4096 * 2, dtype=np.uint8)
    out = np.zeros(16, dtype=np.uint8)
    fr = C.c_size_t()
    rc = L.d2d_translate(e._h, buf.ctypes.data, 4096, out.ctypes.data, out.size, C.byref(fr))
    assert rc == -20 and b"too small" in L.d2d_last_error(e._h)          # D2D_ERR_CAPACITY
    eb = engine_lib.Engine(n_files=2, **KW)
    rc = L.d2d_translate(eb._h, buf.ctypes.data, 2048, out.ctypes.data, out.size, C.byref(fr))
    assert rc == -40                                                     # single-file entry point on a batch engine
    e.reset()
    g1, _ = e.translate(buf)
    e.reset()
    g2, _ = e.translate(buf)
    assert np.array_equal(g1, g2)


def test_engines_on_concurrent_threads(engine_lib, oracle_mod):
    """one engine per worker thread, created, used and dropped there (src/main.rs:361-394,429): different
    configurations at the same time must not disturb each other (launch preparation is shared state)"""
    import threading
    configs = [dict(KW, output_rate=88200, kernel=2), dict(KW, output_rate=96000, kernel=2), dict(KW, output_rate=176400, kernel=1),
               dict(KW, output_rate=88200, channels=1, kernel=2), dict(KW, output_rate=192000, kernel=2), dict(KW, output_rate=352800, bit_depth=32, dither="F", kernel=2)]
    n = 4096 * 6
    data, want = [], []
    for i, kw in enumerate(configs):
        chans = [synth("sine", n, seed=40 + i), synth("pink", n, seed=50 + i, amp=0.098)][:kw["channels"]]
        buf = pack_layout(chans, "P", 4096)
        data.append(buf)
        okw = {k: v for k, v in kw.items() if k != "kernel"}
        want.append(oracle_mod.Oracle(**okw).translate(buf))
    got, errs = [None] * len(configs), []

    def work(i):
        try:
            for _ in range(3):                                   # fresh engine each time: creation races too
                e = engine_lib.Engine(n_files=1, **configs[i])
                pieces, total = [], 0
                for part in np.array_split(data[i].reshape(-1, 4096 * configs[i]["channels"]), 3):   # whole block groups per call
                    out, fr = e.translate(part.reshape(-1))
                    pieces.append(out.copy()); total += fr
                got[i] = (np.concatenate(pieces), total)
                del e
        except Exception as ex:                                  # noqa: BLE001
            errs.append((i, repr(ex)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(configs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(len(configs)):
        r, rf = want[i]
        fb = {16: 2, 32: 4}.get(configs[i]["bit_depth"], 3) * configs[i]["channels"]
        assert got[i][1] == rf and np.array_equal(got[i][0], r[:rf * fb]), i


@pytest.mark.parametrize("out_rate,fmt,pinned,staged", [(88200, "P", True, True), (88200, "P", True, False), (96000, "I", True, False), (96000, "I", False, False),
                                                        (176400, "P", False, False), (88200, "N24", True, False)])
def test_host_resident_batch_pipeline(engine_lib, oracle_mod, out_rate, fmt, pinned, staged):
    """d2d_translate_batch_host: ragged files, many slices (upload / convert / download overlapped), state
    carried from slice to slice; the bytes equal the oracle's one-shot conversion of each file.  Pinned buffers are
    read and written by the kernels themselves (no staging, one call) unless the D2D_DBG_HOST_STAGED flag keeps the pipeline."""
    import torch
    debug = engine_lib.DBG_HOST_STAGED if staged else 0
    extra = {}
    if fmt == "N24":
        fmt, extra = "P", dict(dither="N")
    kw = dict(KW, output_rate=out_rate, fmt=fmt, endianness="L" if fmt == "P" else "M", **extra)
    lens = [4096 * 7 + 123, 4096 * 3, 0, 4096 * 12 + 4000]
    files = [pack_layout([random_bytes(n, 60 + i), random_bytes(n, 70 + i)], fmt, 4096 if fmt == "P" else 1) for i, n in enumerate(lens)]
    e = engine_lib.Engine(n_files=len(lens), kernel=2, debug=debug, **kw)
    fb = e.frame_bytes
    want = [oracle_mod.Oracle(**kw).translate(f) for f in files]
    ins, outs = [], []
    for f, (r, rf) in zip(files, want):
        ti = torch.from_numpy(f.copy()) if f.size else torch.zeros(16, dtype=torch.uint8)
        to = torch.zeros(rf * fb + 64, dtype=torch.uint8)
        if pinned:
            ti, to = ti.pin_memory(), to.pin_memory()
        ins.append(ti); outs.append(to)
    ios = (engine_lib.FileIO * len(lens))()
    for i, n in enumerate(lens):
        ios[i].dsd = ins[i].data_ptr(); ios[i].bytes_per_channel = n
        ios[i].pcm = outs[i].data_ptr(); ios[i].pcm_capacity_bytes = outs[i].numel()
    e.translate_batch_host(ios, 8192)                            # 2-block slices: up to 7 of them
    for i, (r, rf) in enumerate(want):
        assert ios[i].frames_out == rf, (i, ios[i].frames_out, rf)
        assert np.array_equal(outs[i][:rf * fb].numpy(), r[:rf * fb]), i
    # too small an output buffer is reported, not overrun
    e2 = engine_lib.Engine(n_files=len(lens), kernel=2, debug=debug, **kw)
    ios[3].pcm_capacity_bytes = 64
    with pytest.raises(Exception, match="pcm buffer too small"):
        e2.translate_batch_host(ios, 8192)


@pytest.mark.parametrize("out_rate,dither", [(88200, "T"), (96000, "T"), (88200, "N")])
def test_per_block_calls_on_pinned_buffers_need_no_staging(engine_lib, oracle_mod, out_rate, dither):
    """d2d_translate with buffers the GPU can address (hipHostMalloc / hipHostRegister: torch's pinned tensors): the kernels read
    and write them in place, call after call with carried state; same bytes as the staged route and the oracle"""
    import torch
    kw = dict(KW, output_rate=out_rate, dither=dither)
    blocks = [4096 * 2, 4096, 4096 * 5, 4096 * 3]
    data = [pack_layout([random_bytes(n, 300 + i), random_bytes(n, 310 + i)], "P", 4096) for i, n in enumerate(blocks)]
    o = oracle_mod.Oracle(**kw)
    want = [o.translate(b) for b in data]
    got = {}
    for staged in ("0", "1"):
        e = engine_lib.Engine(n_files=1, kernel=2, debug=engine_lib.DBG_HOST_STAGED if staged == "1" else 0, **kw)
        fb = e.frame_bytes
        res = []
        for b, n in zip(data, blocks):
            ti = torch.from_numpy(b.copy()).pin_memory()
            to = torch.zeros(e.next_frames(n) * fb + 64, dtype=torch.uint8).pin_memory()
            fr = e.translate_into(ti.data_ptr(), n, to.data_ptr(), to.numel())
            res.append((to[:fr * fb].numpy().copy(), fr))
        got[staged] = res
    for (r, rf), (a, af), (b, bf) in zip(want, got["0"], got["1"]):
        assert af == bf == rf and np.array_equal(a, r[:rf * 6]) and np.array_equal(b, a)


@pytest.mark.parametrize("staged", [True, False])
@pytest.mark.parametrize("block", [100, 4100, 7])
def test_host_resident_batch_with_odd_block_sizes(engine_lib, oracle_mod, block, staged):
    """slices of d2d_translate_batch_host must be whole planar blocks for ANY block size (-s takes any
    value); rounding the slice to 16 bytes afterwards cut blocks in two and mixed the channels."""
    import torch
    kw = dict(KW, output_rate=88200, fmt="P", endianness="L", block_size=block)
    lens = [block * 37 + 11, block * 5, block * 64]
    files = [pack_layout([random_bytes(n, 160 + i), random_bytes(n, 170 + i)], "P", block) for i, n in enumerate(lens)]
    e = engine_lib.Engine(n_files=len(lens), kernel=2, debug=engine_lib.DBG_HOST_STAGED if staged else 0, **kw)
    fb = e.frame_bytes
    want = [oracle_mod.Oracle(**kw).translate(f) for f in files]
    ins = [torch.from_numpy(f.copy()).pin_memory() for f in files]
    outs = [torch.zeros(rf * fb + 64, dtype=torch.uint8).pin_memory() for _, rf in want]
    ios = (engine_lib.FileIO * len(lens))()
    for i, n in enumerate(lens):
        ios[i].dsd = ins[i].data_ptr(); ios[i].bytes_per_channel = n
        ios[i].pcm = outs[i].data_ptr(); ios[i].pcm_capacity_bytes = outs[i].numel()
    e.translate_batch_host(ios, max(block * 3 + 1, 1000))       # a slice that is NOT a block multiple by itself
    for i, (r, rf) in enumerate(want):
        assert ios[i].frames_out == rf, (i, ios[i].frames_out, rf)
        assert np.array_equal(outs[i][:rf * fb].numpy(), r[:rf * fb]), (block, i)


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("out_rate,fmt,bits", [(88200, "I", 24), (96000, "I", 24), (176400, "P", 16), (352800, "P", 32)])
def test_channel_subsets_reproduce_the_full_conversion(engine_lib, oracle_mod, kernel, out_rate, fmt, bits):
    """SURVEY.md 8e / BASELINE config 5: the ranks of a multi-GPU job take a few channels each of ONE
    multichannel stream.  Every subset engine (fed the whole input) must emit exactly the columns of the
    full conversion -- same filter state, same per-channel dither, same peaks -- call after call."""
    chn = 6
    kw = dict(KW, output_rate=out_rate, channels=chn, fmt=fmt, endianness="M" if fmt == "I" else "L", bit_depth=bits,
              dither="F" if bits == 32 else "T")
    block = 4096 if fmt == "P" else 1
    nbytes = 4096 * 3 + 300
    chans = [random_bytes(nbytes, 500 + c) for c in range(chn)]
    cuts = [0, 4096 * 2, nbytes]
    bufs = [pack_layout([ch[a:b] for ch in chans], fmt, block) for a, b in zip(cuts[:-1], cuts[1:])]
    o = oracle_mod.Oracle(**kw)
    sb = {16: 2, 32: 4}.get(bits, 3)
    full = []
    for b in bufs:
        r, rf = o.translate(b)
        full.append(r[:rf * sb * chn].reshape(rf, chn, sb))
    full = np.concatenate(full)
    for first, count in ((0, 1), (1, 2), (3, 3), (5, 0), (2, 4)):
        e = engine_lib.Engine(n_files=1, kernel=kernel, channel_first=first, channel_count=count, **kw)
        n = count or chn - first
        assert e.out_channels == n and e.frame_bytes == sb * n
        got = []
        for b in bufs:
            out, fr = e.translate(b)
            got.append(out.reshape(fr, n, sb))
        got = np.concatenate(got)
        assert np.array_equal(got, full[:, first:first + n, :]), (first, count)
        for c in range(n):
            assert e.peak(c) == o.peak(first + c)
    with pytest.raises(engine_lib.D2DError, match="Invalid channel subset"):
        engine_lib.Engine(n_files=1, kernel=kernel, channel_first=4, channel_count=3, **kw)


def test_profile_read_all_brackets_every_kernel_of_a_call(engine_lib):
    """d2d_profile_read_all: the FIR kernel's device time and the time of every kernel of the calls (the
    bench priced multi-kernel workloads with the FIR launch alone).  For the 48k cascade (DSD256 / DSD512 input) the step holds stage A,
    stage B and the carries; for a plain 44.1k conversion, or the composed 48k filter of DSD64 / DSD128, the FIR kernel plus the history carry."""
    import torch
    for dsd_rate, out_rate, more in ((1, 88200, 1.0), (1, 96000, 1.0), (8, 96000, 1.3)):
        kw = dict(KW, dsd_rate=dsd_rate, output_rate=out_rate, fmt="P", endianness="L")
        e = engine_lib.Engine(n_files=2, kernel=2, **kw)
        n = 4096 * 64
        buf = torch.from_numpy(np.concatenate([pack_layout([random_bytes(n, 1), random_bytes(n, 2)], "P", 4096)] * 1)).cuda()
        frames = e.next_frames(n)
        out = torch.zeros((2, (frames * e.frame_bytes + 31) // 16 * 16), dtype=torch.uint8, device="cuda")
        ios = (engine_lib.FileIO * 2)()
        for i in range(2):
            ios[i].dsd = buf.data_ptr(); ios[i].bytes_per_channel = n
            ios[i].pcm = out[i].data_ptr(); ios[i].pcm_capacity_bytes = frames * e.frame_bytes
        e.profile_enable(True)
        e.translate_batch_device(ios)
        torch.cuda.synchronize()
        fir, step, launches = e.profile_read_all()
        assert launches == 1 and fir > 0 and step >= fir * more, (out_rate, fir, step)
        assert e.profile_read_all() == (0.0, 0.0, 0)              # read clears
