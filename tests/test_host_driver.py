"""The host side above the C ABI: container readers, PCM sinks, the Rdsd2Pcm mirror and
its small CLI (dsd2dxd_amd/dsd2dxd_amd_cli).  CPU tests cover what needs no GPU (the readers, found
through `probe`); GPU tests convert whole files and compare the written audio with the oracle."""
import hashlib
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import decode_pcm, pack_layout, random_bytes, synth, write_dff, write_dsf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "dsd2dxd_amd", "dsd2dxd_amd_cli")
REF = "/root/reference"


@pytest.fixture(scope="module")
def cli(engine_lib):
    if not os.path.exists(CLI):
        engine_lib.build_library()
    return CLI


def probe(cli, *paths):
    return json.loads(subprocess.check_output([cli, "probe", *paths]))


def test_probe_synthetic_containers(cli, tmp_path):
    chans = [random_bytes(4096 * 3 + 100, 1), random_bytes(4096 * 3 + 100, 2)]
    p1, p2 = str(tmp_path / "a.dsf"), str(tmp_path / "b.dff")
    write_dsf(p1, chans, dsd_rate=2, lsb_first=True, sample_count=(4096 * 3 + 100) * 8 - 3, id3=b"ID3" + bytes(20))
    write_dff(p2, chans, dsd_rate=4, tail=b"ID3 " + struct.pack(">Q", 5000) + b"xx")     # truncated tag
    a, b = probe(cli, p1, p2)
    assert a["error"] == "" and (a["format"], a["channels"], a["dsd_rate"], a["planar"], a["msb_first"], a["block_size"]) == ("dsf", 2, 2, True, False, 4096)
    assert a["data_offset"] == 92 and a["data_bytes"] == 4 * 4096 * 2 and a["bytes_per_channel"] == 4096 * 3 + 100
    assert a["metadata_offset"] == 92 + a["data_bytes"] and not a["metadata_truncated"]
    assert b["error"] == "" and (b["format"], b["channels"], b["dsd_rate"], b["planar"], b["msb_first"], b["block_size"]) == ("dff", 2, 4, False, True, 1)
    assert b["bytes_per_channel"] == 4096 * 3 + 100 and b["metadata_truncated"] and "audio is intact" in b["warning"]
    bad = str(tmp_path / "c.dsf")
    open(bad, "wb").write(b"RIFFxxxx" + bytes(200))
    assert "bad 'DSD ' chunk" in probe(cli, bad)[0]["error"]


def _syncsafe(n):
    return bytes([(n >> 21) & 0x7F, (n >> 14) & 0x7F, (n >> 7) & 0x7F, n & 0x7F])


def make_id3(ver, frames, padding=0):
    """frames: [(id, payload bytes)]; v2.3 sizes are plain big-endian, v2.4 sizes are syncsafe"""
    body = b""
    for fid, pay in frames:
        body += fid + (struct.pack(">I", len(pay)) if ver == 3 else _syncsafe(len(pay))) + b"\x00\x00" + pay
    body += bytes(padding)
    return b"ID3" + bytes([ver, 0, 0]) + _syncsafe(len(body)) + body


def tags(cli, *args):
    return json.loads(subprocess.check_output([cli, "tags", *args]))


PNG = b"\x89PNG\r\n\x1a\n" + bytes(range(64))


def test_tags_are_read_decoded_and_album_marked(cli, tmp_path):
    """README.md:7 (tags are copied where possible) and README.md:170-173 (-a marks the album)"""
    chans = [random_bytes(4096, 1), random_bytes(4096, 2)]
    t23 = make_id3(3, [(b"TIT2", b"\x00Caf\xe9 tone"), (b"TALB", b"\x01\xff\xfe" + "Hits ♫".encode("utf-16-le") + b"\x00\x00"),
                       (b"TRCK", b"\x003/12"), (b"TYER", b"\x002024"), (b"COMM", b"\x00engshort\x00made by a test"),
                       (b"APIC", b"\x00image/png\x00\x03cover\x00" + PNG)], padding=300)
    t24 = make_id3(4, [(b"TIT2", b"\x03" + "Sinus ♪".encode()), (b"TPE1", b"\x03someone\x00"), (b"TALB", b"\x03Album\x00"),
                       (b"TDRC", b"\x032025-01-02")], padding=130)
    p1, p2, p3 = str(tmp_path / "a.dsf"), str(tmp_path / "b.dff"), str(tmp_path / "c.dsf")
    write_dsf(p1, chans, id3=t23)
    write_dff(p2, chans, tail=b"ID3 " + struct.pack(">Q", len(t24)) + t24 + (b"\x00" if len(t24) & 1 else b""))
    write_dsf(p3, chans)
    a, b, c = tags(cli, p1, p2, p3)
    assert a["error"] == "" and a["warning"] == "" and a["pictures"] == 1 and a["tag_bytes"] == len(t23) - 300   # padding dropped
    assert a["fields"] == {"TITLE": "Café tone", "ALBUM": "Hits ♫", "TRACKNUMBER": "3", "TRACKTOTAL": "12", "DATE": "2024",
                           "COMMENT": "made by a test"}
    assert b["fields"] == {"TITLE": "Sinus ♪", "ARTIST": "someone", "ALBUM": "Album", "DATE": "2025-01-02"} and b["tag_bytes"] == len(t24) - 130
    assert c["tag_bytes"] == 0 and c["fields"] == {} and c["warning"] == ""
    a, b, c = tags(cli, "-a", "88200", p1, p2, p3)
    assert a["album_edited"] and a["fields"]["ALBUM"] == "Hits ♫ [88.2K]" and a["fields"]["TITLE"] == "Café tone" and a["pictures"] == 1
    assert b["album_edited"] and b["fields"]["ALBUM"] == "Album [88.2K]" and b["fields"]["DATE"] == "2025-01-02"
    assert not c["album_edited"]
    assert tags(cli, "-a", "96000", p2)[0]["fields"]["ALBUM"] == "Album [96K]"
    # a tag without an album frame, and one this editor must not rewrite (unsynchronised): copied untouched
    t_noalb = make_id3(3, [(b"TIT2", b"\x00x")])
    t_unsync = bytearray(make_id3(3, [(b"TALB", b"\x00y")])); t_unsync[5] = 0x80
    write_dsf(p1, chans, id3=t_noalb)
    write_dsf(p3, chans, id3=bytes(t_unsync))
    a, c = tags(cli, "-a", "88200", p1, p3)
    assert not a["album_edited"] and a["tag_bytes"] == len(t_noalb) and not c["album_edited"] and c["tag_bytes"] == len(t_unsync)
    # a tag that claims more than the file holds is dropped with a warning, the audio stays convertible
    write_dsf(p1, chans, id3=t23[:60])
    a = tags(cli, p1)[0]
    assert a["error"] == "" and a["tag_bytes"] == 0 and "damaged" in a["warning"]


@pytest.mark.skipif(not os.path.isdir(REF + "/id3_test"), reason="reference fixtures not present")
def test_tags_of_reference_fixtures(cli):
    good, bad_dff, bad_dsf = tags(cli, "-a", "176400", REF + "/id3_test/dff/1kHz_stereo_i.dff",
                                  REF + "/id3_test/dff/1kHz_stereo_i_brokenid3.dff", REF + "/id3_test/1kHz_mono_brokenid3.dsf")
    assert good["fields"] == {"TITLE": "1kHz Test Tone DSD64", "ARTIST": "clone206",
                              "ALBUM": "clone206's Greatest Test Tone Hits [176.4K]"} and good["album_edited"]
    for t in (bad_dff, bad_dsf):
        assert t["error"] == "" and t["tag_bytes"] == 0 and "damaged" in t["warning"]


@pytest.mark.skipif(not os.path.isdir(REF + "/test"), reason="reference fixtures not present")
def test_probe_reference_fixtures(cli):
    """the facts SURVEY.md 4.3 measured from the fixture files, including the damaged-ID3 ones"""
    r = {os.path.basename(d["path"]): d for d in probe(
        cli, REF + "/test/1kHz_mono_p.dsf", REF + "/test/1kHz_stereo_128.dsf", REF + "/test/pinknoise_mono_128.dsf",
        REF + "/id3_test/1kHz_mono_brokenid3.dsf", REF + "/id3_test/dff/1kHz_stereo_i.dff", REF + "/id3_test/dff/1kHz_stereo_i_brokenid3.dff")}
    m = r["1kHz_mono_p.dsf"]
    assert (m["channels"], m["dsd_rate"], m["msb_first"], m["block_size"], m["sample_count"], m["data_offset"], m["data_bytes"]) == (1, 1, False, 4096, 14112000, 92, 1765376)
    s = r["1kHz_stereo_128.dsf"]
    assert (s["channels"], s["dsd_rate"], s["sample_count"], s["data_bytes"]) == (2, 2, 11289600, 2 * 345 * 4096)
    assert r["1kHz_mono_brokenid3.dsf"]["error"] == "" and r["1kHz_mono_brokenid3.dsf"]["metadata_truncated"]
    d = r["1kHz_stereo_i.dff"]
    assert (d["channels"], d["dsd_rate"], d["planar"], d["msb_first"], d["data_offset"], d["bytes_per_channel"]) == (2, 1, False, True, 130, 1058400)
    assert r["1kHz_stereo_i_brokenid3.dff"]["error"] == "" and r["1kHz_stereo_i_brokenid3.dff"]["metadata_truncated"]
    # payload identities the README states (README.md:205): the .dsd files are the containers' payloads
    raw = np.fromfile(REF + "/test/1kHz_mono_p.dsd", dtype=np.uint8)
    dsf = np.fromfile(REF + "/test/1kHz_mono_p.dsf", dtype=np.uint8)
    assert np.array_equal(raw, dsf[92:92 + 1765376])
    dff = np.fromfile(REF + "/id3_test/dff/1kHz_stereo_i.dff", dtype=np.uint8)
    assert np.array_equal(np.fromfile(REF + "/test/1kHz_stereo_i.dsd", dtype=np.uint8), dff[130:130 + 2116800])


# ---- GPU: whole-file conversions through the CLI --------------------------------------------------

def _wav_payload(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos = 12
    fmt = None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
        if cid == b"data":
            return fmt, np.frombuffer(b[pos + 8:pos + 8 + sz], dtype=np.uint8)
        pos += 8 + sz + (sz & 1)
    raise AssertionError("no data chunk")


def _aiff_payload(path):
    b = open(path, "rb").read()
    assert b[:4] == b"FORM" and b[8:12] in (b"AIFF", b"AIFC")
    pos = 12
    comm = None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack(">I", b[pos + 4:pos + 8])[0]
        if cid == b"COMM":
            comm = struct.unpack(">hIh", b[pos + 8:pos + 16])
        if cid == b"SSND":
            return comm, np.frombuffer(b[pos + 16:pos + 8 + sz], dtype=np.uint8)
        pos += 8 + sz + (sz & 1)
    raise AssertionError("no SSND chunk")


def _flac_decode(path):
    """decoder for the subset the sink writes: fixed order 0/2 or verbatim, one Rice partition"""
    b = open(path, "rb").read()
    assert b[:4] == b"fLaC"
    si = b[8:42]
    v = int.from_bytes(si[10:18], "big")
    rate, ch, bps, total = v >> 44, ((v >> 41) & 7) + 1, ((v >> 36) & 31) + 1, v & ((1 << 36) - 1)
    mpos, last = 4, False
    _flac_decode.blocks = []
    while not last:                                   # metadata blocks: STREAMINFO first, tags may follow
        last, btype, blen = bool(b[mpos] & 0x80), b[mpos] & 0x7F, int.from_bytes(b[mpos + 1:mpos + 4], "big")
        _flac_decode.blocks.append((btype, b[mpos + 4:mpos + 4 + blen]))
        mpos += 4 + blen
    assert _flac_decode.blocks[0][0] == 0
    bits = "".join(f"{x:08b}" for x in b[mpos:])
    pos = 0

    def rd(n):
        nonlocal pos
        x = int(bits[pos:pos + n], 2) if n else 0
        pos += n
        return x

    def sgn(x, n):
        return x - (1 << n) if x >> (n - 1) else x
    def crc(data, poly, width):
        c, top, mask = 0, 1 << (width - 1), (1 << width) - 1
        for byte in data:
            c ^= byte << (width - 8)
            for _ in range(8):
                c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
        return c
    raw = b[mpos:]
    out = []
    frame_no = 0
    while len(out) < total:
        f0 = pos // 8                                  # frames start on byte boundaries
        assert pos % 8 == 0 and rd(16) == 0xFFF8
        bsc = rd(4); rd(4); assert rd(4) == ch - 1; rd(3); rd(1)
        # the frame number, "UTF-8" coded: consecutive from 0 (fixed block size stream)
        first = rd(8)
        if first < 0x80:
            num = first
        else:
            n = 1 if first < 0xE0 else 2 if first < 0xF0 else 3 if first < 0xF8 else 4 if first < 0xFC else 5
            num = first & (0x3F >> n)
            for _ in range(n):
                c = rd(8)
                assert c >> 6 == 2
                num = (num << 6) | (c & 0x3F)
        assert num == frame_no, (num, frame_no)
        frame_no += 1
        n = 4096 if bsc == 0xC else rd(16) + 1
        # CRC-8 (x^8 + x^2 + x + 1) over the frame header up to here
        assert pos % 8 == 0
        rd(8)
        hdr_end = pos // 8 - 1
        assert raw[hdr_end] == crc(raw[f0:hdr_end], 0x07, 8), "frame header CRC-8"
        chans = []
        for _ in range(ch):
            assert rd(1) == 0
            t = rd(6); assert rd(1) == 0
            if t == 1:
                s = [sgn(rd(bps), bps) for _ in range(n)]
            else:
                order = t - 8
                s = [sgn(rd(bps), bps) for _ in range(order)]
                assert rd(2) == 1 and rd(4) == 0
                k = rd(5)
                for i in range(order, n):
                    q = 0
                    while bits[pos] == "0":
                        q += 1; pos += 1
                    pos += 1
                    u = (q << k) | rd(k)
                    r = (u >> 1) if not (u & 1) else -((u + 1) >> 1)
                    s.append(r + (2 * s[i - 1] - s[i - 2] if order == 2 else 0))
            chans.append(s)
        pos = (pos + 7) & ~7
        # CRC-16 (x^16 + x^15 + x^2 + 1) over the whole frame before it
        assert rd(16) == crc(raw[f0:pos // 8 - 2], 0x8005, 16), "frame CRC-16"
        out.extend(zip(*chans))
    return rate, ch, bps, np.array(out[:total], dtype=np.int64)


@pytest.mark.gpu
def test_dsf_to_wav_respects_sample_count(cli, oracle_mod, tmp_path):
    n = 4096 * 5 + 1000
    chans = [synth("sine", n, seed=1), synth("pink", n, seed=2, amp=0.098)]
    src = str(tmp_path / "tone.dsf")
    write_dsf(src, chans, dsd_rate=1, lsb_first=True)            # last block group is padded inside the file
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    subprocess.check_call([cli, "-o", "w", "-r", "88200", "-b", "24", "-d", "T", "-a", "-p", str(out_dir), "-q", src])
    fmt, pay = _wav_payload(str(out_dir / "tone_88_2K.wav"))
    assert fmt[:4] == (1, 2, 88200, 88200 * 6) and fmt[5] == 24
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096,
                          bit_depth=24, dither="T", seed=0)
    r, rf = o.translate(pack_layout(chans, "P", 4096))            # the padding must NOT be converted
    assert rf == n * 8 // 32 and np.array_equal(pay, r[:rf * 6])


@pytest.mark.gpu
def test_cli_with_32_bit_taps(cli, oracle_mod, tmp_path):
    """--tap-bits 32 (Rdsd2Pcm::set_tap_bits / d2dh_set_tap_bits -> d2d_params.tap_bits): the whole driver on the finer tap grid"""
    n = 4096 * 4 + 500
    chans = [synth("sine", n, seed=3), synth("pink", n, seed=4, amp=0.098)]
    src = str(tmp_path / "fine.dsf")
    write_dsf(src, chans, dsd_rate=1, lsb_first=True)
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    subprocess.check_call([cli, "-o", "w", "-r", "88200", "-b", "24", "-d", "T", "-p", str(out_dir), "-q", "--tap-bits", "32", src])
    fmt, pay = _wav_payload(str(out_dir / "fine.wav"))
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096,
                          bit_depth=24, dither="T", seed=0, tap_bits=32)
    r, rf = o.translate(pack_layout(chans, "P", 4096))
    assert rf == n * 8 // 32 and np.array_equal(pay, r[:rf * 6])
    p = subprocess.run([cli, "-o", "w", "-r", "96000", "-p", str(out_dir), "-q", "--tap-bits", "32", src], stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"32-bit taps serve the 44.1k-family rates" in p.stderr


@pytest.mark.gpu
def test_dff_to_aiff_and_raw_stdin_to_stdout(cli, oracle_mod, tmp_path):
    n = 4096 * 4
    chans = [synth("sine", n, seed=3, msb_first=True), synth("pink", n, seed=4, amp=0.098, msb_first=True)]
    src = str(tmp_path / "x.dff")
    write_dff(src, chans)
    subprocess.check_call([cli, "-o", "a", "-r", "176400", "-b", "16", "-d", "X", "-q", src])
    comm, pay = _aiff_payload(str(tmp_path / "x.aif"))
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=176400, channels=2, fmt="I", endianness="M", bit_depth=16, dither="X")
    r, rf = o.translate(pack_layout(chans, "I", 1))
    assert comm == (2, rf, 16)
    assert np.array_equal(pay.reshape(-1, 2)[:, ::-1].reshape(-1), r[:rf * 4])        # AIFF is big-endian
    # the README's pipe example: raw planar LSB-first on stdin, f32 on stdout (build_test_stereo_flt.sh)
    raw = pack_layout([synth("sine", n, seed=5), synth("sine", n, seed=6, freq=3000.0)], "P", 4096)
    got = subprocess.run([cli, "-f", "P", "-e", "L", "-b", "32", "-d", "F", "--level=-3", "-q"], input=raw.tobytes(),
                         stdout=subprocess.PIPE, check=True).stdout
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=352800, channels=2, fmt="P", endianness="L", block_size=4096,
                          bit_depth=32, dither="F", level_db=-3.0, seed=0)
    r, rf = o.translate(raw)
    assert np.array_equal(np.frombuffer(got, dtype=np.uint8), r[:rf * 8])


@pytest.mark.gpu
def test_flac_and_wav20_sinks(cli, oracle_mod, tmp_path):
    n = 4096 * 6
    chans = [synth("sine", n, seed=7), synth("pink", n, seed=8, amp=0.098)]
    src = str(tmp_path / "y.dsf")
    write_dsf(src, chans)
    subprocess.check_call([cli, "-o", "f", "-r", "88200", "-b", "24", "-d", "T", "-q", src])
    rate, ch, bps, pcm = _flac_decode(str(tmp_path / "y.flac"))
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096,
                          bit_depth=24, dither="T", seed=0)
    r, rf = o.translate(pack_layout(chans, "P", 4096))
    assert (rate, ch, bps, len(pcm)) == (88200, 2, 24, rf)
    assert np.array_equal(pcm, decode_pcm(r[:rf * 6], 24, 2))
    assert open(str(tmp_path / "y.flac"), "rb").read()[26:42] == hashlib.md5(r[:rf * 6].tobytes()).digest()   # STREAMINFO MD5
    assert os.path.getsize(str(tmp_path / "y.flac")) < rf * 6           # the fixed predictor + Rice coding do compress
    subprocess.check_call([cli, "-o", "w", "-r", "88200", "-b", "20", "-d", "X", "-q", src])
    fmt, pay = _wav_payload(str(tmp_path / "y.wav"))
    assert fmt[0] == 0xFFFE and fmt[5] == 24                              # EXTENSIBLE: 20 valid bits in 24
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096, bit_depth=20, dither="X")
    r, rf = o.translate(pack_layout(chans, "P", 4096))
    assert np.array_equal(pay, r[:rf * 6])


@pytest.mark.gpu
def test_levels_tool_and_error_exit(cli, oracle_mod, tmp_path):
    n = 4096 * 8
    a = [synth("sine", n, seed=9, amp=0.5), synth("sine", n, seed=10, amp=0.2)]
    b = [synth("pink", n, seed=11, amp=0.098)]
    pa, pb = str(tmp_path / "a.dsf"), str(tmp_path / "b.dsf")
    write_dsf(pa, a)
    write_dsf(pb, b)
    out = subprocess.check_output([cli, "levels", "-r", "88200", pa, pb]).decode().splitlines()
    want = []
    for chans in (a, b):
        o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=len(chans), fmt="P", endianness="L", block_size=4096,
                              bit_depth=32, dither="X")
        o.translate(pack_layout(chans, "P", 4096))
        want.append(float(o.peak_dbfs()))
    assert out[0] == "a.dsf: %.4f dBFS" % want[0] and out[1] == "b.dsf: %.4f dBFS" % want[1]
    assert out[2] == "Highest peak: %.4f dBFS" % max(want)
    # errors: message on stderr, failure exit code (src/lib.rs:26-35)
    p = subprocess.run([cli, "-r", "705600", "-o", "w", pa], stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"705600 output needs DSD128 or DSD256 input" in p.stderr
    p = subprocess.run([cli, "-d", "Q", pa], stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Invalid dither type; must be T, R, F, or X" in p.stderr


def _chunks(path, big_endian):
    b = open(path, "rb").read()
    pos, out = 12, {}
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack(">I" if big_endian else "<I", b[pos + 4:pos + 8])[0]
        out[cid] = b[pos + 8:pos + 8 + sz]
        pos += 8 + sz + (sz & 1)
    assert pos == len(b) and struct.unpack(">I" if big_endian else "<I", b[4:8])[0] == len(b) - 8
    return out


@pytest.mark.gpu
def test_tags_and_artwork_travel_with_the_audio(cli, oracle_mod, tmp_path):
    """README.md:7 (ID3v2 copied to the destination), :115-119 (-p copies artwork), :170-173 (-a marks the album)"""
    n = 4096 * 4
    chans = [synth("sine", n, seed=21), synth("pink", n, seed=22, amp=0.098)]
    tag = make_id3(3, [(b"TIT2", b"\x00tone"), (b"TPE1", b"\x00me"), (b"TALB", b"\x00Hits"), (b"TRCK", b"\x002"),
                       (b"APIC", b"\x00image/png\x00\x03\x00" + PNG)], padding=50)
    src_dir = tmp_path / "in" / "album"
    src_dir.mkdir(parents=True)
    src = str(src_dir / "t.dsf")
    write_dsf(src, chans, id3=tag)
    (src_dir / "folder.JPG").write_bytes(b"\xff\xd8 not really a jpeg")
    (src_dir / "notes.txt").write_bytes(b"not artwork")
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096,
                          bit_depth=24, dither="T", seed=0)
    r, rf = o.translate(pack_layout(chans, "P", 4096))
    plain = tag[:-50]
    plain = plain[:6] + _syncsafe(len(plain) - 10) + plain[10:]
    marked = make_id3(3, [(b"TIT2", b"\x00tone"), (b"TPE1", b"\x00me"), (b"TALB", b"\x00Hits [88.2K]"), (b"TRCK", b"\x002"),
                          (b"APIC", b"\x00image/png\x00\x03\x00" + PNG)])
    # WAV with -a and -p: tag chunk with the marked album, artwork copied, audio untouched
    subprocess.check_call([cli, "-o", "w", "-r", "88200", "-b", "24", "-d", "T", "-a", "-p", str(out_dir), "-q", src])
    wav = str(out_dir / "t_88_2K.wav")
    ck = _chunks(wav, False)
    assert ck[b"id3 "] == marked and np.array_equal(np.frombuffer(ck[b"data"], dtype=np.uint8), r[:rf * 6])
    assert (out_dir / "folder.JPG").read_bytes() == b"\xff\xd8 not really a jpeg" and not (out_dir / "notes.txt").exists()
    # AIFF without -a: the tag as it was (padding dropped)
    subprocess.check_call([cli, "-o", "a", "-r", "88200", "-b", "24", "-d", "T", "-q", src])
    ck = _chunks(str(src_dir / "t.aif"), True)
    assert ck[b"ID3 "] == plain
    assert np.array_equal(np.frombuffer(ck[b"SSND"][8:], dtype=np.uint8).reshape(-1, 3)[:, ::-1].reshape(-1), r[:rf * 6])
    # FLAC: Vorbis comments + picture, and the audio frames still decode to the same samples
    subprocess.check_call([cli, "-o", "f", "-r", "88200", "-b", "24", "-d", "T", "-a", "-q", src])
    rate, ch, bps, pcm = _flac_decode(str(src_dir / "t_88_2K.flac"))
    assert np.array_equal(pcm, decode_pcm(r[:rf * 6], 24, 2))
    blocks = dict(_flac_decode.blocks)
    vc = blocks[4]
    vlen = struct.unpack("<I", vc[:4])[0]
    cnt = struct.unpack("<I", vc[4 + vlen:8 + vlen])[0]
    pos, got = 8 + vlen, []
    for _ in range(cnt):
        ln = struct.unpack("<I", vc[pos:pos + 4])[0]
        got.append(vc[pos + 4:pos + 4 + ln].decode())
        pos += 4 + ln
    assert got == ["TITLE=tone", "ARTIST=me", "ALBUM=Hits [88.2K]", "TRACKNUMBER=2"] and pos == len(vc)
    pic = blocks[6]
    assert pic[:4] == struct.pack(">I", 3) and pic[8:17] == b"image/png" and pic.endswith(PNG)
    # a damaged tag: warning on stderr, conversion succeeds, no tag chunk
    bad = str(src_dir / "bad.dsf")
    write_dsf(bad, chans, id3=tag[:40])
    p = subprocess.run([cli, "-o", "w", "-r", "88200", "-b", "24", "-d", "T", bad], stderr=subprocess.PIPE, check=True)
    assert b"damaged" in p.stderr
    ck = _chunks(str(src_dir / "bad.wav"), False)
    assert b"id3 " not in ck and np.array_equal(np.frombuffer(ck[b"data"], dtype=np.uint8), r[:rf * 6])


@pytest.mark.gpu
def test_host_driver_c_abi_converts_like_the_cli(cli, engine_lib, oracle_mod, tmp_path):
    """include/rdsd2pcm_c.h (what the rdsd2pcm shim crate binds): from_container + do_conversion with a
    progress callback, check_level, cancel"""
    import ctypes as C
    engine_lib.lib()                                            # loads torch's HIP runtime first (see _capi.lib)
    L = C.CDLL(engine_lib.library_path())
    L.d2dh_last_error.restype = C.c_char_p
    L.d2dh_output_path.restype = C.c_char_p
    n = 4096 * 5
    chans = [synth("sine", n, seed=31), synth("pink", n, seed=32, amp=0.098)]
    src = str(tmp_path / "t.dsf")
    write_dsf(src, chans)
    h = C.c_void_p()
    rc = L.d2dh_from_container(24, ord("W"), C.c_double(0.0), 88200, None, ord("T"), ord("E"), 1, b".", src.encode(), C.byref(h))
    assert rc == 0, L.d2dh_last_error()
    seen = []
    PCB = C.CFUNCTYPE(None, C.c_void_p, C.c_float)
    pcb = PCB(lambda u, p: seen.append(p))
    assert L.d2dh_do_conversion(h, None, pcb, None) == 0, L.d2dh_last_error()
    assert seen and seen[-1] == 100.0 and all(a <= b for a, b in zip(seen, seen[1:]))
    out = L.d2dh_output_path(h).decode()
    assert out.endswith("t_88_2K.wav")
    L.d2dh_free(h)
    o = oracle_mod.Oracle(dsd_rate=1, output_rate=88200, channels=2, fmt="P", endianness="L", block_size=4096,
                          bit_depth=24, dither="T", seed=0)
    r, rf = o.translate(pack_layout(chans, "P", 4096))
    fmt, pay = _wav_payload(out)
    assert np.array_equal(pay, r[:rf * 6])
    # the level check (dsd_levels): same number as the CLI prints
    assert L.d2dh_new_level_check(88200, src.encode(), ord("P"), ord("L"), 2, 4096, 1, C.byref(h)) == 0
    db = C.c_float()
    assert L.d2dh_check_level(h, None, None, None, C.byref(db)) == 0
    L.d2dh_free(h)
    line = subprocess.check_output([cli, "levels", "-r", "88200", src]).decode().splitlines()[0]
    assert line == "t.dsf: %.4f dBFS" % db.value
    # a raised cancel flag stops the run with the reference's message
    assert L.d2dh_from_container(24, ord("W"), C.c_double(0.0), 88200, None, ord("T"), ord("E"), 0, b".", src.encode(), C.byref(h)) == 0
    flag = C.c_int(1)
    assert L.d2dh_do_conversion(h, C.byref(flag), None, None) == -30 and L.d2dh_last_error() == b"Conversion cancelled"
    L.d2dh_free(h)
