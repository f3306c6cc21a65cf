"""The container and tag parsers of the host driver read untrusted files: a mutation fuzzer over
synthetic DSF / DFF files with ID3v2 tags, built with AddressSanitizer + UBSan (CPU only)."""
import os
import struct
import subprocess

import pytest

from helpers import random_bytes, write_dff, write_dsf
from test_host_driver import make_id3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_survive_mutated_files(tmp_path):
    exe = str(tmp_path / "fuzz")
    src = [os.path.join(ROOT, "tools", "fuzz_host_parsers.cpp"),
           os.path.join(ROOT, "dsd2dxd_amd", "csrc", "host", "dsd_reader.cpp"),
           os.path.join(ROOT, "dsd2dxd_amd", "csrc", "host", "id3_tag.cpp")]
    cmd = ["g++", "-g", "-O1", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe] + src
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "tools"), capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    chans = [random_bytes(4096 * 2 + 100, 1), random_bytes(4096 * 2 + 100, 2)]
    tag = make_id3(3, [(b"TIT2", b"\x00tone"), (b"TALB", b"\x01\xff\xfeH\x00i\x00"), (b"APIC", b"\x00image/png\x00\x03c\x00" + bytes(50)),
                       (b"COMM", b"\x00engd\x00text")], padding=40)
    dsf, dff = str(tmp_path / "base.dsf"), str(tmp_path / "base.dff")
    write_dsf(dsf, chans, id3=tag)
    write_dff(dff, chans, tail=b"ID3 " + struct.pack(">Q", len(tag)) + tag)
    for seed_file, seed in ((dsf, 1), (dff, 2)):
        p = subprocess.run([exe, seed_file, str(seed), "400", str(tmp_path)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "fuzz ok" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


def test_dff_chunk_sizes_near_2_64_do_not_hang(tmp_path):
    """a chunk size such as 0xFFFFFFFFFFFFFFF4 wrapped the walker's position back into the file and
    probe() never returned; sub-chunk sizes inside PROP and the DSD chunk's size had the same arithmetic."""
    exe = str(tmp_path / "probe1")
    src = [os.path.join(ROOT, "tools", "fuzz_host_parsers.cpp"),
           os.path.join(ROOT, "dsd2dxd_amd", "csrc", "host", "dsd_reader.cpp"),
           os.path.join(ROOT, "dsd2dxd_amd", "csrc", "host", "id3_tag.cpp")]
    r = subprocess.run(["g++", "-g", "-O1", "-std=c++17", "-o", exe] + src, cwd=os.path.join(ROOT, "tools"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    chans = [random_bytes(4096 + 10, 1), random_bytes(4096 + 10, 2)]
    base = str(tmp_path / "base.dff")
    write_dff(base, chans)
    blob = bytearray(open(base, "rb").read())
    huge = [0xFFFFFFFFFFFFFFF4, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFEC, 1 << 63]
    # every 12-byte chunk header of the file (top level and inside PROP) gets each of the huge sizes in turn
    offs = [16]
    pos = 16
    while pos + 12 <= len(blob):
        cid = bytes(blob[pos:pos + 4]); sz = struct.unpack(">Q", blob[pos + 4:pos + 12])[0]
        if cid == b"PROP":
            q = pos + 16
            while q + 12 <= pos + 12 + sz:
                offs.append(q)
                q += 12 + struct.unpack(">Q", blob[q + 4:q + 12])[0]
        pos += 12 + sz + (sz & 1)
        if pos + 12 <= len(blob):
            offs.append(pos)
    assert len(offs) >= 5
    for k, off in enumerate(offs):
        for h in huge:
            b = bytearray(blob)
            b[off + 4:off + 12] = struct.pack(">Q", h)
            f = str(tmp_path / f"bad_{k}.dff")
            open(f, "wb").write(b)
            # zero mutation iterations: the driver probes and reads the seed file as it is; must come back quickly
            p = subprocess.run([exe, f, "1", "0", str(tmp_path)], capture_output=True, text=True, timeout=20)
            assert p.returncode == 0 and "fuzz ok" in p.stdout, (k, hex(h), p.stderr[-500:])
