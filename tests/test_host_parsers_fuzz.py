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
