"""Committed golden vectors (tests/golden, made by tests/golden/make_golden.py from this repo's
oracle -- parity with the reference itself is unpinned, see that script's header).

CPU: the oracle still reproduces them byte for byte (and the fixture-derived digests when the
reference's fixtures are present).  GPU (-m gpu): both kernels reproduce them through the C ABI with
no oracle in the loop.
"""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(HERE, "golden_meta.json")))
VEC = np.load(os.path.join(HERE, "golden_vectors.npz"))
CASES = sorted(k for k in META if not k.startswith("_"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle_mod, name):
    m = META[name]
    o = oracle_mod.Oracle(**m["kw"])
    for i in range(m["calls"]):
        pcm, fr = o.translate(VEC[f"{name}/in{i}"])
        assert fr == m["frames"][i]
        assert np.array_equal(pcm[:fr * o.frame_bytes], VEC[f"{name}/out{i}"])
    assert [o.peak(c) for c in range(m["kw"]["channels"])] == m["peaks"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/test"), reason="reference fixtures not present")
def test_oracle_reproduces_fixture_digests(oracle_mod):
    for fname, m in META["_fixtures"].items():
        raw = np.fromfile(os.path.join("/root/reference/test", fname), dtype=np.uint8)[m["skip"]:][:m["nbytes"]]
        o = oracle_mod.Oracle(**m["kw"])
        pcm, fr = o.translate(raw)
        pcm = pcm[:fr * o.frame_bytes]
        assert fr == m["frames"]
        assert hashlib.sha256(pcm.tobytes()).hexdigest() == m["sha256"]
        assert pcm[:96].tolist() == m["head"]


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [pytest.param(1, id="lut"), pytest.param(2, id="mfma")])
@pytest.mark.parametrize("name", CASES)
def test_engine_reproduces_golden(engine_lib, name, kernel):
    m = META[name]
    e = engine_lib.Engine(kernel=kernel, **m["kw"])
    for i in range(m["calls"]):
        pcm, fr = e.translate(VEC[f"{name}/in{i}"])
        assert fr == m["frames"][i]
        assert np.array_equal(pcm, VEC[f"{name}/out{i}"])
    assert [e.peak(c) for c in range(m["kw"]["channels"])] == m["peaks"]
    assert e.peak_dbfs() == np.float32(m["peak_dbfs"])
