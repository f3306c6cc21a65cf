#!/bin/bash
# pipelined kernel: parity tests, then A/B bench: sparse chain (default), dense chain (D2D_NO_SPARSE=1), two-group kernel (D2D_NO_PIPE=1)
TAG=${1:-r3a}; TESTS=${2:-"tests/test_gpu_parity.py tests/test_gpu_api.py"}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest $TESTS -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/$TAG/pytest.log
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 --reps 3 $EXTRA > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { echo "$name FAILED"; tail -3 gpurun_out/$TAG/$name.err; return; }
python - <<PY
import json; j=json.load(open("gpurun_out/$TAG/$name.json")); r=j["roofline"]
print("%-14s" % "$name", r.get("kernel"), "kernel_ms", r.get("kernel_ms", r.get("fir_kernel_ms")), "frac", r["frac"], "ms_per_step", j["ms_per_step"], "value", j["value"])
PY
}
run sparse A=1; run dense D2D_NO_SPARSE=1; run nopipe D2D_NO_PIPE=1; run sparse2 A=1
