#!/bin/bash
# A/B on the GPU box: tools/ab_run.sh <tag> <lib name ...>   ("tree" = the in-tree library); EXTRA = extra bench.py arguments
TAG=$1; shift; mkdir -p gpurun_out/$TAG
for n in "$@"; do
  if [ "$n" = tree ]; then L=""; else L=$PWD/ab/$n/libdsd2dxd_amd.so; fi
  D2D_AMD_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-pcie --sustain 0 --steps 10 --warmup 2 --reps 3 $EXTRA > gpurun_out/$TAG/$n.json 2> gpurun_out/$TAG/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/$TAG/$n.err; continue; }
  python - <<PY
import json; j=json.load(open("gpurun_out/$TAG/$n.json")); r=j["roofline"]
print("%-14s" % "$n", r.get("kernel"), "kernel_ms", r.get("kernel_ms", r.get("fir_kernel_ms")), "frac", r["frac"], "ms_per_step", j["ms_per_step"], "value", j["value"])
PY
done
