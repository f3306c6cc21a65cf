#!/bin/bash
# A/B on the GPU box: the M = 8 shapes with more waves per SIMD.  Lines: <label> <lib|tree> <D2D_MFMA_WAVES|-> <workload>
mkdir -p gpurun_out/occ
run() {
  local label=$1 lib=$2 waves=$3 wl=$4
  local L=""; [ "$lib" != tree ] && L=$PWD/ab/$lib/libdsd2dxd_amd.so
  local W=""; [ "$waves" != - ] && W=$waves
  D2D_AMD_LIB=$L D2D_MFMA_WAVES=$W timeout -k 10 300 python bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 --reps 3 --workload $wl > gpurun_out/occ/$label.json 2> gpurun_out/occ/$label.err || { echo "$label FAILED"; tail -3 gpurun_out/occ/$label.err; return 1; }
  python - <<PY
import json; j=json.load(open("gpurun_out/occ/$label.json")); r=j["roofline"]
print("%-22s" % "$label", r.get("kernel"), "kernel_ms", r.get("kernel_ms", r.get("fir_kernel_ms")), "frac", r["frac"], "ms_per_step", j["ms_per_step"], "Gs/s", round(j["value"]/1e3,1))
PY
}
while read -r label lib waves wl; do
  [ -z "$label" ] && continue
  run $label $lib $waves $wl || exit 1
done
