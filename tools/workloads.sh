#!/bin/bash
# bench lines of the other workloads (full 64 x 60 s size unless noted)
TAG=${1:-wl}; mkdir -p gpurun_out/$TAG   # usage: tools/workloads.sh <tag>
for w in dsd64_to_88k2_s16_stereo dsd64_to_88k2_f32_stereo dsd64_to_88k2_s24_stereo_nodither dsd64_to_176k4_s24_stereo dsd64_to_352k8_s24_stereo dsd64_to_352k8_f32_stereo dsd128_to_88k2_s24_stereo dsd128_to_88k2_s24_stereo_ns dsd64_to_96k_s24_stereo dsd64_to_192k_s24_stereo dsd128_to_384k_s24_stereo dsd512_to_96k_s24_8ch; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-pcie --sustain 0 --steps 5 --warmup 2 --reps 3 --distinct 8 > gpurun_out/$TAG/$w.json 2> gpurun_out/$TAG/$w.err; echo "$w rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/$TAG/$w.json")); r=j["roofline"]; print("   %.1f Gsamples/s  step kernels %.3f ms (fir %.3f)  frac %.3f  scope %s  %s" % (j["value"]/1e3, r["step_kernels_ms"], r["fir_kernel_ms"], r["frac"], r["scope"], j["config"]["kernel"]))
except Exception as e: print("   failed", e)
PY
done
