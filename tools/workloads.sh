#!/bin/bash
# bench lines of the other workloads (full 64 x 60 s size, 64 distinct files, unless noted)
TAG=${1:-wl}; mkdir -p gpurun_out/$TAG   # usage: tools/workloads.sh <tag>
run() {   # run <name> <bench args...>
  local n=$1; shift
  timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-pcie --sustain 0 --steps 5 --warmup 2 --reps 3 > gpurun_out/$TAG/$n.json 2> gpurun_out/$TAG/$n.err; echo "$n rc=$?"
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/$TAG/$n.json")); r=j["roofline"]; print("   %.1f Gsamples/s  step kernels %.3f ms (fir %.3f)  frac %.3f  scope %s  %s" % (j["value"]/1e3, r["step_kernels_ms"], r["fir_kernel_ms"], r["frac"], r["scope"], j["config"]["kernel"]))
except Exception as e: print("   failed", e)
PY
}
for w in dsd64_to_88k2_s16_stereo dsd64_to_88k2_f32_stereo dsd64_to_88k2_s24_stereo_nodither dsd64_to_176k4_s24_stereo dsd64_to_352k8_s24_stereo dsd64_to_352k8_f32_stereo dsd128_to_88k2_s24_stereo dsd128_to_88k2_s24_stereo_ns dsd64_to_96k_s24_stereo dsd64_to_192k_s24_stereo dsd128_to_384k_s24_stereo dsd128_to_96k_s24_stereo dsd64_to_88k2_s24_stereo_dff dsd64_to_352k8_s24_stereo_dff dsd64_to_96k_s24_stereo_dff; do
  run $w --workload $w
done
run dsd512_to_96k_s24_8ch --workload dsd512_to_96k_s24_8ch --distinct 8             # (64 distinct 8-channel files would be 87 GB of host memory)
run dsd512_to_96k_s24_8ch_rank0of8 --workload dsd512_to_96k_s24_8ch --distinct 8 --shard channels --as-rank 0/8     # config 5's own partition: one channel per GPU ...
run dsd512_to_96k_s24_8ch_rank0of4 --workload dsd512_to_96k_s24_8ch --distinct 8 --shard channels --as-rank 0/4     # ... or a pair
run dsd64_to_88k2_s24_6ch --workload dsd64_to_88k2_s24_6ch
run dsd64_to_96k_s24_6ch --workload dsd64_to_96k_s24_6ch
run dsd64_to_88k2_s24_4ch --workload dsd64_to_88k2_s24_4ch
run dsd512_to_352k8_s24_8ch --workload dsd512_to_352k8_s24_8ch --distinct 8      # eight channels byte-interleaved: the planar pre-pass, then whole frames from one wave (four pairs)
run dsd64_to_88k2_s24_mono --workload dsd64_to_88k2_s24_mono --files 128
run dsd64_to_352k8_s24_mono_level4 --workload dsd64_to_352k8_s24_mono --files 128 --level 4
run dsd256_to_88k2_s24_stereo --workload dsd256_to_88k2_s24_stereo --seconds 30      # M = 128
run dsd256_to_176k4_s24_stereo --workload dsd256_to_176k4_s24_stereo --seconds 30
run dsd256_to_192k_s24_stereo --workload dsd256_to_192k_s24_stereo --seconds 30
run dsd64_to_88k2_s24_stereo_level-3 --workload dsd64_to_88k2_s24_stereo --level -3
run dsd64_to_352k8_s24_stereo_level-3 --workload dsd64_to_352k8_s24_stereo --level -3
run dsd64_to_88k2_s24_stereo_taps32 --workload dsd64_to_88k2_s24_stereo --tap-bits 32
run dsd128_to_88k2_s24_stereo_taps32 --workload dsd128_to_88k2_s24_stereo --tap-bits 32
