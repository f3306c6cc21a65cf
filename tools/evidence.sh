#!/bin/bash
# round evidence (GPU box, repo root): tools/evidence.sh <round tag, e.g. r03>
#   rocprofv3 kernel stats + PMC passes (tools/prof.sh) of the default bench command and of the other BASELINE shapes, the default bench line
#   with the CPU baseline and the PCIe-inclusive rate, the other workloads' bench lines.  File them with tools/collect.py afterwards.
R=${1:-r03}
mkdir -p gpurun_out/$R
timeout -k 10 500 python bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err; echo "default rc=$?"
bash tools/prof.sh ${R} --steps 5 --warmup 2 --reps 1 > gpurun_out/$R/prof_default.txt 2>&1
for w in dsd64_to_352k8_f32_stereo dsd128_to_88k2_s24_stereo_ns dsd512_to_96k_s24_8ch dsd64_to_96k_s24_stereo; do
  D=0; [ $w = dsd512_to_96k_s24_8ch ] && D=8   # (config 5: 64 distinct 8-channel files would be 87 GB of host memory)
  bash tools/prof.sh ${R}_$w --workload $w --steps 3 --warmup 1 --reps 1 --distinct $D > gpurun_out/$R/prof_$w.txt 2>&1; echo "$w profiled"
done
bash tools/workloads.sh ${R}_workloads > gpurun_out/$R/workloads.txt 2>&1
tail -30 gpurun_out/$R/workloads.txt
