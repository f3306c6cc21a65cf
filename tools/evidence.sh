#!/bin/bash
# round evidence (GPU box, repo root): tools/evidence.sh <round tag, e.g. r04> <part>
#   prof1 | prof2   rocprofv3 kernel stats + PMC passes (tools/prof.sh) of the default bench command and of the other judged shapes
#   workloads       the other workloads' bench lines (tools/workloads.sh)
#   default         the default bench line with the CPU baseline and the PCIe-inclusive rate -- run it AFTER tools/collect.py has filed the
#                   default profile, so that the line cites the counter traffic stamped with these kernel sources
# One gpurun call per part (a call is limited to 20 minutes); file the profiles with tools/collect.py afterwards.
R=${1:-r04}; PART=${2:-prof1}
mkdir -p gpurun_out/$R
prof() {   # prof <workload> [extra bench args]
  local w=$1; shift
  bash tools/prof.sh ${R}_$w --workload $w --steps 3 --warmup 1 --reps 1 "$@" > gpurun_out/$R/prof_$w.txt 2>&1; echo "$w profiled"
}
case $PART in
  prof1)
    bash tools/prof.sh ${R} --steps 5 --warmup 2 --reps 1 > gpurun_out/$R/prof_default.txt 2>&1; echo "default profiled"
    prof dsd64_to_96k_s24_stereo; prof dsd64_to_192k_s24_stereo; prof dsd128_to_384k_s24_stereo; prof dsd64_to_352k8_f32_stereo ;;
  prof2)
    prof dsd128_to_88k2_s24_stereo_ns
    prof dsd512_to_96k_s24_8ch --distinct 8      # (config 5: 64 distinct 8-channel files would be 87 GB of host memory)
    prof dsd64_to_88k2_s24_6ch                   # whole 5.1 frames from one wave
    bash tools/prof.sh ${R}_taps32 --workload dsd64_to_88k2_s24_stereo --tap-bits 32 --steps 3 --warmup 1 --reps 1 > gpurun_out/$R/prof_taps32.txt 2>&1; echo "taps32 profiled" ;;   # 32-bit taps in one pass
  workloads)
    bash tools/workloads.sh ${R}_workloads > gpurun_out/$R/workloads.txt 2>&1; tail -40 gpurun_out/$R/workloads.txt ;;
  default)
    timeout -k 10 500 python bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err; echo "default rc=$?"; tail -c 1500 gpurun_out/$R/bench_default.json ;;
esac
