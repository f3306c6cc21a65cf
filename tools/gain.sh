#!/bin/bash
# usage (GPU box, repo root): tools/gain.sh <tag>: a level other than 0 dB, pipelined epilogue against the older kernels (bench.py --debug 2 = D2D_DBG_NO_GAINQ)
OUT=gpurun_out/$1; mkdir -p $OUT
for w in dsd64_to_352k8_s24_stereo dsd64_to_176k4_s24_stereo dsd64_to_352k8_f32_stereo dsd64_to_88k2_s24_stereo; do
  for off in 0 1; do
    timeout -k 10 200 python bench.py --debug $((off * 2)) --workload $w --level -3 --steps 20 --warmup 5 --reps 3 --no-pcie --no-cpu-baseline > $OUT/${w}_$off.json 2> $OUT/${w}_$off.err
    python3 -c "
import json;d=json.loads(open('$OUT/${w}_$off.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$w', 'NO_GAINQ=$off', d['ms_per_step'], r['frac'], d['config']['kernel'])"
  done
done
