#!/bin/bash
# usage (GPU box, repo root): tools/dff.sh <tag>: the byte-interleaved stereo shapes (DFF / -f I) beside their planar twins
OUT=gpurun_out/$1; mkdir -p $OUT
for w in dsd64_to_88k2_s24_stereo dsd64_to_88k2_s24_stereo_dff dsd64_to_352k8_s24_stereo dsd64_to_352k8_s24_stereo_dff; do
  timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --reps 3 --no-pcie --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err
  python3 -c "
import json;d=json.loads(open('$OUT/$w.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$w', d['ms_per_step'], 'fir', r['fir_kernel_ms'], 'step', r['step_kernels_ms'], r['frac'], r['scope'], d['config']['kernel'])"
done
