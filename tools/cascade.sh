#!/bin/bash
# usage (GPU box, repo root): tools/cascade.sh <tag>: the 48k-family shapes and the noise-shaped one (stage A in the scratch flavour)
OUT=gpurun_out/$1; mkdir -p $OUT
for w in dsd64_to_96k_s24_stereo dsd64_to_192k_s24_stereo dsd128_to_384k_s24_stereo dsd128_to_88k2_s24_stereo_ns; do
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --reps 3 --no-pcie --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err
  python3 -c "
import json;d=json.loads(open('$OUT/$w.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$w', d['ms_per_step'], 'fir', r['fir_kernel_ms'], 'step', r['step_kernels_ms'], r['frac'], d['config']['kernel'])"
done
