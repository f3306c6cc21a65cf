#!/bin/bash
# stage-B ablations at the full workload (DIAG build): D2D_DBG bits 256 no staging, 512 8 steps only, 1024 no quantise, 2048 no stores
for d in 0 256 512 1024 2048 3840 768 1792; do
  D2D_DBG=$d python bench.py --workload dsd64_to_96k_s24_stereo --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-pcie --distinct 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print($d, 'step kernels', r['step_kernels_ms'], 'fir', r['fir_kernel_ms'], 'rest', round(r['step_kernels_ms']-r['fir_kernel_ms'],3))"
done
