#!/bin/bash
# needs a diagnostic build: make -C dsd2dxd_amd/csrc clean && make -C dsd2dxd_amd/csrc DIAG=1 (rebuild without DIAG afterwards)
# resampler ablations (diagnostic): D2D_DBG bits 256 no staging, 512 8 steps only, 1024 no quantise, 2048 no stores
for d in 0 256 512 1024 2048 3840 768; do
  D2D_DBG=$d python bench.py --workload dsd64_to_96k_s24_stereo --steps 5 --warmup 2 --no-cpu-baseline --distinct 8 --files 32 --seconds 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($d, d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"], round(d[\"ms_per_step\"]-d[\"roofline\"][\"kernel_ms\"],3))"
done
