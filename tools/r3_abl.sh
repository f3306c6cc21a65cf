#!/bin/bash
# compile-time ablations of the pipelined kernel (tools/ab_build.sh <name> -DD2D_M3_ABL=<mask>): usage tools/r3_abl.sh <tag> <lib names...>
TAG=$1; shift; mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for n in tree "$@"; do
  if [ "$n" = tree ]; then L=""; else L=$PWD/ab/$n/libdsd2dxd_amd.so; fi
  D2D_AMD_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 --reps 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$n kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"
done | tee gpurun_out/$TAG/abl.txt
if [ -e ab/st/libdsd2dxd_amd.so ]; then D2D_AMD_LIB=$PWD/ab/st/libdsd2dxd_amd.so python tools/stamps3.py 2>&1 | tail -3 | tee -a gpurun_out/$TAG/abl.txt; fi
