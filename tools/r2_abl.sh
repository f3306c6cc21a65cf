#!/bin/bash
# usage: tools/r2_abl.sh <tag> "<dbg list>" "<waves list>"   (library built with DIAG=1)
TAG=${1:-abl}; DBGS=${2:-"0 1 2 4 3 5 6 7"}; WAVES=${3:-"16"}
mkdir -p gpurun_out/r2abl
for d in $DBGS; do
  for w in $WAVES; do
    D2D_DBG=$d D2D_MFMA_WAVES=$w timeout -k 10 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --distinct 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dbg $d waves $w kernel_ms', j['roofline']['kernel_ms'], j['roofline']['kernel'])"
  done
done | tee gpurun_out/r2abl/$TAG.txt
