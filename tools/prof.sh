#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# 1) kernel-trace + stats  2) four PMC passes (separate runs, as the pool requires; each within one pass's slots, see tools/pmc.sh)
# tools/collect.py <tag> <name> then files the summaries under profiles/ and the HBM traffic into profiles/pmc_traffic.json
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-pcie --sustain 0 "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || true
for i in 1 2 3 4; do
  case $i in
    1) PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU";;
    2) PMC="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM";;
    3) PMC="FETCH_SIZE GRBM_GUI_ACTIVE";;
    4) PMC="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
  esac
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py --no-cpu-baseline --no-pcie --sustain 0 --reps 1 --steps 3 --warmup 1 "$@" > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err || true
done
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
