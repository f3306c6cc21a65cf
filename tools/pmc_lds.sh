#!/bin/bash
# LDS / wait counters of the FIR kernel: usage tools/pmc_lds.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
           "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $OUT/b$i.json 2> $OUT/e$i.err || true
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fir_mfma" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print("%-32s mean %.4g (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
