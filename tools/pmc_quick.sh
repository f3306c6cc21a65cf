#!/bin/bash
# needs a diagnostic build: make -C dsd2dxd_amd/csrc clean && make -C dsd2dxd_amd/csrc DIAG=1 (rebuild without DIAG afterwards)
# usage: tools/pmc_quick.sh <tag> "<counters>" [bench args]   (env is inherited: D2D_DBG, D2D_MFMA_NO_REG)
TAG=$1; PMC=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "d2d_fir" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$TAG", {k: "%.4g"%(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
