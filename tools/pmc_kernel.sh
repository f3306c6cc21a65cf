#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-name-substring> "<counters>" [bench args]: per-dispatch PMC means of one kernel
TAG=$1; KSUB=$2; PMC=$3; shift; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-pcie --reps 1 --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "$KSUB" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$TAG", {k: "%.4g"%(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
