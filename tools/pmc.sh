#!/bin/bash
# per-dispatch PMC means of one kernel (GPU box, repo root):  tools/pmc.sh <tag> <kernel-name-substring> "<counters>" [bench args]
#   LIB=<name under ab/> runs a variant library.  One pass holds 8 SQ, 4 TCC (FETCH_SIZE costs 3, WRITE_SIZE 2) and 2 GRBM counters
#   (MI355X_MICROARCH.md): a longer list makes rocprofv3 abort in rocprofiler_create_counter_config (error 38) and burns the lease,
#   so it is refused here.
TAG=$1; KSUB=$2; PMC=$3; shift; shift; shift
sq=0; tcc=0; grbm=0
for c in $PMC; do
  case $c in
    SQ_*) sq=$((sq+1));; GRBM_*) grbm=$((grbm+1));;
    FETCH_SIZE) tcc=$((tcc+3));; WRITE_SIZE) tcc=$((tcc+2));; TCC_*|TCP_*) tcc=$((tcc+1));;
  esac
done
if [ $sq -gt 8 ] || [ $tcc -gt 4 ] || [ $grbm -gt 2 ]; then echo "tools/pmc.sh: \"$PMC\" needs $sq SQ / $tcc TCC / $grbm GRBM slots; one pass has 8 / 4 / 2 -- split it"; exit 2; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
[ -n "$LIB" ] && export D2D_AMD_LIB=$GRAFT_REPO_ROOT/ab/$LIB/libdsd2dxd_amd.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-pcie --sustain 0 --reps 1 --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "$KSUB" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$TAG", {k: "%.4g"%(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
