#!/bin/bash
# A/B: usage tools/r2_ab.sh <tag> ; env variants listed below
TAG=${1:-ab}; mkdir -p gpurun_out/$TAG
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 $EXTRA > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; python - <<PY
import json; j=json.load(open("gpurun_out/$TAG/$name.json")); print("%-28s" % "$name", j["roofline"]["kernel"], j["roofline"]["kernel_ms"], "ms frac", j["roofline"]["frac"], "step", j["ms_per_step"])
PY
}
EXTRA="--distinct 8"; run v2_d8 A=1; run v2_d8_w4 D2D_MFMA_WAVES=4
EXTRA=""; run v2_d64 A=1
