#!/bin/bash
# usage (GPU box, repo root): tools/quick.sh <tag> [bench args]: the pipelined-kernel parity test, then a short bench line
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined or depths or layouts" > $OUT/parity.txt 2>&1; tail -3 $OUT/parity.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-pcie --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel'],d['repetitions']['ms_per_step_all'])"
