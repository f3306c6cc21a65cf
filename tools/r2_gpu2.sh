#!/bin/bash
# parity (all gpu tests) + headline bench
TAG=${1:-r2b}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/$TAG/bench_$i.json 2> gpurun_out/$TAG/bench_$i.err; python - <<PY
import json; j=json.load(open("gpurun_out/$TAG/bench_$i.json")); print("bench $i:", j["value"], "Msamples/s", j["roofline"]["kernel"], j["roofline"]["kernel_ms"], "ms frac", j["roofline"]["frac"])
PY
done
