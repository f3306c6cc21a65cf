#!/bin/bash
mkdir -p gpurun_out/ns
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "noise or golden or fuzz" 2>&1 | tail -4
timeout -k 10 300 python bench.py --workload dsd128_to_88k2_s24_stereo_ns --no-cpu-baseline --no-pcie --steps 5 --warmup 2 --reps 3 --distinct 8 > gpurun_out/ns/bench_ns.json 2> gpurun_out/ns/bench_ns.err; echo rc=$?
python - <<PY
import json; j=json.load(open("gpurun_out/ns/bench_ns.json")); print(j["value"], "Msamples/s; step kernels", j["roofline"]["step_kernels_ms"], "fir", j["roofline"]["fir_kernel_ms"], "frac", j["roofline"]["frac"], j["roofline"]["scope"])
PY
