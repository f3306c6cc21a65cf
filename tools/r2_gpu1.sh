#!/bin/bash
# round 2, GPU call 1: parity of the two-group kernel, then A/B bench against the one-group kernel
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --distinct 8 > gpurun_out/r2a/bench_v2.json 2> gpurun_out/r2a/bench_v2.err; echo "v2 rc=$?"; cat gpurun_out/r2a/bench_v2.json
D2D_NO_INTQ=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --distinct 8 > gpurun_out/r2a/bench_v2_f64epi.json 2> gpurun_out/r2a/bench_v2_f64epi.err; echo "v2 f64 epilogue rc=$?"; cat gpurun_out/r2a/bench_v2_f64epi.json
D2D_MFMA_V1=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --distinct 8 > gpurun_out/r2a/bench_v1.json 2> gpurun_out/r2a/bench_v1.err; echo "v1 rc=$?"; cat gpurun_out/r2a/bench_v1.json
D2D_MFMA_WAVES=12 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --distinct 8 > gpurun_out/r2a/bench_v2_w12.json 2> gpurun_out/r2a/bench_v2_w12.err; echo "v2 12 waves rc=$?"; cat gpurun_out/r2a/bench_v2_w12.json
D2D_MFMA_WAVES=8 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --distinct 8 > gpurun_out/r2a/bench_v2_w8.json 2> gpurun_out/r2a/bench_v2_w8.err; echo "v2 8 waves rc=$?"; cat gpurun_out/r2a/bench_v2_w8.json
