#!/bin/bash
# rehearsal of the N-rank bench on the one-GPU box: gloo collectives, ranks folded onto GPU 0
mkdir -p gpurun_out/ranks
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --cpu-budget 6 > gpurun_out/ranks/n1.json 2> gpurun_out/ranks/n1.err; echo "n1 rc=$?"; cat gpurun_out/ranks/n1.json
D2D_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --files 24 --seconds 20 --steps 4 --warmup 1 --reps 3 --sustain 0 --no-cpu-baseline > gpurun_out/ranks/n2_gloo.json 2> gpurun_out/ranks/n2_gloo.err; echo "n2 rc=$?"; cat gpurun_out/ranks/n2_gloo.json; tail -3 gpurun_out/ranks/n2_gloo.err
D2D_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --shard channels --workload dsd512_to_96k_s24_8ch --files 2 --seconds 5 --steps 3 --warmup 1 --reps 3 --sustain 0 --no-cpu-baseline > gpurun_out/ranks/n4_channels_gloo.json 2> gpurun_out/ranks/n4_channels_gloo.err; echo "n4 rc=$?"; cat gpurun_out/ranks/n4_channels_gloo.json; tail -3 gpurun_out/ranks/n4_channels_gloo.err
