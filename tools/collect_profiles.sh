#!/bin/bash
# copy the artefacts of tools/r2_profile.sh (gpurun_out/) into the tracked profiles/ directory as the round-2 evidence set
set -e
R=r02
cp gpurun_out/prof_r02_console.txt profiles/${R}_summary.txt
cp $(ls -t gpurun_out/prof_r02/trace/*/*_kernel_stats.csv | head -1) profiles/${R}_kernel_stats.csv
cp gpurun_out/prof_r02/bench_trace.json profiles/${R}_bench_under_rocprof_trace.json
cp gpurun_out/r02_bench_default.json profiles/${R}_bench_default.json
cat gpurun_out/r02_workloads/*.json > profiles/${R}_bench_other_workloads.json
cat gpurun_out/ranks/n1.json gpurun_out/ranks/n2_gloo.json gpurun_out/ranks/n4_channels_gloo.json > profiles/${R}_bench_ranks_rehearsal.json
python3 - <<'PY'
import json
cur = json.load(open("profiles/pmc_traffic.json"))
new = json.load(open("gpurun_out/pmc_traffic_r02.json"))
cur.update(new)
json.dump(cur, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
echo collected
