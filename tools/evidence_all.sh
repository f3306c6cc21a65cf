#!/bin/bash
# The whole round's evidence in one go, from the development container (not the GPU box): four gpurun calls (tools/evidence.sh prof1, prof2, workloads, default),
# each retried while the pool is busy, the profiles filed by tools/collect.py in between so that the default line, taken last, cites the counter traffic
# stamped with these kernel sources.  Do not edit dsd2dxd_amd/csrc while it runs (every call snapshots the tree).   usage: tools/evidence_all.sh
cd "$(dirname "$0")/.."
rm -rf gpurun_out/prof_r04* gpurun_out/r04 gpurun_out/r04_workloads
G() { for i in 1 2 3 4 5 6 7 8; do /usr/local/graft/bin/gpurun "$@" > gpurun_out/.evidence_call.out 2>&1; if grep -q "status=transient" gpurun_out/.evidence_call.out; then sleep 150; continue; fi; cat gpurun_out/.evidence_call.out; return 0; done; cat gpurun_out/.evidence_call.out; return 1; }
mvset() { for f in summary.txt kernel_stats.csv bench_under_rocprof_trace.json; do b=${f%.*}; e=${f##*.}; mv profiles/r04_$1_$f profiles/r04_${b}_$1.$e; done; }
G --timeout 1200 -- 'tools/evidence.sh r04 prof1 2>&1 | tail -6' 2>&1 | tail -7 || exit 1
python tools/collect.py r04 r04 | head -3
for pair in "dsd64_to_96k_s24_stereo 96k" "dsd64_to_192k_s24_stereo 192k" "dsd128_to_384k_s24_stereo 384k" "dsd64_to_352k8_f32_stereo c2"; do set -- $pair; python tools/collect.py r04_$1 r04_$2 | head -1; mvset $2; done
G --timeout 1200 -- 'tools/evidence.sh r04 prof2 2>&1 | tail -6' 2>&1 | tail -7 || exit 1
for pair in "dsd128_to_88k2_s24_stereo_ns c3" "dsd512_to_96k_s24_8ch c5" "dsd64_to_88k2_s24_6ch 6ch"; do set -- $pair; python tools/collect.py r04_$1 r04_$2 | head -1; mvset $2; done
python tools/collect.py r04_taps32 r04_taps32 | head -1; mvset taps32
G --timeout 1200 -- 'tools/evidence.sh r04 workloads 2>&1 | tail -64' 2>&1 | tail -66 || exit 1
python - <<'PY'
import glob, json
out=[]
for f in sorted(glob.glob('gpurun_out/r04_workloads/*.json')):
    ls=[l for l in open(f).read().splitlines() if l.startswith('{')]
    if ls: out.append(ls[-1])
    else: print('EMPTY', f)
open('profiles/r04_bench_other_workloads.json','w').write('\n'.join(out)+'\n')
print(len(out), 'lines')
PY
G --timeout 900 -- 'tools/evidence.sh r04 default 2>&1 | tail -3' 2>&1 | tail -5 || exit 1
cp gpurun_out/r04/bench_default.json profiles/r04_bench_default.json
python tools/bench_table.py r04 > gpurun_out/bench_table.md; head -3 gpurun_out/bench_table.md
echo EVIDENCE DONE
