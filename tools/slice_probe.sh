#!/bin/bash
# Does a step get faster per sample when its working set fits the 256 MB Infinity Cache?  (decides whether slicing a call in time pays)
mkdir -p gpurun_out
out=gpurun_out/slice_probe.txt
: > $out
for wl in dsd512_to_96k_s24_8ch dsd64_to_96k_s24_stereo; do
  for s in 60 4 1 0.5 0.25 0.1 0.05; do
    echo "== $wl seconds=$s" >> $out
    timeout -k 10 200 python3 bench.py --workload $wl --seconds $s --steps 20 --warmup 3 --reps 3 --no-cpu-baseline --no-pcie --distinct 8 2>>gpurun_out/slice_probe.err | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d['roofline']; print(d['value'], d['ms_per_step'], r.get('fir_kernel_ms'), r.get('step_kernels_ms'))
" >> $out || exit 1
    echo "$wl $s done"
  done
done
cat $out
