#!/bin/bash
cp dsd2dxd_amd/libdsd2dxd_amd.so /tmp/lib_orig.so
for f in dsd2dxd_amd/lib_flag*.so.keep; do
  cp $f dsd2dxd_amd/libdsd2dxd_amd.so
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 --reps 3 --distinct 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$f', j['roofline']['kernel_ms'], j['config']['kernel'])"
  done
done
cp /tmp/lib_orig.so dsd2dxd_amd/libdsd2dxd_amd.so
python bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 --reps 3 --distinct 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('orig', j['roofline']['kernel_ms'], j['config']['kernel'])"
