python bench.py --steps 20 --warmup 3 --no-cpu-baseline --distinct 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bound4', d['ms_per_step'], d['value'])"
# needs a diagnostic build: make -C dsd2dxd_amd/csrc clean && make -C dsd2dxd_amd/csrc DIAG=1 (rebuild without DIAG afterwards)
