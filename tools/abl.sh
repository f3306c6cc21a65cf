for w in 4 8; do
  D2D_MFMA_WAVES=$w python bench.py --steps 10 --warmup 2 --no-cpu-baseline --distinct 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves=$w', d['ms_per_step'], d['value'], d['config']['kernel'])"
done
