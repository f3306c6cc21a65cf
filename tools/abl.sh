for w in 4 8 16; do
  D2D_MFMA_NO_REG=1 D2D_MFMA_WAVES=$w python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('noreg waves=$w', d['ms_per_step'], d['value'], d['config']['kernel'])"
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reg', d['ms_per_step'], d['value'], d['config']['kernel'])"
