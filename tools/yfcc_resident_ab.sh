for i in 1 2; do
python bench.py --workload yfcc --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident  :', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms/step')"
PATS_AMD_DIAG_LIB=1 PATS_STREAM_RESIDENT=0 python bench.py --workload yfcc --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-launch:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms/step')"
done
