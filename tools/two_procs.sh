python bench.py --no-cpu-baseline --steps 6 --warmup 2 --inflight 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A',d['value'],d['ms_per_step'])" &
python bench.py --no-cpu-baseline --steps 6 --warmup 2 --inflight 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B',d['value'],d['ms_per_step'])" &
wait
