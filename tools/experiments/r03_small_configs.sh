NOX="--no-cpu-baseline --no-query-extra --no-loglinear-extra --no-c4-extra --no-live-pmc --no-seed-extra"
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,2), 'M pairs/s', round(1000*d['ms_per_step'],1), 'us/step')"; }
python bench.py --batch 4096 --entities 32768 --dim 300 --entity-dim 128 --steps 300 --warmup 30 $NOX 2>/dev/null | p product_search
python bench.py --model loglinear --batch 1024 --dim 300 --entities 715 --window 8 --steps 300 --warmup 30 $NOX 2>/dev/null | p w3c
python tools/bench_c1.py 2>/dev/null | tail -3
python tools/bench_c4.py --kinds loglinear --steps 10 2>/dev/null | grep -E "ms_per_step|pairs_per_s" | head -2
