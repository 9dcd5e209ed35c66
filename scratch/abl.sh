timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -3
for w in C2 C1 C4; do python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'Gpx/s', round(d['value']/1e3,1), 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"; done
bash scratch/pmc3.sh 2>&1 | grep -E "^(a0|b0|c0|d0)"
