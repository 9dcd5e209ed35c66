timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', 'kernel_ms', d['roofline']['kernel_ms'], 'Gpx/s', d['value']/1e3)"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload C1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1', 'kernel_ms', d['roofline']['kernel_ms'], 'Gpx/s', d['value']/1e3)"
bash scratch/pmc3.sh 2>&1 | grep -E "^(a0|e0)"
