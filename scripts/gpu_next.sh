# next-row operators: parity tests + their timings
timeout 900 python -m pytest tests -m gpu -q -k "smooth or decoder_tail or post_process or flip_right or fixture" 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 --no_cpu_baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']); print(json.dumps(d['next_rows']))"
