timeout -k 10 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -2
for i in 1 2; do timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"trace_kernel": [0-9.]*'; done
