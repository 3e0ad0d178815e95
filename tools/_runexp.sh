mkdir -p gpurun_out/exp24
timeout -k 10 600 python -m pytest tests -q -x -m gpu > gpurun_out/exp24/pytest.log 2>&1; tail -3 gpurun_out/exp24/pytest.log
for i in 1 2; do
timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/exp24/b$i.json 2>gpurun_out/exp24/b$i.err || exit 1
echo $(grep -o '"trace_kernel": [0-9.]*' gpurun_out/exp24/b$i.json)
done
