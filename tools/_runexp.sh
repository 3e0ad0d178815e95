mkdir -p gpurun_out/exp17
timeout -k 10 600 python -m pytest tests -q -x -m gpu > gpurun_out/exp17/pytest.log 2>&1; tail -3 gpurun_out/exp17/pytest.log
for parts in 1 2 4 8; do
  AOB_TRACE_PARTS=$parts timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp17/p${parts}.json 2>gpurun_out/exp17/p${parts}.err || exit 1
  echo parts $parts $(grep -o '"trace_kernel": [0-9.]*' gpurun_out/exp17/p${parts}.json)
done
