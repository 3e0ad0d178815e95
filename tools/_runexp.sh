mkdir -p gpurun_out/exp9
timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/exp9/pytest.log 2>&1; tail -5 gpurun_out/exp9/pytest.log
timeout -k 10 120 python bench.py --steps 3 --warmup 1 > gpurun_out/exp9/bench.json 2>gpurun_out/exp9/bench.err; cat gpurun_out/exp9/bench.json
