mkdir -p gpurun_out/exp32
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/exp32/pytest.log 2>&1; tail -3 gpurun_out/exp32/pytest.log
for i in 1 2; do
timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/exp32/b$i.json 2>gpurun_out/exp32/b$i.err || exit 1
echo $(grep -o '"trace_kernel": [0-9.]*' gpurun_out/exp32/b$i.json) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/exp32/b$i.json) $(grep -o '"bvh_build": [0-9.]*' gpurun_out/exp32/b$i.json) $(grep -o '"inner_visits_per_ray": [0-9.]*' gpurun_out/exp32/b$i.json)
done
