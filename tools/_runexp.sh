mkdir -p gpurun_out/exp5
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/exp5/pytest.log 2>&1; tail -3 gpurun_out/exp5/pytest.log
for r in 4 8 12 16 24; do for l in 8 16; do
  AOB_TRACE_REFILL_IDLE=$r AOB_TRACE_LEAF_WAIT=$l timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp5/r${r}_l$l.json 2>gpurun_out/exp5/r${r}_l$l.err || exit 1
  echo refill $r leaf $l $(grep -o '"trace_kernel": [0-9.]*' gpurun_out/exp5/r${r}_l$l.json)
done; done
