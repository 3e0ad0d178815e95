mkdir -p gpurun_out/exp21
for r in 20 24 28 32 40; do for l in 8 10 12 14; do
  AOB_TRACE_REFILL_IDLE=$r AOB_TRACE_LEAF_WAIT=$l timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp21/r${r}_l$l.json 2>gpurun_out/exp21/r${r}_l$l.err || exit 1
  echo refill $r leaf $l $(grep -o '"trace_kernel": [0-9.]*' gpurun_out/exp21/r${r}_l$l.json)
done; done
