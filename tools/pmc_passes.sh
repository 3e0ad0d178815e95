#!/bin/bash
# tools/pmc_passes.sh OUTDIR "GROUP1" "GROUP2" ... — one rocprofv3 counter pass per group over one bench step.
#
# Each GROUP is a space-separated list of PMC counters small enough for one pass (too many in one pass abort
# with "Request exceeds the capabilities of the hardware").  Counter passes use --kernel-trace only (never a
# hip/hsa/sys trace) and each is bounded by `timeout -k 10 150`.  tools/pmc_summary.py sums the counters of the
# k_trace launches.  AOB_PMC_PROGRAM: the python program (and its arguments) measured instead of one bench step — it goes
# straight after `-- python3` (no shell, no env in between).  Run from the repo root on the GPU box, e.g.
#   bash tools/pmc_passes.sh gpurun_out/pmc "TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
#   AOB_PMC_PROGRAM="tools/time_trace.py --child --modes cfg5 --reps 1" bash tools/pmc_passes.sh gpurun_out/pmc5 "TA_BUSY_avr"
set -u
R=$PWD
O=$R/$1; shift
mkdir -p "$O"
PROGRAM=${AOB_PMC_PROGRAM:-bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cpp-host --no-wallclock}
cd /tmp && export TMPDIR=/tmp
n=0
for c in "$@"; do
    n=$((n + 1))
    timeout -k 10 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pass$n" -- python3 "$R"/$PROGRAM > "$O/pass$n.log" 2>&1 || { echo "pass $n ($c) failed"; exit 1; }
    echo "pass $n done: $c"
done
python3 "$R/tools/pmc_summary.py" "$O" > "$O/summary.json" && cat "$O/summary.json"
