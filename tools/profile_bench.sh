#!/bin/bash
# tools/profile_bench.sh [OUTDIR] — everything DESIGN.md §6 and profiles/ quote, from one command on the GPU box:
#   the -m gpu suite (stops if it is not green), tools/run_shapes.py, `bench.py --steps 20 --warmup 5`, the rocprofv3
#   --kernel-trace --stats summary of the bench command, the FETCH_SIZE calibration kernels, one --pmc pass per counter
#   group over one bench step, and tools/pmc_record.py -> OUTDIR/k_trace_counters.json (copy it to profiles/).
# Counter passes use --kernel-trace only and each is bounded by `timeout -k 10`.  Run from the repo root:
#   bash tools/profile_bench.sh gpurun_out/profile
set -u
R=$PWD
O=${1:-gpurun_out/profile}
mkdir -p $O $O/calib
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/run_shapes.py > $O/shapes.jsonl 2> $O/shapes.err; echo "shapes rc=$?"
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cpp-host --no-wallclock > $R/$O/stats_bench.json 2> $R/$O/stats.err; echo "stats rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib/pass -- $R/tools/_fetch_calib > $R/$O/calib/run.log 2> $R/$O/calib/run.err; echo "calib rc=$?"
cd $R
bash tools/pmc_passes.sh $O/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" > $O/pmc_passes.log 2>&1; echo "pmc rc=$?"
python3 tools/pmc_record.py $O/pmc $O/calib $O/k_trace_counters.json > $O/record.log 2>&1; echo "record rc=$?"
grep -E "valu_|hbm_side_bytes_per_launch\"|L2_hit" $O/record.log
