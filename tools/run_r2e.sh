#!/bin/bash
# GPU call: full GPU suite on the current build, bench, kernel stats, PMC passes + counters record
set -u
R=$PWD
O=gpurun_out/r2h
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
