#!/bin/bash
# one GPU call: knob sweep, kernel stats of the bench command, PMC passes, FETCH_SIZE calibration, counters record
set -u
R=$PWD
O=gpurun_out/r2c
mkdir -p $O $O/calib
timeout -k 10 300 python3 tools/sweep_trace_knobs.py > $O/knobs.jsonl 2> $O/knobs.err; echo "sweep rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cpp-host --no-wallclock > $R/$O/stats_bench.json 2> $R/$O/stats.err; echo "stats rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib/pass -- $R/tools/_fetch_calib > $R/$O/calib/run.log 2> $R/$O/calib/run.err; echo "calib rc=$?"
cd $R
bash tools/pmc_passes.sh $O/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" > $O/pmc_passes.log 2>&1; echo "pmc rc=$?"
python3 tools/pmc_record.py $O/pmc $O/calib $O/k_trace_counters.json > $O/record.log 2>&1; echo "record rc=$?"
tail -5 $O/record.log
cat $O/knobs.jsonl
