#!/bin/bash
# tools/profile_bench.sh [OUTDIR] — everything DESIGN.md §6 and profiles/ quote, from one command on the GPU box:
#   the -m gpu suite (stops if it is not green), tools/run_shapes.py, `bench.py --steps 20 --warmup 5`, the rocprofv3
#   --kernel-trace --stats summary of the bench command, the FETCH_SIZE calibration kernels, one --pmc pass per counter
#   group over one bench step, the slope builds of k_trace (tools/build_variants.py must have made tools/_variants/lib_{base,
#   valu8,valu16,valu32,slow4,slow8,slow16,load1,load2,stats}.so and tools/_variants/valu_mix.json (tools/valu_mix.py) HERE, before the snapshot travels) and tools/pmc_record.py ->
#   OUTDIR/k_trace_counters.json (copy it to profiles/).
# Counter passes use --kernel-trace only and each is bounded by `timeout -k 10`.  Run from the repo root:
#   bash tools/profile_bench.sh gpurun_out/profile
set -u
R=$PWD
O=${1:-gpurun_out/profile}
mkdir -p $O $O/calib $O/roofs
python3 tools/source_hash.py > $O/source_hash.json
if [ -z "${AOB_PROFILE_SKIP_TESTS:-}" ]; then       # (set it when the suite has just run green on this very tree)
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
fi
timeout -k 10 400 python tools/run_shapes.py > $O/shapes.jsonl 2> $O/shapes.err; echo "shapes rc=$?"
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
timeout -k 10 600 python tools/time_trace.py --modes anyhit --reps 3 --knob bvh_builder=2 tools/_variants/lib_base.so tools/_variants/lib_valu8.so tools/_variants/lib_valu16.so tools/_variants/lib_valu32.so tools/_variants/lib_slow4.so tools/_variants/lib_slow8.so tools/_variants/lib_slow16.so tools/_variants/lib_load1.so tools/_variants/lib_load2.so > $O/roofs/slopes.jsonl 2> $O/roofs/slopes.err; echo "slopes rc=$?"
timeout -k 10 300 python tools/time_trace.py --modes anyhit --reps 1 --knob bvh_builder=2 tools/_variants/lib_stats.so > $O/roofs/loop_counters.jsonl 2> $O/roofs/loop_counters.err; echo "loop counters rc=$?"
timeout -k 10 120 tools/_valu_probe > $O/roofs/valu_probe.jsonl 2> $O/roofs/valu_probe.err; echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cpp-host --no-wallclock > $R/$O/stats_bench.json 2> $R/$O/stats.err; echo "stats rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib/pass -- $R/tools/_fetch_calib > $R/$O/calib/run.log 2> $R/$O/calib/run.err; echo "calib rc=$?"
cd $R
bash tools/pmc_passes.sh $O/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TA_BUSY_avr TA_BUSY_max" "TA_TOTAL_WAVEFRONTS_sum" "TD_TD_BUSY_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" > $O/pmc_passes.log 2>&1; echo "pmc rc=$?"
cp tools/_variants/valu_mix.json $O/roofs/valu_mix.json 2>/dev/null
python3 tools/pmc_record.py $O/pmc $O/calib $O/roofs/slopes.jsonl $O/k_trace_counters.json $O/roofs/valu_mix.json > $O/record.log 2>&1; echo "record rc=$?"
grep -E "\"bound\"|\"frac\"|fabric_side_bytes_per_launch\"|L2_hit|\"mean\"" $O/record.log
