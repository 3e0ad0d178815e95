#!/bin/bash
# tools/profile_cfg5.sh [OUTDIR] — the closest-hit instantiation of k_trace on BASELINE configs[4]'s scene (10 M triangles,
# 8192^2 atlas, falloff scoring, 16 of its 1024 spp per launch), on the GPU box:
#   its time (tools/time_trace.py cfg5), the rocprofv3 --kernel-trace --stats summary of that program, and one --pmc pass per
#   counter group (pmc_passes.sh; --kernel-trace only, each pass bounded by `timeout -k 10`), summed by tools/pmc_summary.py.
# Run from the repo root:  bash tools/profile_cfg5.sh gpurun_out/cfg5
set -u
R=$PWD
O=${1:-gpurun_out/cfg5}
mkdir -p $O
python3 tools/source_hash.py > $O/source_hash.json
timeout -k 10 400 python tools/time_trace.py --modes cfg5 --reps 2 --knob bvh_builder=1 - > $O/time_builder1.jsonl 2> $O/time.err; echo "time(builder 1) rc=$?"
timeout -k 10 400 python tools/time_trace.py --modes cfg5 --reps 2 --knob bvh_builder=2 - > $O/time_builder2.jsonl 2>> $O/time.err; echo "time(builder 2) rc=$?"
timeout -k 10 400 python tools/time_trace.py --modes cfg5 --reps 1 --knob bvh_builder=1 --knob trace_cut=1 - > $O/time_builder1_from_the_root.jsonl 2>> $O/time.err; echo "time(builder 1, cuts off) rc=$?"
# the kernel's own loop counters (tools/build_variants.py stats=-DAOB_TRACE_STATS=1 must have made the library HERE, before the snapshot travels)
[ -f tools/_variants/lib_stats.so ] && { timeout -k 10 400 python tools/time_trace.py --modes cfg5 --reps 1 --knob bvh_builder=1 tools/_variants/lib_stats.so > $O/loop_counters.jsonl 2>> $O/time.err; echo "loop counters rc=$?"; }
cat $O/time_builder1.jsonl $O/time_builder2.jsonl $O/time_builder1_from_the_root.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/tools/time_trace.py --child --modes cfg5 --reps 1 > $R/$O/stats_run.json 2> $R/$O/stats.err; echo "stats rc=$?"
cd $R
AOB_PMC_PROGRAM="tools/time_trace.py --child --modes cfg5 --reps 1" bash tools/pmc_passes.sh $O/pmc "FETCH_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "TA_BUSY_avr TA_BUSY_max" "TD_TD_BUSY_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > $O/pmc_passes.log 2>&1; echo "pmc rc=$?"
tail -30 $O/pmc/summary.json
python3 tools/roofline_closest.py $O $O/roofline.json
