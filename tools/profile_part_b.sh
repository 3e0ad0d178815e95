#!/bin/bash
# tools/profile_part_b.sh OUTDIR — the second half of tools/profile_bench.sh: the FETCH_SIZE calibration kernels and one --pmc pass per counter group over one
# bench step (counter passes with --kernel-trace only).  tools/pmc_record.py then runs wherever both halves' outputs are (it needs no GPU):
#   python3 tools/pmc_record.py OUTDIR/pmc OUTDIR/calib OUTDIR/roofs/slopes.jsonl OUTDIR/k_trace_counters.json OUTDIR/roofs/valu_mix.json
set -u
R=$PWD
O=${1:-gpurun_out/profile}
mkdir -p $O $O/calib
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib/pass -- $R/tools/_fetch_calib > $R/$O/calib/run.log 2> $R/$O/calib/run.err; echo "calib rc=$?"
cd $R
bash tools/pmc_passes.sh $O/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TA_BUSY_avr TA_BUSY_max" "TA_TOTAL_WAVEFRONTS_sum" "TD_TD_BUSY_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" > $O/pmc_passes.log 2>&1; echo "pmc rc=$?"
tail -3 $O/pmc_passes.log
