#!/bin/bash
# tools/profile_part_a.sh OUTDIR — the first half of tools/profile_bench.sh (one gpurun call holds 20 minutes: the whole script does not fit since round 6):
# shapes, the bench line, the slope builds, the loop counters, the issue probe, the rocprofv3 kernel stats of the bench command.
set -u
R=$PWD
O=${1:-gpurun_out/profile}
mkdir -p $O $O/roofs
python3 tools/source_hash.py > $O/source_hash.json
timeout -k 10 400 python tools/run_shapes.py > $O/shapes.jsonl 2> $O/shapes.err; echo "shapes rc=$?"
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
timeout -k 10 420 python tools/time_trace.py --modes anyhit --reps 3 --knob bvh_builder=2 tools/_variants/lib_base.so tools/_variants/lib_valu8.so tools/_variants/lib_valu16.so tools/_variants/lib_valu32.so tools/_variants/lib_slow4.so tools/_variants/lib_slow8.so tools/_variants/lib_slow16.so tools/_variants/lib_load1.so tools/_variants/lib_load2.so > $O/roofs/slopes.jsonl 2> $O/roofs/slopes.err; echo "slopes rc=$?"
timeout -k 10 200 python tools/time_trace.py --modes anyhit --reps 1 --knob bvh_builder=2 tools/_variants/lib_stats.so > $O/roofs/loop_counters.jsonl 2> $O/roofs/loop_counters.err; echo "loop counters rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cpp-host --no-wallclock > $R/$O/stats_bench.json 2> $R/$O/stats.err; echo "stats rc=$?"
cd $R
cp tools/_variants/valu_mix.json $O/roofs/valu_mix.json 2>/dev/null
