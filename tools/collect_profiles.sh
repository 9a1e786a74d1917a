#!/bin/bash
# rocprofv3 kernel stats + trace of the default bench pass, PMC fabric traffic (two separate passes, as
# MI355X_MICROARCH.md prescribes), SQ MFMA-busy counters, in-kernel clock of the tiled step.
#   bash tools/collect_profiles.sh <dir under gpurun_out> <round label>; copy the results to profiles/rNN_*
OUT=${1:-r06_final}
ROUND=${2:-6}
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT
mkdir -p $D
B="bench.py --steps 7 --warmup 2 --cpu_batches 0 --train_steps 0 --host_steps 0 --rank_check 0 --cached_steps 0 --api_steps 0 --fast_steps 0 --power_steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/$B > $D/bench_under_rocprof.json 2>/dev/null
B2="bench.py --steps 2 --warmup 1 --cpu_batches 0 --train_steps 0 --host_steps 0 --rank_check 0 --cached_steps 0 --api_steps 0 --fast_steps 0 --power_steps 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_FETCH_SIZE -- python3 $R/$B2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_WRITE_SIZE -- python3 $R/$B2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $D/pmc_SQ -- python3 $R/$B2 > $D/pmc_SQ.log 2>&1
cd $R
python tools/pmc_traffic.py $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE > $D/pmc_hbm_traffic.json
python tools/pmc_sq.py $D/pmc_SQ > $D/pmc_sq_summary.md 2>&1
rm -rf $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE $D/pmc_SQ
python tools/summarize_rocprof.py $D/stats/*/*kernel_stats.csv "round $ROUND, final kernels: rocprofv3 --kernel-trace --stats -- python3 $B (anet_icep_val, exact fp32, every pass rebuilds its schedules)" > $D/kernel_stats.md
python tools/trace_timeline.py $D/stats/*/*kernel_trace.csv > $D/pass_timeline.txt
rm -rf $D/stats
python tools/tile_trace.py 22419 2048 1024 2> $D/tile_trace.err | grep -v "amdgpu.ids" > $D/tile_trace.txt
head -30 $D/kernel_stats.md; tail -3 $D/bench_under_rocprof.json | cut -c1-400
