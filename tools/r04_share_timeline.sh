#!/bin/bash
# kernel timeline of a rank's share of the split (615 videos)
OUT=${1:-r04_share_tl}
NV=${2:-615}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/bench.py --n_videos $NV --steps 6 --warmup 3 --cpu_batches 0 --train_steps 0 --host_steps 0 --rank_check 0 --cached_steps 0 > $D/bench_$NV.json 2>/dev/null
cd $R
python tools/trace_timeline.py $D/stats/*/*kernel_trace.csv > $D/pass_timeline_$NV.txt 2>&1
rm -rf $D/stats
head -70 $D/pass_timeline_$NV.txt
