#!/bin/bash
# rocprofv3 kernel trace of K training steps -> per-kernel table of the timed region
# (tools/trace_busy.py).  usage: [FEED=resident|pull|upload|prefetch] bash tools/profile_train.sh <out dir under gpurun_out> [config ...]
OUT=${1:-train_prof}; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for c in ${@:-c3d}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/train_$c -- python3 $R/tools/train_profile.py --config $c --steps 10 --feed ${FEED:-resident} > $D/train_$c.txt 2>/dev/null
  (cd $R && python tools/trace_busy.py $D/train_$c/*/*kernel_trace.csv 10 --timeline > $D/train_step_$c.md)
  rm -rf $D/train_$c
done
