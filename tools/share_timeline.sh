#!/bin/bash
# kernel timeline of one rank's share of the split under the whole split's step plan (tools/rank_share.py)
W=${1:-8}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/${2:-share}; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools/rank_share.py --world $W --steps 6 --warmup 3 > $D/share_w$W.json 2>/dev/null
cd $R
python tools/trace_timeline.py $D/stats/*/*kernel_trace.csv > $D/pass_timeline_share_w$W.txt 2>&1
rm -rf $D/stats
head -60 $D/pass_timeline_share_w$W.txt
