#!/bin/bash
# step-chain launch (gru_step_chain_kernel) against per-step launches: the validation pass at the full split and at
# a rank's share of it, arms interleaved; then the kernel table of a chained pass.
OUT=${1:-r04_chain}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd $R
M="tune.chain_min_steps=0;tune.chain_min_steps=2"
for nv in 0 615 1230 2460; do
  echo "== n_videos $nv" >> $D/ab.txt
  timeout 900 python tools/ab_pass.py --modes "$M" --rounds 4 --passes 3 --n_videos $nv >> $D/ab.txt 2>&1
done
grep -v amdgpu.ids $D/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/bench.py --steps 5 --warmup 2 --cpu_batches 0 --train_steps 0 --host_steps 0 --rank_check 0 --cached_steps 0 > $D/bench_under_rocprof.json 2>/dev/null
cd $R
python tools/summarize_rocprof.py $D/stats/*/*kernel_stats.csv "chained pass" | head -14 > $D/kernel_stats.md
python tools/trace_timeline.py $D/stats/*/*kernel_trace.csv > $D/pass_timeline.txt 2>&1
rm -rf $D/stats
cat $D/kernel_stats.md
