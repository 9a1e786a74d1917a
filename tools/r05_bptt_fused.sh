#!/bin/bash
# Round 5, item 2: the BPTT step as ONE launch (split-K + last-arriver epilogue) against two.
set -u
out=gpurun_out/r05b
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "one_launch_with_a_last_arriver or resident_chain_kernel or weight_gradients_in_time" > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee $out/summary.txt
grep -h "bit-identical" $out/pytest.log | tee -a $out/summary.txt
A="tune.bwd_fused_step=0,side_streams=0;tune.bwd_fused_step=1,side_streams=0"
for shape in "152 80 2048" "152 80 500" "100 80 2048" "170 80 2048" "256 80 2048" "64 80 2048"; do
  set -- $shape
  echo "== standalone S=$1 T=$2 I=$3" | tee -a $out/summary.txt
  python tools/bench_bptt.py --S $1 --T $2 --I $3 --arms "$A" --rounds 6 2>&1 | tee -a $out/summary.txt
done
B="tune.bwd_fused_step=0;tune.bwd_fused_step=1"
for cfg in c3d icep icep_recon didemo_recon; do
  echo "== train step $cfg" | tee -a $out/summary.txt
  python tools/ab_train.py --config $cfg --modes "$B" --rounds 6 --steps 10 2>&1 | tail -4 | tee -a $out/summary.txt
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof$f -- python3 $R/tools/bench_bptt.py --S 152 --T 80 --I 2048 --arms "tune.bwd_fused_step=$f,side_streams=0" --rounds 6 > /dev/null 2>&1
  python3 $R/tools/summarize_rocprof.py $R/$out/prof$f/*/*kernel_stats.csv 2>/dev/null | head -12 > $R/$out/kernel_stats_fused$f.md
  rm -rf $R/$out/prof$f
done
cat $R/$out/kernel_stats_fused0.md $R/$out/kernel_stats_fused1.md
