#!/bin/bash
OUT=${1:-r04_chain3}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd $R
M="tune.tiny_max_seqs=1024,tune.mid_max_seqs=1024;tune.tiny_max_seqs=512,tune.mid_max_seqs=512;tune.tiny_max_seqs=256,tune.mid_max_seqs=256;tune.tiny_max_seqs=128,tune.mid_max_seqs=128;tune.tiny_max_seqs=64,tune.mid_max_seqs=64"
for nv in 615 2460 0; do
  echo "== n_videos $nv" >> $D/ab.txt
  timeout 900 python tools/ab_pass.py --modes "$M" --rounds 3 --passes 3 --n_videos $nv >> $D/ab.txt 2>&1
done
grep -v amdgpu.ids $D/ab.txt
