#!/bin/bash
OUT=${1:-r04_tn_ab}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd $R
M="tune.tn_rows_bm=128;tune.tn_rows_bm=192;tune.tn_rows_bm=192,tune.bwd_chunk_rows=3072;tune.tn_rows_bm=128,tune.bwd_chunk_rows=3072"
for c in icep c3d icep_recon didemo_recon; do
  echo "== $c" >> $D/ab.txt
  timeout 600 python tools/ab_train.py --config $c --rounds 5 --steps 10 --modes "$M" >> $D/ab.txt 2>&1
done
cat $D/ab.txt
