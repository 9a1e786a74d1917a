#!/bin/bash
OUT=${1:-r04_chain2}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd $R
M="tune.tall_tile_min_wgs=2048;tune.tall_tile_min_wgs=1024;tune.tall_tile_min_wgs=256;tune.tall_tile_min_wgs=0"
for nv in 615 1230 2460 0; do
  echo "== n_videos $nv" >> $D/ab.txt
  timeout 900 python tools/ab_pass.py --modes "$M" --rounds 3 --passes 3 --n_videos $nv >> $D/ab.txt 2>&1
done
grep -v amdgpu.ids $D/ab.txt
