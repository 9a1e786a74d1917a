#!/bin/bash
out=gpurun_out/r05d
mkdir -p $out
M="tune.chain_attention=0;tune.chain_attention=0,two_streams=1;tune.chain_attention=1,two_streams=1"
echo "== full split" | tee $out/ab2.txt
python tools/ab_pass.py --modes "$M" --rounds 3 --passes 3 2>&1 | grep -v amdgpu | tee -a $out/ab2.txt
for nv in 615 2460; do
  echo "== n_videos $nv" | tee -a $out/ab2.txt
  python tools/ab_pass.py --modes "$M" --rounds 4 --passes 4 --n_videos $nv 2>&1 | grep -v amdgpu | tee -a $out/ab2.txt
done
