#!/bin/bash
# The mid-size step's shape space on the GPU box: ring depth (compile time) x waves per workgroup
# (CMHSE_MID_WAVES) x unit tile (CMHSE_MID_UNITS), us per dependent step.  Rebuilds the library
# three times and restores the default build at the end.
#   bash tools/mid_shape_sweep.sh > gpurun_out/mid_shape_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for ring in 2 4 8; do
  CMHSE_HIPCC_FLAGS="-DCMHSE_MID_RING=$ring" python -m cmhse_amd.build --force > /dev/null
  echo "== 16-k blocks in flight per wave: $ring"
  python tools/step_sweep.py --sizes 8,16,32,64,128,512,1024 --dims 300 --T 24 \
    --arms "CMHSE_MID_WAVES=8,CMHSE_MID_UNITS=16;CMHSE_MID_WAVES=4,CMHSE_MID_UNITS=16;CMHSE_MID_WAVES=8;CMHSE_MID_WAVES=4" 2>&1 | grep -v amdgpu
done
python -m cmhse_amd.build --force > /dev/null
echo "== default build (ring 2): old small-batch kernel | 16-unit tiles | auto"
python tools/step_sweep.py --sizes 1,8,16,32,64,152,320,512,1024 --dims 500,300,1024 \
  --arms "CMHSE_MID_MAX_SEQS=0;CMHSE_MID_UNITS=16;CMHSE_X=1" 2>&1 | grep -v amdgpu
