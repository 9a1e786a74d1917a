#!/bin/bash
out=gpurun_out/r05h
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_plan or tuning_contexts or stream_schedules or step_chain_failure or full_val_split" > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
M="tune.early_xproj=0;tune.early_xproj=1"
python tools/ab_pass.py --modes "$M" --rounds 4 --passes 3 2>&1 | grep -v amdgpu | tee -a $out/summary.txt
for nv in 615 1230 2460; do
  python tools/ab_pass.py --modes "$M" --rounds 4 --passes 4 --n_videos $nv 2>&1 | grep -v amdgpu | tee -a $out/summary.txt
done
