#!/bin/bash
out=gpurun_out/r05c
mkdir -p $out
python -m pytest tests -x -q -m gpu -s > $out/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee $out/summary.txt
grep -h "passed\|failed" $out/pytest_gpu.log | tail -3 | tee -a $out/summary.txt
grep -h "rank noise floor" $out/pytest_gpu.log | cut -c1-400 | tee -a $out/summary.txt
