#!/bin/bash
out=gpurun_out/r05j
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "bf16x6 or bf16x3" > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee $out/summary.txt
grep -h "passed\|failed\|max |error|" $out/pytest.log | tail -14 | tee -a $out/summary.txt
python bench.py --steps 5 --warmup 2 --fast_steps 5 --train_steps 0 --host_steps 0 --cpu_batches 0 --cached_steps 0 > $out/bench_fast.json 2> $out/bench_fast.err
python - <<'PY' | tee -a gpurun_out/r05j/summary.txt
import json
d = json.loads([l for l in open('gpurun_out/r05j/bench_fast.json') if l.startswith('{')][0])
print('exact ms', round(d['ms_per_step'], 2))
for k in ('fast_mode', 'fast_mode_bf16x6'):
  f = d.get(k, {})
  print(k, {a: f.get(a) for a in ('ms_per_step', 'max_abs_embedding_diff_vs_fp32', 'rank_rows_moved_vs_fp32_random_init', 'rank_rows_moved_on_correlated_embeddings', 'error')}, f.get('roofline', {}).get('achieved'))
PY
