#!/bin/bash
# Round 5, item 1: partition-independent ranks by default.  Tests, then the 1- and 8-rank bench at
# N = 4917 with default flags (CRC must agree), then a rank's share under the whole split's plan.
set -u
out=gpurun_out/r05a
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_plan or step_chain_launch or superbatch or sharded" > $out/pytest_plan.log 2>&1
echo "pytest plan rc=$?" | tee -a $out/summary.txt
python -m pytest tests/test_boundary.py -x -q -m gpu -k "bench_starts" > $out/pytest_boundary.log 2>&1
echo "pytest boundary rc=$?" | tee -a $out/summary.txt
LEGS="--fast_steps 0 --train_steps 0 --host_steps 0 --cpu_batches 0 --rank_check 0 --cached_steps 0"
python bench.py --gpus 1 --steps 10 --warmup 3 $LEGS > $out/w1.json 2> $out/w1.err
python bench.py --gpus 8 --steps 3 --warmup 1 $LEGS > $out/w8.json 2> $out/w8.err
python bench.py --gpus 2 --steps 3 --warmup 1 $LEGS > $out/w2.json 2> $out/w2.err
python - <<'PY' | tee -a gpurun_out/r05a/summary.txt
import json
for n in ('w1', 'w8', 'w2'):
  try:
    d = json.loads([l for l in open('gpurun_out/r05a/%s.json' % n) if l.startswith('{')][0])
    print(n, 'ms', round(d['ms_per_step'], 2), 'crc', d['ranks_crc32'], 'frac', round(d['roofline']['frac'], 4),
          [round(r.get('encode_ms', 0), 1) for r in d['per_rank']])
  except Exception as e:
    print(n, 'FAILED', e)
PY
for w in 8 4 2; do
  for plan in 1 0; do
    python tools/rank_share.py --world $w --plan $plan --steps 12 --warmup 3 >> $out/share.jsonl 2>> $out/share.err
  done
done
cat $out/share.jsonl | tee -a $out/summary.txt
