#!/bin/bash
out=gpurun_out/r05d
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_chain or step_plan or full_val_split or superbatch or stream_schedules or small_batch_chain_beside or bf16x3" > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
LEGS="--fast_steps 0 --train_steps 0 --host_steps 0 --cpu_batches 0 --rank_check 1 --cached_steps 0"
python bench.py --steps 10 --warmup 3 $LEGS > $out/w1.json 2> $out/w1.err
python tools/ab_pass.py --help > /dev/null 2>&1
for w in 8 4 2; do
  python tools/rank_share.py --world $w --plan 1 --steps 12 --warmup 3 >> $out/share.jsonl 2>> $out/share.err
done
python - <<'PY' | tee -a gpurun_out/r05d/summary.txt
import json
d = json.loads([l for l in open('gpurun_out/r05d/w1.json') if l.startswith('{')][0])
print('w1 ms', round(d['ms_per_step'], 2), 'crc', d['ranks_crc32'], 'frac', round(d['roofline']['frac'], 4), 'launches', d['roofline']['launches'], 'rank_check', d.get('rank_check'))
for l in open('gpurun_out/r05d/share.jsonl'):
  r = json.loads(l); print('share world', r['world'], 'videos', r['videos'], 'ms', round(r['ms_per_pass'], 2))
PY
