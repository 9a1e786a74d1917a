#!/bin/bash
out=gpurun_out/r05k
mkdir -p $out
LEGS="--fast_steps 0 --train_steps 0 --host_steps 3 --cpu_batches 0 --rank_check 1 --cached_steps 0"
python bench.py --workload anet_c3d_val --steps 8 --warmup 3 $LEGS > $out/c3d.json 2> $out/c3d.err
python bench.py --workload didemo_icep_val --steps 10 --warmup 3 $LEGS > $out/didemo.json 2> $out/didemo.err
python bench.py --workload anet_icep_val --rnn_type maxout --steps 6 --warmup 2 $LEGS > $out/icep_maxout.json 2> $out/icep_maxout.err
python bench.py --workload plumbing --embed 256 --steps 5 --warmup 2 $LEGS > $out/plumbing.json 2> $out/plumbing.err
python - <<'PY'
import json
for n in ('c3d', 'didemo', 'icep_maxout', 'plumbing'):
  try:
    d = json.loads([l for l in open('gpurun_out/r05k/%s.json' % n) if l.startswith('{')][0])
    print(n, 'ms', round(d['ms_per_step'], 2), 'M pairs/s', round(d['value'] / 1e6, 2), 'frac', round(d['roofline']['frac'], 3), 'rank_check', d.get('rank_check', {}).get('mismatches'), 'pcie', round(d.get('pcie_inclusive', {}).get('ms_per_step', 0), 1))
  except Exception as e:
    print(n, 'FAILED', e, open('gpurun_out/r05k/%s.err' % n).read()[-600:])
PY
