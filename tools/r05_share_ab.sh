#!/bin/bash
out=gpurun_out/r05h
mkdir -p $out; rm -f $out/share_ab.jsonl
for rep in 1 2; do
  for w in 8 4 2; do
    for t in early_xproj=0 early_xproj=1; do
      python tools/rank_share.py --world $w --steps 12 --warmup 3 --tune $t >> $out/share_ab.jsonl 2>/dev/null
    done
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r05h/share_ab.jsonl'):
  r = json.loads(l); print('world', r['world'], 'videos', r['videos'], r['tune'], 'ms %.2f (min %.2f)' % (r['ms_per_pass'], r['pass_ms_min']))
PY
