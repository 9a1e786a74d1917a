#!/bin/bash
# The few-sequence tail of an inference chain as one resident kernel (infer_tail_min_steps): a rank's share and the whole split, arms alternated.
O=gpurun_out/r05n; mkdir -p $O
for rep in 1 2; do
  for v in 0 16; do
    for w in 8 4; do
      python tools/rank_share.py --world $w --steps 12 --tune infer_tail_min_steps=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('share W=%d infer_tail_min_steps=$v: %.2f ms (min %.2f median %.2f)' % (d['world'], d['ms_per_pass'], d['pass_ms_min'], d['pass_ms_median']))"
    done
  done
done
for v in 0 16 0 16; do
  python tools/rank_share.py --world 1 --steps 8 --tune infer_tail_min_steps=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split (W=1) infer_tail_min_steps=$v: %.2f ms (min %.2f median %.2f)' % (d['ms_per_pass'], d['pass_ms_min'], d['pass_ms_median']))"
done
