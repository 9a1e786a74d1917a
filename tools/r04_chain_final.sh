#!/bin/bash
# round 4, after the step chain: a rank's share at 615 / 1230 / 2460 videos and the full split (12 passes each), the
# 8-rank functional run sharing one GPU over gloo (CRC against the single process, default kernels and --pin_shapes 1)
OUT=${1:-r04_chain_final}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=$R/gpurun_out/$OUT; mkdir -p $D
cd $R
Q="--cpu_batches 0 --host_steps 0 --cached_steps 0 --rank_check 0 --train_steps 0"
for n in 615 1230 2460 4917; do python bench.py --n_videos $n --steps 12 --warmup 3 $Q > $D/share_$n.json 2>/dev/null; done
timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 $Q > $D/w8_shared_gpu.json 2> $D/w8.err
python bench.py --steps 3 --warmup 2 $Q > $D/w1_ref.json 2>/dev/null
timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --pin_shapes 1 $Q > $D/w8_shared_gpu_pinned.json 2> $D/w8p.err
python bench.py --steps 3 --warmup 2 --pin_shapes 1 $Q > $D/w1_ref_pinned.json 2>/dev/null
python - <<'PY' "$D"
import json, sys, os
D = sys.argv[1]
def last(p):
  try:
    return json.loads(open(os.path.join(D, p)).read().strip().split('\n')[-1])
  except Exception as e:
    return {'error': str(e)}
full = None
for n in (4917, 2460, 1230, 615):
  d = last('share_%d.json' % n)
  if n == 4917: full = d['ms_per_step']
  print('%5d videos: %8.2f ms / pass  (min %.2f median %.2f max %.2f)  roofline.frac %.3f  kernel share %.2f  implied efficiency %.3f'
        % (n, d['ms_per_step'], d['pass_ms']['min'], d['pass_ms']['median'], d['pass_ms']['max'], d['roofline']['frac'],
           d['roofline']['kernel_time_share'], full / (4917.0 / n * d['ms_per_step']) if n != 4917 else 1.0))
for a, b in (('w8_shared_gpu.json', 'w1_ref.json'), ('w8_shared_gpu_pinned.json', 'w1_ref_pinned.json')):
  x, y = last(a), last(b)
  print(a, 'ranks_crc32', x.get('ranks_crc32'), 'n_gpus', x.get('n_gpus'), '|', b, y.get('ranks_crc32'),
        '| reports equal:', x.get('report_i2t_random_init') == y.get('report_i2t_random_init'), x.get('error', ''), y.get('error', ''))
  if 'per_rank' in x: print('   per-rank videos', [r.get('videos') for r in x['per_rank']])
PY
tail -n 2 $D/w8.err; tail -n 2 $D/w8p.err
