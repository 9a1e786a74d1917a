#!/usr/bin/env python3
"""Per-pass wall time of the validation pass against the time the host spends inside the level-1
call (cmhse_gru_pool_fwd_multi): does a pass whose launches block the host run slower?

  python tools/pass_jitter.py [--passes 16] [--plan] [--no_freeze]   (default: gc.freeze() after pass 2, as bench.py did until late round 4)
"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from cmhse_amd import evaluation, ops  # noqa: E402
from cmhse_amd import synthetic  # noqa: E402
from cmhse_amd.evaluation import encode_data_device  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402


def main():
  passes = int(sys.argv[sys.argv.index('--passes') + 1]) if '--passes' in sys.argv else 16
  plan = {} if '--plan' in sys.argv else None
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  wl = dict(bench.WORKLOADS['anet_icep_val'])
  opt = bench.make_opt(wl, 'attention', 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset='anet')
  nb = (spec.n_videos + 31) // 32
  batches = bench.build_loader(spec, wl, dev, 0, nb)
  quiet = lambda *a, **k: None
  orig = ops.gru_pool_fwd_multi
  marks = []

  def wrapped(*a, **k):
    marks.append(time.perf_counter())
    r = orig(*a, **k)
    marks.append(time.perf_counter())
    return r
  ops.gru_pool_fwd_multi = wrapped
  evaluation.ops.gru_pool_fwd_multi = wrapped
  import collections
  import gc
  import threading
  import traceback
  main_id = threading.get_ident()
  samples, stop = [], [False]

  def sampler():
    while not stop[0]:
      fr = sys._current_frames().get(main_id)
      if fr is not None:
        samples.append((time.perf_counter(), ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno)
                                                          for f in reversed(traceback.extract_stack(fr)[-4:]))))
      time.sleep(0.002)
  th = threading.Thread(target=sampler, daemon=True)
  th.start()
  log = []
  for i in range(passes):
    torch.cuda.synchronize()
    if i == 3 and '--no_freeze' not in sys.argv:
      gc.collect()
      gc.freeze()
    marks.clear()
    del samples[:]
    t0 = time.perf_counter()
    cat, _, _, fin = encode_data_device(opt, model, batches, logging=quiet, defer_logging=True, plan=plan)
    t1 = time.perf_counter()
    r_i, t_i = ops.sim_rank(cat['vid_emb'], cat['para_emb'])
    r_t, t_t = ops.sim_rank(cat['para_emb'], cat['vid_emb'])
    fin()
    torch.stack([r_i, t_i, r_t, t_t]).cpu()
    t2 = time.perf_counter()
    print('pass %2d: %.2f ms | host: -> level-1 call %.2f, inside it %.2f, encode returned %.2f'
          % (i, (t2 - t0) * 1e3, (marks[0] - t0) * 1e3, (marks[1] - marks[0]) * 1e3, (t1 - t0) * 1e3))
    if (t1 - t0) > 0.030 and i > 0:
      # where was the host while it queued the encoders?  (stack samples every 2 ms, innermost first)
      hist = collections.Counter(s_ for ts, s_ in samples if ts <= t1)
      for s_, n in hist.most_common(6):
        print('      %3d samples  %s' % (n, s_))
  stop[0] = True


if __name__ == '__main__':
  main()
