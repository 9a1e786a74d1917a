#!/usr/bin/env python3
"""Host side of one validation pass (encode_data_device over the full split): when is the first
level-1 launch queued, how long does the host spend inside each call, and where (cProfile)."""
import cProfile
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from cmhse_amd import ops, synthetic  # noqa: E402
from cmhse_amd.evaluation import encode_data_device  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402


def main():
  n_videos = int(sys.argv[sys.argv.index('--n_videos') + 1]) if '--n_videos' in sys.argv else 0
  plan = {} if '--plan' in sys.argv else None
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  wl = dict(bench.WORKLOADS['anet_icep_val'])
  opt = bench.make_opt(wl, 'attention', 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(n_videos or wl['n_videos'], seed=0, dataset='anet')
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
  from cmhse_amd import evaluation
  evaluation.ops.gru_pool_fwd_multi = wrapped
  for i in range(5):
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    cat, _, _, fin = encode_data_device(opt, model, batches, logging=quiet, defer_logging=True, plan=plan)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fin()
    print('pass %d: entry -> level-1 call %.2f ms | level-1 call (host) %.2f | -> level-2 call %.2f | '
          'encode returned %.2f | GPU done %.2f' % (i, (marks[0] - t0) * 1e3, (marks[1] - marks[0]) * 1e3,
                                                   (marks[2] - marks[1]) * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
  pr = cProfile.Profile()
  torch.cuda.synchronize()
  pr.enable()
  cat, _, _, fin = encode_data_device(opt, model, batches, logging=quiet, defer_logging=True, plan=plan)
  pr.disable()
  torch.cuda.synchronize()
  fin()
  pstats.Stats(pr).sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
  main()
