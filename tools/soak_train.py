#!/usr/bin/env python3
"""Soak: epochs of train_emb steps interleaved with validation passes (the shape of train.py's
loop), checking that nothing hangs, traps or drifts: loss meters finite, the validation report
identical between two passes over the same weights, allocator footprint flat.
  python tools/soak_train.py --config icep_recon --epochs 3 --steps 300"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import bench  # noqa: E402
from cmhse_amd import synthetic  # noqa: E402
from cmhse_amd.evaluation import LogCollector, encode_data, i2t, t2i  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402
from train_profile import CONFIGS  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--config', default='icep_recon', choices=sorted(CONFIGS))
  ap.add_argument('--epochs', type=int, default=3)
  ap.add_argument('--steps', type=int, default=300)
  ap.add_argument('--val_videos', type=int, default=256)
  ap.add_argument('--pin', type=int, default=1, help='1: training batches in pinned host memory, as a '
                  'DataLoader(pin_memory=True) hands them over (the host-fed path of train_emb); 0: pageable')
  args = ap.parse_args()
  cfg = dict(CONFIGS[args.config])
  wl = dict(bench.WORKLOADS[cfg.pop('workload')])
  opt = bench.make_opt(wl, 'attention', 1024)
  for k, v in cfg.items():
    setattr(opt, k, v)
  torch.cuda.set_device(0)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(32 * 8, seed=0, dataset=wl['dataset'])
  train = synthetic.make_batches(spec, 32, wl['img_dim'], wl['vocab'], seed=0, feat=wl['feat'])
  if args.pin:
    train = [tuple(t.pin_memory() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b))
             for b in train]
  vspec = synthetic.anet_like_spec(args.val_videos, seed=5, dataset=wl['dataset'])
  val = synthetic.make_batches(vspec, 32, wl['img_dim'], wl['vocab'], seed=5, feat=wl['feat'])
  quiet = lambda *a, **k: None

  class Sink(object):
    def __init__(self):
      self.n, self.bad = 0, 0

    def log_value(self, name, value, step=None):
      self.n += 1
      self.bad += 0 if np.isfinite(value) else 1
  sink = Sink()
  t0 = time.time()
  for epoch in range(args.epochs):
    model.logger = LogCollector()
    model.train_start(opt)
    for i in range(args.steps):
      model.train_emb(opt, *train[i % len(train)])
      model.logger.tb_log(sink, step=model.Eiters)        # train.py:215
    meters = {k: m.val for k, m in model.logger.meters.items()}
    assert all(np.isfinite(v) for v in meters.values()), meters
    model.val_start(opt)
    reports = []
    for _ in range(2):
      out = encode_data(opt, model, val, log_step=10 ** 9, logging=quiet, contextual_model=True)
      vid, para = out[0], out[1]
      (rep_i, top_i, ranks_i), (rep_t, top_t, ranks_t) = i2t(vid, para), t2i(vid, para)
      reports.append((dict(rep_i), dict(rep_t), [top_i, ranks_i, top_t, ranks_t]))
    assert reports[0][0] == reports[1][0] and reports[0][1] == reports[1][1] and all(
        np.array_equal(a, b) for a, b in zip(reports[0][2], reports[1][2])), 'validation is not reproducible'
    torch.cuda.synchronize()
    print('epoch %d: %d steps, Le_vid %.4f, r1 i2t %.2f t2i %.2f, alloc %.0f MB, %.1f s'
          % (epoch, args.steps, meters.get('Le_vid', float('nan')), float(reports[0][0].get('r1', float('nan'))), float(reports[0][1].get('r1', float('nan'))),
             torch.cuda.memory_allocated() / 2 ** 20, time.time() - t0))
  from cmhse_amd import ops
  assert ops.async_status() == 0, 'a resident kernel timed out'
  assert sink.bad == 0 and sink.n >= args.epochs * args.steps * 9, (sink.n, sink.bad)
  print('soak ok (%d tensorboard values, all finite)' % sink.n)


if __name__ == '__main__':
  main()
