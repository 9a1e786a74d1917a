#!/usr/bin/env python3
"""A/B of schedules and kernel-shape crossovers on the TRAINING step (VSE.train_emb), inside one
process with the arms interleaved (tools/_arms.py: cmhse_tune crossovers and the host-side schedule
attributes).

  python tools/ab_train.py --config c3d --modes "schedule=towers;schedule=interleaved,side_streams=0;"
"""
import argparse
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import bench  # noqa: E402
from cmhse_amd import synthetic  # noqa: E402
from cmhse_amd.evaluation import LogCollector  # noqa: E402
from cmhse_amd import model as model_mod  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402
from train_profile import CONFIGS  # noqa: E402
import _arms  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--config', default='c3d', choices=sorted(CONFIGS))
  ap.add_argument('--modes', required=True, help='arms separated by ";", each "K=V,K=V"')
  ap.add_argument('--rounds', type=int, default=5)
  ap.add_argument('--steps', type=int, default=8)
  ap.add_argument('--rnn_type', default='attention')
  ap.add_argument('--resident', type=int, default=1, help='1: batches resident in HBM (what bench.py times); 0: pinned host batches')
  args = ap.parse_args()
  arms = _arms.parse(args.modes)
  cfg = dict(CONFIGS[args.config])
  wl = dict(bench.WORKLOADS[cfg.pop('workload')])
  opt = bench.make_opt(wl, args.rnn_type, 1024)
  for k, v in cfg.items():
    setattr(opt, k, v)
  torch.cuda.set_device(0)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(32 * 4, seed=0, dataset=wl['dataset'])
  batches = synthetic.make_batches(spec, 32, wl['img_dim'], wl['vocab'], seed=0, feat=wl['feat'])
  if args.resident:
    batches = [tuple(t.cuda() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b)) for b in batches]
  else:
    batches = [tuple(t.pin_memory() if isinstance(t, torch.Tensor) else t for t in b) for b in batches]
  model.logger = LogCollector()
  model.train_start(opt)

  def run(arm, n):
    _arms.apply(arm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
      model.train_emb(opt, *batches[i % len(batches)])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

  for a in arms:
    run(a, 3)
  res = [[] for _ in arms]
  for _ in range(args.rounds):
    for i, a in enumerate(arms):
      res[i].append(run(a, args.steps))
  print('%-56s %10s %10s   (ms per train_emb step, %s)' % ('arm', 'median', 'min', args.config))
  for a, r in zip(arms, res):
    print('%-56s %10.2f %10.2f' % (_arms.label(a),
                                   statistics.median(r), min(r)))


if __name__ == '__main__':
  main()
