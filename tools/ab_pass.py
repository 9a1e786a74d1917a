#!/usr/bin/env python3
"""A/B of kernel-shape crossovers / schedules on the validation pass, inside ONE process with the
arms interleaved (cdna_hip_programming.md §5.4 rule 24; arms: tools/_arms.py).

  python tools/ab_pass.py --modes "tune.tall_tile_min_wgs=0;tune.tall_tile_min_wgs=2048" --rounds 3

Each arm: `--passes` timed passes (encode_data_device + i2t + t2i) per round; prints per arm the
median / min ms per pass and the LDS-tiled step kernel's event-timed TFLOP/s.
"""
import argparse
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from cmhse_amd import ops, synthetic  # noqa: E402
from cmhse_amd.evaluation import encode_data_device  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402
sys.path.insert(0, os.path.join(REPO, 'tools'))
import _arms  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='anet_icep_val')
  ap.add_argument('--modes', required=True, help='arms separated by ";", each "K=V,K=V"')
  ap.add_argument('--rounds', type=int, default=3)
  ap.add_argument('--passes', type=int, default=2)
  ap.add_argument('--n_videos', type=int, default=0)
  args = ap.parse_args()
  arms = _arms.parse(args.modes)
  device = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  wl = dict(bench.WORKLOADS[args.workload])
  if args.n_videos:
    wl['n_videos'] = args.n_videos
  opt = bench.make_opt(wl, 'attention', 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset=wl['dataset'])
  nb = (spec.n_videos + wl['batch'] - 1) // wl['batch']
  batches = bench.build_loader(spec, wl, device, 0, nb)
  quiet = lambda *a, **k: None

  def one_pass():
    cat, _, _ = encode_data_device(opt, model, batches, logging=quiet)
    ops.sim_rank(cat['vid_emb'], cat['para_emb'])
    ops.sim_rank(cat['para_emb'], cat['vid_emb'])
    return cat

  def set_arm(a):
    _arms.apply(a)

  ref = None
  res = [dict(ms=[], tf=[]) for _ in arms]
  for rnd in range(args.rounds + 1):     # round 0 = warm-up of every arm
    for i, a in enumerate(arms):
      set_arm(a)
      if rnd == 0:
        cat = one_pass()
        torch.cuda.synchronize()
        if ref is None:
          ref = cat['vid_emb'].clone()
        else:
          print('arm %d max |vid_emb - arm0| = %.3g' % (i, float((cat['vid_emb'] - ref).abs().max())))
        continue
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      with ops.StepTimers() as timers:
        for _ in range(args.passes):
          one_pass()
        torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / args.passes
      spans = timers.collect()
      ms = sum(s[3][0] for s in spans)
      fl = sum(s[3][1] for s in spans)
      res[i]['ms'].append(dt * 1e3)
      res[i]['tf'].append(fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
  print('%-40s %10s %10s %12s' % ('arm', 'median ms', 'min ms', 'tiled TF/s'))
  for a, r in zip(arms, res):
    print('%-40s %10.2f %10.2f %12.1f' % (','.join('%s=%s' % kv for kv in a.items()),
                                          statistics.median(r['ms']), min(r['ms']),
                                          statistics.median(r['tf'])))


if __name__ == '__main__':
  main()
