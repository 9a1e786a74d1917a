#!/usr/bin/env python3
"""Stand-alone cost of one encoder's backward pass at a training batch (S sequences x T steps), with
the BPTT step as one launch (tune.bwd_split_min_seqs=0) or as two (default).  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations.

  python tools/bench_bptt.py --S 152 --T 80 --I 500 --arms "tune.bwd_split_min_seqs=0;tune.bwd_split_min_seqs=33"
"""
import argparse
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

from cmhse_amd import layers  # noqa: E402
import _arms  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--S', type=int, default=152)
  ap.add_argument('--T', type=int, default=80)
  ap.add_argument('--I', type=int, default=500)
  ap.add_argument('--H', type=int, default=1024)
  ap.add_argument('--pool', default='seq2seq')
  ap.add_argument('--arms', default='tune.bwd_split_min_seqs=0;tune.bwd_split_min_seqs=33')
  ap.add_argument('--rounds', type=int, default=4)
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  cls = {'seq2seq': layers.Seq2Seq, 'attention': layers.Attention, 'maxout': layers.Maxout}[args.pool]
  torch.manual_seed(0)
  layer = cls(args.I, args.H).to(dev)
  x = torch.randn(args.S, args.T, args.I, device=dev)
  lens = torch.full((args.S,), args.T, dtype=torch.int64)
  arms = _arms.parse(args.arms)
  res = [[] for _ in arms]
  for rnd in range(args.rounds + 1):
    for i, a in enumerate(arms):
      _arms.apply(a)
      out = layer(x, lens)
      g = torch.ones_like(out)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      out.backward(g)
      e1.record()
      torch.cuda.synchronize()
      layer.zero_grad()
      if rnd:
        res[i].append(e0.elapsed_time(e1))
  for a, r in zip(arms, res):
    print('%-44s backward %.3f ms (min %.3f) = %.1f us per step' % (_arms.label(a), statistics.median(r), min(r),
                                                                    statistics.median(r) / args.T * 1e3))


if __name__ == '__main__':
  main()
