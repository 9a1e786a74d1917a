#!/usr/bin/env python3
"""Full-size check of the step chain (gru_step_chain_kernel) against per-step launches with NEW input
values every round, the chained pass FIRST: the workspaces come back from the allocator holding the
previous round's states, so a tile that read a state row too early (or from a stale cache line)
would see the old round's value and the six embedding tensors would differ from the per-step pass
that follows.  (bench.py and the A/B tools repeat one computation, which cannot show that.)

  python tools/chain_stress.py [--rounds 4] [--n_videos 0] [--workload anet_icep_val]
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import bench  # noqa: E402
from cmhse_amd import _lib, ops, synthetic  # noqa: E402
from cmhse_amd.evaluation import encode_data_device  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rounds', type=int, default=4)
  ap.add_argument('--n_videos', type=int, default=0)
  ap.add_argument('--workload', default='anet_icep_val')
  ap.add_argument('--rnn_type', default='attention')
  ap.add_argument('--sizes', default='', help='comma-separated n_videos to run one after the other (overrides --n_videos)')
  args = ap.parse_args()
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  wl = dict(bench.WORKLOADS[args.workload])
  opt = bench.make_opt(wl, args.rnn_type, 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  sizes = [int(x) for x in args.sizes.split(',') if x] or [args.n_videos or wl['n_videos']]
  bad = 0
  for i, nv in enumerate(sizes):
    bad += one_size(args, wl, opt, model, dev, nv, seed=i)
  ops.tune('chain_min_steps', 2)
  sys.exit(1 if bad else 0)


def one_size(args, wl, opt, model, dev, n_videos, seed):
  spec = synthetic.anet_like_spec(n_videos, seed=seed, dataset=wl['dataset'])
  batches = bench.build_loader(spec, wl, dev, 0, (spec.n_videos + 31) // 32)
  g = torch.Generator(device=dev).manual_seed(5)
  quiet = lambda *a, **k: None
  keys = ('vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx')

  def run(min_steps):
    ops.tune('chain_min_steps', min_steps)
    cat, _, _, fin = encode_data_device(opt, model, batches, logging=quiet, defer_logging=True)
    fin()
    torch.cuda.synchronize()
    assert _lib.load().cmhse_async_status(0) == 0
    return {k: cat[k].clone() for k in keys}

  bad = 0
  for r in range(args.rounds):
    for b in batches:
      b[0].normal_(generator=g)                    # clip features
      b[2].normal_(generator=g)                    # whole-video features
      b[1].random_(0, wl['vocab'], generator=g)    # sentence tokens
      b[3].random_(0, wl['vocab'], generator=g)    # paragraph tokens
    chained = run(2)
    per_step = run(0)
    diffs = {k: float((chained[k] - per_step[k]).abs().max()) for k in keys}
    same = all(torch.equal(chained[k], per_step[k]) for k in keys)
    bad += 0 if same else 1
    print('%5d videos, round %d: %s  max |diff| %s' % (n_videos, r, 'bit-identical' if same else 'MISMATCH',
                                                        ' '.join('%s %.2g' % kv for kv in diffs.items())))
  return bad


if __name__ == '__main__':
  main()
