#!/usr/bin/env python3
"""One rank's share of the sharded validation pass, on ONE MI355X (the single-GPU proxy for the
N-GPU strong-scaling efficiency before communication, SURVEY 8e / VERDICT r04 item 1c).

This process plays rank `--rank` of `--world`: it is dealt its batches of the split by work
(parallel_eval.assign_batches, as bench.py --gpus N does), encodes them with the WHOLE split's step
plan (evaluation.split_step_plan: the kernel kind of every time step is the one the single process
picks, which is what makes the integer ranks independent of the world size), places its normalised
rows in [N, D] gallery matrices whose other rows are random unit vectors (what the all-gather would
bring), ranks its row stripe in both directions and brings the ranks to the host.  Timed like
bench.py: W warm-up passes, K timed passes between synchronisations.

  python tools/rank_share.py --world 8 [--rank 0] [--plan 1] [--steps 12] [--warmup 3]
Prints one JSON line.  --plan 0 encodes the share with its OWN step counts (round 4's behaviour).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

from bench_common import WORKLOADS, build_loader, make_opt  # noqa: E402
from cmhse_amd import evaluation, ops, parallel_eval, synthetic  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402


def plan_of_lengths(lengths, num_clips):
  """split_step_plan from the sizes alone (synthetic.batch_lengths), nothing materialised."""
  v1 = np.concatenate([np.concatenate([lc, lv]) for lc, lv, _, _ in lengths])
  t1 = np.concatenate([np.concatenate([lw, lp]) for _, _, lw, lp in lengths])
  v2 = np.asarray(num_clips, dtype=np.int64)
  return dict(v1=ops.step_counts(v1), t1=ops.step_counts(t1), v2=ops.step_counts(v2), t2=ops.step_counts(v2))


def main(argv=None, quiet_print=False):
  ap = argparse.ArgumentParser()
  ap.add_argument('--world', type=int, default=8)
  ap.add_argument('--rank', type=int, default=0)
  ap.add_argument('--plan', type=int, default=1)
  ap.add_argument('--steps', type=int, default=12)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--workload', default='anet_icep_val', choices=sorted(WORKLOADS))
  ap.add_argument('--rnn_type', default='attention')
  ap.add_argument('--embed', type=int, default=1024)
  ap.add_argument('--deal', default='tail', choices=['tail', 'lpt'],
                  help="tail: the deal prices a rank's longest paragraph (round 6); lpt: work only, longest paragraph first (round 5)")
  ap.add_argument('--tune', default='', help='crossovers to move for the run: name=value,name=value (ops.tune)')
  ap.add_argument('--all_ranks', type=int, default=0,
                  help='1: every rank of the deal one after another in this process, then one summary line '
                       '(slowest rank = what an N-GPU pass costs before communication) beside the whole split (world 1)')
  args = ap.parse_args(argv)
  if args.all_ranks:
    base = list(argv if argv is not None else sys.argv[1:])

    def strip(opts, name):      # drop `name value`
      out, skip = [], False
      for a in opts:
        if skip:
          skip = False
          continue
        if a == name:
          skip = True
          continue
        out.append(a)
      return out
    base = strip(strip(strip(base, '--rank'), '--all_ranks'), '--world')
    whole = main(base + ['--world', '1', '--rank', '0'], quiet_print=True)
    rows = [main(base + ['--world', str(args.world), '--rank', str(r)], quiet_print=True) for r in range(args.world)]
    slow = max(r['ms_per_pass'] for r in rows)
    print(json.dumps({'world': args.world, 'deal': args.deal, 'whole_split_ms': whole['ms_per_pass'],
                      'slowest_rank_ms': slow, 'implied_efficiency': whole['ms_per_pass'] / (args.world * slow),
                      'ranks': [{k: r[k] for k in ('rank', 'videos', 'longest_paragraph', 'gru_tflop', 'ms_per_pass')} for r in rows]}))
    return None
  for kv in [x for x in args.tune.split(',') if x]:
    k, v = kv.split('=')
    ops.tune(k, int(v))
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  wl = dict(WORKLOADS[args.workload])
  opt = make_opt(wl, args.rnn_type, args.embed)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset=wl['dataset'])
  lengths = synthetic.batch_lengths(spec, wl['batch'])
  costs = [parallel_eval.batch_cost(lc, lv, lw, lp, wl['img_dim'], 300, args.embed) for lc, lv, lw, lp in lengths]
  tail_from = parallel_eval.tail_horizon(np.concatenate([np.concatenate([lw, lp]) for _, _, lw, lp in lengths]))
  assignment = parallel_eval.assign_batches(costs, args.world, tail_from=tail_from if args.deal == 'tail' else None)
  own = assignment[args.rank]
  batches = [b for i, b in enumerate(build_loader(spec, wl, dev, own)) if i in set(own)]
  step_plan = plan_of_lengths(lengths, spec.num_clips) if args.plan else None
  N, D = spec.n_videos, args.embed
  n_own = sum(len(b[8]) for b in batches)
  row0 = sum(sum(len(range(i * wl['batch'], min(N, (i + 1) * wl['batch']))) for i in assignment[r])
             for r in range(args.rank))
  g = torch.Generator(device=dev).manual_seed(7)
  V = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
  P = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
  quiet = lambda *a, **k: None

  def one_pass():
    cat, _, _, fin = evaluation.encode_data_device(opt, model, batches, logging=quiet, defer_logging=True,
                                                   step_plan=step_plan)
    V[row0:row0 + n_own] = cat['vid_emb']          # (the all-gather's copy of the own rows)
    P[row0:row0 + n_own] = cat['para_emb']
    r_i, t_i = ops.sim_rank(V, P, row0, n_own)
    r_t, t_t = ops.sim_rank(P, V, row0, n_own)
    packed = torch.stack([r_i, t_i, r_t, t_t])
    fin()
    return packed.cpu().numpy()

  for _ in range(args.warmup):
    one_pass()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  marks = [t0]
  for _ in range(args.steps):
    one_pass()
    marks.append(time.perf_counter())
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / args.steps * 1e3
  per = sorted((b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:]))
  result = ({'world': args.world, 'rank': args.rank, 'deal': args.deal, 'plan': bool(args.plan), 'videos': n_own,
                    'longest_paragraph': int(max(costs[i][1] for i in own)),
                    'stripe': '%d x %d' % (n_own, N), 'ms_per_pass': ms, 'pass_ms_min': per[0],
                    'pass_ms_median': per[len(per) // 2], 'pass_ms_max': per[-1], 'steps': args.steps,
                    'workload': args.workload, 'rnn_type': args.rnn_type, 'tune': args.tune,
                    'gru_tflop': float(sum(costs[i][0] for i in own)) / 1e12})
  del batches, V, P, model
  torch.cuda.empty_cache()
  if not quiet_print:
    print(json.dumps(result))
  return result


if __name__ == '__main__':
  main()
