#!/usr/bin/env python3
"""Where the milliseconds of the reference-API validation pass go (train.py:223-236 through
cmhse_amd.evaluation): host wall-clock marks inside encode_data / i2t / t2i (evaluation.TRACE) over a
few passes of the bench workload, next to the device-resident pass of bench.py.

  python tools/api_path_profile.py [--n_videos N] [--passes K] [--host 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

from bench_common import WORKLOADS, build_loader, make_opt  # noqa: E402
from cmhse_amd import evaluation as ev, ops, synthetic  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n_videos', type=int, default=4917)
  ap.add_argument('--passes', type=int, default=6)
  ap.add_argument('--host', type=int, default=0)
  ap.add_argument('--push_wgs', type=int, default=8, help='workgroups of cmhse_push_rows')
  ap.add_argument('--push_waves', type=int, default=1, help='wavefronts per workgroup of cmhse_push_rows')
  ap.add_argument('--late_wgs', type=int, default=32, help='workgroups for the two level-2 matrices')
  ap.add_argument('--pull_waves', type=int, default=0, help='cmhse_tune pull_waves (0 = default)')
  ap.add_argument('--brief', type=int, default=0)
  ap.add_argument('--blit', type=int, default=0, help='1: stage with hipMemcpyAsync instead (A/B)')
  args = ap.parse_args()
  ops.PUSH_WORKGROUPS[0] = args.push_wgs
  ops.PUSH_WAVES[0] = args.push_waves
  ops.PUSH_WORKGROUPS_LATE[0] = args.late_wgs
  if args.pull_waves:
    ops.tune('pull_waves', args.pull_waves)
  ev.PUSH_KERNEL[0] = not args.blit
  print('staging: %s' % ('hipMemcpyAsync' if args.blit else
                         'cmhse_push_rows, %d workgroups x %d waves' % (args.push_wgs, args.push_waves)))
  wl = dict(WORKLOADS['anet_icep_val'], n_videos=args.n_videos)
  opt = make_opt(wl, 'attention', 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset='anet')
  dev = torch.device('cuda', 0)
  n_b = (spec.n_videos + wl['batch'] - 1) // wl['batch']
  batches = build_loader(spec, wl, dev, range(n_b))
  if args.host:
    pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t)
    batches = [tuple(pin(t) if isinstance(t, torch.Tensor) and t.is_cuda else t for t in b) for b in batches]
  quiet = lambda *a, **k: None

  def device_pass():
    cat, _, _, fin = ev.encode_data_device(opt, model, batches, logging=quiet, defer_logging=True)
    r_i, t_i = ops.sim_rank(cat['vid_emb'], cat['para_emb'])
    r_t, t_t = ops.sim_rank(cat['para_emb'], cat['vid_emb'])
    packed = torch.stack([r_i, t_i, r_t, t_t])
    fin()
    return packed.cpu().numpy()

  def api_pass():
    out = ev.encode_data(opt, model, batches, 10, quiet)
    ev.TRACE and ev.TRACE.append(('encode_data returned', time.perf_counter()))
    a = ev.i2t(out[0], out[1])
    ev.TRACE and ev.TRACE.append(('i2t returned', time.perf_counter()))
    b = ev.t2i(out[0], out[1])
    ev.TRACE and ev.TRACE.append(('t2i returned', time.perf_counter()))
    return a, b

  for fn in (device_pass, api_pass):
    fn(); fn()
    torch.cuda.synchronize()
  import gc
  gc_log = []
  gc.callbacks.append(lambda phase, info, _t=[0.0]: (_t.__setitem__(0, time.perf_counter()) if phase == 'start' else
                                                      gc_log.append((info.get('generation'), (time.perf_counter() - _t[0]) * 1e3))))
  for name, fn in (('device pass', device_pass), ('api pass', api_pass)):
    ts = []
    for _ in range(args.passes):
      torch.cuda.synchronize()
      del gc_log[:]
      if name == 'api pass':
        ev.TRACE = [('start', time.perf_counter())]
      t0 = time.perf_counter()
      fn()
      torch.cuda.synchronize()
      ts.append((time.perf_counter() - t0) * 1e3)
      if name == 'api pass':
        marks, ev.TRACE = ev.TRACE, None
        if ts[-1] > 1.1 * min(ts):
          print('  slow api pass %.1f ms: gc %s; phases %s' % (ts[-1], [(g, round(m, 1)) for g, m in gc_log if m > 1.0],
                ['%s %.1f' % (b[0][:28], (b[1] - a[1]) * 1e3) for a, b in zip(marks[:-1], marks[1:]) if (b[1] - a[1]) > 2e-3]))
    print('%-12s ms per pass: %s  median %.2f' % (name, ' '.join('%.2f' % t for t in ts), float(np.median(ts))))
  if args.brief:
    return
  # phase marks of the api pass
  acc = {}
  for _ in range(args.passes):
    torch.cuda.synchronize()
    ev.TRACE = [('start', time.perf_counter())]
    api_pass()
    marks, ev.TRACE = ev.TRACE, None
    for (n0, t0), (n1, t1) in zip(marks[:-1], marks[1:]):
      acc.setdefault(n1, []).append((t1 - t0) * 1e3)
  print('phase (host ms since the previous mark; median of %d passes)' % args.passes)
  for k, v in acc.items():
    print('  %-44s %8.3f' % (k, float(np.median(v))))


if __name__ == '__main__':
  main()
