#!/usr/bin/env python3
"""K timed VSE.train_emb steps after a warm-up and a 300 ms idle gap (the gap lets
tools/trace_busy.py find the timed region in a rocprofv3 kernel trace of this script).

  python tools/train_profile.py --config icep_recon --steps 10
  rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/train_profile.py ...
"""
import argparse
import copy
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from cmhse_amd import synthetic  # noqa: E402
from cmhse_amd.evaluation import LogCollector  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402

CONFIGS = {
    # BASELINE configs[1]: ActivityNet C3D, HSE tau=0, batch 32 (bench.py's train_step line)
    'c3d': dict(workload='anet_c3d_val', low_level_loss=True, norm=True),
    # BASELINE configs[2]: ActivityNet ICEP, tau=5e-4, --low_level_loss --reconstruct_loss --norm
    'icep_recon': dict(workload='anet_icep_val', low_level_loss=True, norm=True,
                       reconstruct_loss=True, weight_recon=0.0005, decode_rnn_type='seq2seq'),
    'icep': dict(workload='anet_icep_val', low_level_loss=True, norm=True),
    # BASELINE configs[3]: DiDeMo ICEP, all clips 80 frames
    'didemo_recon': dict(workload='didemo_icep_val', low_level_loss=True, norm=True,
                         reconstruct_loss=True, weight_recon=0.0005, decode_rnn_type='seq2seq'),
}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--config', default='icep_recon', choices=sorted(CONFIGS))
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--rnn_type', default='attention')
  ap.add_argument('--resident', type=int, default=1, help='1: batches resident in HBM (what bench.py times); 0: pinned host batches')
  ap.add_argument('--feed', default='', choices=['', 'resident', 'pull', 'auto', 'ahead', 'upload', 'prefetch'],
                  help="how a batch reaches train_emb: resident (in HBM); pinned host tensors pulled "
                       "under the chain (pull) or .cuda()'d in front of the step (upload); pinned host "
                       "tensors through collate.DevicePrefetcher + prepare_batch (prefetch)")
  ap.add_argument('--timeline', type=int, default=0, help='host (h) / GPU (g) ms at phase marks')
  args = ap.parse_args()
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
  feed = args.feed or ('resident' if args.resident else 'auto')
  if feed == 'resident':
    batches = [tuple(t.cuda() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b)) for b in batches]
  else:
    batches = [tuple(t.pin_memory() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b)) for b in batches]
  from cmhse_amd import collate, model as model_mod
  model_mod.HOST_PULL[0] = feed != 'upload'
  if feed in ('pull', 'auto', 'ahead'):
    model_mod.HOST_FEED[0] = feed
  wrap = (lambda bs: collate.DevicePrefetcher(bs, prepare=model.prepare_batch)) if feed == 'prefetch' else (lambda bs: bs)
  model.logger = LogCollector()
  model.train_start(opt)
  use = [batches[i % len(batches)] for i in range(args.steps + 6)]
  for b in wrap(use[:6]):
    model.train_emb(opt, *b)
  torch.cuda.synchronize()
  time.sleep(0.3)
  t0 = time.perf_counter()
  for b in wrap(use[6:]):
    model.train_emb(opt, *b)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / args.steps
  print('%s [%s]: %.2f ms per train_emb step (%d steps, batch 32, img_dim %d, %s)'
        % (args.config, feed, dt * 1e3, args.steps, wl['img_dim'], args.rnn_type))
  if args.timeline:
    from cmhse_amd import model as model_mod
    for b in use[6:9]:
      torch.cuda.synchronize()
      model_mod.TRACE = []
      from cmhse_amd import layers as layers_mod

      def mark(name, stream, _tr=model_mod.TRACE):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        _tr.append((name, time.perf_counter(), ev))
      layers_mod.MARK = mark
      model.train_emb(opt, *b)
      torch.cuda.synchronize()
      layers_mod.MARK = None
      tr, model_mod.TRACE = model_mod.TRACE, None
      h0, e0 = tr[0][1], tr[0][2]
      print('  ' + '  '.join('%s h%.2f g%.2f' % (n, (h - h0) * 1e3, e0.elapsed_time(e))
                             for n, h, e in tr))
  if args.timeline >= 2:
    # Steady state, NO synchronisation between steps (the host keeps its lead): GPU time of every
    # phase mark relative to the step's first mark, mean over the steps.  Marks on the caller's
    # stream (model._tick) are join points; marks [0] / [1] sit on the visual / text tower's stream.
    from cmhse_amd import layers as layers_mod, model as model_mod
    n = args.steps
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(32 * n)]
    for ev in pool:
      ev.record()
    torch.cuda.synchronize()
    trace = []

    def mark(name, stream=None):
      ev = pool.pop()
      ev.record(stream) if stream is not None else ev.record()
      trace.append((name, time.perf_counter(), ev))
    model_mod._tick = lambda name: mark(name)
    layers_mod.MARK = mark
    t0 = time.perf_counter()
    for b in wrap(use[6:]):
      model.train_emb(opt, *b)
    torch.cuda.synchronize()
    layers_mod.MARK = None
    dt2 = (time.perf_counter() - t0) / n
    steps_tr, cur = [], None
    for name, h, ev in trace:
      if name == 'step:start':
        cur = []
        steps_tr.append(cur)
      if cur is not None:
        cur.append((name, h, ev))
    names = [x[0] for x in steps_tr[-1]]
    print('  steady state, %.2f ms per step (marks on): mean GPU ms after step:start [host ms]' % (dt2 * 1e3))
    for k, nm in enumerate(names):
      g = [st[0][2].elapsed_time(st[k][2]) for st in steps_tr[2:] if len(st) == len(names)]
      h = [(st[k][1] - st[0][1]) * 1e3 for st in steps_tr[2:] if len(st) == len(names)]
      if g:
        print('    %-16s g %6.2f  h %6.2f' % (nm, sum(g) / len(g), sum(h) / len(h)))
    # how far the host is ahead: GPU time of step k's start minus host time of its queueing, as lag
    lag = []
    for st in steps_tr[2:]:
      lag.append(steps_tr[2][0][2].elapsed_time(st[0][2]) - (st[0][1] - steps_tr[2][0][1]) * 1e3)
    print('    GPU start of a step lags its host queueing by (ms, relative to the first): ' +
          ' '.join('%.1f' % v for v in lag))
  print(str(model.logger))


if __name__ == '__main__':
  main()
