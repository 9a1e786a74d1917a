#!/usr/bin/env python3
"""Is a training step host-bound, and what does each way of feeding it cost?  

Arms, each `--steps` un-synchronised steps after warm-up, interleaved `--rounds` times on one box:
  resident          batches in HBM (what bench.py's train_steps legs time)
  resident+spin N   the same with N us of host busy-wait in front of every step: if the step time
                    grows by N, the host is on the critical path
  upload            pinned host 12-tuples, `.cuda()` in front of the step (the reference, model.py:225-227)
  pull              pinned host 12-tuples pulled under the chain (model.HOST_PULL)
  prefetch:<when>:<stream>[:noprep]
                    collate.DevicePrefetcher: when = before | mid; stream = copy (ops.copy_stream,
                    the default) | s3 (stream_set[3]) | new (a fresh torch stream) | s2; noprep =
                    without prepare_batch
  pull:<stream>     model.HOST_FEED 'pull' on that stream
  <arm>+tb          the arm with model.logger.tb_log(sink, step) after every step, as train.py:215 does
  auto | ahead      model.HOST_FEED 'auto' (the default: the batch copied ahead on the copy stream while
                    the host leads the GPU, else the pull) | 'ahead' (always the copy)
Also prints the host time spent inside train_emb per step (the launch-queueing cost).

  python tools/host_lead.py --config icep --steps 30 --rounds 2
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import bench  # noqa: E402
from cmhse_amd import collate, model as model_mod, ops, synthetic  # noqa: E402
from cmhse_amd.evaluation import LogCollector  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402
from train_profile import CONFIGS  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--config', default='icep', choices=sorted(CONFIGS))
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--rounds', type=int, default=2)
  ap.add_argument('--arms', default='resident,resident+spin1000,upload,auto,ahead,pull:copy,pull:s3,'
                                    'prefetch:before:copy,prefetch:before:s3,prefetch:before:new,'
                                    'prefetch:mid:copy')
  args = ap.parse_args()
  cfg = dict(CONFIGS[args.config])
  wl = dict(bench.WORKLOADS[cfg.pop('workload')])
  opt = bench.make_opt(wl, 'attention', 1024)
  for k, v in cfg.items():
    setattr(opt, k, v)
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  torch.manual_seed(1)
  model = VSE(opt)
  model.logger = LogCollector()
  model.train_start(opt)
  spec = synthetic.anet_like_spec(32 * 10, seed=0, dataset=wl['dataset'])
  batches = synthetic.make_batches(spec, 32, wl['img_dim'], wl['vocab'], seed=0, feat=wl['feat'])
  host = [tuple(t.pin_memory() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b))
          for b in batches]
  res = [tuple(t.cuda() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b))
         for b in batches]
  extra = torch.cuda.Stream(dev)

  class Sink(object):
    def log_value(self, name, value, step=None):
      pass
  sink = Sink()
  n = args.steps
  pick = lambda src: [src[i % len(src)] for i in range(n)]

  def loop(arm):
    model_mod.HOST_PULL[0] = True
    spin = 0
    tb_every_step = arm.endswith('+tb')      # train.py:215: model.logger.tb_log(...) after every step
    arm = arm[:-3] if tb_every_step else arm
    if arm.startswith('resident'):
      src = pick(res)
      if '+spin' in arm:
        spin = int(arm.split('+spin')[1])
    elif arm == 'upload':
      model_mod.HOST_PULL[0] = False
      src = pick(host)
    elif arm in ('auto', 'ahead'):
      model_mod.HOST_FEED[0] = arm
      model_mod.HOST_PULL_STREAM[0] = None
      src = pick(host)
    elif arm.startswith('pull'):
      model_mod.HOST_FEED[0] = 'pull'
      which = arm.split(':')[1] if ':' in arm else 'copy'
      model_mod.HOST_PULL_STREAM[0] = {'copy': None, 's3': ops.stream_set(dev)[3], 'new': extra,
                                       's2': ops.stream_set(dev)[2]}[which]
      src = pick(host)
    else:
      parts = arm.split(':')
      stream = {'copy': None, 's3': ops.stream_set(dev)[3], 'new': extra, 's2': ops.stream_set(dev)[2]}[parts[2]]
      src = collate.DevicePrefetcher(pick(host), model=model, when=parts[1], stream=stream,
                                     prepare=(lambda b: b) if 'noprep' in parts else None)
    in_step = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in src:
      if spin:
        t_end = time.perf_counter() + spin * 1e-6
        while time.perf_counter() < t_end:
          pass
      h0 = time.perf_counter()
      model.train_emb(opt, *b)
      if tb_every_step:
        model.logger.tb_log(sink, step=model.Eiters)
      in_step += time.perf_counter() - h0
    str(model.logger)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, in_step / n * 1e3

  arms = [a for a in args.arms.split(',') if a]
  for a in arms:          # warm-up: every arm's shapes, pools and streams seen once
    loop(a)
  out = {a: [] for a in arms}
  for _ in range(args.rounds):
    for a in arms:
      out[a].append(loop(a))
  print('%s, %d steps per arm and round; ms per step (host ms inside train_emb per step)' % (args.config, n))
  base = min(x[0] for x in out[arms[0]])
  for a in arms:
    print('  %-32s %s   best %.2f (x%.3f)' % (a, '  '.join('%.2f (%.2f)' % x for x in out[a]),
                                                min(x[0] for x in out[a]), min(x[0] for x in out[a]) / base))


if __name__ == '__main__':
  main()
