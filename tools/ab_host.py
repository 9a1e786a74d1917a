#!/usr/bin/env python3
"""PCIe-inclusive validation pass (loader batches in pinned host memory) under different settings,
arms interleaved in one process.  Arms: "PIPE=0|1,CHUNK=n;..." (evaluation.PIPELINE_UPLOAD /
UPLOAD_CHUNK)."""
import argparse
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from cmhse_amd import evaluation, ops, synthetic  # noqa: E402
from cmhse_amd.model import VSE  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='anet_icep_val')
  ap.add_argument('--modes', default='')
  ap.add_argument('--rounds', type=int, default=2)
  args = ap.parse_args()
  arms = [dict(kv.split('=') for kv in m.split(',') if kv) for m in args.modes.split(';')]
  keys = sorted({k for a in arms for k in a})
  device = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  wl = dict(bench.WORKLOADS[args.workload])
  opt = bench.make_opt(wl, 'attention', 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset=wl['dataset'])
  nb = (spec.n_videos + wl['batch'] - 1) // wl['batch']
  batches = bench.build_loader(spec, wl, device, 0, nb)
  host = [tuple(t.cpu().pin_memory() if isinstance(t, torch.Tensor) and t.is_cuda else t for t in b)
          for b in batches]
  quiet = lambda *a, **k: None

  def one_pass(src):
    cat, _, _ = evaluation.encode_data_device(opt, model, src, logging=quiet)
    ops.sim_rank(cat['vid_emb'], cat['para_emb'])
    ops.sim_rank(cat['para_emb'], cat['vid_emb'])
    return cat

  def timed(src, n=2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      one_pass(src)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

  def pull_only():
    """The hand-over alone (no compute beside it): GB/s of valid rows."""
    import numpy as np
    lens = np.concatenate([np.asarray(b[4]) for b in host] + [np.asarray(b[6]) for b in host])
    hs = [b[0] for b in host] + [b[2] for b in host]
    ds = [torch.empty(h.shape, dtype=torch.float32, device=device) for h in hs]
    sched = ops.SeqSchedule(lens, device,
                            x_ptrs=np.concatenate([ops.padded_row_ptrs(d) for d in ds]),
                            src_ptrs=np.concatenate([ops.padded_row_ptrs(h) for h in hs]))
    copy = evaluation._copy_stream(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.pull_steps(sched, wl['img_dim'], copy, 8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return float(lens.sum()) * wl['img_dim'] * 4 / dt / 1e9

  ref = one_pass(batches)['vid_emb'].clone()
  print('resident: %.2f ms' % timed(batches))
  res = [[] for _ in arms]
  for rnd in range(args.rounds + 1):
    for i, a in enumerate(arms):
      evaluation.PIPELINE_UPLOAD[0] = a.get('PIPE', '1') == '1'
      evaluation.UPLOAD_CHUNK[0] = int(a.get('CHUNK', '8'))
      if rnd == 0:
        cat = one_pass(host)
        torch.cuda.synchronize()
        print('arm %d identical to resident: %s; pull alone %.1f GB/s'
              % (i, bool(torch.equal(cat['vid_emb'], ref)), pull_only()))
        continue
      res[i].append(timed(host))
      if rnd == 1 and evaluation.PIPELINE_UPLOAD[0]:   # when did the chunks land in a pass?
        ops.PULL_EVENT_TIMING[0] = True
        ops.LAST_PULL_EVENTS[0] = None
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        one_pass(host)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        torch.cuda.synchronize()
        ev = ops.LAST_PULL_EVENTS[0]
        ks = sorted(ev)
        print('  arm %d: pass %.1f ms; chunk landed at ms: %s'
              % (i, e0.elapsed_time(e1), ' '.join('%d:%.0f' % (k, e0.elapsed_time(ev[k])) for k in ks)))
        ops.PULL_EVENT_TIMING[0] = False
  print('resident again: %.2f ms' % timed(batches))
  for a, r in zip(arms, res):
    print('%-50s median %8.2f  min %8.2f ms' % (','.join('%s=%s' % kv for kv in a.items()),
                                                 statistics.median(r), min(r)))


if __name__ == '__main__':
  main()
