#!/usr/bin/env python3
"""Where does the HOST lose time inside a bench pass?  Runs bench.py's main() with the usual
suspects wrapped (allocations, staging uploads, schedule builds, the library's launch calls) and
prints every call that took longer than a few ms, between the per-pass host marks
(CMHSE_BENCH_DEBUG=1).  Found round 2's mid-pass page-locking stalls.

  python tools/host_stalls.py [bench.py arguments]
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402
from cmhse_amd import evaluation, ops  # noqa: E402


def wrap(mod, name, thresh=4.0, label=None):
  fn = getattr(mod, name)

  def w(*a, **k):
    t = time.perf_counter()
    r = fn(*a, **k)
    d = (time.perf_counter() - t) * 1e3
    if d > thresh:
      sys.stderr.write('  SLOW %s %.1f ms\n' % (label or name, d))
    return r
  setattr(mod, name, w)


def main():
  wrap(ops, 'upload')
  wrap(ops, '_prepare_fwd', 8.0)
  wrap(ops, 'gru_pool_fwd_multi', 15.0)
  wrap(ops, 'sim_rank', 4.0)
  wrap(ops, 'contrastive_blocks_fwd', 4.0)
  wrap(evaluation, 'encode_group', 15.0)
  wrap(evaluation, '_group_batches')
  wrap(torch, 'empty')
  wrap(torch, 'zeros')
  wrap(torch, 'cat')
  init = ops.SeqSchedule.__init__

  def timed_init(self, *a, **k):
    t = time.perf_counter()
    init(self, *a, **k)
    d = (time.perf_counter() - t) * 1e3
    if d > 4:
      sys.stderr.write('  SLOW SeqSchedule %.1f ms\n' % d)
  ops.SeqSchedule.__init__ = timed_init
  os.environ['CMHSE_BENCH_DEBUG'] = '1'
  sys.argv = ['bench.py'] + sys.argv[1:]
  bench.main()


if __name__ == '__main__':
  main()
