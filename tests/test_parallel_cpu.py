"""World-size-2 (and 3) gloo test of the sharded validation: partitioning, the all-gather of
unequal shards, stripe ranking and the merge — with the CPU oracle standing in for the two HIP
calls (tests only; the product defaults have no fallback)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, n_videos, batch, out_dir, scramble=False):
  sys.path.insert(0, REPO)
  sys.path.insert(0, os.path.join(REPO, 'oracle'))
  import cmhse_oracle as oracle
  from cmhse_amd import parallel_eval, synthetic
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  a, b = synthetic.correlated_embeddings(n_videos, 48, 2.0, seed=11)
  # fake "loader": batch k carries its slice of precomputed embeddings; num_clips in slot 8
  batches = []
  for k0 in range(0, n_videos, batch):
    k1 = min(n_videos, k0 + batch)
    bt = [None] * 12
    bt[8] = tuple([1] * (k1 - k0))
    bt[0] = (a[k0:k1], b[k0:k1])
    batches.append(tuple(bt))

  def encode_fn(opt, model, mine):
    if not mine:
      return None
    return (torch.from_numpy(np.concatenate([m[0][0] for m in mine])),
            torch.from_numpy(np.concatenate([m[0][1] for m in mine])))

  def rank_fn(q, g, row0, nrows):
    d = q[row0:row0 + nrows].numpy().astype(np.float64) @ g.numpy().astype(np.float64).T
    diag = d[np.arange(nrows), row0 + np.arange(nrows)][:, None]
    return (torch.from_numpy((d > diag).sum(1).astype(np.int32)),
            torch.from_numpy(d.argmax(1).astype(np.int32)))

  assignment = None
  if scramble:   # a non-contiguous deal: exercises the permutation back to loader order
    nb = len(batches)
    assignment = [[i for i in range(nb) if (i * 7 + 3) % world == r] for r in range(world)]
  res = parallel_eval.validate_sharded(None, None, batches, encode_fn=encode_fn, rank_fn=rank_fn,
                                       device='cpu', dim=48, assignment=assignment)
  np.savez(os.path.join(out_dir, 'r%d.npz' % rank), ranks_i=res[2], ranks_t=res[3],
           top1_i=res[4], top1_t=res[5], rep_i=np.array(sorted(res[0].items()), dtype=object)[:, 1]
           .astype(np.float64))
  dist.destroy_process_group()


@pytest.mark.parametrize('world,n_videos,batch,scramble',
                         [(2, 37, 5, False), (3, 10, 4, False), (2, 3, 4, False), (2, 37, 5, True),
                          (3, 41, 4, True)])
def test_sharded_validation_matches_single_process(tmp_path, oracle, world, n_videos, batch,
                                                   scramble):
  from cmhse_amd import synthetic
  port = _free_port()
  mp.spawn(_worker, args=(world, port, n_videos, batch, str(tmp_path), scramble), nprocs=world,
           join=True)
  a, b = synthetic.correlated_embeddings(n_videos, 48, 2.0, seed=11)
  _, top1_i, ranks_i = oracle.i2t(a, b, np.float64)
  _, top1_t, ranks_t = oracle.t2i(a, b, np.float64)
  for r in range(world):
    got = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
    np.testing.assert_array_equal(got['ranks_i'], ranks_i)
    np.testing.assert_array_equal(got['ranks_t'], ranks_t)
    np.testing.assert_array_equal(got['top1_i'], top1_i)
    np.testing.assert_array_equal(got['top1_t'], top1_t)


def _stub_worker(rank, world, port, out_dir, lie):
  """Each rank materialises only the batches a count-based deal gives it (the others are stubs
  that carry only num_clips) and calls validate_sharded WITHOUT an assignment."""
  sys.path.insert(0, REPO)
  from cmhse_amd import parallel_eval, synthetic
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  n_videos, batch = 23, 4
  a, b = synthetic.correlated_embeddings(n_videos, 16, 2.0, seed=5)
  sizes = [min(n_videos, k0 + batch) - k0 for k0 in range(0, n_videos, batch)]
  own = parallel_eval.assign_batches([(float(s), 0) for s in sizes], world)[rank]
  batches = []
  for i, k0 in enumerate(range(0, n_videos, batch)):
    k1 = min(n_videos, k0 + batch)
    bt = [None] * 12
    bt[8] = tuple([1] * (k1 - k0))
    if i in own:   # materialised: it carries length arrays, the stubs do not
      for slot in (4, 5, 6, 7):
        bt[slot] = np.full(k1 - k0, 3 + i + (rank if lie else 0), dtype=np.int64)
      bt[0] = (a[k0:k1], b[k0:k1])
    batches.append(tuple(bt))
  if lie:          # every rank has "all lengths" but sees different ones: the deals differ
    batches = [tuple(np.full(len(bt[8]), 5 + 7 * ((i + rank) % 3), dtype=np.int64)
                     if slot in (4, 5, 6, 7) else bt[slot] for slot in range(12))
               for i, bt in enumerate(batches)]

  def encode_fn(opt, model, mine):
    if not mine:
      return None
    return (torch.from_numpy(np.concatenate([m[0][0] for m in mine])),
            torch.from_numpy(np.concatenate([m[0][1] for m in mine])))

  def rank_fn(q, g, row0, nrows):
    d = q[row0:row0 + nrows].numpy().astype(np.float64) @ g.numpy().astype(np.float64).T
    diag = d[np.arange(nrows), row0 + np.arange(nrows)][:, None]
    return (torch.from_numpy((d > diag).sum(1).astype(np.int32)),
            torch.from_numpy(d.argmax(1).astype(np.int32)))

  try:
    res = parallel_eval.validate_sharded(None, None, batches, encode_fn=encode_fn,
                                         rank_fn=rank_fn, device='cpu', dim=16)
    np.savez(os.path.join(out_dir, 's%d.npz' % rank), ranks_i=res[2], ranks_t=res[3])
  except RuntimeError as e:
    open(os.path.join(out_dir, 'err%d.txt' % rank), 'w').write(str(e))
  dist.destroy_process_group()


def test_default_deal_with_per_rank_materialised_loaders(tmp_path, oracle):
  """a loader that materialises only the rank's own batches hands every rank a
  different mix of stubs; the default deal must still be the same on all ranks."""
  from cmhse_amd import synthetic
  mp.spawn(_stub_worker, args=(2, _free_port(), str(tmp_path), False), nprocs=2, join=True)
  a, b = synthetic.correlated_embeddings(23, 16, 2.0, seed=5)
  _, _, ranks_i = oracle.i2t(a, b, np.float64)
  _, _, ranks_t = oracle.t2i(a, b, np.float64)
  for r in range(2):
    got = np.load(os.path.join(str(tmp_path), 's%d.npz' % r))
    np.testing.assert_array_equal(got['ranks_i'], ranks_i)
    np.testing.assert_array_equal(got['ranks_t'], ranks_t)


def test_diverging_default_deals_raise_instead_of_hanging(tmp_path):
  mp.spawn(_stub_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
  for r in range(2):
    assert 'different batch-to-rank deals' in open(os.path.join(str(tmp_path), 'err%d.txt' % r)).read()


def test_costs_of_prices_by_count_when_any_batch_is_a_stub():
  from cmhse_amd import parallel_eval
  full = [None] * 12
  full[0] = np.zeros((2, 3, 8), np.float32)
  full[8] = (1, 1)
  for slot in (4, 5, 6, 7):
    full[slot] = np.array([3, 2])
  stub = [None] * 12
  stub[8] = (1, 1, 1)
  assert parallel_eval.costs_of([tuple(full), tuple(stub)]) == [(2.0, 0), (3.0, 0)]
  assert parallel_eval.costs_of([tuple(full)])[0][0] > 1e3


def test_shard_range_is_a_partition():
  from cmhse_amd.parallel_eval import shard_range
  for n in [0, 1, 7, 154]:
    for w in [1, 2, 3, 8]:
      pieces = [shard_range(n, r, w) for r in range(w)]
      assert pieces[0][0] == 0 and pieces[-1][1] == n
      assert all(pieces[i][1] == pieces[i + 1][0] for i in range(w - 1))
      sizes = [b - a for a, b in pieces]
      assert max(sizes) - min(sizes) <= 1


def test_assign_batches_balances_work_and_spreads_the_long_chains():
  """Without a tail horizon (a loader that only says how many videos a batch has) the deal is a
  partition, deterministic, balanced by work to within one batch, and puts the batches with the
  longest paragraphs on different ranks (round 5's deal)."""
  from cmhse_amd import parallel_eval, synthetic
  spec = synthetic.anet_like_spec(4917, seed=0)
  costs = [parallel_eval.batch_cost(lc, lv, lw, lp, 2048)
           for lc, lv, lw, lp in synthetic.batch_lengths(spec, 32)]
  assert len(costs) == 154
  for world in (1, 2, 3, 4, 8):
    a = parallel_eval.assign_batches(costs, world)
    assert a == parallel_eval.assign_batches(list(costs), world)
    flat = sorted(i for r in a for i in r)
    assert flat == list(range(len(costs)))
    load = [sum(costs[i][0] for i in r) for r in a]
    assert max(load) - min(load) <= max(c[0] for c in costs) + 1e-6
    top = sorted(range(len(costs)), key=lambda i: -costs[i][1])[:world]
    owners = {r for r in range(world) for i in a[r] if i in top}
    assert len(owners) == world


def test_tail_aware_deal_gathers_the_long_paragraphs_and_lowers_the_slowest_rank():
  """With the split's tail horizon (round 6) a rank is priced work + TAIL_STEP_FLOPS x the words its
  longest paragraph has beyond the horizon: the deal stays a deterministic partition, the outlier
  paragraphs (309-435 words on this split; a typical batch's longest: 140) share ONE rank, which is
  dealt less work, and the modelled finish time of the slowest rank drops by ~6 % at 8 ranks against
  the work-only deal (measured on the GPU, every rank's share: profiles/r06_rank_share.txt)."""
  from cmhse_amd import parallel_eval, synthetic
  spec = synthetic.anet_like_spec(4917, seed=0)
  lens = synthetic.batch_lengths(spec, 32)
  costs = [parallel_eval.batch_cost(lc, lv, lw, lp, 2048) for lc, lv, lw, lp in lens]
  horizon = parallel_eval.tail_horizon(np.concatenate([np.concatenate([lw, lp]) for _, _, lw, lp in lens]))
  assert 40 <= horizon <= 120                       # (72 on this split: where 22 k text sequences drop to 1024 active)
  K = parallel_eval.TAIL_STEP_FLOPS
  finish = lambda deal: [sum(costs[i][0] for i in r) + K * max(0, max(costs[i][1] for i in r) - horizon) for r in deal]
  for world in (2, 4, 8):
    new = parallel_eval.assign_batches(costs, world, tail_from=horizon)
    assert new == parallel_eval.assign_batches(list(costs), world, tail_from=horizon)        # (memoised or not)
    assert sorted(i for r in new for i in r) == list(range(len(costs)))
    old = parallel_eval.assign_batches(costs, world)
    assert max(finish(new)) < max(finish(old))
    longest = [max(costs[i][1] for i in r) for r in new]
    assert longest == sorted(longest, reverse=True)
    outliers = [i for i in range(len(costs)) if costs[i][1] >= 300]
    assert len(outliers) == 5 and all(i in new[0] for i in outliers)
    if world == 8:
      assert max(finish(new)) < 0.95 * max(finish(old))
      assert sum(costs[i][0] for i in new[0]) < 0.9 * max(sum(costs[i][0] for i in r) for r in new[1:])
  # one rank, or no horizon: the old deal
  assert parallel_eval.assign_batches(costs, 1, tail_from=horizon) == [list(range(len(costs)))]
  # a split whose text tower never leaves the tiled regime pays no tail: the deal is balanced by work
  short = [(c[0], 10) for c in costs]
  d = parallel_eval.assign_batches(short, 4, tail_from=horizon)
  load = [sum(short[i][0] for i in r) for r in d]
  assert max(load) - min(load) <= max(c[0] for c in short) + 1e-6


def test_batch_lengths_match_the_materialised_batches():
  from cmhse_amd import synthetic
  spec = synthetic.ragged_spec(11, seed=3)
  lens = synthetic.batch_lengths(spec, 4)
  batches = synthetic.make_batches(spec, 4, 8, 50, seed=0)
  assert len(lens) == len(batches)
  for (lc, lv, lw, lp), b in zip(lens, batches):
    assert list(lc) == b[4].tolist() and list(lw) == b[5].tolist()
    assert list(lv) == b[6].tolist() and list(lp) == b[7].tolist()


def _plan_worker(rank, world, port, out_dir):
  sys.path.insert(0, REPO)
  from cmhse_amd import evaluation, parallel_eval, synthetic
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  spec = synthetic.ragged_spec(41, seed=9, max_clips=6, max_frames=17, max_words=11, max_video=23)
  batches = synthetic.make_batches(spec, 4, 8, 50, seed=0)
  own = [i for i in range(len(batches)) if (i * 5 + 1) % world == rank]
  if rank == world - 1:
    own = []                       # a rank with no batches at all still takes part
  # this rank's loader: its own batches materialised, stubs (num_clips only) for the others
  mine = []
  for i, b in enumerate(batches):
    if i in own:
      mine.append(b)
    else:
      stub = [None] * 12
      stub[8] = b[8]
      mine.append(tuple(stub))
  plan = parallel_eval.global_step_plan([mine[i] for i in own], None, 'cpu')
  np.savez(os.path.join(out_dir, 'p%d.npz' % rank), **plan)
  dist.destroy_process_group()


def test_global_step_plan_is_the_whole_splits_on_every_rank(tmp_path):
  """parallel_eval.global_step_plan: ranks that hold different batches (and one that holds none)
  agree on evaluation.split_step_plan of the union of the ranks' batches — the per-time-step active
  counts of the four encoders from which every rank picks its kernels."""
  from cmhse_amd import evaluation, synthetic
  world = 3
  mp.spawn(_plan_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  spec = synthetic.ragged_spec(41, seed=9, max_clips=6, max_frames=17, max_words=11, max_video=23)
  batches = synthetic.make_batches(spec, 4, 8, 50, seed=0)
  held = [b for i, b in enumerate(batches) if (i * 5 + 1) % world != world - 1]
  want = evaluation.split_step_plan(held)
  for r in range(world):
    got = np.load(os.path.join(str(tmp_path), 'p%d.npz' % r))
    assert sorted(got.files) == sorted(evaluation.TOWERS)
    for k in evaluation.TOWERS:
      np.testing.assert_array_equal(got[k], want[k])


def test_split_step_plan_counts_active_sequences_per_step():
  from cmhse_amd import evaluation, ops, synthetic
  spec = synthetic.ragged_spec(13, seed=2)
  batches = synthetic.make_batches(spec, 5, 8, 50, seed=0)
  plan = evaluation.split_step_plan(batches)
  v1 = np.concatenate([np.asarray(b[4]) for b in batches] + [np.asarray(b[6]) for b in batches])
  t1 = np.concatenate([np.asarray(b[5]) for b in batches] + [np.asarray(b[7]) for b in batches])
  v2 = np.concatenate([np.asarray(b[8]) for b in batches])
  for k, lens in (('v1', v1), ('t1', t1), ('v2', v2), ('t2', v2)):
    want = [int((lens > t).sum()) for t in range(int(lens.max()))]
    assert plan[k].tolist() == want
    assert ops.step_counts(lens).tolist() == want
    # ... exactly what a schedule over all of them hands to the library as its own counts
  # histograms add over any cut of the split
  a = evaluation.length_histograms(batches[:1])
  b = evaluation.length_histograms(batches[1:])
  for k in evaluation.TOWERS:
    n = max(len(a[k]), len(b[k]))
    tot = np.zeros(n, dtype=np.int64)
    tot[:len(a[k])] += a[k]
    tot[:len(b[k])] += b[k]
    np.testing.assert_array_equal(tot, evaluation.length_histograms(batches)[k])


# ------------------------------------------------------------------------------------------------
# RCCL first contact: `bench.py --gpus 8` runs for the first time on a box the
# builder never sees.  Everything of that run that is not a kernel or an RCCL call, at the REAL
# sizes: the rank environments, the by-work deal of the 154 loader batches of N = 4917, the step
# plan all ranks must agree on, the padded all-gather shapes, the stripes, and the merge back to
# loader order — an 8-rank gloo world over length-only batches.
# ------------------------------------------------------------------------------------------------
def test_rank_environments_of_a_bare_multi_gpu_launch():
  import bench
  base = {'PATH': '/usr/bin', 'CMHSE_BENCH_BACKEND': 'stale'}
  envs = bench.rank_environments(8, 8, 29511, base)
  assert [e['RANK'] for e in envs] == [str(r) for r in range(8)]
  assert [e['LOCAL_RANK'] for e in envs] == [str(r) for r in range(8)]      # one GPU per rank
  for e in envs:
    assert e['WORLD_SIZE'] == '8' and e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '29511'
    assert e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'          # dmabuf IPC: RCCL fails without it here
    assert 'CMHSE_BENCH_BACKEND' not in e                  # -> nccl (= RCCL)
  shared = bench.rank_environments(8, 1, 1, {'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
  assert all(e['CMHSE_BENCH_BACKEND'] == 'gloo' and e['LOCAL_RANK'] == '0' for e in shared)
  assert base['CMHSE_BENCH_BACKEND'] == 'stale'           # the caller's environment is not edited


N_FULL, B_FULL, D_FAKE = 4917, 32, 24


def _full_split_lengths():
  from cmhse_amd import parallel_eval, synthetic
  spec = synthetic.anet_like_spec(N_FULL, seed=0, dataset='anet')
  lens = synthetic.batch_lengths(spec, B_FULL)
  costs = [parallel_eval.batch_cost(lc, lv, lw, lp, 2048, 300, 1024) for lc, lv, lw, lp in lens]
  return spec, lens, costs


def _first_contact_worker(rank, world, port, out_dir):
  sys.path.insert(0, REPO)
  from cmhse_amd import parallel_eval, synthetic
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.set_num_threads(1)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  spec, lens, costs = _full_split_lengths()
  horizon = parallel_eval.tail_horizon(np.concatenate([np.concatenate([lw, lp]) for _, _, lw, lp in lens]))
  assignment = parallel_eval.assign_batches(costs, world, tail_from=horizon)      # bench.py's deal
  own = set(assignment[rank])
  a, b = synthetic.correlated_embeddings(N_FULL, D_FAKE, 2.0, seed=3)
  # the rank's loader as bench_common.build_loader makes it: own batches carry their length members
  # (features stay absent: length-only), the others are stubs with num_clips alone
  batches, clip_pos = [], 0
  for bi, b0 in enumerate(range(0, N_FULL, B_FULL)):
    b1 = min(N_FULL, b0 + B_FULL)
    nclips = tuple(spec.num_clips[b0:b1])
    bt = [None] * 12
    bt[8] = nclips
    if bi in own:
      lc, lv, lw, lp = lens[bi]
      bt[4], bt[5], bt[6], bt[7] = (np.asarray(lc), np.asarray(lw), np.asarray(lv), np.asarray(lp))
      bt[9] = nclips
      bt[0] = (b0, b1)
    batches.append(tuple(bt))
    clip_pos += sum(nclips)
  seen = {}

  def encode_fn(opt, model, mine):
    rows = np.concatenate([np.arange(m[0][0], m[0][1]) for m in mine])
    seen['rows'] = len(rows)
    return torch.from_numpy(a[rows]), torch.from_numpy(b[rows])

  def rank_fn(q, g, row0, nrows):
    seen.setdefault('stripes', []).append((int(q.shape[0]), int(g.shape[0]), int(row0), int(nrows)))
    d = q[row0:row0 + nrows].numpy().astype(np.float64) @ g.numpy().astype(np.float64).T
    diag = d[np.arange(nrows), row0 + np.arange(nrows)][:, None]
    return (torch.from_numpy((d > diag).sum(1).astype(np.int32)),
            torch.from_numpy(d.argmax(1).astype(np.int32)))

  plan = parallel_eval.global_step_plan([batches[i] for i in assignment[rank]], None, 'cpu')
  res = parallel_eval.validate_sharded(None, None, batches, encode_fn=encode_fn, rank_fn=rank_fn,
                                       device='cpu', dim=D_FAKE, assignment=assignment)
  np.savez(os.path.join(out_dir, 'fc%d.npz' % rank), ranks_i=res[2], ranks_t=res[3], top1_i=res[4],
           top1_t=res[5], rows=seen['rows'], stripes=np.array(seen['stripes']),
           **{'plan_' + k: v for k, v in plan.items()})
  dist.destroy_process_group()


def test_eight_rank_first_contact_at_the_real_split_sizes(tmp_path, oracle):
  """8 gloo ranks over the N = 4917 split's LENGTHS (no features, fake embeddings keyed by the
  global video index): every rank derives bench.py's deal (work + the tail of the rank's longest paragraph), holds a 480-704 video share,
  agrees on the whole split's step plan, all-gathers shards padded to the largest, scores its own
  [row0, row0 + nrows) stripe of the full 4917-row gallery, and the merged ranks / top-1 come out
  in loader order equal to the single-process oracle's."""
  from cmhse_amd import evaluation, parallel_eval, synthetic
  world = 8
  mp.spawn(_first_contact_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  spec, lens, costs = _full_split_lengths()
  horizon = parallel_eval.tail_horizon(np.concatenate([np.concatenate([lw, lp]) for _, _, lw, lp in lens]))
  assignment = parallel_eval.assign_batches(costs, world, tail_from=horizon)
  sizes = [min(N_FULL, b0 + B_FULL) - b0 for b0 in range(0, N_FULL, B_FULL)]
  counts = [sum(sizes[i] for i in assignment[r]) for r in range(world)]
  assert sum(counts) == N_FULL and min(counts) >= 480 and max(counts) <= 704, counts
  work = [sum(costs[i][0] for i in assignment[r]) for r in range(world)]
  assert max(work[1:]) / min(work[1:]) < 1.06         # the deal balances GRU work (to one batch), not video counts
  assert work[0] < min(work[1:])                      # ... and the rank that pays for the longest paragraphs gets less of it
  # the whole split's plan from its lengths alone
  full = []
  for bi, (lc, lv, lw, lp) in enumerate(lens):
    bt = [None] * 12
    bt[4], bt[5], bt[6], bt[7] = np.asarray(lc), np.asarray(lw), np.asarray(lv), np.asarray(lp)
    b0 = bi * B_FULL
    bt[8] = bt[9] = tuple(spec.num_clips[b0:min(N_FULL, b0 + B_FULL)])
    full.append(tuple(bt))
  want_plan = evaluation.split_step_plan(full)
  a, b = synthetic.correlated_embeddings(N_FULL, D_FAKE, 2.0, seed=3)
  _, top1_i, ranks_i = oracle.i2t(a, b, np.float64)
  _, top1_t, ranks_t = oracle.t2i(a, b, np.float64)
  for r in range(world):
    got = np.load(os.path.join(str(tmp_path), 'fc%d.npz' % r))
    assert int(got['rows']) == counts[r]
    row0 = sum(counts[:r])
    assert got['stripes'].tolist() == [[N_FULL, N_FULL, row0, counts[r]]] * 2
    for k in evaluation.TOWERS:
      np.testing.assert_array_equal(got['plan_' + k], want_plan[k])
    np.testing.assert_array_equal(got['ranks_i'], ranks_i)
    np.testing.assert_array_equal(got['ranks_t'], ranks_t)
    np.testing.assert_array_equal(got['top1_i'], top1_i)
    np.testing.assert_array_equal(got['top1_t'], top1_t)
