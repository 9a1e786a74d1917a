"""Multi-GPU validation: the val-set all-pairs scoring row-sharded over ranks.

The reference is single-GPU (train.py:101) and has no collective anywhere; this is the one
data-parallel sharding the path offers: videos/paragraphs are independent units.

  1. the loader's batches are dealt to ranks by WORK, not by count (`assign_batches`): the cost of a
     batch is its frame steps and word steps priced by the GRU step they feed, plus — per RANK — the
     few-sequence tail its longest paragraph brings (which does not shrink with the world size): the
     batches with the few very long paragraphs share a rank that is dealt less work, the deal
     minimises the slowest rank; rank r encodes its batches with replicated weights;
  2. ONE exchange step: all-gather of the L2-normalised [n_r, D] video and paragraph embeddings
     (RCCL over xGMI under backend "nccl"; `all_gather_into_tensor`, shards padded to the largest);
  3. rank r scores its own ROW STRIPE: V[stripe] . P^T for i2t and P[stripe] . V^T for t2i with the
     fused rank/top-1 epilogue — every row's diagonal lies inside its stripe's columns, so no
     further data-path communication is needed;
  4. all-gather of the int32 ranks / top-1 (a few KB), un-permuted to loader order, and the
     Recall@K report on every rank, computed exactly as evaluation.py:173-184.

Ranks do not depend on the partition: each score row is computed independently, a common
permutation of videos and paragraphs keeps every diagonal pair together, and every rank encodes
its share with the kernel kinds the WHOLE split selects per time step (`global_step_plan`: a
share's own, smaller active counts would otherwise move some steps from the LDS-tiled kernel to the
small-batch one, whose sums are ordered differently — embeddings equal to fp32 rounding only, a
handful of the 9834 rank rows flipped at N = 4917 in round 4).  So the result is identical, bit for
bit, for any world size.  `encode_fn` / `rank_fn` exist so the partition/merge logic can be exercised on CPU
with gloo in tests; the defaults are the HIP path and have no fallback.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.distributed as dist

from . import evaluation, ops


def shard_range(n_items, rank, world):
  """Contiguous, balanced slice [lo, hi) of n_items for `rank` (first ranks get the remainder)."""
  base, rem = divmod(n_items, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def batch_cost(len_clip, len_vid, len_cap, len_par, img_dim, word_dim=300, hidden=1024):
  """(work, chain) of one loader batch: work = GRU FLOPs of its level-1 steps (SURVEY §8d:
  2*3H*(I+H) per sequence-step, frames at I = img_dim, words at I = word_dim); chain = its longest
  paragraph, the length of the dependent tail it brings to whichever rank encodes it."""
  fv = 6.0 * hidden * (img_dim + hidden)
  ft = 6.0 * hidden * (word_dim + hidden)
  work = fv * (float(np.sum(len_clip)) + float(np.sum(len_vid))) + \
      ft * (float(np.sum(len_cap)) + float(np.sum(len_par)))
  return work, int(np.max(len_par)) if len(len_par) else 0


# What ONE step of the text tower's few-sequence tail costs a rank, expressed in GRU FLOPs of its step
# chain.  Fitted to the measured shares of all ranks of an 8- and a 4-rank deal of the ICEP split
# (24 points, profiles/r06_rank_share.txt): ms = 8.18 x TFLOP + 0.0105 x tail steps + 4.2, i.e. one
# tail step (10.5 us: a dependent small-batch launch, partly hidden under the attention pass) is worth
# 1.3 GFLOP of chain work at the 122 TFLOP/s a share's chain runs at.  It prices the tail a rank pays
# for its longest paragraph against the work it is dealt.
TAIL_STEP_FLOPS = 1.3e9


def tail_horizon(text_lens, threshold=1024):
  """The time step from which the text tower of the WHOLE split runs on the small-batch kernels:
  the number of leading steps with more than `threshold` (tiny_max_seqs) sequences still active over
  all sentence and paragraph lengths.  A rank's tail is what its longest paragraph has beyond it."""
  text_lens = np.asarray(text_lens, dtype=np.int64).reshape(-1)
  if text_lens.size == 0:
    return 0
  return int((ops.step_counts(text_lens) > threshold).sum())


def _lpt(costs, world):
  order = sorted(range(len(costs)), key=lambda i: (-costs[i][1], -costs[i][0], i))
  load = [0.0] * world
  out = [[] for _ in range(world)]
  for i in order:
    r = min(range(world), key=lambda q: (load[q], q))
    out[r].append(i)
    load[r] += max(costs[i][0], 1e-9)
  return [sorted(o) for o in out]


def assign_batches(costs, world, tail_from=None, tail_flops=TAIL_STEP_FLOPS):
  """Deal batches to ranks.  `costs` = [(work, chain)] per batch: GRU FLOPs, longest paragraph.
  Returns `world` ascending batch-index lists; deterministic, so every rank computes the same one.

  A rank's pass is its step chain (its WORK at the matrix pipe's rate) followed by the text tower's
  tail: one dependent small-batch launch per word its LONGEST paragraph has beyond `tail_from` (the
  step from which the whole split's text tower is small-batch, tail_horizon()) — a cost per rank,
  not per batch, that does not shrink with the world size (ActivityNet val: the longest paragraphs
  have 309-435 words, a typical batch's longest 140; 363 tail steps are 7.7 of a 608-video share's
  40 ms).  With `tail_from` the deal minimises max over ranks of work + tail_flops * tail steps:
  the few batches with very long paragraphs go to the SAME rank, which is dealt that much less work,
  instead of one to every rank (round 6; 8 ranks: 40.3 -> 36-37 ms per share,
  profiles/r06_rank_share.txt).  Without it (a loader that only says how many videos a batch has):
  longest chain first onto the least-loaded rank, as before.  Integer ranks do not depend on the
  deal (step plan)."""
  n = len(costs)
  if tail_from is None or world <= 1 or n == 0:
    return _lpt(costs, world)
  key = (world, int(tail_from), float(tail_flops), tuple(costs))
  if key in _DEALS:                      # (a validation loop deals the same split every epoch)
    return [list(r) for r in _DEALS[key]]
  out = _tail_aware_deal(costs, world, tail_from, tail_flops)
  if len(_DEALS) >= 8:
    _DEALS.pop(next(iter(_DEALS)))
  _DEALS[key] = [tuple(r) for r in out]
  return out


_DEALS = {}


def _tail_aware_deal(costs, world, tail_from, tail_flops):
  """min over deals of max over ranks of (work + tail_flops x tail steps of the rank's longest
  paragraph), approximately: first-fit of the batches in order of their longest paragraph (so a rank's
  FIRST batch fixes the tail it pays, every later batch fits under it for free, and the outliers end
  up together on the first rank) into ranks of capacity T, the smallest T that needs no more than
  `world` ranks found by bisection."""
  n = len(costs)
  tail = lambda chain: tail_flops * max(0, int(chain) - int(tail_from))
  order = sorted(range(n), key=lambda i: (-costs[i][1], -costs[i][0], i))
  total = float(sum(c[0] for c in costs))

  def first_fit(limit):
    load, out = [], []
    for i in order:
      w, chain = costs[i]
      for r in range(len(load)):
        if load[r] + w <= limit:
          load[r] += w
          out[r].append(i)
          break
      else:
        if len(load) >= world or tail(chain) + w > limit:
          return None
        load.append(tail(chain) + w)
        out.append([i])
    return out + [[] for _ in range(world - len(out))]

  lo = total / world
  hi = total + tail(costs[order[0]][1])
  for _ in range(32):
    mid = 0.5 * (lo + hi)
    if first_fit(mid) is None:
      lo = mid
    else:
      hi = mid
  return [sorted(r) for r in first_fit(hi)]


def costs_of(batches, img_dim=None):
  """Per-batch (work, chain) from the loader's own length tensors (slots 4-7 of the 12-tuple).
  If ANY batch carries none (a stub that only says how many videos it has — a loader that
  materialises only the rank's own batches), EVERY batch is priced by its video count: the deal
  must be a function of what all ranks see alike, and a stub is all some ranks know of a batch."""
  if any(b[4] is None or b[7] is None for b in batches):
    return [(float(len(b[8])), 0) for b in batches]
  out = []
  for b in batches:
    I = img_dim if img_dim is not None else (int(b[0].shape[2]) if hasattr(b[0], 'shape') else 1024)
    out.append(batch_cost(np.asarray(b[4]), np.asarray(b[6]), np.asarray(b[5]), np.asarray(b[7]), I))
  return out


def tail_horizon_of(batches):
  """tail_horizon() of a loader's batches from their sentence / paragraph length members, or None
  when any batch carries none (a stub: the deal must be a function of what all ranks see alike)."""
  if any(b[5] is None or b[7] is None for b in batches):
    return None
  return tail_horizon(np.concatenate([np.asarray(b[5], dtype=np.int64).reshape(-1) for b in batches] +
                                     [np.asarray(b[7], dtype=np.int64).reshape(-1) for b in batches]))


def _default_encode(opt, model, batches, plan=None, step_plan=None):
  """(video embeddings, paragraph embeddings, finish): `finish()` replays the per-batch 'Letest'
  meters (evaluation.py:129); validate_sharded calls it after the exchange and the ranking kernels
  are queued, so no host sync stands in front of them."""
  if not batches:
    return None
  cat, _, _, finish = evaluation.encode_data_device(opt, model, batches, logging=lambda *a: None,
                                                    defer_logging=True, plan=plan, step_plan=step_plan)
  return cat['vid_emb'], cat['para_emb'], finish


def global_step_plan(own_batches, group=None, device=None):
  """evaluation.split_step_plan of the WHOLE split, agreed on by all ranks, from each rank's own
  batches: the length histograms of the four encoders add over the ranks (two small control
  collectives: the longest length per encoder, then the histograms — a few hundred integers).  A
  rank never needs the lengths of batches it does not encode, so loaders that materialise only
  their own batches work.  With it every rank picks, for every time step, the kernel the single
  process would pick for the whole split, and a sequence's embedding is the same bit pattern on
  any number of ranks."""
  hists = evaluation.length_histograms(own_batches)
  if dist.get_world_size(group) > 1:
    dev = _collective_device(group, device)
    sizes = torch.tensor([len(hists[k]) for k in evaluation.TOWERS], dtype=torch.int64, device=dev)
    dist.all_reduce(sizes, op=dist.ReduceOp.MAX, group=group)
    sizes = [int(v) for v in sizes.cpu().tolist()]
    flat = np.zeros(sum(sizes), dtype=np.int64)
    pos = 0
    for k, n in zip(evaluation.TOWERS, sizes):
      flat[pos:pos + len(hists[k])] = hists[k]
      pos += n
    t = torch.from_numpy(flat).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    flat = t.cpu().numpy()
    pos = 0
    for k, n in zip(evaluation.TOWERS, sizes):
      hists[k] = flat[pos:pos + n].copy()
      pos += n
  return evaluation.plan_from_histograms(hists)


def _default_rank(queries, gallery, row0, nrows):
  return ops.sim_rank(queries, gallery, row0, nrows)


def all_gather_rows(local, counts, group=None):
  """All-gather row blocks of unequal height into one tensor: pad to max(counts), ONE
  all_gather_into_tensor, drop the padding."""
  world = dist.get_world_size(group)
  width = tuple(local.shape[1:])
  mx = max(max(counts), 1)
  # (a gloo group moves device tensors through the host: debugging runs of the N-rank path on a
  # box with fewer GPUs than ranks; RCCL gathers in place)
  via_host = local.is_cuda and dist.get_backend(group) == 'gloo'
  dev = torch.device('cpu') if via_host else local.device
  padded = torch.zeros((mx,) + width, dtype=local.dtype, device=dev)
  padded[:local.shape[0]] = local
  out = torch.empty((world * mx,) + width, dtype=local.dtype, device=dev)
  dist.all_gather_into_tensor(out, padded, group=group)
  if via_host:
    out = out.to(local.device)
  if all(c == mx for c in counts):
    return out
  return torch.cat([out[r * mx:r * mx + c] for r, c in enumerate(counts)], 0)


def all_gather_pair(a, b, counts, group=None):
  """The exchange step: rows of `a` and `b` (same shape [n_r, D]) of every rank in ONE collective —
  each rank contributes [max(counts), 2, D] (its a rows and b rows side by side, zero padded) —
  returned as two contiguous [sum(counts), D] matrices, rank after rank."""
  world = dist.get_world_size(group)
  D = int(a.shape[1])
  mx = max(max(counts), 1)
  via_host = a.is_cuda and dist.get_backend(group) == 'gloo'
  dev = torch.device('cpu') if via_host else a.device
  n = int(a.shape[0])
  padded = torch.empty((mx, 2, D), dtype=a.dtype, device=dev)
  padded[:n, 0] = a
  padded[:n, 1] = b
  if n < mx:
    padded[n:].zero_()
  out = torch.empty((world * mx, 2, D), dtype=a.dtype, device=dev)
  dist.all_gather_into_tensor(out, padded, group=group)
  if via_host:
    out = out.to(a.device)
  keep = [out[r * mx:r * mx + c] for r, c in enumerate(counts) if c > 0]
  if not keep:
    z = torch.zeros(0, D, dtype=a.dtype, device=a.device)
    return z, z.clone()
  # the split into two row-major matrices is the one copy (it also drops the padding)
  A = torch.cat([k[:, 0] for k in keep], 0)
  B = torch.cat([k[:, 1] for k in keep], 0)
  return A, B


def _collective_device(group, device):
  """Where a small control tensor of a collective must live: a gloo group reduces CPU tensors; any
  other backend (nccl = RCCL) only device tensors — on `device`, or, when the caller gave none
  (validate_sharded(opt, model, val_loader) under backend='nccl'), on the
  calling thread's current GPU (a CPU tensor there raises "No backend type associated with device
  type cpu")."""
  if dist.get_backend(group) == 'gloo':
    return torch.device('cpu')
  if device is not None and torch.device(device).type == 'cuda':
    return torch.device(device)
  return torch.device('cuda', torch.cuda.current_device())


def _same_on_all_ranks(assignment, group, device):
  """Raise on every rank if the ranks derived different deals (a collective entered with
  mismatched shapes would hang instead)."""
  import zlib
  h = zlib.crc32(repr(assignment).encode()) & 0x7fffffff
  t = torch.tensor([h, -h], dtype=torch.int64, device=_collective_device(group, device))
  dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
  if int(t[0]) != h or int(t[1]) != -h:
    raise RuntimeError('validate_sharded: the ranks derived different batch-to-rank deals from '
                       'their loaders; pass the same `assignment` on every rank')


_PHASE_EVENTS = {}    # device index -> timing events ready for reuse (creating one while the GPU
                      # is busy can stall the host for tens of ms, see ops.upload)


class _Phases(object):
  """Phase marks of one sharded pass.  On a GPU they are HIP events on the current stream, read
  after the pass's own final device-to-host copy — the timed pass is never drained for them."""

  def __init__(self, use_events):
    self.use_events, self.marks = use_events, []
    self.pool = _PHASE_EVENTS.setdefault(torch.cuda.current_device(), []) if use_events else None

  def mark(self):
    if self.use_events:
      ev = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
      ev.record()
      self.marks.append(ev)
    else:
      self.marks.append(time.perf_counter())

  def spans_ms(self):
    m = self.marks
    if self.use_events:
      m[-1].synchronize()
      out = [m[i].elapsed_time(m[i + 1]) for i in range(len(m) - 1)]
      self.pool.extend(m)
      return out
    return [(m[i + 1] - m[i]) * 1e3 for i in range(len(m) - 1)]


def validate_sharded(opt, model, data_loader, group=None, encode_fn=None, rank_fn=None,
                     device=None, dim=None, assignment=None, timings=None, plan=None, step_plan=None):
  """Sharded counterpart of train.validate's encode_data + i2t + t2i (train.py:223-236).
  Returns (report_i2t, report_t2i, ranks_i2t, ranks_t2i, top1_i2t, top1_t2i) on every rank, rows in
  loader order.  `assignment`: per-rank batch-index lists (default: assign_batches on the loader's
  lengths, checked to agree across ranks).  `encode_fn(opt, model, batches)` returns None (no
  batches), (V, P) or (V, P, finish) — `finish()` is called once the ranking is queued.
  `timings`: a dict that receives this rank's encode_ms / exchange_ms / score_ms — on a GPU from
  HIP events on the current stream (device time between the marks; the pass is not synchronised
  for them) — measurement only.  `plan`: a dict the caller keeps between passes over the SAME
  resident batches (evaluation.encode_group): this rank's schedules are built once.  `step_plan`:
  evaluation.split_step_plan of the whole split if the caller has it; by default the ranks agree on
  it themselves (global_step_plan) — it is what makes the integer ranks independent of the world
  size bit for bit (each rank encodes its share with the kernels the whole split selects)."""
  rank_fn = rank_fn or _default_rank
  world = dist.get_world_size(group)
  me = dist.get_rank(group)
  batches = list(data_loader)
  if assignment is None:
    assignment = assign_batches(costs_of(batches), world, tail_from=tail_horizon_of(batches))
    _same_on_all_ranks(assignment, group, device)
  if encode_fn is None:
    if step_plan is None:
      step_plan = global_step_plan([batches[i] for i in assignment[me]], group, device)
    encode_fn = lambda o, m, b: _default_encode(o, m, b, plan, step_plan)
  sizes = [len(b[8]) for b in batches]
  starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
  # global (loader-order) video index of every row of the gathered matrices, rank after rank
  perm = np.concatenate([np.arange(starts[i], starts[i + 1]) for r in range(world)
                         for i in assignment[r]] or [np.zeros(0, dtype=np.int64)]).astype(np.int64)
  counts = [int(sum(sizes[i] for i in assignment[r])) for r in range(world)]
  on_gpu = torch.cuda.is_available() and device is not None and torch.device(device).type == 'cuda'
  ph = _Phases(on_gpu) if timings is not None else None

  finish, ok = None, False
  try:
    if ph:
      ph.mark()
    enc = encode_fn(opt, model, [batches[i] for i in assignment[me]])
    if enc is None:
      if device is None and dist.get_backend(group) != 'gloo' and torch.cuda.is_available():
        device = torch.device('cuda', torch.cuda.current_device())
      if device is None or dim is None:
        raise ValueError('a rank with an empty shard needs `dim` (and, on a gloo group, `device`)')
      v_loc = torch.zeros(0, dim, dtype=torch.float32, device=device)
      p_loc = torch.zeros(0, dim, dtype=torch.float32, device=device)
    else:
      v_loc, p_loc = enc[0], enc[1]
      finish = enc[2] if len(enc) > 2 else None
    if ph:
      ph.mark()
    V, P = all_gather_pair(v_loc, p_loc, counts, group)
    if ph:
      ph.mark()
    row0 = sum(counts[:me])
    nrows = counts[me]
    r_i, t_i = rank_fn(V, P, row0, nrows)
    r_t, t_t = rank_fn(P, V, row0, nrows)
    packed = torch.stack([r_i, t_i, r_t, t_t], 1).to(torch.int32)
    if ph:
      ph.mark()
    full = all_gather_rows(packed, counts, group).cpu().numpy()
    ok = True
  finally:
    # the deferred 'Letest' replay belongs to THIS call: run it, or drop it on an error path (a
    # failed pass must not leave a closure behind that pins its loader batches and would be
    # replayed into the next call)
    fin, finish = finish, None
    if fin is not None and ok:
      fin()
  if ph:
    enc_ms, exch_ms, score_ms = ph.spans_ms()
    timings.update(encode_ms=enc_ms, exchange_ms=exch_ms, score_ms=score_ms, videos=nrows)
  # rows are in rank-major (permuted) order: put them back in loader order; a top-1 is an index
  # into the permuted gallery and maps through the same permutation
  n = len(perm)
  ranks_i, ranks_t = np.empty(n, np.float64), np.empty(n, np.float64)
  top1_i, top1_t = np.empty(n, np.float64), np.empty(n, np.float64)
  ranks_i[perm], ranks_t[perm] = full[:, 0], full[:, 2]
  top1_i[perm], top1_t[perm] = perm[full[:, 1]], perm[full[:, 3]]
  return (evaluation.report_from_ranks(ranks_i), evaluation.report_from_ranks(ranks_t),
          ranks_i, ranks_t, top1_i, top1_t)

