"""Multi-GPU validation: the val-set all-pairs scoring row-sharded over ranks.

The reference is single-GPU (train.py:101) and has no collective anywhere; this is the one
data-parallel sharding the path offers: videos/paragraphs are independent units.

  1. rank r encodes a contiguous slice of the loader's batches (replicated weights);
  2. ONE exchange step: all-gather of the L2-normalised [n_r, D] video and paragraph embeddings
     (RCCL over xGMI under backend "nccl"; shards are padded to the largest n_r);
  3. rank r scores its own ROW STRIPE: V[stripe] . P^T for i2t and P[stripe] . V^T for t2i with the
     fused rank/top-1 epilogue — every row's diagonal lies inside its stripe's columns, so no
     further data-path communication is needed;
  4. all-gather of the int32 ranks / top-1 (a few KB) and the Recall@K report on every rank,
     computed exactly as evaluation.py:173-184.

Ranks do not depend on the partition (each row is computed independently), so the result is
identical for any world size.  `encode_fn` / `rank_fn` exist so the partition/merge logic can be
exercised on CPU with gloo in tests; the defaults are the HIP path and have no fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import evaluation, ops


def shard_range(n_items, rank, world):
  """Contiguous, balanced slice [lo, hi) of n_items for `rank` (first ranks get the remainder)."""
  base, rem = divmod(n_items, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def _default_encode(opt, model, batches):
  if not batches:
    return None
  cat, _, _ = evaluation.encode_data_device(opt, model, batches, logging=lambda *a: None)
  return cat['vid_emb'], cat['para_emb']


def _default_rank(queries, gallery, row0, nrows):
  return ops.sim_rank(queries, gallery, row0, nrows)


def all_gather_rows(local, counts, group=None):
  """All-gather row blocks of unequal height: pad to max(counts), gather, drop the padding."""
  world = dist.get_world_size(group)
  width = local.shape[1:]
  mx = max(counts)
  padded = torch.zeros((mx,) + tuple(width), dtype=local.dtype, device=local.device)
  padded[:local.shape[0]] = local
  bufs = [torch.empty_like(padded) for _ in range(world)]
  dist.all_gather(bufs, padded, group=group)
  return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def validate_sharded(opt, model, data_loader, group=None, encode_fn=None, rank_fn=None,
                     device=None, dim=None):
  """Sharded counterpart of train.validate's encode_data + i2t + t2i (train.py:223-236).
  Returns (report_i2t, report_t2i, ranks_i2t, ranks_t2i, top1_i2t, top1_t2i) on every rank."""
  encode_fn = encode_fn or _default_encode
  rank_fn = rank_fn or _default_rank
  world = dist.get_world_size(group)
  me = dist.get_rank(group)
  batches = list(data_loader)
  lo, hi = shard_range(len(batches), me, world)
  mine = batches[lo:hi]
  # videos per rank follow from the loader alone, so every rank can compute all counts locally
  counts = []
  for r in range(world):
    a, b = shard_range(len(batches), r, world)
    counts.append(sum(len(bb[8]) for bb in batches[a:b]))
  enc = encode_fn(opt, model, mine)
  if enc is None:
    if device is None or dim is None:
      raise ValueError('a rank with an empty shard needs `device` and `dim`')
    v_loc = torch.zeros(0, dim, dtype=torch.float32, device=device)
    p_loc = torch.zeros(0, dim, dtype=torch.float32, device=device)
  else:
    v_loc, p_loc = enc
  V = all_gather_rows(v_loc, counts, group)
  P = all_gather_rows(p_loc, counts, group)
  row0 = sum(counts[:me])
  nrows = counts[me]
  r_i, t_i = rank_fn(V, P, row0, nrows)
  r_t, t_t = rank_fn(P, V, row0, nrows)
  packed = torch.stack([r_i, t_i, r_t, t_t], 1).to(torch.int32)
  full = all_gather_rows(packed, counts, group).cpu().numpy()
  ranks_i, top1_i = full[:, 0].astype(np.float64), full[:, 1].astype(np.float64)
  ranks_t, top1_t = full[:, 2].astype(np.float64), full[:, 3].astype(np.float64)
  return (evaluation.report_from_ranks(ranks_i), evaluation.report_from_ranks(ranks_t),
          ranks_i, ranks_t, top1_i, top1_t)
