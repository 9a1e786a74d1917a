"""Similarity / ranking / loss kernels against the oracle, and the C ABI's argument checks.
(One family of the former tests/test_gpu_parity.py; helpers in tests/gpu_common.py, fixtures in conftest.py.)"""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import EMB_TOL, assert_emb_close, load_golden, golden_state_dicts, golden_batches  # noqa: F401

from gpu_common import *  # noqa: F401,F403,E402
from gpu_common import (_blas_threads, _check_train_step_vs_oracle, _full_opt, _nccl_worker, _np_batches,  # noqa: F401,E402
                        _np_state_dicts, _plan_setup, _RecordForward, _robust_rank_rows)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,m,d', [(1, 1, 8), (129, 300, 64), (515, 515, 1024), (257, 400, 30)])
def test_sim_rank_vs_oracle(dev, oracle, n, m, d):
  """Rectangular and non-tile-multiple shapes; ranks compared exactly on tie-free rows."""
  from cmhse_amd import ops, synthetic
  rng = np.random.RandomState(n + m)
  a = rng.standard_normal((n, d)).astype(np.float32)
  b = rng.standard_normal((m, d)).astype(np.float32)
  k = min(n, m)
  b[:k] += 2.0 * a[:k]
  a /= np.linalg.norm(a, axis=1, keepdims=True)
  b /= np.linalg.norm(b, axis=1, keepdims=True)
  d64 = a.astype(np.float64) @ b.astype(np.float64).T
  diag = d64[np.arange(n), np.arange(n)][:, None]
  gap = np.abs(d64 - diag)
  gap[np.arange(n), np.arange(n)] = 1.0
  ok = gap.min(axis=1) > 1e-5
  srt = np.sort(d64, axis=1)
  ok_top = (srt[:, -1] - srt[:, -2]) > 1e-5 if m > 1 else np.ones(n, bool)
  rank, top1 = ops.sim_rank(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
  want_rank = ((d64 > diag).sum(axis=1) - 0).astype(np.int64)
  np.testing.assert_array_equal(rank.cpu().numpy()[ok], want_rank[ok])
  np.testing.assert_array_equal(top1.cpu().numpy()[ok_top], d64.argmax(axis=1)[ok_top])
  assert ok.mean() > 0.99


def test_sim_rank_stripes_match_full(dev):
  """Row-stripe calls (the multi-GPU sharding unit) give exactly the full-matrix ranks."""
  from cmhse_amd import ops, synthetic
  a, b = synthetic.correlated_embeddings(700, 256, 3.0, seed=4)
  ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
  rank, top1 = ops.sim_rank(ta, tb)
  for parts in (2, 3, 8):
    bounds = np.linspace(0, 700, parts + 1).astype(int)
    rs, ts = [], []
    for p in range(parts):
      r, t = ops.sim_rank(ta, tb, int(bounds[p]), int(bounds[p + 1] - bounds[p]))
      rs.append(r)
      ts.append(t)
    assert torch.equal(torch.cat(rs), rank) and torch.equal(torch.cat(ts), top1)


def test_sim_rank_tie_rule(dev):
  """Documented tie rule: strict '>' for the rank, smallest column for top1."""
  from cmhse_amd import ops
  a = torch.zeros(4, 8, device=dev)
  b = torch.zeros(4, 8, device=dev)
  a[:, 0] = 1.0
  b[:, 0] = 1.0          # every score equals 1.0
  rank, top1 = ops.sim_rank(a, b)
  assert rank.tolist() == [0, 0, 0, 0] and top1.tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize('n', [1, 2, 64, 65, 300])
def test_contrastive_vs_oracle(dev, oracle, n):
  from cmhse_amd import ops
  rng = np.random.RandomState(n)
  a = rng.standard_normal((n, 128)).astype(np.float32)
  b = (a + rng.standard_normal((n, 128))).astype(np.float32)
  a /= np.linalg.norm(a, axis=1, keepdims=True)
  b /= np.linalg.norm(b, axis=1, keepdims=True)
  ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
  for mv in (False, True):
    for nm in (False, True):
      got = ops.contrastive_fwd(ta, tb, 0.2, mv, nm).item()
      want = oracle.contrastive_loss(a, b, 0.2, mv, nm, np.float64)
      assert loss_close(got, want), (n, mv, nm, got, want)


@pytest.mark.parametrize('seed', range(8))
def test_scoring_and_loss_fuzz_vs_oracle(dev, oracle, seed):
  """Seeded random (n, m, d) — single rows, widths of 1 and of non-multiples of 4, sizes around
  the 128-row tile: ranks / top-1 exact on tie-free rows, the stored score matrix, the loss value
  and both loss gradients against the float64 oracle for every (max_violation, norm)."""
  from cmhse_amd import ops
  from cmhse_amd.loss import ContrastiveLoss
  rng = np.random.RandomState(500 + seed)
  n = int(rng.choice([1, 2, 31, 127, 128, 129, 260]))
  m = int(rng.choice([n, n + 1, 2 * n + 3]))
  d = int(rng.choice([1, 3, 8, 30, 64, 257]))
  a = rng.standard_normal((n, d)).astype(np.float32)
  b = rng.standard_normal((m, d)).astype(np.float32)
  b[:n] += 1.5 * a
  a /= np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
  b /= np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-12)
  ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
  d64 = a.astype(np.float64) @ b.astype(np.float64).T
  diag = d64[np.arange(n), np.arange(n)][:, None]
  gap = np.abs(d64 - diag)
  gap[np.arange(n), np.arange(n)] = 1.0
  ok = gap.min(axis=1) > 1e-5
  rank, top1 = ops.sim_rank(ta, tb)
  np.testing.assert_array_equal(rank.cpu().numpy()[ok], (d64 > diag).sum(axis=1)[ok])
  if m > 1:
    srt = np.sort(d64, axis=1)
    ok_top = (srt[:, -1] - srt[:, -2]) > 1e-5
    np.testing.assert_array_equal(top1.cpu().numpy()[ok_top], d64.argmax(axis=1)[ok_top])
  np.testing.assert_allclose(ops.cosine_sim(ta, tb).cpu().numpy(), d64, atol=2e-6, rtol=0)
  bs = b[:n]
  for mv in (False, True):
    for nm in (False, True):
      x = ta.clone().requires_grad_(True)
      y = tb[:n].clone().requires_grad_(True)
      loss = ContrastiveLoss(margin=0.2, measure='cosine', max_violation=mv, norm=nm)(x, y)
      want = oracle.contrastive_loss(a, bs, 0.2, mv, nm, np.float64)
      assert loss_close(loss.item(), want), (n, d, mv, nm, loss.item(), want)
      # hinge kinks / arg-max ties move the subgradient: only compare where the oracle's own
      # float32 and float64 evaluations agree on the active set
      g64 = oracle.contrastive_loss_backward(a, bs, 0.2, mv, nm, np.float64)
      g32 = oracle.contrastive_loss_backward(a, bs, 0.2, mv, nm, np.float32)
      if not (np.allclose(g64[0], g32[0], atol=1e-5) and np.allclose(g64[1], g32[1], atol=1e-5)):
        continue
      loss.backward()
      grad_close(x.grad.cpu().numpy(), g64[0], 'fuzz da n%d d%d mv%d nm%d' % (n, d, mv, nm))
      grad_close(y.grad.cpu().numpy(), g64[1], 'fuzz db n%d d%d mv%d nm%d' % (n, d, mv, nm))


def test_contrastive_blocks_equal_single_calls(dev):
  """The batched per-loader-batch loss equals one cmhse_contrastive_fwd call per block, bitwise."""
  from cmhse_amd import ops
  rng = np.random.RandomState(7)
  sizes = [32, 32, 7, 1, 130, 64]
  n = sum(sizes)
  a = torch.from_numpy(rng.standard_normal((n, 96)).astype(np.float32)).to(dev)
  b = torch.from_numpy(rng.standard_normal((n, 96)).astype(np.float32)).to(dev)
  a, b = ops.l2norm_rows(a), ops.l2norm_rows(b)
  for mv in (False, True):
    for nm in (False, True):
      got = ops.contrastive_blocks_fwd(a, b, sizes, 0.2, mv, nm).cpu().numpy()
      pos = 0
      for i, sz in enumerate(sizes):
        want = ops.contrastive_fwd(a[pos:pos + sz], b[pos:pos + sz], 0.2, mv, nm).item()
        assert got[i] == np.float32(want), (mv, nm, i)
        pos += sz


def test_cpu_tensor_is_rejected_loudly(dev):
  from cmhse_amd import ops
  with pytest.raises(RuntimeError):
    ops.l2norm_rows(torch.zeros(2, 4))
  with pytest.raises(RuntimeError):
    ops.sim_rank(torch.zeros(2, 4), torch.zeros(2, 4))


def test_weak_low_level_loss_train_step(dev, oracle):
  """train_emb with --weak_low_level_loss runs end to end and logs '_wlow_lvel' with the oracle's
  value."""
  g = load_golden('model_maxout.npz')
  batch = torch_batches(golden_batches(g))[1]
  opt, model = golden_model('maxout', g, low_level_loss=True, weak_low_level_loss=True, norm=True)
  model.logger = MeterLog()
  model.train_start(opt)
  model.train_emb(opt, *batch)
  names = [c[0] for c in model.logger.calls if c[0].startswith('Le')]
  assert names == ['Le_vid', 'Le_ctx_low_lvel', 'Le_vid_inloss', 'Le_para_inloss', 'Le_wlow_lvel',
                   'Le_clip_inloss', 'Le_cap_inloss']
  sds = golden_state_dicts(g)
  nb = golden_batches(g)[1]
  clip_emb, cap_emb, _ = oracle.forward_emb('maxout', sds, nb[0], nb[1], nb[4], nb[5], np.float64)
  want = oracle.groupwise_contrastive_loss(oracle.l2_normalize(clip_emb, np.float64),
                                           oracle.l2_normalize(cap_emb, np.float64), nb[8], nb[9],
                                           0.2, False, True)
  got = [c[1] for c in model.logger.calls if c[0] == 'Le_wlow_lvel'][0]
  assert loss_close(got, want)


def test_abi_error_codes_on_device(dev):
  """Error behaviour of the C ABI with real device buffers: too-small / misaligned workspace,
  bad stripe, bad pooling mode -> negative codes, nothing launched, no exception across the ABI."""
  import ctypes
  from cmhse_amd import _lib
  lib = _lib.load()
  a = torch.randn(8, 16, device=dev)
  rank = torch.empty(8, dtype=torch.int32, device=dev)
  top1 = torch.empty(8, dtype=torch.int32, device=dev)
  ws = torch.empty(4096, dtype=torch.uint8, device=dev)
  args = (a.data_ptr(), a.data_ptr(), 8, 8, 16)
  assert lib.cmhse_sim_rank(*args, 0, 8, rank.data_ptr(), top1.data_ptr(), ws.data_ptr(), 8,
                            None) == -2                       # workspace too small
  assert lib.cmhse_sim_rank(*args, 0, 8, rank.data_ptr(), top1.data_ptr(), ws.data_ptr() + 4,
                            4000, None) == -2                 # misaligned
  assert lib.cmhse_sim_rank(*args, 4, 8, rank.data_ptr(), top1.data_ptr(), ws.data_ptr(), 4096,
                            None) == -1                       # stripe beyond N
  loss = torch.empty((), device=dev)
  assert lib.cmhse_contrastive_fwd(a.data_ptr(), a.data_ptr(), 8, 16, 0.2, 0, 0, loss.data_ptr(),
                                   None, ws.data_ptr(), 16, None) == -2
  sb, gw = _lib.SeqBatch(), _lib.GruWeights()
  assert lib.cmhse_gru_pool_fwd(ctypes.byref(sb), ctypes.byref(gw), 7, a.data_ptr(),
                                ws.data_ptr(), 4096, None) == -1


def test_python_wrappers_validate_shapes(dev):
  from cmhse_amd import ops
  from cmhse_amd.loss import ContrastiveLoss, GroupWiseContrastiveLoss
  a, b = torch.randn(4, 8, device=dev), torch.randn(5, 8, device=dev)
  with pytest.raises(ValueError):
    ContrastiveLoss(margin=0.2)(a, b)                 # im / s must pair up (diag view)
  with pytest.raises(ValueError):
    ops.sim_rank(a, torch.randn(4, 9, device=dev))
  with pytest.raises(ValueError):
    GroupWiseContrastiveLoss(margin=0.2)(a, a, [2, 1], [2, 2])
  with pytest.raises(NotImplementedError):
    ContrastiveLoss(margin=0.2, measure='order')


def test_euclid_rows_target_outlives_caller_locals(dev, oracle):
  """EuclideanLoss.forward_rows addresses its targets by raw device address; the graph node must
  keep that storage alive.  Drop every caller reference, churn the allocator with same-sized
  buffers full of garbage, and only then run backward()."""
  import gc
  from cmhse_amd.decoder import EuclideanLoss
  rng = np.random.RandomState(5)
  rows, cols = 257, 300
  a_np = rng.standard_normal((rows, cols)).astype(np.float32)
  b_np = rng.standard_normal((rows + 40, cols)).astype(np.float32)
  idx = np.sort(rng.choice(rows + 40, rows, replace=False))
  a = torch.from_numpy(a_np).to(dev).requires_grad_(True)

  def make_loss():
    target = torch.from_numpy(b_np).to(dev)
    addrs = np.uint64(target.data_ptr()) + idx.astype(np.uint64) * np.uint64(cols * 4)
    return EuclideanLoss(norm=True).forward_rows(a, addrs, target) * 3.0

  loss = make_loss()
  gc.collect()
  junk = [torch.full((rows + 40, cols), 1e30, device=dev) for _ in range(8)]   # would reuse the block
  torch.cuda.synchronize()
  loss.backward()
  want = 3.0 * oracle.euclidean_loss_backward(a_np, b_np[idx], True, np.float64)
  grad_close(a.grad.cpu().numpy(), want, 'd_a')
  del junk


@pytest.mark.gpu
@pytest.mark.parametrize('max_violation', [False, True])
def test_step_losses_node_equals_normalize_plus_criterion(dev, max_violation):
  """loss.step_losses (cmhse_step_losses_fwd / _bwd: F.normalize, the step's contrastive terms and
  their weighted total as one autograd node) against normalize() + criterion(a, b) term by term
  (model.py:333-343): values bit for bit, total and gradients wrt the un-normalised encoder outputs
  to fp32 rounding (the node sums a row's uses in term order, autograd in its own), and both
  against the NumPy oracle."""
  from cmhse_amd.loss import ContrastiveLoss, normalize, step_losses
  from oracle import cmhse_oracle as O
  torch.manual_seed(5)
  crit = ContrastiveLoss(margin=0.2, max_violation=max_violation, norm=True)
  D = 160
  rows = [32, 32, 32, 32, 117, 117, 9]
  raw = [torch.randn(n, D, device=dev) * (0.5 + e) for e, n in enumerate(rows)]
  raw[1] = raw[0] + 0.8 * torch.randn_like(raw[0])
  raw[5] = raw[4] + 0.8 * torch.randn_like(raw[4])
  terms = [(0, 1, 1.0), (2, 3, 1.0), (0, 0, 0.5), (1, 1, 0.5), (4, 5, 1.0), (4, 4, 0.5), (5, 5, 0.5)]
  sep = [x.clone().requires_grad_(True) for x in raw]
  ns = [normalize(x) for x in sep]
  sep_vals = torch.stack([crit(ns[a], ns[b]) for a, b, _ in terms])
  sep_total = sum(w * sep_vals[k] for k, (_, _, w) in enumerate(terms))
  (sep_total * 1.5).backward()
  fus = [x.clone().requires_grad_(True) for x in raw]
  total, vals = step_losses(crit, fus, terms)
  assert torch.equal(vals, sep_vals)
  assert not vals.requires_grad
  (total * 1.5).backward()
  assert abs(float(total) - float(sep_total)) <= 1e-6 * max(1.0, abs(float(sep_total)))
  for e in range(len(rows)):
    g, r = fus[e].grad, sep[e].grad
    if r is None:                      # an embedding no term uses: zeros
      assert e == 6 and float(g.abs().max()) == 0.0
      continue
    assert float((g - r).abs().max()) <= 2e-6 * max(1e-3, float(r.abs().max())), e
  # oracle: values of the seven terms, and the gradient wrt every encoder output (fp64)
  host = [x.cpu().numpy().astype(np.float64) for x in raw]
  y = [O.l2_normalize(x, dtype=np.float64) for x in host]
  gy = [np.zeros_like(x) for x in host]
  for k, (a, b, w) in enumerate(terms):
    want = O.contrastive_loss(y[a], y[b], margin=0.2, max_violation=max_violation, norm=True,
                              dtype=np.float64)
    assert abs(float(vals[k]) - float(want)) <= 2e-5 * max(1.0, abs(float(want))), k
    d_im, d_s = O.contrastive_loss_backward(y[a], y[b], margin=0.2, max_violation=max_violation,
                                            norm=True)
    gy[a] += 1.5 * w * d_im
    gy[b] += 1.5 * w * d_s
  for e in range(6):
    want = O.l2_normalize_backward(host[e], gy[e])
    got = fus[e].grad.cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5 * max(1e-3, np.abs(want).max()), e
  # mismatched pair sizes are refused, not mis-scored
  with pytest.raises(ValueError):
    step_losses(crit, raw, [(0, 4, 1.0)])


@pytest.mark.gpu
@pytest.mark.parametrize('max_violation', [False, True])
def test_batched_losses_equal_the_separate_calls(dev, max_violation):
  """loss.contrastive_losses (cmhse_contrastive_blocks_fwd / _bwd: one launch set for several
  ContrastiveLoss evaluations) against the separate criterion(a, b) calls: values bit for bit,
  gradients bit for bit per operand."""
  from cmhse_amd.loss import ContrastiveLoss, contrastive_losses
  torch.manual_seed(3)
  crit = ContrastiveLoss(margin=0.2, max_violation=max_violation, norm=True)
  sizes = [32, 120, 32, 7, 129]
  D = 96
  a = [torch.nn.functional.normalize(torch.randn(n, D, device=dev), dim=1) for n in sizes]
  b = [torch.nn.functional.normalize(x + 0.7 * torch.randn_like(x), dim=1) for x in a]
  w = torch.tensor([1.0, 0.5, 2.0, 1.0, 0.25], device=dev)
  sep_a = [x.clone().requires_grad_(True) for x in a]
  sep_b = [x.clone().requires_grad_(True) for x in b]
  sep = torch.stack([crit(x, y) for x, y in zip(sep_a, sep_b)])
  torch.dot(sep, w).backward()
  bat_a = [x.clone().requires_grad_(True) for x in a]
  bat_b = [x.clone().requires_grad_(True) for x in b]
  bat = contrastive_losses(crit, list(zip(bat_a, bat_b)))
  assert torch.equal(bat, sep)
  torch.dot(bat, w).backward()
  for k in range(len(sizes)):
    assert torch.equal(bat_a[k].grad, sep_a[k].grad), k
    assert torch.equal(bat_b[k].grad, sep_b[k].grad), k
  # the self-similarity form CL(x, x) of model.py:335-336: both operands are the same tensor
  x1 = a[1].clone().requires_grad_(True)
  x2 = a[1].clone().requires_grad_(True)
  crit(x1, x1).backward()
  contrastive_losses(crit, [(x2, x2)])[0].backward()
  assert torch.equal(x1.grad, x2.grad)
