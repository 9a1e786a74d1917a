"""Shared helpers of the GPU parity tests (tests/test_gpu_*.py): golden-model builders, oracle
adapters, tolerances.  Bars (BASELINE.json north_star): integer ranks / top-1 / R@K / medr bit-identical;
embeddings and losses within 1e-4 (fp32) — and within GOLDEN_TOL where the comparison is against the
reference's own outputs or the oracle (conftest.assert_emb_close); loss tolerance is relative for
|loss| > 1 (an fp32 loss of magnitude 5e3 has an ulp of 5e-4)."""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import EMB_TOL, assert_emb_close, load_golden, golden_state_dicts, golden_batches  # noqa: F401


def loss_close(got, want):
  return abs(float(got) - float(want)) <= 1e-4 * max(1.0, abs(float(want)))


def make_layer(cls_name, I, H, sd, dev):
  from cmhse_amd import layers
  layer = getattr(layers, cls_name)(I, H)
  layer.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in sd.items()})
  return layer.to(dev)


def golden_opt(rnn_type, **kw):
  opt = argparse.Namespace(
      margin=0.2, word_dim=12, embed_size=32, grad_clip=0.0, learning_rate=0.001,
      max_violation=False, img_dim=24, measure='cosine', rnn_type=rnn_type, img_first_size=32,
      cap_first_size=32, low_level_loss=False, weak_low_level_loss=False, reconstruct_loss=False,
      lowest_reconstruct_loss=False, norm=False, data_name='anet_precomp', vocab_size=60)
  for k, v in kw.items():
    setattr(opt, k, v)
  return opt


def golden_model(rnn_type, g, **kw):
  from cmhse_amd.model import VSE
  opt = golden_opt(rnn_type, **kw)
  model = VSE(opt)
  sds = golden_state_dicts(g)
  model.load_state_dict([{k: torch.from_numpy(v) for k, v in sd.items()} for sd in sds], opt)
  return opt, model


def torch_batches(batches):
  return [tuple(torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in b)
          for b in batches]


class MeterLog(object):
  def __init__(self):
    self.calls = []

  def update(self, k, v, n=0):
    self.calls.append((k, float(v), int(n)))


def _plan_setup(dev, n_videos=1500, H=256, img_dim=64, rnn_type='attention'):
  from cmhse_amd import synthetic
  from cmhse_amd.model import VSE
  opt = argparse.Namespace(
      margin=0.2, word_dim=300, embed_size=H, grad_clip=0.0, learning_rate=0.001, max_violation=False,
      img_dim=img_dim, measure='cosine', rnn_type=rnn_type, img_first_size=H, cap_first_size=H,
      low_level_loss=False, weak_low_level_loss=False, reconstruct_loss=False,
      lowest_reconstruct_loss=False, norm=False, data_name='anet_precomp', vocab_size=300)
  torch.manual_seed(4)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(n_videos, seed=2)
  batches = synthetic.make_batches(spec, 32, img_dim, opt.vocab_size, seed=3)
  batches = [tuple(x.to(dev) if isinstance(x, torch.Tensor) and i < 4 else x for i, x in enumerate(b))
             for b in batches]
  return opt, model, batches


KEYS6 = ['vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx']


def grad_close(got, want, name=''):
  got = np.asarray(got, dtype=np.float64)
  want = np.asarray(want, dtype=np.float64)
  assert got.shape == want.shape, (name, got.shape, want.shape)
  tol = 2e-4 * max(1e-30, np.abs(want).max()) + 2e-6
  err = np.abs(got - want).max()
  assert err <= tol, '%s: max |diff| %.3e > tol %.3e' % (name, err, tol)


def _blas_threads(n=16):
  """The oracle's per-step GEMMs are small: all cores of a big host oversubscribe OpenBLAS."""
  from threadpoolctl import threadpool_limits
  return threadpool_limits(limits=n)


def _full_opt(rnn_type, img_dim, vocab, **kw):
  opt = argparse.Namespace(
      margin=0.2, word_dim=300, embed_size=1024, grad_clip=0.0, learning_rate=0.001,
      max_violation=False, img_dim=img_dim, measure='cosine', rnn_type=rnn_type,
      img_first_size=1024, cap_first_size=1024, low_level_loss=False, weak_low_level_loss=False,
      reconstruct_loss=False, lowest_reconstruct_loss=False, norm=False,
      data_name='anet_precomp', vocab_size=vocab)
  for k, v in kw.items():
    setattr(opt, k, v)
  return opt


def _np_state_dicts(model, opt):
  return [{k: v.detach().cpu().numpy() for k, v in sd.items()} for sd in model.state_dict(opt)]


def _np_batches(batches):
  return [tuple(x.cpu().numpy() if isinstance(x, torch.Tensor) else x for x in b) for b in batches]


def _robust_rank_rows(q64, g64, eps):
  """fp64 ranks of every row and the mask of rows whose diagonal score is further than `eps` from
  every other score of the row (their rank cannot change under perturbations < eps / 2)."""
  d = q64 @ g64.T
  n = d.shape[0]
  dii = d[np.arange(n), np.arange(n)]
  ranks = (d > dii[:, None]).sum(1)
  gap = np.abs(d - dii[:, None])
  gap[np.arange(n), np.arange(n)] = np.inf
  return ranks, gap.min(1) > eps, d


class _RecordForward(object):
  """Records every ops.gru_pool_fwd_multi call of a train_emb step (requests' forward contexts), so a
  test can read what the HIP forward kept — here the arg-max steps of the max pooling."""

  def __init__(self, monkeypatch):
    from cmhse_amd import ops
    self.calls = []
    real = ops.gru_pool_fwd_multi

    def wrapper(requests, *a, **kw):
      res = real(requests, *a, **kw)
      self.calls.append([ctx for _, ctx in res])
      return res
    monkeypatch.setattr(ops, 'gru_pool_fwd_multi', wrapper)

  def argmax_routes(self, n_clip, n_cap):
    """The routing of the six max-pooled encoder passes, keyed like oracle.apply_argmax_route, rows
    in input order.  'interleaved' training schedule: call 0 = level 1 (visual, text), call 1 =
    level 2 (visual, text)."""
    from cmhse_amd import ops

    def in_order(ctx):
      a = ops.saved_region(ctx, 'argmax')
      assert a is not None
      out = np.empty(tuple(a.shape), dtype=np.int64)
      out[ctx['sched'].order] = a.cpu().numpy()
      return out
    (v1, t1), (v2, t2) = self.calls[0], self.calls[1]
    av, at = in_order(v1), in_order(t1)
    return dict(clip=av[:n_clip], vid=av[n_clip:], cap=at[:n_cap], par=at[n_cap:],
                v2=in_order(v2), p2=in_order(t2))


def _check_train_step_vs_oracle(model, opt, batch, oracle, rnn_type, recorder, recon):
  """ONE VSE.train_emb step against the fp64 oracle: the logged (name, value, n) stream to 1e-4 and
  EVERY parameter gradient element-wise (grad_close).  Max pooling routes each output's gradient
  through the arg-max step, a discrete choice: the oracle's backward is given the routing the HIP
  forward used (read back from its workspace), and the pairs it routes differently from its own
  fp64 arg-max must be near-ties — their number and largest gap are returned."""
  sds = _np_state_dicts(model, opt)
  model.logger = MeterLog()
  model.train_start(opt)
  model.train_emb(opt, *batch)
  torch.cuda.synchronize()
  nb = _np_batches([batch])[0]
  routes, report = None, {}
  if rnn_type == 'maxout':
    routes = recorder.argmax_routes(len(batch[4]), len(batch[5]))
  kw = dict(margin=0.2, max_violation=False, norm=True, low_level_loss=True, argmax_route=routes,
            route_report=report)
  with _blas_threads():
    if recon:
      log, _, grads = oracle.train_step_recon(rnn_type, sds, nb, lowest=False, weight_recon=0.0005, **kw)
    else:
      grads = oracle.train_step_grads(rnn_type, sds, nb, **kw)
      log = oracle.train_losses(rnn_type, sds, nb, margin=0.2, max_violation=False, norm=True,
                                low_level_loss=True, dtype=np.float64)[0]
  calls = [c for c in model.logger.calls if c[0].startswith('Le')]
  assert [c[0] for c in calls] == [l[0] for l in log]
  for c, l in zip(calls, log):
    assert loss_close(c[1], l[1]), (c, l)
    assert c[2] == l[2]
  flipped = pairs = 0
  for name, (n_diff, gap, n_pairs) in report.items():
    # a pair routed differently from the fp64 arg-max is a near-tie: the two steps' values agree to
    # within the fp32 forward's own error on a hidden state (measured <= 2e-6 after 80 steps)
    assert gap <= 1e-5, 'encoder %s: routed away from the fp64 arg-max across a gap of %.3e' % (name, gap)
    flipped += n_diff
    pairs += n_pairs
  if rnn_type == 'maxout':
    # measured 0-2 of ~350,000 (sequence, unit) pairs per step (profiles/r05_maxout_route_flips.txt)
    assert pairs > 0 and flipped <= 8, '%d of %d (sequence, unit) pairs routed away from the fp64 arg-max' % (flipped, pairs)
  for i, m in enumerate(model._modules()):
    for pn, pp in m.named_parameters():
      assert pp.grad is not None, (i, pn)
      grad_close(pp.grad.cpu().numpy(), grads[i][pn], 'mod%d %s' % (i, pn))
  return flipped, pairs


def _nccl_worker(rank, world, port, out_dir):
  import os
  import sys
  import torch.distributed as dist
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from cmhse_amd import parallel_eval, synthetic
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.cuda.set_device(rank)
  dev = torch.device('cuda', rank)
  try:
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)             # the communicator really works on this box
    torch.cuda.synchronize()
  except Exception as e:               # RCCL / peer-access set-up of the box, not this library
    open(os.path.join(out_dir, 'infra_r%d.txt' % rank), 'w').write(repr(e))
    return
  try:
    g = load_golden('model_maxout.npz')
    opt, model = golden_model('maxout', g)
    spec = synthetic.ragged_spec(29, seed=6)
    batches = synthetic.make_batches(spec, 4, opt.img_dim, opt.vocab_size, seed=2)
    out = parallel_eval.validate_sharded(opt, model, batches, device=dev, dim=opt.embed_size)
    np.savez(os.path.join(out_dir, 'r%d.npz' % rank), ranks_i=out[2], ranks_t=out[3],
             top1_i=out[4], top1_t=out[5])
  finally:
    dist.destroy_process_group()
