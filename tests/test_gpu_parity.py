"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors produced by the
reference (tests/golden) and against the CPU oracle on seeded inputs.

Bars (BASELINE.json north_star): integer ranks / top-1 / R@K / medr bit-identical; embeddings and
losses within 1e-4 (fp32); loss tolerance is relative for |loss| > 1 (an fp32 loss of magnitude
5e3 has an ulp of 5e-4, so an absolute 1e-4 is not representable there).
"""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import EMB_TOL, assert_emb_close, load_golden, golden_state_dicts, golden_batches

pytestmark = pytest.mark.gpu



def loss_close(got, want):
  return abs(float(got) - float(want)) <= 1e-4 * max(1.0, abs(float(want)))


@pytest.fixture(scope='module')
def dev():
  assert torch.cuda.is_available(), 'GPU tests need the MI355X'
  from cmhse_amd import _lib
  _lib.load()     # fail loudly if the HIP library is missing
  return torch.device('cuda', 0)


@pytest.fixture
def tune(dev):
  """Move kernel-shape crossovers of the library (cmhse_tune) for one test; restored afterwards."""
  from cmhse_amd import ops
  saved = {}

  def _set(**kw):
    for k, v in kw.items():
      old = ops.tune(k, v)
      saved.setdefault(k, old)
  yield _set
  for k, v in saved.items():
    ops.tune(k, v)


def make_layer(cls_name, I, H, sd, dev):
  from cmhse_amd import layers
  layer = getattr(layers, cls_name)(I, H)
  layer.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in sd.items()})
  return layer.to(dev)


@pytest.mark.parametrize('cls', ['Attention', 'Maxout', 'Seq2Seq'])
@pytest.mark.parametrize('tag', ['ragged', 'equal', 'one'])
def test_layers_vs_golden(dev, cls, tag):
  g = load_golden('layers.npz')
  sd = {k[len(cls) + 4:]: g[k] for k in g.files if k.startswith(cls + '.sd.')}
  layer = make_layer(cls, 24, 32, sd, dev)
  key = '%s.%s' % (cls, tag)
  x = torch.from_numpy(g[key + '.x']).to(dev)
  lens = torch.from_numpy(g[key + '.lens'])
  h0 = torch.from_numpy(g[key + '.h0']).to(dev)
  with torch.no_grad():
    y = layer(x, lens).cpu().numpy()
    y_h0 = layer(x, lens, h0).cpu().numpy()
  assert_emb_close(y, g[key + '.out'])
  assert_emb_close(y_h0, g[key + '.out_h0'])


@pytest.mark.parametrize('n', [5, 16, 37])
def test_loss_vs_golden(dev, n):
  from cmhse_amd import ops
  from cmhse_amd.loss import ContrastiveLoss, cosine_sim
  g = load_golden('loss.npz')
  a = torch.from_numpy(g['n%d.a' % n]).to(dev)
  b = torch.from_numpy(g['n%d.b' % n]).to(dev)
  an, bn = ops.l2norm_rows(a), ops.l2norm_rows(b)
  np.testing.assert_allclose(an.cpu().numpy(), g['n%d.a_norm' % n], atol=1e-6, rtol=0)
  np.testing.assert_allclose(cosine_sim(an, bn).cpu().numpy(), g['n%d.scores' % n], atol=1e-5,
                             rtol=0)
  for mv in (0, 1):
    for nm in (0, 1):
      crit = ContrastiveLoss(margin=0.2, measure='cosine', max_violation=bool(mv), norm=bool(nm))
      tag = 'n%d.mv%d.norm%d' % (n, mv, nm)
      assert loss_close(crit(an, bn).item(), g[tag + '.ab']), tag
      assert loss_close(crit(an, an).item(), g[tag + '.aa']), tag


def test_normalize_zero_rows(dev):
  from cmhse_amd import ops
  g = load_golden('loss.npz')
  y = ops.l2norm_rows(torch.from_numpy(g['normalize.zero_rows.x']).to(dev)).cpu().numpy()
  np.testing.assert_allclose(y, g['normalize.zero_rows.y'], atol=1e-7, rtol=0)


@pytest.mark.parametrize('n', [50, 203])
def test_rank_vs_golden_bit_exact(dev, n):
  from cmhse_amd.evaluation import i2t, t2i
  g = load_golden('rank.npz')
  a, b = g['n%d.images' % n], g['n%d.captions' % n]
  for nm, fn in [('i2t', i2t), ('t2i', t2i)]:
    rep, top1, ranks = fn(a, b)
    np.testing.assert_array_equal(ranks, g['n%d.%s.ranks' % (n, nm)])
    np.testing.assert_array_equal(top1, g['n%d.%s.top1' % (n, nm)])
    got = np.array([rep[k] for k in ['r1', 'r5', 'r10', 'medr', 'meanr', 'sum']])
    np.testing.assert_array_equal(got, g['n%d.%s.report' % (n, nm)])
    assert ranks.dtype == np.float64 and top1.dtype == np.float64


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


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_model_vs_golden(dev, rnn_type):
  """VSE.forward_emb / structure_emb / encode_data / i2t / t2i against the reference's outputs."""
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data, i2t, t2i, LogCollector
  g = load_golden('model_%s.npz' % rnn_type)
  opt, model = golden_model(rnn_type, g)
  batches = torch_batches(golden_batches(g))
  b = batches[0]
  with torch.no_grad():
    clip_emb, cap_emb, word = model.forward_emb(b[0], b[1], b[4], b[5], return_word=True)
    vid_ctx, para_ctx = model.forward_emb(b[2], b[3], b[6], b[7])
    vid_emb, para_emb = model.structure_emb(clip_emb, cap_emb, b[8], b[9], vid_ctx, para_ctx)
    vid_nc, para_nc = model.structure_emb(clip_emb, cap_emb, b[8], b[9])
  for nm, v in [('clip_emb', clip_emb), ('cap_emb', cap_emb), ('word', word),
                ('vid_context', vid_ctx), ('para_context', para_ctx), ('vid_emb', vid_emb),
                ('para_emb', para_emb), ('vid_emb_noctx', vid_nc), ('para_emb_noctx', para_nc)]:
    assert_emb_close(v.cpu().numpy(), g['fwd.' + nm], nm)

  res = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  for i, nm in enumerate(['vid_embs', 'para_embs', 'clip_embs', 'cap_embs', 'vid_contexts',
                          'para_contexts']):
    assert res[i].dtype == np.float32
    assert_emb_close(res[i], g['enc.' + nm], nm)
  assert list(res[6]) == list(g['enc.num_clips_total'])
  # per-batch 'Letest' meter: last value and running average, like the reference's LogCollector
  want = g['enc.test_losses']
  meter = model.logger.meters['Letest']
  assert loss_close(meter.val, want[-1])
  sizes = [len(b_[8]) for b_ in batches]
  avg = sum(w * s for w, s in zip(want, sizes)) / (.0001 + sum(sizes))
  assert abs(meter.avg - avg) < 1e-4
  for nm, fn in [('i2t', i2t), ('t2i', t2i)]:
    rep, top1, ranks = fn(res[0], res[1])
    np.testing.assert_array_equal(ranks, g['enc.%s.ranks' % nm])
    np.testing.assert_array_equal(top1, g['enc.%s.top1' % nm])
    got = np.array([rep[k] for k in ['r1', 'r5', 'r10', 'medr', 'meanr', 'sum']])
    np.testing.assert_array_equal(got, g['enc.%s.report' % nm])


class MeterLog(object):
  def __init__(self):
    self.calls = []

  def update(self, k, v, n=0):
    self.calls.append((k, float(v), int(n)))


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_train_loss_meters_vs_golden(dev, rnn_type):
  """Forward half of VSE.train_emb: the (name, value, n) stream sent to the logger."""
  g = load_golden('model_%s.npz' % rnn_type)
  batch = torch_batches(golden_batches(g))[1]
  for mv in (0, 1):
    for nm in (0, 1):
      opt, model = golden_model(rnn_type, g, max_violation=bool(mv), norm=bool(nm),
                                low_level_loss=True)
      model.logger = MeterLog()
      with torch.no_grad():
        model.train_losses(opt, *batch)
      tag = 'train.mv%d.norm%d' % (mv, nm)
      assert [c[0] for c in model.logger.calls] == [str(s) for s in g[tag + '.names']]
      for c, want in zip(model.logger.calls, g[tag + '.values']):
        assert loss_close(c[1], want), (tag, c)
      assert [c[2] for c in model.logger.calls] == list(g[tag + '.n'])


# ------------------------------------------------------------------------------------------
# seeded comparisons with the oracle at sizes that exercise tiling edges
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('pool', ['attention', 'maxout', 'seq2seq'])
@pytest.mark.parametrize('S,T,I,H', [(3, 5, 10, 33),       # odd widths: scalar-load path
                                     (150, 9, 500, 96),    # > one M tile, H not a tile multiple
                                     (70, 17, 300, 256),
                                     (2300, 4, 36, 72),    # > 1024 active: LDS-tiled kernel,
                                     (2100, 3, 10, 33)])   #   then the tiny kernel on the tail
def test_gru_pool_vs_oracle(dev, oracle, pool, S, T, I, H):
  from cmhse_amd import layers
  rng = np.random.RandomState(S + T)
  cls = {'attention': 'Attention', 'maxout': 'Maxout', 'seq2seq': 'Seq2Seq'}[pool]
  torch.manual_seed(3)
  layer = getattr(layers, cls)(I, H)
  with torch.no_grad():
    layer.rnn.bias_ih_l0.normal_(0, 0.1)
    layer.rnn.bias_hh_l0.normal_(0, 0.1)
  sd = {'rnn.' + k: v.detach().numpy() for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = rng.randint(1, T + 1, size=S)
  lens[0] = T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32)
  with torch.no_grad():
    y = layer(torch.from_numpy(x).to(dev), torch.from_numpy(lens)).cpu().numpy()
    y0 = layer(torch.from_numpy(x).to(dev), torch.from_numpy(lens),
               torch.from_numpy(h0).to(dev)).cpu().numpy()
  want = oracle.pooled_gru_forward(pool, x, lens, sd, None, np.float64)
  want0 = oracle.pooled_gru_forward(pool, x, lens, sd, h0, np.float64)
  assert_emb_close(y, want)
  assert_emb_close(y0, want0)


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


def test_superbatch_equals_per_batch(dev):
  """encode_data's fused super-batch gives the same embeddings as per-batch VSE calls."""
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data
  g = load_golden('model_attention.npz')
  opt, model = golden_model('attention', g)
  spec = synthetic.ragged_spec(23, seed=5)
  batches = synthetic.make_batches(spec, 6, opt.img_dim, opt.vocab_size, seed=1)
  res = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  from cmhse_amd import ops
  pos = 0
  with torch.no_grad():
    for b in batches:
      clip_emb, cap_emb = model.forward_emb(b[0], b[1], b[4], b[5])
      vc, pc = model.forward_emb(b[2], b[3], b[6], b[7])
      ve, pe = model.structure_emb(clip_emb, cap_emb, b[8], b[9], vc, pc)
      B = len(b[8])
      np.testing.assert_allclose(ops.l2norm_rows(ve).cpu().numpy(), res[0][pos:pos + B],
                                 atol=1e-6, rtol=0)
      np.testing.assert_allclose(ops.l2norm_rows(pe).cpu().numpy(), res[1][pos:pos + B],
                                 atol=1e-6, rtol=0)
      pos += B


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


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout'])
def test_step_plan_makes_every_share_of_a_split_bit_identical(dev, rnn_type):
  """"ranks must be identical for G in {1,2,4,8}".  A 1500-video split whose shares
  straddle the 1024-sequence crossover between the LDS-tiled step kernel and the small-batch one
  (level 2: 1500 videos against 750 / 500; level 1: the whole split's active count passes 1024 many
  steps after a share's): every share encoded with the WHOLE split's step plan
  (evaluation.split_step_plan -> cmhse_seq_batch.step_plan_host) gives all six embedding matrices
  bit for bit as the single call over the split does — and without the plan it does not (the test
  has teeth: the two kernels order their sums differently)."""
  from cmhse_amd import evaluation
  opt, model, batches = _plan_setup(dev, rnn_type=rnn_type)
  quiet = lambda *a: None
  whole, _, _ = evaluation.encode_data_device(opt, model, batches, logging=quiet)
  plan = evaluation.split_step_plan(batches)
  assert plan['v2'][0] == 1500 and plan['v1'][0] > 1024 and plan['t1'][-1] <= 1024
  sizes = np.cumsum([0] + [len(b[8]) for b in batches])
  csizes = np.cumsum([0] + [sum(b[8]) for b in batches])
  differs_without = False
  for world in (2, 3):
    for r in range(world):
      own = [i for i in range(len(batches)) if i % world == r]        # a scrambled deal
      mine = [batches[i] for i in own]
      got, _, _ = evaluation.encode_data_device(opt, model, mine, logging=quiet, step_plan=plan)
      bare, _, _ = evaluation.encode_data_device(opt, model, mine, logging=quiet)
      vid_rows = np.concatenate([np.arange(sizes[i], sizes[i + 1]) for i in own])
      clip_rows = np.concatenate([np.arange(csizes[i], csizes[i + 1]) for i in own])
      for k in KEYS6:
        rows = torch.from_numpy(clip_rows if k in ('clip_emb', 'cap_emb') else vid_rows).to(dev)
        assert torch.equal(got[k], whole[k][rows]), (world, r, k)
        differs_without = differs_without or not torch.equal(bare[k], whole[k][rows])
  assert differs_without, 'no share crossed a kernel crossover: the test does not test the plan'
  # the same holds for a single process that cuts its loader into several super-batches: the plan
  # of the whole loader is the default there
  cut, _, _ = evaluation.encode_data_device(opt, model, batches, logging=quiet,
                                            superbatch_bytes=int(batches[0][0].numel() * 4 * 9))
  for k in KEYS6:
    assert torch.equal(cut[k], whole[k]), k
  # the hoisted input projection of the small-batch steps on the side stream before the first step
  # (early_xproj, the default) or in order in front of those steps: launch order only
  from cmhse_amd import ops
  for _ in range(2):
    with ops.tuned(early_xproj=0):
      inorder, _, _ = evaluation.encode_data_device(opt, model, batches, logging=quiet)
    early, _, _ = evaluation.encode_data_device(opt, model, batches, logging=quiet)
    for k in KEYS6:
      assert torch.equal(inorder[k], whole[k]) and torch.equal(early[k], whole[k]), k
  # ... and in the opt-in bf16x3 math mode (the attention projection's choice between the
  # bf16x3 and the fp32 tile must follow the plan too, not the share's own packed rows)
  ops.set_math_mode('bf16x3')
  try:
    whole3, _, _ = evaluation.encode_data_device(opt, model, batches, logging=quiet)
    for r in range(3):
      own = [i for i in range(len(batches)) if i % 3 == r]
      got, _, _ = evaluation.encode_data_device(opt, model, [batches[i] for i in own], logging=quiet, step_plan=plan)
      vid_rows = np.concatenate([np.arange(sizes[i], sizes[i + 1]) for i in own])
      clip_rows = np.concatenate([np.arange(csizes[i], csizes[i + 1]) for i in own])
      for k in KEYS6:
        rows = torch.from_numpy(clip_rows if k in ('clip_emb', 'cap_emb') else vid_rows).to(dev)
        assert torch.equal(got[k], whole3[k][rows]), ('bf16x3', r, k)
  finally:
    ops.set_math_mode('fp32')


def test_step_plan_is_validated_by_the_library(dev):
  """A plan below the batch's own counts (built from other lengths) is an argument error, not a
  silently different schedule."""
  from cmhse_amd import ops
  H, I, S, T = 64, 16, 40, 5
  x = torch.randn(S, T, I, device=dev)
  w = dict(w_ih=torch.randn(3 * H, I, device=dev), w_hh=torch.randn(3 * H, H, device=dev),
           b_ih=torch.zeros(3 * H, device=dev), b_hh=torch.zeros(3 * H, device=dev))
  lens = np.full(S, T, dtype=np.int64)
  ok, _ = ops.gru_pool_fwd(w, ops.POOL_LAST, lens, I, H, dev, x_ptrs=ops.padded_row_ptrs(x),
                           step_plan=np.full(T + 3, 5000))
  ref, _ = ops.gru_pool_fwd(w, ops.POOL_LAST, lens, I, H, dev, x_ptrs=ops.padded_row_ptrs(x))
  with ops.tuned(tiny_max_seqs=0, mid_max_seqs=0):
    tiled, _ = ops.gru_pool_fwd(w, ops.POOL_LAST, lens, I, H, dev, x_ptrs=ops.padded_row_ptrs(x))
  assert torch.equal(ok, tiled)                 # a plan above 1024 selects the LDS-tiled kernel
  assert torch.allclose(ok, ref, atol=1e-5)
  with pytest.raises(RuntimeError):
    ops.gru_pool_fwd(w, ops.POOL_LAST, lens, I, H, dev, x_ptrs=ops.padded_row_ptrs(x),
                     step_plan=np.full(T, S - 1))
  with pytest.raises(RuntimeError):
    ops.gru_pool_fwd(w, ops.POOL_LAST, lens, I, H, dev, x_ptrs=ops.padded_row_ptrs(x),
                     step_plan=np.array([50, 60, 60, 60, 60]))


def test_stream_schedules_are_bit_identical(dev):
  """The side-stream schedule of encode_group (the two towers on two streams), the grouped launches
  of cmhse_gru_pool_fwd_multi and the early attention pass of the shorter chain on a side stream
  change launch order only, never a bit of the result."""
  from cmhse_amd import synthetic, evaluation
  g = load_golden('model_attention.npz')
  opt, model = golden_model('attention', g)
  spec = synthetic.ragged_spec(37, seed=9)
  batches = synthetic.make_batches(spec, 8, opt.img_dim, opt.vocab_size, seed=2)
  keys = ('vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx')
  saved = evaluation.TWO_STREAMS[0], evaluation.GROUP_TOWERS[0], evaluation.EARLY_POOL[0]
  outs = []
  try:
    for two, group, early in ((False, False, False), (True, False, False), (False, True, False),
                              (False, True, True)):
      evaluation.TWO_STREAMS[0], evaluation.GROUP_TOWERS[0] = two, group
      evaluation.EARLY_POOL[0] = early
      for _ in range(3):   # repeat: a missing stream dependency shows up as a flaky mismatch
        with torch.no_grad():
          r = evaluation.encode_group(model, batches)
        torch.cuda.synchronize()
        outs.append({k: r[k].cpu().numpy() for k in keys})
  finally:
    evaluation.TWO_STREAMS[0], evaluation.GROUP_TOWERS[0], evaluation.EARLY_POOL[0] = saved
  for o in outs[1:]:
    for k in keys:
      assert np.array_equal(o[k], outs[0][k]), k


def test_cpu_tensor_is_rejected_loudly(dev):
  from cmhse_amd import ops
  with pytest.raises(RuntimeError):
    ops.l2norm_rows(torch.zeros(2, 4))
  with pytest.raises(RuntimeError):
    ops.sim_rank(torch.zeros(2, 4), torch.zeros(2, 4))


def test_full_size_rank_properties(dev, oracle):
  """BASELINE full-val size (4917 x 4917 x 1024): size-independent properties.
  (1) ranks of A vs A are all zero (a row's best match is itself);
  (2) permuting the gallery permutes top1 and leaves ranks unchanged;
  (3) a 512-row sample of the stripe agrees exactly with the fp64 oracle on tie-free rows."""
  from cmhse_amd import ops, synthetic
  n = 4917
  a, b = synthetic.correlated_embeddings(n, 1024, 3.0, seed=0)
  ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
  r_self, t_self = ops.sim_rank(ta, ta)
  assert int(r_self.abs().sum()) == 0
  assert torch.equal(t_self.cpu(), torch.arange(n, dtype=torch.int32))
  rank, top1 = ops.sim_rank(ta, tb)
  rep = oracle.recall_report(rank.cpu().numpy())
  assert 25.0 < rep['r1'] < 40.0          # R@1 ~ 33 %
  sample = np.arange(0, n, n // 512)[:512]
  d64 = a[sample].astype(np.float64) @ b.astype(np.float64).T
  diag = d64[np.arange(len(sample)), sample][:, None]
  gap = np.abs(d64 - diag)
  gap[np.arange(len(sample)), sample] = 1.0
  ok = gap.min(axis=1) > 1e-6
  want = (d64 > diag).sum(axis=1)
  np.testing.assert_array_equal(rank.cpu().numpy()[sample][ok], want[ok])
  assert ok.mean() > 0.98


# ------------------------------------------------------------------------------------------
# backward: HIP gradients vs the reference's autograd gradients (golden)
# ------------------------------------------------------------------------------------------
def grad_close(got, want, name=''):
  got = np.asarray(got, dtype=np.float64)
  want = np.asarray(want, dtype=np.float64)
  assert got.shape == want.shape, (name, got.shape, want.shape)
  tol = 2e-4 * max(1e-30, np.abs(want).max()) + 2e-6
  err = np.abs(got - want).max()
  assert err <= tol, '%s: max |diff| %.3e > tol %.3e' % (name, err, tol)


@pytest.mark.parametrize('cls', ['Attention', 'Maxout', 'Seq2Seq'])
@pytest.mark.parametrize('tag', ['ragged', 'equal', 'one'])
def test_layer_backward_vs_golden(dev, cls, tag):
  g = load_golden('layers.npz')
  sd = {k[len(cls) + 4:]: g[k] for k in g.files if k.startswith(cls + '.sd.')}
  layer = make_layer(cls, 24, 32, sd, dev)
  key = '%s.%s' % (cls, tag)
  x = torch.from_numpy(g[key + '.x']).to(dev).requires_grad_(True)
  lens = torch.from_numpy(g[key + '.lens'])
  h0 = torch.from_numpy(g[key + '.h0']).to(dev).requires_grad_(True)
  w = torch.from_numpy(g[key + '.bwd.w']).to(dev)
  out = layer(x, lens, h0)
  assert_emb_close(out.detach().cpu().numpy(), g[key + '.out_h0'])
  (out * w).sum().backward()
  grad_close(x.grad.cpu().numpy(), g[key + '.bwd.dx'], 'dx')
  grad_close(h0.grad.cpu().numpy(), g[key + '.bwd.dh0'], 'dh0')
  for pn, pp in layer.named_parameters():
    grad_close(pp.grad.cpu().numpy(), g[key + '.bwd.grad.rnn.' + pn], pn)


@pytest.mark.parametrize('n', [5, 16, 37])
def test_loss_backward_vs_golden(dev, n):
  from cmhse_amd.loss import ContrastiveLoss, normalize
  g = load_golden('loss.npz')
  for mv in (0, 1):
    for nm in (0, 1):
      crit = ContrastiveLoss(margin=0.2, measure='cosine', max_violation=bool(mv), norm=bool(nm))
      tag = 'n%d.mv%d.norm%d' % (n, mv, nm)
      a = torch.from_numpy(g['n%d.a' % n]).to(dev).requires_grad_(True)
      b = torch.from_numpy(g['n%d.b' % n]).to(dev).requires_grad_(True)
      crit(normalize(a), normalize(b)).backward()
      grad_close(a.grad.cpu().numpy(), g[tag + '.da'], tag + '.da')
      grad_close(b.grad.cpu().numpy(), g[tag + '.db'], tag + '.db')
      a2 = torch.from_numpy(g['n%d.a' % n]).to(dev).requires_grad_(True)
      na = normalize(a2)
      crit(na, na).backward()
      grad_close(a2.grad.cpu().numpy(), g[tag + '.da_self'], tag + '.da_self')


@pytest.mark.parametrize('schedule', ['interleaved', 'levels', 'towers', 'grouped', 'serial'])
@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_train_emb_gradients_vs_golden(dev, rnn_type, schedule, monkeypatch):  # noqa: C901
  """One full VSE.train_emb step (forward, 7 losses, backward, Adam): the parameter gradients
  left in .grad equal the reference's for every encoder, and the parameters moved — with the two
  towers as interleaved chains on their own streams inside one call per level (default), as one
  call per tower on two HIP streams, grouped into shared per-step launches, and on one stream."""
  from cmhse_amd import model as model_mod
  monkeypatch.setattr(model_mod, 'TRAIN_SCHEDULE', [schedule])
  g = load_golden('model_%s.npz' % rnn_type)
  batch = torch_batches(golden_batches(g))[1]
  for mv in (0, 1):
    for nm in (0, 1):
      opt, model = golden_model(rnn_type, g, max_violation=bool(mv), norm=bool(nm),
                                low_level_loss=True)
      model.logger = MeterLog()
      before = [p.detach().clone() for p in model.params]
      model.train_start(opt)
      model.train_emb(opt, *batch)
      tag = 'train.mv%d.norm%d' % (mv, nm)
      for c, want in zip([c for c in model.logger.calls if c[0].startswith('Le')],
                         g[tag + '.values']):
        assert loss_close(c[1], want), (tag, c)
      for i, enc in enumerate([model.clip_enc, model.txt_enc, model.vid_seq_enc,
                               model.txt_seq_enc]):
        for pn, pp in enc.named_parameters():
          grad_close(pp.grad.cpu().numpy(), g['%s.grad%d.%s' % (tag, i, pn)],
                     '%s enc%d %s' % (tag, i, pn))
      assert any(not torch.equal(a, b) for a, b in zip(before, model.params))
      assert model.Eiters == 1


def test_gru_backward_vs_oracle_tiled_sizes(dev, oracle):
  """Sizes that cross tile boundaries in the backward GEMMs (H, I not tile multiples, > 32 seqs)."""
  from cmhse_amd import layers
  rng = np.random.RandomState(5)
  S, T, I, H = 45, 6, 20, 40
  for pool, cls in [('attention', 'Attention'), ('maxout', 'Maxout'), ('seq2seq', 'Seq2Seq')]:
    torch.manual_seed(4)
    layer = getattr(layers, cls)(I, H)
    with torch.no_grad():
      layer.rnn.bias_ih_l0.normal_(0, 0.1)
      layer.rnn.bias_hh_l0.normal_(0, 0.1)
    sd = {'rnn.' + k: v.detach().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    layer = layer.to(dev)
    lens = rng.randint(1, T + 1, size=S)
    lens[0] = T
    x = np.zeros((S, T, I), dtype=np.float32)
    for i, l in enumerate(lens):
      x[i, :l] = rng.standard_normal((l, I))
    h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32)
    w = rng.standard_normal((S, H)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    ht = torch.from_numpy(h0).to(dev).requires_grad_(True)
    (layer(xt, torch.from_numpy(lens), ht) * torch.from_numpy(w).to(dev)).sum().backward()
    _, c = oracle.pooled_gru_forward_cache(pool, x, lens, sd, h0)
    grads, dx, dh0 = oracle.pooled_gru_backward(c, w.astype(np.float64))
    grad_close(xt.grad.cpu().numpy()[:, :dx.shape[1]], dx, pool + ' dx')
    grad_close(ht.grad.cpu().numpy(), dh0, pool + ' dh0')
    for pn, pp in layer.named_parameters():
      grad_close(pp.grad.cpu().numpy(), grads['rnn.' + pn], pool + ' ' + pn)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(8))
def test_resident_tails_fuzz(dev, seed, tune):
  """Random small batches (1-32 sequences, ragged lengths, every pooling, with and without an
  initial state, H = 32 ... 128) through a chain on its own stream: resident tail kernels on
  (forward and backward) against one launch per step — outputs and gradients to fp32 rounding."""
  from cmhse_amd import layers, ops
  rng = np.random.RandomState(100 + seed)
  cls = ['Attention', 'Maxout', 'Seq2Seq'][seed % 3]
  H = int(rng.choice([32, 48, 64, 128]))
  S, T, I = int(rng.randint(1, 33)), int(rng.randint(5, 41)), int(rng.choice([8, 20, 36]))
  torch.manual_seed(seed)
  layer = getattr(layers, cls)(I, H).to(dev)
  lens = rng.randint(1, T + 1, size=S)
  lens[rng.randint(S)] = T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32) if seed % 2 else None
  w = rng.standard_normal((S, H)).astype(np.float32)
  stream = ops.stream_set(dev)[1]

  def run(min_steps):
    tune(fwd_tail_min_steps=min_steps, bwd_tail_min_steps=min_steps)
    layer.zero_grad()
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    ht = torch.from_numpy(h0).to(dev).requires_grad_(True) if h0 is not None else None
    spec = layers.SeqInput('padded', lens.astype(np.int64), layer.POOL)
    out, = layers.run_grouped([(layer, spec, xt, ht, None)], [stream])
    (out * torch.from_numpy(w).to(dev)).sum().backward()
    torch.cuda.synchronize()
    return ([out.detach().clone(), xt.grad.clone()] + ([ht.grad.clone()] if ht is not None else []) +
            [p.grad.clone() for p in layer.parameters()])

  per_step, resident, again = run(0), run(2), run(2)
  for a, b, c in zip(per_step, resident, again):
    assert float((a - b).abs().max()) <= 2e-5 * max(1e-6, float(a.abs().max())), (cls, H, S, T)
    assert torch.equal(b, c)


@pytest.mark.gpu
def test_tall_mid_step_tile_is_bit_identical(dev, tune):
  """A training chain with more than 128 active sequences (DiDeMo: every clip has 80 frames, ~220
  sequences at every step) takes 64 sequences per workgroup in the mid-size forward step
  (gru_step_mid_kernel<4, 16, 8>: one round of workgroups instead of two): outputs and gradients
  bit-identical to the 32-sequence tile (mid_tall_min_seqs = 0)."""
  from cmhse_amd import layers, ops
  rng = np.random.RandomState(3)
  S, T, I, H = 203, 9, 40, 1024
  torch.manual_seed(2)
  layer = layers.Attention(I, H).to(dev)
  lens = rng.randint(5, T + 1, size=S)
  lens[:150] = T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  w = rng.standard_normal((S, H)).astype(np.float32)
  stream = ops.stream_set(dev)[0]

  def run(tall, n_seq=S):
    tune(mid_tall_min_seqs=tall)
    lens_ = lens[:n_seq]
    layer.zero_grad()
    xt = torch.from_numpy(x[:n_seq]).to(dev).requires_grad_(True)
    spec = layers.SeqInput('padded', lens_.astype(np.int64), layer.POOL)
    out, = layers.run_grouped([(layer, spec, xt, None, None)], [stream])
    (out * torch.from_numpy(w[:n_seq]).to(dev)).sum().backward()
    torch.cuda.synchronize()
    return [out.detach().clone(), xt.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

  for a, b in zip(run(0), run(129)):
    assert torch.equal(a, b)
  for n_seq in (150, 131):
    for a, b in zip(run(0, n_seq), run(129, n_seq)):
      assert torch.equal(a, b), n_seq


@pytest.mark.gpu
@pytest.mark.parametrize('pool,cls,H,S,n_long', [('attention', 'Attention', 32, 11, 13), ('maxout', 'Maxout', 48, 40, 13),
                                                 ('seq2seq', 'Seq2Seq', 256, 23, 13), ('attention', 'Attention', 1024, 32, 13),
                                                 ('maxout', 'Maxout', 64, 40, 29), ('attention', 'Attention', 1024, 32, 30)])
def test_forward_tail_as_one_resident_kernel(dev, oracle, pool, cls, H, S, n_long, tune):
  """The few-sequence tail of a training chain's FORWARD pass inside one resident kernel
  (gru_fwd_tail_kernel: chains on a stream of their own, cmhse_gru_job.stream) against one launch
  per step (fwd_tail_min_steps = 0): outputs and every gradient (the kernel also writes the gate
  activations the backward pass consumes) equal to fp32 rounding, bitwise reproducible, and the
  outputs against the float64 oracle."""
  from cmhse_amd import layers, ops
  rng = np.random.RandomState(11 + H)
  I = 20 if H < 1024 else 64
  T = 37
  torch.manual_seed(9)
  layer = getattr(layers, cls)(I, H)
  with torch.no_grad():
    layer.rnn.bias_ih_l0.normal_(0, 0.1)
    layer.rnn.bias_hh_l0.normal_(0, 0.1)
  sd = {'rnn.' + k: v.detach().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = rng.randint(1, 9, size=S)
  long_ones = rng.permutation(S)[:min(S, n_long)]   # n_long > 16: a tail of two 16-row blocks
  lens[long_ones] = rng.randint(10, T + 1, size=len(long_ones))
  lens[long_ones[0]] = T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32)
  w = rng.standard_normal((S, H)).astype(np.float32)
  stream = ops.stream_set(dev)[0]

  def run(min_steps):
    tune(fwd_tail_min_steps=min_steps)
    layer.zero_grad()
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    ht = torch.from_numpy(h0).to(dev).requires_grad_(True)
    spec = layers.SeqInput('padded', lens.astype(np.int64), layer.POOL)
    out, = layers.run_grouped([(layer, spec, xt, ht, None)], [stream])
    (out * torch.from_numpy(w).to(dev)).sum().backward()
    torch.cuda.synchronize()
    return [out.detach().clone(), xt.grad.clone(), ht.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

  per_step, resident, again = run(0), run(4), run(1)
  for a, b, c in zip(per_step, resident, again):
    # (the hidden states differ in their last bits; the attention softmax and 37 steps of BPTT
    # carry that into the gradients)
    assert float((a - b).abs().max()) <= 2e-5 * max(1e-6, float(a.abs().max()))
    assert torch.equal(b, c), 'not reproducible from run to run'
  assert float((per_step[0] - resident[0]).abs().max()) <= 2e-6 * float(per_step[0].abs().max())
  if H <= 256:
    want, _ = oracle.pooled_gru_forward_cache(pool, x, lens, sd, h0)
    assert np.abs(resident[0].cpu().numpy() - want).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('pool,cls,H,S,n_long', [('attention', 'Attention', 32, 11, 13), ('maxout', 'Maxout', 48, 40, 13),
                                                 ('seq2seq', 'Seq2Seq', 128, 23, 13), ('attention', 'Attention', 1024, 32, 13),
                                                 ('maxout', 'Maxout', 64, 40, 29), ('attention', 'Attention', 1024, 32, 30)])
def test_bptt_tail_as_one_resident_kernel(dev, oracle, pool, cls, H, S, n_long, tune):
  """The few-sequence tail of a BPTT chain — the steps with at most 16 active sequences at the end
  of whole-paragraph / whole-video sequences (up to 32: one or two 16-row blocks per workgroup) — inside ONE resident kernel (gru_bwd_tail_kernel:
  grid barrier per step, the rows that cross workgroups written through / read past the
  non-coherent L2s) against one launch per step (bwd_tail_min_steps = 0): every gradient equal to
  fp32 rounding and bitwise reproducible run after run, for H = 32 ... 1024 (2 ... 24 16-k blocks per wave, ragged ownership at the
  small ones), a chain that is ALL tail (S = 11), one whose tail starts mid-way, and against the
  float64 oracle."""
  from cmhse_amd import layers
  rng = np.random.RandomState(5 + H)
  I = 20 if H < 1024 else 64
  T = 37
  torch.manual_seed(9)
  layer = getattr(layers, cls)(I, H)
  with torch.no_grad():
    layer.rnn.bias_ih_l0.normal_(0, 0.1)
    layer.rnn.bias_hh_l0.normal_(0, 0.1)
  sd = {'rnn.' + k: v.detach().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = rng.randint(1, 9, size=S)                 # most sequences end early ...
  long_ones = rng.permutation(S)[:min(S, n_long)]   # n_long > 16: a tail of two 16-row blocks
  lens[long_ones] = rng.randint(10, T + 1, size=len(long_ones))   # ... at most 13 run on
  lens[long_ones[0]] = T
  assert (lens > 9).sum() <= 32 and lens.max() == T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32)
  w = rng.standard_normal((S, H)).astype(np.float32)

  def run(min_steps):
    tune(bwd_tail_min_steps=min_steps)
    layer.zero_grad()
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    ht = torch.from_numpy(h0).to(dev).requires_grad_(True)
    (layer(xt, torch.from_numpy(lens), ht) * torch.from_numpy(w).to(dev)).sum().backward()
    torch.cuda.synchronize()
    return [xt.grad.clone(), ht.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

  per_step, resident, again = run(0), run(4), run(1)
  for a, b, c in zip(per_step, resident, again):
    # same block ownership and accumulation order as the per-step kernel; the compiler contracts
    # the gate arithmetic of the two kernels into different FMAs: equal to fp32 rounding
    assert float((a - b).abs().max()) <= 4e-6 * max(1e-6, float(a.abs().max()))
    assert torch.equal(b, c), 'not reproducible from run to run'
  if H <= 128:
    _, cache = oracle.pooled_gru_forward_cache(pool, x, lens, sd, h0)
    grads, dx, dh0 = oracle.pooled_gru_backward(cache, w.astype(np.float64))
    grad_close(resident[0].cpu().numpy()[:, :dx.shape[1]], dx, pool + ' dx')
    grad_close(resident[1].cpu().numpy(), dh0, pool + ' dh0')
    for (pn, _), got in zip(layer.named_parameters(), resident[2:]):
      grad_close(got.cpu().numpy(), grads['rnn.' + pn], pool + ' ' + pn)


@pytest.mark.parametrize('cls,pool,I,H,S,T', [
    ('Seq2Seq', 'seq2seq', 24, 64, 37, 9),        # 3H = 192: one tall tile, N < 128
    ('Attention', 'attention', 200, 128, 150, 7),   # 3H = 384: two tall tiles; rows split into parts
    ('Maxout', 'maxout', 36, 40, 21, 5),          # 3H = 120: a tall tile forced onto a ragged M
    ('Seq2Seq', 'seq2seq', 130, 192, 300, 4),     # N = 130: a second column tile of 2; > 1024 rows
])
def test_weight_gradient_tall_tile_vs_small_tile(dev, oracle, tune, cls, pool, I, H, S, T):
  """The weight-gradient products (gemm_tn_rows_kernel, tn_rows.hpp) on their 192-row tile (each
  wave 96 x 64 of C, two workgroups per CU) against the 128-row tile: every gradient equal to fp32
  rounding (the row split into parts is chosen per tile count, so the order of the partial sums may
  differ), each bitwise reproducible, and against the float64 oracle.  Ragged M (3H = 120 on a
  192-row tile), N below and just above a column tile, row counts that are not a multiple of 16."""
  from cmhse_amd import layers
  rng = np.random.RandomState(5 + H + S)
  torch.manual_seed(9)
  layer = getattr(layers, cls)(I, H)
  with torch.no_grad():
    layer.rnn.bias_ih_l0.normal_(0, 0.1)
    layer.rnn.bias_hh_l0.normal_(0, 0.1)
  sd = {'rnn.' + k: v.detach().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = rng.randint(1, T + 1, size=S)
  lens[0] = T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  w = rng.standard_normal((S, H)).astype(np.float32)

  def run(bm):
    tune(tn_rows_bm=bm)
    layer.zero_grad()
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    (layer(xt, torch.from_numpy(lens)) * torch.from_numpy(w).to(dev)).sum().backward()
    torch.cuda.synchronize()
    return [xt.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

  small, tall, again = run(128), run(192), run(192)
  for a, b, c in zip(small, tall, again):
    assert float((a - b).abs().max()) <= 2e-5 * max(1e-6, float(a.abs().max()))
    assert torch.equal(b, c), 'not reproducible from run to run'
  _, cache = oracle.pooled_gru_forward_cache(pool, x, lens, sd, None)
  grads, dx, _ = oracle.pooled_gru_backward(cache, w.astype(np.float64))
  grad_close(tall[0].cpu().numpy()[:, :dx.shape[1]], dx, pool + ' dx')
  for (pn, _), got in zip(layer.named_parameters(), tall[1:]):
    grad_close(got.cpu().numpy(), grads['rnn.' + pn], pool + ' ' + pn)


@pytest.mark.parametrize('pool,cls', [('attention', 'Attention'), ('maxout', 'Maxout')])
def test_weight_gradients_in_time_chunks_beside_the_chain(dev, oracle, pool, cls, monkeypatch, tune):
  """The weight-gradient products (gemm_tn_rows_kernel) of a batch long enough to be taken in
  several chunks of time steps (sum T ~ 3.5 k packed rows: chunks close every >= 1024 rows, the
  last one mid-tile), with widths that are not tile multiples (3H = 216 rows of C, I = 36) and so
  few tiles that every launch is row-split (parts + ordered reduce): every gradient against the
  float64 oracle; bit-identical with the products on the chain's own stream instead of the side
  stream; and bitwise reproducible from run to run (no atomics on this path)."""
  from cmhse_amd import layers, ops
  tune(bwd_chunk_rows=1024)      # (default 2048: this batch would be two chunks)
  rng = np.random.RandomState(77)
  S, T, I, H = 330, 14, 36, 72
  torch.manual_seed(8)
  layer = getattr(layers, cls)(I, H)
  with torch.no_grad():
    layer.rnn.bias_ih_l0.normal_(0, 0.1)
    layer.rnn.bias_hh_l0.normal_(0, 0.1)
  sd = {'rnn.' + k: v.detach().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = rng.randint(1, T + 1, size=S)
  lens[:200] = rng.randint(T - 2, T + 1, size=200)
  lens[0] = T
  assert lens.sum() > 3 * 1024
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32)
  w = rng.standard_normal((S, H)).astype(np.float32)

  def run(side):
    monkeypatch.setattr(ops, 'SIDE_STREAMS', [side])
    layer.zero_grad()
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    ht = torch.from_numpy(h0).to(dev).requires_grad_(True)
    (layer(xt, torch.from_numpy(lens), ht) * torch.from_numpy(w).to(dev)).sum().backward()
    torch.cuda.synchronize()
    return [xt.grad.clone(), ht.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

  beside, again, inline = run(True), run(True), run(False)
  for a, b, c in zip(beside, again, inline):
    assert torch.equal(a, b), 'not reproducible from run to run'
    assert torch.equal(a, c), 'side stream changed the result'
  _, cache = oracle.pooled_gru_forward_cache(pool, x, lens, sd, h0)
  grads, dx, dh0 = oracle.pooled_gru_backward(cache, w.astype(np.float64))
  grad_close(beside[0].cpu().numpy()[:, :dx.shape[1]], dx, pool + ' dx')
  grad_close(beside[1].cpu().numpy(), dh0, pool + ' dh0')
  for (pn, _), got in zip(layer.named_parameters(), beside[2:]):
    grad_close(got.cpu().numpy(), grads['rnn.' + pn], pool + ' ' + pn)


@pytest.mark.parametrize('seed', range(12))
def test_gru_pool_fuzz_forward_backward_vs_oracle(dev, oracle, seed):
  """Seeded random shapes, deliberately awkward: single sequences and single steps, widths that
  are not multiples of the 4-float vector path, the 8-unit / 32-sequence / 64-unit tile edges,
  ragged lengths with many short sequences.  Forward output and every gradient (inputs, initial
  state, all weights) against the float64 oracle, for the three pooling modes."""
  from cmhse_amd import layers
  rng = np.random.RandomState(1000 + seed)
  S = int(rng.choice([1, 2, 7, 31, 33, 65, 130]))
  T = int(rng.choice([1, 2, 5, 11]))
  I = int(rng.choice([1, 3, 8, 17, 36, 64]))
  H = int(rng.choice([1, 5, 8, 9, 31, 64, 66]))
  use_h0 = bool(rng.randint(2))
  lens = rng.randint(1, T + 1, size=S)
  lens[rng.randint(S)] = T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32) if use_h0 else None
  w = rng.standard_normal((S, H)).astype(np.float32)
  for pool, cls in [('attention', 'Attention'), ('maxout', 'Maxout'), ('seq2seq', 'Seq2Seq')]:
    torch.manual_seed(seed)
    layer = getattr(layers, cls)(I, H)
    with torch.no_grad():
      layer.rnn.bias_ih_l0.normal_(0, 0.1)
      layer.rnn.bias_hh_l0.normal_(0, 0.1)
    sd = {'rnn.' + k: v.detach().numpy().astype(np.float64) for k, v in layer.state_dict().items()}
    layer = layer.to(dev)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    ht = torch.from_numpy(h0).to(dev).requires_grad_(True) if use_h0 else None
    y = layer(xt, torch.from_numpy(lens), ht)
    (y * torch.from_numpy(w).to(dev)).sum().backward()
    want, c = oracle.pooled_gru_forward_cache(pool, x, lens, sd, h0)
    tag = '%s S%d T%d I%d H%d h0=%d' % (pool, S, T, I, H, use_h0)
    assert_emb_close(y.detach().cpu().numpy(), want, tag)
    grads, dx, dh0 = oracle.pooled_gru_backward(c, w.astype(np.float64))
    grad_close(xt.grad.cpu().numpy()[:, :dx.shape[1]], dx, tag + ' dx')
    if use_h0:
      grad_close(ht.grad.cpu().numpy(), dh0, tag + ' dh0')
    for pn, pp in layer.named_parameters():
      grad_close(pp.grad.cpu().numpy(), grads['rnn.' + pn], tag + ' ' + pn)


# ------------------------------------------------------------------------------------------
# reconstruction path: BASELINE configs 3/4 use --reconstruct_loss
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('lowest', [0, 1])
def test_reconstruction_train_step_vs_golden(dev, lowest):
  from cmhse_amd.model import VSE
  g = load_golden('model_recon.npz')
  tag = 'lowest%d' % lowest
  n_sd = 8 if lowest else 6
  sds = [dict() for _ in range(n_sd)]
  for k in g.files:
    if k.startswith(tag + '.sd'):
      i, key = k[len(tag) + 3:].split('.', 1)
      sds[int(i)][key] = torch.from_numpy(g[k])
  p = tag + '.batch0.'
  nc = tuple(int(c) for c in g[p + 'num_clips'])
  batch = tuple(torch.from_numpy(g[p + nm]) for nm in
                ['clips', 'captions', 'videos', 'paragraphs', 'lengths_clip', 'lengths_cap',
                 'lengths_video', 'lengths_paragraph']) + (
                     nc, tuple(int(c) for c in g[p + 'num_caps']), tuple(range(len(nc))),
                     tuple('v%d' % j for j in range(len(nc))))
  opt = golden_opt('maxout', reconstruct_loss=True, lowest_reconstruct_loss=bool(lowest),
                   low_level_loss=True, norm=True, word_dim=300 if lowest else 12,
                   weight_recon=0.0005, lowest_weight_recon=0.0001, decode_rnn_type='seq2seq')
  model = VSE(opt)
  assert len(model.state_dict(opt)) == n_sd
  model.load_state_dict(sds, opt)
  model.logger = MeterLog()
  model.train_start(opt)
  model.train_emb(opt, *batch)
  calls = [c for c in model.logger.calls if c[0].startswith('Le')]
  assert [c[0] for c in calls] == [str(s) for s in g[tag + '.names']]
  for c, want in zip(calls, g[tag + '.values']):
    assert loss_close(c[1], want), (tag, c, want)
  assert [c[2] for c in calls] == list(g[tag + '.n'])
  for i, m in enumerate(model._modules()):
    for pn, pp in m.named_parameters():
      grad_close(pp.grad.cpu().numpy(), g['%s.grad%d.%s' % (tag, i, pn)],
                 '%s mod%d %s' % (tag, i, pn))


def test_decoder_forward_matches_oracle(dev, oracle):
  """DecoderSequence on a padded, NON-constant input (the reference's generic entry)."""
  from cmhse_amd.decoder import DecoderSequence
  rng = np.random.RandomState(3)
  torch.manual_seed(2)
  dec = DecoderSequence(20, 36)
  sd = {k: v.detach().numpy() for k, v in dec.state_dict().items()}
  dec = dec.to(dev)
  lens = np.array([3, 1, 5, 2])
  x = np.zeros((4, 5, 20), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, 20))
  with torch.no_grad():
    y = dec(torch.from_numpy(x).to(dev), torch.from_numpy(lens)).cpu().numpy()
  c = oracle.gru_forward_cache(x, lens, sd, None, np.float64)
  want = np.concatenate([c['hs'][i, :l] for i, l in enumerate(lens)], 0)
  assert_emb_close(y, want)


@pytest.mark.parametrize('n', [11, 9])
def test_groupwise_loss_vs_golden(dev, n):
  """GroupWiseContrastiveLoss (--weak_low_level_loss): value and gradients vs the reference."""
  from cmhse_amd.loss import GroupWiseContrastiveLoss, normalize
  g = load_golden('loss.npz')
  nc, ncap = list(g['gw%d.num_clips' % n]), list(g['gw%d.num_caps' % n])
  for mv in (0, 1):
    for nm in (0, 1):
      tag = 'gw%d.mv%d.norm%d' % (n, mv, nm)
      crit = GroupWiseContrastiveLoss(margin=0.2, measure='cosine', max_violation=bool(mv),
                                      norm=bool(nm))
      a = torch.from_numpy(g['gw%d.a' % n]).to(dev).requires_grad_(True)
      b = torch.from_numpy(g['gw%d.b' % n]).to(dev).requires_grad_(True)
      loss = crit(normalize(a), normalize(b), nc, ncap)
      assert loss_close(loss.item(), g[tag + '.loss']), tag
      loss.backward()
      grad_close(a.grad.cpu().numpy(), g[tag + '.da'], tag + '.da')
      grad_close(b.grad.cpu().numpy(), g[tag + '.db'], tag + '.db')


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


def test_checkpoint_round_trip_in_reference_format(dev, tmp_path):
  """train.save_checkpoint's payload (train.py:166-172: {'epoch','model': [state_dicts],...})
  round-trips through torch.save / load_state_dict and reproduces the embeddings."""
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data
  g = load_golden('model_attention.npz')
  opt, model = golden_model('attention', g)
  path = str(tmp_path / 'ckpt.pth.tar')
  torch.save({'epoch': 3, 'model': model.state_dict(opt), 'best_rsum': 1.0, 'opt': opt,
              'Eiters': 7}, path)
  ck = torch.load(path, weights_only=False)
  from cmhse_amd.model import VSE
  torch.manual_seed(99)
  model2 = VSE(opt)
  model2.load_state_dict(ck['model'], opt)
  batches = torch_batches(golden_batches(g))
  r1 = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  r2 = encode_data(opt, model2, synthetic.ListLoader(batches), logging=lambda *a: None)
  np.testing.assert_array_equal(r1[0], r2[0])
  assert_emb_close(r2[0], g['enc.vid_embs'])


# ------------------------------------------------------------------------------------------
# bf16x3 math mode (CMHSE_MATH_BF16X3): fp32-grade split products on the bf16 matrix pipe
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('pool', ['attention', 'maxout', 'seq2seq'])
@pytest.mark.parametrize('S,T,I,H', [(2300, 5, 36, 72), (2200, 3, 500, 128)])
def test_bf16x3_mode_vs_oracle(dev, oracle, pool, S, T, I, H):
  """Same parity bar (1e-4) as the exact path; also reports how close the split really is."""
  from cmhse_amd import layers, ops
  rng = np.random.RandomState(S + I)
  cls = {'attention': 'Attention', 'maxout': 'Maxout', 'seq2seq': 'Seq2Seq'}[pool]
  torch.manual_seed(3)
  layer = getattr(layers, cls)(I, H)
  with torch.no_grad():
    layer.rnn.bias_ih_l0.normal_(0, 0.1)
    layer.rnn.bias_hh_l0.normal_(0, 0.1)
  sd = {'rnn.' + k: v.detach().numpy() for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = rng.randint(1, T + 1, size=S)
  lens[0] = T
  x = np.zeros((S, T, I), dtype=np.float32)
  for i, l in enumerate(lens):
    x[i, :l] = rng.standard_normal((l, I))
  h0 = (0.5 * rng.standard_normal((S, H))).astype(np.float32)
  want = oracle.pooled_gru_forward(pool, x, lens, sd, h0, np.float64)
  xt, ht = torch.from_numpy(x).to(dev), torch.from_numpy(h0).to(dev)
  try:
    ops.set_math_mode('bf16x3')
    with torch.no_grad():
      y3 = layer(xt, torch.from_numpy(lens), ht).cpu().numpy()
  finally:
    ops.set_math_mode('fp32')
  with torch.no_grad():
    y = layer(xt, torch.from_numpy(lens), ht).cpu().numpy()
  err3, err = np.abs(y3 - want).max(), np.abs(y - want).max()
  assert err <= EMB_TOL and err3 <= EMB_TOL, (err, err3)
  assert err3 <= 5e-5, 'bf16x3 should be ~1e-5 on these UN-normalised outputs, got %g' % err3
  assert not np.array_equal(y3, y), 'bf16x3 mode did not engage'


# ------------------------------------------------------------------------------------------
# BASELINE full sizes (H = 1024, C3D 500-d / words 300-d, T <= 80): size-independent properties
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('pool', ['attention', 'maxout'])
def test_full_size_encoder_properties(dev, oracle, pool):
  """S = 3000 ragged sequences at embed 1024, img_dim 500, T <= 80 (both step kernels in play):
  (1) a 12-sequence sample equals the fp64 oracle within 1e-4 after L2 normalisation;
  (2) the embedding of a sequence does not depend on the batch it is in or on its position:
      permuting the batch permutes the outputs bit for bit;
  (3) the same 12 sequences encoded alone (tiny kernel only) agree with their in-batch values."""
  from cmhse_amd import layers, ops
  rng = np.random.RandomState(17)
  S, T, I, H = 3000, 80, 500, 1024
  cls = {'attention': 'Attention', 'maxout': 'Maxout'}[pool]
  torch.manual_seed(8)
  layer = getattr(layers, cls)(I, H)
  sd = {'rnn.' + k: v.detach().numpy() for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = np.where(rng.uniform(size=S) < 0.5, T, rng.randint(1, T, size=S))
  x = torch.randn(S, T, I, generator=torch.Generator().manual_seed(3))
  x = x * (torch.arange(T)[None, :] < torch.from_numpy(lens)[:, None])[:, :, None]
  xd = x.to(dev)
  with torch.no_grad():
    y = layer(xd, torch.from_numpy(lens))
    perm = torch.from_numpy(rng.permutation(S))
    yp = layer(xd[perm.to(dev)], torch.from_numpy(lens)[perm])
    assert torch.equal(yp, y[perm.to(dev)])
    sample = np.sort(rng.choice(S, 12, replace=False))
    ys = layer(xd[torch.from_numpy(sample).to(dev)], torch.from_numpy(lens[sample]))
  yn = ops.l2norm_rows(y).cpu().numpy()
  want = oracle.pooled_gru_forward(pool, x[sample].numpy(), lens[sample], sd, None, np.float64)
  want = want / np.linalg.norm(want, axis=1, keepdims=True)
  assert_emb_close(yn[sample], want)
  np.testing.assert_allclose(ops.l2norm_rows(ys).cpu().numpy(), yn[sample], atol=2e-6, rtol=0)


def test_sharded_validation_on_device_world1(dev):
  """parallel_eval.validate_sharded through RCCL (backend 'nccl') with a 1-rank group: the real
  device code path (encode shard, all-gather, stripe ranking, merge) equals encode_data + i2t/t2i."""
  import os
  import socket
  import torch.distributed as dist
  from cmhse_amd import parallel_eval, synthetic
  from cmhse_amd.evaluation import encode_data, i2t, t2i
  g = load_golden('model_maxout.npz')
  opt, model = golden_model('maxout', g)
  spec = synthetic.ragged_spec(19, seed=6)
  batches = synthetic.make_batches(spec, 5, opt.img_dim, opt.vocab_size, seed=2)
  res = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  rep_i, top1_i, ranks_i = i2t(res[0], res[1])
  rep_t, top1_t, ranks_t = t2i(res[0], res[1])
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
  try:
    out = parallel_eval.validate_sharded(opt, model, batches, device=dev, dim=opt.embed_size)
    # the call INTEGRATION.md section C documents: no device=, no assignment= — the agreement check
    # of the derived deal must then reduce on the current GPU (a CPU tensor under an NCCL-only group
    # raises "No backend type associated with device type cpu": ADVICE r03)
    out_doc = parallel_eval.validate_sharded(opt, model, batches)
  finally:
    dist.destroy_process_group()
  for o in (out, out_doc):
    assert o[0] == rep_i and o[1] == rep_t
    np.testing.assert_array_equal(o[2], ranks_i)
    np.testing.assert_array_equal(o[3], ranks_t)
    np.testing.assert_array_equal(o[4], top1_i)
    np.testing.assert_array_equal(o[5], top1_t)


def test_gru_pool_fwd_multi_equals_separate_calls(dev):
  """cmhse_gru_pool_fwd_multi: four unrelated encoders of different widths, pooling modes and
  batch sizes (token input with an embedding table, an initial state, a > 1024-sequence batch that
  starts on the LDS-tiled kernel while the others are on the small-batch kernel, the decoder's
  all-states mode on a constant input) in ONE call give bit-identical outputs and hidden states to
  four separate calls; request-count errors are reported, not launched."""
  import ctypes
  from cmhse_amd import _lib, ops
  rng = np.random.RandomState(11)
  g = torch.Generator().manual_seed(5)

  def weights(I, H, attn):
    w = dict(w_ih=torch.randn(3 * H, I, generator=g).mul_(0.2), w_hh=torch.randn(3 * H, H, generator=g).mul_(0.2),
             b_ih=torch.randn(3 * H, generator=g).mul_(0.1), b_hh=torch.randn(3 * H, generator=g).mul_(0.1))
    if attn:
      w.update(w_lin=torch.randn(H, H, generator=g).mul_(0.2), b_lin=torch.randn(H, generator=g).mul_(0.1),
               w_att=torch.randn(1, H, generator=g).mul_(0.2))
    return {k: v.to(dev) for k, v in w.items()}

  keep = []

  def padded(S, T, I):
    lens = rng.randint(1, T + 1, size=S).astype(np.int64)
    lens[rng.randint(S)] = T
    x = torch.randn(S, T, I, generator=g).to(dev)
    keep.append(x)
    return lens, ops.padded_row_ptrs(x)

  reqs = []
  lens, ptrs = padded(1300, 5, 36)                                  # tiled kernel, then tiny
  reqs.append(dict(weights=weights(36, 72, True), pool_mode=ops.POOL_ATTN, lens=lens, I=36, H=72,
                   device=dev, x_ptrs=ptrs))
  S, T, V = 37, 9, 50                                               # tokens + table, max pooling
  tok = torch.randint(0, V, (S, T), generator=g).to(dev)
  table = torch.randn(V, 20, generator=g).to(dev)
  keep += [tok, table]
  lens = rng.randint(1, T + 1, size=S).astype(np.int64)
  reqs.append(dict(weights=weights(20, 40, False), pool_mode=ops.POOL_MAX, lens=lens, I=20, H=40,
                   device=dev, tok_ptrs=ops.padded_row_ptrs(tok), emb_table=table))
  lens, ptrs = padded(5, 12, 10)                                    # odd widths + initial state
  h0 = torch.randn(5, 33, generator=g).to(dev)
  keep.append(h0)
  reqs.append(dict(weights=weights(10, 33, False), pool_mode=ops.POOL_LAST, lens=lens, I=10, H=33,
                   device=dev, x_ptrs=ptrs, h0_ptrs=ops.padded_row_ptrs(h0)))
  emb = torch.randn(6, 16, generator=g).to(dev)                     # decoder: constant input, all states
  keep.append(emb)
  lens = rng.randint(1, 8, size=6).astype(np.int64)
  reqs.append(dict(weights=weights(16, 24, False), pool_mode=ops.POOL_ALL, lens=lens, I=16, H=24,
                   device=dev, x_ptrs=ops.padded_row_ptrs(emb), constant_input=True))

  single = [ops.gru_pool_fwd(**r) for r in reqs]
  multi = ops.gru_pool_fwd_multi(reqs)
  pair = ops.gru_pool_fwd_multi(reqs[1:3])
  torch.cuda.synchronize()
  for k, ((o1, c1), (o2, c2)) in enumerate(zip(single, multi)):
    assert torch.equal(o1, o2), k
    n_hs = c1['sched'].sum_T * c1['H'] * 4
    assert torch.equal(c1['ws'][:n_hs], c2['ws'][:n_hs]), k
  for (o1, _), (o2, _) in zip(single[1:3], pair):
    assert torch.equal(o1, o2)
  with pytest.raises(ValueError):
    ops.gru_pool_fwd_multi(reqs + reqs[:1])
  lib = _lib.load()
  jobs = (_lib.GruJob * 1)()
  assert lib.cmhse_gru_pool_fwd_multi(jobs, 0, None) == -1
  assert lib.cmhse_gru_pool_fwd_multi(jobs, _lib.MAX_JOBS + 1, None) == -1
  assert lib.cmhse_gru_pool_fwd_multi(jobs, 1, None) == -1          # null request fields
  assert lib.cmhse_gru_pool_fwd_multi(None, 1, None) == -1


@pytest.mark.parametrize('shape', ['one_xcd_queue', 'two_requests', 'full_width', 'scalar_loads', 'long_chain',
                                   'many_rounds', 'uneven_256', 'uneven_768', 'uneven_128_long',
                                   'attention_2048', 'attention_share'])
def test_step_chain_launch_is_bit_identical_to_per_step_launches(dev, tune, shape):
  """The LDS-tiled steps of a call as ONE launch (gru_step_chain_kernel: a workgroup per (step,
  request, row tile, column tile) task, per-XCD task queues, the previous step's rows awaited
  between the x phase and the h phase, state rows written through the non-coherent L2s) against one
  launch per time step (chain_min_steps = 0): outputs and every hidden state equal bit for bit,
  repeated (a missing dependency shows up as a flaky mismatch), and no timeout recorded.
    one_xcd_queue  H = 64: one column tile, so seven XCDs' workgroups take tasks of another queue
    two_requests   attention, last-state, all-states and max-pooling requests of different lengths in
                   one call; the chain is cut where one of them ends; initial states; tokens + table
    full_width     H = 1024 (16 column tiles: two per XCD queue), 3000 sequences, 128-row tiles
    scalar_loads   I not a multiple of 4 (the scalar-load variant of the tile loop; H = 96: a chain needs
                   state rows of whole cache lines, H % 32 == 0 — other widths keep per-step launches)
    long_chain     more steps than one launch covers (96): the chain is cut and resumed
    many_rounds    H = 1024, 6000 + 5000 sequences: ~15 rounds of workgroups per launch, so tasks wait
                   for tiles that run later on other XCDs (the validation pass's regime)
    attention_2048 / attention_share   the attention energies of a chain's steps as tasks of the same launch
                   (phase 2 s + 3 of every queue: H = 2048, one column tile of W_lin per queue; H = 1024, two
                   queues per column tile by row-tile parity, odd counts padded with no-op tickets); also
                   exercised by full_width and many_rounds, whose first request is attention-pooled
    uneven_256 / uneven_768 / uneven_128_long   4, 12 and 2 column tiles — not a whole multiple of the 8
                   XCD queues — at sizes far beyond what the chip holds at once (47-94 row tiles x 12-20
                   steps): with per-XCD queues of unequal length the long queues ran ahead and could fill
                   every slot with waiting workgroups (deadlock in a model of the ticket
                   logic); these counts now share ONE queue whose tickets are a topological order"""
  from cmhse_amd import _lib, ops
  rng = np.random.RandomState(3)
  g = torch.Generator().manual_seed(8)
  g_dev = torch.Generator(device=dev).manual_seed(9)
  keep, fresh = [], []          # fresh: the input tensors whose values are redrawn between rounds

  def weights(I, H, attn):
    w = dict(w_ih=torch.randn(3 * H, I, generator=g).mul_(0.2), w_hh=torch.randn(3 * H, H, generator=g).mul_(0.1),
             b_ih=torch.randn(3 * H, generator=g).mul_(0.1), b_hh=torch.randn(3 * H, generator=g).mul_(0.1))
    if attn:
      w.update(w_lin=torch.randn(H, H, generator=g).mul_(0.1), b_lin=torch.randn(H, generator=g).mul_(0.1),
               w_att=torch.randn(1, H, generator=g).mul_(0.2))
    return {k: v.to(dev) for k, v in w.items()}

  def request(S, T, I, H, mode, h0=False, tokens=False, full=0):
    lens = rng.randint(1, T + 1, size=S).astype(np.int64)
    lens[:max(1, full)] = T
    r = dict(weights=weights(I, H, mode == ops.POOL_ATTN), pool_mode=mode, lens=lens, I=I, H=H, device=dev)
    if tokens:
      tok = torch.randint(0, 40, (S, T), generator=g).to(dev)
      table = torch.randn(40, I, generator=g).to(dev)
      keep.extend([tok, table])
      fresh.extend([tok, table])
      r.update(tok_ptrs=ops.padded_row_ptrs(tok), emb_table=table)
    else:
      x = torch.randn(S, T, I, generator=g).to(dev)
      keep.append(x)
      fresh.append(x)
      r.update(x_ptrs=ops.padded_row_ptrs(x))
    if h0:
      h = torch.randn(S, H, generator=g).to(dev)
      keep.append(h)
      fresh.append(h)
      r.update(h0_ptrs=ops.padded_row_ptrs(h))
    return r

  tune(tiny_max_seqs=0, mid_max_seqs=0)       # every step on the LDS-tiled kernel
  if shape == 'one_xcd_queue':
    reqs = [request(300, 9, 24, 64, ops.POOL_ATTN)]
  elif shape == 'two_requests':
    reqs = [request(700, 11, 36, 128, ops.POOL_ATTN, h0=True), request(450, 5, 20, 128, ops.POOL_LAST, tokens=True),
            request(90, 7, 16, 128, ops.POOL_ALL), request(520, 9, 24, 128, ops.POOL_MAX)]
  elif shape == 'full_width':
    tune(tall_tile_min_wgs=64)                # 128-row tiles
    reqs = [request(3000, 6, 64, 1024, ops.POOL_ATTN, full=1500), request(2100, 4, 32, 1024, ops.POOL_LAST)]
  elif shape == 'scalar_loads':
    reqs = [request(200, 6, 10, 96, ops.POOL_LAST, h0=True), request(150, 8, 10, 96, ops.POOL_ATTN)]
  elif shape == 'many_rounds':
    reqs = [request(6000, 10, 256, 1024, ops.POOL_ATTN, full=3000), request(5000, 7, 64, 1024, ops.POOL_LAST, full=1200)]
  elif shape == 'attention_2048':
    # H = 2048: 8 attention column tiles, one per queue; 4 GRU column tiles per queue
    reqs = [request(1500, 5, 64, 2048, ops.POOL_ATTN, full=900)]
  elif shape == 'attention_share':
    # a rank's share: steps of one round of workgroups or less, where the attention tasks of the
    # previous step fill the slots the recurrence leaves empty; odd row-tile counts (no-op tickets)
    reqs = [request(700, 14, 96, 1024, ops.POOL_ATTN, h0=True, full=200), request(330, 9, 40, 1024, ops.POOL_ATTN, tokens=True)]
  elif shape == 'uneven_256':
    reqs = [request(3000, 12, 48, 256, ops.POOL_ATTN, full=2000)]
  elif shape == 'uneven_768':
    reqs = [request(3000, 12, 32, 768, ops.POOL_MAX, full=2500), request(1500, 10, 32, 768, ops.POOL_ATTN, full=700)]
  elif shape == 'uneven_128_long':
    reqs = [request(6000, 20, 16, 128, ops.POOL_LAST, full=5000)]
  else:
    reqs = [request(70, 130, 8, 32, ops.POOL_ATTN, full=3)]

  def run(min_steps):
    tune(chain_min_steps=min_steps)
    res = ops.gru_pool_fwd_multi(reqs)
    torch.cuda.synchronize()
    assert _lib.load().cmhse_async_status(0) == 0
    out = []
    for o, c in res:
      out.append((o.clone(), c['ws'][:c['sched'].sum_T * c['H'] * 4].clone()))
    return out

  per_step = run(0)
  for _ in range(3):
    for (o1, h1), (o2, h2) in zip(per_step, run(2)):
      assert torch.equal(o1, o2)
      assert torch.equal(h1, h2)
  with ops.StepTimers() as timers:            # the timed form (an event pair around the launch)
    timed = run(2)
  spans = timers.collect()
  assert spans and all(torch.equal(a[0], b[0]) for a, b in zip(per_step, timed))
  # New input VALUES in the same tensors, the chained run FIRST: the workspace blocks come back from
  # the allocator with the previous round's states in them, so a tile that read a state row before
  # its producer's store had reached memory (or from a stale cache line) would see the old round's
  # value and differ from the per-step run that follows.
  for _ in range(4):
    for t_ in fresh:
      t_.normal_(generator=g_dev) if t_.dtype == torch.float32 else t_.random_(0, 40, generator=g_dev)
    chained = run(2)
    for (o1, h1), (o2, h2) in zip(run(0), chained):
      assert torch.equal(o1, o2)
      assert torch.equal(h1, h2)


def test_step_chain_failure_modes_are_an_error_or_a_correct_result(dev, tune):
  """the chain's two assumptions, forced.  (a) A chain while another stream
  saturates the chip with GEMMs (the workgroups of the chain start late and far apart, a dependency
  may be waited for much longer): the result is bit-identical to per-step launches and no timeout
  is recorded.  (b) The abort path for real: with resident_timeout_ms = 0 a dependency wait gives
  up at its second clock check, so a chain of many rounds cannot complete — the call's result is
  then garbage BY CONTRACT, cmhse_async_status reports CMHSE_ERR_TIMEOUT, the next library call
  raises instead of launching, and after the caller has cleared the status the library has fallen
  back to one launch per step (bit-identical again); re-enabled explicitly, the chain works again.
  Never a wrong embedding without an error."""
  from cmhse_amd import _lib, ops
  lib = _lib.load()
  g = torch.Generator().manual_seed(12)
  I, H = 128, 1024
  w = {k: v.to(dev) for k, v in dict(
      w_ih=torch.randn(3 * H, I, generator=g).mul_(0.2), w_hh=torch.randn(3 * H, H, generator=g).mul_(0.1),
      b_ih=torch.randn(3 * H, generator=g).mul_(0.1), b_hh=torch.randn(3 * H, generator=g).mul_(0.1)).items()}
  S, T = 6000, 10
  lens = np.full(S, T, dtype=np.int64)
  lens[3000:] = np.random.RandomState(2).randint(1, T + 1, size=S - 3000)
  x = torch.randn(S, T, I, generator=g).to(dev)
  req = dict(weights=w, pool_mode=ops.POOL_MAX, lens=lens, I=I, H=H, device=dev, x_ptrs=ops.padded_row_ptrs(x))
  tune(tiny_max_seqs=0, mid_max_seqs=0, chain_min_steps=0)
  ref, _ = ops.gru_pool_fwd(**req)
  ref = ref.clone()
  # (a) beside a chip-filling stream
  tune(chain_min_steps=2)
  hog = torch.cuda.Stream()
  a = torch.randn(4096, 4096, device=dev)
  with torch.cuda.stream(hog):
    for _ in range(40):
      a = torch.mm(a, a).mul_(1e-4)
  out, _ = ops.gru_pool_fwd(**req)
  torch.cuda.synchronize()
  assert lib.cmhse_async_status(0) == 0
  assert torch.equal(out, ref)
  # (b) the abort path
  tune(resident_timeout_ms=0)
  out, _ = ops.gru_pool_fwd(**req)
  torch.cuda.synchronize()
  status = lib.cmhse_async_status(0)
  if status == 0:
    assert torch.equal(out, ref)              # no wait was long enough to give up: then it must be right
  else:
    assert status == -5                       # CMHSE_ERR_TIMEOUT
    with pytest.raises(RuntimeError):         # sticky: the next call refuses to launch
      ops.gru_pool_fwd(**req)
    assert lib.cmhse_async_status(1) == -5    # the caller acknowledges ...
    assert lib.cmhse_async_status(0) == 0
    assert ops.tune('multi_step_off') == 1    # ... and the library has fallen back to per-step launches on this device
    tune(resident_timeout_ms=5000)
    out, _ = ops.gru_pool_fwd(**req)
    torch.cuda.synchronize()
    assert lib.cmhse_async_status(0) == 0 and torch.equal(out, ref)
    ops.tune('fwd_tail_min_steps', 4)
    ops.tune('bwd_tail_min_steps', 4)
  tune(resident_timeout_ms=5000, chain_min_steps=2)
  out, _ = ops.gru_pool_fwd(**req)
  torch.cuda.synchronize()
  assert lib.cmhse_async_status(0) == 0 and torch.equal(out, ref)


def test_tuning_contexts_do_not_share_state(dev):
  """SURVEY 8b "re-entrant, no global state" (VERDICT r04 weak 8): two tuning contexts with
  different crossovers, used alternately in one process, each keep their own kernel choice —
  visible as each context's own bit pattern — while the process defaults (ops.tune) stay what they
  were; a backward pass re-enters the context its forward ran in (autograd's thread)."""
  from cmhse_amd import layers, ops
  g = torch.Generator().manual_seed(5)
  I, H, S, T = 24, 64, 90, 6
  x = torch.randn(S, T, I, generator=g).to(dev)
  w = {k: v.to(dev) for k, v in dict(
      w_ih=torch.randn(3 * H, I, generator=g).mul_(0.3), w_hh=torch.randn(3 * H, H, generator=g).mul_(0.3),
      b_ih=torch.zeros(3 * H), b_hh=torch.zeros(3 * H)).items()}
  lens = np.full(S, T, dtype=np.int64)
  req = dict(weights=w, pool_mode=ops.POOL_LAST, lens=lens, I=I, H=H, device=dev, x_ptrs=ops.padded_row_ptrs(x))
  defaults = {k: ops.tune(k) for k in ('tiny_max_seqs', 'mid_max_seqs')}
  small, _ = ops.gru_pool_fwd(**req)                       # process defaults: the small-batch kernel
  tiled_ctx = ops.TuneContext(tiny_max_seqs=0, mid_max_seqs=0)
  other_ctx = ops.TuneContext()
  with tiled_ctx:
    tiled, _ = ops.gru_pool_fwd(**req)                     # this context: the LDS-tiled kernel
    with other_ctx:
      nested, _ = ops.gru_pool_fwd(**req)                  # a nested context with the defaults
    again, _ = ops.gru_pool_fwd(**req)
  after, _ = ops.gru_pool_fwd(**req)
  assert torch.equal(small, nested) and torch.equal(small, after)
  assert torch.equal(tiled, again)
  assert not torch.equal(small, tiled) and torch.allclose(small, tiled, atol=1e-5)
  assert {k: ops.tune(k) for k in defaults} == defaults    # nothing leaked into the process defaults
  assert tiled_ctx.tune('tiny_max_seqs') == 0 and other_ctx.tune('tiny_max_seqs') == defaults['tiny_max_seqs']
  # a context created now copies the defaults of NOW
  ops.tune('mid_units', 8)
  try:
    assert ops.TuneContext().tune('mid_units') == 8 and other_ctx.tune('mid_units') == 0
  finally:
    ops.tune('mid_units', 0)
  # training: the backward pass (autograd thread) runs inside the forward's context
  layer = layers.Seq2Seq(I, H).to(dev)
  ctx = ops.TuneContext(bwd_split_min_seqs=0)
  def grads(c):
    layer.zero_grad()
    xt = x.clone().requires_grad_(True)
    if c is None:
      layer(xt, torch.from_numpy(lens)).sum().backward()
    else:
      with c:
        out = layer(xt, torch.from_numpy(lens)).sum()
      out.backward()                                       # outside the `with`: re-entered from the saved state
    return [xt.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
  with ops.tuned(bwd_split_min_seqs=0):
    want = grads(None)                                     # the one-launch BPTT step via the process defaults
  got = grads(ctx)
  for a, b in zip(want, got):
    assert torch.equal(a, b)


def test_multi_step_kernels_under_a_cu_mask(dev):
  """VERDICT r04 weak 6 / ADVICE r04: the assumptions of csrc/grid_sync.hpp on a chip that gives the
  process fewer CUs than it reports.  tools/cu_mask_check.py in a child process under
  HSA_CU_MASK=0:0-31 (32 of the 256 CUs; the device still reports 256): the step chain — one
  workgroup per task, no co-residency requirement — stays bit-identical to per-step launches with no
  timeout (5x slower, as it should be); the resident tail kernels of a training step (64 workgroups
  that must all be on the chip, one per CU) cannot fit, and the library surfaces CMHSE_ERR_TIMEOUT
  instead of wrong gradients, falls back to per-step launches once the caller acknowledges it, and
  then reproduces the reference gradients bit for bit."""
  import subprocess
  import sys
  from conftest import REPO
  env = dict(os.environ, HSA_CU_MASK='0:0-31')
  res = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'cu_mask_check.py')], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-2000:]
  assert 'step chain under the mask: bit-identical True, status 0' in res.stdout, res.stdout[-2000:]
  assert ('CMHSE_ERR_TIMEOUT surfaced, fallback to per-step launches True, gradients after the acknowledgement equal the '
          'reference True' in res.stdout) or 'resident tails fitted under the mask: gradients equal True' in res.stdout, \
      res.stdout[-2000:]


def test_concurrent_calls_do_not_disturb_each_other(dev):
  """tools/bystander_check.py: a complete attention-pooled encoder call (step chain, attention
  projection, pooling) stays bit-identical while another encoder's per-step launches run on a second
  stream, in every math mode that ships.  (An abandoned bf16x6 mode failed exactly this in 2 of 3
  repetitions, profiles/r05_bf16x6_rate.txt; the far more sensitive form of the check is
  test_no_lost_updates_in_a_bystander_beside_any_math_mode.)"""
  import subprocess
  import sys
  from conftest import REPO
  res = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'bystander_check.py'), '--reps', '12'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-2000:]
  assert res.stdout.count('0 of 12 repetitions') == 2, res.stdout[-2000:]


@pytest.mark.parametrize('neighbour', ['steps', 'chain'])
def test_no_lost_updates_in_a_bystander_beside_any_math_mode(dev, neighbour):
  """profiles/r05_bf16_mfma_bystander.txt: beside gfx950's double-rate matrix instructions a v_pk_fma_f32 of
  another wave on the same SIMD loses updates (lanes 48-63 of one result register).  The bf16x3 tile loop
  did that to bystanders (376-650 wrong sums of 6e9 beside one encoder call) until it moved to
  v_mfma_f32_32x32x8_bf16_1k pairs.  tools/pkfma_canary.py: attn_pool_kernel's inner loop on exact data
  (2e9 sums here), its v_pk_fma_f32 kept, beside an encoder call in each mode that ships — every sum right."""
  import subprocess
  import sys
  from conftest import REPO
  res = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pkfma_canary.py'), '--modes', 'fp32,bf16x3',
                        '--reps', '20', '--neighbour', neighbour],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
  assert res.returncode == 0, res.stdout[-2000:]
  lines = [l for l in res.stdout.splitlines() if l.startswith('neighbour ')]
  assert len(lines) == 2, res.stdout[-2000:]
  for l in lines:
    assert ': 0 wrong sums of 2013265920 ' in l, res.stdout[-2000:]


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


# ------------------------------------------------------------------------------------------
# BASELINE configs[2] / [3] / [4] at their real shapes (ICEP 2048-d, DiDeMo, the full val split)
# ------------------------------------------------------------------------------------------
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


@pytest.mark.parametrize('pool', ['attention', 'maxout', 'seq2seq'])
def test_full_size_encoder_properties_icep(dev, oracle, pool):
  """configs[2]/[4] encoder shape: S = 2200 ragged sequences, img_dim 2048 (post-ReLU-like
  non-negative features), embed 1024, T <= 80 — all three poolings: a 12-sequence sample against
  the fp64 oracle, permutation equivariance bit for bit, in-batch == alone."""
  from cmhse_amd import layers, ops
  rng = np.random.RandomState(23)
  S, T, I, H = 2200, 80, 2048, 1024
  cls = {'attention': 'Attention', 'maxout': 'Maxout', 'seq2seq': 'Seq2Seq'}[pool]
  torch.manual_seed(9)
  layer = getattr(layers, cls)(I, H)
  sd = {'rnn.' + k: v.detach().numpy() for k, v in layer.state_dict().items()}
  layer = layer.to(dev)
  lens = np.where(rng.uniform(size=S) < 0.53, T, rng.randint(1, T, size=S))
  gen = torch.Generator(device=dev).manual_seed(5)
  xd = (0.5 * torch.randn(S, T, I, generator=gen, device=dev)).abs_()
  xd = xd * (torch.arange(T, device=dev)[None, :] < torch.from_numpy(lens).to(dev)[:, None])[:, :, None]
  with torch.no_grad():
    y = layer(xd, torch.from_numpy(lens))
    perm = torch.from_numpy(rng.permutation(S))
    yp = layer(xd[perm.to(dev)], torch.from_numpy(lens)[perm])
    assert torch.equal(yp, y[perm.to(dev)])
    sample = np.sort(rng.choice(S, 12, replace=False))
    ys = layer(xd[torch.from_numpy(sample).to(dev)], torch.from_numpy(lens[sample]))
  yn = ops.l2norm_rows(y).cpu().numpy()
  with _blas_threads():
    want = oracle.pooled_gru_forward(pool, xd[torch.from_numpy(sample).to(dev)].cpu().numpy(),
                                     lens[sample], sd, None, np.float64)
  want = want / np.linalg.norm(want, axis=1, keepdims=True)
  assert_emb_close(yn[sample], want)
  np.testing.assert_allclose(ops.l2norm_rows(ys).cpu().numpy(), yn[sample], atol=2e-6, rtol=0)


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


def test_didemo_icep_encode_and_rank_vs_oracle(dev, oracle):
  """configs[3]: DiDeMo-shaped split (all-80-frame clips, 1-7 clips per video, short sentences,
  vocab 7205, img_dim 2048): encode_data + i2t / t2i on 256 videos against the oracle."""
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data, i2t, t2i
  from cmhse_amd.model import VSE
  opt = _full_opt('attention', 2048, synthetic.DIDEMO_VOCAB)
  torch.manual_seed(4)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(256, seed=2, dataset='didemo')
  assert set(spec.frames_per_clip) == {80}
  batches = synthetic.make_batches(spec, 32, 2048, synthetic.DIDEMO_VOCAB, seed=7, feat='relu')
  res = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  with _blas_threads():
    want = oracle.encode_data('attention', _np_state_dicts(model, opt), _np_batches(batches),
                              margin=0.2)
  err = 0.0
  for i in range(6):
    assert res[i].shape == want[i].shape
    assert_emb_close(res[i], want[i])
    err = max(err, float(np.abs(res[i] - want[i]).max()))
  assert list(res[6]) == list(want[6])
  v64, p64 = want[0].astype(np.float64), want[1].astype(np.float64)
  hv64, hp64 = res[0].astype(np.float64), res[1].astype(np.float64)
  for fn, (q, g), (hq, hg) in [(i2t, (v64, p64), (hv64, hp64)), (t2i, (p64, v64), (hp64, hv64))]:
    rep, top1, ranks = fn(res[0], res[1])
    # (1) the scoring kernel on its own inputs: exact on rows an fp32 dot cannot flip
    r_self, ok_self, d_self = _robust_rank_rows(hq, hg, 2e-6)
    np.testing.assert_array_equal(ranks[ok_self], r_self[ok_self])
    assert np.array_equal(top1[ok_self], d_self.argmax(1)[ok_self])
    # (2) end to end against the oracle's embeddings: exact on rows the embedding error cannot flip
    r_or, ok_or, _ = _robust_rank_rows(q, g, 4 * err + 2e-6)
    np.testing.assert_array_equal(ranks[ok_or], r_or[ok_or])
    assert ok_self.mean() > 0.5 and ok_or.mean() > 0.25, (ok_self.mean(), ok_or.mean())


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


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout'])
def test_icep_recon_train_step_full_dims_vs_oracle(dev, oracle, monkeypatch, rnn_type):
  """configs[2] at its real dimensions: ONE VSE.train_emb step, batch 32, embed 1024, img_dim 2048,
  --low_level_loss --reconstruct_loss --norm, weight_recon 5e-4 — the logged losses and every
  parameter gradient (4 encoders, 2 decoders, the word table) against the fp64 oracle, element-wise
  for both poolings (max pooling: under the HIP forward's own arg-max routing, see
  _check_train_step_vs_oracle)."""
  from cmhse_amd import synthetic
  from cmhse_amd.model import VSE
  opt = _full_opt(rnn_type, 2048, synthetic.ANET_VOCAB, low_level_loss=True, reconstruct_loss=True,
                  norm=True, weight_recon=0.0005, lowest_weight_recon=0.0001,
                  decode_rnn_type='seq2seq')
  torch.manual_seed(11)
  model = VSE(opt)
  assert len(model.state_dict(opt)) == 6
  spec = synthetic.anet_like_spec(32, seed=3)
  batch = synthetic.make_batches(spec, 32, 2048, synthetic.ANET_VOCAB, seed=1, feat='relu')[0]
  rec = _RecordForward(monkeypatch)
  flipped, pairs = _check_train_step_vs_oracle(model, opt, batch, oracle, rnn_type, rec, recon=True)
  print('configs[2] %s: %d of %d (sequence, unit) pairs routed at a near-tie' % (rnn_type, flipped, pairs))


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout'])
def test_c3d_tau0_train_step_full_dims_vs_oracle(dev, oracle, monkeypatch, rnn_type):
  """configs[1] at its real dimensions (README "HSE tau=0 on ActivityNet with C3D": --low_level_loss
  --norm, img_dim 500, embed 1024, batch 32): one VSE.train_emb step, its seven logged losses and
  every parameter gradient of the four encoders and the word table against the fp64 oracle
  (model.py:309-369)."""
  from cmhse_amd import synthetic
  from cmhse_amd.model import VSE
  opt = _full_opt(rnn_type, 500, synthetic.ANET_VOCAB, low_level_loss=True, norm=True)
  torch.manual_seed(12)
  model = VSE(opt)
  assert len(model.state_dict(opt)) == 4
  spec = synthetic.anet_like_spec(32, seed=5)
  batch = synthetic.make_batches(spec, 32, 500, synthetic.ANET_VOCAB, seed=2)[0]
  rec = _RecordForward(monkeypatch)
  flipped, pairs = _check_train_step_vs_oracle(model, opt, batch, oracle, rnn_type, rec, recon=False)
  print('configs[1] %s: %d of %d (sequence, unit) pairs routed at a near-tie' % (rnn_type, flipped, pairs))


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout'])
def test_didemo_recon_train_step_full_dims_vs_oracle(dev, oracle, monkeypatch, rnn_type):
  """configs[3] at its real dimensions: a DiDeMo-shaped batch of 32 (1-7 clips per video, every clip
  80 frames, short sentences, vocab 7205), img_dim 2048, embed 1024, --low_level_loss
  --reconstruct_loss --norm, weight_recon 5e-4 — the step kernels' all-sequences-full-length regime
  (didemo_dev/data.py:127-165 batches; model.py:309-369)."""
  from cmhse_amd import synthetic
  from cmhse_amd.model import VSE
  vocab = 7205
  opt = _full_opt(rnn_type, 2048, vocab, low_level_loss=True, reconstruct_loss=True, norm=True,
                  weight_recon=0.0005, lowest_weight_recon=0.0001, decode_rnn_type='seq2seq',
                  data_name='didemo_precomp')
  torch.manual_seed(13)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(32, seed=7, dataset='didemo')
  batch = synthetic.make_batches(spec, 32, 2048, vocab, seed=3, feat='relu')[0]
  assert int(np.asarray(batch[4]).min()) == 80      # every clip is 80 frames
  rec = _RecordForward(monkeypatch)
  flipped, pairs = _check_train_step_vs_oracle(model, opt, batch, oracle, rnn_type, rec, recon=True)
  print('configs[3] %s: %d of %d (sequence, unit) pairs routed at a near-tie' % (rnn_type, flipped, pairs))


@pytest.mark.parametrize('argv', [['--rounds', '2'], ['--rounds', '2', '--n_videos', '615'],
                                  ['--rounds', '1', '--rnn_type', 'maxout', '--workload', 'anet_c3d_val']])
def test_step_chain_at_full_size_on_new_inputs_every_round(dev, monkeypatch, argv):
  """tools/chain_stress.py: the validation pass at bench.py's sizes (the full split, a rank's
  615-video share of it, the C3D split with max pooling) with the LDS-tiled steps as step chains
  against per-step launches, new input values every round and the chained pass first — all six
  embedding tensors bit-identical, no timeout recorded."""
  import importlib
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
  mod = importlib.import_module('chain_stress')
  monkeypatch.setattr(sys, 'argv', ['chain_stress.py'] + argv)
  with pytest.raises(SystemExit) as e:
    mod.main()
  assert e.value.code == 0


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout'])
def test_full_val_split_icep_encode_sample_vs_oracle(dev, oracle, rnn_type):
  """configs[4], encode half at full size: the whole N = 4917 ActivityNet-val-shaped split at
  img_dim 2048 through encode_data_device as ONE super-batch (what bench.py times), then the six
  embedding matrices of 128 videos (4 loader batches spread over the split) against the oracle
  encoding those batches on their own.  Both for attention pooling (the bench's model) and for the
  reference's DEFAULT pooling, maxout (train.py:68): max pooling inside a step chain is its own code
  path (the running maximum read and written at agent scope from tile to tile)."""
  import bench
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data_device
  from cmhse_amd.model import VSE
  wl = dict(bench.WORKLOADS['anet_icep_val'])
  opt = bench.make_opt(wl, rnn_type, 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset='anet')
  n_batches = (spec.n_videos + wl['batch'] - 1) // wl['batch']
  batches = bench.build_loader(spec, wl, dev, 0, n_batches)
  cat, num_clips_total, _ = encode_data_device(opt, model, batches, logging=lambda *a: None)
  assert cat['vid_emb'].shape == (4917, 1024) and len(num_clips_total) == 4917
  pick = [0, 51, 102, n_batches - 1]
  clip_start = np.concatenate([[0], np.cumsum(num_clips_total)])
  with _blas_threads():
    want = oracle.encode_data(rnn_type, _np_state_dicts(model, opt),
                              _np_batches([batches[i] for i in pick]), margin=0.2)
  v0 = c0 = 0
  for i in pick:
    lo, hi = i * wl['batch'], min(spec.n_videos, (i + 1) * wl['batch'])
    nv, nc = hi - lo, int(clip_start[hi] - clip_start[lo])
    for key, w_idx, a, b, n, o in [('vid_emb', 0, lo, hi, nv, v0), ('para_emb', 1, lo, hi, nv, v0),
                                   ('vid_ctx', 4, lo, hi, nv, v0), ('para_ctx', 5, lo, hi, nv, v0),
                                   ('clip_emb', 2, clip_start[lo], clip_start[hi], nc, c0),
                                   ('cap_emb', 3, clip_start[lo], clip_start[hi], nc, c0)]:
      assert_emb_close(cat[key][a:b].cpu().numpy(), want[w_idx][o:o + n], '%s batch %d' % (key, i))
    v0 += nv
    c0 += nc


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout'])
def test_rank_noise_floor_of_the_exact_path(dev, oracle, rnn_type):
  """"bit-identical ranks" as a measured statement.  The HIP path and the
  torch-CPU oracle encode the same 96 ICEP-shaped videos and rank them END TO END, each with its
  own scorer (bench.py's rank_noise_floor leg, same function): the embeddings agree to 1e-4 (they
  measure ~1e-6); the HIP scorer reproduces an fp64 scorer on the same embeddings (up to a near-tie or two of
  the random-init scores; on separable data exactly: bench.py's rank_check, test_full_size_rank_properties); the
  deviation, applied to separable (correlated) embeddings of the full split's size, moves 0-1 of
  9834 rank rows by one position — the ruler on which bf16x3 moved 6 of 9834.  On the random-init
  embeddings themselves (every score within ~1e-3 of every other) the two fp32 evaluation orders
  may disagree on a few rows: reported, bounded, not asserted to be zero."""
  import bench
  from bench_legs import rank_noise_floor
  from cmhse_amd import synthetic
  from cmhse_amd.model import VSE
  wl = dict(bench.WORKLOADS['anet_icep_val'])
  opt = bench.make_opt(wl, rnn_type, 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(wl['n_videos'], seed=0, dataset='anet')
  with _blas_threads():
    nf = rank_noise_floor(wl, opt, model, spec, 3, wl['n_videos'])
  print('rank noise floor (%s):' % rnn_type, nf)
  assert nf['videos'] == 96
  assert nf['max_abs_embedding_diff'] < EMB_TOL, nf['embedding_diff_by_matrix']
  assert nf['scorer_only']['rank_rows_differing_from_fp64'] <= 2, nf['scorer_only']   # (near-ties of random-init scores)
  # measured: 0 of 9834 rows (attention), 1 of 9834 by one position (maxout) — an fp32 path whose
  # embeddings are 1.6e-7 from the oracle's already sits at the floor of this ruler
  assert nf['correlated']['rank_rows_moved'] <= 2 and nf['correlated']['max_abs_rank_diff'] <= 1, nf['correlated']
  assert nf['random_init']['rank_rows_differing_from_hip'] <= nf['rank_rows'] // 8, nf['random_init']


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


# ------------------------------------------------------------------------------------------
# host hand-over (§8f-3): chunked pull of the loader's pinned tensors under the step pipeline
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('I', [24, 7])
def test_pull_steps_moves_exactly_the_valid_rows(dev, I):
  """cmhse_pull_steps: every valid (sequence, step) row of the pinned source lands in the device
  buffer, chunk by chunk; padding rows are neither read nor written (sentinel survives)."""
  from cmhse_amd import ops
  rng = np.random.RandomState(I)
  S, T = 37, 11
  lens = rng.randint(1, T + 1, size=S)
  lens[3] = T
  src = torch.from_numpy(rng.standard_normal((S, T, I)).astype(np.float32)).pin_memory()
  dst = torch.full((S, T, I), -7.0, device=dev)
  sched = ops.SeqSchedule(lens, dev, x_ptrs=ops.padded_row_ptrs(dst),
                          src_ptrs=ops.padded_row_ptrs(src))
  copy = torch.cuda.Stream(dev)
  copy.wait_stream(torch.cuda.current_stream())
  events = ops.pull_steps(sched, I, copy, chunk=4)
  assert sorted(events) == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10]   # single steps first, then <= chunk
  for ev in events.values():
    ev.synchronize()
  got = dst.cpu().numpy()
  for s in range(S):
    np.testing.assert_array_equal(got[s, :lens[s]], src.numpy()[s, :lens[s]])
    assert (got[s, lens[s]:] == -7.0).all()


@pytest.mark.parametrize('chunk', [2, 8])
def test_pinned_host_batches_encode_bit_identically(dev, chunk, monkeypatch):
  """encode_data_device fed the loader's pinned HOST tensors (features pulled chunk by chunk on a
  copy stream while earlier steps compute) == the same batches resident on the device."""
  from cmhse_amd import evaluation, synthetic
  g = load_golden('model_attention.npz')
  opt, model = golden_model('attention', g)
  spec = synthetic.ragged_spec(23, seed=9, max_frames=13, max_video=17)
  batches = synthetic.make_batches(spec, 6, opt.img_dim, opt.vocab_size, seed=4)
  pinned = [tuple(t.pin_memory() if isinstance(t, torch.Tensor) else t for t in b) for b in batches]
  on_dev = [tuple(t.to(dev) if isinstance(t, torch.Tensor) and t.dim() > 1 else t for t in b)
            for b in batches]
  monkeypatch.setattr(evaluation, 'UPLOAD_CHUNK', [chunk])
  quiet = lambda *a, **k: None
  want, nc_w, _ = evaluation.encode_data_device(opt, model, on_dev, logging=quiet)
  monkeypatch.setattr(evaluation, 'PIPELINE_UPLOAD', [True])
  got, nc_g, _ = evaluation.encode_data_device(opt, model, pinned, logging=quiet)
  monkeypatch.setattr(evaluation, 'PIPELINE_UPLOAD', [False])
  plain, _, _ = evaluation.encode_data_device(opt, model, pinned, logging=quiet)
  assert nc_w == nc_g
  for k in want:
    assert torch.equal(got[k], want[k]), k
    assert torch.equal(plain[k], want[k]), k


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


def test_sharded_validation_two_gpus_rccl(dev, tmp_path):
  """World size 2 over RCCL (backend 'nccl'), one process per GPU: work-balanced deal, all-gather
  of the embeddings, row stripes, merge == the single-GPU encode_data + i2t / t2i.  Skips itself
  on a one-GPU box (the gloo tests cover the same logic at world 2 and 3 on CPU)."""
  if torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs')
  import socket
  import torch.multiprocessing as mp
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data, i2t, t2i
  g = load_golden('model_maxout.npz')
  opt, model = golden_model('maxout', g)
  spec = synthetic.ragged_spec(29, seed=6)
  batches = synthetic.make_batches(spec, 4, opt.img_dim, opt.vocab_size, seed=2)
  res = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  _, top1_i, ranks_i = i2t(res[0], res[1])
  _, top1_t, ranks_t = t2i(res[0], res[1])
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  mp.spawn(_nccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  infra = [f for f in os.listdir(str(tmp_path)) if f.startswith('infra_')]
  if infra:
    pytest.skip('RCCL could not be brought up on this box: ' +
                open(os.path.join(str(tmp_path), infra[0])).read()[:200])
  for r in range(2):
    got = np.load(os.path.join(str(tmp_path), 'r%d.npz' % r))
    np.testing.assert_array_equal(got['ranks_i'], ranks_i)
    np.testing.assert_array_equal(got['ranks_t'], ranks_t)
    np.testing.assert_array_equal(got['top1_i'], top1_i)
    np.testing.assert_array_equal(got['top1_t'], top1_t)


def test_grouped_backward_equals_separate_calls(dev):
  """cmhse_gru_pool_bwd_multi (BPTT steps of independent encoders in shared launches, chains of
  different lengths aligned at their last step) == one cmhse_gru_pool_bwd per encoder, bit for
  bit: every parameter gradient, d input, d h0, d embedding table."""
  from cmhse_amd import layers
  rng = np.random.RandomState(12)
  torch.manual_seed(5)
  H = 64
  enc_a = layers.Attention(24, H).to(dev)
  enc_b = layers.Maxout(20, H).to(dev)
  enc_c = layers.Seq2Seq(H, H).to(dev)
  table = torch.randn(50, 20, device=dev, requires_grad=True)
  xa = torch.randn(37, 9, 24, device=dev)
  la = rng.randint(1, 10, size=37)
  tok = torch.from_numpy(rng.randint(0, 50, size=(21, 17))).to(dev)
  lb = rng.randint(1, 18, size=21)
  rows = torch.randn(30, H, device=dev, requires_grad=True)
  counts = [5, 1, 9, 3, 12]
  h0 = torch.randn(5, H, device=dev, requires_grad=True)

  def calls():
    return [enc_a.call_multi([xa], [la]), enc_b.call_tokens_multi([tok], [lb], table),
            enc_c.call_rows(rows, counts, h0)]

  def grads_of(outs):
    params = [p for e in (enc_a, enc_b, enc_c) for p in e.parameters()] + [table, rows, h0]
    for p in params:
      p.grad = None
    w = [torch.linspace(-1, 1, o.numel(), device=dev).reshape(o.shape) for o in outs]
    sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
    return [p.grad.clone() for p in params], [o.detach().clone() for o in outs]

  g_grp, o_grp = grads_of(layers.run_grouped(calls()))
  g_sep, o_sep = grads_of([layers._PackedGRUPoolFn.apply(c[1], c[2], c[3], c[4],
                                                         c[0].rnn.weight_ih_l0, c[0].rnn.weight_hh_l0,
                                                         c[0].rnn.bias_ih_l0, c[0].rnn.bias_hh_l0,
                                                         *c[0]._extra_weights())
                           for c in [tuple(x) for x in calls()] if c[1].__setattr__('need_grad', True) is None])
  for a, b in zip(o_grp, o_sep):
    assert torch.equal(a, b)
  for i, (a, b) in enumerate(zip(g_grp, g_sep)):
    if i == len(g_grp) - 3:     # the embedding table: float atomics (order-dependent last bits)
      np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5, rtol=1e-5)
    else:
      assert torch.equal(a, b), i


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_bf16x3_mode_on_the_reference_goldens(dev, rnn_type, tune):
  """The optional bf16x3 math mode where parity means something: the REFERENCE's own outputs.
  The golden fixtures are small, so the LDS-tiled kernels (the only ones the mode touches) are
  forced onto them; encode_data then has to reproduce the reference's embeddings within the 1e-4
  bar and the reference's integer ranks / top-1 exactly, with pre-split inputs, hidden states,
  initial states (level 2) and weights all in play."""
  from cmhse_amd import ops, synthetic
  from cmhse_amd.evaluation import encode_data, i2t, t2i
  g = load_golden('model_%s.npz' % rnn_type)
  opt, model = golden_model(rnn_type, g)
  batches = torch_batches(golden_batches(g))
  tune(tiny_max_seqs=0, mid_max_seqs=0)
  exact = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  try:
    ops.set_math_mode('bf16x3')
    res = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  finally:
    ops.set_math_mode('fp32')
  engaged = False
  for i, nm in enumerate(['vid_embs', 'para_embs', 'clip_embs', 'cap_embs', 'vid_contexts',
                          'para_contexts']):
    np.testing.assert_allclose(res[i], g['enc.' + nm], atol=EMB_TOL, rtol=0, err_msg=nm)
    assert_emb_close(exact[i], g['enc.' + nm], nm)
    engaged = engaged or not np.array_equal(res[i], exact[i])
  assert engaged, 'bf16x3 mode did not engage'
  for nm, fn in [('i2t', i2t), ('t2i', t2i)]:
    rep, top1, ranks = fn(res[0], res[1])
    np.testing.assert_array_equal(ranks, g['enc.%s.ranks' % nm])
    np.testing.assert_array_equal(top1, g['enc.%s.top1' % nm])


def test_small_batch_chain_beside_tiled_chain_is_bit_identical(dev, monkeypatch, tune):
  """cmhse_gru_pool_fwd_multi moves a chain that has dropped to small-batch steps onto the side
  stream while the other chain still launches LDS-tiled steps (a rank's share of the split on 8
  GPUs), and projects the still-running chain's rows early when the other one ends.  With the
  small / tiled crossover lowered so that both happen on a small fixture, the result must not
  change by a bit against the one-stream schedule — repeated, because a missing stream dependency
  shows up as a flaky mismatch."""
  from cmhse_amd import synthetic, evaluation
  g = load_golden('model_attention.npz')
  opt, model = golden_model('attention', g)
  spec = synthetic.ragged_spec(41, seed=13, max_frames=14, max_words=5, max_video=16)
  batches = synthetic.make_batches(spec, 8, opt.img_dim, opt.vocab_size, seed=3)
  tune(tiny_max_seqs=40, mid_max_seqs=40)
  keys = ('vid_emb', 'para_emb', 'clip_emb', 'cap_emb', 'vid_ctx', 'para_ctx')
  outs = []
  for early in (False, True, True, True):
    monkeypatch.setattr(evaluation, 'EARLY_POOL', [early])
    with torch.no_grad():
      r = evaluation.encode_group(model, batches)
    torch.cuda.synchronize()
    outs.append({k: r[k].cpu().numpy() for k in keys})
  for o in outs[1:]:
    for k in keys:
      assert np.array_equal(o[k], outs[0][k]), k


def test_deferred_logging_reports_the_same_meters(dev):
  """encode_data_device(defer_logging=True): same embeddings, and after finish() the same
  'Letest' meter (last value, weighted average, count) as the immediate form."""
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data_device
  g = load_golden('model_maxout.npz')
  opt, model = golden_model('maxout', g)
  spec = synthetic.ragged_spec(26, seed=2)
  batches = synthetic.make_batches(spec, 7, opt.img_dim, opt.vocab_size, seed=5)
  quiet = lambda *a, **k: None
  cat_a, nc_a, cv_a = encode_data_device(opt, model, batches, logging=quiet)
  meter_a = model.logger.meters['Letest']
  ref = (meter_a.val, meter_a.avg, meter_a.count)
  cat_b, nc_b, cv_b, finish = encode_data_device(opt, model, batches, logging=quiet,
                                                 defer_logging=True)
  assert 'Letest' not in model.logger.meters      # nothing logged before finish()
  finish()
  meter_b = model.logger.meters['Letest']
  assert (meter_b.val, meter_b.avg, meter_b.count) == ref
  assert nc_a == nc_b and cv_a == cv_b
  for k in cat_a:
    assert torch.equal(cat_a[k], cat_b[k]), k


def test_pinned_host_batches_at_icep_width(dev):
  """The chunked pull at the real feature width (2048 floats = 8 KB rows, every chunk size of the
  schedule in play: 80-frame clips) == the resident pass, bit for bit."""
  from cmhse_amd import evaluation, synthetic
  from cmhse_amd.model import VSE
  opt = _full_opt('attention', 2048, 500, embed_size=128, img_first_size=128, cap_first_size=128)
  torch.manual_seed(3)
  model = VSE(opt)
  spec = synthetic.anet_like_spec(48, seed=4)
  batches = synthetic.make_batches(spec, 16, 2048, 500, seed=6, feat='relu')
  pinned = [tuple(t.pin_memory() if isinstance(t, torch.Tensor) else t for t in b) for b in batches]
  on_dev = [tuple(t.to(dev) if isinstance(t, torch.Tensor) and t.dim() > 1 else t for t in b)
            for b in batches]
  quiet = lambda *a, **k: None
  want, _, _ = evaluation.encode_data_device(opt, model, on_dev, logging=quiet)
  for _ in range(2):
    got, _, _ = evaluation.encode_data_device(opt, model, pinned, logging=quiet)
    for k in want:
      assert torch.equal(got[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('img_dim', [10, 2048])
def test_pad_rows_kernel_rebuilds_collate_fn_tensors(dev, img_dim):
  """collate_packed -> ONE upload -> cmhse_pad_rows on the device == the reference-golden-checked
  collate_fn tensors, bit for bit (float rows on the 4-byte and on the 16-byte path, int64 ids)."""
  from cmhse_amd import collate, ops, synthetic
  samples = synthetic.dataset_samples(5, img_dim, 7)
  ref = collate.collate_fn(samples)
  pk = collate.upload_packed(collate.collate_packed(samples, pin=True), dev)
  for k in range(4):
    assert isinstance(pk[k], ops.Ragged) and pk[k].is_cuda
    got = pk[k].padded()
    assert got.dtype == ref[k].dtype and tuple(got.shape) == tuple(ref[k].shape)
    assert torch.equal(got.cpu(), ref[k]), k
  # the four members are views of one device block
  assert len({pk[k].data.untyped_storage().data_ptr() for k in range(4)}) == 1


@pytest.mark.gpu
@pytest.mark.parametrize('lowest', [0, 1])
def test_train_step_on_a_packed_batch_is_bit_identical(dev, lowest):
  """VSE.train_emb fed the collate_packed 12-tuple (no padding anywhere) == fed collate_fn's padded
  12-tuple: the logged losses bit for bit (the kernels read the same rows through different base
  pointers), every parameter gradient bit for bit where the backward pass is deterministic,
  reconstruction losses included."""
  import copy
  from cmhse_amd import collate, synthetic
  from cmhse_amd.model import VSE
  opt = golden_opt('attention', reconstruct_loss=True, lowest_reconstruct_loss=bool(lowest),
                   low_level_loss=True, norm=True, weight_recon=0.0005, lowest_weight_recon=0.0001,
                   decode_rnn_type='seq2seq')
  torch.manual_seed(5)
  model_a = VSE(opt)
  model_b = VSE(opt)
  model_b.load_state_dict(copy.deepcopy(model_a.state_dict(opt)), opt)
  spec = synthetic.ragged_spec(9, seed=2, max_frames=11, max_video=13)
  padded = synthetic.make_batches(spec, 9, opt.img_dim, opt.vocab_size, seed=3)[0]
  samples = collate.split_samples(padded)
  packed = collate.upload_packed(collate.collate_packed(samples), dev)
  again = collate.collate_fn(samples)
  for k in range(8):
    assert torch.equal(again[k], padded[k])
  logs = []
  for model, batch in [(model_a, padded), (model_b, packed)]:
    model.logger = MeterLog()
    model.train_start(opt)
    model.train_emb(opt, *batch)
    logs.append([c for c in model.logger.calls if c[0].startswith('Le')])
  assert logs[0] == logs[1] and len(logs[0]) >= 9
  for ma, mb in zip(model_a._modules(), model_b._modules()):
    for (na, pa), (nb, pb) in zip(ma.named_parameters(), mb.named_parameters()):
      assert na == nb and pa.grad is not None
      # the forward pass is deterministic (logs equal above); two backward scatters use float
      # atomics — the embedding-table gradient (repeated tokens) and the gradient of a decoder's
      # time-constant input — so those and everything upstream of them vary in the last bits
      # from run to run of the SAME batch
      exact = na != 'embed.weight' and not lowest
      if exact:
        assert torch.equal(pa.grad, pb.grad), na
      else:
        scale = float(pa.grad.abs().max())
        assert float((pa.grad - pb.grad).abs().max()) <= 1e-5 * scale + 1e-12, na


@pytest.mark.gpu
@pytest.mark.parametrize('whole_tower', [False, True])
def test_train_step_with_a_frozen_encoder(dev, whole_tower):
  """A fine-tuning set-up the reference allows (requires_grad = False on one encoder's parameters):
  the two encoder levels then cannot be one autograd node (layers.run_towers declines) and the
  step falls back to a node per level — same loss values, no gradient on the frozen parameters,
  the other gradients equal to the all-trainable step's wherever they do not pass through the
  frozen encoder's inputs."""
  import copy
  from cmhse_amd import synthetic
  from cmhse_amd.model import VSE
  opt = golden_opt('attention', low_level_loss=True, norm=True)
  torch.manual_seed(3)
  model_a = VSE(opt)
  model_b = VSE(opt)
  model_b.load_state_dict(copy.deepcopy(model_a.state_dict(opt)), opt)
  frozen = list(model_b.txt_enc.rnn.parameters())
  if whole_tower:      # nothing of the text tower trains: the towers disagree on requires_grad
    frozen = list(model_b.txt_enc.parameters()) + list(model_b.txt_seq_enc.parameters())
  for p in frozen:
    p.requires_grad_(False)
  spec = synthetic.ragged_spec(9, seed=6, max_frames=11, max_video=13)
  batch = synthetic.make_batches(spec, 9, opt.img_dim, opt.vocab_size, seed=7)[0]
  logs = []
  for model in (model_a, model_b):
    model.logger = MeterLog()
    model.train_start(opt)
    model.train_emb(opt, *batch)
    logs.append([c for c in model.logger.calls if c[0].startswith('Le')])
  assert len(logs[0]) >= 7
  for (ka, va, na), (kb, vb, nb) in zip(logs[0], logs[1]):
    assert ka == kb and na == nb and va == pytest.approx(vb, rel=1e-6, abs=1e-9)
  assert all(p.grad is None for p in frozen)
  for (na, pa), (nb, pb) in zip(model_a.clip_enc.named_parameters(), model_b.clip_enc.named_parameters()):
    assert pb.grad is not None
    assert float((pa.grad - pb.grad).abs().max()) <= 2e-5 * max(1e-6, float(pa.grad.abs().max())), na


@pytest.mark.gpu
def test_late_loss_values_reach_the_collector_in_the_reference_order(dev):
  """VSE.train_emb with this package's LogCollector (the step's loss values leave the device as a
  copy that is still in flight when train_emb returns) against a plain logger object (values
  delivered before train_emb returns): after three steps the meters hold the same sequence —
  names in the same first-use order, last value, running average and count — and a logger swapped
  in mid-way (evaluation.encode_data does that, evaluation.py:101) does not lose a step."""
  import copy
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import LogCollector
  from cmhse_amd.model import VSE
  opt = golden_opt('attention', low_level_loss=True, norm=True, reconstruct_loss=True,
                   weight_recon=0.0005, decode_rnn_type='seq2seq')
  torch.manual_seed(7)
  model_a = VSE(opt)
  model_b = VSE(opt)
  model_b.load_state_dict(copy.deepcopy(model_a.state_dict(opt)), opt)
  spec = synthetic.ragged_spec(9, seed=4, max_frames=11, max_video=13)
  batches = synthetic.make_batches(spec, 3, opt.img_dim, opt.vocab_size, seed=5)
  model_a.logger = MeterLog()
  late = model_b.logger = LogCollector()
  other = LogCollector()
  for model in (model_a, model_b):
    model.train_start(opt)
  for k, b in enumerate(batches):
    model_a.train_emb(opt, *b)
    if k == 2:
      model_b.logger = other          # the third step logs elsewhere; the second is still in flight
    model_b.train_emb(opt, *b)
  calls = model_a.logger.calls
  first_two = [c for c in calls if c[0] not in ('Eit', 'lr')]
  per_step = len(first_two) // 3
  assert per_step >= 9
  want = {}
  for key, v, n in calls[:2 * (per_step + 2)]:
    want.setdefault(key, []).append((v, n))
  assert list(late.meters) == list(want)
  for key, seq in want.items():
    # (the embedding-table gradient is scattered with float atomics: from the second step on the
    # two models agree to rounding, not bit for bit)
    m = late.meters[key]
    assert m.val == pytest.approx(seq[-1][0], rel=1e-4, abs=1e-7), key
    if key.startswith('Le'):
      assert m.count == sum(n for _, n in seq), key
      assert m.avg == pytest.approx(sum(v * n for v, n in seq) / (m.count + 1e-4), rel=1e-4, abs=1e-7)
  third = {k: v for k, v, _ in calls[2 * (per_step + 2):]}
  got = {k: m.val for k, m in other.meters.items()}
  assert list(got) == list(third)
  for k in third:
    assert got[k] == pytest.approx(third[k], rel=1e-4, abs=1e-7), k


@pytest.mark.gpu
def test_packed_loader_encodes_bit_identically(dev, monkeypatch):
  """evaluation.encode_data_device over a loader of collate_packed batches — pinned on the host
  (features pulled step-chunk by step-chunk) and already resident on the device — == over the
  padded batches."""
  from cmhse_amd import collate, evaluation, synthetic
  g = load_golden('model_attention.npz')
  opt, model = golden_model('attention', g)
  spec = synthetic.ragged_spec(23, seed=9, max_frames=13, max_video=17)
  batches = synthetic.make_batches(spec, 6, opt.img_dim, opt.vocab_size, seed=4)
  on_dev = [tuple(t.to(dev) if isinstance(t, torch.Tensor) and t.dim() > 1 else t for t in b)
            for b in batches]
  packed_host = [collate.collate_packed(collate.split_samples(b), pin=True) for b in batches]
  packed_dev = [collate.upload_packed(b, dev) for b in packed_host]
  quiet = lambda *a, **k: None
  want, nc_w, _ = evaluation.encode_data_device(opt, model, on_dev, logging=quiet)
  for pipe, loader in [(True, packed_host), (False, packed_host), (False, packed_dev)]:
    monkeypatch.setattr(evaluation, 'PIPELINE_UPLOAD', [pipe])
    got, nc_g, _ = evaluation.encode_data_device(opt, model, loader, logging=quiet)
    assert nc_g == nc_w
    for k in want:
      assert torch.equal(got[k], want[k]), (pipe, k)


@pytest.mark.gpu
def test_encode_plan_over_a_resident_loader_is_bit_identical(dev):
  """evaluation.encode_data_device(plan=...): a caller that encodes the SAME resident batches pass
  after pass (a validation set kept in HBM; bench.py) keeps the level-1 schedules of the first
  pass.  Planned passes == an unplanned pass bit for bit; a different loader under the same plan
  rebuilds (the key does not match) instead of reusing stale tables."""
  from cmhse_amd import evaluation, synthetic
  g = load_golden('model_attention.npz')
  opt, model = golden_model('attention', g)
  quiet = lambda *a, **k: None

  def loader(seed):
    spec = synthetic.ragged_spec(23, seed=seed, max_frames=13, max_video=17)
    batches = synthetic.make_batches(spec, 6, opt.img_dim, opt.vocab_size, seed=seed + 1)
    return [tuple(t.to(dev) if isinstance(t, torch.Tensor) and t.dim() > 1 else t for t in b)
            for b in batches]

  a, b = loader(9), loader(31)
  want_a, _, _ = evaluation.encode_data_device(opt, model, a, logging=quiet)
  want_b, _, _ = evaluation.encode_data_device(opt, model, b, logging=quiet)
  plan = {}
  for _ in range(3):
    got, _, _ = evaluation.encode_data_device(opt, model, a, logging=quiet, plan=plan)
    for k in want_a:
      assert torch.equal(got[k], want_a[k]), k
  assert plan[0]['key'] == evaluation._plan_key(a)
  got, _, _ = evaluation.encode_data_device(opt, model, b, logging=quiet, plan=plan)
  for k in want_b:
    assert torch.equal(got[k], want_b[k]), k
  assert plan[0]['key'] == evaluation._plan_key(b)
  # an in-place edit of a MIDDLE batch's lengths under a live plan: the key changes, the schedules
  # are rebuilt, and the result is that of an unplanned pass over the edited loader
  mid = b[len(b) // 2]
  i = int(np.argmax(np.asarray(mid[4]) > 1))
  mid[4][i] -= 1
  want_e, _, _ = evaluation.encode_data_device(opt, model, b, logging=quiet)
  got, _, _ = evaluation.encode_data_device(opt, model, b, logging=quiet, plan=plan)
  for k in want_e:
    assert torch.equal(got[k], want_e[k]), k
  assert not torch.equal(want_e['clip_emb'], want_b['clip_emb'])


@pytest.mark.gpu
def test_dataloader_with_collate_packed_feeds_train_emb(dev):
  """The reference's loader construction (activity_net/data.py:157-162: DataLoader(collate_fn=...,
  pin_memory=True)) with collate_packed in place of collate_fn: the pin thread pins the Ragged
  members, and train_emb on such a batch logs the same losses as on collate_fn's batch."""
  import copy
  from cmhse_amd import collate, ops, synthetic
  from cmhse_amd.model import VSE
  opt = golden_opt('maxout', low_level_loss=True, norm=True)
  torch.manual_seed(3)
  model_a = VSE(opt)
  model_b = VSE(opt)
  model_b.load_state_dict(copy.deepcopy(model_a.state_dict(opt)), opt)
  samples = synthetic.dataset_samples(11, opt.img_dim, 12)
  for s in samples:   # token ids inside the model's vocabulary
    assert max(float(c.max()) for c in s[1]) < opt.vocab_size
  logs = []
  for model, fn in [(model_a, collate.collate_fn), (model_b, collate.collate_packed)]:
    loader = torch.utils.data.DataLoader(samples, batch_size=6, shuffle=False, pin_memory=True,
                                         collate_fn=fn, num_workers=0)
    model.logger = MeterLog()
    model.train_start(opt)
    for batch in loader:
      if fn is collate.collate_packed:
        assert isinstance(batch[0], ops.Ragged) and batch[0].is_pinned() and batch[1].is_pinned()
      model.train_emb(opt, *batch)
    logs.append([c for c in model.logger.calls if c[0].startswith('Le')])
  assert len(logs[0]) == 2 * 7
  for a, b in zip(logs[0], logs[1]):
    assert a[0] == b[0] and a[2] == b[2]
    assert loss_close(a[1], b[1]), (a, b)     # second step: after an Adam update with atomics upstream
  assert logs[0][:7] == logs[1][:7]           # first step: bit-identical forward


@pytest.mark.gpu
def test_fused_adam_is_the_same_update(dev, monkeypatch):
  """VSE's optimizer is torch.optim.Adam(params, lr) as upstream (model.py:160); the fused
  implementation it selects on the GPU applies the same update as torch's default one."""
  import copy
  from cmhse_amd import synthetic
  from cmhse_amd.model import VSE
  opt = golden_opt('attention', low_level_loss=True, norm=True)
  from cmhse_amd import model as model_mod
  torch.manual_seed(9)
  monkeypatch.setattr(model_mod, 'FUSED_ADAM', [True])
  model_a = VSE(opt)
  monkeypatch.setattr(model_mod, 'FUSED_ADAM', [False])
  model_b = VSE(opt)
  model_b.load_state_dict(copy.deepcopy(model_a.state_dict(opt)), opt)
  assert model_a.optimizer.defaults.get('fused') and not model_b.optimizer.defaults.get('fused')
  assert model_a.optimizer.param_groups[0]['lr'] == model_b.optimizer.param_groups[0]['lr'] == 0.001
  spec = synthetic.ragged_spec(8, seed=4, max_frames=9, max_video=11)
  batch = synthetic.make_batches(spec, 8, opt.img_dim, opt.vocab_size, seed=6)[0]
  for model in (model_a, model_b):
    model.logger = MeterLog()
    model.train_start(opt)
    for _ in range(3):
      model.train_emb(opt, *batch)
  for ma, mb in zip(model_a._modules(), model_b._modules()):
    for (na, pa), (nb, pb) in zip(ma.named_parameters(), mb.named_parameters()):
      if na == 'embed.weight':   # its gradient is scattered with float atomics
        continue
      assert float((pa.detach() - pb.detach()).abs().max()) <= 2e-6, na


@pytest.mark.gpu
@pytest.mark.parametrize('S,H', [(5, 1024), (29, 1024), (70, 256), (200, 64)])
def test_small_batch_step_shapes_are_bit_identical(dev, S, H, tune):
  """The mid-size step's launch shapes — 16 / 8 / 4 hidden units per workgroup, 8 waves x 1 K slice
  or 4 waves x 2 — are scheduling choices: every combination gives the same bits, forward and
  (through the saved gates) backward."""
  from cmhse_amd import layers
  torch.manual_seed(S)
  I = 40
  layer = layers.Attention(I, H).to(dev)
  lens = torch.randint(1, 9, (S,), dtype=torch.int64)
  lens[0] = 8
  x = torch.randn(S, 8, I, device=dev)
  h0 = torch.randn(S, H, device=dev)
  outs = []
  for units, waves in [(16, 8), (16, 4), (8, 8), (8, 4), (4, 8), (4, 4), (0, 0)]:
    tune(mid_units=units, mid_waves=waves)
    xr = x.clone().requires_grad_(True)
    layer.zero_grad()
    y = layer(xr, lens, h0)
    y.square().sum().backward()
    outs.append((y.detach().clone(), xr.grad.clone(), layer.rnn.weight_hh_l0.grad.clone()))
  for o in outs[1:]:
    for a, b in zip(outs[0], o):
      assert torch.equal(a, b)


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



@pytest.mark.gpu
def test_host_fed_train_step_is_bit_identical(dev, monkeypatch):
  """a training step fed from the loader's pinned HOST tensors (model.py:225-227,
  activity_net/data.py:157-162) must equal the step on resident tensors bit for bit — logged losses
  and every parameter gradient (the word table's is scattered with float atomics: close, not
  equal) — whichever way the batch crosses PCIe:
    pull      train.py unchanged: train_emb pulls the frame rows time-chunk by time-chunk under the
              visual chain (model.HOST_FEED 'pull'; the projection chunks wait for exactly their rows);
    ahead     train.py unchanged: train_emb copies the batch into one of its two device slots on
              the copy stream ('ahead'; what 'auto', the default, does while the host runs ahead of
              the GPU — 'auto' itself is exercised too: pull for the first step, then either);
    upload    the reference's `.cuda()` in front of the step (HOST_PULL off);
    prefetch  collate.DevicePrefetcher(loader, prepare=model.prepare_batch): one batch ahead on the
              copy stream, schedules built a step early;
    packed    the same through collate_packed's un-padded block (ops.Ragged members), pulled.
  Sized so that the forward projection IS cut into time chunks (>= 6144 packed rows)."""
  import copy
  from cmhse_amd import collate, model as model_mod, ops, synthetic
  from cmhse_amd.model import VSE
  opt = golden_opt('attention', low_level_loss=True, norm=True, img_dim=64, embed_size=64,
                   img_first_size=64, cap_first_size=64, reconstruct_loss=True, weight_recon=0.0005)
  spec = synthetic.anet_like_spec(64, seed=2)
  batches = synthetic.make_batches(spec, 32, opt.img_dim, opt.vocab_size, seed=3)
  assert len(batches) == 2 and int(np.asarray(batches[0][4]).sum() + np.asarray(batches[0][6]).sum()) >= 6144
  pin = lambda b: tuple(t.pin_memory() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b))
  host = [pin(b) for b in batches]
  resident = [tuple(t.to(dev) if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b))
              for b in batches]
  packed = [collate.collate_packed(collate.split_samples(b), pin=True) for b in batches]
  assert isinstance(packed[0][0], ops.Ragged) and packed[0][0].is_pinned()
  torch.manual_seed(5)
  ref = VSE(opt)
  sd0 = copy.deepcopy(ref.state_dict(opt))

  def run(feed):
    model = VSE(opt)
    model.load_state_dict(copy.deepcopy(sd0), opt)
    model.logger = MeterLog()
    model.train_start(opt)
    pulls = []
    real = ops.pull_steps
    monkeypatch.setattr(ops, 'pull_steps', lambda *a, **k: (pulls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(model_mod, 'HOST_PULL', [feed != 'upload'])
    monkeypatch.setattr(model_mod, 'HOST_FEED', [{'ahead': 'ahead', 'auto': 'auto', 'packed_ahead': 'ahead'}.get(feed, 'pull')])
    loader = {'resident': resident, 'pull': host, 'upload': host, 'packed': packed, 'ahead': host,
              'auto': host, 'packed_ahead': packed,
              'prefetch': collate.DevicePrefetcher(host, prepare=model.prepare_batch)}[feed]
    grads = None
    for k, b in enumerate(loader):
      if feed == 'prefetch':
        assert b[0].is_cuda and hasattr(b[0], '_cmhse_prep')
      model.train_emb(opt, *b)
      if k == 0:
        torch.cuda.synchronize()
        grads = {(i, n): p.grad.detach().clone() for i, m in enumerate(model._modules())
                 for n, p in m.named_parameters()}
    torch.cuda.synchronize()
    if feed != 'auto':
      assert bool(pulls) == (feed in ('pull', 'packed')), (feed, len(pulls))
    return [c for c in model.logger.calls if c[0].startswith('Le')], grads

  want_log, want_g = run('resident')
  n_first = len(want_log) // 2
  for feed in ['pull', 'upload', 'prefetch', 'packed', 'ahead', 'auto', 'packed_ahead']:
    log, g = run(feed)
    assert log[:n_first] == want_log[:n_first], feed          # first step: bit-identical losses
    for a, b in zip(log[n_first:], want_log[n_first:]):       # second: after an update with atomics upstream
      assert a[0] == b[0] and a[2] == b[2] and loss_close(a[1], b[1]), (feed, a, b)
    for key, w in want_g.items():
      if key[1] == 'embed.weight':
        assert float((g[key] - w).abs().max()) <= 1e-6 * max(1.0, float(w.abs().max())), (feed, key)
      else:
        assert torch.equal(g[key], w), (feed, key)


@pytest.mark.gpu
@pytest.mark.parametrize('feed', ['pull', 'ahead'])
def test_host_fed_batch_may_be_dropped_right_after_the_call(dev, monkeypatch, feed):
  """A DataLoader's pinned batch is released by the loop as soon as train_emb returns, and torch's
  pinned-memory allocator hands the block to the next batch — while the GPU, a step behind the
  host, may not have read it yet.  The hand-overs must keep what they read alive themselves: the
  DMA copies ('ahead') through torch's own bookkeeping, the pull kernels ('pull'), which read the
  tensors by address, through VSE's (model._host_rows).  Here, after a warm-up step, the copy
  stream is kept busy for a few hundred ms so that the hand-over runs LATE, the batch is dropped,
  and same-sized pinned blocks full of garbage are allocated at once (they reuse a freed block);
  the step's losses must still be those of the resident step (checked to FAIL without the
  keep-alive: round 4)."""
  import copy
  import gc
  from cmhse_amd import model as model_mod, ops, synthetic
  from cmhse_amd.model import VSE
  opt = golden_opt('attention', low_level_loss=True, norm=True, img_dim=64, embed_size=64,
                   img_first_size=64, cap_first_size=64)
  spec = synthetic.anet_like_spec(64, seed=8)
  batches = synthetic.make_batches(spec, 32, opt.img_dim, opt.vocab_size, seed=9)
  on_dev = lambda b: [t.to(dev) if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b)]
  pinned = lambda b: [t.pin_memory() if isinstance(t, torch.Tensor) and i < 4 else t for i, t in enumerate(b)]
  torch.manual_seed(6)
  ref = VSE(opt)
  sd0 = copy.deepcopy(ref.state_dict(opt))
  ref.logger = MeterLog()
  ref.train_start(opt)
  for b in batches:
    ref.train_emb(opt, *on_dev(b))
  torch.cuda.synchronize()
  want = [c for c in ref.logger.calls if c[0].startswith('Le')]
  want = want[len(want) // 2:]                     # the second step

  class LateLog(MeterLog):       # takes the values late, like evaluation.LogCollector: train_emb does not wait
    def __init__(self):
      MeterLog.__init__(self)
      self.pending = []

    def defer(self, thunk):
      self.pending.append(thunk)

    def settle(self):
      while self.pending:
        self.pending.pop(0)()

    def _update(self, k, v, n=0):
      self.update(k, v, n)

  model = VSE(opt)
  model.load_state_dict(copy.deepcopy(sd0), opt)
  model.logger = LateLog()
  model.train_start(opt)
  monkeypatch.setattr(model_mod, 'HOST_FEED', [feed])
  model.train_emb(opt, *pinned(batches[0]))        # warm-up: arenas, streams, allocator pools
  torch.cuda.synchronize()
  model.logger.settle()
  model.logger.calls = []
  shapes = [(t.shape, t.dtype) for t in batches[1][:4]]
  host = pinned(batches[1])
  ptrs = {t.data_ptr() for t in host[:4]}
  with torch.cuda.stream(ops.copy_stream(dev)):    # ~0.3 s of work in front of whatever is queued there next
    a = torch.randn(8192, 8192, device=dev)
    for _ in range(30):
      a = (a @ a) * 1e-4
  model.train_emb(opt, *host)
  busy = not model._step_done.query()
  del host
  gc.collect()
  junk = [torch.full(s, 7 if d == torch.int64 else 1e30, dtype=d).pin_memory() for s, d in shapes for _ in range(2)]
  reused = any(t.data_ptr() in ptrs for t in junk)
  torch.cuda.synchronize()
  model.logger.settle()
  got = [c for c in model.logger.calls if c[0].startswith('Le')]
  assert [c[0] for c in got] == [c[0] for c in want]
  for g, w in zip(got, want):
    assert loss_close(g[1], w[1]), (feed, g, w, 'GPU still busy when the batch was dropped: %s, a pinned '
                                    'block was reused: %s' % (busy, reused))
  assert busy, 'the hand-over was not late: the test did not exercise what it is for'
  del junk, a


@pytest.mark.gpu
def test_grid_barrier_timeout_is_an_error_not_a_trap(dev, tune):
  """csrc/grid_sync.hpp: the grid barrier of the resident kernels.  With every workgroup
  present the barriers complete (256 workgroups x 50 rounds); with one arrival missing nobody can
  complete them — the wall-time bound (resident_timeout_ms) must end the kernel through the abort
  path (no trap, no hang), raise the device's status word, make the next gru forward / backward
  call return CMHSE_ERR_TIMEOUT (RuntimeError) without launching, and everything works again once
  the status is cleared."""
  import ctypes
  from cmhse_amd import _lib, layers, ops
  lib = _lib.load()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  ws = torch.zeros(64, dtype=torch.int32, device=dev)
  assert ops.async_status() == 0
  assert lib.cmhse_selftest_grid_sync(ws.data_ptr(), 256, 0, 50, stream) == 0
  torch.cuda.synchronize()
  assert ws[2].item() == 256 and ws[3].item() == 0 and ops.async_status() == 0
  layer = layers.Seq2Seq(8, 32).to(dev)
  x, lens = torch.randn(3, 4, 8, device=dev), torch.tensor([4, 2, 1])
  tune(resident_timeout_ms=30)
  try:
    import time
    t0 = time.time()
    assert lib.cmhse_selftest_grid_sync(ws.data_ptr(), 64, 1, 3, stream) == 0
    torch.cuda.synchronize()
    assert time.time() - t0 < 5.0                      # bounded: ~30 ms, not the default 5 s, not forever
    assert ws[2].item() == 0 and ws[3].item() == 64    # every workgroup left through the abort path
    assert ops.async_status() == _lib.load().cmhse_async_status(0) == -5
    with pytest.raises(RuntimeError, match='grid barrier'):
      with torch.no_grad():
        layer(x, lens)
    assert ops.async_status(clear=True) == -5 and ops.async_status() == 0
    # acknowledging a timeout switches the multi-step kernels off on THIS device — for every tuning
    # context, the knobs themselves untouched (round 6)
    assert ops.tune('multi_step_off') == 1
    assert [ops.tune(k) for k in ('chain_min_steps', 'fwd_tail_min_steps', 'bwd_tail_min_steps')] == [2, 4, 4]
    with ops.TuneContext() as other:
      assert other.tune('multi_step_off') == 1
    with torch.no_grad():
      y = layer(x, lens)
    assert torch.isfinite(y).all()
  finally:
    ops.async_status(clear=True)
    ops.tune('chain_min_steps', 2)       # (this box is fine: a positive set re-enables the multi-step kernels here)
    assert ops.tune('multi_step_off') == 0
