"""Pins the CPU oracle (oracle/cmhse_oracle.py) to outputs of the reference itself
(tests/golden/*.npz, produced by tools/make_golden.py from /root/reference)."""
import numpy as np
import pytest

from conftest import load_golden, golden_state_dicts, golden_batches

TOL = 2e-6   # fp32 restatement vs fp32 reference: summation-order noise only


@pytest.mark.parametrize('cls,fn', [('Attention', 'attention_forward'),
                                    ('Maxout', 'maxout_forward'),
                                    ('Seq2Seq', 'seq2seq_forward')])
@pytest.mark.parametrize('tag', ['ragged', 'equal', 'one'])
def test_layers_forward(oracle, cls, fn, tag):
  g = load_golden('layers.npz')
  p = {k[len(cls) + 4:]: g[k] for k in g.files if k.startswith(cls + '.sd.')}
  key = '%s.%s' % (cls, tag)
  x, lens, h0 = g[key + '.x'], g[key + '.lens'], g[key + '.h0']
  y = getattr(oracle, fn)(x, lens, p)
  y_h0 = getattr(oracle, fn)(x, lens, p, h0)
  np.testing.assert_allclose(y, g[key + '.out'], atol=TOL, rtol=0)
  np.testing.assert_allclose(y_h0, g[key + '.out_h0'], atol=TOL, rtol=0)


@pytest.mark.parametrize('n', [5, 16, 37])
def test_normalize_and_contrastive(oracle, n):
  g = load_golden('loss.npz')
  a, b = g['n%d.a' % n], g['n%d.b' % n]
  an, bn = oracle.l2_normalize(a), oracle.l2_normalize(b)
  np.testing.assert_allclose(an, g['n%d.a_norm' % n], atol=1e-7, rtol=0)
  np.testing.assert_allclose(oracle.cosine_sim(an, bn), g['n%d.scores' % n], atol=1e-6, rtol=0)
  for mv in (0, 1):
    for nm in (0, 1):
      tag = 'n%d.mv%d.norm%d' % (n, mv, nm)
      ab = oracle.contrastive_loss(an, bn, 0.2, bool(mv), bool(nm))
      aa = oracle.contrastive_loss(an, an, 0.2, bool(mv), bool(nm))
      np.testing.assert_allclose(ab, g[tag + '.ab'], rtol=2e-6, atol=1e-6)
      np.testing.assert_allclose(aa, g[tag + '.aa'], rtol=2e-6, atol=1e-6)


def test_normalize_zero_rows_and_euclid(oracle):
  g = load_golden('loss.npz')
  np.testing.assert_array_equal(oracle.l2_normalize(g['normalize.zero_rows.x']),
                                g['normalize.zero_rows.y'])
  np.testing.assert_allclose(oracle.euclidean_loss(g['euclid.a'], g['euclid.b'], True),
                             g['euclid.norm1'], rtol=1e-6)
  np.testing.assert_allclose(oracle.euclidean_loss(g['euclid.a'], g['euclid.b'], False),
                             g['euclid.norm0'], rtol=1e-6)


@pytest.mark.parametrize('n', [50, 203])
def test_rank_bit_exact(oracle, n):
  g = load_golden('rank.npz')
  a, b = g['n%d.images' % n], g['n%d.captions' % n]
  for nm, fn in [('i2t', oracle.i2t), ('t2i', oracle.t2i)]:
    rep, top1, ranks = fn(a, b)
    np.testing.assert_array_equal(ranks, g['n%d.%s.ranks' % (n, nm)])
    np.testing.assert_array_equal(top1, g['n%d.%s.top1' % (n, nm)])
    want = g['n%d.%s.report' % (n, nm)]
    got = np.array([rep[k] for k in ['r1', 'r5', 'r10', 'medr', 'meanr', 'sum']])
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_model_forward_and_encode(oracle, rnn_type):
  g = load_golden('model_%s.npz' % rnn_type)
  sds = golden_state_dicts(g)
  batches = golden_batches(g)
  b = batches[0]
  clip_emb, cap_emb, word = oracle.forward_emb(rnn_type, sds, b[0], b[1], b[4], b[5])
  vid_ctx, para_ctx, _ = oracle.forward_emb(rnn_type, sds, b[2], b[3], b[6], b[7])
  vid_emb, para_emb = oracle.structure_emb(rnn_type, sds, clip_emb, cap_emb, b[8], b[9],
                                           vid_ctx, para_ctx)
  vid_nc, para_nc = oracle.structure_emb(rnn_type, sds, clip_emb, cap_emb, b[8], b[9])
  for nm, v in [('clip_emb', clip_emb), ('cap_emb', cap_emb), ('word', word),
                ('vid_context', vid_ctx), ('para_context', para_ctx), ('vid_emb', vid_emb),
                ('para_emb', para_emb), ('vid_emb_noctx', vid_nc), ('para_emb_noctx', para_nc)]:
    np.testing.assert_allclose(v, g['fwd.' + nm], atol=TOL, rtol=0, err_msg=nm)

  res = oracle.encode_data(rnn_type, sds, batches, margin=0.2, max_violation=False, norm=False)
  for i, nm in enumerate(['vid_embs', 'para_embs', 'clip_embs', 'cap_embs', 'vid_contexts',
                          'para_contexts']):
    np.testing.assert_allclose(res[i], g['enc.' + nm], atol=TOL, rtol=0, err_msg=nm)
  assert list(res[6]) == list(g['enc.num_clips_total'])
  np.testing.assert_allclose(res[8], g['enc.test_losses'], rtol=1e-5, atol=1e-6)
  for nm, fn in [('i2t', oracle.i2t), ('t2i', oracle.t2i)]:
    rep, top1, ranks = fn(res[0], res[1])
    np.testing.assert_array_equal(ranks, g['enc.%s.ranks' % nm])
    np.testing.assert_array_equal(top1, g['enc.%s.top1' % nm])


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_train_loss_meters(oracle, rnn_type):
  g = load_golden('model_%s.npz' % rnn_type)
  sds = golden_state_dicts(g)
  batch = golden_batches(g)[1]
  for mv in (0, 1):
    for nm in (0, 1):
      tag = 'train.mv%d.norm%d' % (mv, nm)
      log, _ = oracle.train_losses(rnn_type, sds, batch, margin=0.2, max_violation=bool(mv),
                                   norm=bool(nm), low_level_loss=True)
      assert [l[0] for l in log] == [str(s) for s in g[tag + '.names']]
      np.testing.assert_allclose([l[1] for l in log], g[tag + '.values'], rtol=1e-5, atol=2e-6)
      assert [l[2] for l in log] == list(g[tag + '.n'])


# ------------------------------------------------------------------------------------------
# backward: the oracle's written-out derivatives against the reference's autograd gradients
# ------------------------------------------------------------------------------------------
GTOL = dict(rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize('cls,rnn_type', [('Attention', 'attention'), ('Maxout', 'maxout'),
                                          ('Seq2Seq', 'seq2seq')])
@pytest.mark.parametrize('tag', ['ragged', 'equal', 'one'])
def test_layer_backward(oracle, cls, rnn_type, tag):
  g = load_golden('layers.npz')
  p = {k[len(cls) + 4:]: g[k] for k in g.files if k.startswith(cls + '.sd.')}
  key = '%s.%s' % (cls, tag)
  x, lens, h0, w = g[key + '.x'], g[key + '.lens'], g[key + '.h0'], g[key + '.bwd.w']
  out, c = oracle.pooled_gru_forward_cache(rnn_type, x, lens, p, h0)
  np.testing.assert_allclose(out, g[key + '.out_h0'], atol=TOL, rtol=0)
  grads, dx, dh0 = oracle.pooled_gru_backward(c, w.astype(np.float64))
  np.testing.assert_allclose(dx, g[key + '.bwd.dx'][:, :dx.shape[1]], **GTOL)
  np.testing.assert_allclose(dh0, g[key + '.bwd.dh0'], **GTOL)
  for k in g.files:
    if k.startswith(key + '.bwd.grad.'):
      name = k[len(key + '.bwd.grad.'):]
      np.testing.assert_allclose(grads[name], g[k], err_msg=name, **GTOL)


@pytest.mark.parametrize('n', [5, 16, 37])
def test_loss_backward(oracle, n):
  g = load_golden('loss.npz')
  a, b = g['n%d.a' % n].astype(np.float64), g['n%d.b' % n].astype(np.float64)
  an, bn = oracle.l2_normalize(a, np.float64), oracle.l2_normalize(b, np.float64)
  for mv in (0, 1):
    for nm in (0, 1):
      tag = 'n%d.mv%d.norm%d' % (n, mv, nm)
      ga, gb = oracle.contrastive_loss_backward(an, bn, 0.2, bool(mv), bool(nm))
      np.testing.assert_allclose(oracle.l2_normalize_backward(a, ga), g[tag + '.da'], **GTOL)
      np.testing.assert_allclose(oracle.l2_normalize_backward(b, gb), g[tag + '.db'], **GTOL)
      g1, g2 = oracle.contrastive_loss_backward(an, an, 0.2, bool(mv), bool(nm))
      np.testing.assert_allclose(oracle.l2_normalize_backward(a, g1 + g2), g[tag + '.da_self'],
                                 **GTOL)


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_train_step_gradients(oracle, rnn_type):
  g = load_golden('model_%s.npz' % rnn_type)
  sds = golden_state_dicts(g)
  batch = golden_batches(g)[1]
  for mv in (0, 1):
    for nm in (0, 1):
      tag = 'train.mv%d.norm%d' % (mv, nm)
      grads = oracle.train_step_grads(rnn_type, sds, batch, margin=0.2, max_violation=bool(mv),
                                      norm=bool(nm), low_level_loss=True)
      for i in range(4):
        for k, v in grads[i].items():
          np.testing.assert_allclose(v, g['%s.grad%d.%s' % (tag, i, k)], rtol=5e-4, atol=5e-6,
                                     err_msg='%s enc%d %s' % (tag, i, k))


def recon_golden(lowest):
  g = load_golden('model_recon.npz')
  tag = 'lowest%d' % lowest
  n_sd = 8 if lowest else 6
  sds = [dict() for _ in range(n_sd)]
  for k in g.files:
    if k.startswith(tag + '.sd'):
      i, key = k[len(tag) + 3:].split('.', 1)
      sds[int(i)][key] = g[k]
  p = tag + '.batch0.'
  nc = tuple(int(c) for c in g[p + 'num_clips'])
  batch = (g[p + 'clips'], g[p + 'captions'], g[p + 'videos'], g[p + 'paragraphs'],
           g[p + 'lengths_clip'], g[p + 'lengths_cap'], g[p + 'lengths_video'],
           g[p + 'lengths_paragraph'], nc, tuple(int(c) for c in g[p + 'num_caps']),
           tuple(range(len(nc))), tuple('v%d' % j for j in range(len(nc))))
  return g, tag, sds, batch


@pytest.mark.parametrize('lowest', [0, 1])
def test_reconstruction_train_step(oracle, lowest):
  """--reconstruct_loss (+ --lowest_reconstruct_loss): logger triples and every gradient."""
  g, tag, sds, batch = recon_golden(lowest)
  log, total, grads = oracle.train_step_recon('maxout', sds, batch, margin=0.2, norm=True,
                                              low_level_loss=True, lowest=bool(lowest))
  assert [l[0] for l in log] == [str(s) for s in g[tag + '.names']]
  np.testing.assert_allclose([l[1] for l in log], g[tag + '.values'], rtol=2e-5, atol=2e-6)
  assert [l[2] for l in log] == list(g[tag + '.n'])
  for i, gd in enumerate(grads):
    for k, v in gd.items():
      want = g['%s.grad%d.%s' % (tag, i, k)]
      tol = 5e-4 * np.abs(want).max() + 1e-9
      assert np.abs(v - want).max() <= tol, (tag, i, k, np.abs(v - want).max(), tol)


@pytest.mark.parametrize('n', [11, 9])
def test_groupwise_loss(oracle, n):
  g = load_golden('loss.npz')
  a, b = g['gw%d.a' % n].astype(np.float64), g['gw%d.b' % n].astype(np.float64)
  nc, ncap = list(g['gw%d.num_clips' % n]), list(g['gw%d.num_caps' % n])
  an, bn = oracle.l2_normalize(a, np.float64), oracle.l2_normalize(b, np.float64)
  for mv in (0, 1):
    for nm in (0, 1):
      tag = 'gw%d.mv%d.norm%d' % (n, mv, nm)
      loss, ga, gb = oracle.groupwise_contrastive_loss(an, bn, nc, ncap, 0.2, bool(mv), bool(nm),
                                                       want_grad=True)
      np.testing.assert_allclose(loss, g[tag + '.loss'], rtol=2e-6, atol=1e-6)
      np.testing.assert_allclose(oracle.l2_normalize_backward(a, ga), g[tag + '.da'], **GTOL)
      np.testing.assert_allclose(oracle.l2_normalize_backward(b, gb), g[tag + '.db'], **GTOL)


# ---- the torch-CPU baseline (oracle/cmhse_torch_cpu.py: bench.py's `cpu_baseline`) ----------------
@pytest.fixture(scope='module')
def torch_cpu():
  import os
  import sys
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
  import cmhse_torch_cpu
  return cmhse_torch_cpu


@pytest.mark.parametrize('cls,rnn_type', [('Attention', 'attention'), ('Maxout', 'maxout'),
                                          ('Seq2Seq', 'seq2seq')])
@pytest.mark.parametrize('tag', ['ragged', 'equal', 'one'])
def test_torch_cpu_layers_forward(torch_cpu, cls, rnn_type, tag):
  import torch
  g = load_golden('layers.npz')
  p = {k[len(cls) + 4:]: g[k] for k in g.files if k.startswith(cls + '.sd.')}
  enc = torch_cpu.Encoder(rnn_type, p)
  key = '%s.%s' % (cls, tag)
  x, lens, h0 = torch.from_numpy(g[key + '.x']), g[key + '.lens'], torch.from_numpy(g[key + '.h0'])
  with torch.no_grad():
    np.testing.assert_allclose(enc(x, lens).numpy(), g[key + '.out'], atol=TOL, rtol=0)
    np.testing.assert_allclose(enc(x, lens, h0).numpy(), g[key + '.out_h0'], atol=TOL, rtol=0)


@pytest.mark.parametrize('rnn_type', ['attention', 'maxout', 'seq2seq'])
def test_torch_cpu_encode_and_rank(torch_cpu, rnn_type):
  g = load_golden('model_%s.npz' % rnn_type)
  res = torch_cpu.encode_data(rnn_type, golden_state_dicts(g), golden_batches(g), margin=0.2,
                              max_violation=False, norm=False)
  for i, nm in enumerate(['vid_embs', 'para_embs', 'clip_embs', 'cap_embs', 'vid_contexts',
                          'para_contexts']):
    np.testing.assert_allclose(res[i], g['enc.' + nm], atol=TOL, rtol=0, err_msg=nm)
  np.testing.assert_allclose(res[8], g['enc.test_losses'], rtol=1e-5, atol=1e-6)
  for nm, fn in [('i2t', torch_cpu.i2t), ('t2i', torch_cpu.t2i)]:
    rep, top1, ranks = fn(res[0], res[1])
    np.testing.assert_array_equal(ranks, g['enc.%s.ranks' % nm])
    np.testing.assert_array_equal(top1, g['enc.%s.top1' % nm])


def test_argmax_route_override(oracle):
  """oracle.apply_argmax_route (what the full-dimension maxout gradient tests hand the oracle: the
  routing the fp32 forward used).  Its own routing handed back changes nothing and reports no
  difference; a routing moved off the arg-max for one (sequence, unit) pair is reported with the
  fp64 gap it jumps across and changes the gradients; a step outside the sequence is refused."""
  g = load_golden('model_maxout.npz')
  sds = golden_state_dicts(g)
  batch = golden_batches(g)[1]
  kw = dict(margin=0.2, max_violation=False, norm=True, low_level_loss=True)
  base = oracle.train_step_grads('maxout', sds, batch, **kw)
  # recover the oracle's own routing of the clip encoder
  _, c = oracle.pooled_gru_forward_cache('maxout', batch[0], batch[4], sds[0], None, np.float64)
  own = c['argmax'].copy()
  rep = {}
  same = oracle.train_step_grads('maxout', sds, batch, argmax_route={'clip': own}, route_report=rep, **kw)
  assert rep['clip'][0] == 0 and rep['clip'][2] == own.size
  for i in range(4):
    for k in base[i]:
      np.testing.assert_array_equal(same[i][k], base[i][k])
  lens = np.asarray(batch[4])
  s = int(np.argmax(lens > 1))
  moved = own.copy()
  moved[s, 0] = (own[s, 0] + 1) % lens[s]
  rep = {}
  other = oracle.train_step_grads('maxout', sds, batch, argmax_route={'clip': moved}, route_report=rep, **kw)
  assert rep['clip'][0] == 1 and rep['clip'][1] > 0
  assert np.abs(other[0]['rnn.rnn.weight_hh_l0'] - base[0]['rnn.rnn.weight_hh_l0']).max() > 0
  bad = own.copy()
  bad[s, 0] = lens[s]
  with pytest.raises(ValueError):
    oracle.train_step_grads('maxout', sds, batch, argmax_route={'clip': bad}, **kw)


@pytest.mark.parametrize('a', [6, 9, 12])
def test_attention_epsilon_quirk_cases(oracle, a):
  """tests/golden/quirks.npz: energies of about -a, where the 0.0001 of layers.py:158-162 is as
  large as exp(e) itself.  The oracle follows the reference; an oracle WITHOUT the epsilon (the
  textbook masked softmax) misses the same vectors by more than a thousand tolerances — what makes
  these cases a detector for the quirk on the HIP side too (tests/test_quirks_tight.py)."""
  g = load_golden('quirks.npz')
  tag = 'a%d' % a
  p = {k[len(tag) + 4:]: g[k] for k in g.files if k.startswith(tag + '.sd.')}
  x, lens, h0 = g['x'], g['lens'], g['h0']
  np.testing.assert_allclose(oracle.attention_forward(x, lens, p), g[tag + '.out'], atol=TOL, rtol=0)
  np.testing.assert_allclose(oracle.attention_forward(x, lens, p, h0), g[tag + '.out_h0'], atol=TOL, rtol=0)
  # the textbook softmax: weights of a length-1 sequence are exactly 1 -> out = h_1
  hs = oracle.gru_forward(x, lens, p['rnn.rnn.weight_ih_l0'], p['rnn.rnn.weight_hh_l0'],
                          p['rnn.rnn.bias_ih_l0'], p['rnn.rnn.bias_hh_l0'])
  hs = hs[0]
  one = np.flatnonzero(lens == 1)
  miss = np.abs(hs[one, 0] - g[tag + '.out'][one]).max()
  assert miss > 1000 * TOL, miss


def test_bidirectional_seq2seq_golden(oracle):
  """layers.Seq2Seq(rnn_bidirectional=True) (layers.py:31-34,58-59): the final states of the forward
  GRU and of the `_reverse` GRU over the time-reversed valid steps, side by side — restated with the
  oracle's unidirectional GRU, against the reference's output.  Maxout ignores the flag upstream."""
  g = load_golden('quirks.npz')
  p = {k[len('bidir.sd.'):]: g[k] for k in g.files if k.startswith('bidir.sd.')}
  x, lens = g['x'], g['lens']
  _, h_f = oracle.gru_forward(x, lens, p['rnn.rnn.weight_ih_l0'], p['rnn.rnn.weight_hh_l0'],
                              p['rnn.rnn.bias_ih_l0'], p['rnn.rnn.bias_hh_l0'])
  xr = np.zeros_like(x)
  for s, l in enumerate(lens):
    xr[s, :l] = x[s, :l][::-1]
  _, h_b = oracle.gru_forward(xr, lens, p['rnn.rnn.weight_ih_l0_reverse'], p['rnn.rnn.weight_hh_l0_reverse'],
                              p['rnn.rnn.bias_ih_l0_reverse'], p['rnn.rnn.bias_hh_l0_reverse'])
  np.testing.assert_allclose(np.concatenate([h_f, h_b], 1), g['bidir.out'], atol=TOL, rtol=0)
  pm = {k[len('bidir_maxout.sd.'):]: g[k] for k in g.files if k.startswith('bidir_maxout.sd.')}
  assert not any(k.endswith('_reverse') for k in pm)
  np.testing.assert_allclose(oracle.maxout_forward(x, lens, pm), g['bidir_maxout.out'], atol=TOL, rtol=0)
