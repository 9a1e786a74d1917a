"""Reference quirks pinned on the HIP side (VERDICT r05 "what's weak" 1, "missing" 3).

The 1e-4 contract cannot see a kernel that drops the `+ 0.0001` of the attention softmax
(layers.py:158-162) on ordinary inputs; these cases can: golden vectors made by the reference in
which the epsilon carries most of the answer, held at GOLDEN_TOL, plus `norm=True` dividing by
n * m (loss.py:114-115) and BASELINE configs[0] at its stated dimensions.
"""
import argparse

import numpy as np
import pytest
import torch

from conftest import GOLDEN_TOL, assert_emb_close, load_golden

pytestmark = pytest.mark.gpu


def _quirk_layer(g, tag, dev):
  from cmhse_amd import layers
  layer = layers.Attention(24, 32)
  layer.load_state_dict({k[len(tag) + 8:]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith(tag + '.sd.rnn.')})
  return layer.to(dev)


@pytest.mark.parametrize('a', [6, 9, 12])
def test_attention_softmax_epsilon_on_short_sequences_with_small_energies(dev, a):
  """tests/golden/quirks.npz (made by the reference): energies of about -a, so exp(e) is of the
  order of the 0.0001 added to the denominator; length-1 sequences come out as 0.96 / 0.55 / 0.06
  of their hidden state.  Forward at the tight bar, and the reference-autograd gradients (the
  epsilon is in the backward's denominator too, bwd.hip)."""
  g = load_golden('quirks.npz')
  tag = 'a%d' % a
  layer = _quirk_layer(g, tag, dev)
  x = torch.from_numpy(g['x']).to(dev)
  lens = torch.from_numpy(g['lens'])
  h0 = torch.from_numpy(g['h0']).to(dev)
  with torch.no_grad():
    y = layer(x, lens).cpu().numpy()
    y0 = layer(x, lens, h0).cpu().numpy()
  assert_emb_close(y, g[tag + '.out'], tag)
  assert_emb_close(y0, g[tag + '.out_h0'], tag + ' h0')
  # the case is a detector: without the epsilon a length-1 row would be h_1 itself, i.e. the golden
  # row divided by its weight exp(e) / (exp(e) + 1e-4) < 1 — far outside the tight bar
  one = np.flatnonzero(g['lens'] == 1)
  wgt = np.exp(-float(a)) / (np.exp(-float(a)) + 1e-4)
  assert np.abs(g[tag + '.out'][one] * (1.0 / wgt - 1.0)).max() > 100 * GOLDEN_TOL
  xg = x.clone().requires_grad_(True)
  hg = h0.clone().requires_grad_(True)
  layer.zero_grad()
  (layer(xg, lens, hg) * torch.from_numpy(g['w']).to(dev)).sum().backward()

  def close(got, want, name):
    scale = max(1e-6, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    assert err < 2e-5, (name, err)
  close(xg.grad.cpu().numpy(), g[tag + '.bwd.dx'], 'dx')
  close(hg.grad.cpu().numpy(), g[tag + '.bwd.dh0'], 'dh0')
  for pn, pp in layer.named_parameters():
    close(pp.grad.cpu().numpy(), g[tag + '.bwd.grad.rnn.' + pn], pn)


@pytest.mark.parametrize('n', [5, 16, 37])
def test_norm_divides_by_n_times_m_at_a_tight_bar(dev, n):
  """loss.py:114-115: `norm=True` divides by n * m (not n).  The reference's values at 2e-6
  relative — `norm` and plain sums in the exact ratio n * n."""
  from cmhse_amd.loss import ContrastiveLoss
  g = load_golden('loss.npz')
  an = torch.from_numpy(g['n%d.a_norm' % n]).to(dev)
  bn = torch.from_numpy(g['n%d.b_norm' % n]).to(dev)
  for mv in (0, 1):
    vals = {}
    for nm in (0, 1):
      crit = ContrastiveLoss(margin=0.2, measure='cosine', max_violation=bool(mv), norm=bool(nm))
      for pair, key in (((an, bn), 'ab'), ((an, an), 'aa')):
        got = float(crit(*pair))
        want = float(g['n%d.mv%d.norm%d.%s' % (n, mv, nm, key)])
        assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (n, mv, nm, key, got, want)
        vals[(nm, key)] = got
    for key in ('ab', 'aa'):
      assert abs(vals[(0, key)] / (n * n) - vals[(1, key)]) <= 2e-6 * max(1.0, abs(vals[(1, key)]))


def test_config0_plumbing_at_its_stated_dimensions(dev, oracle):
  """BASELINE configs[0] as written: 64 videos x 4 clips x 10 frames of 500-d features, 4
  sentences x 12 words, vocabulary 13 058, batch 16, embed 1024 (activity_net/data.py:114-150
  shapes; bench.WORKLOADS['plumbing']).  encode_data's six matrices against the oracle at the tight
  bar; i2t / t2i ranks, top-1 and report against an fp64 ranking (every row the embedding error
  cannot flip; the whole report when that is every row)."""
  import sys
  import os
  from conftest import REPO
  sys.path.insert(0, os.path.join(REPO, 'tools'))
  from bench_common import WORKLOADS, make_opt
  from cmhse_amd import synthetic
  from cmhse_amd.evaluation import encode_data, i2t, t2i, report_from_ranks
  from cmhse_amd.model import VSE
  wl = WORKLOADS['plumbing']
  assert (wl['n_videos'], wl['batch'], wl['img_dim'], wl['vocab']) == (64, 16, 500, 13058)
  opt = make_opt(wl, 'attention', 1024)
  torch.manual_seed(1)
  model = VSE(opt)
  spec = synthetic.uniform_spec(64, clips=4, frames=10, words=12)
  batches = synthetic.make_batches(spec, 16, 500, 13058, seed=0)
  assert len(batches) == 4 and batches[0][0].shape == (64, 10, 500) and batches[0][1].shape == (64, 12)
  assert batches[0][2].shape == (16, 40, 500) and batches[0][3].shape == (16, 48)
  res = encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  sds = [{k: v.detach().cpu().numpy() for k, v in sd.items()} for sd in model.state_dict(opt)]
  nb = [tuple(t.numpy() if isinstance(t, torch.Tensor) else t for t in b) for b in batches]
  want = oracle.encode_data('attention', sds, nb, margin=0.2)
  err = 0.0
  for i, nm in enumerate(['vid', 'para', 'clip', 'cap', 'vid_ctx', 'para_ctx']):
    assert res[i].shape == want[i].shape == ((64, 1024) if i in (0, 1, 4, 5) else (256, 1024))
    assert_emb_close(res[i], want[i], nm)
    err = max(err, float(np.abs(res[i] - want[i]).max()))
  assert list(res[6]) == [4] * 64
  v64, p64 = want[0].astype(np.float64), want[1].astype(np.float64)
  for fn, q, gal in [(i2t, v64, p64), (t2i, p64, v64)]:
    rep, top1, ranks = fn(res[0], res[1])
    d = q @ gal.T
    dii = np.diag(d)
    r64 = (d > dii[:, None]).sum(1)
    gap = np.abs(d - dii[:, None])
    np.fill_diagonal(gap, np.inf)
    ok = gap.min(1) > 4 * err + 2e-6
    np.testing.assert_array_equal(ranks[ok], r64[ok])
    assert ok.mean() > 0.5
    if ok.all():
      assert rep == report_from_ranks(r64.astype(np.float64))


def test_bidirectional_seq2seq_vs_the_reference(dev):
  """layers.Seq2Seq(rnn_bidirectional=True).forward (layers.py:47-66): [S, 2H] = forward | reverse
  final states, and the reference-autograd gradients of sum(out * w) wrt the input and all eight GRU
  parameters; Maxout with the flag set equals the unidirectional layer (upstream ignores it)."""
  from cmhse_amd import layers
  g = load_golden('quirks.npz')
  layer = layers.Seq2Seq(24, 32, rnn_bidirectional=True)
  layer.load_state_dict({k[len('bidir.sd.rnn.'):]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith('bidir.sd.rnn.')})
  layer = layer.to(dev)
  x = torch.from_numpy(g['x']).to(dev)
  lens = torch.from_numpy(g['lens'])
  with torch.no_grad():
    y = layer(x, lens).cpu().numpy()
  assert y.shape == (len(g['lens']), 64)
  assert_emb_close(y, g['bidir.out'], 'bidirectional Seq2Seq')
  xg = x.clone().requires_grad_(True)
  layer.zero_grad()
  (layer(xg, lens) * torch.from_numpy(g['bidir.w']).to(dev)).sum().backward()

  def close(got, want, name):
    err = float(np.abs(got - want).max()) / max(1e-6, float(np.abs(want).max()))
    assert err < 2e-5, (name, err)
  close(xg.grad.cpu().numpy(), g['bidir.bwd.dx'], 'dx')
  for pn, pp in layer.named_parameters():
    close(pp.grad.cpu().numpy(), g['bidir.bwd.grad.rnn.' + pn], pn)
  mo = layers.Maxout(24, 32, rnn_bidirectional=True)
  mo.load_state_dict({k[len('bidir_maxout.sd.rnn.'):]: torch.from_numpy(g[k]) for k in g.files
                      if k.startswith('bidir_maxout.sd.rnn.')})
  with torch.no_grad():
    assert_emb_close(mo.to(dev)(x, lens).cpu().numpy(), g['bidir_maxout.out'], 'Maxout ignores the flag')
