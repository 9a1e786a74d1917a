"""The HIP path against the golden vectors the REFERENCE produced (tests/golden, tools/make_golden.py).
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
