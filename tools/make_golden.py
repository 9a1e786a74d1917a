#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (read-only, /root/reference).

Runs only in the build container (the reference does not exist on the GPU box).  The reference is
imported unmodified; what it needs from the environment is supplied in memory:
  1. stub modules for absent imports (IPython, torchvision, nltk, h5py, tensorboard_logger);
  2. `.cuda()` made an identity (the reference calls it unconditionally, layers.py:97,117,156);
  3. decoder/layers.py:16 is a Python-2 print statement: that module is compiled from its text
     with that one line blanked and registered before `import model`;
  4. vocab/<data>_w2v_total.npz is loaded relative to cwd (model.py:90): a synthetic table is
     written to a scratch cwd;
  5. evaluation.LogCollector.__str__ uses .iteritems() (evaluation.py:63): replaced in memory.
Only inputs/outputs (data) are written to tests/golden/; no reference source is copied.

Usage: python tools/make_golden.py        (writes tests/golden/*.npz)
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REPO)

from cmhse_amd import synthetic  # noqa: E402


def install_shims(vocab_size, word_dim, scratch):
  for name in ['IPython', 'torchvision', 'torchvision.models', 'torchvision.transforms', 'nltk',
               'h5py', 'tensorboard_logger']:
    if name not in sys.modules:
      sys.modules[name] = types.ModuleType(name)
  sys.modules['IPython'].embed = lambda *a, **k: None
  sys.modules['torchvision'].models = sys.modules['torchvision.models']
  sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
  torch.Tensor.cuda = lambda self, *a, **k: self
  torch.nn.Module.cuda = lambda self, *a, **k: self

  sys.path.insert(0, REF)
  # (3) decoder/layers.py with the py2 print line neutralised
  import decoder  # noqa: F401  (namespace package under /root/reference)
  src = open(os.path.join(REF, 'decoder', 'layers.py')).read().split('\n')
  src = [('    pass' if ln.strip().startswith('print ') else ln) for ln in src]
  mod = types.ModuleType('decoder.layers')
  mod.__file__ = os.path.join(REF, 'decoder', 'layers.py')
  exec(compile('\n'.join(src), mod.__file__, 'exec'), mod.__dict__)
  sys.modules['decoder.layers'] = mod
  import decoder.model as dmodel
  dmodel.Seq2Seq_Decode = mod.Seq2Seq_Decode

  # (4) scratch cwd with a synthetic word-vector table
  os.makedirs(os.path.join(scratch, 'vocab'), exist_ok=True)
  rng = np.random.RandomState(1234)
  table = (0.1 * rng.standard_normal((vocab_size, word_dim))).astype(np.float32)
  np.savez(os.path.join(scratch, 'vocab', 'anet_precomp_w2v_total.npz'), table)
  os.chdir(scratch)

  import layers as ref_layers
  import loss as ref_loss
  import model as ref_model
  import evaluation as ref_eval

  def _str(self):
    return '  '.join(k + ' ' + str(v) for k, v in self.meters.items())
  ref_eval.LogCollector.__str__ = _str
  return ref_layers, ref_loss, ref_model, ref_eval


def sd_np(module, prefix=''):
  return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def randomize_biases(module, gen):
  """The reference zero-initialises GRU biases (layers.py:38-39); released checkpoints have
  trained, non-zero biases, so goldens use random ones to exercise the bias terms."""
  for name, p in module.named_parameters():
    if 'bias' in name:
      with torch.no_grad():
        p.copy_(0.1 * torch.randn(p.shape, generator=gen))


def make_opt(**kw):
  opt = argparse.Namespace(
      margin=0.2, word_dim=12, embed_size=32, grad_clip=0.0, learning_rate=0.001,
      max_violation=False, img_dim=24, measure='cosine', rnn_type='maxout',
      img_first_size=32, cap_first_size=32, weight_recon=0.0005, lowest_weight_recon=0.0001,
      decode_rnn_type='seq2seq', low_level_loss=False, weak_low_level_loss=False,
      reconstruct_loss=False, lowest_reconstruct_loss=False, norm=False,
      data_name='anet_precomp', vocab_size=60)
  for k, v in kw.items():
    setattr(opt, k, v)
  return opt


def golden_layers(ref_layers, out):
  """layers.{Attention,Maxout,Seq2Seq}.forward: ragged lengths incl. len=1, with/without h0."""
  gen = torch.Generator().manual_seed(7)
  I, H = 24, 32
  cases = {}
  for cls_name in ['Attention', 'Maxout', 'Seq2Seq']:
    torch.manual_seed(11)
    layer = getattr(ref_layers, cls_name)(I, H)
    randomize_biases(layer, gen)
    sd = sd_np(layer, 'rnn.')      # keys as they appear under an Encoder* wrapper
    for tag, lens in [('ragged', [5, 1, 9, 3, 9, 2, 7]), ('equal', [4, 4, 4]), ('one', [1, 1])]:
      S, T = len(lens), max(lens)
      x = torch.zeros(S, T, I)
      for i, l in enumerate(lens):
        x[i, :l] = torch.randn(l, I, generator=gen)
      h0 = 0.5 * torch.randn(S, H, generator=gen)
      with torch.no_grad():
        y = layer(x, torch.tensor(lens))
        y_h0 = layer(x, torch.tensor(lens), h0)
      key = '%s.%s' % (cls_name, tag)
      cases[key + '.x'] = x.numpy()
      cases[key + '.lens'] = np.array(lens, dtype=np.int64)
      cases[key + '.h0'] = h0.numpy()
      cases[key + '.out'] = y.numpy()
      cases[key + '.out_h0'] = y_h0.numpy()
      # gradients of sum(out * w) wrt parameters, input and h0 (reference autograd)
      w = torch.randn(S, H, generator=gen)
      xg = x.clone().requires_grad_(True)
      hg = h0.clone().requires_grad_(True)
      layer.zero_grad()
      (layer(xg, torch.tensor(lens), hg) * w).sum().backward()
      cases[key + '.bwd.w'] = w.numpy()
      cases[key + '.bwd.dx'] = xg.grad.numpy()
      cases[key + '.bwd.dh0'] = hg.grad.numpy()
      for pn, pp in layer.named_parameters():
        cases[key + '.bwd.grad.rnn.' + pn] = pp.grad.detach().numpy().copy()
    for k, v in sd.items():
      cases['%s.sd.%s' % (cls_name, k)] = v
  np.savez_compressed(os.path.join(out, 'layers.npz'), **cases)


def golden_quirks(ref_layers, out):
  """Cases in which a reference quirk carries most of the answer, so that an implementation without
  it misses by far more than fp32 rounding: layers.Attention's softmax adds
  0.0001 to the denominator and does not subtract the maximum (layers.py:158-162).  With strongly
  NEGATIVE energies (lin.bias = +3 saturates the tanh, att_w = -a / H gives e ~ -a) exp(e) is of
  the order of that 0.0001: a length-1 sequence comes out as h * exp(e) / (exp(e) + 1e-4) — 0.96,
  0.55 and 0.06 of h for a = 6, 9, 12 — where a softmax without the epsilon returns h itself."""
  gen = torch.Generator().manual_seed(23)
  I, H = 24, 32
  cases = {}
  lens = [1, 1, 2, 3, 1, 5]
  S, T = len(lens), max(lens)
  x = torch.zeros(S, T, I)
  for i, l in enumerate(lens):
    x[i, :l] = torch.randn(l, I, generator=gen)
  h0 = 0.5 * torch.randn(S, H, generator=gen)
  w = torch.randn(S, H, generator=gen)
  cases['x'] = x.numpy(); cases['lens'] = np.array(lens, dtype=np.int64)
  cases['h0'] = h0.numpy(); cases['w'] = w.numpy()
  for a in [6, 9, 12]:
    torch.manual_seed(40 + a)
    layer = ref_layers.Attention(I, H)
    randomize_biases(layer, gen)
    with torch.no_grad():
      layer.lin.bias.fill_(3.0)
      layer.att_w.weight.fill_(-float(a) / H)
    tag = 'a%d' % a
    for k, v in sd_np(layer, 'rnn.').items():
      cases['%s.sd.%s' % (tag, k)] = v
    with torch.no_grad():
      cases[tag + '.out'] = layer(x, torch.tensor(lens)).numpy()
      cases[tag + '.out_h0'] = layer(x, torch.tensor(lens), h0).numpy()
    xg = x.clone().requires_grad_(True)
    hg = h0.clone().requires_grad_(True)
    layer.zero_grad()
    (layer(xg, torch.tensor(lens), hg) * w).sum().backward()
    cases[tag + '.bwd.dx'] = xg.grad.numpy()
    cases[tag + '.bwd.dh0'] = hg.grad.numpy()
    for pn, pp in layer.named_parameters():
      cases[tag + '.bwd.grad.rnn.' + pn] = pp.grad.detach().numpy().copy()
  # rnn_bidirectional=True: upstream only Seq2Seq.forward(q_emb, q_len) works with it (layers.py:58-59:
  # the two directions' final states side by side); Maxout ignores the flag (:167-172)
  torch.manual_seed(77)
  bi = ref_layers.Seq2Seq(I, H, rnn_bidirectional=True)
  randomize_biases(bi, gen)
  for k, v in sd_np(bi, 'rnn.').items():
    cases['bidir.sd.' + k] = v
  with torch.no_grad():
    cases['bidir.out'] = bi(x, torch.tensor(lens)).numpy()
  w2 = torch.randn(S, 2 * H, generator=gen)
  cases['bidir.w'] = w2.numpy()
  xg = x.clone().requires_grad_(True)
  bi.zero_grad()
  (bi(xg, torch.tensor(lens)) * w2).sum().backward()
  cases['bidir.bwd.dx'] = xg.grad.numpy()
  for pn, pp in bi.named_parameters():
    cases['bidir.bwd.grad.rnn.' + pn] = pp.grad.detach().numpy().copy()
  torch.manual_seed(78)
  mo = ref_layers.Maxout(I, H, rnn_bidirectional=True)
  randomize_biases(mo, gen)
  for k, v in sd_np(mo, 'rnn.').items():
    cases['bidir_maxout.sd.' + k] = v
  with torch.no_grad():
    cases['bidir_maxout.out'] = mo(x, torch.tensor(lens)).numpy()
  np.savez_compressed(os.path.join(out, 'quirks.npz'), **cases)


def golden_loss(ref_loss, out):
  """loss.ContrastiveLoss x {max_violation} x {norm}, (im,s) and CL(x,x); F.normalize;
  decoder.loss.EuclideanLoss."""
  import torch.nn.functional as F
  gen = torch.Generator().manual_seed(3)
  cases = {}
  for n in [5, 16, 37]:
    a = torch.randn(n, 32, generator=gen)
    b = a + 0.8 * torch.randn(n, 32, generator=gen)
    an, bn = F.normalize(a), F.normalize(b)
    cases['n%d.a' % n] = a.numpy(); cases['n%d.b' % n] = b.numpy()
    cases['n%d.a_norm' % n] = an.numpy(); cases['n%d.b_norm' % n] = bn.numpy()
    for mv in [False, True]:
      for nm in [False, True]:
        crit = ref_loss.ContrastiveLoss(margin=0.2, measure='cosine', max_violation=mv, norm=nm)
        tag = 'n%d.mv%d.norm%d' % (n, int(mv), int(nm))
        cases[tag + '.ab'] = np.float32(crit(an, bn).item())
        cases[tag + '.aa'] = np.float32(crit(an, an).item())
    cases['n%d.scores' % n] = ref_loss.cosine_sim(an, bn).numpy()
    for mv in [False, True]:
      for nm in [False, True]:
        crit = ref_loss.ContrastiveLoss(margin=0.2, measure='cosine', max_violation=mv, norm=nm)
        ag = a.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
        crit(F.normalize(ag), F.normalize(bg)).backward()
        tag = 'n%d.mv%d.norm%d' % (n, int(mv), int(nm))
        cases[tag + '.da'] = ag.grad.numpy(); cases[tag + '.db'] = bg.grad.numpy()
        ag2 = a.clone().requires_grad_(True)
        crit(F.normalize(ag2), F.normalize(ag2)).backward()
        cases[tag + '.da_self'] = ag2.grad.numpy()
  # GroupWiseContrastiveLoss (loss.py:15-72): block max / mean of clip x caption scores
  for n, counts in [(11, [3, 1, 4, 3]), (9, [2, 2, 5])]:
    a = torch.randn(n, 32, generator=gen)
    b = a + 0.8 * torch.randn(n, 32, generator=gen)
    caps = counts[::-1] if n == 9 else counts      # also a case with num_clips != num_caps blocks
    if sum(caps) != n:
      caps = counts
    cases['gw%d.a' % n] = a.numpy(); cases['gw%d.b' % n] = b.numpy()
    cases['gw%d.num_clips' % n] = np.array(counts); cases['gw%d.num_caps' % n] = np.array(caps)
    for mv in [False, True]:
      for nm in [False, True]:
        crit = ref_loss.GroupWiseContrastiveLoss(margin=0.2, measure='cosine', max_violation=mv,
                                                 norm=nm)
        ag = a.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
        val = crit(F.normalize(ag), F.normalize(bg), counts, caps)
        val.backward()
        tag = 'gw%d.mv%d.norm%d' % (n, int(mv), int(nm))
        cases[tag + '.loss'] = np.float32(val.item())
        cases[tag + '.da'] = ag.grad.numpy(); cases[tag + '.db'] = bg.grad.numpy()
  from decoder.loss import EuclideanLoss
  a = torch.randn(13, 24, generator=gen); b = torch.randn(13, 24, generator=gen)
  cases['euclid.a'] = a.numpy(); cases['euclid.b'] = b.numpy()
  cases['euclid.norm1'] = np.float32(EuclideanLoss(norm=True)(a, b).item())
  cases['euclid.norm0'] = np.float32(EuclideanLoss(norm=False)(a, b).item())
  z = torch.zeros(3, 8); z[1] = torch.randn(8, generator=gen)
  cases['normalize.zero_rows.x'] = z.numpy()
  cases['normalize.zero_rows.y'] = F.normalize(z).numpy()
  np.savez_compressed(os.path.join(out, 'loss.npz'), **cases)


def golden_rank(ref_eval, out):
  """evaluation.i2t / t2i on tie-free data (asserted in float64 with a margin)."""
  cases = {}
  for n, dim, sigma in [(50, 32, 1.0), (203, 64, 2.0)]:
    a, b = synthetic.correlated_embeddings(n, dim, sigma, seed=n)
    d64 = a.astype(np.float64) @ b.astype(np.float64).T
    for d in (d64, d64.T):
      gap = np.abs(d - np.diag(d)[:, None]); np.fill_diagonal(gap, 1.0)
      assert gap.min() > 1e-5, 'golden ranking data must be tie-free'
      srt = np.sort(d, axis=1)
      assert (srt[:, -1] - srt[:, -2]).min() > 1e-5
    r_i2t, top1_i2t, ranks_i2t = ref_eval.i2t(a, b)
    r_t2i, top1_t2i, ranks_t2i = ref_eval.t2i(a, b)
    tag = 'n%d' % n
    cases[tag + '.images'] = a; cases[tag + '.captions'] = b
    for nm, (rep, top1, ranks) in [('i2t', (r_i2t, top1_i2t, ranks_i2t)),
                                   ('t2i', (r_t2i, top1_t2i, ranks_t2i))]:
      cases['%s.%s.top1' % (tag, nm)] = top1
      cases['%s.%s.ranks' % (tag, nm)] = ranks
      cases['%s.%s.report' % (tag, nm)] = np.array(
          [rep[k] for k in ['r1', 'r5', 'r10', 'medr', 'meanr', 'sum']], dtype=np.float64)
  np.savez_compressed(os.path.join(out, 'rank.npz'), **cases)


def batches_np(batches):
  out = {}
  for bi, b in enumerate(batches):
    names = ['clips', 'captions', 'videos', 'paragraphs', 'lengths_clip', 'lengths_cap',
             'lengths_video', 'lengths_paragraph']
    for nm, t in zip(names, b[:8]):
      out['batch%d.%s' % (bi, nm)] = t.numpy()
    out['batch%d.num_clips' % bi] = np.array(b[8], dtype=np.int64)
    out['batch%d.num_caps' % bi] = np.array(b[9], dtype=np.int64)
  return out


class MeterLog(object):
  """Stands in for evaluation.LogCollector to capture (name, value, n) triples (model.py:291)."""

  def __init__(self):
    self.calls = []

  def update(self, k, v, n=0):
    self.calls.append((k, float(v), int(n)))


def golden_model(ref_model, ref_eval, out):
  """VSE.forward_emb / structure_emb / encode_data 8-tuple / train_emb loss meters, for every
  rnn_type, on a small ragged synthetic split."""
  gen = torch.Generator().manual_seed(5)
  spec = synthetic.ragged_spec(10, seed=2)
  for rnn_type in ['attention', 'maxout', 'seq2seq']:
    opt = make_opt(rnn_type=rnn_type)
    torch.manual_seed(21)
    model = ref_model.VSE(opt)
    for enc in [model.clip_enc, model.txt_enc, model.vid_seq_enc, model.txt_seq_enc]:
      randomize_biases(enc, gen)
    batches = synthetic.make_batches(spec, 4, opt.img_dim, opt.vocab_size, seed=9)
    cases = batches_np(batches)
    cases['n_batches'] = np.int64(len(batches))
    for i, sd in enumerate(model.state_dict(opt)):
      for k, v in sd.items():
        cases['sd%d.%s' % (i, k)] = v.detach().numpy()

    # forward_emb + structure_emb on batch 0 (model.py:222-255)
    b = batches[0]
    with torch.no_grad():
      clip_emb, cap_emb, word = model.forward_emb(b[0], b[1], b[4], b[5], return_word=True)
      vid_ctx, para_ctx = model.forward_emb(b[2], b[3], b[6], b[7])
      vid_emb, para_emb = model.structure_emb(clip_emb, cap_emb, b[8], b[9], vid_ctx, para_ctx)
      vid_nc, para_nc = model.structure_emb(clip_emb, cap_emb, b[8], b[9])
    for nm, t in [('clip_emb', clip_emb), ('cap_emb', cap_emb), ('word', word),
                  ('vid_context', vid_ctx), ('para_context', para_ctx), ('vid_emb', vid_emb),
                  ('para_emb', para_emb), ('vid_emb_noctx', vid_nc), ('para_emb_noctx', para_nc)]:
      cases['fwd.' + nm] = t.numpy()

    # encode_data over the whole split (evaluation.py:80-158) + per-batch 'Letest'
    test_log = []
    orig_fl = model.forward_loss

    def fl(a, b_, name, **kw):
      loss = orig_fl(a, b_, name, **kw)
      test_log.append(loss.item())
      return loss
    model.forward_loss = fl
    with torch.no_grad():
      res = ref_eval.encode_data(opt, model, synthetic.ListLoader(batches), log_step=1000,
                                 logging=lambda *a, **k: None)
    model.forward_loss = orig_fl
    for nm, arr in zip(['vid_embs', 'para_embs', 'clip_embs', 'cap_embs', 'vid_contexts',
                        'para_contexts'], res[:6]):
      cases['enc.' + nm] = arr
    cases['enc.num_clips_total'] = np.array(res[6], dtype=np.int64)
    cases['enc.test_losses'] = np.array(test_log, dtype=np.float64)
    for nm, fn in [('i2t', ref_eval.i2t), ('t2i', ref_eval.t2i)]:
      rep, top1, ranks = fn(res[0], res[1])
      cases['enc.%s.ranks' % nm] = ranks
      cases['enc.%s.top1' % nm] = top1
      cases['enc.%s.report' % nm] = np.array(
          [rep[k] for k in ['r1', 'r5', 'r10', 'medr', 'meanr', 'sum']], dtype=np.float64)

    # train_emb loss meters (model.py:309-343) for flag combinations; weights restored after
    saved = [{k: v.clone() for k, v in sd.items()} for sd in model.state_dict(opt)]
    for mv in [False, True]:
      for nm_ in [False, True]:
        topt = make_opt(rnn_type=rnn_type, max_violation=mv, norm=nm_, low_level_loss=True)
        model.criterion = ref_model.ContrastiveLoss(margin=topt.margin, measure=topt.measure,
                                                    max_violation=mv, norm=nm_)
        model.load_state_dict(saved, topt)
        model.optimizer = torch.optim.Adam(model.params, lr=0.0)
        model.logger = MeterLog()
        model.train_start(topt)
        model.train_emb(topt, *batches[1])
        tag = 'train.mv%d.norm%d' % (int(mv), int(nm_))
        calls = [c for c in model.logger.calls if c[0].startswith('Le')]
        for i, enc in enumerate([model.clip_enc, model.txt_enc, model.vid_seq_enc,
                                 model.txt_seq_enc]):
          for pn, pp in enc.named_parameters():
            cases['%s.grad%d.%s' % (tag, i, pn)] = pp.grad.detach().numpy().copy()
        cases[tag + '.names'] = np.array([c[0] for c in calls])
        cases[tag + '.values'] = np.array([c[1] for c in calls], dtype=np.float64)
        cases[tag + '.n'] = np.array([c[2] for c in calls], dtype=np.int64)
    np.savez_compressed(os.path.join(out, 'model_%s.npz' % rnn_type), **cases)


def golden_recon(ref_model, out):
  """VSE.train_emb with --reconstruct_loss (+ --lowest_reconstruct_loss): decoders'
  state-dicts, logger triples, total-loss pieces and every parameter gradient."""
  gen = torch.Generator().manual_seed(15)
  spec = synthetic.ragged_spec(6, seed=4)
  cases = {}
  for lowest in [False, True]:
    # model.py:357 hard-codes a word width of 300 for --lowest_reconstruct_loss
    wd = 300 if lowest else 12
    opt = make_opt(rnn_type='maxout', reconstruct_loss=True, lowest_reconstruct_loss=lowest,
                   low_level_loss=True, norm=True, word_dim=wd)
    rng = np.random.RandomState(77)
    np.savez(os.path.join(os.getcwd(), 'vocab', 'anet_precomp_w2v_total.npz'),
             (0.1 * rng.standard_normal((opt.vocab_size, wd))).astype(np.float32))
    torch.manual_seed(31)
    model = ref_model.VSE(opt)
    mods = [model.clip_enc, model.txt_enc, model.vid_seq_enc, model.txt_seq_enc,
            model.vid_seq_dec, model.txt_seq_dec]
    if lowest:
      mods += [model.clip_seq_dec, model.sent_seq_dec]
    for m in mods:
      randomize_biases(m, gen)
    batches = synthetic.make_batches(spec, 6, opt.img_dim, opt.vocab_size, seed=3)
    tag = 'lowest%d' % int(lowest)
    for k, v in batches_np(batches).items():
      cases[tag + '.' + k] = v
    for i, sd in enumerate(model.state_dict(opt)):
      for k, v in sd.items():
        cases['%s.sd%d.%s' % (tag, i, k)] = v.detach().numpy().copy()
    model.optimizer = torch.optim.Adam(model.params, lr=0.0)
    model.logger = MeterLog()
    model.train_start(opt)
    model.train_emb(opt, *batches[0])
    calls = [c for c in model.logger.calls if c[0].startswith('Le')]
    cases[tag + '.names'] = np.array([c[0] for c in calls])
    cases[tag + '.values'] = np.array([c[1] for c in calls], dtype=np.float64)
    cases[tag + '.n'] = np.array([c[2] for c in calls], dtype=np.int64)
    for i, m in enumerate(mods):
      for pn, pp in m.named_parameters():
        cases['%s.grad%d.%s' % (tag, i, pn)] = pp.grad.detach().numpy().copy()
  np.savez_compressed(os.path.join(out, 'model_recon.npz'), **cases)


def golden_collate(out):
  """collate_fn of activity_net/data.py:114-150 and didemo_dev/data.py:127-165 on seeded samples."""
  import importlib
  if 'PIL' not in sys.modules:
    try:
      importlib.import_module('PIL')
    except ImportError:
      sys.modules['PIL'] = types.ModuleType('PIL')
      sys.modules['PIL'].Image = types.ModuleType('PIL.Image')
      sys.modules['PIL.Image'] = sys.modules['PIL'].Image
  rec = {}
  for tag, modname, didemo, img_dim in [('anet', 'activity_net.data', False, 6),
                                        ('didemo', 'didemo_dev.data', True, 2048)]:
    ref_data = importlib.import_module(modname)
    samples = synthetic.dataset_samples(7 if not didemo else 8, img_dim, 5, didemo)
    res = ref_data.collate_fn(samples)
    rec[tag + '_seed'] = np.int64(7 if not didemo else 8)
    rec[tag + '_img_dim'] = np.int64(img_dim)
    for k, name in enumerate(['clips', 'captions', 'videos', 'paragraphs', 'lengths_clip',
                              'lengths_cap', 'lengths_video', 'lengths_paragraph']):
      rec['%s_%s' % (tag, name)] = res[k].numpy()
      rec['%s_%s_dtype' % (tag, name)] = np.array(str(res[k].dtype))
    rec[tag + '_num_clips'] = np.asarray(res[8], dtype=np.int64)
    rec[tag + '_num_caps'] = np.asarray(res[9], dtype=np.int64)
    rec[tag + '_index'] = np.asarray(res[10], dtype=np.int64)
    rec[tag + '_last'] = (res[11].numpy() if isinstance(res[11], torch.Tensor)
                          else np.array(list(res[11])))
    rec[tag + '_last_is_tensor'] = np.bool_(isinstance(res[11], torch.Tensor))
  np.savez_compressed(os.path.join(out, 'collate.npz'), **rec)


def main():
  out = os.path.join(REPO, 'tests', 'golden')
  os.makedirs(out, exist_ok=True)
  scratch = tempfile.mkdtemp(prefix='cmhse_ref_cwd_')
  ref_layers, ref_loss, ref_model, ref_eval = install_shims(60, 12, scratch)
  golden_layers(ref_layers, out)
  golden_loss(ref_loss, out)
  golden_rank(ref_eval, out)
  golden_model(ref_model, ref_eval, out)
  golden_recon(ref_model, out)
  golden_collate(out)
  golden_quirks(ref_layers, out)
  for f in sorted(os.listdir(out)):
    print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == '__main__':
  main()
