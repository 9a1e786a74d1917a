"""The drop-in boundary as a train.py-style driver uses it (SURVEY.md §8b, B-outer):
`from model import VSE`, `from evaluation import i2t, t2i, AverageMeter, LogCollector, encode_data,
LogReporter` (train.py:9-10) resolved through cmhse_amd/dropin/, a training / validation /
checkpoint / resume loop shaped like train.py:124-172,185-257 run through those imports, and
bench.py started the way the driver starts it (a bare `python bench.py --gpus N`).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

DROPIN = os.path.join(REPO, 'cmhse_amd', 'dropin')
# every name a driver written against the reference pulls from the four hot-path modules
# (train.py:9-10; evaluation.py:15 `from model import VSE`; model.py:15-17 `from layers import *`,
# `from loss import *`)
NAMES = {
    'model': ['VSE', 'EncoderImage', 'EncoderSequence', 'EncoderText'],
    'evaluation': ['i2t', 't2i', 'AverageMeter', 'LogCollector', 'encode_data', 'LogReporter'],
    'loss': ['ContrastiveLoss', 'GroupWiseContrastiveLoss', 'cosine_sim'],
    'layers': ['Attention', 'Maxout', 'Seq2Seq'],
}


def test_dropin_directory_resolves_every_name_train_py_imports():
  """With cmhse_amd/dropin first on sys.path, the reference's import lines work unchanged and
  bind to the cmhse_amd implementations (not to anything else called `model` on the path).  Run in
  a child interpreter so the top-level module names do not leak into this test session."""
  prog = ['import sys', 'sys.path.insert(0, %r)' % REPO, 'sys.path.insert(0, %r)' % DROPIN,
          'from model import VSE',
          'from evaluation import i2t, t2i, AverageMeter, LogCollector, encode_data, LogReporter',
          'import model, evaluation, loss, layers, cmhse_amd.model, cmhse_amd.evaluation, '
          'cmhse_amd.loss, cmhse_amd.layers']
  for mod, names in NAMES.items():
    for n in names:
      prog.append('assert %s.%s is cmhse_amd.%s.%s, %r' % (mod, n, mod, n, mod + '.' + n))
  prog.append('assert model.__file__.startswith(%r)' % DROPIN)
  prog.append('print("resolved")')
  res = subprocess.run([sys.executable, '-c', '\n'.join(prog)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
  assert res.returncode == 0 and 'resolved' in res.stdout, res.stderr[-2000:]


def test_bare_multi_gpu_launch_without_a_gpu_fails_cleanly():
  """`python bench.py --gpus 2` with no WORLD_SIZE goes down the self-launch path; on a box
  without any GPU that path must return an error code without trying to start ranks."""
  if torch.cuda.device_count() > 0:
    pytest.skip('covered on a GPU box by test_bench_starts_its_own_ranks')
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
  res = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
  assert res.returncode == 2 and 'no GPU visible' in res.stderr


class _Tb(object):
  """Stand-in for tensorboard_logger (train.py:13): records log_value calls."""

  def __init__(self):
    self.rows = []

  def log_value(self, name, value, step=None):
    self.rows.append((name, float(value), step))


@pytest.fixture
def dropin_modules():
  """`model` / `evaluation` imported the way train.py does, with the shim directory first on the
  path; removed from sys.modules afterwards."""
  sys.path.insert(0, DROPIN)
  try:
    import evaluation
    import model
    yield model, evaluation
  finally:
    sys.path.remove(DROPIN)
    for name in ('model', 'evaluation', 'loss', 'layers'):
      sys.modules.pop(name, None)


@pytest.mark.gpu
@pytest.mark.parametrize('rnn_type,recon', [('attention', False), ('maxout', True)])
def test_train_py_shaped_loop_through_the_dropin_imports(dropin_modules, tmp_path, rnn_type, recon):
  """The calls train.py makes, in its order (train.py:124-172 main, :185-221 train, :223-257
  validate, :259-260 save_checkpoint): construct, print the encoders, set the learning rate through
  optimizer.param_groups, two train_emb steps with the logger assigned from outside, validate
  (encode_data + i2t + t2i + LogReporter), save the list-of-state-dicts checkpoint, resume it into a
  fresh VSE, validate again: the resumed model must reproduce the embeddings bit for bit."""
  import argparse
  from cmhse_amd import synthetic
  model_mod, ev = dropin_modules
  assert torch.cuda.is_available()
  opt = argparse.Namespace(
      margin=0.2, word_dim=300, embed_size=64, grad_clip=2.0, learning_rate=0.001, lr_update=10,
      max_violation=False, img_dim=24, measure='cosine', rnn_type=rnn_type, img_first_size=64,
      cap_first_size=64, low_level_loss=True, weak_low_level_loss=False, reconstruct_loss=recon,
      lowest_reconstruct_loss=False, norm=True, weight_recon=0.0005, lowest_weight_recon=0.0001,
      decode_rnn_type='seq2seq', data_name='anet_precomp', vocab_size=60, log_step=1, val_step=500)
  spec = synthetic.ragged_spec(12, seed=4)
  train_loader = synthetic.ListLoader(synthetic.make_batches(spec, 6, opt.img_dim, opt.vocab_size, seed=1))
  val_loader = synthetic.ListLoader(synthetic.make_batches(spec, 5, opt.img_dim, opt.vocab_size, seed=2))
  tb = _Tb()
  lines = []

  torch.manual_seed(3)
  model = model_mod.VSE(opt)
  for enc in (model.clip_enc, model.txt_enc, model.vid_seq_enc, model.txt_seq_enc):
    assert 'GRU' in str(enc)                                     # train.py:126-130 prints them

  def adjust_learning_rate(optimizer, epoch):                    # train.py:263-270
    lr = opt.learning_rate * (0.1 ** (epoch // opt.lr_update))
    for group in optimizer.param_groups:
      group['lr'] = lr

  def validate(m):                                               # train.py:223-257
    embs = ev.encode_data(opt, m, val_loader, opt.log_step, lines.append, contextual_model=True)
    vid, para = embs[0], embs[1]
    rep_v, top1_v, rank_v = ev.i2t(vid, para, measure=opt.measure)
    rep_p, top1_p, rank_p = ev.t2i(vid, para, measure=opt.measure)
    ev.LogReporter(tb, rep_v, m.Eiters, 'seq')
    ev.LogReporter(tb, rep_p, m.Eiters, 'seqi')
    return rep_v['sum'] + rep_p['sum'], embs, (rank_v, rank_p)

  adjust_learning_rate(model.optimizer, 0)
  batch_time, train_logger = ev.AverageMeter(), ev.LogCollector()
  model.train_start(opt)
  before = [p.detach().clone() for p in model.params]
  for i, train_data in enumerate(train_loader):
    model.logger = train_logger                                  # train.py:190
    model.train_emb(opt, *train_data)
    batch_time.update(0.01)
    lines.append('Epoch: [0][%d/%d]\t%s\tTime %.3f' % (i, len(train_loader), str(model.logger),
                                                      batch_time.val))
    model.logger.tb_log(tb, step=model.Eiters)
  assert model.Eiters == 2
  assert any(not torch.equal(a, b.detach()) for a, b in zip(before, model.params))   # Adam stepped
  want = {'Eit', 'lr', 'Le_vid', 'Le_ctx_low_lvel', 'Le_vid_inloss', 'Le_para_inloss', 'Le_low_lvel',
          'Le_clip_inloss', 'Le_cap_inloss'} | ({'Le_clip_recon', 'Le_cap_recon'} if recon else set())
  assert set(train_logger.meters) == want
  assert all(np.isfinite(m.val) for m in train_logger.meters.values())

  # train.py's own loop reads the logger only on log steps but calls tb_log after EVERY step
  # (train.py:198-215): the tensorboard sink must still receive every step's values under that
  # step's number — delivered late (the call queues behind the loss values in flight), at the
  # latest when the loop switches modes for validation
  for i, train_data in enumerate(train_loader):
    model.logger = train_logger
    model.train_emb(opt, *train_data)
    model.logger.tb_log(tb, step=model.Eiters)
  assert model.Eiters == 4
  score, embs, ranks = validate(model)
  for step in (1, 2, 3, 4):
    got = {r[0]: r[1] for r in tb.rows if r[2] == step}
    assert want <= set(got) and got['Eit'] == step and np.isfinite(got['Le_vid']), (step, sorted(got))
  assert np.isfinite(score) and any(r[0] == 'seqr1' for r in tb.rows)
  assert any(l.startswith('Test: [0/') for l in lines)            # evaluation.py:135-141 log line

  path = str(tmp_path / '0checkpoint.pth.tar')                    # train.py:166-172, :259-260
  torch.save({'epoch': 1, 'model': model.state_dict(opt), 'best_rsum': score, 'opt': opt,
              'Eiters': model.Eiters}, path)
  sd = model.state_dict(opt)
  assert isinstance(sd, list) and len(sd) == (6 if recon else 4)  # model.py:166-179

  checkpoint = torch.load(path, weights_only=False)               # train.py:139-147
  torch.manual_seed(99)                                           # a different init, overwritten
  resumed = model_mod.VSE(opt)
  resumed.load_state_dict(checkpoint['model'], opt)
  resumed.Eiters = checkpoint['Eiters']
  score2, embs2, ranks2 = validate(resumed)
  for a, b in zip(embs[:6], embs2[:6]):
    np.testing.assert_array_equal(a, b)
  np.testing.assert_array_equal(ranks[0], ranks2[0])
  np.testing.assert_array_equal(ranks[1], ranks2[1])
  assert score2 == score and embs[6] == embs2[6] and embs[7] == embs2[7]


def _run_bench(extra, timeout=1200):
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'CMHSE_BENCH_BACKEND'):
    env.pop(k, None)
  cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '1', '--warmup', '1',
         '--fast_steps', '0', '--train_steps', '0', '--host_steps', '0', '--cpu_batches', '0',
         '--rank_check', '0'] + extra
  res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
  assert res.returncode == 0, res.stderr[-3000:]
  lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, res.stdout[-2000:]
  return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_matches_the_single_process_ranks():
  """`python bench.py --gpus 2` from a bare shell (no torchrun, no WORLD_SIZE): the parent starts
  two rank processes (sharing this box's GPU over gloo when it has only one), ONE JSON line comes
  out with two per_rank entries, and the integer ranks of both directions at ICEP width are those
  of the single-process run (train.py:223-236's validate, sharded) — at 1500 videos, where a rank's
  share and the whole split sit on different sides of the 1024-sequence kernel crossover for many
  time steps (level 2: 750 against 1500 videos): the ranks agree on the whole split's step plan
  (parallel_eval.global_step_plan, two small control collectives) and encode their shares with it."""
  one = _run_bench(['--gpus', '1', '--n_videos', '1500'])
  two = _run_bench(['--gpus', '2', '--n_videos', '1500'])
  assert one['n_gpus'] == 1 and two['n_gpus'] == 2
  assert len(two['per_rank']) == 2
  assert sum(r['videos'] for r in two['per_rank']) == 1500
  assert all(r['encode_ms'] > 0 and r['score_ms'] > 0 for r in two['per_rank'])
  assert two['ranks_crc32'] == one['ranks_crc32']
  assert two['report_i2t_random_init'] == one['report_i2t_random_init']
  assert two['report_t2i_random_init'] == one['report_t2i_random_init']
