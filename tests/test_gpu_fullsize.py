"""BASELINE configs[1..4] at their real shapes: full-dimension encoders, training steps and the N = 4917 split against the oracle.
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
