"""The packed GRU + pooling kernels, forward and backward, against the oracle on seeded inputs; shapes that must be bit-identical.
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


@pytest.mark.parametrize('pool', ['attention', 'maxout', 'seq2seq'])
@pytest.mark.parametrize('S,T,I,H', [(2300, 5, 36, 72), (2200, 3, 500, 128), (2100, 4, 12, 64), (2100, 3, 20, 96)])
def test_bf16x3_mode_vs_oracle(dev, oracle, pool, S, T, I, H):
  """Same parity bar (1e-4) as the exact path; also reports how close the split really is.  (I = 12 / 20:
  one / two 16-k chunks in the input phase — the LDS-DMA ring's clamped re-loads, nt_phase_bf3_ring.)"""
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


@pytest.mark.parametrize('I,H', [(300, 1024), (2048, 1024), (36, 96)])
def test_bf16x3_ring_is_deterministic_over_many_runs(dev, I, H):
  """The bf16x3 loops stage operands by LDS-DMA into a 3-stage ring ordered by counted vmcnt waits and
  raw barriers (nt_phase_bf3_ring): a read placed one step too early would pass a single comparison
  whenever the DMA happens to land first.  Screen for that: 25 runs of an attention-pooled encoder whose
  step launches co-reside two workgroups per CU (19 / 128 / 3 chunks in the input phase), every output
  bit-identical to the first run's, with a competing exact-fp32 call between the runs."""
  from cmhse_amd import layers, ops
  torch.manual_seed(11)
  S, T = 2600, 4
  layer = layers.Attention(I, H).to(dev)
  rng = np.random.RandomState(5)
  lens = np.sort(rng.randint(2, T + 1, size=S))[::-1].copy()
  lens[0] = T
  x = torch.randn(S, T, I, device=dev)
  lens_t = torch.from_numpy(lens)
  other = layers.Maxout(I, H).to(dev)
  try:
    ops.set_math_mode('bf16x3')
    with torch.no_grad():
      first = layer(x, lens_t).clone()
    for _ in range(25):
      ops.set_math_mode('fp32')
      with torch.no_grad():
        other(x, lens_t)
      ops.set_math_mode('bf16x3')
      with torch.no_grad():
        again = layer(x, lens_t)
      assert torch.equal(again, first)
  finally:
    ops.set_math_mode('fp32')
  with torch.no_grad():
    exact = layer(x, lens_t)
  assert not torch.equal(exact, first), 'bf16x3 mode did not engage'
  assert float((exact - first).abs().max()) < 5e-5
