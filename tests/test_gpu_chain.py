"""The multi-step kernels (step chain, resident tails), the step plan, tuning contexts and their failure paths.
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
