"""train.validate's calls through the reference API (train.py:223-236): encode_data -> six NumPy
matrices -> i2t(vid, para) -> t2i(vid, para).  The matrices leave the device through page-locked
staging; the ranking of both directions is queued by encode_data and served to i2t / t2i — only
when they are handed the very arrays encode_data returned, holding what it wrote.
"""
import argparse
import gc

import numpy as np
import pytest
import torch

from cmhse_amd import synthetic

pytestmark = pytest.mark.gpu


def _opt(rnn_type='attention', embed=64, img_dim=24, vocab=60):
  return argparse.Namespace(
      margin=0.2, word_dim=300, embed_size=embed, grad_clip=2.0, learning_rate=0.001,
      max_violation=False, img_dim=img_dim, measure='cosine', rnn_type=rnn_type, img_first_size=embed,
      cap_first_size=embed, low_level_loss=False, weak_low_level_loss=False, reconstruct_loss=False,
      lowest_reconstruct_loss=False, norm=False, data_name='anet_precomp', vocab_size=vocab, log_step=10)


@pytest.fixture
def small():
  from cmhse_amd import evaluation as ev
  from cmhse_amd.model import VSE
  opt = _opt()
  torch.manual_seed(5)
  model = VSE(opt)
  spec = synthetic.ragged_spec(37, seed=11)
  loader = synthetic.ListLoader(synthetic.make_batches(spec, 8, opt.img_dim, opt.vocab_size, seed=3))
  ev._forget_last_encode()
  yield opt, model, loader, ev
  ev._forget_last_encode()
  ev.SPECULATE_RANKS[0] = True


def _quiet(*a, **k):
  pass


def test_staged_arrays_equal_the_device_matrices(small):
  """The six arrays encode_data returns (page-locked staging on the copy stream, level-1 rows
  early) are bit for bit the matrices encode_data_device leaves on the GPU — also when the loader
  is cut into several super-batches, and for host-resident (pinned) loader batches."""
  opt, model, loader, ev = small
  cat, ncl, cvt = ev.encode_data_device(opt, model, loader, logging=_quiet)
  want = {k: cat[k].cpu().numpy() for k in ev.MATRICES}
  got = ev.encode_data(opt, model, loader, 10, _quiet)
  assert got[6] == ncl and got[7] == cvt
  for k, a in zip(ev.MATRICES, got[:6]):
    assert a.dtype == np.float32 and a.flags.writeable and a.flags.c_contiguous
    np.testing.assert_array_equal(a, want[k])
  # several super-batches: rows land at their offsets
  old = ev.SUPERBATCH_BYTES[0]
  ev.SUPERBATCH_BYTES[0] = 1
  try:
    got2 = ev.encode_data(opt, model, loader, 10, _quiet)
  finally:
    ev.SUPERBATCH_BYTES[0] = old
  for a, b in zip(got[:6], got2[:6]):
    np.testing.assert_array_equal(a, b)
  # pinned host batches (what the reference's DataLoader hands over)
  pin = lambda t: t.pin_memory() if isinstance(t, torch.Tensor) and t.dtype == torch.float32 else t
  host_loader = synthetic.ListLoader([tuple(pin(t) for t in b) for b in loader])
  got3 = ev.encode_data(opt, model, host_loader, 10, _quiet)
  for a, b in zip(got[:6], got3[:6]):
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('rnn_type,longer', [('attention', 'visual'), ('attention', 'text'), ('maxout', 'text'),
                                             ('seq2seq', 'visual')])
def test_staging_follows_whichever_tower_ends_first(rnn_type, longer):
  """Each tower's level-1 rows leave for the host behind ITS ready event (cmhse_gru_job.out_ready_event):
  recorded early for an attention-pooled tower whose chain ends while the other still steps — the
  visual tower of an ActivityNet batch (long paragraphs), the text tower of a DiDeMo one (80-frame
  clips, short sentences) — and at the end of the call otherwise (max / last pooling).  Either way the
  arrays equal the device matrices bit for bit, at a width that runs the LDS-tiled kernels too."""
  from cmhse_amd import evaluation as ev, ops
  from cmhse_amd.model import VSE
  opt = _opt(rnn_type=rnn_type, embed=128, img_dim=40)
  torch.manual_seed(9)
  model = VSE(opt)
  if longer == 'visual':
    spec = synthetic.ragged_spec(41, seed=3, max_clips=4, max_frames=33, max_words=3, max_video=40)
  else:
    spec = synthetic.ragged_spec(41, seed=3, max_clips=6, max_frames=5, max_words=17, max_video=7)
  loader = synthetic.ListLoader(synthetic.make_batches(spec, 9, opt.img_dim, opt.vocab_size, seed=5))
  for tiled in (False, True):
    with ops.tuned(**(dict(tiny_max_seqs=0, mid_max_seqs=0) if tiled else {})):
      cat, _, _ = ev.encode_data_device(opt, model, loader, logging=_quiet)
      want = {k: cat[k].cpu().numpy() for k in ev.MATRICES}
      got = ev.encode_data(opt, model, loader, 10, _quiet)
    for k, a in zip(ev.MATRICES, got[:6]):
      np.testing.assert_array_equal(a, want[k], err_msg='%s tiled=%s' % (k, tiled))
  ev._forget_last_encode()


def test_i2t_t2i_are_served_from_the_last_encode_bit_identically(small):
  """train.py:234-236 on the arrays encode_data returned: both calls are served from the ranking
  encode_data queued (2 hits), and equal — report, top-1, ranks — what the calls compute from
  scratch on copies of the same arrays."""
  opt, model, loader, ev = small
  vid, para = ev.encode_data(opt, model, loader, 10, _quiet)[:2]
  h0 = ev.CACHE_STATS['hits']
  rep_i, top_i, rk_i = ev.i2t(vid, para, measure='cosine')
  rep_t, top_t, rk_t = ev.t2i(vid, para, measure='cosine')
  assert ev.CACHE_STATS['hits'] == h0 + 2
  # copies are different objects: never served, always recomputed
  rep_i2, top_i2, rk_i2 = ev.i2t(vid.copy(), para.copy())
  rep_t2, top_t2, rk_t2 = ev.t2i(vid.copy(), para.copy())
  assert ev.CACHE_STATS['hits'] == h0 + 2
  assert rep_i == rep_i2 and rep_t == rep_t2
  for a, b in ((top_i, top_i2), (rk_i, rk_i2), (top_t, top_t2), (rk_t, rk_t2)):
    assert a.dtype == np.float64
    np.testing.assert_array_equal(a, b)
  # and with the speculation off the same arrays are simply uploaded and ranked
  ev.SPECULATE_RANKS[0] = False
  vid3, para3 = ev.encode_data(opt, model, loader, 10, _quiet)[:2]
  np.testing.assert_array_equal(vid3, vid)
  rep_i3, top_i3, rk_i3 = ev.i2t(vid3, para3)
  assert ev.CACHE_STATS['hits'] == h0 + 2 and rep_i3 == rep_i
  np.testing.assert_array_equal(rk_i3, rk_i)


def test_an_array_modified_in_place_is_not_served(small):
  """The arrays are ordinary writable ndarrays; a caller may edit them between encode_data and
  i2t / t2i.  The content check sees it: the call ranks what the arrays hold NOW, and the entry is
  gone for good (t2i afterwards recomputes too)."""
  opt, model, loader, ev = small
  vid, para = ev.encode_data(opt, model, loader, 10, _quiet)[:2]
  _, _, rk_before = ev.i2t(vid, para)
  vid[:] = vid[::-1].copy()                      # reverse the video rows in place
  h0, s0 = ev.CACHE_STATS['hits'], ev.CACHE_STATS['stale']
  rep, top1, rk = ev.i2t(vid, para)
  assert ev.CACHE_STATS['hits'] == h0 and ev.CACHE_STATS['stale'] == s0 + 1
  rep_w, top1_w, rk_w = ev.i2t(vid.copy(), para.copy())
  np.testing.assert_array_equal(rk, rk_w)
  np.testing.assert_array_equal(top1, top1_w)
  assert not np.array_equal(rk, rk_before)
  rep_t, _, rk_t = ev.t2i(vid, para)
  assert ev.CACHE_STATS['hits'] == h0            # the entry did not come back
  np.testing.assert_array_equal(rk_t, ev.t2i(vid.copy(), para.copy())[2])
  # a single flipped bit in the paragraph matrix is enough
  vid, para = ev.encode_data(opt, model, loader, 10, _quiet)[:2]
  para.view(np.uint32)[3, 5] ^= 1
  s1 = ev.CACHE_STATS['stale']
  ev.t2i(vid, para)
  assert ev.CACHE_STATS['stale'] == s1 + 1


def test_the_entry_dies_with_the_arrays(small):
  """Nothing of a pass stays on the device once the caller has dropped the arrays."""
  opt, model, loader, ev = small
  out = ev.encode_data(opt, model, loader, 10, _quiet)
  assert ev._LAST_ENCODE[0] is not None
  del out
  gc.collect()
  assert ev._LAST_ENCODE[0] is None


@pytest.mark.parametrize('nbytes', [16, 4 * 1024 * 37 + 4, 8 << 20, (8 << 20) + 12])
def test_push_rows_moves_exactly_the_bytes(nbytes):
  """cmhse_push_rows: device bytes -> page-locked host bytes, sizes that are not a multiple of 16
  included; bytes behind the end stay untouched; the runtime's copy gives the same result."""
  from cmhse_amd import ops
  dev = torch.device('cuda', 0)
  src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
  dst = torch.full((nbytes + 64,), 7, dtype=torch.uint8).pin_memory()
  stream = ops.stream_set(dev)[1]
  stream.wait_stream(torch.cuda.current_stream())
  ops.push_rows(src, dst[:nbytes], stream)
  stream.synchronize()
  assert torch.equal(dst[:nbytes], src.cpu())
  assert bool((dst[nbytes:] == 7).all())
  with pytest.raises(RuntimeError):
    ops.push_rows(src, torch.empty(nbytes, dtype=torch.uint8), stream)        # pageable destination


def test_rows_differ_sees_single_bits_anywhere():
  """cmhse_rows_differ: pinned-host against device bytes, sizes with ragged ends, one flipped bit at
  the first, a middle and the very last byte."""
  from cmhse_amd import ops
  dev = torch.device('cuda', 0)
  for nbytes in (16, 4096 * 5 + 16 * 3, (1 << 20) + 16):
    base = torch.randint(0, 256, (nbytes,), dtype=torch.uint8)
    d = base.to(dev)
    h = base.clone().pin_memory()
    assert ops.rows_differ([(h, d)]) is False
    assert ops.rows_differ([(d.clone(), d)]) is False            # device against device
    for pos in (0, nbytes // 2 + 1, nbytes - 1):
      h2 = base.clone()
      h2[pos] ^= 0x10
      assert ops.rows_differ([(h2.pin_memory(), d)]) is True, (nbytes, pos)
    if nbytes > 16:
      assert ops.rows_differ([(h, d), (h[:nbytes - 16].clone().pin_memory(), d)]) is True   # sizes differ
