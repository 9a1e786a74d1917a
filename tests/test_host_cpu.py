"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares,
host-side schedule/meters/report logic, the synthetic batch contract, loud failure without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden


def test_library_exports_every_declared_symbol():
  from cmhse_amd import _lib, build
  build.build()
  lib = _lib.load()
  header = open(os.path.join(REPO, 'include', 'cmhse_hip.h')).read()
  header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)      # drop comments
  declared = set(re.findall(r'\b(cmhse_[a-z0-9_]+)\s*\(', header))
  assert declared, 'no declarations parsed'
  assert declared == set(_lib.SIGNATURES.keys())
  for name in declared:
    assert hasattr(lib, name), name
  assert lib.cmhse_version().decode().endswith('gfx950')
  assert lib.cmhse_strerror(-2).decode().startswith('workspace')
  # size queries are pure host functions
  assert lib.cmhse_gru_pool_workspace(4, 3, 10, 24, 32, 0) >= 10 * 32 * 4
  assert lib.cmhse_gru_pool_workspace(4, 3, 10, 24, 32, 1) > lib.cmhse_gru_pool_workspace(4, 3, 10, 24, 32, 0)
  assert lib.cmhse_gru_pool_workspace(4, 3, 10, 24, 32, 0x200) > lib.cmhse_gru_pool_workspace(4, 3, 10, 24, 32, 0)
  assert lib.cmhse_sim_rank_workspace(100) >= 100 * 12
  assert lib.cmhse_contrastive_workspace(10) >= 400


def test_argument_validation_without_gpu():
  """Bad arguments are rejected with error codes before anything touches the device."""
  from cmhse_amd import _lib
  lib = _lib.load()
  assert lib.cmhse_l2norm_rows(None, None, 1, 1, 1, None) == -1
  assert lib.cmhse_sim_rank(None, None, 1, 1, 1, 0, 1, None, None, None, 0, None) == -1
  assert lib.cmhse_contrastive_fwd(None, None, 1, 1, 0.2, 0, 0, None, None, None, 0, None) == -1
  assert lib.cmhse_gru_pool_fwd(None, None, 0, None, None, 0, None) == -1


def test_seq_schedule_matches_pack_padded_sequence():
  from cmhse_amd.ops import SeqSchedule
  from torch.nn.utils.rnn import pack_padded_sequence
  lens = np.array([3, 7, 1, 7, 4, 2])
  ptrs = np.arange(len(lens), dtype=np.uint64) * 1000
  s = SeqSchedule(lens, 'cpu', x_ptrs=ptrs)
  x = torch.zeros(len(lens), 7, 1)
  packed = pack_padded_sequence(x[torch.from_numpy(s.order)], s.lens_sorted.tolist(),
                                batch_first=True)
  assert packed.batch_sizes.tolist() == s.step_count_host.tolist()
  assert s.sum_T == lens.sum() and s.Tmax == 7
  meta = s.meta.numpy()
  S = len(lens)
  rows = meta[:S * 8].view(np.uint64)
  assert rows.tolist() == (ptrs[s.order]).tolist()
  v32 = meta[2 * S * 8:].view(np.int32)
  assert v32[:S].tolist() == sorted(lens.tolist(), reverse=True)
  assert v32[S:2 * S].tolist() == s.order.tolist()
  assert v32[2 * S:].tolist() == np.concatenate([[0], np.cumsum(s.step_count_host)]).tolist()
  with pytest.raises(ValueError):
    SeqSchedule(np.array([2, 0]), 'cpu', x_ptrs=ptrs[:2])


def test_seq_schedule_order_is_the_stable_descending_sort():
  """The schedule's order is torch.sort(lengths, 0, True)'s for the reference (layers.py:94) up to
  ties, and among equal lengths the input order (stable) — with the 16-bit radix keys used for
  ordinary lengths and with the comparison sort used for lengths >= 65536 alike."""
  from cmhse_amd.ops import SeqSchedule
  rng = np.random.RandomState(4)
  for hi in (7, 435, 70000):
    lens = rng.randint(1, hi + 1, size=5000).astype(np.int64)
    lens[17] = hi
    if hi > 65535:
      lens[:4000] = rng.randint(1, 50, size=4000)     # keep the packed batch below 2^31 rows
    s = SeqSchedule(lens, 'cpu', x_ptrs=np.arange(len(lens), dtype=np.uint64))
    assert s.order.tolist() == np.argsort(-lens, kind='stable').tolist()
    assert s.lens_sorted.tolist() == sorted(lens.tolist(), reverse=True)
    for t in range(0, s.Tmax, max(1, s.Tmax // 50)):
      assert int(s.step_count_host[t]) == int((lens > t).sum())


def test_validation_pass_switches_the_cyclic_collector_off_and_back_on():
  """evaluation._no_gc_pause: no cyclic collection while a pass's launches are queued; the
  collector's state is what it was afterwards, also after an exception, also when it was off."""
  import gc
  from cmhse_amd.evaluation import _no_gc_pause
  assert gc.isenabled()
  with _no_gc_pause():
    assert not gc.isenabled()
  assert gc.isenabled()
  with pytest.raises(RuntimeError):
    with _no_gc_pause():
      raise RuntimeError('x')
  assert gc.isenabled()
  gc.disable()
  try:
    with _no_gc_pause():
      assert not gc.isenabled()
    assert not gc.isenabled()
  finally:
    gc.enable()


def test_meters_follow_reference_quirks():
  from cmhse_amd.evaluation import AverageMeter, LogCollector
  m = AverageMeter()
  m.update(5)                       # n defaults to 0: records val only (evaluation.py:30)
  assert str(m) == '5' and m.count == 0
  m.update(2.0, 4)
  assert abs(m.avg - 8.0 / 4.0001) < 1e-12
  lc = LogCollector()
  lc.update('Le_vid', 1.5, 2)
  lc.update('Eit', 3)
  assert str(lc) == 'Le_vid 1.5000 (1.4999)  Eit 3'   # avg = 3 / 2.0001


@pytest.mark.parametrize('n', [50, 203])
def test_report_from_ranks_vs_golden(n):
  from cmhse_amd.evaluation import report_from_ranks
  g = load_golden('rank.npz')
  for nm in ['i2t', 't2i']:
    rep = report_from_ranks(g['n%d.%s.ranks' % (n, nm)])
    got = np.array([rep[k] for k in ['r1', 'r5', 'r10', 'medr', 'meanr', 'sum']])
    np.testing.assert_array_equal(got, g['n%d.%s.report' % (n, nm)])


def test_synthetic_batches_honour_the_12_tuple_contract():
  from cmhse_amd import synthetic
  spec = synthetic.anet_like_spec(50, seed=0)
  batches = synthetic.make_batches(spec, 16, 20, 100, seed=0)
  assert len(batches) == 4
  seen = 0
  for b in batches:
    clips, caps, vids, pars, lc, lcap, lv, lp, nc, ncap, ind, cur = b
    assert clips.shape[0] == caps.shape[0] == sum(nc) == len(lc) == len(lcap)
    assert clips.shape[1] == int(lc.max()) and caps.shape[1] == int(lcap.max())
    assert vids.shape[0] == pars.shape[0] == len(nc) == len(ind) == len(cur)
    assert int(lc.max()) <= 80 and int(lv.max()) <= 80 and int(lc.min()) >= 1
    j = 0
    for v, c in enumerate(nc):          # paragraph = concatenation of its sentences
      want = torch.cat([caps[j + k, :lcap[j + k]] for k in range(c)])
      assert torch.equal(pars[v, :lp[v]], want)
      assert float(clips[j, lc[j]:].abs().sum()) == 0.0      # zero padding
      j += c
    seen += len(nc)
  assert seen == 50
  assert spec.totals()['clips'] == sum(spec.num_clips)


def test_product_path_fails_loudly_without_gpu():
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  import argparse
  from cmhse_amd import ops
  from cmhse_amd.model import VSE
  from cmhse_amd.evaluation import i2t
  with pytest.raises(RuntimeError):
    ops.l2norm_rows(torch.zeros(2, 4))
  with pytest.raises(RuntimeError):
    i2t(np.zeros((2, 4), np.float32), np.zeros((2, 4), np.float32))
  opt = argparse.Namespace(norm=False, grad_clip=0, img_dim=8, img_first_size=8, vocab_size=10,
                           word_dim=4, cap_first_size=8, rnn_type='maxout', embed_size=8,
                           data_name='x', margin=0.2, measure='cosine', max_violation=False,
                           learning_rate=1e-3)
  with pytest.raises(RuntimeError):
    VSE(opt)


def test_product_never_imports_the_oracle():
  """The oracle is test infrastructure: no file of the package may reference it."""
  pkg = os.path.join(REPO, 'cmhse_amd')
  for root, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.hpp', '.h')):
        text = open(os.path.join(root, f)).read()
        assert 'cmhse_oracle' not in text and 'import oracle' not in text, f


def test_schedule_and_wrappers_reject_bad_input():
  """Host-side validation happens before any device work."""
  from cmhse_amd.ops import SeqSchedule
  with pytest.raises(ValueError):
    SeqSchedule(np.array([], dtype=np.int64), 'cpu', x_ptrs=np.array([], dtype=np.uint64))
  with pytest.raises(ValueError):
    SeqSchedule(np.array([3, -1]), 'cpu', x_ptrs=np.zeros(2, dtype=np.uint64))
  from cmhse_amd import ops
  with pytest.raises(ValueError):
    ops.set_math_mode('fp16')
  assert ops.math_mode() in ('fp32', 'bf16x3')


def test_pool_all_schedule_rows_are_sequence_starts():
  """CMHSE_POOL_ALL: out_row[s] is the first output row of (sorted) sequence s."""
  from cmhse_amd.ops import SeqSchedule
  lens = np.array([2, 5, 1, 3])
  s = SeqSchedule(lens, 'cpu', x_ptrs=np.zeros(4, dtype=np.uint64), out_rows_are_starts=True)
  S = len(lens)
  v32 = s.meta.numpy()[2 * S * 8:].view(np.int32)
  starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
  assert v32[S:2 * S].tolist() == starts[s.order].tolist()


def test_error_strings_cover_all_codes():
  from cmhse_amd import _lib
  lib = _lib.load()
  seen = {lib.cmhse_strerror(c).decode() for c in (0, -1, -2, -3, -4, -99)}
  assert len(seen) == 6


def test_profile_tools_reduce_rocprof_csvs(tmp_path):
  """tools/pmc_traffic.py and tools/summarize_rocprof.py on hand-made rocprofv3 CSVs: KiB units,
  the gfx950 doubling of FETCH_SIZE, per-launch averages, and the markdown table."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  head = 'Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n'
  for name, vals in (('FETCH_SIZE', (1024.0, 3072.0)), ('WRITE_SIZE', (512.0, 512.0))):
    d = tmp_path / ('pmc_' + name) / 'box'
    d.mkdir(parents=True)
    rows = ''.join('%d,"void cmhse::gru_step_kernel<true, 1, false>(cmhse::GruStepGroup)",%s,%f\n'
                   % (i, name, v) for i, v in enumerate(vals))
    rows += '9,"void at::native::fill()",%s,7.0\n' % name
    (d / '1_counter_collection.csv').write_text(head + rows)
  out = subprocess.check_output([sys.executable, os.path.join(root, 'tools', 'pmc_traffic.py'),
                                 str(tmp_path / 'pmc_FETCH_SIZE'), str(tmp_path / 'pmc_WRITE_SIZE')])
  res = json.loads(out)
  assert list(res) == ['void cmhse::gru_step_kernel<true, 1, false>']       # foreign kernels dropped
  k = res['void cmhse::gru_step_kernel<true, 1, false>']
  assert k['launches'] == 2
  assert k['hbm_read_bytes_per_launch_corrected'] == 2.0 * 4096.0 * 1024.0 / 2
  assert k['hbm_write_bytes_per_launch'] == 1024.0 * 1024.0 / 2
  stats = tmp_path / 'kernel_stats.csv'
  stats.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                   '"void cmhse::gru_step_kernel<true, 1, false>(cmhse::GruStepGroup)",4,8000000,'
                   '2000000.0,80.0,1000000,3000000,1.0\n"k2",1,2000000,2000000.0,20.0,2000000,2000000,0.0\n')
  md = subprocess.check_output([sys.executable, os.path.join(root, 'tools', 'summarize_rocprof.py'),
                                str(stats), 'title']).decode()
  assert 'total kernel time 10.000 ms' in md
  assert '| 4 | 8.000 | 2000.00 | 1000.00 | 3000.00 | 80.00 |' in md


@pytest.mark.parametrize('tag,didemo', [('anet', False), ('didemo', True)])
def test_collate_fn_vs_reference_golden(tag, didemo):
  """collate.collate_fn against the 12-tuples the reference's own collate_fn produced on the same
  seeded samples (tools/make_golden.py: activity_net/data.py:114-150, didemo_dev/data.py:127-165):
  values, shapes, dtypes, member types — bit for bit."""
  from cmhse_amd import collate, synthetic
  g = load_golden('collate.npz')
  samples = synthetic.dataset_samples(int(g[tag + '_seed']), int(g[tag + '_img_dim']), 5, didemo)
  res = collate.collate_fn(samples)
  assert len(res) == 12
  names = ['clips', 'captions', 'videos', 'paragraphs', 'lengths_clip', 'lengths_cap',
           'lengths_video', 'lengths_paragraph']
  for k, name in enumerate(names):
    want = g['%s_%s' % (tag, name)]
    assert isinstance(res[k], torch.Tensor)
    assert str(res[k].dtype) == str(g['%s_%s_dtype' % (tag, name)])
    assert tuple(res[k].shape) == want.shape
    np.testing.assert_array_equal(res[k].numpy(), want)
  assert isinstance(res[8], tuple) and isinstance(res[9], tuple) and isinstance(res[10], tuple)
  np.testing.assert_array_equal(np.asarray(res[8]), g[tag + '_num_clips'])
  np.testing.assert_array_equal(np.asarray(res[9]), g[tag + '_num_caps'])
  np.testing.assert_array_equal(np.asarray(res[10]), g[tag + '_index'])
  if bool(g[tag + '_last_is_tensor']):
    assert isinstance(res[11], torch.Tensor) and res[11].dtype == torch.int64
    np.testing.assert_array_equal(res[11].numpy(), g[tag + '_last'])
  else:
    assert isinstance(res[11], tuple) and list(res[11]) == list(g[tag + '_last'])


def test_collate_packed_holds_the_same_batch_without_padding():
  """collate_packed: one host block, the sequences back to back; padding it (host path of
  Ragged.padded) gives collate_fn's tensors exactly; split_samples inverts collate_fn."""
  from cmhse_amd import collate, ops, synthetic
  samples = synthetic.dataset_samples(3, 10, 6)
  ref = collate.collate_fn(samples)
  pk = collate.collate_packed(samples, pin=False)
  base = pk[0].data.untyped_storage().data_ptr()
  for k in range(4):
    assert isinstance(pk[k], ops.Ragged)
    assert pk[k].data.untyped_storage().data_ptr() == base          # ONE block
    assert pk[k].shape == tuple(ref[k].shape)
    assert torch.equal(pk[k].padded(), ref[k])
    np.testing.assert_array_equal(pk[k].lens, np.asarray(ref[4 + [0, 1, 2, 3][k]]))
  # no padding stored: the block is exactly the valid rows (plus < 8 bytes of alignment)
  valid = int(ref[4].sum() + ref[6].sum()) * 10 * 4 + int(ref[5].sum() + ref[7].sum()) * 8
  assert valid <= pk[0].data.untyped_storage().nbytes() < valid + 8
  for k in range(4, 8):
    assert torch.equal(pk[k], ref[k])
  assert pk[8:] == ref[8:]
  # round trip through the per-sample form
  again = collate.collate_fn(collate.split_samples(ref))
  for k in range(8):
    assert torch.equal(again[k], ref[k])
  # row pointers: sequence s starts first[s] rows into the block
  ptrs = pk[0].row_ptrs()
  assert int(ptrs[0]) == pk[0].data.data_ptr()
  assert int(ptrs[1] - ptrs[0]) == int(pk[0].lens[0]) * 10 * 4


def test_tune_is_the_one_configuration_entry_point():
  """cmhse_tune (include/cmhse_hip.h): reads, sets and restores a kernel-shape crossover without a
  GPU; unknown names are an argument error, not a silent no-op."""
  from cmhse_amd import ops
  default = ops.tune('tiny_max_seqs')                 # read only
  assert default == 1024
  assert ops.tune('tiny_max_seqs', 7) == default      # set: returns the previous value
  assert ops.tune('tiny_max_seqs') == 7
  with ops.tuned(tiny_max_seqs=0, mid_units=8):
    assert ops.tune('tiny_max_seqs') == 0 and ops.tune('mid_units') == 8
  assert ops.tune('tiny_max_seqs') == 7 and ops.tune('mid_units') == 0
  ops.tune('tiny_max_seqs', default)
  with pytest.raises(RuntimeError):
    ops.tune('no_such_tunable', 1)


def test_the_library_reads_no_environment_variable():
  """Experiment switches were retired in round 3: the native sources contain no getenv, and the
  package reads exactly one variable (the library path)."""
  import glob
  import re
  root = os.path.join(REPO, 'cmhse_amd')
  native = ''.join(open(f).read() for f in glob.glob(os.path.join(root, 'csrc', '*')))
  assert 'getenv' not in native
  py = ''.join(open(f).read() for f in glob.glob(os.path.join(root, '*.py')))
  assert sorted(set(re.findall(r"environ(?:\.get)?\(\s*'(CMHSE_[A-Z0-9_]+)'", py))) == ['CMHSE_HIP_LIB']


def test_log_collector_replays_late_values_in_order():
  """evaluation.LogCollector with values still in flight (VSE.train_emb hands the step's losses
  over as a pending copy): updates made meanwhile queue behind it, every reader settles first, and
  the meters end up exactly as with immediate updates (evaluation.py:48-72 semantics)."""
  from cmhse_amd.evaluation import LogCollector
  a, b = LogCollector(), LogCollector()
  ran = []
  # immediate
  a.update('Eit', 1); a.update('lr', 0.5); a.update('Le_vid', 2.0, 4); a.update('Eit', 2)
  a.update('Le_vid', 4.0, 4)
  # deferred: step 1's losses arrive late, step 2's 'Eit' is updated before they are read
  b.update('Eit', 1); b.update('lr', 0.5)
  b.defer(lambda: (ran.append(1), b._update('Le_vid', 2.0, 4)))
  b.update('Eit', 2)
  assert not ran and list(b._meters) == ['Eit', 'lr'] and b._meters['Eit'].val == 1
  b.defer(lambda: (ran.append(2), b._update('Le_vid', 4.0, 4)))
  assert str(b) == str(a) and ran == [1, 2]
  assert list(b.meters) == ['Eit', 'lr', 'Le_vid']
  assert b.meters['Le_vid'].avg == a.meters['Le_vid'].avg and abs(a.meters['Le_vid'].avg - 3.0) < 1e-3 and b.meters['Eit'].val == 2
  b.update('Eit', 3)            # nothing outstanding: immediate again
  assert b._meters['Eit'].val == 3

  class Tb(object):
    def __init__(self): self.got = []
    def log_value(self, k, v, step=None): self.got.append((k, v, step))
  # tb_log after every step (train.py:215) does not wait for values in flight: it queues behind them
  # and delivers the same triples when they have arrived
  b.defer(lambda: b._update('Le_vid', 6.0, 4))
  tb = Tb(); b.tb_log(tb, prefix='t/', step=7)
  assert tb.got == []
  b.update('Eit', 4)
  b.defer(lambda: b._update('Le_vid', 7.0, 4))
  b.tb_log(tb, prefix='t/', step=8)
  b.settle()
  assert ('t/Le_vid', 6.0, 7) in tb.got and ('t/Eit', 3, 7) in tb.got
  assert ('t/Le_vid', 7.0, 8) in tb.got and ('t/Eit', 4, 8) in tb.got
  assert [g[2] for g in tb.got] == [7] * 3 + [8] * 3
  b.tb_log(tb, prefix='t/', step=9)      # nothing outstanding: immediate
  assert tb.got[-1][2] == 9
  # a collector with values still outstanding can be pickled / deep-copied: both settle first
  import copy
  import pickle
  b.defer(lambda: b._update('Le_vid', 8.0, 4))
  c = copy.deepcopy(b)
  assert c.meters['Le_vid'].val == 8.0 and not b._deferred
  b.defer(lambda: b._update('Le_vid', 10.0, 4))
  d = pickle.loads(pickle.dumps(b))
  assert d.meters['Le_vid'].val == 10.0 and str(d) == str(b)
  # a reset (`collector.meters = OrderedDict()`) delivers what was queued for the OLD meters first:
  # the tensorboard rows of the last step are not dropped
  from collections import OrderedDict
  b.defer(lambda: b._update('Le_vid', 11.0, 4))
  b.tb_log(tb, prefix='t/', step=10)
  n_before = len(tb.got)
  b.meters = OrderedDict()
  assert len(tb.got) > n_before and ('t/Le_vid', 11.0, 10) in tb.got
  assert len(b.meters) == 0
  b.defer(lambda: b._update('Le_vid', 12.0, 4))
  b.flush()
  assert b.meters['Le_vid'].val == 12.0


def test_plan_key_sees_every_batch():
  """evaluation._plan_key (the validity key of a cached encode plan): an in-place edit of a MIDDLE
  batch's lengths, a swapped middle batch, a replaced middle tensor and a shorter list all change
  it; an untouched list does not (VERDICT r03 weak 3b / ADVICE: the old key looked at the first
  and last batch only)."""
  from cmhse_amd import evaluation, synthetic
  spec = synthetic.ragged_spec(40, seed=3, max_frames=9, max_video=11)
  batches = synthetic.make_batches(spec, 8, 12, 50, seed=4)
  assert len(batches) == 5
  k0 = evaluation._plan_key(batches)
  assert k0 == evaluation._plan_key(list(batches))
  mid = batches[2]
  lens = mid[4]
  i = int(np.argmax(np.asarray(lens) > 1))
  old = int(lens[i])
  lens[i] = old - 1                         # in place: same objects, same first / last batch
  assert evaluation._plan_key(batches) != k0
  lens[i] = old
  assert evaluation._plan_key(batches) == k0
  swapped = [batches[0], batches[3], batches[2], batches[1], batches[4]]
  assert evaluation._plan_key(swapped) != k0
  replaced = list(batches)
  replaced[2] = (mid[0].clone(),) + tuple(mid[1:])      # same values, new storage
  assert evaluation._plan_key(replaced) != k0
  assert evaluation._plan_key(batches[:-1] ) != k0


def test_tuning_contexts_at_the_abi_without_gpu():
  """cmhse_ctx_* (include/cmhse_hip.h): a context is a private copy of the crossovers taken at
  creation; tuning it or the process defaults does not touch the other; cmhse_ctx_enter makes it the
  calling thread's current context (what every crossover read inside the library then uses) and
  returns the previous one; other threads keep the defaults.  No GPU involved."""
  import ctypes
  import threading
  from cmhse_amd import _lib
  lib = _lib.load()
  old = ctypes.c_int32(0)
  assert lib.cmhse_tune(b'mid_units', -1, ctypes.byref(old)) == 0
  default = old.value
  ctx = lib.cmhse_ctx_create()
  assert ctx
  try:
    assert lib.cmhse_ctx_tune(ctx, b'mid_units', 8, ctypes.byref(old)) == 0 and old.value == default
    assert lib.cmhse_ctx_tune(ctx, b'mid_units', -1, ctypes.byref(old)) == 0 and old.value == 8
    assert lib.cmhse_tune(b'mid_units', -1, ctypes.byref(old)) == 0 and old.value == default   # defaults untouched
    assert lib.cmhse_ctx_tune(ctx, b'no_such_knob', 1, None) == -1
    assert lib.cmhse_ctx_tune(None, b'mid_units', 1, None) == -1
    # workspace sizing reads the CURRENT context: mid_max_seqs bounds the hoisted-projection region
    size = lambda: lib.cmhse_gru_pool_workspace(5000, 40, 100000, 512, 256, 0)
    outside = size()
    assert lib.cmhse_ctx_tune(ctx, b'mid_max_seqs', 0, None) == 0
    prev = lib.cmhse_ctx_enter(ctx)
    assert not prev
    try:
      inside = size()
      seen = []
      t = threading.Thread(target=lambda: seen.append(size()))     # another thread: no context
      t.start(); t.join()
    finally:
      assert lib.cmhse_ctx_enter(prev) == ctx
    assert inside < outside and seen == [outside] and size() == outside
  finally:
    lib.cmhse_ctx_destroy(ctx)


def test_device_code_holds_neither_half_of_the_lost_update_pair():
  """profiles/r05_bf16_mfma_bystander.txt: on gfx950 a v_pk_fma_f32 loses updates while another wave of its SIMD
  issues the double-rate matrix instructions.  The shipped library contains neither (build.py DEVICE_FLAGS,
  nt_core.hpp::mfma_bf16_16k): disassemble its three code objects and look."""
  from cmhse_amd import build
  lib = build.build()
  assert len(build.code_objects(lib)) == len(build.SOURCES)
  assert build.audit_isa(lib) == {}
  # the audit does see what it is looking for: the matrix instructions the library DOES use
  assert build.audit_isa(lib, forbidden=('v_mfma_f32_32x32x2_f32', 'v_mfma_f32_32x32x8_bf16_1k'))


def test_bidirectional_flag_means_what_it_means_upstream():
  """layers.py:27-34, 70-79, 165-172: Seq2Seq and Attention build a bidirectional nn.GRU (state-dict
  keys `rnn.*_reverse`), Maxout stores the flag and stays unidirectional; Attention's forward cannot
  work with it upstream (its `lin` is H -> H) and says so here; a bidirectional layer is refused by
  the fused / grouped entry points (no reference call site uses one, model.py:107-114)."""
  import torch
  from cmhse_amd import layers
  s2s = layers.Seq2Seq(6, 8, rnn_bidirectional=True)
  att = layers.Attention(6, 8, rnn_bidirectional=True)
  mo = layers.Maxout(6, 8, rnn_bidirectional=True)
  assert s2s.bidirectional and att.bidirectional and mo.bidirectional
  assert 'rnn.weight_hh_l0_reverse' in s2s.state_dict() and 'rnn.weight_ih_l0_reverse' in att.state_dict()
  assert not any(k.endswith('_reverse') for k in mo.state_dict())
  x, lens = torch.zeros(2, 3, 6), torch.tensor([3, 1])
  with pytest.raises(RuntimeError, match='fails upstream'):
    att(x, lens)
  with pytest.raises(RuntimeError, match='Expected hidden size'):
    s2s(x, lens, torch.zeros(2, 8))
  with pytest.raises(RuntimeError, match='bidirectional'):
    s2s.call_rows(torch.zeros(4, 6), [3, 1])
  assert layers.Seq2Seq(6, 8).state_dict().keys() == layers.Maxout(6, 8, rnn_bidirectional=True).state_dict().keys()
