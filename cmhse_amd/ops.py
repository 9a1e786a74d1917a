"""Thin host-side wrappers over the C ABI (include/cmhse_hip.h): argument marshalling, the
pack_padded_sequence-style schedule, workspaces.  No arithmetic happens here and there is no CPU
fallback — tensors must live on the MI355X.
"""
from __future__ import annotations

import ctypes
import threading

import numpy as np
import torch

from . import _lib
from ._lib import (MATH_BF16X3, MAX_JOBS, POOL_ALL, POOL_ATTN, POOL_LAST, POOL_MAX, POOL_OF,  # noqa: F401
                   SAVE_FOR_BACKWARD)


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Pinned staging of upload(): ONE page-locked arena cut into fixed slots, taken round-robin; a
# slot is reused only after the copy that last read it has completed (its event).  Page-locking
# memory (hipHostMalloc) while the GPU is busy was measured at 5-100 ms per call on the GPU box,
# and torch's own pinned cache hands a block back only after the copy's event has completed — late,
# when the copy is queued behind a pass's kernels — so it kept allocating in the middle of passes.
# The arena is allocated once, at the first upload (model set-up / warm-up).
_SLOT_BYTES = 2 << 20
_N_SLOTS = 64
_ARENA = {'buf': None, 'np': None, 'events': None, 'next': 0}


def _staging_slot(nbytes):
  """((pinned uint8 tensor view of >= nbytes, the same bytes as a NumPy array), event to record
  after the copy)."""
  if nbytes > _SLOT_BYTES:    # larger than any schedule array of this path: a block of its own
    block = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    return (block, block.numpy()), torch.cuda.Event()
  if _ARENA['buf'] is None:
    # everything that can stall later is paid here, once: page-locking, the first touch of every
    # page (a fresh slot's first copy ran at 0.1 GB/s), and the creation of the slots' events
    # (torch creates an event at its first record; one such creation was seen to take 65 ms)
    _ARENA['buf'] = torch.empty(_SLOT_BYTES * _N_SLOTS, dtype=torch.uint8).pin_memory()
    _ARENA['buf'].zero_()
    _ARENA['np'] = _ARENA['buf'].numpy()      # the same pinned bytes as a NumPy array (upload())
    _ARENA['events'] = [torch.cuda.Event() for _ in range(_N_SLOTS)]
    for ev in _ARENA['events']:
      ev.record()
  i = _ARENA['next']
  _ARENA['next'] = (i + 1) % _N_SLOTS
  ev = _ARENA['events'][i]
  ev.synchronize()            # 63 uploads ago: complete unless the host is far ahead of the GPU
  return (_ARENA['buf'][i * _SLOT_BYTES:(i + 1) * _SLOT_BYTES],
          _ARENA['np'][i * _SLOT_BYTES:(i + 1) * _SLOT_BYTES]), ev


def upload(arr, device):
  """Small host array -> device without blocking the host: pinned staging + async copy (a pageable
  copy would make the host wait for everything already queued on the stream)."""
  if torch.device(device).type != 'cuda':
    return torch.from_numpy(arr).to(device)
  arr = np.ascontiguousarray(arr)
  n = arr.nbytes
  if n == 0:
    return torch.from_numpy(arr).to(device)
  (buf, buf_np), ev = _staging_slot(n)
  # host -> pinned staging with NumPy (one memcpy on this thread).  torch's CPU copy_ splits a
  # copy of this size (a 22 k-sequence schedule is 0.5 MB) over its intra-op thread pool, and
  # waking that pool on a many-core host took 60-190 ms once in ~20 validation passes — with the
  # GPU idle behind it (tools/pass_jitter.py: stack samples of the slow passes).
  buf_np[:n] = arr.reshape(-1).view(np.uint8)
  view = buf[:n].view(torch.from_numpy(arr.reshape(-1)[:0]).dtype).view(arr.shape)
  dev = view.to(device, non_blocking=True)
  ev.record(torch.cuda.current_stream(dev.device))
  return dev


def _require_cuda(t, name):
  if not isinstance(t, (torch.Tensor, Ragged)) or not t.is_cuda:
    raise RuntimeError('cmhse_amd: `%s` must be a tensor on the GPU (no CPU fallback; the HIP '
                       'path is the only implementation)' % name)


def _f32c(t, name):
  _require_cuda(t, name)
  if t.dtype != torch.float32:
    t = t.float()
  return t.contiguous()


def tune(name, value=-1):
  """cmhse_tune / cmhse_ctx_tune: set (value >= 0) or read (value < 0) one kernel-shape crossover;
  returns its previous value.  Inside a `with TuneContext(...)` block the call goes to THAT context
  (the one every library call of this thread reads); outside, to the process defaults new contexts
  are initialised from.  Names: tiny_max_seqs, mid_max_seqs, mid_units, mid_waves,
  tall_tile_min_wgs, mid_tall_min_seqs, bwd_mid_max_seqs, bwd_split_min_seqs, bwd_tail_min_steps,
  fwd_tail_min_steps, bwd_chunk_rows, xproj_chunk_rows, tn_rows_bm, chain_min_steps, early_xproj,
  chain_tall_min_wgs, pull_waves, resident_timeout_ms (include/cmhse_hip.h)."""
  ctx = TuneContext.current()
  if ctx is not None:
    return ctx.tune(name, value)
  old = ctypes.c_int32(0)
  _lib.check(_lib.load().cmhse_tune(name.encode(), int(value), ctypes.byref(old)), 'cmhse_tune(%s)' % name)
  return int(old.value)


def async_status(clear=False):
  """cmhse_async_status: 0, or CMHSE_ERR_TIMEOUT (-5) once a resident-kernel launch on the current
  device has given up at a grid barrier (its workgroups were not all on the chip: a shared GPU, a CU
  mask).  The status is asynchronous — it reflects launches that have run — and sticky until
  cleared; the gru forward / backward entry points check it themselves and raise.  Clearing a
  raised status also turns the multi-step kernels off for the process (one launch per time step
  from then on; ops.tune('chain_min_steps', 2) etc. turn them back on)."""
  return int(_lib.load().cmhse_async_status(1 if clear else 0))


class tuned(object):
  """Context manager: `with ops.tuned(tiny_max_seqs=0, mid_max_seqs=0): ...` moves crossovers for
  the block and puts the previous values back."""

  def __init__(self, **kw):
    self.kw, self.old = kw, {}

  def __enter__(self):
    for k, v in self.kw.items():
      self.old[k] = tune(k, v)
    return self

  def __exit__(self, *a):
    for k, v in self.old.items():
      tune(k, v)


class TuneContext(object):
  """A private copy of the library's kernel-shape crossovers (cmhse_ctx_*): `with ctx: ...` makes it
  the calling thread's current context for every library call in the block — workspace sizing and
  launches — whatever cmhse_tune, other threads or other contexts do meanwhile.  Two models tuned
  differently can live in one process, each running inside its own context.  A forward pass made
  inside a context remembers it, and its backward pass (which autograd runs on another thread)
  re-enters it.  Initialised from the process defaults (ops.tune) at creation."""
  _current = threading.local()

  def __init__(self, **kw):
    self._lib = _lib.load()
    self.handle = self._lib.cmhse_ctx_create()
    if not self.handle:
      raise MemoryError('cmhse_ctx_create failed')
    self._depth = 0          # live `with` blocks over all threads (the handle outlives them)
    for k, v in kw.items():
      self.tune(k, v)

  def tune(self, name, value=-1):
    old = ctypes.c_int32(0)
    _lib.check(self._lib.cmhse_ctx_tune(self.handle, name.encode(), int(value), ctypes.byref(old)),
               'cmhse_ctx_tune(%s)' % name)
    return int(old.value)

  # The enter / exit stack lives with the THREAD: two threads inside the same context
  # at once — autograd's device threads re-entering a forward pass's context, workers sharing one —
  # each restore their own previous context.
  def __enter__(self):
    tl = TuneContext._current
    stack = getattr(tl, 'stack', None)
    if stack is None:
      stack = tl.stack = []
    stack.append((self, self._lib.cmhse_ctx_enter(self.handle), getattr(tl, 'ctx', None)))
    tl.ctx = self
    self._depth += 1
    return self

  def __exit__(self, *a):
    tl = TuneContext._current
    me, prev_handle, prev_obj = tl.stack.pop()
    assert me is self, 'TuneContext blocks must nest'
    self._lib.cmhse_ctx_enter(prev_handle)
    tl.ctx = prev_obj
    self._depth -= 1

  def __del__(self):
    try:
      if self.handle and self._depth == 0:
        self._lib.cmhse_ctx_destroy(self.handle)
        self.handle = None
    except Exception:      # noqa: BLE001  (interpreter shutdown)
      pass

  @staticmethod
  def current():
    """The TuneContext the calling thread is inside of, or None."""
    return getattr(TuneContext._current, 'ctx', None)


class _in_ctx(object):
  """`with _in_ctx(ctx):` enters `ctx` unless it is None or already current (backward passes)."""

  def __init__(self, ctx):
    self.ctx = ctx if (ctx is not None and TuneContext.current() is not ctx) else None

  def __enter__(self):
    if self.ctx is not None:
      self.ctx.__enter__()

  def __exit__(self, *a):
    if self.ctx is not None:
      self.ctx.__exit__(*a)


_MATH_MODE = ['fp32']


def math_mode():
  """'fp32' (exact, default) or 'bf16x3' (3-term bf16 split on the matrix pipe for the large
  inference GEMMs; ~1e-6 on the embeddings).  Set with set_math_mode()."""
  return _MATH_MODE[0]


def set_math_mode(mode):
  if mode not in ('fp32', 'bf16x3'):
    raise ValueError("math mode must be 'fp32' or 'bf16x3'")
  _MATH_MODE[0] = mode


class SeqSchedule(object):
  """Host-side schedule of one packed GRU launch sequence.

  Mirrors what the reference does with `torch.sort(lengths, 0, True)` +
  `pack_padded_sequence(..., lens.tolist())` (/root/reference/layers.py:94-97): sequences sorted
  longest-first, so the sequences still active at step t are the prefix [0, step_count[t]).
  All small per-sequence arrays go to the device in ONE copy.
  """

  def __init__(self, lens, device, x_ptrs=None, tok_ptrs=None, h0_ptrs=None,
               out_rows_are_starts=False, src_ptrs=None):
    lens = np.asarray(lens, dtype=np.int64).reshape(-1)
    if lens.size == 0 or lens.min() < 1:
      raise ValueError('all sequence lengths must be >= 1 (pack_padded_sequence contract)')
    S = lens.size
    if int(lens.max()) < 65536:
      # 16-bit keys: numpy's stable sort is then a radix sort (10x faster on a 22k-sequence tower;
      # this runs on the host in front of every encoder launch)
      order = np.argsort((65535 - lens).astype(np.uint16), kind='stable')
    else:
      order = np.argsort(-lens, kind='stable')
    ls = lens[order]
    Tmax = int(ls[0])
    # step_count[t] = #{s : len_s > t}
    hist = np.bincount(ls, minlength=Tmax + 1)
    step_count = (S - np.cumsum(hist)[:Tmax]).astype(np.int32)
    step_off = np.zeros(Tmax + 1, dtype=np.int64)
    np.cumsum(step_count, out=step_off[1:])
    if step_off[-1] >= 2 ** 31:
      raise ValueError('packed batch too large for int32 row offsets')
    self.S, self.Tmax, self.sum_T = S, Tmax, int(step_off[-1])
    self.order = order
    self.step_count_host = np.ascontiguousarray(step_count)
    self.lens_sorted = ls

    # one packed metadata buffer:
    #   [x_rows u64 | h0_rows u64 | (src_rows u64) | lens i32 | out_row i32 | step_off i32]
    n64 = (3 if src_ptrs is not None else 2) * S
    n32 = 2 * S + (Tmax + 1)
    buf = np.zeros(n64 * 8 + n32 * 4, dtype=np.uint8)
    v64 = buf[:n64 * 8].view(np.uint64)
    v32 = buf[n64 * 8:].view(np.int32)
    src = x_ptrs if x_ptrs is not None else tok_ptrs
    v64[:S] = np.asarray(src, dtype=np.uint64)[order]
    if h0_ptrs is not None:
      v64[S:2 * S] = np.asarray(h0_ptrs, dtype=np.uint64)[order]
    if src_ptrs is not None:   # pinned-host source of every sequence (pull_steps)
      v64[2 * S:3 * S] = np.asarray(src_ptrs, dtype=np.uint64)[order]
    v32[:S] = ls
    if out_rows_are_starts:
      # CMHSE_POOL_ALL: sequence i (input order) owns output rows [start_i, start_i + len_i)
      starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
      v32[S:2 * S] = starts[order]
    else:
      v32[S:2 * S] = order
    v32[2 * S:] = step_off
    self.meta = upload(buf, device)
    base = self.meta.data_ptr()
    self.p_rows = base
    self.p_h0 = base + S * 8 if h0_ptrs is not None else None
    self.p_src = base + 2 * S * 8 if src_ptrs is not None else None
    self.p_lens = base + n64 * 8
    self.p_out_row = self.p_lens + S * 4
    self.p_step_off = self.p_out_row + S * 4
    self.is_tokens = x_ptrs is None


def step_counts(lens):
  """#{s : lens[s] > t} for t = 0 .. max(lens) - 1 (int64): the active sequences per time step of a
  length-sorted packed batch — SeqSchedule.step_count_host, without the schedule."""
  lens = np.asarray(lens, dtype=np.int64).reshape(-1)
  if lens.size == 0:
    return np.zeros(0, dtype=np.int64)
  hist = np.bincount(lens, minlength=int(lens.max()) + 1)
  return (lens.size - np.cumsum(hist)[:-1]).astype(np.int64)


def padded_row_ptrs(t):
  """Base address of every sequence of a contiguous padded batch [S, T, ...]."""
  S = t.shape[0]
  stride = t.stride(0) * t.element_size()
  return (np.uint64(t.data_ptr()) + np.arange(S, dtype=np.uint64) * np.uint64(stride))


class Ragged(object):
  """S sequences stored back to back WITHOUT padding: `data` is [sum(lens), I] float32 (frame
  features) or [sum(lens)] int64 (token ids); sequence s owns rows first[s] .. first[s]+lens[s].
  Stands in for a padded [S, Tmax, ...] loader tensor wherever the encoders take one: they address
  every sequence through its own base pointer (cmhse_seq_batch.x_rows / tok_rows), so the padding
  the reference's collate_fn writes (activity_net/data.py:114-150) is neither uploaded nor read."""

  def __init__(self, data, lens):
    self.data = data
    self.lens = np.asarray(lens, dtype=np.int64).reshape(-1)
    self.first = np.concatenate([[0], np.cumsum(self.lens)[:-1]]).astype(np.int64)
    if int(self.lens.sum()) != int(data.shape[0]):
      raise ValueError('Ragged: sum(lens) != rows of data')

  # the handful of tensor members the hot path uses on its big inputs
  @property
  def shape(self):
    tmax = int(self.lens.max()) if len(self.lens) else 0
    return (len(self.lens), tmax) + tuple(self.data.shape[1:])

  @property
  def is_cuda(self):
    return self.data.is_cuda

  @property
  def device(self):
    return self.data.device

  @property
  def dtype(self):
    return self.data.dtype

  def cuda(self, non_blocking=False):
    return self if self.data.is_cuda else Ragged(self.data.cuda(non_blocking=non_blocking), self.lens)

  def to(self, device, non_blocking=False):
    return Ragged(self.data.to(device, non_blocking=non_blocking), self.lens)

  def pin_memory(self):
    return Ragged(self.data.pin_memory(), self.lens)

  def is_pinned(self):
    return self.data.is_pinned()

  def is_contiguous(self):
    return self.data.is_contiguous()

  def numel(self):
    return self.data.numel()       # stored elements (no padding)

  def record_stream(self, stream):
    self.data.record_stream(stream)

  def row_bytes(self):
    return int(np.prod(self.data.shape[1:], dtype=np.int64)) * self.data.element_size()

  def row_ptrs(self):
    """Base address of every sequence (what padded_row_ptrs is for a padded batch)."""
    return np.uint64(self.data.data_ptr()) + self.first.astype(np.uint64) * np.uint64(self.row_bytes())

  def padded(self):
    """The reference's zero-padded [S, Tmax, ...] tensor: cmhse_pad_rows on the device; plain
    slicing for a host tensor (what collate_fn itself does)."""
    S, tmax = self.shape[0], self.shape[1]
    out_shape = (S, tmax) + tuple(self.data.shape[1:])
    if not self.data.is_cuda:
      out = torch.zeros(out_shape, dtype=self.data.dtype)
      for s in range(S):
        out[s, :self.lens[s]] = self.data[self.first[s]:self.first[s] + self.lens[s]]
      return out
    lib = _lib.load()
    data = self.data.contiguous()
    out = torch.empty(out_shape, dtype=data.dtype, device=data.device)
    first = upload(self.first, data.device)
    lens32 = upload(self.lens.astype(np.int32), data.device)
    rc = lib.cmhse_pad_rows(data.data_ptr(), first.data_ptr(), lens32.data_ptr(), S, tmax,
                            self.row_bytes(), out.data_ptr(), _stream())
    _lib.check(rc, 'cmhse_pad_rows')
    return out


def seq_row_ptrs(t):
  """Per-sequence base addresses of a loader tensor: padded [S, T, ...] or Ragged."""
  return t.row_ptrs() if isinstance(t, Ragged) else padded_row_ptrs(t)


def seq_row_ptrs_many(tensors):
  """np.concatenate([seq_row_ptrs(t) for t in tensors]) without one small NumPy array per tensor
  (a validation split hands over hundreds of loader batches per encoder)."""
  if not tensors:
    return np.zeros(0, dtype=np.uint64)
  if any(isinstance(t, Ragged) for t in tensors):
    return np.concatenate([seq_row_ptrs(t) for t in tensors])
  n = len(tensors)
  bases = np.fromiter((t.data_ptr() for t in tensors), dtype=np.uint64, count=n)
  counts = np.fromiter((t.shape[0] for t in tensors), dtype=np.int64, count=n)
  strides = np.fromiter((t.stride(0) * t.element_size() for t in tensors), dtype=np.uint64, count=n)
  starts = np.cumsum(counts) - counts
  within = (np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(starts, counts)).astype(np.uint64)
  return np.repeat(bases, counts) + within * np.repeat(strides, counts)


def seq_keep(t, dtype):
  """The storage to keep alive / hand to the kernels for one loader tensor, as `dtype`, contiguous."""
  if isinstance(t, Ragged):
    d = t.data.detach()
    d = d if d.dtype == dtype else d.to(dtype)
    return Ragged(d.contiguous(), t.lens)
  t = t.detach()
  t = t if t.dtype == dtype else t.to(dtype)
  return t.contiguous()


class StepTimers(object):
  """Measurement aid for bench.py: while active, every cmhse_gru_pool_fwd call gets a
  cmhse_timer around its per-step GRU kernels; `collect()` returns a list of
  (elapsed_ms, n_launches, [(Tmax, sum_T, I, H, had_h0, S) per request of the call],
  (tiled_ms, tiled_flops, tiled_bytes, tiled_launches)) and frees the timers."""
  active = None

  def __init__(self):
    self.items = []

  def __enter__(self):
    StepTimers.active = self
    return self

  def __exit__(self, *a):
    StepTimers.active = None

  def collect(self):
    lib = _lib.load()
    out = []
    for handle, meta in self.items:
      ms = ctypes.c_float(0.0)
      _lib.check(lib.cmhse_timer_elapsed_ms(handle, ctypes.byref(ms)), 'cmhse_timer_elapsed_ms')
      launches = int(lib.cmhse_timer_launches(handle))
      t_ms, t_flops, t_bytes = ctypes.c_float(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
      t_n = ctypes.c_int32(0)
      _lib.check(lib.cmhse_timer_tiled(handle, ctypes.byref(t_ms), ctypes.byref(t_flops),
                                       ctypes.byref(t_bytes), ctypes.byref(t_n)),
                 'cmhse_timer_tiled')
      lib.cmhse_timer_destroy(handle)
      out.append((ms.value, launches, meta, (t_ms.value, t_flops.value, t_bytes.value, t_n.value)))
    self.items = []
    return out


PULL_EVENT_TIMING = [False]     # tools/ab_host.py: timed events to see when each chunk landed
LAST_PULL_EVENTS = [None]
_LAST_PULL_TIMED = [False]


def pull_steps(sched, row_floats, copy_stream, chunk=8):
  """Queue the host -> HBM hand-over of a schedule built with `src_ptrs` (pinned host rows) on
  `copy_stream` and return {t0: torch event}: the event of the chunk that starts at step t0 (what
  cmhse_seq_batch.step_events_host takes).  Chunks grow from single steps up to `chunk` steps, so
  the first step's rows arrive after one step's worth of PCIe time, not after a whole chunk's."""
  lib = _lib.load()
  if sched.p_src is None:
    raise ValueError('schedule was built without src_ptrs')
  events = {}
  # the previous call's events are recycled (creating one while the GPU is busy can stall the host
  # for tens of ms): the step launcher's waits on them were captured when they were queued
  pool = []
  if not PULL_EVENT_TIMING[0] and LAST_PULL_EVENTS[0] and not _LAST_PULL_TIMED[0]:
    pool = list(LAST_PULL_EVENTS[0].values())
  sched.meta.record_stream(copy_stream)
  bounds, t = [], 0
  while t < sched.Tmax:
    # a consumer trails the copy by one chunk, and it can only stall while the pipeline fills:
    # single steps first, the full chunk once the copy is safely ahead
    c = min(chunk, 1 if t < 8 else (2 if t < 16 else (4 if t < 32 else chunk)))
    bounds.append((t, min(sched.Tmax, t + c)))
    t += c
  for t0, t1 in bounds:
    rc = lib.cmhse_pull_steps(sched.p_src, sched.p_rows, sched.p_lens,
                              int(sched.step_count_host[t0]), row_floats, t0, t1,
                              ctypes.c_void_p(copy_stream.cuda_stream))
    _lib.check(rc, 'cmhse_pull_steps')
    ev = pool.pop() if pool else torch.cuda.Event(enable_timing=PULL_EVENT_TIMING[0])
    ev.record(copy_stream)
    events[t0] = ev
  LAST_PULL_EVENTS[0] = events
  _LAST_PULL_TIMED[0] = PULL_EVENT_TIMING[0]
  return events


PUSH_WORKGROUPS = [8]      # workgroups x wavefronts of cmhse_push_rows (tools/api_path_profile.py --push_wgs / --push_waves)
PUSH_WAVES = [1]
PUSH_WORKGROUPS_LATE = [32]   # ... for the two level-2 matrices, which leave under the ranking (no chain to disturb)


def push_rows(src, dst_pinned, stream, workgroups=None):
  """cmhse_push_rows: contiguous device tensor `src` -> page-locked host tensor `dst_pinned` (same
  byte count) as a small kernel on `stream` (a torch stream).  Asynchronous: the caller orders
  `stream` behind the producer of `src` and synchronises it before reading `dst_pinned`.
  `workgroups`: PUSH_WORKGROUPS by default (beside a step chain: few); more when nothing latency-
  bound runs beside the copy."""
  _require_cuda(src, 'src')
  if dst_pinned.is_cuda or not dst_pinned.is_pinned():
    raise RuntimeError('push_rows: the destination must be page-locked host memory')
  if not src.is_contiguous() or not dst_pinned.is_contiguous():
    raise ValueError('push_rows: contiguous tensors only')
  nbytes = src.numel() * src.element_size()
  if nbytes != dst_pinned.numel() * dst_pinned.element_size():
    raise ValueError('push_rows: byte counts differ')
  rc = _lib.load().cmhse_push_rows(src.data_ptr(), dst_pinned.data_ptr(), nbytes,
                                   PUSH_WORKGROUPS[0] if workgroups is None else int(workgroups),
                                   PUSH_WAVES[0], ctypes.c_void_p(stream.cuda_stream))
  _lib.check(rc, 'cmhse_push_rows')


def rows_differ(pairs):
  """cmhse_rows_differ over [(a, b)]: a = page-locked host tensor (read in place over PCIe) or device
  tensor, b = device tensor of the same byte count.  True when any pair differs in any byte.  One
  small readback; synchronises the current stream."""
  lib = _lib.load()
  dev = pairs[0][1].device
  flag = torch.zeros(1, dtype=torch.int32, device=dev)
  for a, b in pairs:
    _require_cuda(b, 'b')
    if not (a.is_cuda or a.is_pinned()) or not a.is_contiguous() or not b.is_contiguous():
      raise RuntimeError('rows_differ: `a` must be contiguous device or page-locked host memory')
    na, nb = a.numel() * a.element_size(), b.numel() * b.element_size()
    if na != nb:
      return True
    _lib.check(lib.cmhse_rows_differ(a.data_ptr(), b.data_ptr(), na, flag.data_ptr(), _stream()),
               'cmhse_rows_differ')
  return bool(flag.item())


def _prepare_fwd(weights, pool_mode, lens, I, H, device, x_ptrs=None, tok_ptrs=None,
                 emb_table=None, h0_ptrs=None, out=None, save_for_backward=False,
                 constant_input=False, sched=None, step_events=None, side=True, step_plan=None):
  """Build the ctypes request of one cmhse_gru_pool_fwd call.  Returns (job dict, timer meta).
  `sched`: a prebuilt SeqSchedule for these sequences (else built here); `step_events`:
  {step: torch.cuda.Event} the step's launch must wait for (chunked upload, pull_steps);
  `step_plan`: active sequences per time step of the WHOLE set these sequences are a share of
  (step_counts() of all its lengths; cmhse_seq_batch.step_plan_host) — the kernel kind of every step
  is then chosen from it, so a sequence is encoded bit for bit the same in any share."""
  lib = _lib.load()
  if sched is None:
    sched = SeqSchedule(lens, device, x_ptrs, tok_ptrs, h0_ptrs,
                        out_rows_are_starts=(pool_mode == POOL_ALL))
  S = sched.S
  if out is None:
    n_out = sched.sum_T if pool_mode == POOL_ALL else S
    out = torch.empty(n_out, H, dtype=torch.float32, device=device)
  mode_flags = pool_mode | (SAVE_FOR_BACKWARD if save_for_backward else 0)
  if math_mode() == 'bf16x3' and not save_for_backward:
    mode_flags |= MATH_BF16X3
  ws_bytes = lib.cmhse_gru_pool_workspace(S, sched.Tmax, sched.sum_T, I, H, mode_flags)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)

  w = _lib.GruWeights()
  keep = []
  for name in ['w_ih', 'w_hh', 'b_ih', 'b_hh']:
    t = _f32c(weights[name], name)
    keep.append(t)
    setattr(w, name, t.data_ptr())
  if pool_mode == POOL_ATTN:
    for name in ['w_lin', 'b_lin', 'w_att']:
      t = _f32c(weights[name], name)
      keep.append(t)
      setattr(w, name, t.data_ptr())
  b = _lib.SeqBatch()
  b.S, b.Tmax, b.I, b.H = S, sched.Tmax, I, H
  if sched.is_tokens:
    emb_table = _f32c(emb_table, 'emb_table')
    keep.append(emb_table)
    b.tok_rows = sched.p_rows
    b.emb_table = emb_table.data_ptr()
    b.vocab = emb_table.shape[0]
  else:
    b.x_rows = sched.p_rows
    b.x_step_floats = 0 if constant_input else I
  b.h0_rows = sched.p_h0
  b.lens, b.out_row, b.step_off = sched.p_lens, sched.p_out_row, sched.p_step_off
  b.step_count_host = sched.step_count_host.ctypes.data
  if step_plan is not None and not save_for_backward:
    plan = np.zeros(sched.Tmax, dtype=np.int32)
    sp = np.asarray(step_plan, dtype=np.int32)[:sched.Tmax]
    plan[:len(sp)] = sp
    # (a share's own counts are a lower bound: a caller whose plan was built from other lengths gets
    # CMHSE_ERR_ARG from the library rather than a silently different schedule)
    b.step_plan_host = plan.ctypes.data
    keep.append(plan)
  ev_arr = None
  if step_events:
    ev_arr = (ctypes.c_void_p * sched.Tmax)()
    for t, ev in step_events.items():
      ev_arr[t] = ev.cuda_event
    b.step_events_host = ctypes.cast(ev_arr, ctypes.c_void_p)
    keep.append((ev_arr, step_events))
  ctx = dict(sched=sched, ws=ws, keep=keep, H=H, I=I, batch=b, weights=w, pool_mode=pool_mode,
             device=device, mode_flags=mode_flags, tune_ctx=TuneContext.current())
  job = dict(b=b, w=w, mode_flags=mode_flags, out=out, ws=ws, ws_bytes=ws_bytes, ctx=ctx, side=side)
  meta = (sched.Tmax, sched.sum_T, I, H, h0_ptrs is not None, S)
  return job, meta


def gru_pool_fwd(weights, pool_mode, lens, I, H, device, **kw):
  """cmhse_gru_pool_fwd.  `weights`: dict with w_ih, w_hh, b_ih, b_hh (+ w_lin, b_lin, w_att).
  Keywords: x_ptrs | tok_ptrs + emb_table, h0_ptrs, out, save_for_backward, constant_input.
  Returns (out [S,H], ctx) where ctx keeps the workspace (packed hidden states) and
  schedule; with `save_for_backward` the workspace also keeps what gru_pool_bwd needs."""
  return gru_pool_fwd_multi([dict(weights=weights, pool_mode=pool_mode, lens=lens, I=I, H=H,
                                  device=device, **kw)])[0]


def gru_pool_fwd_multi(requests, tail_stream=None, job_streams=None, join=True, hold=None, ready_events=None):
  """cmhse_gru_pool_fwd_multi: `requests` is a list of keyword dicts (the arguments of
  gru_pool_fwd) for INDEPENDENT encoders; their time steps share launches.  Returns a list of
  (out, ctx), bit-identical to separate gru_pool_fwd calls.  With `tail_stream` (a torch stream,
  ideally high priority) the steps left over when an attention-pooled request's shorter chain
  has ended continue there while that request's pooling pass runs on the current stream; the
  call rejoins the current stream before it returns.  With `job_streams` (one torch stream per
  request) every request runs on its own stream, the launches of all requests interleaved step by
  step (cmhse_gru_job.stream); forked from and joined into the current stream inside the call.
  join=False (CMHSE_NO_JOIN) leaves the job streams un-joined: the outputs are ready on their job
  stream only — for a caller that queues the consumer on the same stream next and joins later; it
  must pass `hold`, a list that receives everything the queued work uses and that it keeps until
  that later join.  `ready_events`: one torch.cuda.Event (already recorded once, so that it owns a
  handle) or None per request; the call records it where that request's output becomes final
  (cmhse_gru_job.out_ready_event) — early for a request whose chain ends while others still step."""
  lib = _lib.load()
  if not 1 <= len(requests) <= MAX_JOBS:
    raise ValueError('gru_pool_fwd_multi takes 1..%d requests' % MAX_JOBS)
  prepared = [_prepare_fwd(**r) for r in requests]
  jobs = (_lib.GruJob * len(prepared))()
  for k, (job, _) in enumerate(prepared):
    jobs[k].seqs = ctypes.pointer(job['b'])
    jobs[k].weights = ctypes.pointer(job['w'])
    jobs[k].pool_mode = job['mode_flags'] | (0 if join or job_streams is None else _lib.NO_JOIN)
    jobs[k].out = job['out'].data_ptr()
    jobs[k].workspace = job['ws'].data_ptr()
    jobs[k].workspace_bytes = job['ws_bytes']
    if job_streams is not None:
      jobs[k].stream = ctypes.c_void_p(job_streams[k].cuda_stream)
    if ready_events is not None and ready_events[k] is not None:
      if not ready_events[k].cuda_event:
        raise ValueError('ready_events: record the event once before handing it over')
      jobs[k].out_ready_event = ctypes.c_void_p(ready_events[k].cuda_event)
    if job['mode_flags'] & SAVE_FOR_BACKWARD and job['side']:   # a training call: throughput work beside the chain
      side = side_stream(job_streams[k] if job_streams is not None else None)
      if side is not None:
        jobs[k].side_stream = ctypes.c_void_p(side.cuda_stream)
    if tail_stream is not None and len(prepared) > 1:
      jobs[k].tail_stream = ctypes.c_void_p(tail_stream.cuda_stream)
      for t in [job['out'], job['ws'], job['ctx']['sched'].meta] + job['ctx']['keep']:
        if isinstance(t, torch.Tensor):
          t.record_stream(tail_stream)
  if StepTimers.active is not None:
    handle = lib.cmhse_timer_create()
    prepared[0][0]['b'].step_timer = handle
    StepTimers.active.items.append((handle, [m for _, m in prepared]))
  if len(prepared) == 1 and job_streams is None and not jobs[0].side_stream and not jobs[0].out_ready_event:
    job = prepared[0][0]
    rc = lib.cmhse_gru_pool_fwd(ctypes.byref(job['b']), ctypes.byref(job['w']), job['mode_flags'],
                                job['out'].data_ptr(), job['ws'].data_ptr(), job['ws_bytes'],
                                _stream())
    _lib.check(rc, 'cmhse_gru_pool_fwd')
  else:
    rc = lib.cmhse_gru_pool_fwd_multi(jobs, len(prepared), _stream())
    _lib.check(rc, 'cmhse_gru_pool_fwd_multi')
  prepared[0][0]['b'].step_timer = None
  if not join:
    if hold is None:
      raise ValueError('gru_pool_fwd_multi(join=False) needs a `hold` list')
    hold.append(prepared)
  return [(job['out'], job['ctx']) for job, _ in prepared]


def saved_region(fctx, name):
  """A named region of a forward call's workspace (cmhse_gru_pool_ws_region) as a tensor view:
  'hs' [sum_T, H], 'gates' [sum_T, 4H], 'v' [sum_T, H] float32, 'argmax' [S, H] int32 (rows in the
  schedule's SORTED order: fctx['sched'].order maps them to input order).  None when the call's mode
  did not keep it.  For tests and tools."""
  lib = _lib.load()
  sched, H, I = fctx['sched'], fctx['H'], fctx['I']
  flags = fctx.get('mode_flags', fctx['pool_mode'])
  off, nbytes = ctypes.c_size_t(0), ctypes.c_size_t(0)
  rc = lib.cmhse_gru_pool_ws_region(sched.S, sched.Tmax, sched.sum_T, I, H, flags, name.encode(),
                                    ctypes.byref(off), ctypes.byref(nbytes))
  _lib.check(rc, 'cmhse_gru_pool_ws_region(%s)' % name)
  if nbytes.value == 0:
    return None
  raw = fctx['ws'][off.value:off.value + nbytes.value]
  if name == 'argmax':
    return raw.view(torch.int32).view(sched.S, H)
  return raw.view(torch.float32).view(-1, 4 * H if name == 'gates' else H)


def l2norm_rows(x, out=None):
  """torch.nn.functional.normalize(x) on device (cmhse_l2norm_rows)."""
  lib = _lib.load()
  x = _f32c(x, 'x')
  if x.dim() != 2:
    raise ValueError('l2norm_rows expects [rows, cols]')
  if out is None:
    out = torch.empty_like(x)
  rc = lib.cmhse_l2norm_rows(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.shape[1],
                             _stream())
  _lib.check(rc, 'cmhse_l2norm_rows')
  return out


def gather_rows(table, ids):
  """nn.Embedding lookup on device (cmhse_gather_rows): table[ids] with ids of any shape."""
  lib = _lib.load()
  table = _f32c(table, 'table')
  _require_cuda(ids, 'ids')
  ids = ids.contiguous()
  if ids.dtype != torch.int64:
    ids = ids.long()
  out = torch.empty(tuple(ids.shape) + (table.shape[1],), dtype=torch.float32,
                    device=table.device)
  rc = lib.cmhse_gather_rows(table.data_ptr(), ids.data_ptr(), ids.numel(), table.shape[1],
                             table.shape[0], out.data_ptr(), _stream())
  _lib.check(rc, 'cmhse_gather_rows')
  return out


class SimTimers(object):
  """Measurement aid for bench.py: while active, every sim_rank call gets a cmhse_timer around its
  counting pass (sim_kernel<Rank>); collect() returns [(elapsed_ms, nrows, M, D)] and frees them."""
  active = None

  def __init__(self):
    self.items = []

  def __enter__(self):
    SimTimers.active = self
    return self

  def __exit__(self, *a):
    SimTimers.active = None

  def collect(self):
    lib = _lib.load()
    out = []
    for handle, meta in self.items:
      ms = ctypes.c_float(0.0)
      _lib.check(lib.cmhse_timer_elapsed_ms(handle, ctypes.byref(ms)), 'cmhse_timer_elapsed_ms')
      lib.cmhse_timer_destroy(handle)
      out.append((ms.value,) + meta)
    self.items = []
    return out


def sim_rank(a, b, row0=0, nrows=None):
  """Ranks / top-1 of the stripe [row0,row0+nrows) of a @ b.T (cmhse_sim_rank).
  Returns int32 device tensors (rank, top1)."""
  lib = _lib.load()
  a = _f32c(a, 'a')
  b = _f32c(b, 'b')
  N, D = a.shape
  M = b.shape[0]
  if b.shape[1] != D:
    raise ValueError('embedding widths differ')
  if nrows is None:
    nrows = N - row0
  rank = torch.empty(nrows, dtype=torch.int32, device=a.device)
  top1 = torch.empty(nrows, dtype=torch.int32, device=a.device)
  if nrows == 0:
    return rank, top1
  ws_bytes = lib.cmhse_sim_rank_workspace(nrows)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
  timer = None
  if SimTimers.active is not None:
    timer = lib.cmhse_timer_create()
    SimTimers.active.items.append((timer, (nrows, M, D)))
  rc = lib.cmhse_sim_rank_ex(a.data_ptr(), b.data_ptr(), N, M, D, row0, nrows, rank.data_ptr(),
                             top1.data_ptr(), ws.data_ptr(), ws_bytes, _stream(), timer)
  _lib.check(rc, 'cmhse_sim_rank')
  return rank, top1


def cosine_sim(im, s):
  """loss.cosine_sim on device: im @ s.T, exact fp32 (cmhse_cosine_sim)."""
  lib = _lib.load()
  im = _f32c(im, 'im')
  s = _f32c(s, 's')
  n, D = im.shape
  m = s.shape[0]
  out = torch.empty(n, m, dtype=torch.float32, device=im.device)
  rc = lib.cmhse_cosine_sim(im.data_ptr(), s.data_ptr(), n, m, D, out.data_ptr(), _stream())
  _lib.check(rc, 'cmhse_cosine_sim')
  return out


def contrastive_fwd(im, s, margin, max_violation, norm, want_scores=False):
  """loss.ContrastiveLoss.forward on device (cmhse_contrastive_fwd): 0-d loss tensor
  (and the score matrix when `want_scores`)."""
  lib = _lib.load()
  im = _f32c(im, 'im')
  s = _f32c(s, 's')
  n, D = im.shape
  if s.shape[0] != n or s.shape[1] != D:
    raise ValueError('ContrastiveLoss needs im and s of identical shape (diag view, loss.py:89)')
  loss = torch.empty((), dtype=torch.float32, device=im.device)
  scores = torch.empty(n, n, dtype=torch.float32, device=im.device) if want_scores else None
  ws_bytes = lib.cmhse_contrastive_workspace(n)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=im.device)
  rc = lib.cmhse_contrastive_fwd(im.data_ptr(), s.data_ptr(), n, D, float(margin),
                                 int(bool(max_violation)), int(bool(norm)), loss.data_ptr(),
                                 scores.data_ptr() if want_scores else None, ws.data_ptr(),
                                 ws_bytes, _stream())
  _lib.check(rc, 'cmhse_contrastive_fwd')
  return (loss, scores) if want_scores else loss


def contrastive_blocks_fwd(im, s, block_sizes, margin, max_violation, norm, keep=False):
  """Per-block ContrastiveLoss over consecutive row blocks of im / s (cmhse_contrastive_blocks_fwd):
  returns a float32 device tensor [len(block_sizes)]; with `keep` also the state
  contrastive_blocks_bwd needs (the stored score blocks live in the call's workspace)."""
  lib = _lib.load()
  im = _f32c(im, 'im')
  s = _f32c(s, 's')
  sizes = np.asarray(block_sizes, dtype=np.int64)
  if sizes.sum() != im.shape[0] or im.shape != s.shape:
    raise ValueError('block sizes must tile the rows of im and s')
  nb, max_n = len(sizes), int(sizes.max())
  off = np.zeros(nb + 1, dtype=np.int32)
  np.cumsum(sizes, out=off[1:])
  off_d = upload(off, im.device)
  losses = torch.empty(nb, dtype=torch.float32, device=im.device)
  ws_bytes = lib.cmhse_contrastive_blocks_workspace(nb, max_n)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=im.device)
  rc = lib.cmhse_contrastive_blocks_fwd(im.data_ptr(), s.data_ptr(), off_d.data_ptr(), nb, max_n,
                                        im.shape[1], float(margin), int(bool(max_violation)),
                                        int(bool(norm)), losses.data_ptr(), ws.data_ptr(),
                                        ws_bytes, _stream())
  _lib.check(rc, 'cmhse_contrastive_blocks_fwd')
  if keep:
    return losses, dict(scores=ws, off=off_d, nb=nb, max_n=max_n)
  return losses


def contrastive_blocks_bwd(im, s, state, margin, max_violation, norm, grad_losses):
  """d(sum_b grad_losses[b] * loss_b) / d im, d s (cmhse_contrastive_blocks_bwd); `state` from
  contrastive_blocks_fwd(keep=True)."""
  lib = _lib.load()
  im = _f32c(im, 'im')
  s = _f32c(s, 's')
  g = _f32c(grad_losses, 'grad_losses').reshape(-1)
  nb, max_n = state['nb'], state['max_n']
  if g.numel() != nb:
    raise ValueError('one upstream gradient per block')
  d_im = torch.empty_like(im)
  d_s = torch.empty_like(s)
  ws_bytes = lib.cmhse_contrastive_blocks_bwd_workspace(nb, max_n)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=im.device)
  rc = lib.cmhse_contrastive_blocks_bwd(im.data_ptr(), s.data_ptr(), state['scores'].data_ptr(),
                                        state['off'].data_ptr(), nb, max_n, im.shape[1],
                                        float(margin), int(bool(max_violation)), int(bool(norm)),
                                        g.data_ptr(), d_im.data_ptr(), d_s.data_ptr(),
                                        ws.data_ptr(), ws_bytes, _stream())
  _lib.check(rc, 'cmhse_contrastive_blocks_bwd')
  return d_im, d_s


def step_losses_fwd(xs, terms, margin, max_violation, norm):
  """cmhse_step_losses_fwd: xs = the step's encoder outputs ([rows_e, D] float32, un-normalised),
  terms = [(a, b, weight)] indices into xs.  Returns (values [n_terms], total [1], state); the
  state (descriptor + workspace) is what step_losses_bwd needs."""
  lib = _lib.load()
  xs = [_f32c(x, 'x') for x in xs]
  if not 0 < len(xs) <= _lib.STEP_LOSS_MAX or not 0 < len(terms) <= _lib.STEP_LOSS_MAX:
    raise ValueError('step_losses: 1..%d embeddings and terms' % _lib.STEP_LOSS_MAX)
  d = _lib.StepLosses()
  d.n_emb, d.n_terms, d.D = len(xs), len(terms), int(xs[0].shape[1])
  for e, x in enumerate(xs):
    if x.dim() != 2 or x.shape[1] != d.D:
      raise ValueError('step_losses: every embedding is [rows, %d]' % d.D)
    d.x[e], d.rows[e] = x.data_ptr(), int(x.shape[0])
  for k, (a, b, w) in enumerate(terms):
    d.term_a[k], d.term_b[k], d.weight[k] = int(a), int(b), float(w)
  d.margin, d.max_violation, d.norm = float(margin), int(bool(max_violation)), int(bool(norm))
  ws_bytes = lib.cmhse_step_losses_workspace(ctypes.byref(d))
  if ws_bytes == 0:
    raise ValueError('step_losses: terms must pair embeddings with equal row counts')
  dev = xs[0].device
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
  out = torch.empty(len(terms) + 1, dtype=torch.float32, device=dev)
  rc = lib.cmhse_step_losses_fwd(ctypes.byref(d), out.data_ptr(), out.data_ptr() + 4 * len(terms),
                                 ws.data_ptr(), ws_bytes, _stream())
  _lib.check(rc, 'cmhse_step_losses_fwd')
  return out[:len(terms)], out[len(terms):], dict(desc=d, ws=ws, ws_bytes=ws_bytes, xs=xs)


def step_losses_bwd(state, grad_total):
  """d (grad_total * total) / d xs[e] for every e (cmhse_step_losses_bwd)."""
  lib = _lib.load()
  g = _f32c(grad_total, 'grad_total').reshape(-1)
  xs = state['xs']
  dxs = [torch.empty_like(x) for x in xs]
  ptrs = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in dxs])
  rc = lib.cmhse_step_losses_bwd(ctypes.byref(state['desc']), g.data_ptr(), ptrs,
                                 state['ws'].data_ptr(), state['ws_bytes'], _stream())
  _lib.check(rc, 'cmhse_step_losses_bwd')
  return dxs


def gru_pool_bwd(fctx, dout, dx_ptrs=None, d_emb_table=None, want_dh0=False):
  """cmhse_gru_pool_bwd for a forward run with save_for_backward=True.
  dx_ptrs: numpy uint64 [S] (input order) addresses receiving d x of step 0 of each sequence.
  Returns (grads dict of fresh tensors, dh0 [S,H] or None)."""
  return gru_pool_bwd_multi([dict(fctx=fctx, dout=dout, dx_ptrs=dx_ptrs, d_emb_table=d_emb_table,
                                  want_dh0=want_dh0)])[0]


# The package's streams.  The HIP runtime serves a process with a handful of hardware queues
# (four by default) and binds a stream to one of them when the stream is first used; which queue
# it gets depends on what ran before.  Streams that must run SIDE BY SIDE — the two towers of a
# training step, a chain and the throughput work beside it, the text tail / the host pull beside the
# caller's stream in a validation pass — must not share a queue: streams first used after a
# validation pass had used a high-priority stream all landed on ONE queue (training steps 2x
# slower), and binding the training streams first cost the validation pass 3-10 % the same way.
# So the package owns exactly FOUR streams per device, created together and each used once right
# away (first call: VSE.__init__), and every role is one of them:
#   [0] tower A (visual) of a training step | the text tail of a validation pass (tail_stream)
#   [1] tower B (text)                      | the host-pull copy stream of a validation pass
#   [2] companion of [0] (and of any other stream): weight-gradient products, projection chunks
#   [3] companion of [1]
# With the default (null) stream that is five streams on four queues: [3] shares the null stream's
# queue, which is idle while a training step's towers run.  All four have the default priority:
# what rounds 1-2 bought with high-priority side streams was a queue of their own, not the
# priority (validation pass 286.7-290.0 ms with default priority against 286.2-293.1 with [0], [1]
# raised, alternating runs on one box) — and raised tower streams cost a training step 1-2 ms
# (profiles/r03_stream_binding.txt).
SIDE_STREAMS = [True]
_STREAM_SET = {}


def stream_set(device=None):
  device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
  key = device.index if device.index is not None else torch.cuda.current_device()
  st = _STREAM_SET.get(key)
  if st is None:
    dev = torch.device('cuda', key)
    st = tuple(torch.cuda.Stream(dev) for _ in range(4))
    for s in st:
      with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
    _STREAM_SET[key] = st
  return st


_COPY_STREAM = {}


def copy_stream(device=None):
  """The package's host-to-device hand-over stream for TRAINING (collate.DevicePrefetcher, the host
  pull of VSE.train_emb): a fifth stream, created after the set of four and used once right away.
  Why not [3] of the set: [3] shares the caller's (null) stream's hardware queue, and a 1.7 ms
  batch upload queued there sits in front of the next step's first launches on the null stream
  (measured, tools/host_lead.py: a prefetched ICEP step 1.087x the resident one on [3], 1.010x on a
  stream of its own)."""
  device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
  key = device.index if device.index is not None else torch.cuda.current_device()
  st = _COPY_STREAM.get(key)
  if st is None:
    stream_set(device)
    dev = torch.device('cuda', key)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
      torch.zeros(1, device=dev)
    _COPY_STREAM[key] = st
  return st


def side_stream(cur=None):
  """The companion stream of `cur` (default: the current stream) for throughput work beside a
  latency chain (cmhse_gru_job.side_stream / cmhse_gru_bwd_job.side_stream); None when disabled.
  The library forks to it and joins it back inside the call, so the caller's stream semantics do
  not change and no record_stream bookkeeping is needed (nothing the side stream touches is freed
  before the join)."""
  if not SIDE_STREAMS[0]:
    return None
  cur = cur or torch.cuda.current_stream()
  st = stream_set(cur.device)
  return st[3] if cur.cuda_stream == st[1].cuda_stream else st[2]


def prepare_bwd(requests):
  """The allocations, zero fills and table uploads of gru_pool_bwd_multi(requests), done now —
  everything of that call that is queued on the caller's stream; pass the result as `prepared`."""
  return [_prepare_bwd(**r) for r in requests]


def gru_pool_bwd_multi(requests, job_streams=None, join=True, hold=None, prepared=None):
  """cmhse_gru_pool_bwd_multi: `requests` = keyword dicts of gru_pool_bwd for INDEPENDENT encoders
  (the two towers of a training step); their BPTT steps share launches — or, with `job_streams`
  (one torch stream per request), run as separate chains on those streams with their launches
  interleaved step by step.  Returns [(grads, dh0)].  join / hold: as in gru_pool_fwd_multi.
  Runs inside the TuneContext the forward pass was made in (autograd calls this on its own thread)."""
  with _in_ctx(requests[0]['fctx'].get('tune_ctx') if requests else None):
    return _gru_pool_bwd_multi(requests, job_streams, join, hold, prepared)


def _gru_pool_bwd_multi(requests, job_streams, join, hold, prepared):
  lib = _lib.load()
  if not 1 <= len(requests) <= MAX_JOBS:
    raise ValueError('gru_pool_bwd_multi takes 1..%d requests' % MAX_JOBS)
  if not join and hold is None:
    raise ValueError('gru_pool_bwd_multi(join=False) needs a `hold` list')
  jobs = (_lib.GruBwdJob * len(requests))()
  keep, out = [], []
  if prepared is None:
    prepared = prepare_bwd(requests)
  for k, r in enumerate(requests):
    grads, dh0, g, dx_dev, ws, ws_bytes, dout = prepared[k]
    fctx = r['fctx']
    jobs[k].seqs = ctypes.pointer(fctx['batch'])
    jobs[k].weights = ctypes.pointer(fctx['weights'])
    jobs[k].pool_mode = fctx['pool_mode'] | (0 if join or job_streams is None else _lib.NO_JOIN)
    jobs[k].dout = dout.data_ptr()
    jobs[k].fwd_workspace = fctx['ws'].data_ptr()
    jobs[k].grads = ctypes.pointer(g)
    jobs[k].dx_rows = dx_dev.data_ptr() if dx_dev is not None else None
    d_emb = r.get('d_emb_table')
    jobs[k].d_emb_table = d_emb.data_ptr() if d_emb is not None else None
    jobs[k].dh0 = dh0.data_ptr() if dh0 is not None else None
    jobs[k].workspace = ws.data_ptr()
    jobs[k].workspace_bytes = ws_bytes
    if job_streams is not None:
      jobs[k].stream = ctypes.c_void_p(job_streams[k].cuda_stream)
    side = side_stream(job_streams[k] if job_streams is not None else None)
    jobs[k].side_stream = ctypes.c_void_p(side.cuda_stream) if side is not None else None
    keep.append((g, dx_dev, ws, dout))
    out.append((grads, dh0))
  rc = lib.cmhse_gru_pool_bwd_multi(jobs, len(requests), _stream())
  _lib.check(rc, 'cmhse_gru_pool_bwd_multi')
  if not join:
    hold.append(keep)
  return out


def _prepare_bwd(fctx, dout, dx_ptrs=None, d_emb_table=None, want_dh0=False):
  lib = _lib.load()
  sched, b, w = fctx['sched'], fctx['batch'], fctx['weights']
  H, I, device, pool_mode = fctx['H'], fctx['I'], fctx['device'], fctx['pool_mode']
  S = sched.S
  dout = _f32c(dout, 'dout')
  g = _lib.GruGrads()
  grads = dict(w_ih=torch.empty(3 * H, I, dtype=torch.float32, device=device),
               w_hh=torch.empty(3 * H, H, dtype=torch.float32, device=device),
               b_ih=torch.empty(3 * H, dtype=torch.float32, device=device),
               b_hh=torch.empty(3 * H, dtype=torch.float32, device=device))
  if pool_mode == POOL_ATTN:
    grads.update(w_lin=torch.empty(H, H, dtype=torch.float32, device=device),
                 b_lin=torch.empty(H, dtype=torch.float32, device=device),
                 w_att=torch.empty(H, dtype=torch.float32, device=device))
  for k, t in grads.items():
    setattr(g, 'd' + k, t.data_ptr())
  dx_dev = None
  if dx_ptrs is not None:
    dx_dev = upload(np.asarray(dx_ptrs, dtype=np.uint64)[sched.order].view(np.int64).copy(),
                    device)
  dh0 = torch.empty(S, H, dtype=torch.float32, device=device) if want_dh0 else None
  ws_bytes = lib.cmhse_gru_pool_bwd_workspace(S, sched.Tmax, sched.sum_T, I, H, pool_mode)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
  return grads, dh0, g, dx_dev, ws, ws_bytes, dout


def l2norm_rows_bwd(x, g):
  lib = _lib.load()
  x = _f32c(x, 'x')
  g = _f32c(g, 'g')
  dx = torch.empty_like(x)
  rc = lib.cmhse_l2norm_rows_bwd(x.data_ptr(), g.data_ptr(), dx.data_ptr(), x.shape[0],
                                 x.shape[1], _stream())
  _lib.check(rc, 'cmhse_l2norm_rows_bwd')
  return dx


def contrastive_bwd(im, s, scores, margin, max_violation, norm, grad_out):
  lib = _lib.load()
  im = _f32c(im, 'im')
  s = _f32c(s, 's')
  n, D = im.shape
  grad_out = _f32c(grad_out, 'grad_out').reshape(1)
  d_im = torch.empty_like(im)
  d_s = torch.empty_like(s)
  ws_bytes = lib.cmhse_contrastive_bwd_workspace(n)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=im.device)
  rc = lib.cmhse_contrastive_bwd(im.data_ptr(), s.data_ptr(), scores.data_ptr(), n, D,
                                 float(margin), int(bool(max_violation)), int(bool(norm)),
                                 grad_out.data_ptr(), d_im.data_ptr(), d_s.data_ptr(),
                                 ws.data_ptr(), ws_bytes, _stream())
  _lib.check(rc, 'cmhse_contrastive_bwd')
  return d_im, d_s


def euclid_fwd(a, b=None, b_rows=None, norm=True):
  """decoder.loss.EuclideanLoss forward (cmhse_euclid_fwd).  `b_rows`: numpy uint64 addresses of
  the target rows when they are not one contiguous [rows, cols] tensor."""
  lib = _lib.load()
  a = _f32c(a, 'a')
  rows, cols = a.shape
  loss = torch.empty((), dtype=torch.float32, device=a.device)
  scratch = torch.empty(rows, dtype=torch.float32, device=a.device)
  bd = None
  if b_rows is not None:
    bd = upload(np.asarray(b_rows, dtype=np.uint64).view(np.int64).copy(), a.device)
  else:
    b = _f32c(b, 'b')
  rc = lib.cmhse_euclid_fwd(a.data_ptr(), b.data_ptr() if b is not None else None,
                            bd.data_ptr() if bd is not None else None, rows, cols,
                            int(bool(norm)), loss.data_ptr(), scratch.data_ptr(), _stream())
  _lib.check(rc, 'cmhse_euclid_fwd')
  return loss, bd


def euclid_bwd(a, b, bd, norm, grad_out):
  lib = _lib.load()
  a = _f32c(a, 'a')
  rows, cols = a.shape
  d_a = torch.empty_like(a)
  g = _f32c(grad_out, 'grad_out').reshape(1)
  rc = lib.cmhse_euclid_bwd(a.data_ptr(), b.data_ptr() if b is not None else None,
                            bd.data_ptr() if bd is not None else None, rows, cols,
                            int(bool(norm)), g.data_ptr(), d_a.data_ptr(), _stream())
  _lib.check(rc, 'cmhse_euclid_bwd')
  return d_a


def groupwise_fwd(im, s, num_clips, num_caps, margin, max_violation, norm):
  """loss.GroupWiseContrastiveLoss forward (cmhse_groupwise_fwd).  Returns (loss, saved state)."""
  lib = _lib.load()
  im = _f32c(im, 'im')
  s = _f32c(s, 's')
  n, D = im.shape
  B = len(num_clips)
  if len(num_caps) != B or sum(num_clips) != n or sum(num_caps) != s.shape[0] or s.shape[0] != n:
    raise ValueError('GroupWiseContrastiveLoss: block sizes must tile the rows (loss.py:33)')
  off = np.zeros(2 * (B + 1), dtype=np.int32)
  np.cumsum(np.asarray(num_clips), out=off[1:B + 1])
  np.cumsum(np.asarray(num_caps), out=off[B + 2:])
  off_d = upload(off, im.device)
  loss = torch.empty((), dtype=torch.float32, device=im.device)
  reduced = torch.empty(B, B, dtype=torch.float32, device=im.device)
  arg = torch.empty(B, B, dtype=torch.int32, device=im.device)
  ws_bytes = lib.cmhse_groupwise_workspace(n, B)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=im.device)
  rc = lib.cmhse_groupwise_fwd(im.data_ptr(), s.data_ptr(), n, D, off_d.data_ptr(),
                               off_d.data_ptr() + 4 * (B + 1), B, float(margin),
                               int(bool(max_violation)), int(bool(norm)), loss.data_ptr(),
                               reduced.data_ptr(), arg.data_ptr(), None, ws.data_ptr(), ws_bytes,
                               _stream())
  _lib.check(rc, 'cmhse_groupwise_fwd')
  return loss, dict(off=off_d, reduced=reduced, arg=arg, B=B)


def groupwise_bwd(im, s, st, margin, max_violation, norm, grad_out):
  lib = _lib.load()
  n, D = im.shape
  B = st['B']
  g = _f32c(grad_out, 'grad_out').reshape(1)
  d_im = torch.empty_like(im)
  d_s = torch.empty_like(s)
  ws_bytes = lib.cmhse_groupwise_bwd_workspace(n, B)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=im.device)
  rc = lib.cmhse_groupwise_bwd(im.data_ptr(), s.data_ptr(), n, D, st['off'].data_ptr(),
                               st['off'].data_ptr() + 4 * (B + 1), B, float(margin),
                               int(bool(max_violation)), int(bool(norm)),
                               st['reduced'].data_ptr(), st['arg'].data_ptr(), g.data_ptr(),
                               d_im.data_ptr(), d_s.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
  _lib.check(rc, 'cmhse_groupwise_bwd')
  return d_im, d_s
