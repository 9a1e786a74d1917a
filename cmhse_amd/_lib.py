"""ctypes binding of libcmhse_hip.so (include/cmhse_hip.h).

This is the binding a maintainer of the reference would add.  There is no CPU
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CMHSE_HIP_LIB: another build of the same sources (kernel experiments); same ABI, still HIP-only
LIB_PATH = os.environ.get('CMHSE_HIP_LIB') or os.path.join(_HERE, 'libcmhse_hip.so')

POOL_LAST, POOL_ATTN, POOL_MAX, POOL_ALL = 0, 1, 2, 3
SAVE_FOR_BACKWARD = 0x100
MATH_BF16X3 = 0x200
NO_JOIN = 0x400
MAX_JOBS = 4            # requests per cmhse_gru_pool_fwd_multi call
POOL_OF = {'seq2seq': POOL_LAST, 'attention': POOL_ATTN, 'maxout': POOL_MAX}

c_void_p, c_int32, c_int64, c_size_t, c_float = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                  ctypes.c_size_t, ctypes.c_float)


class GruWeights(ctypes.Structure):
  _fields_ = [('w_ih', c_void_p), ('w_hh', c_void_p), ('b_ih', c_void_p), ('b_hh', c_void_p),
              ('w_lin', c_void_p), ('b_lin', c_void_p), ('w_att', c_void_p)]


class GruGrads(ctypes.Structure):
  _fields_ = [('dw_ih', c_void_p), ('dw_hh', c_void_p), ('db_ih', c_void_p), ('db_hh', c_void_p),
              ('dw_lin', c_void_p), ('db_lin', c_void_p), ('dw_att', c_void_p)]


class SeqBatch(ctypes.Structure):
  _fields_ = [('S', c_int32), ('Tmax', c_int32), ('I', c_int32), ('H', c_int32),
              ('x_rows', c_void_p), ('x_step_floats', c_int32), ('tok_rows', c_void_p),
              ('emb_table', c_void_p),
              ('vocab', c_int32), ('h0_rows', c_void_p), ('lens', c_void_p),
              ('out_row', c_void_p), ('step_off', c_void_p), ('step_count_host', c_void_p),
              ('step_timer', c_void_p), ('step_events_host', c_void_p),
              ('step_plan_host', c_void_p)]


class GruJob(ctypes.Structure):
  _fields_ = [('seqs', ctypes.POINTER(SeqBatch)), ('weights', ctypes.POINTER(GruWeights)),
              ('pool_mode', c_int32), ('out', c_void_p), ('workspace', c_void_p),
              ('workspace_bytes', c_size_t), ('tail_stream', c_void_p), ('stream', c_void_p),
              ('side_stream', c_void_p), ('out_ready_event', c_void_p)]


class GruBwdJob(ctypes.Structure):
  _fields_ = [('seqs', ctypes.POINTER(SeqBatch)), ('weights', ctypes.POINTER(GruWeights)),
              ('pool_mode', c_int32), ('dout', c_void_p), ('fwd_workspace', c_void_p),
              ('grads', ctypes.POINTER(GruGrads)), ('dx_rows', c_void_p), ('d_emb_table', c_void_p),
              ('dh0', c_void_p), ('workspace', c_void_p), ('workspace_bytes', c_size_t),
              ('stream', c_void_p), ('side_stream', c_void_p)]


STEP_LOSS_MAX = 8       # CMHSE_STEP_LOSS_MAX


class StepLosses(ctypes.Structure):
  _fields_ = [('n_emb', c_int32), ('n_terms', c_int32), ('D', c_int32),
              ('x', c_void_p * STEP_LOSS_MAX), ('rows', c_int32 * STEP_LOSS_MAX),
              ('term_a', c_int32 * STEP_LOSS_MAX), ('term_b', c_int32 * STEP_LOSS_MAX),
              ('weight', c_float * STEP_LOSS_MAX), ('margin', c_float),
              ('max_violation', c_int32), ('norm', c_int32)]


# every symbol include/cmhse_hip.h declares: (restype, argtypes)
SIGNATURES = {
    'cmhse_gru_pool_workspace': (c_size_t, [c_int32, c_int32, c_int64, c_int32, c_int32, c_int32]),
    'cmhse_async_status': (ctypes.c_int, [c_int32]),
    'cmhse_selftest_grid_sync': (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    'cmhse_gru_pool_ws_region': (ctypes.c_int, [c_int32, c_int32, c_int64, c_int32, c_int32, c_int32,
                                                ctypes.c_char_p, ctypes.POINTER(c_size_t),
                                                ctypes.POINTER(c_size_t)]),
    'cmhse_gru_pool_fwd': (ctypes.c_int, [ctypes.POINTER(SeqBatch), ctypes.POINTER(GruWeights),
                                          c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cmhse_gru_pool_fwd_multi': (ctypes.c_int, [ctypes.POINTER(GruJob), c_int32, c_void_p]),
    'cmhse_pull_steps': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                        c_int32, c_void_p]),
    'cmhse_push_rows': (ctypes.c_int, [c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_void_p]),
    'cmhse_rows_differ': (ctypes.c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'cmhse_pad_rows': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                      c_void_p, c_void_p]),
    'cmhse_l2norm_rows': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p]),
    'cmhse_gather_rows': (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p,
                                         c_void_p]),
    'cmhse_sim_rank_workspace': (c_size_t, [c_int32]),
    'cmhse_sim_rank': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                      c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cmhse_sim_rank_ex': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                         c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                         c_void_p]),
    'cmhse_cosine_sim': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                        c_void_p]),
    'cmhse_contrastive_workspace': (c_size_t, [c_int32]),
    'cmhse_contrastive_fwd': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_float,
                                             c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                             c_size_t, c_void_p]),
    'cmhse_contrastive_blocks_workspace': (c_size_t, [c_int32, c_int32]),
    'cmhse_contrastive_blocks_fwd': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                    c_int32, c_float, c_int32, c_int32, c_void_p,
                                                    c_void_p, c_size_t, c_void_p]),
    'cmhse_step_losses_workspace': (c_size_t, [ctypes.POINTER(StepLosses)]),
    'cmhse_step_losses_fwd': (ctypes.c_int, [ctypes.POINTER(StepLosses), c_void_p, c_void_p,
                                             c_void_p, c_size_t, c_void_p]),
    'cmhse_step_losses_bwd': (ctypes.c_int, [ctypes.POINTER(StepLosses), c_void_p,
                                             ctypes.POINTER(c_void_p), c_void_p, c_size_t,
                                             c_void_p]),
    'cmhse_contrastive_blocks_bwd_workspace': (c_size_t, [c_int32, c_int32]),
    'cmhse_contrastive_blocks_bwd': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                                    c_int32, c_int32, c_float, c_int32, c_int32,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                    c_void_p]),
    'cmhse_gru_pool_bwd_workspace': (c_size_t, [c_int32, c_int32, c_int64, c_int32, c_int32,
                                                c_int32]),
    'cmhse_gru_pool_bwd': (ctypes.c_int, [ctypes.POINTER(SeqBatch), ctypes.POINTER(GruWeights),
                                          c_int32, c_void_p, c_void_p, ctypes.POINTER(GruGrads),
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                          c_void_p]),
    'cmhse_gru_pool_bwd_multi': (ctypes.c_int, [ctypes.POINTER(GruBwdJob), c_int32, c_void_p]),
    'cmhse_l2norm_rows_bwd': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                             c_void_p]),
    'cmhse_contrastive_bwd_workspace': (c_size_t, [c_int32]),
    'cmhse_contrastive_bwd': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                             c_float, c_int32, c_int32, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_size_t, c_void_p]),
    'cmhse_groupwise_workspace': (c_size_t, [c_int32, c_int32]),
    'cmhse_groupwise_fwd': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                           c_int32, c_float, c_int32, c_int32, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cmhse_groupwise_bwd_workspace': (c_size_t, [c_int32, c_int32]),
    'cmhse_groupwise_bwd': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                           c_int32, c_float, c_int32, c_int32, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    'cmhse_euclid_fwd': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                        c_void_p, c_void_p, c_void_p]),
    'cmhse_euclid_bwd': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                        c_void_p, c_void_p, c_void_p]),
    'cmhse_timer_create': (c_void_p, []),
    'cmhse_timer_destroy': (None, [c_void_p]),
    'cmhse_timer_elapsed_ms': (ctypes.c_int, [c_void_p, ctypes.POINTER(c_float)]),
    'cmhse_timer_launches': (c_int32, [c_void_p]),
    'cmhse_timer_tiled': (ctypes.c_int, [c_void_p, ctypes.POINTER(c_float),
                                         ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(c_int32)]),
    'cmhse_tune': (ctypes.c_int, [ctypes.c_char_p, c_int32, ctypes.POINTER(c_int32)]),
    'cmhse_ctx_create': (c_void_p, []),
    'cmhse_ctx_destroy': (None, [c_void_p]),
    'cmhse_ctx_tune': (ctypes.c_int, [c_void_p, ctypes.c_char_p, c_int32, ctypes.POINTER(c_int32)]),
    'cmhse_ctx_enter': (c_void_p, [c_void_p]),
    'cmhse_strerror': (ctypes.c_char_p, [ctypes.c_int]),
    'cmhse_version': (ctypes.c_char_p, []),
}

_lib = None


def load():
  """Load the HIP library (after torch, so both share one HIP runtime).  Fails loudly."""
  global _lib
  if _lib is not None:
    return _lib
  import torch  # noqa: F401  torch's bundled libamdhip64.so.7 must be the runtime we bind to
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        'cmhse_amd: %s is missing. Build it with `python -m cmhse_amd.build` (needs hipcc); '
        'there is no CPU fallback for the hot path.' % LIB_PATH)
  lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)   # AttributeError if the library does not export the symbol
    fn.restype = res
    fn.argtypes = args
  _lib = lib
  return lib


def check(rc, what):
  if rc != 0:
    msg = load().cmhse_strerror(rc).decode()
    raise RuntimeError('cmhse_hip: %s failed: %s (%d)' % (what, msg, rc))
