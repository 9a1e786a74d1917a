"""RCCL first contact on ONE GPU: the collectives of parallel_eval under
backend 'nccl' (= RCCL) in a 1-rank group, with the shard shapes an 8-rank run can meet — a rank
whose share is EMPTY included (counts with a zero), which pads to one row and must vanish again."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture
def nccl_world1():
  assert torch.cuda.is_available()
  dev = torch.device('cuda', 0)
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
  try:
    yield dev
  finally:
    dist.destroy_process_group()


def test_exchange_step_with_an_empty_and_a_full_shard(nccl_world1):
  from cmhse_amd import parallel_eval
  dev = nccl_world1
  D = 1024
  # a rank with nothing to contribute: [0, D] rows, counts = [0] -> one padded row travels, none returns
  z = torch.zeros(0, D, device=dev)
  A, B = parallel_eval.all_gather_pair(z, z.clone(), [0])
  assert A.shape == (0, D) and B.shape == (0, D) and A.is_cuda
  packed = torch.zeros(0, 4, dtype=torch.int32, device=dev)
  assert parallel_eval.all_gather_rows(packed, [0]).shape[0] in (0, 1)   # (all counts equal the padded height)
  # the real shard size of an 8-rank run (608-640 videos)
  torch.manual_seed(0)
  a, b = torch.randn(615, D, device=dev), torch.randn(615, D, device=dev)
  A, B = parallel_eval.all_gather_pair(a, b, [615])
  assert torch.equal(A, a) and torch.equal(B, b) and A.is_contiguous() and B.is_contiguous()
  r = torch.arange(615 * 4, dtype=torch.int32, device=dev).view(615, 4)
  assert torch.equal(parallel_eval.all_gather_rows(r, [615]), r)


def test_validate_sharded_with_a_stub_only_rank_view(nccl_world1):
  """A 1-rank world whose loader mixes materialised batches with none missing: the documented call
  (no device=, no assignment=) and the step-plan collectives run on RCCL tensors of the current GPU."""
  import argparse
  from cmhse_amd import evaluation, parallel_eval, synthetic
  from cmhse_amd.model import VSE
  dev = nccl_world1
  opt = argparse.Namespace(
      margin=0.2, word_dim=300, embed_size=64, grad_clip=0.0, learning_rate=0.001, max_violation=False,
      img_dim=24, measure='cosine', rnn_type='attention', img_first_size=64, cap_first_size=64,
      low_level_loss=False, weak_low_level_loss=False, reconstruct_loss=False,
      lowest_reconstruct_loss=False, norm=False, data_name='anet_precomp', vocab_size=60)
  torch.manual_seed(2)
  model = VSE(opt)
  spec = synthetic.ragged_spec(21, seed=8)
  batches = synthetic.make_batches(spec, 6, opt.img_dim, opt.vocab_size, seed=4)
  plan = parallel_eval.global_step_plan(batches)
  want = evaluation.split_step_plan(batches)
  for k in evaluation.TOWERS:
    np.testing.assert_array_equal(plan[k], want[k])
  out = parallel_eval.validate_sharded(opt, model, batches)
  res = evaluation.encode_data(opt, model, synthetic.ListLoader(batches), logging=lambda *a: None)
  rep_i, top1_i, ranks_i = evaluation.i2t(res[0], res[1])
  rep_t, top1_t, ranks_t = evaluation.t2i(res[0], res[1])
  assert out[0] == rep_i and out[1] == rep_t
  np.testing.assert_array_equal(out[2], ranks_i)
  np.testing.assert_array_equal(out[3], ranks_t)
  np.testing.assert_array_equal(out[4], top1_i)
  np.testing.assert_array_equal(out[5], top1_t)
