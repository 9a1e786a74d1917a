"""Builds tools/microbench/canary.so, the bystander canary of the lost-update tests
(tools/pkfma_canary.py, tests/test_gpu_chain.py).  No torch import: __graft_entry__.build() calls
this in the build step.  Built WITHOUT the library's device flags on purpose: the
victim loop must keep its v_pk_fma_f32."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANARY_SRC = os.path.join(R, 'tools', 'microbench', 'canary.hip')
CANARY_LIB = os.path.join(R, 'tools', 'microbench', 'canary.so')


def _hipcc():
  import importlib.util
  spec = importlib.util.spec_from_file_location('_cmhse_build', os.path.join(R, 'cmhse_amd', 'build.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)          # (the file alone: importing the package would pull torch in)
  return mod._hipcc()


def build_canary():
  """Rebuilt when missing or older than its source; returns the path."""
  if os.path.exists(CANARY_LIB) and os.path.getmtime(CANARY_LIB) >= os.path.getmtime(CANARY_SRC):
    return CANARY_LIB
  res = subprocess.run([_hipcc(), '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', CANARY_LIB + '.tmp',
                        CANARY_SRC], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
  if res.returncode != 0:
    raise RuntimeError('hipcc failed:\n' + res.stdout)
  os.replace(CANARY_LIB + '.tmp', CANARY_LIB)
  return CANARY_LIB


if __name__ == '__main__':
  print(build_canary())
