// launch_gap.hip — GPU-side gap between DEPENDENT kernels of one stream: plain launches against the
// same chain replayed from a captured hipGraph (the CDNA4 playbook's "capture launch-bound inner
// loops in hipGraphs").  Each kernel: `wgs` workgroups that spin ~`us` microseconds (s_memrealtime)
// and write one word, so a chain's wall time is n x (kernel + gap).
//   hipcc --offload-arch=gfx950 -O3 -o launch_gap.bin launch_gap.hip && ./launch_gap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(unsigned* out, unsigned ticks, const unsigned* in) {
  const unsigned long long t0 = wall_clock64();
  unsigned v = in ? in[blockIdx.x] : 0u;          // a dependence on the previous kernel's output
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) out[blockIdx.x] = v + 1u;
}

static float run_stream(hipStream_t st, unsigned* a, unsigned* b, int n, int wgs, unsigned ticks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < n; ++i)
    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st, (i & 1) ? a : b, ticks, (i & 1) ? b : a);
  CHECK(hipEventRecord(e1, st));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  const int n = 400;
  unsigned *a, *b;
  CHECK(hipMalloc(&a, 4096 * 4)); CHECK(hipMalloc(&b, 4096 * 4));
  CHECK(hipMemset(a, 0, 4096 * 4)); CHECK(hipMemset(b, 0, 4096 * 4));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  const int wg_list[] = {64, 256, 1024};
  const unsigned us_list[] = {2, 10, 20};
  printf("%d dependent launches of one stream; per launch: us total (kernel spin + gap)\n", n);
  printf("  workgroups  spin us   stream launches   graph replay\n");
  for (int wi = 0; wi < 3; ++wi)
    for (int ui = 0; ui < 3; ++ui) {
      const int wgs = wg_list[wi];
      const unsigned ticks = us_list[ui] * 100u;
      run_stream(st, a, b, n, wgs, ticks);                       // warm-up
      const float ms_s = run_stream(st, a, b, n, wgs, ticks);
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, st, (i & 1) ? a : b, ticks, (i & 1) ? b : a);
      CHECK(hipStreamEndCapture(st, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipGraphLaunch(ge, st));                             // warm-up
      CHECK(hipStreamSynchronize(st));
      CHECK(hipEventRecord(e0, st));
      CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      float ms_g = 0.f;
      CHECK(hipEventElapsedTime(&ms_g, e0, e1));
      printf("  %9d  %7u   %15.2f   %12.2f\n", wgs, us_list[ui], ms_s * 1e3f / n, ms_g * 1e3f / n);
      CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
  return 0;
}
