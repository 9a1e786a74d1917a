// What does one step of a kernel that stays resident over a recurrence's chain cost on MI355X?
// A step of such a kernel = every workgroup publishes a few rows (h_t), a grid-wide barrier, every
// workgroup reads ALL rows of the step (the A operand of step t + 1) — across the 8 XCDs, whose L2s
// are not coherent with each other.  Variants of the publish / barrier / read triple:
//   fence    plain stores and loads, __threadfence() by every wave before the arrive and after the wait
//            (what the HIP memory model asks for: L2 write-back + invalidate per fence)
//   scoped   agent-scope relaxed atomic stores / loads for the exchanged data (write-through, L2-bypass
//            on read), s_waitcnt before the arrive, no cache maintenance — valid because every row is
//            written once, to a fresh address, and read only after the barrier
// Every spin is bounded (a lost arrive ends the kernel with an error flag, never a hang), and every
// value read is checked.  Compare with the 7.6 us (forward) / 12.8 us (backward) that one dependent
// LAUNCH of the small-batch step kernels costs (profiles/r03_step_latency.json).
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier.bin grid_barrier.hip && ./grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned kMaxSpins = 1u << 22;

template <bool FENCE>
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
  if (FENCE) __threadfence(); else __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kMaxSpins) { good = 0; *err = 1; break; }
    }
    ok = good;
  }
  __syncthreads();
  if (FENCE) __threadfence();
  return ok != 0;
}

// rows: [steps][n_wg * rows_per_wg][cols]; workgroup w publishes rows_per_wg rows per step and, after
// the barrier, reads `read_rows` rows of the step (all of them by default)
template <bool FENCE>
__global__ __launch_bounds__(512) void chain(float* rows, int steps, int rows_per_wg, int cols,
                                             int read_rows, unsigned* counter, unsigned* err,
                                             unsigned long long* bad) {
  const int n_rows = gridDim.x * rows_per_wg;
  unsigned long long wrong = 0;
  for (int t = 0; t < steps; ++t) {
    float* step = rows + static_cast<size_t>(t) * n_rows * cols;
    for (int i = threadIdx.x; i < rows_per_wg * cols; i += blockDim.x) {
      const int r = blockIdx.x * rows_per_wg + i / cols, c = i % cols;
      const float v = static_cast<float>((t * 131 + r * 7 + c) & 0xffff);
      if (FENCE) step[static_cast<size_t>(r) * cols + c] = v;
      else __hip_atomic_store(&step[static_cast<size_t>(r) * cols + c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_barrier<FENCE>(counter, static_cast<unsigned>(t + 1) * gridDim.x, err)) return;
    for (int i = threadIdx.x; i < read_rows * cols; i += blockDim.x) {
      const int r = (blockIdx.x * rows_per_wg + i / cols) % n_rows, c = i % cols;
      const float* p = &step[static_cast<size_t>(r) * cols + c];
      const float v = FENCE ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wrong += (v != static_cast<float>((t * 131 + r * 7 + c) & 0xffff));
    }
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <bool FENCE>
static void run(const char* name, int n_wg, int steps, int rows_per_wg, int cols, int read_rows) {
  const size_t n = static_cast<size_t>(steps) * n_wg * rows_per_wg * cols;
  float* rows; unsigned* counter; unsigned* err; unsigned long long* bad;
  CK(hipMalloc(&rows, n * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  unsigned herr = 0; unsigned long long hbad = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(rows, 0xff, n * 4)); CK(hipMemset(counter, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain<FENCE>, dim3(n_wg), dim3(512), 0, 0, rows, steps, rows_per_wg, cols, read_rows, counter, err, bad);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    unsigned e; unsigned long long b;
    CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
    herr |= e; hbad += b;
  }
  printf("%-7s %4d workgroups, %2d rows x %4d floats each, %4d rows read per workgroup: %6.2f us per step%s%s\n",
         name, n_wg, rows_per_wg, cols, read_rows, best * 1e3f / steps, herr ? "  BARRIER TIMED OUT" : "",
         hbad ? "  STALE VALUES READ" : "");
  CK(hipFree(rows)); CK(hipFree(counter)); CK(hipFree(err)); CK(hipFree(bad));
}

int main() {
  const int steps = 200;
  for (int n_wg : {64, 128, 256}) {
    // 16 sequences x H = 1024: workgroup w owns 1024 / n_wg units of every sequence's h_t -> model as
    // rows_per_wg = 16 rows of (1024 / n_wg) floats; everybody reads all 16 x 1024
    const int cols = 1024 / n_wg * 16;   // 16 rows x units, flattened
    run<true>("fence", n_wg, steps, 1, cols, n_wg);
    run<false>("scoped", n_wg, steps, 1, cols, n_wg);
  }
  // barrier alone (nothing exchanged)
  run<true>("fence", 256, steps, 1, 4, 0);
  run<false>("scoped", 256, steps, 1, 4, 0);
  run<false>("scoped", 64, steps, 1, 4, 0);
  return 0;
}
