// grid_sync.hpp — the grid-wide barrier of the kernels that stay RESIDENT over a recurrence's chain
// (gru_fwd_tail_kernel / gru_fwd_chain_kernel, gru.hip; gru_bwd_tail_kernel / gru_bwd_chain_kernel,
// bwd.hip), and what happens when it cannot complete.
//
// Mechanism.  One 32-bit arrival counter per kernel launch (zeroed by the caller): after its
// stores of the step have left the CU (`s_waitcnt 0` by every wave, then the workgroup barrier),
// thread 0 of each workgroup adds 1 at agent scope and spins until the counter reaches the step's
// target.  No cache maintenance: the 8 XCDs' L2s are not coherent with each other and the fences
// the HIP memory model prescribes for that (release = L2 write-back, acquire = L2 invalidate) cost
// 24-74 us per step (tools/microbench/grid_barrier.hip) — as much as the launches the resident
// kernels replace.  Instead every value that crosses workgroups is written with an agent-scope
// (sc1, write-through) store and read with an agent-scope (sc1) load, which the non-coherent
// caches do not serve; everything else a step touches is private to its workgroup or was
// written by an earlier kernel.  The arrive / spin themselves are relaxed agent-scope atomics: the
// ordering they need — "my sc1 stores are performed before my arrival is visible" — is what
// `s_waitcnt vmcnt(0)` in front of the arrive provides on this hardware, and the reader's sc1 loads
// are issued after the spin has seen the target (program order, no speculation across the spin's
// s_waitcnt).  This is a gfx950 contract, not the portable HIP memory model, and it is stated here
// on purpose.
//
// Co-residency.  Every workgroup of the launch must be on the chip at once.  The launchers check
// what can be checked (CU count against the grid, hipOccupancy for the kernel); what cannot —
// another process holding CUs, a CU mask, a queue pre-empted for seconds — ends in a TIMEOUT, not
// a hang and not a trap: the workgroup whose spin exceeds the wall-time bound (s_memrealtime;
// Tunables::resident_timeout_ms, default 5 s — a barrier completes in microseconds) raises the
// launch's abort word and the device's host-visible status word, every other workgroup sees the
// abort word in its own spin, and all of them leave the kernel.  The step's results are then
// garbage; the NEXT call into the library on that device returns CMHSE_ERR_TIMEOUT (and so does
// cmhse_async_status), so a training loop stops with an error instead of a dead context.  On a
// GPU shared with other tenants set the *_tail_min_steps / *_chain_min_steps tunables to 0: every
// step is then its own launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmhse {

struct GridSync {
  unsigned* counter;       // arrivals of this launch (zeroed by the caller)
  unsigned* abort_word;    // device word shared by the launch's workgroups (zeroed by the caller)
  unsigned* status_host;   // pinned host word (device-visible) or NULL: set to 1 on a timeout
  uint64_t timeout_ticks;  // s_memrealtime ticks (100 MHz) one barrier may take
};

// Returns false when the barrier was abandoned (every thread of the workgroup gets the same answer):
// the caller leaves the kernel.  `target` = arrivals expected so far (the caller adds gridDim.x per
// barrier).  All waves must have drained their stores (s_waitcnt) before calling.
__device__ __forceinline__ bool grid_sync_wait(const GridSync& g, unsigned target) {
  __shared__ int s_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(g.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    unsigned spins = 0;
    uint64_t t0 = 0;
    while (__hip_atomic_load(g.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) != 0) continue;
      if (__hip_atomic_load(g.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        ok = 0;
        break;
      }
      const uint64_t now = wall_clock64();
      if (t0 == 0) {
        t0 = now;
      } else if (now - t0 > g.timeout_ticks) {
        __hip_atomic_store(g.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g.status_host != nullptr)
          __hip_atomic_store(g.status_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

// One-sided wait of the step-chain kernel (gru_step_chain_kernel, gru.hip): every thread of the
// workgroup returns once `*flag` has reached `need` (other workgroups add to it with
// flag_signal() after their write-through stores have left the CU), or false when the wait was
// abandoned — the same wall-time bound, abort word and host status word as grid_sync_wait.
__device__ __forceinline__ bool flag_wait(const GridSync& g, const unsigned* flag, unsigned need) {
  __shared__ int s_flag_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    uint64_t t0 = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) != 0) continue;
      if (__hip_atomic_load(g.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        ok = 0;
        break;
      }
      const uint64_t now = wall_clock64();
      if (t0 == 0) {
        t0 = now;
      } else if (now - t0 > g.timeout_ticks) {
        __hip_atomic_store(g.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g.status_host != nullptr)
          __hip_atomic_store(g.status_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
    }
    s_flag_ok = ok;
  }
  __syncthreads();
  return s_flag_ok != 0;
}

// All waves must have drained their stores (s_waitcnt) before calling.
__device__ __forceinline__ void flag_signal(unsigned* flag) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Host side (gru.hip): the device's status word and the GridSync of a launch.
unsigned* resident_status_word();                 // pinned, device-visible; NULL if it cannot be allocated
GridSync make_grid_sync(unsigned* counter, unsigned* abort_word);
int resident_check();                             // CMHSE_OK, or CMHSE_ERR_TIMEOUT once a launch has timed out

}  // namespace cmhse
