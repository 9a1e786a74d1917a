// gru_chain.hpp
//
// The step CHAIN: the LDS-tiled steps of a call as ONE launch with per-row-tile dependencies
// (gru_step_chain_kernel).  Included by gru.hip only, after gru_step_tile.hpp.
#pragma once

namespace cmhse {

// ---------------------------------------------------------------------------------------------
// Step CHAIN: the LDS-tiled steps t0 .. t0 + nsteps - 1 of up to kMaxJobs encoders in ONE launch.
//
// Per-step launches drain the chip at every time step: the last round of a step's workgroups runs
// on a partly empty chip (a full split: ~2 % of the kernel's time; a rank's 615-video share, whose
// steps are one or two rounds each: 15 %), although row tile r of step t + 1 needs nothing but row
// tile r of step t — the sequences are sorted by length, so the active set of a step is a prefix
// of the previous one's — and two thirds of its work (the x phase, K = I) nothing at all.  Here
// every (step, request, row tile, column tile) is a TASK; a workgroup takes the next task of its
// XCD's queue (tasks in step order; column tile c belongs to queue c % 8, so an XCD's L2 keeps
// re-serving the same weight rows exactly as with the per-step launches' block order), runs the
// tile's x phase, waits until the counter of (request, step - 1, row tile) has reached the number
// of column tiles, runs the h phase and the epilogue, writes the new state rows through to memory
// (agent-scope stores: the next step's tiles run on other XCDs, whose L2s are not coherent with
// this one; nobody has read those addresses — whole cache lines: H % 32 == 0 is a condition of the
// chain — before they were written, so the readers' plain loads miss their L2 and are served
// from memory) and bumps its own counter.  Results are
// bit-identical to the per-step launches (same tiles, same k order).
//
// Progress: a workgroup takes its task when it starts (queue = its index modulo 8), workgroups
// start in index order, every queue lists its tasks in step order, and a task depends only on
// tasks of the previous step.  So the queues advance in step with each other, and the earliest
// unfinished task overall is either running (everything it waits for is earlier, hence done) or
// the next one its queue hands out, with every task that is already held at most a step ahead of
// it — some held task can always run.  No co-residency requirement (the grid is one workgroup per
// task, dispatched as slots free up); a workgroup whose queue is exhausted takes a task of another
// queue.  That argument needs EQUAL queues: it holds when the column tiles are a whole multiple of
// the 8 XCDs (H = 512, 1024, 1536 ...); for every other count there is one queue for the whole
// chip (chain_queues), whose tickets are a topological order of the tasks.  The wait is bounded
// like the resident kernels' barrier (grid_sync.hpp): CMHSE_ERR_TIMEOUT, not a hang.
// ---------------------------------------------------------------------------------------------
constexpr int kChainMaxSteps = kChainMaxStepsWs;
constexpr int kXcds = 8;
// Tasks of a queue come in PHASES, one per time step, the same number of tickets in every queue:
// phase s (s < nsteps) = the GRU tiles of step t0 + s: (request, row tile) x the queue's column tiles.
// tick[p] = tickets of a queue in front of phase p.
constexpr int kChainPhases = kChainMaxSteps;
struct GruChainGroup {
  GruStepParams j[kMaxJobs];            // (t, S_t, off_prev, off_cur unused: derived per task)
  const int32_t* step_off[kMaxJobs];    // device: first packed row of every step of request k
  unsigned* done[kMaxJobs];             // zeroed counters [nsteps][rt_stride[k]] of request k
  int32_t rt_stride[kMaxJobs];          // row tiles of request k at step t0 (its maximum)
  uint32_t tick[kChainPhases + 1];
  unsigned* ticket;                     // [kXcds] zeroed: next task of every queue
  GridSync sync;
  int32_t n, t0, nsteps, n_tiles;
};

// Queues.  n_tiles % 8 == 0: eight, column tile c of the GRU step in queue c % 8 (an XCD's L2 keeps
// re-serving the same weight rows, as with the per-step launches' block order), every queue the same
// number of tickets.  Any other count (H = 128, 192, 256, 320, 768, 1280 ...): ONE queue holds all
// the tasks in (step, row tile, column tile) order — with uneven queues the workgroups of the XCDs
// with fewer (or no) columns overflow into the others, those queues run steps ahead of the short
// ones and can fill every resident slot with workgroups waiting for tasks nobody is left to start
// (a discrete-event model of the ticket logic deadlocks at n_tiles = 2, 4, 12, 20).
// With one ticket every held task depends on earlier tickets only, so the earliest unfinished one
// can always run.
__device__ __host__ __forceinline__ int chain_queues(int n_tiles) { return (n_tiles % kXcds == 0) ? kXcds : 1; }

template <bool VEC, int MSUB>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(MSUB == 1 ? 3 : 2)))
void gru_step_chain_kernel(const GruChainGroup g) {
  constexpr int BM = 64 * MSUB;
  __shared__ unsigned s_task[2];
  const unsigned nq = static_cast<unsigned>(chain_queues(g.n_tiles));
  const unsigned cols = static_cast<unsigned>(g.n_tiles) / nq;
  const int n_phases = g.nsteps;
  const unsigned per_queue = g.tick[n_phases];
  if (threadIdx.x == 0) {
    // home queue: workgroups are dealt to the XCDs round-robin by their index (b and b + 8 share an
    // XCD — what the per-step kernels' block order relies on too), so this IS the workgroup's XCD on
    // an unpartitioned MI355X; derived from the index rather than read from XCC_ID so that the
    // queues advance in step with the dispatch order whatever the partition mode
    const unsigned x = blockIdx.x & (nq - 1);
    unsigned got = 0xffffffffu, queue = 0xffffffffu;
    for (unsigned d = 0; d < nq; ++d) {
      const unsigned y = (x + d) & (nq - 1);
      const unsigned tk = __hip_atomic_fetch_add(g.ticket + y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tk < per_queue) {
        got = tk;
        queue = y;
        break;
      }
    }
    s_task[0] = got;
    s_task[1] = queue;
  }
  __syncthreads();
  const unsigned queue = __builtin_amdgcn_readfirstlane(s_task[1]);
  if (queue == 0xffffffffu) return;      // every queue is empty (cannot happen: one workgroup per ticket)
  const unsigned tk = __builtin_amdgcn_readfirstlane(s_task[0]);
  int lo = 0, hi = n_phases - 1;         // the last phase whose first ticket is <= tk
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (g.tick[mid] <= tk) lo = mid; else hi = mid - 1;
  }
  // ---- the GRU tile of step s this ticket stands for ----
  const int s = lo, t = g.t0 + s;
  const unsigned local = tk - g.tick[s];
  unsigned rem = local / cols;
  const int c = static_cast<int>(queue + nq * (local % cols));
  int k = 0, S_t = 0;
  for (; k < g.n; ++k) {
    S_t = g.step_off[k][t + 1] - g.step_off[k][t];
    const unsigned rt = static_cast<unsigned>((S_t + BM - 1) / BM);
    if (rem < rt || k == g.n - 1) break;
    rem -= rt;
  }
  const GruStepParams& p = g.j[k];
  const int64_t off_cur = g.step_off[k][t];
  const int64_t off_prev = (t > 0) ? g.step_off[k][t - 1] : 0;
  ChainDep dep;
  dep.sync = g.sync;
  dep.need = static_cast<unsigned>(g.n_tiles);
  dep.done = g.done[k] + static_cast<size_t>(s) * g.rt_stride[k] + rem;
  dep.wait = (s > 0) ? g.done[k] + static_cast<size_t>(s - 1) * g.rt_stride[k] + rem : nullptr;
  gru_step_tile<VEC, MSUB, false, true>(p, rem * static_cast<unsigned>(g.n_tiles) + static_cast<unsigned>(c), t, S_t,
                                        off_prev, off_cur, dep);
}

}  // namespace cmhse
