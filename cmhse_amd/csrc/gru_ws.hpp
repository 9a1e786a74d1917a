// gru_ws.hpp — layout of cmhse_gru_pool_fwd's workspace, shared by the forward and backward
// launchers.  Regions (each 256-byte aligned):
//   hs      [sumT, H]     hidden states, time-major packed (always)
//   e_part  [ceil(H/256), sumT]  attention energy partials (CMHSE_POOL_ATTN)
// and, with CMHSE_SAVE_FOR_BACKWARD:
//   gates   [sumT, 4H]    r, z, n, (W_hn h + b_hn) per packed row
//   argmax  [S, H] int32  step of the maximum (CMHSE_POOL_MAX)
//   v       [sumT, H]     tanh(W_lin h + b_lin) (CMHSE_POOL_ATTN)
// with CMHSE_MATH_BF16X3: the hi/lo pre-split copies of W_ih, W_hh (and W_lin), of the input rows
// (xs [sumT, split_ld(I)]) and of the hidden states (hs_s [sumT, split_ld(H)]);
// and last   gx  [rows of the small-batch steps, 3H]   their hoisted input projection x W_ih^T.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <vector>

#include "../../include/cmhse_hip.h"

// Wave priority of the kernels on a recurrence's dependent chain (s_setprio; 0 = the hardware
// default).  A chain step is a short latency-bound kernel that shares its CUs with throughput work
// — the weight-gradient products and projection chunks on the side stream, the other tower's
// chain: raised, its waves win the SIMD's issue arbitration against the resident GEMM waves, which
// fill the bubbles.  Training step, priority 3 against 0 (tools/ab_train.py, two runs each): ICEP
// 9.83 / 9.76 -> 9.58 / 9.61 ms, C3D 8.62 / 8.62 -> 8.49 / 8.58 ms (priorities 1, 2 and 3 measure equal;
// the validation pass does not move: a 615-video share 41.9-42.0 ms at 0, 42.1 at 3).  The LDS-tiled step kernel of the
// validation pass is throughput work itself and stays at the default.
#ifndef CHAIN_PRIO
#define CHAIN_PRIO 3
#endif
#define CHAIN_WAVE_PRIORITY()                                   \
  do {                                                          \
    if (CHAIN_PRIO > 0) __builtin_amdgcn_s_setprio(CHAIN_PRIO); \
  } while (0)

namespace cmhse {

constexpr int kChainMaxStepsWs = 96;   // steps one launch of gru_step_chain_kernel covers at most (gru.hip: kChainMaxSteps)
constexpr int kAttBN = 256;  // columns of W_lin per attention-energy workgroup

struct GruWs {
  size_t hs, e_part, gates, argmax, v, wih_s, whh_s, wlin_s, xs, hs_s, h0_s, gx, tail_sync, chain_sync, total;
};

// Kernel-shape crossovers a caller may move (cmhse_tune / cmhse_ctx_tune; documented one by one in
// include/cmhse_hip.h): values, never results.  Read with relaxed atomics at every call.  ONE table:
// the struct's members, the name lookup of cmhse_tune and the copy a new context starts from are all
// generated from it, so a knob cannot exist in one of them only.
//   X(name, default)
#define CMHSE_TUNABLES(X)                                                                                          \
  X(tiny_max_seqs, 1024)      /* active sequences at or below which a forward step runs on the small-batch kernels */ \
  X(mid_max_seqs, 1024)       /* ... on gru_step_mid_kernel (hoisted input projection + split-K recurrent part); 0 disables it */ \
  X(mid_units, 0)             /* 16 | 8 | 4 forces the mid-size step's unit tile (0 = by grid size) */             \
  X(mid_waves, 0)             /* 4 | 8 forces its waves per workgroup (0 = by schedule) */                         \
  X(tall_tile_min_wgs, 2048)  /* 64-row workgroups from which a tiled per-step launch uses 128-row tiles (0 = never) */ \
  X(bwd_mid_max_seqs, 512)    /* active sequences at or below which a BPTT step runs on gru_bwd_step_mid_kernel */  \
  X(bwd_chunk_rows, 2048)     /* packed rows a weight-gradient chunk spans before it is issued beside the chain */ \
  X(bwd_split_min_seqs, 33)   /* ... from which (up to bwd_mid_max_seqs) the BPTT step is two launches with K split over the grid */ \
  X(bwd_tail_min_steps, 4)    /* <= 32-sequence steps at the end of a chain from which its BPTT runs them in one resident kernel */ \
  X(fwd_tail_min_steps, 4)    /* the same for the forward chain of a training call (gru_fwd_tail_kernel) */        \
  X(mid_tall_min_seqs, 129)   /* active sequences from which the mid-size forward step takes 64 sequences per workgroup */ \
  X(resident_timeout_ms, 5000) /* wall time one wait of a multi-step kernel may take before the launch gives up (grid_sync.hpp) */ \
  X(xproj_chunk_rows, 1536)   /* packed rows per chunk of a training chain's hoisted input projection beside the chain */ \
  X(tn_rows_bm, 0)            /* tile height of the weight-gradient products: 128, 192, or 0 = by shape (tn_rows.hpp) */ \
  X(chain_min_steps, 2)       /* consecutive LDS-tiled inference steps from which they are ONE launch of gru_step_chain_kernel */ \
  X(early_xproj, 1)           /* 1: an inference call's hoisted projection starts on the side stream before the first step */ \
  X(chain_tall_min_wgs, 256)  /* 64-row workgroups per step from which a step chain uses 128-row tiles */ \
  X(pull_waves, 32)           /* single-wave workgroups of one cmhse_pull_steps launch (host -> HBM over PCIe) */

struct Tunables {
#define CMHSE_TUNABLE_MEMBER_(name, dflt) std::atomic<int> name{dflt};
  CMHSE_TUNABLES(CMHSE_TUNABLE_MEMBER_)
#undef CMHSE_TUNABLE_MEMBER_
};
Tunables& tunables();
// A multi-step knob (chain_min_steps, *_tail_min_steps) as the current device sees it: 0 once a timeout
// of a multi-step kernel has been acknowledged there (cmhse_async_status(1)), else the knob's value.
int multi_step_knob(const std::atomic<int>& knob);
static inline int mid_max_seqs() { return tunables().mid_max_seqs.load(std::memory_order_relaxed); }

// Upper bound of the packed rows whose input projection is hoisted (the rows of the steps with at
// most mid_max_seqs() active sequences; all rows when the whole batch is small enough).
static inline int64_t gx_rows_bound(int32_t S, int32_t Tmax, int64_t sum_T) {
  const int64_t mm = mid_max_seqs();
  if (mm <= 0) return 0;
  if (S <= mm || Tmax <= 0) return sum_T;
  const int64_t b = mm * static_cast<int64_t>(Tmax);
  return b < sum_T ? b : sum_T;
}

constexpr int32_t kModeMask = ~(CMHSE_SAVE_FOR_BACKWARD | CMHSE_MATH_BF16X3 | CMHSE_NO_JOIN);

// row length (in float units) of a bf16x3 pre-split weight row: K rounded up to whole 16-k chunks
__host__ __device__ static inline int64_t split_ld(int K) { return (static_cast<int64_t>(K) + 15) / 16 * 16; }

static inline size_t ws_align(size_t v) { return (v + 255) / 256 * 256; }

static inline GruWs gru_ws_layout(int32_t S, int64_t sum_T, int32_t H, int32_t mode_flags,
                                  int32_t I = 0, int32_t Tmax = 0) {
  const bool save = (mode_flags & CMHSE_SAVE_FOR_BACKWARD) != 0;
  const bool bf3 = (mode_flags & CMHSE_MATH_BF16X3) != 0;
  const int mode = mode_flags & kModeMask;
  GruWs L;
  size_t off = 0;
  L.hs = off;
  off += ws_align(static_cast<size_t>(sum_T) * H * sizeof(float));
  L.e_part = off;
  if (mode == CMHSE_POOL_ATTN)
    off += ws_align(static_cast<size_t>((H + kAttBN - 1) / kAttBN) * sum_T * sizeof(float));
  L.gates = off;
  if (save) off += ws_align(static_cast<size_t>(sum_T) * 4 * H * sizeof(float));
  L.argmax = off;
  if (save && mode == CMHSE_POOL_MAX) off += ws_align(static_cast<size_t>(S) * H * sizeof(int32_t));
  L.v = off;
  if (save && mode == CMHSE_POOL_ATTN) off += ws_align(static_cast<size_t>(sum_T) * H * sizeof(float));
  L.wih_s = off;
  if (bf3) off += ws_align(static_cast<size_t>(3) * H * split_ld(I) * sizeof(float));
  L.whh_s = off;
  if (bf3) off += ws_align(static_cast<size_t>(3) * H * split_ld(H) * sizeof(float));
  L.wlin_s = off;
  if (bf3 && mode == CMHSE_POOL_ATTN) off += ws_align(static_cast<size_t>(H) * split_ld(H) * sizeof(float));
  // bf16x3: the inputs and the hidden states of the LDS-tiled steps in pre-split (hi | lo) rows
  L.xs = off;
  if (bf3) off += ws_align(static_cast<size_t>(sum_T) * split_ld(I) * sizeof(float));
  L.hs_s = off;
  if (bf3) off += ws_align(static_cast<size_t>(sum_T) * split_ld(H) * sizeof(float));
  L.h0_s = off;  // ... and of the caller's initial hidden states
  if (bf3) off += ws_align(static_cast<size_t>(S) * split_ld(H) * sizeof(float));
  L.gx = off;   // last region: the backward pass never looks at it (it calls this without Tmax)
  if (I % 4 == 0 && H % 4 == 0)
    off += ws_align(static_cast<size_t>(gx_rows_bound(S, Tmax, sum_T)) * 3 * H * sizeof(float));
  L.tail_sync = off;   // barrier counter of gru_fwd_tail_kernel
  off += 256;
  // step-chain kernel (inference calls): 8 task tickets + the abort word, then one counter per
  // (step of a chain launch, 64-row tile)
  L.chain_sync = off;
  if (!save && Tmax > 0)
    off += ws_align(256 + sizeof(unsigned) * static_cast<size_t>(Tmax < kChainMaxStepsWs ? Tmax : kChainMaxStepsWs) *
                              static_cast<size_t>((S + 63) / 64));
  L.total = off;
  return L;
}

// HIP events are recycled through per-device free lists (gru.hip; the library's one piece of
// process-wide state, mutex-guarded): creating an event can stall the host for tens of
// milliseconds when the runtime grows its pools while the GPU is busy (measured: 65 ms inside a
// validation pass), and the step launcher needs one per stream hand-over, a timer two per tiled
// launch.  event_get() pops a recycled event of the wanted kind for the CURRENT device or creates
// one; event_put() returns it to that device's list (an event may be re-recorded as soon as
// nothing waits for its old use: the launcher's wait-events are consumed by hipStreamWaitEvent at
// enqueue time, the timers' events are read before their handle is destroyed).
// A resident kernel (gru_fwd_tail_kernel / gru_bwd_tail_kernel) needs all its `wgs` workgroups on the
// chip at once, one per CU, and the package may run four such chains side by side: true when the
// current device has the CUs for that (a full MI355X: 256; a partitioned one may not).
bool resident_fits(int wgs);
// ... and for the chain kernels, whose workgroups share their CU with others: at least `wgs` CUs
bool resident_fits_wgs(int wgs);

hipEvent_t event_get(bool timing);
void event_put(hipEvent_t ev, bool timing);

// Orders `waiter` behind everything queued on `signal` so far (one pooled event; falls back to a
// host-side hipStreamSynchronize(signal) if the event cannot be recorded or waited for).
void stream_after(hipStream_t waiter, hipStream_t signal);

// cmhse_timer handle (measurement aid, include/cmhse_hip.h): HIP events owned by the handle.
struct Timer {
  hipEvent_t start, stop;
  int32_t launches;  // step kernels launched inside the bracket
  // one event pair around every launch of the LDS-tiled step kernel (the dominant kernel), with the
  // algorithmic FLOPs of those launches, so bench.py can price that kernel alone
  std::vector<hipEvent_t> tiled_events;
  double tiled_flops, tiled_bytes;
};

}  // namespace cmhse
