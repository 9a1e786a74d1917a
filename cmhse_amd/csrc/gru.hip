// gru.hip — hierarchical-encoder hot path: packed GRU steps + last / attention / max pooling.
//
// Replaces the bodies of layers.Seq2Seq / Attention / Maxout .forward of the reference
// (/root/reference/layers.py:47-66, 93-119, 185-204), i.e. torch.nn.GRU over a
// pack_padded_sequence batch followed by the pooling, with hand-written gfx950 kernels.
//
// Data layout in HBM
//   * inputs are consumed in place through one base pointer per sequence (cmhse_seq_batch);
//   * hidden states live in ONE time-major packed buffer hs[sumT][H] — exactly the order of
//     pack_padded_sequence: step t occupies rows step_off[t] .. step_off[t]+S_t-1, where the
//     S_t still-active sequences are a prefix of the length-sorted batch.  h_{t-1} of the active
//     prefix is therefore a contiguous row block: the A operand of step t is read coalesced, and
//     the same buffer feeds the attention pooling (and a later BPTT) without any copy;
//   * weights stay in the reference's checkpoint layout ([3H,I], [3H,H], gate rows r,z,n).
//
// Kernels
//   gru_step_kernel   one launch per time step: fused [x_t | h_{t-1}] x [W_ih | W_hh]^T exact-fp32
//                     MFMA GEMM (nt_core.hpp) over the active prefix; the r/z pre-activations
//                     accumulate over both K phases, the two n-gate terms are kept apart; gate
//                     math, the state update, the hs store and the last/max pooling are the
//                     epilogue (no gate pre-activation ever goes to HBM).  Bound: fp32 MFMA.
//   attn_energy_kernel  e = w_att . tanh(W_lin h + b_lin) for all packed rows at once
//                     ([sumT,H] x [H,H]^T MFMA GEMM, tanh-dot epilogue, wave-shuffle row sums).
//   attn_pool_kernel  masked exp-softmax (no max-subtraction, +1e-4: layers.py:158-162) and the
//                     weighted sum over time; HBM-bound, one pass over hs.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/cmhse_hip.h"
#include "gru_ws.hpp"
#include "grid_sync.hpp"
#include "nt_core.hpp"

#include "gru_step_tile.hpp"
#include "gru_attention.hpp"
#include "gru_chain.hpp"
#include "gru_small_batch.hpp"
#include "gru_rows.hpp"

namespace cmhse {

// small-batch / tiled crossover of the forward steps (Tunables::tiny_max_seqs)
static int tiny_max_seqs() { return tunables().tiny_max_seqs.load(std::memory_order_relaxed); }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_event_mutex;
std::vector<hipEvent_t> g_event_free[kMaxDevices][2];   // [device][0 = hipEventDisableTiming, 1 = timing]
int event_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  return dev;
}
}  // namespace

// An event belongs to the device that was current when it was created; recording it on another
// device's stream fails.  The free lists are therefore per device, and both calls use the device
// that is current in the calling thread (the one whose streams the caller passes).
hipEvent_t event_get(bool timing) {
  const int dev = event_device();
  if (dev >= 0) {
    std::lock_guard<std::mutex> lock(g_event_mutex);
    std::vector<hipEvent_t>& fl = g_event_free[dev][timing ? 1 : 0];
    if (!fl.empty()) {
      hipEvent_t ev = fl.back();
      fl.pop_back();
      return ev;
    }
  }
  hipEvent_t ev = nullptr;
  const hipError_t rc = timing ? hipEventCreate(&ev) : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  return rc == hipSuccess ? ev : nullptr;
}

void event_put(hipEvent_t ev, bool timing) {
  if (ev == nullptr) return;
  const int dev = event_device();
  if (dev < 0) {
    (void)hipEventDestroy(ev);
    return;
  }
  std::lock_guard<std::mutex> lock(g_event_mutex);
  g_event_free[dev][timing ? 1 : 0].push_back(ev);
}

static int device_cus();
bool resident_fits(int wgs) { return device_cus() >= 4 * wgs; }

static int device_cus() {
  static std::atomic<int> cus[64];      // per device, 0 = not asked yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

bool resident_fits_wgs(int wgs) { return device_cus() >= wgs; }

// The device's host-visible status word (grid_sync.hpp): 64 bytes of pinned, mapped host memory per
// device, allocated at the first resident launch and kept for the life of the process.
namespace {
std::mutex g_status_mutex;
unsigned* g_status_host[kMaxDevices];
unsigned* g_status_dev[kMaxDevices];
}  // namespace

unsigned* resident_status_word() {
  const int dev = event_device();
  if (dev < 0) return nullptr;
  std::lock_guard<std::mutex> lock(g_status_mutex);
  if (g_status_host[dev] == nullptr) {
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess ||
        hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (h != nullptr) (void)hipHostFree(h);
      return nullptr;
    }
    memset(h, 0, 64);
    g_status_host[dev] = static_cast<unsigned*>(h);
    g_status_dev[dev] = static_cast<unsigned*>(d);
  }
  return g_status_dev[dev];
}

GridSync make_grid_sync(unsigned* counter, unsigned* abort_word) {
  GridSync g;
  g.counter = counter;
  g.abort_word = abort_word;
  g.status_host = resident_status_word();
  const int ms = tunables().resident_timeout_ms.load(std::memory_order_relaxed);
  // s_memrealtime: 100 MHz.  0 = a wait gives up at its second clock check (~512 polls, tens of
  // microseconds): the value tests use to walk the abort path of a chain for real
  g.timeout_ticks = static_cast<uint64_t>(ms > 0 ? ms : 0) * 100000ull;
  return g;
}

static Tunables& global_tunables();

// Per device: the multi-step kernels (step chain, resident tails) are off after an acknowledged timeout.
static std::atomic<int> g_multi_off[kMaxDevices];

int multi_step_knob(const std::atomic<int>& knob) {
  const int dev = event_device();
  if (dev >= 0 && g_multi_off[dev].load(std::memory_order_relaxed) != 0) return 0;
  return knob.load(std::memory_order_relaxed);
}

static int resident_status(bool clear) {
  const int dev = event_device();
  if (dev < 0) return CMHSE_OK;
  std::lock_guard<std::mutex> lock(g_status_mutex);
  volatile unsigned* w = g_status_host[dev];
  if (w == nullptr || *w == 0) return CMHSE_OK;
  if (clear) {
    *w = 0;
    // A timeout means THIS device does not give the multi-step kernels what they need (workgroups
    // started in index order / all resident: a CU mask, another tenant).  The caller has been told
    // (this status); from here on every call on this device — whatever tuning context it runs in —
    // uses one launch per time step, which needs neither.  Other devices are not touched.
    // Setting one of the three knobs to a positive value (cmhse_tune / cmhse_ctx_tune) with this
    // device current re-enables them here.
    g_multi_off[dev].store(1, std::memory_order_relaxed);
  }
  return CMHSE_ERR_TIMEOUT;
}

int resident_check() { return resident_status(false); }

void stream_after(hipStream_t waiter, hipStream_t signal) {
  hipEvent_t ev = event_get(false);
  bool ordered = false;
  if (ev != nullptr) {
    // (the wait captures the record: the event may be re-recorded right after)
    ordered = hipEventRecord(ev, signal) == hipSuccess && hipStreamWaitEvent(waiter, ev, 0) == hipSuccess;
    event_put(ev, false);
  }
  if (!ordered) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(signal);     // host-side ordering: slower, never wrong
  }
}

// The process-wide defaults (cmhse_tune) and, while a thread is inside cmhse_ctx_enter ...
// cmhse_ctx_leave, that thread's CONTEXT: a private copy of the crossovers (cmhse_ctx_create).  Every
// read of a crossover anywhere in the library goes through tunables(), so a call made inside a
// context — workspace sizing included — sees that context's values and nothing another thread or
// another context does to its own.
static Tunables& global_tunables() {
  static Tunables t;
  return t;
}
static thread_local Tunables* tl_ctx = nullptr;

Tunables& tunables() { return tl_ctx != nullptr ? *tl_ctx : global_tunables(); }

}  // namespace cmhse

using namespace cmhse;

extern "C" size_t cmhse_gru_pool_workspace(int32_t S, int32_t Tmax, int64_t sum_T, int32_t I,
                                           int32_t H, int32_t pool_mode) {
  if (sum_T <= 0 || H <= 0 || S <= 0 || I <= 0) return 0;
  return gru_ws_layout(S, sum_T, H, pool_mode, I, Tmax).total;
}

extern "C" int cmhse_gru_pool_ws_region(int32_t S, int32_t Tmax, int64_t sum_T, int32_t I, int32_t H,
                                        int32_t pool_mode, const char* name, size_t* offset,
                                        size_t* bytes) {
  if (sum_T <= 0 || H <= 0 || S <= 0 || I <= 0 || !name || !offset || !bytes) return CMHSE_ERR_ARG;
  const GruWs L = gru_ws_layout(S, sum_T, H, pool_mode, I, Tmax);
  if (!strcmp(name, "hs")) {
    *offset = L.hs, *bytes = static_cast<size_t>(sum_T) * H * sizeof(float);
  } else if (!strcmp(name, "gates")) {
    *offset = L.gates, *bytes = L.argmax > L.gates ? static_cast<size_t>(sum_T) * 4 * H * sizeof(float) : 0;
  } else if (!strcmp(name, "argmax")) {
    *offset = L.argmax, *bytes = L.v > L.argmax ? static_cast<size_t>(S) * H * sizeof(int32_t) : 0;
  } else if (!strcmp(name, "v")) {
    *offset = L.v, *bytes = L.wih_s > L.v ? static_cast<size_t>(sum_T) * H * sizeof(float) : 0;
  } else {
    return CMHSE_ERR_ARG;
  }
  return CMHSE_OK;
}

namespace {

// One validated cmhse_gru_pool_fwd request: step-kernel parameters plus what the pooling tail needs.
struct FwdJob {
  GruStepParams p;
  const cmhse_seq_batch* b;
  const cmhse_gru_weights* w;
  float* out;
  char* wsb;
  GruWs L;
  int64_t sum_T, off;
  int32_t pool_mode;
  bool vec, bf3, save;
  int32_t t_mid;             // first step served by the mid-size kernel (Tmax: none)
  int64_t rows_split;        // bf16x3: packed rows of the steps the tiled bf16x3 kernel serves
  hipStream_t tail_stream;   // optional stream the call's remaining steps move to when this chain ends early
  hipStream_t own_stream;    // optional stream ALL launches of this request go to (forked from / joined into the call's)
  hipStream_t side_stream;   // optional stream for throughput work beside a small-batch chain (projection chunks, attention)
  bool pooled;               // attention already launched (early, beside the others' tail)
  hipEvent_t ready_event;    // optional: recorded where `out` of this request becomes final (cmhse_gru_job.out_ready_event)
  bool ready_marked;
  int64_t att_rows_done;     // packed rows whose attention energies are already launched
  const int32_t* kind_count; // HOST [Tmax]: active sequences the KIND of step t's kernel is chosen from — the batch's own
                             // step counts, or the caller's step_plan_host (the counts of the whole split this batch is a share of)
  int32_t tail_lo;           // steps >= tail_lo run inside ONE resident kernel (gru_fwd_tail_kernel); -1 = none
  int32_t chain_until;       // steps < chain_until are inside a queued gru_step_chain_kernel launch
};

// 128-row tiles (2 workgroups per CU, 230 registers per lane) halve the weight bytes and cut the
// LDS fragment reads per MFMA by a third; 64-row tiles (3 per CU) have half the work per wave, so
// a launch of only a round or two of workgroups ends sooner.  Measured on boxes that hold
// 1.75-2.0 GHz under this load: full split 287-290 ms per pass with 128 rows against 297-303 with
// 64 (C3D 185 / 190.7); a 615-video share of the split 48.4 against 42.9; 1230 videos equal.
// Launches of at least Tunables::tall_tile_min_wgs (2048) 64-row workgroups (all fp32 tiled
// requests of the time step together) therefore use the 128-row tile.
int gru_msub_for(int wgs64) {
  const int thr = tunables().tall_tile_min_wgs.load(std::memory_order_relaxed);
  return (thr > 0 && wgs64 >= thr) ? 2 : 1;
}

int prepare_job(const cmhse_seq_batch* b, const cmhse_gru_weights* w, int32_t pool_mode, float* out,
                void* workspace, size_t workspace_bytes, hipStream_t stream, FwdJob* job) {
  if (!b || !w || !out || !workspace) return CMHSE_ERR_ARG;
  if (b->S <= 0 || b->Tmax <= 0 || b->I <= 0 || b->H <= 0) return CMHSE_ERR_ARG;
  if ((b->x_rows == nullptr) == (b->tok_rows == nullptr)) return CMHSE_ERR_ARG;
  if (b->tok_rows && (!b->emb_table || b->vocab <= 0)) return CMHSE_ERR_ARG;
  if (!b->lens || !b->out_row || !b->step_off || !b->step_count_host) return CMHSE_ERR_ARG;
  if (!w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh) return CMHSE_ERR_ARG;
  const int32_t mode_flags = pool_mode;
  const bool save = (pool_mode & CMHSE_SAVE_FOR_BACKWARD) != 0;
  bool bf3 = (pool_mode & CMHSE_MATH_BF16X3) != 0;
  pool_mode &= kModeMask;
  if (pool_mode != CMHSE_POOL_LAST && pool_mode != CMHSE_POOL_ATTN && pool_mode != CMHSE_POOL_MAX &&
      pool_mode != CMHSE_POOL_ALL)
    return CMHSE_ERR_ARG;
  if (pool_mode == CMHSE_POOL_ATTN && (!w->w_lin || !w->b_lin || !w->w_att)) return CMHSE_ERR_ARG;
  if (b->x_rows && b->x_step_floats != 0 && b->x_step_floats < b->I) return CMHSE_ERR_ARG;
  int64_t sum_T = 0;
  for (int t = 0; t < b->Tmax; ++t) {
    const int c = b->step_count_host[t];
    if (c <= 0 || c > b->S || (t > 0 && c > b->step_count_host[t - 1])) return CMHSE_ERR_ARG;
    sum_T += c;
  }
  if (b->step_count_host[0] != b->S) return CMHSE_ERR_ARG;
  if (b->step_plan_host != nullptr)   // a share cannot have more active sequences than the whole
    for (int t = 0; t < b->Tmax; ++t)
      if (b->step_plan_host[t] < b->step_count_host[t] || (t > 0 && b->step_plan_host[t] > b->step_plan_host[t - 1]))
        return CMHSE_ERR_ARG;
  if (sum_T * b->H >= (int64_t(1) << 40)) return CMHSE_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_gru_pool_workspace(b->S, b->Tmax, sum_T, b->I, b->H, mode_flags))
    return CMHSE_ERR_WORKSPACE;
  job->b = b;
  job->w = w;
  job->out = out;
  job->kind_count = b->step_plan_host != nullptr ? b->step_plan_host : b->step_count_host;
  job->L = gru_ws_layout(b->S, sum_T, b->H, mode_flags, b->I, b->Tmax);
  job->wsb = static_cast<char*>(workspace);
  job->sum_T = sum_T;
  job->off = 0;
  job->pool_mode = pool_mode;
  job->save = save;
  const GruWs& L = job->L;
  char* wsb = job->wsb;

  GruStepParams& p = job->p;
  p.x_rows = b->x_rows;
  p.tok_rows = b->tok_rows;
  p.emb = b->emb_table;
  p.h0_rows = b->h0_rows;
  p.lens = b->lens;
  p.out_row = b->out_row;
  p.w_ih = w->w_ih;
  p.w_hh = w->w_hh;
  p.b_ih = w->b_ih;
  p.b_hh = w->b_hh;
  p.hs = reinterpret_cast<float*>(wsb + L.hs);
  p.out = out;
  p.gates = save ? reinterpret_cast<float*>(wsb + L.gates) : nullptr;
  p.argmax = (save && pool_mode == CMHSE_POOL_MAX) ? reinterpret_cast<int32_t*>(wsb + L.argmax) : nullptr;
  p.I = b->I;
  p.H = b->H;
  p.vocab = b->vocab;
  p.pool_mode = pool_mode;
  p.x_step = b->x_step_floats;
  p.n_tiles = (b->H + kGruBU - 1) / kGruBU;
  // dwordx4 operand loads need K % 4 == 0 in both phases (row bases are then 4-float multiples)
  job->vec = (b->I % 4 == 0) && (b->H % 4 == 0);
  // bf16x3 serves the LDS-tiled kernels of inference calls only (the latency-shaped tiny kernel stays
  // exact fp32; a training call's backward consumes what the exact forward saved)
  job->bf3 = bf3 && !save && job->vec && (job->kind_count[0] > tiny_max_seqs() || pool_mode == CMHSE_POOL_ATTN);
  // steps with few active sequences: mid-size kernel on a hoisted input projection
  job->t_mid = b->Tmax;
  p.gx = nullptr;
  p.gx_p0 = 0;
  p.gx_per_seq = 0;
  if (job->vec && mid_max_seqs() > 0) {
    int64_t p0 = 0;
    for (int t = 0; t < b->Tmax; ++t) {
      if (job->kind_count[t] <= mid_max_seqs()) {
        job->t_mid = t;
        break;
      }
      p0 += b->step_count_host[t];
    }
    p.gx = reinterpret_cast<float*>(wsb + L.gx);
    p.gx_p0 = p0;
    p.gx_per_seq = (b->x_rows != nullptr && b->x_step_floats == 0) ? 1 : 0;
  }
  p.w_ih_s = nullptr;
  p.w_hh_s = nullptr;
  p.xs = nullptr;
  p.hs_s = nullptr;
  p.h0_s = nullptr;
  job->rows_split = 0;
  if (job->bf3) {
    float* wih_s = reinterpret_cast<float*>(wsb + L.wih_s);
    float* whh_s = reinterpret_cast<float*>(wsb + L.whh_s);
    launch_split(w->w_ih, wih_s, 3 * b->H, b->I, stream);
    launch_split(w->w_hh, whh_s, 3 * b->H, b->H, stream);
    p.w_ih_s = wih_s;
    p.w_hh_s = whh_s;
    // the steps the tiled bf16x3 kernel serves are a prefix (S_t is non-increasing)
    for (int t = 0; t < b->Tmax && t < job->t_mid && job->kind_count[t] > tiny_max_seqs(); ++t)
      job->rows_split += b->step_count_host[t];
    p.xs = reinterpret_cast<float*>(wsb + L.xs);
    p.hs_s = reinterpret_cast<float*>(wsb + L.hs_s);
    p.h0_s = reinterpret_cast<float*>(wsb + L.h0_s);
    if (job->rows_split > 0) {
      // their input rows pre-split in one pass (an upload still in flight must land first)
      if (b->step_events_host != nullptr)
        for (int t = 0; t < b->Tmax; ++t)
          if (b->step_events_host[t] != nullptr)
            (void)hipStreamWaitEvent(stream, static_cast<hipEvent_t>(const_cast<void*>(b->step_events_host[t])), 0);
      SplitRowsParams sr;
      sr.x_rows = b->x_rows; sr.tok_rows = b->tok_rows; sr.emb = b->emb_table;
      sr.step_off = b->step_off;
      sr.out = reinterpret_cast<uint32_t*>(wsb + L.xs);
      sr.I = b->I; sr.vocab = b->vocab; sr.x_step = b->x_step_floats; sr.Tmax = b->Tmax;
      hipLaunchKernelGGL(split_rows_kernel, dim3(static_cast<unsigned>(job->rows_split)),
                         dim3(kThreads), 0, stream, sr);
      if (b->h0_rows != nullptr) {   // the caller's initial states, one row per sorted sequence
        SplitRowsParams sh = sr;
        sh.x_rows = b->h0_rows; sh.tok_rows = nullptr;
        sh.out = reinterpret_cast<uint32_t*>(wsb + L.h0_s);
        sh.I = b->H; sh.x_step = 0; sh.Tmax = 1;
        hipLaunchKernelGGL(split_rows_kernel, dim3(static_cast<unsigned>(b->S)), dim3(kThreads), 0,
                           stream, sh);
      }
      if (b->H % 16 != 0)   // the epilogue writes whole (hi, lo) pairs; the chunk padding must read 0
        (void)hipMemsetAsync(wsb + L.hs_s, 0,
                             static_cast<size_t>(job->rows_split) * split_ld(b->H) * sizeof(float), stream);
    }
  }
  return CMHSE_OK;
}

// Which step kernel serves job `j` at its current step: 0 = tiny, 1 = tiled fp32, 2 = tiled bf16x3;
// bit 2 = scalar-load variant.  Jobs of equal kind share a launch.
// Hidden units per workgroup of the mid-size step: the narrowest of 16, 8, 4 whose grid still
// fits the chip in one round (more, smaller tiles = more CUs pulling operands; past one round the
// replicated h rows cost more than the spread gains).  Tunables::mid_units = 16 | 8 | 4 forces one.
constexpr int kChipCUs = 256;
static int mid_m_blocks(int S_t) { return (S_t <= 16) ? 1 : (S_t + 31) / 32; }
static int mid_units(int H, int m_blocks) {
  const int forced = tunables().mid_units.load(std::memory_order_relaxed);
  if (forced == 16 || forced == 8 || forced == 4) return forced;
  for (int bu = 4; bu < 16; bu *= 2)
    if (((H + bu - 1) / bu) * m_blocks <= kChipCUs) return bu;
  return 16;
}

// `mid_blocks`: 16/32-sequence blocks of ALL requests of the call that run a mid-size step at this
// time step (they share the chip, and a launch when their shapes agree).  `alone`: nothing else of
// the call competes for workgroup slots at this step (no LDS-tiled step, no chain moved to the
// side stream, no attention pass started beside the steps) — the 8-wave shape is used (bit 512);
// it and the 4-wave shape compute bit-identical results (kMidSlices).
// `tiled_wgs`: 64-row workgroups of all requests that run an fp32 LDS-tiled step at this time step
// (bit 2048 = 128-row tiles, gru_msub_for()).
int step_kind(const FwdJob& j, int S_t, int mid_blocks, bool alone, int tiled_wgs) {
  if (j.p.t >= j.t_mid) {   // mid-size kernel (vec shapes only)
    const int waves = tunables().mid_waves.load(std::memory_order_relaxed);   // 4 | 8 forces a shape
    if (waves == 4) alone = false;
    if (waves == 8) alone = true;
    const int bu = mid_units(j.b->H, mid_blocks);
    // many sequences: 64 per workgroup, so that the step is ONE round of workgroups (H / 16 x
    // ceil(S_t / 64) <= 256 up to 256 sequences at H = 1024) instead of two of 32-sequence ones
    const int tall = tunables().mid_tall_min_seqs.load(std::memory_order_relaxed);
    // (a 48-sequence tile for 129-192 sequences — one round of 192 / 256 workgroups instead of 192 of
    // this shape — was built in round 4, bit-identical, and changed nothing: profiles/r04_train_ab.txt)
    if (tall > 0 && S_t >= tall && bu == 16 && alone && j.save) return 3 | 512 | 8192;   // (training calls: the validation pass keeps its shapes)
    return 3 | (S_t <= 16 ? 32 : 0) | (bu == 8 ? 128 : 0) | (bu == 4 ? 256 : 0) |
           (alone ? 512 : 0);
  }
  // (WHICH of the differently-ordered sums serves the step comes from kind_count — the whole
  // split's active count when the batch is a share of one — so that a sequence sees the same
  // arithmetic whatever else is in its batch; shapes within a kind are bit-identical)
  int k = (j.kind_count[j.p.t] <= tiny_max_seqs()) ? 0 : (j.bf3 ? 2 : 1);
  if (k == 1 && gru_msub_for(tiled_wgs) == 2) k |= 2048;
  return k | (j.vec ? 0 : 4);
}

// The input projection of the steps >= t_mid of a job: gx rows [m_begin, m_end) (relative to the
// first hoisted row; m_end < 0 = all of them).
void launch_xproj(const FwdJob& j, hipStream_t stream, int64_t m_begin = 0, int64_t m_end = -1) {
  const cmhse_seq_batch* b = j.b;
  const int t_first = j.t_mid;
  XprojParams q;
  q.x_rows = b->x_rows;
  q.tok_rows = b->tok_rows;
  q.emb = b->emb_table;
  q.step_off = b->step_off;
  q.w_ih = j.w->w_ih;
  q.gx = const_cast<float*>(j.p.gx);
  q.p0 = j.p.gx_p0;
  q.per_seq = j.p.gx_per_seq;
  q.rows = q.per_seq ? b->step_count_host[t_first] : (j.sum_T - j.p.gx_p0);
  if (m_end >= 0 && m_end < q.rows) q.rows = m_end;
  q.m_begin = m_begin;
  q.I = b->I;
  q.N = 3 * b->H;
  q.vocab = b->vocab;
  q.x_step = b->x_step_floats;
  q.Tmax = b->Tmax;
  q.t_first = t_first;
  q.n_tiles = (q.N + 191) / 192;
  const int64_t grid = static_cast<int64_t>(q.n_tiles) * ((q.rows - q.m_begin + 63) / 64);
  if (grid <= 0) return;
  const size_t smem = TileSmem<64, 192>::kBytes;
  hipLaunchKernelGGL(xproj_kernel, dim3(static_cast<unsigned>(grid)), dim3(kThreads), smem, stream, q);
}

void launch_group(const GruStepGroup& g, int kind, unsigned grid, hipStream_t stream) {
  const bool vec = (kind & 4) == 0;
  const int msub = (kind & 2048) != 0 ? 2 : 1;
  switch (kind & 3) {
    case 3: {
      const int bu = (kind & 256) != 0 ? 4 : ((kind & 128) != 0 ? 8 : 16);
#define MID_LAUNCH_(MB, BU)                                                                      \
  do {                                                                                                \
    if ((kind & 512) != 0)                                                                            \
      hipLaunchKernelGGL((gru_step_mid_kernel<MB, BU, 8>), dim3(grid), dim3(512), 0, stream, g);      \
    else                                                                                              \
      hipLaunchKernelGGL((gru_step_mid_kernel<MB, BU, 4>), dim3(grid), dim3(256), 0, stream, g);      \
  } while (0)
      if ((kind & 8192) != 0) {       // 64 sequences per workgroup (only with 16 units, 8 waves)
        hipLaunchKernelGGL((gru_step_mid_kernel<4, 16, 8>), dim3(grid), dim3(512), 0, stream, g);
      } else if ((kind & 32) != 0) {
        if (bu == 4) MID_LAUNCH_(1, 4);
        else if (bu == 8) MID_LAUNCH_(1, 8);
        else MID_LAUNCH_(1, 16);
      } else {
        if (bu == 4) MID_LAUNCH_(2, 4);
        else if (bu == 8) MID_LAUNCH_(2, 8);
        else MID_LAUNCH_(2, 16);
      }
#undef MID_LAUNCH_
      break;
    }
    case 0:
      if (vec) {
        hipLaunchKernelGGL((gru_step_tiny_kernel<true, 4>), dim3(grid), dim3(kThreads), 0, stream, g);
      } else {
        hipLaunchKernelGGL((gru_step_tiny_kernel<false, 4>), dim3(grid), dim3(kThreads), 0, stream, g);
      }
      break;
    case 2: {
      // staging-bound loop: the 128-row tile halves the weight bytes per MFMA
      const size_t smem = RingSmem<128, 3 * kGruBU>::kBytes;
      hipLaunchKernelGGL((gru_step_kernel<true, 2, true>), dim3(grid), dim3(kThreads), smem, stream, g);
      break;
    }
    default:
      if (msub == 2) {
        const size_t smem = TileSmem<128, 3 * kGruBU>::kBytes;
        if (vec)
          hipLaunchKernelGGL((gru_step_kernel<true, 2, false>), dim3(grid), dim3(kThreads), smem, stream, g);
        else
          hipLaunchKernelGGL((gru_step_kernel<false, 2, false>), dim3(grid), dim3(kThreads), smem, stream, g);
      } else {
        const size_t smem = TileSmem<64, 3 * kGruBU>::kBytes;
        if (vec)
          hipLaunchKernelGGL((gru_step_kernel<true, 1, false>), dim3(grid), dim3(kThreads), smem, stream, g);
        else
          hipLaunchKernelGGL((gru_step_kernel<false, 1, false>), dim3(grid), dim3(kThreads), smem, stream, g);
      }
  }
}

unsigned step_grid(const FwdJob& j, int kind, int S_t) {
  const int H = j.b->H;
  if ((kind & 3) == 3) {
    const int bm = (kind & 8192) != 0 ? 64 : ((kind & 32) != 0 ? 16 : 32);
    const int bu = (kind & 256) != 0 ? 4 : ((kind & 128) != 0 ? 8 : 16);
    return static_cast<unsigned>((H + bu - 1) / bu) * ((S_t + bm - 1) / bm);
  }
  if ((kind & 3) == 0)
    return static_cast<unsigned>((H + kTinyBU - 1) / kTinyBU) * ((S_t + kTinyBM - 1) / kTinyBM);
  const int bm = ((kind & 3) == 2 || (kind & 2048) != 0) ? 128 : 64;
  const int m_tiles = (S_t + bm - 1) / bm;
  return static_cast<unsigned>(j.p.n_tiles) * m_tiles;
}

// Time steps of all jobs, step t of every still-running job in as few launches as kinds allow.
int launch_attention(FwdJob& job, hipStream_t stream, int64_t row_end, bool pool);

// Rows of the input projection in front of a chunked chain / per later chunk (launch_steps)
constexpr int64_t kXprojFirstRows = 512;
static int64_t xproj_chunk_rows() { return tunables().xproj_chunk_rows.load(std::memory_order_relaxed); }
static void launch_fwd_tail(FwdJob& j, hipStream_t stream) {
  FwdTailParams q;
  q.p = j.p;
  q.step_off = j.b->step_off;
  unsigned* words = reinterpret_cast<unsigned*>(j.wsb + j.L.tail_sync);
  q.sync = make_grid_sync(words, words + 63);
  q.t_lo = j.tail_lo;
  q.t_hi = j.b->Tmax - 1;
  const int H = j.b->H;
  const int kb = 2 * ((H / 16 + 15) / 16);   // 16-k blocks per wave, whole pairs (mid_phase's ownership)
  const dim3 grid(static_cast<unsigned>(H / 16)), block(512);
  const bool two = j.b->step_count_host[j.tail_lo] > 16;   // 17 ... 32 sequences at the tail's first step
#define FWD_TAIL_(KB)                                                                         \
  do {                                                                                        \
    if (two) hipLaunchKernelGGL((gru_fwd_tail_kernel<KB, 2>), grid, block, 0, stream, q);     \
    else hipLaunchKernelGGL((gru_fwd_tail_kernel<KB, 1>), grid, block, 0, stream, q);         \
  } while (0)
  if (kb <= 2) FWD_TAIL_(2);
  else if (kb <= 4) FWD_TAIL_(4);
  else FWD_TAIL_(8);
#undef FWD_TAIL_
}

struct XprojPlan {
  std::vector<int> step;        // first step of chunk c (c >= 1)
  std::vector<hipEvent_t> ev;   // recorded behind chunk c's launch on the side stream
};

// The requests of a call that run an fp32 LDS-tiled step at time step t and will go on doing so
// TOGETHER: which of them (in_chain), up to which step (end, exclusive; 0 = no chain here), with
// which tile height (kind, bit 2048 = 128 rows) and on which stream.
struct ChainPlan {
  bool in_chain[kMaxJobs];
  int end, kind;
  hipStream_t stream;
};

static ChainPlan plan_chain(const FwdJob* jobs, int n, int t, const int* kind, const bool* done,
                            const hipStream_t* js, int tiled_wgs) {
  ChainPlan c;
  for (int k = 0; k < kMaxJobs; ++k) c.in_chain[k] = false;
  c.end = 0;
  c.kind = -1;
  c.stream = nullptr;

  const int min_steps = multi_step_knob(tunables().chain_min_steps);
  int n_c = 0, tiles_c = -1;
  bool ok = min_steps > 0;
  for (int k = 0; k < n && ok; ++k) {
    if (done[k] || (kind[k] & 3) != 1) continue;
    const FwdJob& j = jobs[k];
    // (H % 32: a state row must be whole 128-byte cache lines — a reader that pulled a line
    // shared with the NEXT row tile's first row into its L2 before that row was written would
    // leave a stale copy there for the tile that needs it)
    if (t < j.chain_until || j.save || j.bf3 || j.tail_lo >= 0 || j.b->H % 32 != 0) {
      ok = false;
    } else if (n_c == 0) {
      c.kind = kind[k];
      tiles_c = j.p.n_tiles;
      c.stream = js[k];
    } else if (kind[k] != c.kind || j.p.n_tiles != tiles_c || js[k] != c.stream) {
      ok = false;
    }
    c.in_chain[k] = true;
    ++n_c;
  }
  if (ok && n_c > 0) {
    // Tile height of a chain.  Nothing drains between steps here, so the 128-row tile (the more
    // efficient one) pays from far fewer workgroups per step than with per-step launches:
    // chain_tall_min_wgs (256) 64-row workgroups when every request's x phase is at least as long
    // as its h phase (I >= H: a tile has that much work in front of its wait), four times that
    // otherwise — a chain of h-dominated tiles is a latency chain, and half as many, twice as
    // long tiles lengthen it (one tower alone, I = 300, 1100-2048 sequences: 185-194 us per
    // step with 128 rows against 128-147 with 64; I = 2048: level; profiles/r04_step_chain.txt).
    int tall = tunables().chain_tall_min_wgs.load(std::memory_order_relaxed);
    for (int k = 0; k < n; ++k)
      if (c.in_chain[k] && jobs[k].b->I < jobs[k].b->H) {
        tall *= 4;
        break;
      }
    c.kind = (c.kind & ~2048) | ((tall > 0 && tiled_wgs >= tall) ? 2048 : 0);
    c.end = t + 1;
    while (c.end - t < kChainMaxSteps) {
      bool same = true;
      for (int k = 0; k < n && same; ++k) {
        const FwdJob& j = jobs[k];
        const bool tiled_next = c.end < j.b->Tmax && c.end < j.t_mid && !j.bf3 &&
                                j.kind_count[c.end] > tiny_max_seqs();
        same = tiled_next == c.in_chain[k];
        // a step whose inputs are still crossing PCIe (cmhse_pull_steps: an event per chunk of
        // time steps) starts a new chain, launched behind that event
        if (same && c.in_chain[k] && j.b->step_events_host != nullptr && j.b->step_events_host[c.end] != nullptr)
          same = false;
      }
      if (!same) break;
      ++c.end;
    }
    // one workgroup per task: HIP rejects a launch of more than 2^32 - 1 threads, i.e. 2^24 - 1
    // workgroups of 256 (halve the chain until it fits)
    for (;;) {
      const int bm_c = (c.kind & 2048) != 0 ? 128 : 64;
      uint64_t units_c = 0;
      for (int q = t; q < c.end; ++q)
        for (int k = 0; k < n; ++k)
          if (c.in_chain[k])
            units_c += static_cast<uint64_t>((jobs[k].b->step_count_host[q] + bm_c - 1) / bm_c);
      if (units_c * static_cast<uint64_t>(tiles_c) * kThreads <= 0xffffffffULL || c.end - t <= 1) break;
      c.end = t + (c.end - t) / 2;
    }
    ok = c.end - t >= min_steps;
  }
  if (!ok || n_c == 0) {
    c.end = 0;
    for (int k = 0; k < kMaxJobs; ++k) c.in_chain[k] = false;
  }
  return c;
}

// Queue ONE gru_step_chain_kernel launch for steps [t, c.end) of the requests in `c` (counters and
// tickets zeroed on the chain's stream in front of it; with `timer`, an event pair around it).
static void launch_chain(FwdJob* jobs, int n, int t, const ChainPlan& c, Timer* timer) {
  GruChainGroup cg;
  cg.n = 0;
  cg.t0 = t;
  cg.nsteps = c.end - t;
  cg.ticket = nullptr;
  unsigned* abort_word = nullptr;
  const int bm = (c.kind & 2048) != 0 ? 128 : 64;
  for (int k = 0; k < n; ++k) {
    if (!c.in_chain[k]) continue;
    FwdJob& j = jobs[k];
    const int q = cg.n++;
    cg.j[q] = j.p;
    cg.n_tiles = j.p.n_tiles;
    cg.step_off[q] = j.b->step_off;
    cg.rt_stride[q] = (j.b->step_count_host[t] + bm - 1) / bm;
    unsigned* words = reinterpret_cast<unsigned*>(j.wsb + j.L.chain_sync);
    cg.done[q] = words + 64;
    (void)hipMemsetAsync(words, 0, 256 + sizeof(unsigned) * static_cast<size_t>(cg.nsteps) * cg.rt_stride[q],
                         c.stream);
    if (q == 0) {
      cg.ticket = words;
      abort_word = words + kXcds;
    }
    j.chain_until = c.end;
  }
  for (int q = cg.n; q < kMaxJobs; ++q) {
    cg.step_off[q] = nullptr;
    cg.done[q] = nullptr;
    cg.rt_stride[q] = 0;
  }
  // per-queue tickets, step by step (GruChainGroup)
  const int nq = chain_queues(cg.n_tiles);
  const unsigned cols = static_cast<unsigned>(cg.n_tiles / nq);
  const int n_phases = cg.nsteps;
  uint32_t count[kChainPhases];
  for (int p = 0; p < kChainPhases; ++p) count[p] = 0;
  double flops = 0.0, bytes = 0.0;
  for (int sidx = 0; sidx < cg.nsteps; ++sidx) {
    for (int k = 0; k < n; ++k) {
      if (!c.in_chain[k]) continue;
      const int S_k = jobs[k].b->step_count_host[t + sidx];
      const unsigned rt = static_cast<unsigned>((S_k + bm - 1) / bm);
      count[sidx] += rt * cols;
      const double I = jobs[k].p.I, H = jobs[k].p.H;
      flops += S_k * (2.0 * 3.0 * H * (I + H) + 14.0 * H);
      bytes += S_k * 4.0 * (I + 2.0 * H) + 12.0 * H * (I + H);
    }
  }
  uint64_t total = 0;
  for (int p = 0; p <= kChainPhases; ++p) {
    cg.tick[p] = static_cast<uint32_t>(total);
    if (p < n_phases) total += count[p];
  }
  cg.sync = make_grid_sync(nullptr, abort_word);
  const unsigned cgrid = static_cast<unsigned>(total * static_cast<uint64_t>(nq));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timer != nullptr) {
    e0 = event_get(true);
    e1 = e0 ? event_get(true) : nullptr;
    if (e0 && !e1) {
      event_put(e0, true);
      e0 = nullptr;
    }
    if (e0 && e1) (void)hipEventRecord(e0, c.stream);
  }
  const bool cvec = (c.kind & 4) == 0;
#define CHAIN_LAUNCH_(V, M, SMEM) \
  hipLaunchKernelGGL((gru_step_chain_kernel<V, M>), dim3(cgrid), dim3(kThreads), SMEM, c.stream, cg)
  if (bm == 128) {
    if (cvec) CHAIN_LAUNCH_(true, 2, (TileSmem<128, 3 * kGruBU>::kBytes));
    else CHAIN_LAUNCH_(false, 2, (TileSmem<128, 3 * kGruBU>::kBytes));
  } else {
    if (cvec) CHAIN_LAUNCH_(true, 1, (TileSmem<64, 3 * kGruBU>::kBytes));
    else CHAIN_LAUNCH_(false, 1, (TileSmem<64, 3 * kGruBU>::kBytes));
  }
#undef CHAIN_LAUNCH_
  if (e0 && e1) {
    (void)hipEventRecord(e1, c.stream);
    timer->tiled_events.push_back(e0);
    timer->tiled_events.push_back(e1);
    timer->tiled_flops += flops;
    timer->tiled_bytes += bytes;
  }
}

// `out` of job j is final behind everything queued on `stream` so far: tell the caller's event.
static inline void mark_ready(FwdJob& j, hipStream_t stream) {
  if (j.ready_event == nullptr || j.ready_marked) return;
  (void)hipEventRecord(j.ready_event, stream);
  j.ready_marked = true;
}

int launch_steps(FwdJob* jobs, int n, hipStream_t main_stream, Timer* timer) {
  int Tmax = 0, launches = 0;
  XprojPlan plan[kMaxJobs];
  // Every chain starts on the caller's stream; a chain moves to the call's side stream (at most
  // once, ordered by an event) when it should run BESIDE the others instead of between them:
  //   (a) it has dropped to small-batch steps while another chain still launches LDS-tiled steps
  //       that do not fill the chip (a rank's share of the split: 28-44 row tiles x 16 < 768
  //       workgroup slots) — its short launches, and later its long few-sequence tail, then hide
  //       under the other chain's steps instead of following each of them;
  //   (b) another chain has ended and starts its attention pass on the caller's stream.
  hipStream_t side = nullptr;
  for (int k = 0; k < n; ++k)
    if (jobs[k].tail_stream != nullptr && jobs[k].tail_stream != main_stream) side = jobs[k].tail_stream;
  hipStream_t js[kMaxJobs];
  for (int k = 0; k < kMaxJobs; ++k) js[k] = main_stream;
  // A request with a stream of its own (the towers of a training step: independent latency chains
  // that should advance side by side from their first step) runs there from start to end; the
  // host interleaves the launches of all requests step by step, so no chain waits for another
  // one's launches to be queued.
  for (int k = 0; k < n; ++k)
    if (jobs[k].own_stream != nullptr && jobs[k].own_stream != main_stream) {
      bool seen = false;
      for (int m = 0; m < k; ++m) seen = seen || js[m] == jobs[k].own_stream;
      if (!seen) stream_after(jobs[k].own_stream, main_stream);
      js[k] = jobs[k].own_stream;
    }
  // The few-sequence tail of a TRAINING chain on its own stream as one resident kernel (below).  The tail
  // kernel's arithmetic is the small-batch kernel's (gru_step_mid_kernel<1, 16, 8>), so the choice
  // follows the LOCAL counts and leaves the bits alone.  (For an inference chain the same kernel was
  // measured slower than 7-us launches that use every CU, profiles/r05_rank_share.txt: not offered.)
  for (int k = 0; k < n; ++k) {
    FwdJob& j = jobs[k];
    j.tail_lo = -1;
    j.chain_until = 0;
    const cmhse_seq_batch* b = j.b;
    const int min_steps = multi_step_knob(tunables().fwd_tail_min_steps);
    if (!j.save || min_steps <= 0 || j.bf3 || !j.vec || b->H % 16 != 0 || b->H > 1024 || timer != nullptr ||
        !resident_fits(b->H / 16))
      continue;
    const int floor_t = 1;
    if (js[k] == main_stream || j.t_mid != 0) continue;
    int lo = b->Tmax;
    while (lo - 1 >= floor_t && b->step_count_host[lo - 1] <= kFwdTailMaxSeqs) --lo;
    if (b->Tmax - lo < min_steps) continue;
    j.tail_lo = lo;
    (void)hipMemsetAsync(j.wsb + j.L.tail_sync, 0, 256, js[k]);   // the barrier counter, off the chain's path
  }
  // The hoisted input projection of an inference chain's small-batch steps (x W_ih^T of the rows of
  // the steps >= t_mid) depends on the inputs only.  With a side stream it is launched THERE, now,
  // beside the tiled steps, instead of on the chain's stream when the chain reaches t_mid, where it
  // stood in front of the text tower's few-sequence tail — the end of a rank's share of the split is
  // that tail, 1.7 of its 41 ms were this GEMM (profiles/r05_rank_share.txt).  Step t_mid waits for
  // the event.  (Inputs still crossing PCIe — step_events_host — keep the in-order form.)
  hipEvent_t early_xproj[kMaxJobs];
  for (int k = 0; k < kMaxJobs; ++k) early_xproj[k] = nullptr;
  if (side != nullptr && tunables().early_xproj.load(std::memory_order_relaxed) != 0) {
    bool side_ready = false;
    for (int k = 0; k < n; ++k) {
      const FwdJob& j = jobs[k];
      if (j.save || j.t_mid <= 0 || j.t_mid >= j.b->Tmax || j.b->step_events_host != nullptr ||
          j.own_stream != nullptr)
        continue;
      if (!side_ready) {
        stream_after(side, main_stream);     // the caller's inputs are ready
        side_ready = true;
      }
      launch_xproj(j, side);
      hipEvent_t ev = event_get(false);
      if (ev != nullptr && hipEventRecord(ev, side) == hipSuccess) {
        early_xproj[k] = ev;
      } else {                               // no event: order the whole stream instead
        event_put(ev, false);
        (void)hipGetLastError();
        (void)hipStreamSynchronize(side);
        early_xproj[k] = reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(1));   // done, nothing to wait for
      }
    }
  }
  bool forked = false;
  auto fork = [&](int k) {
    if (side == nullptr || js[k] != main_stream) return;
    stream_after(side, main_stream);
    js[k] = side;
    forked = true;
  };
  for (int k = 0; k < n; ++k) Tmax = jobs[k].b->Tmax > Tmax ? jobs[k].b->Tmax : Tmax;
  for (int t = 0; t < Tmax; ++t) {
    int kind[kMaxJobs];
    bool done[kMaxJobs];
    bool any_tiled = false;
    int mid_blocks = 0;
    for (int k = 0; k < n; ++k)
      if (t < jobs[k].b->Tmax && t >= jobs[k].t_mid && !(jobs[k].tail_lo >= 0 && t >= jobs[k].tail_lo))
        mid_blocks += mid_m_blocks(jobs[k].b->step_count_host[t]);
    bool alone = !forked;
    int tiled_wgs = 0;
    for (int k = 0; k < n; ++k)
      if (t < jobs[k].b->Tmax && t < jobs[k].t_mid) {
        alone = false;   // a tiled / tiny step runs too
        const int S_k = jobs[k].b->step_count_host[t];
        if (jobs[k].kind_count[t] > tiny_max_seqs() && !jobs[k].bf3)
          tiled_wgs += jobs[k].p.n_tiles * ((S_k + 63) / 64);
      }
    for (int k = 0; k < n; ++k) {
      done[k] = t >= jobs[k].b->Tmax;
      if (done[k]) continue;
      FwdJob& j = jobs[k];
      j.p.t = t;
      kind[k] = step_kind(j, j.b->step_count_host[t], mid_blocks, alone, tiled_wgs);
      any_tiled = any_tiled || (kind[k] & 3) == 1 || (kind[k] & 3) == 2;
    }
    // Step chain (gru_step_chain_kernel): when the requests that run an fp32 LDS-tiled step now go
    // on doing so together for at least chain_min_steps steps, those steps are ONE launch, queued
    // below; the requests are skipped by the per-step launches until the chain's last step.
    const ChainPlan chain = plan_chain(jobs, n, t, kind, done, js, tiled_wgs);
    for (int k = 0; k < n; ++k) {
      if (done[k]) continue;
      FwdJob& j = jobs[k];
      const int S_t = j.b->step_count_host[t];
      if (j.tail_lo >= 0 && t >= j.tail_lo) {      // the resident kernel does this step
        if (t == j.tail_lo) {
          if (any_tiled) fork(k);                  // (a), as for a small-batch step below
          for (size_t c = 0; c < plan[k].step.size(); ++c)
            if (plan[k].ev[c] != nullptr) {        // its rows' projection chunks, all of them
              (void)hipStreamWaitEvent(js[k], plan[k].ev[c], 0);
              event_put(plan[k].ev[c], false);
              plan[k].ev[c] = nullptr;
            }
          launch_fwd_tail(j, js[k]);
          ++launches;
        }
        j.off += S_t;
        done[k] = true;
        continue;
      }
      const bool small = (kind[k] & 3) == 0 || (kind[k] & 3) == 3;
      if (small && any_tiled) fork(k);                                       // (a)
      hipStream_t stream = js[k];
      if (j.b->step_events_host != nullptr && j.b->step_events_host[t] != nullptr)
        (void)hipStreamWaitEvent(stream, static_cast<hipEvent_t>(const_cast<void*>(j.b->step_events_host[t])), 0);
      if (t == j.t_mid) {
        // A chain that is small-batch from its first step (a training batch) with a side stream:
        // only the projection of the first steps stands in front of the chain; the rest is cut
        // into chunks of time steps that run on the side stream BESIDE the chain, step t waiting
        // (event) for the chunk that holds its rows.  A GEMM of img_dim = 2048 rows is as long as
        // the whole 80-step chain it used to precede.
        const bool chunked = j.side_stream != nullptr && j.side_stream != stream && t == 0 &&
                             j.t_mid == 0 && !j.p.gx_per_seq && xproj_chunk_rows() > 0 && j.sum_T >= 4 * xproj_chunk_rows();
        // the hoisted projection reads the inputs of the steps it covers: wait for their uploads
        // (cmhse_pull_steps chunks still in flight) — all remaining steps for the one-launch form,
        // chunk by chunk for the chunked one (a host-fed training step: the chain starts as soon as
        // the first steps' rows have crossed PCIe, the rest arrives under it)
        auto wait_uploads = [&](hipStream_t st, int q0, int q1) {
          if (j.b->step_events_host == nullptr) return;
          for (int q = q0; q < q1; ++q)
            if (j.b->step_events_host[q] != nullptr)
              (void)hipStreamWaitEvent(st, static_cast<hipEvent_t>(const_cast<void*>(j.b->step_events_host[q])), 0);
        };
        if (early_xproj[k] != nullptr) {          // launched on the side stream before the first step
          if (early_xproj[k] != reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(1))) {
            (void)hipStreamWaitEvent(stream, early_xproj[k], 0);
            event_put(early_xproj[k], false);
          }
          early_xproj[k] = nullptr;
        } else if (!chunked) {
          wait_uploads(stream, t + 1, j.b->Tmax);
          launch_xproj(j, stream);
        } else {
          XprojPlan& xp = plan[k];
          int64_t row = 0, chunk_begin = 0;
          int t0 = 0;
          for (int q = 0; q < j.b->Tmax; ++q) {
            row += j.b->step_count_host[q];
            const int64_t want = (t0 == 0) ? kXprojFirstRows : xproj_chunk_rows();
            if (row - chunk_begin >= want || q == j.b->Tmax - 1) {
              if (t0 == 0) {
                wait_uploads(stream, 1, q + 1);
                launch_xproj(j, stream, 0, row);
                stream_after(j.side_stream, stream);     // fork: the inputs are ready
              } else {
                wait_uploads(j.side_stream, t0, q + 1);
                launch_xproj(j, j.side_stream, chunk_begin, row);
                hipEvent_t ev = event_get(false);
                if (ev != nullptr && hipEventRecord(ev, j.side_stream) == hipSuccess) {
                  xp.step.push_back(t0);
                  xp.ev.push_back(ev);
                } else {                                  // no event: order the whole stream instead
                  event_put(ev, false);
                  (void)hipGetLastError();
                  (void)hipStreamSynchronize(j.side_stream);
                }
              }
              chunk_begin = row;
              t0 = q + 1;
            }
          }
        }
      }
      for (size_t c = 0; c < plan[k].step.size(); ++c)
        if (plan[k].step[c] == t) {     // the chunk that holds step t's rows
          (void)hipStreamWaitEvent(stream, plan[k].ev[c], 0);
          event_put(plan[k].ev[c], false);
          plan[k].ev[c] = nullptr;
        }
      j.p.S_t = S_t;
      j.p.off_prev = j.off - (t > 0 ? j.b->step_count_host[t - 1] : 0);
      j.p.off_cur = j.off;
      j.off += S_t;
    }
    if (chain.end > t) {
      launch_chain(jobs, n, t, chain, timer);
      for (int k = 0; k < n; ++k)
        if (chain.in_chain[k]) done[k] = true;
      ++launches;
    }
    for (int k = 0; k < n; ++k)
      if (!done[k] && t < jobs[k].chain_until) done[k] = true;   // inside a chain launch queued at an earlier step
    for (int k = 0; k < n; ++k) {
      if (done[k]) continue;
      GruStepGroup g;
      g.n = 0;
      unsigned grid = 0;
      hipStream_t stream = js[k];
      for (int m = k; m < n; ++m) {   // same kernel, same stream: one launch
        if (done[m] || kind[m] != kind[k] || js[m] != stream) continue;
        g.j[g.n] = jobs[m].p;
        g.start[g.n] = grid;
        grid += step_grid(jobs[m], kind[m], jobs[m].p.S_t);
        ++g.n;
        done[m] = true;
      }
      for (int m = g.n; m < kMaxJobs; ++m) g.start[m] = 0xffffffffu;
      const bool stamp = timer != nullptr && ((kind[k] & 3) == 1 || (kind[k] & 3) == 2);
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (stamp) {
        e0 = event_get(true);
        e1 = e0 ? event_get(true) : nullptr;
        if (e0 && !e1) {   // no pair: release the first, time nothing
          event_put(e0, true);
          e0 = nullptr;
        }
        if (e0 && e1) (void)hipEventRecord(e0, stream);
      }
      launch_group(g, kind[k], grid, stream);
      if (stamp && e0 && e1) {
        (void)hipEventRecord(e1, stream);
        timer->tiled_events.push_back(e0);
        timer->tiled_events.push_back(e1);
        for (int q = 0; q < g.n; ++q) {
          const double I = g.j[q].I, H = g.j[q].H;
          timer->tiled_flops += g.j[q].S_t * (2.0 * 3.0 * H * (I + H) + 14.0 * H);
          // x_t in, h_{t-1} in, h_t out per sequence; the weights once per launch
          timer->tiled_bytes += g.j[q].S_t * 4.0 * (I + 2.0 * H) + 12.0 * H * (I + H);
        }
      }
      ++launches;
    }
    // (b) A request whose chain ends here while others go on: the others continue on the side
    // stream and its attention pass starts now on the caller's stream, so the two overlap (the
    // tail is a few sequences per step: latency-bound launches on an otherwise idle chip).
    for (int k = 0; k < n; ++k) {   // a chain on its own stream pools as soon as it ends
      FwdJob& j = jobs[k];
      if (!j.pooled && js[k] != main_stream && js[k] == j.own_stream && j.pool_mode == CMHSE_POOL_ATTN &&
          t == j.b->Tmax - 1 && launch_attention(j, js[k], j.sum_T, true) == CMHSE_OK) {
        j.pooled = true;
        mark_ready(j, js[k]);
      }
    }
    for (int k = 0; k < n; ++k) {
      FwdJob& j = jobs[k];
      if (j.pooled || side == nullptr || js[k] != main_stream || j.pool_mode != CMHSE_POOL_ATTN ||
          t != j.b->Tmax - 1 || t == Tmax - 1)
        continue;
      bool others = false;
      for (int m = 0; m < n; ++m)
        if (m != k && t < jobs[m].b->Tmax - 1) {
          fork(m);
          others = true;
        }
      if (!others) continue;
      if (launch_attention(j, main_stream, j.sum_T, true) == CMHSE_OK) {
        j.pooled = true;
        mark_ready(j, main_stream);      // this request's rows can leave while the others' tail runs
      }
      // ... and behind it the attention projection of what the OTHER attention-pooled chains have
      // produced so far: only the rows of their remaining tail steps are left for after the tail
      for (int m = 0; m < n; ++m) {
        FwdJob& o = jobs[m];
        if (m == k || o.pool_mode != CMHSE_POOL_ATTN || o.pooled || t >= o.b->Tmax - 1 || o.off <= 0 ||
            o.own_stream != nullptr)
          continue;
        if (js[m] != main_stream) stream_after(main_stream, js[m]);   // its steps <= t
        (void)launch_attention(o, main_stream, o.off, false);
      }
    }
  }
  // an early projection nobody waited for (defensive: step t_mid always does) still owns an event, and
  // the side stream then holds work the caller's stream has not been ordered behind
  bool side_pending = false;
  for (int k = 0; k < kMaxJobs; ++k) {
    if (early_xproj[k] == nullptr) continue;
    if (early_xproj[k] != reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(1))) event_put(early_xproj[k], false);
    early_xproj[k] = nullptr;
    side_pending = true;
  }
  if (forked || side_pending) stream_after(main_stream, side);   // rejoin: what follows is ordered on the caller's stream
  return launches;
}

// Attention energies of packed rows [job.att_rows_done, row_end) and, with `pool`, the pooling
// pass over all rows (row_end must then be sum_T).
int launch_attention(FwdJob& job, hipStream_t stream, int64_t row_end, bool pool) {
  const cmhse_seq_batch* b = job.b;
  const cmhse_gru_weights* w = job.w;
  const GruWs& L = job.L;
  char* wsb = job.wsb;
  const int64_t sum_T = job.sum_T;
  const int att_tiles = (b->H + kAttBN - 1) / kAttBN;
  // tile height of the projection: 64 rows (128 measured equal on the full split in round 2: 285.0
  // against 285.8 ms per pass; again in round 6, beside the text tower's tail on the side stream:
  // 271.0-273.4 with 64 rows against 273.3-275.1 with 128, three runs each on one box)
  constexpr int msub = 1;
  float* e_part = reinterpret_cast<float*>(wsb + L.e_part);
  AttnEnergyParams ep;
  ep.hs_s = nullptr;
  ep.hs = job.p.hs;
  ep.w_lin = w->w_lin;
  ep.b_lin = w->b_lin;
  ep.w_att = w->w_att;
  ep.e_part = e_part;
  ep.v = job.save ? reinterpret_cast<float*>(wsb + L.v) : nullptr;
  ep.rows = sum_T;
  ep.row_begin = job.att_rows_done;
  ep.row_end = row_end;
  ep.H = b->H;
  ep.n_tiles = att_tiles;
  // (bf16x3 or fp32 for the projection is a function of the PLAN's packed rows — the whole split's when
  // the batch is a share of one — never of the share's own: the same bits in any share)
  int64_t plan_rows = 0;
  for (int q = 0; q < b->Tmax; ++q) plan_rows += job.kind_count[q];
  const bool att_bf3 = job.bf3 && plan_rows > tiny_max_seqs();
  const int att_bm = att_bf3 ? 128 : 64 * msub;
  if (row_end < job.att_rows_done) row_end = job.att_rows_done;
  ep.row_end = row_end;
  const int64_t m_tiles = (row_end - job.att_rows_done + att_bm - 1) / att_bm;
  job.att_rows_done = row_end;
  if (m_tiles * att_tiles > 0x7fffffffLL) return CMHSE_ERR_UNSUPPORTED;
  const unsigned att_grid = static_cast<unsigned>(m_tiles * att_tiles);
  ep.w_lin_s = nullptr;
  if (att_grid == 0) {
    // nothing left to project (every row was served by an earlier partial launch)
  } else if (att_bf3) {
    float* wlin_s = reinterpret_cast<float*>(wsb + L.wlin_s);
    if (ep.row_begin == 0) launch_split(w->w_lin, wlin_s, b->H, b->H, stream);
    ep.w_lin_s = wlin_s;
    ep.hs_s = job.p.hs_s;
    const size_t att_smem = TileSmem<128, kAttBN>::kBytes;
    // rows the tiled bf16x3 steps produced exist in pre-split form (no conversion in the loop);
    // the rows of the small-batch steps behind them are fp32 only
    const int64_t lo = ep.row_begin, hi = ep.row_end;
    const int64_t cut = job.rows_split < lo ? lo : (job.rows_split > hi ? hi : job.rows_split);
    if (cut > lo) {
      ep.row_begin = lo;
      ep.row_end = cut;
      const unsigned g1 = static_cast<unsigned>(((cut - lo + 127) / 128) * att_tiles);
      // three ring stages of 384 rows x 64 B: more than the 64 KiB a launch may ask for by default
      constexpr size_t ring_smem = RingSmem<128, kAttBN>::kBytes;
      // (per device; a host-side call, no synchronisation)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_energy_kernel<true, 2, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ring_smem)) != hipSuccess)
        return CMHSE_ERR_LAUNCH;
      hipLaunchKernelGGL((attn_energy_kernel<true, 2, true, true>), dim3(g1), dim3(kThreads), ring_smem, stream, ep);
    }
    if (hi > cut) {
      ep.row_begin = cut;
      ep.row_end = hi;
      const unsigned g2 = static_cast<unsigned>(((hi - cut + 127) / 128) * att_tiles);
      hipLaunchKernelGGL((attn_energy_kernel<true, 2, true, false>), dim3(g2), dim3(kThreads), att_smem, stream, ep);
    }
  } else {
    const size_t att_smem = TileSmem<64, kAttBN>::kBytes;
    if (job.vec)
      hipLaunchKernelGGL((attn_energy_kernel<true, 1, false>), dim3(att_grid), dim3(kThreads), att_smem, stream, ep);
    else
      hipLaunchKernelGGL((attn_energy_kernel<false, 1, false>), dim3(att_grid), dim3(kThreads), att_smem, stream, ep);
  }
  if (!pool) return CMHSE_OK;
  AttnPoolParams pp;
  pp.hs = job.p.hs;
  pp.e_part = e_part;
  pp.lens = b->lens;
  pp.out_row = b->out_row;
  pp.step_off = b->step_off;
  pp.out = job.out;
  pp.rows = sum_T;
  pp.H = b->H;
  pp.n_tiles = att_tiles;
  hipLaunchKernelGGL(attn_pool_kernel, dim3(b->S), dim3(kThreads), 0, stream, pp);
  return CMHSE_OK;
}

}  // namespace

extern "C" int cmhse_async_status(int32_t clear) { return resident_status(clear != 0); }

namespace cmhse {
// Self-test of the grid barrier (grid_sync.hpp): every workgroup arrives `rounds` times, but each
// barrier expects `missing` more arrivals than there are workgroups.  missing == 0: the ordinary
// path; missing > 0: nobody ever completes the barrier — the wall-time bound must end the kernel,
// raise the abort word and the device's status word.
__global__ __launch_bounds__(64) void grid_sync_selftest_kernel(GridSync g, unsigned* out, int missing, int rounds) {
  unsigned arrivals = 0;
  for (int r = 0; r < rounds; ++r) {
    __builtin_amdgcn_s_waitcnt(0);
    arrivals += gridDim.x + static_cast<unsigned>(missing);
    if (!grid_sync_wait(g, arrivals)) {
      if (threadIdx.x == 0) atomicAdd(out + 1, 1u);      // workgroups that left through the abort path
      return;
    }
  }
  if (threadIdx.x == 0) atomicAdd(out, 1u);              // workgroups that passed every barrier
}
}  // namespace cmhse

extern "C" int cmhse_selftest_grid_sync(void* workspace, int32_t workgroups, int32_t missing,
                                        int32_t rounds, void* stream_) {
  if (!workspace || workgroups <= 0 || workgroups > 1024 || missing < 0 || rounds <= 0) return CMHSE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return CMHSE_ERR_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (hipMemsetAsync(workspace, 0, 256, stream) != hipSuccess) return CMHSE_ERR_LAUNCH;
  unsigned* words = static_cast<unsigned*>(workspace);
  const GridSync g = make_grid_sync(words, words + 1);
  hipLaunchKernelGGL(grid_sync_selftest_kernel, dim3(static_cast<unsigned>(workgroups)), dim3(64), 0, stream,
                     g, words + 2, missing, rounds);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_gru_pool_fwd_multi(const cmhse_gru_job* reqs, int32_t n_jobs, void* stream_) {
  if (!reqs || n_jobs <= 0 || n_jobs > kMaxJobs) return CMHSE_ERR_ARG;
  if (resident_check() != CMHSE_OK) return CMHSE_ERR_TIMEOUT;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FwdJob jobs[kMaxJobs];
  for (int k = 0; k < n_jobs; ++k) {
    const int rc = prepare_job(reqs[k].seqs, reqs[k].weights, reqs[k].pool_mode, reqs[k].out,
                               reqs[k].workspace, reqs[k].workspace_bytes, stream, &jobs[k]);
    if (rc != CMHSE_OK) return rc;
    jobs[k].tail_stream = static_cast<hipStream_t>(reqs[k].tail_stream);
    jobs[k].own_stream = static_cast<hipStream_t>(reqs[k].stream);
    jobs[k].side_stream = static_cast<hipStream_t>(reqs[k].side_stream);
    jobs[k].pooled = false;
    jobs[k].att_rows_done = 0;
    jobs[k].ready_event = static_cast<hipEvent_t>(reqs[k].out_ready_event);
    jobs[k].ready_marked = false;
  }
  // the first job's step_timer (if any) spans the step launches of the whole group
  Timer* timer = static_cast<Timer*>(jobs[0].b->step_timer);
  if (timer) (void)hipEventRecord(timer->start, stream);
  const int launches = launch_steps(jobs, n_jobs, stream, timer);
  if (timer) {
    (void)hipEventRecord(timer->stop, stream);
    timer->launches = launches;
  }
  // join the requests' own streams (each once) back into the caller's
  for (int k = 0; k < n_jobs; ++k) {
    hipStream_t own = jobs[k].own_stream;
    if (own == nullptr || own == stream || (reqs[k].pool_mode & CMHSE_NO_JOIN)) continue;
    bool seen = false;
    for (int m = 0; m < k; ++m) seen = seen || jobs[m].own_stream == own;
    if (!seen) stream_after(stream, own);
  }
  for (int k = 0; k < n_jobs; ++k) {
    if (jobs[k].pool_mode != CMHSE_POOL_ATTN || jobs[k].pooled) continue;
    const int rc = launch_attention(jobs[k], stream, jobs[k].sum_T, true);
    if (rc != CMHSE_OK) return rc;
  }
  for (int k = 0; k < n_jobs; ++k) mark_ready(jobs[k], stream);     // (whatever was not final earlier)
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_gru_pool_fwd(const cmhse_seq_batch* b, const cmhse_gru_weights* w,
                                  int32_t pool_mode, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
  cmhse_gru_job req;
  req.seqs = b;
  req.weights = w;
  req.pool_mode = pool_mode;
  req.out = out;
  req.workspace = workspace;
  req.workspace_bytes = workspace_bytes;
  req.tail_stream = nullptr;
  req.stream = nullptr;
  req.side_stream = nullptr;
  req.out_ready_event = nullptr;
  return cmhse_gru_pool_fwd_multi(&req, 1, stream_);
}

extern "C" int cmhse_pull_steps(const uint64_t* src_rows_pinned, const uint64_t* dst_rows,
                                const int32_t* lens, int32_t n_active, int32_t row_floats,
                                int32_t t0, int32_t t1, void* stream_) {
  if (!src_rows_pinned || !dst_rows || !lens || n_active < 0 || row_floats <= 0 || t0 < 0 || t1 < t0)
    return CMHSE_ERR_ARG;
  if (n_active == 0 || t1 == t0) return CMHSE_OK;
  PullParams p;
  p.src_rows = src_rows_pinned;
  p.dst_rows = dst_rows;
  p.lens = lens;
  p.n_active = n_active;
  p.row_floats = row_floats;
  p.t0 = t0;
  p.t1 = t1;
  // "pull_waves" (32) single-wave workgroups, 8 x 16 B in flight per lane: 57 GB/s alone (the PCIe Gen5
  // x16 rate).  Since round 6 a pull wave needs 48 registers and no scratch: it fits beside the two
  // 232-register workgroups of the step chain on a CU instead of displacing one (gru_rows.hpp).
  int cap = tunables().pull_waves.load(std::memory_order_relaxed);
  cap = cap < 1 ? 1 : (cap > 1024 ? 1024 : cap);
  constexpr int thr = 64;
  const unsigned grid = static_cast<unsigned>(n_active < cap ? n_active : cap);
  hipLaunchKernelGGL(pull_steps_kernel, dim3(grid), dim3(thr), 0,
                     static_cast<hipStream_t>(stream_), p);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_push_rows(const void* src, void* dst_pinned, size_t bytes, int32_t workgroups,
                               int32_t waves, void* stream_) {
  if ((!src || !dst_pinned) && bytes) return CMHSE_ERR_ARG;
  if (workgroups < 0 || workgroups > 1024 || waves < 0 || waves > 4) return CMHSE_ERR_ARG;
  if (bytes == 0) return CMHSE_OK;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst_pinned)) & 15u)
    return CMHSE_ERR_ARG;      // (rows of fp32 matrices out of the allocator: always 16-byte aligned)
  const size_t n16 = bytes >> 4;
  const int tail = static_cast<int>(bytes & 15u);
  // workgroups x waves single-wave workgroups (a wave of <= 48 registers fits beside the step chain's
  // workgroups; wider workgroups would only tie four of them to one CU)
  unsigned grid = static_cast<unsigned>(workgroups ? workgroups : 8) * static_cast<unsigned>(waves ? waves : 1);
  const size_t need = (n16 + 511) / 512;
  if (need < grid) grid = static_cast<unsigned>(need ? need : 1);
  hipLaunchKernelGGL(push_bytes_kernel, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream_),
                     static_cast<const float4*>(src), static_cast<float4*>(dst_pinned), n16,
                     static_cast<const unsigned char*>(src) + (n16 << 4),
                     static_cast<unsigned char*>(dst_pinned) + (n16 << 4), tail);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_rows_differ(const void* a, const void* b, size_t bytes, int32_t* flag, void* stream_) {
  if (!flag || ((!a || !b) && bytes)) return CMHSE_ERR_ARG;
  if (bytes == 0) return CMHSE_OK;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u) return CMHSE_ERR_ARG;
  const size_t n16 = bytes >> 4;
  const int tail = static_cast<int>(bytes & 15u);
  // nothing latency-bound runs beside this check: enough waves to keep the link full with four
  // 16-byte reads in flight per lane
  unsigned grid = 64;
  const size_t need = (n16 + 255) / 256;
  if (need < grid) grid = static_cast<unsigned>(need ? need : 1);
  hipLaunchKernelGGL(rows_differ_kernel, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream_),
                     static_cast<const uint4*>(a), static_cast<const uint4*>(b), n16,
                     static_cast<const unsigned char*>(a) + (n16 << 4),
                     static_cast<const unsigned char*>(b) + (n16 << 4), tail, flag);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_pad_rows(const void* src, const int64_t* first_row, const int32_t* lens,
                              int32_t S, int32_t Tmax, int32_t row_bytes, void* dst, void* stream_) {
  if (!src || !first_row || !lens || !dst || S < 0 || Tmax < 0 || row_bytes <= 0 || row_bytes % 4 != 0)
    return CMHSE_ERR_ARG;
  if (S == 0 || Tmax == 0) return CMHSE_OK;
  const bool wide = row_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(dst) % 16 == 0;
  PadRowsParams q;
  q.src = static_cast<const char*>(src);
  q.first_row = first_row;
  q.lens = lens;
  q.dst = static_cast<char*>(dst);
  q.Tmax = Tmax;
  q.words_per_row = row_bytes / (wide ? 16 : 4);
  q.words = static_cast<int64_t>(S) * Tmax * q.words_per_row;
  const int64_t blocks = (q.words + kThreads - 1) / kThreads;
  const unsigned grid = static_cast<unsigned>(blocks < 8192 ? blocks : 8192);
  if (wide)
    hipLaunchKernelGGL(pad_rows_kernel<uint4>, dim3(grid), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream_), q);
  else
    hipLaunchKernelGGL(pad_rows_kernel<uint32_t>, dim3(grid), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream_), q);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_l2norm_rows(const float* x, float* y, int32_t rows, int32_t cols, int64_t ld,
                                 void* stream_) {
  if (!x || !y || rows < 0 || cols <= 0 || ld < cols) return CMHSE_ERR_ARG;
  if (rows == 0) return CMHSE_OK;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(rows), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream_), x, y, cols, ld);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_gather_rows(const float* table, const int64_t* ids, int64_t n, int32_t cols,
                                 int32_t vocab, float* out, void* stream_) {
  if (!table || !ids || !out || n < 0 || cols <= 0 || vocab <= 0) return CMHSE_ERR_ARG;
  if (n == 0) return CMHSE_OK;
  if (n > 0x7fffffffLL) return CMHSE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(static_cast<unsigned>(n)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream_), table,
                     reinterpret_cast<const long long*>(ids), out, cols, vocab);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

namespace {
struct TuneEntry { const char* name; std::atomic<int>* v; };
int tune_in(Tunables& t, const char* name, int32_t value, int32_t* old_value);
}  // namespace

extern "C" int cmhse_tune(const char* name, int32_t value, int32_t* old_value) {
  if (!name) return CMHSE_ERR_ARG;
  return tune_in(global_tunables(), name, value, old_value);
}

extern "C" void* cmhse_ctx_create(void) {
  Tunables* c = new (std::nothrow) Tunables;
  if (c == nullptr) return nullptr;
  // a copy of the process defaults as they are now, member by member (the one table, gru_ws.hpp)
  Tunables& g = global_tunables();
#define CMHSE_COPY_(name, dflt) c->name.store(g.name.load(std::memory_order_relaxed), std::memory_order_relaxed);
  CMHSE_TUNABLES(CMHSE_COPY_)
#undef CMHSE_COPY_
  return c;
}

extern "C" void cmhse_ctx_destroy(void* ctx) {
  Tunables* c = static_cast<Tunables*>(ctx);
  if (c != nullptr && tl_ctx == c) tl_ctx = nullptr;
  delete c;
}

extern "C" int cmhse_ctx_tune(void* ctx, const char* name, int32_t value, int32_t* old_value) {
  if (!ctx || !name) return CMHSE_ERR_ARG;
  return tune_in(*static_cast<Tunables*>(ctx), name, value, old_value);
}

extern "C" void* cmhse_ctx_enter(void* ctx) {
  Tunables* prev = tl_ctx;
  tl_ctx = static_cast<Tunables*>(ctx);
  return prev;
}

namespace {
int tune_in(Tunables& t, const char* name, int32_t value, int32_t* old_value) {
  if (strcmp(name, "multi_step_off") == 0) {       // not a crossover: the current DEVICE's fallback flag
    const int dev = event_device();
    if (dev < 0) return CMHSE_ERR_ARG;
    const int old = (value >= 0) ? g_multi_off[dev].exchange(value != 0 ? 1 : 0) : g_multi_off[dev].load();
    if (old_value) *old_value = old;
    return CMHSE_OK;
  }
  TuneEntry table[] = {
#define CMHSE_ENTRY_(name, dflt) {#name, &t.name},
      CMHSE_TUNABLES(CMHSE_ENTRY_)
#undef CMHSE_ENTRY_
  };
  for (auto& e : table)
    if (strcmp(name, e.name) == 0) {
      const int old = (value >= 0) ? e.v->exchange(value) : e.v->load();
      if (old_value) *old_value = old;
      if (value > 0 && (e.v == &t.chain_min_steps || e.v == &t.fwd_tail_min_steps || e.v == &t.bwd_tail_min_steps)) {
        const int dev = event_device();      // an explicit re-enable after an acknowledged timeout
        if (dev >= 0) g_multi_off[dev].store(0, std::memory_order_relaxed);
      }
      return CMHSE_OK;
    }
  return CMHSE_ERR_ARG;
}
}  // namespace

extern "C" void* cmhse_timer_create(void) {
  Timer* t = new (std::nothrow) Timer;
  if (!t) return nullptr;
  t->launches = 0;
  t->tiled_flops = 0.0;
  t->tiled_bytes = 0.0;
  t->start = event_get(true);
  t->stop = t->start ? event_get(true) : nullptr;
  if (!t->start || !t->stop) {
    event_put(t->start, true);
    delete t;
    return nullptr;
  }
  return t;
}

extern "C" void cmhse_timer_destroy(void* timer) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t) return;
  event_put(t->start, true);
  event_put(t->stop, true);
  for (hipEvent_t e : t->tiled_events) event_put(e, true);
  delete t;
}

extern "C" int cmhse_timer_elapsed_ms(void* timer, float* ms_host) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t || !ms_host) return CMHSE_ERR_ARG;
  if (hipEventSynchronize(t->stop) != hipSuccess) return CMHSE_ERR_LAUNCH;
  if (hipEventElapsedTime(ms_host, t->start, t->stop) != hipSuccess) return CMHSE_ERR_LAUNCH;
  return CMHSE_OK;
}

#ifdef TILE_TRACE_BUILD
extern "C" int cmhse_debug_set_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(cmhse::g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int cmhse_timer_tiled(void* timer, float* ms_host, double* flops_host,
                                 double* bytes_host, int32_t* launches_host) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t || !ms_host || !flops_host || !bytes_host || !launches_host) return CMHSE_ERR_ARG;
  if (hipEventSynchronize(t->stop) != hipSuccess) return CMHSE_ERR_LAUNCH;
  double total = 0.0;
  for (size_t i = 0; i + 1 < t->tiled_events.size(); i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t->tiled_events[i], t->tiled_events[i + 1]) != hipSuccess)
      return CMHSE_ERR_LAUNCH;
    total += ms;
  }
  *ms_host = static_cast<float>(total);
  *flops_host = t->tiled_flops;
  *bytes_host = t->tiled_bytes;
  *launches_host = static_cast<int32_t>(t->tiled_events.size() / 2);
  return CMHSE_OK;
}

extern "C" int32_t cmhse_timer_launches(void* timer) {
  Timer* t = static_cast<Timer*>(timer);
  return t ? t->launches : 0;
}
