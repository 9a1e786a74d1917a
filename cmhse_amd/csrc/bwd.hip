// bwd.hip — backward pass of the hot path: what `loss.backward()`
// (/root/reference/model.py:367) computes through layers.{Seq2Seq,Attention,Maxout}.forward,
// F.normalize and ContrastiveLoss.forward, as hand-written gfx950 kernels.
//
//   cmhse_gru_pool_bwd      pooling backward -> dpool[sumT,H]; BPTT over the packed steps in reverse
//                           (dh_{t} = dgates_{t+1} . W_hh + carry and the gate derivatives: one
//                           launch per step, or two with K split over the grid at training-batch
//                           sizes); weight gradients straight from the packed rows in chunks of
//                           time steps beside the chain (tn_rows.hpp); optional d(input) / d(h0) /
//                           d(embedding table).
//   cmhse_l2norm_rows_bwd   F.normalize backward.
//   cmhse_contrastive_bwd   d loss / d im, d loss / d s from the stored score matrix.
//
// Determinism: every reduction runs in a fixed order except the sums into the embedding table and
// into a time-constant decoder input (float atomics, like torch's CUDA embedding backward).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/cmhse_hip.h"
#include "gru_ws.hpp"
#include "grid_sync.hpp"
#include "nt_core.hpp"
#include "tn_core.hpp"
#include "tn_rows.hpp"
#include "step_loss.hpp"

#include "bwd_pool_kernels.hpp"
#include "bwd_step_kernels.hpp"
#include "bwd_loss_kernels.hpp"

namespace cmhse {

// ---------------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------------
static void launch_transpose(const float* in, float* out, int R, int C, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(kThreads), 0, st,
                     in, out, R, C);
}

static void launch_colsum(const float* in, const float* w, float* out, float* scratch,
                          int64_t rows, int cols, int64_t ld, hipStream_t st) {
  const int slabs = static_cast<int>((rows + kColsumRows - 1) / kColsumRows);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 63) / 64, slabs), dim3(kThreads), 0, st,
                     in, w, scratch, rows, cols, ld);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + kThreads - 1) / kThreads), dim3(kThreads),
                     0, st, scratch, out, slabs, cols);
}

static void launch_tn(const float* a, int64_t lda, const float* b, int64_t ldb,
                      const uint64_t* b_addr, float* c, int64_t ldc, int M, int N, int64_t K,
                      const float* scale, bool vec, hipStream_t st) {
  TnParams q;
  q.a = a; q.lda = lda; q.b = b; q.ldb = ldb; q.b_addr = b_addr; q.c = c; q.ldc = ldc;
  q.M = M; q.N = N; q.K = K; q.n_tiles = (N + 127) / 128; q.scale = scale;
  q.blk_off = nullptr; q.blk_stride = 0;
  const unsigned grid = static_cast<unsigned>(q.n_tiles) * ((M + 127) / 128);
  const size_t smem = TnSmem<128, 128>::kBytes;
  if (vec)
    hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(grid), dim3(kThreads), smem, st, q);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(grid), dim3(kThreads), smem, st, q);
}

// Split-K for products with few output tiles and a long K: the workgroups of one tile accumulate
// with float atomics.  `split` = 1: C is a dense [M, ldc = N] matrix this call may zero itself;
// `split` = 2: the caller's output rows are already zero (rows through c_addr, or mode 2 targets).
static void launch_nt_out(const float* a, int64_t lda, const float* b, int64_t ldb, float* c,
                          int64_t ldc, const uint64_t* c_addr, int M, int N, int K, int mode,
                          hipStream_t st, int split = 0) {
  NtOutParams q;
  q.a = a; q.lda = lda; q.b = b; q.ldb = ldb; q.c = c; q.ldc = ldc; q.c_addr = c_addr;
  q.M = M; q.N = N; q.K = K; q.mode = mode; q.n_tiles = (N + 127) / 128;
  q.k_seg = 0;
  const unsigned grid = static_cast<unsigned>(q.n_tiles) * ((M + 127) / 128);
  unsigned splits = 1;
  const bool own_zero = split == 1 && mode == 0 && c_addr == nullptr && ldc == N;
  const bool pre_zero = split == 2 && (mode == 0 || mode == 2);
  if ((own_zero || pre_zero) && grid < 512 && K >= 512) {
    splits = (768 + grid - 1) / grid;
    const unsigned max_splits = static_cast<unsigned>(K / 256);   // >= 16 chunks per segment
    if (splits > max_splits) splits = max_splits;
    if (splits > 1) {
      q.k_seg = ((K + static_cast<int>(splits) - 1) / static_cast<int>(splits) + 15) / 16 * 16;
      splits = static_cast<unsigned>((K + q.k_seg - 1) / q.k_seg);
      q.mode = 2;
      if (own_zero) (void)hipMemsetAsync(c, 0, static_cast<size_t>(M) * N * sizeof(float), st);
    }
  }
  const size_t smem = TileSmem<128, 128>::kBytes;
  const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0);
  if (vec)
    hipLaunchKernelGGL(gemm_nt_out_kernel<true>, dim3(grid, splits), dim3(kThreads), smem, st, q);
  else
    hipLaunchKernelGGL(gemm_nt_out_kernel<false>, dim3(grid, splits), dim3(kThreads), smem, st, q);
}

// A product with few output rows and a long K whose rows are written exactly once (d(input) of the
// level-2 encoders: ~120 rows x K = 3H): one 128-row tile per 128 columns would walk all of K on a
// handful of CUs (200 us for 0.75 GFLOP).  K is cut into segments, every segment's partial goes
// to scratch, and a second small launch adds the segments in order — no atomics, reproducible.
constexpr int kDetSplitRows = 256, kDetSplitMax = 12;
static size_t det_split_scratch_bytes(int64_t sum_T, int N) {
  const int64_t rows = sum_T < kDetSplitRows ? sum_T : kDetSplitRows;
  return static_cast<size_t>(kDetSplitMax) * rows * N * sizeof(float);
}
// rows scattered through c_addr and written once (c == nullptr), or the contiguous rows of c
// added to (the attention backward's dpool += du . W_lin of a level-2 batch: 8 tiles used to walk
// K = H on 8 CUs, 89 us on the path between the levels)
static void launch_nt_rows_once(const float* a, int64_t lda, const float* b, int64_t ldb,
                                const uint64_t* c_addr, int M, int N, int K, float* scratch,
                                hipStream_t st, float* c = nullptr, int64_t ldc = 0) {
  int splits = K / 256;
  if (splits > kDetSplitMax) splits = kDetSplitMax;
  if (M > kDetSplitRows || splits < 2) {
    if (c != nullptr) launch_nt_out(a, lda, b, ldb, c, ldc, nullptr, M, N, K, 1, st);
    else launch_nt_out(a, lda, b, ldb, nullptr, 0, c_addr, M, N, K, 0, st);
    return;
  }
  NtOutParams q;
  q.a = a; q.lda = lda; q.b = b; q.ldb = ldb; q.c = scratch; q.ldc = N; q.c_addr = nullptr;
  q.M = M; q.N = N; q.K = K; q.mode = 3; q.n_tiles = (N + 127) / 128;
  q.k_seg = ((K + splits - 1) / splits + 15) / 16 * 16;
  splits = (K + q.k_seg - 1) / q.k_seg;
  const unsigned grid = static_cast<unsigned>(q.n_tiles) * ((M + 127) / 128);
  const size_t smem = TileSmem<128, 128>::kBytes;
  const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0);
  if (vec)
    hipLaunchKernelGGL(gemm_nt_out_kernel<true>, dim3(grid, splits), dim3(kThreads), smem, st, q);
  else
    hipLaunchKernelGGL(gemm_nt_out_kernel<false>, dim3(grid, splits), dim3(kThreads), smem, st, q);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(M), dim3(kThreads), 0, st, scratch, c_addr, c, ldc,
                     c != nullptr ? 1 : 0, splits, M, N);
}

// One launch of gemm_tn_rows_kernel: up to kTnRowsMaxProblems products over the packed rows
// [p0, p1), each C (and bias) stored (accumulate = 0) or added to (1).
struct TnRowsLaunch {
  TnRowsGroup g;
  unsigned grid;
  TnRowsLaunch() : grid(0) { g.n = 0; }
  void add(const float* a, int64_t lda, const uint64_t* b_addr, float* c, int64_t ldc, float* bias,
           int M, int N) {
    TnRowsProblem& q = g.q[g.n];
    q.a = a; q.lda = lda; q.b_addr = b_addr; q.c = c; q.ldc = ldc;
    q.bias = bias; q.M = M; q.N = N; q.n_tiles = (N + 127) / 128;
    ++g.n;
  }
  // Tile height of this launch (Tunables::tn_rows_bm; 0 = by shape): the tall tile when every
  // product's M is a whole number of them (3H = 3072 is), else the small one
  int tile_bm() const {
    const int want = tunables().tn_rows_bm.load(std::memory_order_relaxed);
    if (want == kTnRowsBmSmall || want == kTnRowsBmTall) return want;
    for (int k = 0; k < g.n; ++k)
      if (g.q[k].M % kTnRowsBmTall != 0) return kTnRowsBmSmall;
    return kTnRowsBmTall;
  }
  // `part`: scratch of `part_floats` floats for the row split's parts (nullptr = never split)
  void launch(int64_t p0, int64_t p1, bool accumulate, hipStream_t st, float* part = nullptr,
              int64_t part_floats = 0, bool beside = false) {
    if (g.n == 0 || p1 <= p0) return;
    const int bm = tile_bm();
    constexpr int resident = 3;   // workgroups a CU holds at once (either tile)
    grid = 0;
    for (int k = 0; k < g.n; ++k) {
      g.start[k] = grid;
      grid += static_cast<unsigned>(g.q[k].n_tiles) * ((g.q[k].M + bm - 1) / bm);
    }
    for (int k = g.n; k < kTnRowsMaxProblems; ++k) g.start[k] = 0xffffffffu;
    // the kernel addresses A rows with 32-bit byte offsets from the first row of the range
    int64_t max_lda = 1, per_split = 0;
    for (int k = 0; k < g.n; ++k) {
      max_lda = g.q[k].lda > max_lda ? g.q[k].lda : max_lda;
      g.part_off[k] = per_split;
      per_split += static_cast<int64_t>(g.q[k].M) * g.q[k].N + g.q[k].M;
    }
    const int64_t max_rows = (0x7fffffffLL / (max_lda * 4)) / kBK * kBK;
    const size_t smem = (bm == kTnRowsBmTall) ? TileSmem<kTnRowsBmTall, 128>::kBytes
                                              : TileSmem<kTnRowsBmSmall, 128>::kBytes;
    for (int64_t a = p0; a < p1; a += max_rows) {
      g.p0 = a;
      g.p1 = (a + max_rows < p1) ? a + max_rows : p1;
      g.accumulate = (accumulate || a > p0) ? 1 : 0;
      // Row split: the number of splits (<= 8, >= 256 rows each, parts fitting the scratch) that
      // leaves the busiest CU the least above the average, three workgroups per CU being
      // resident at once.  A chunk that is not the last one (`beside`: the chain is still running
      // when it is launched) is split only below two tiles per CU: there the parts' extra traffic
      // competes with the chain for the cache fabric both are bound by (train_emb step, ms, same
      // box, split below 256 / 512 / 1024 tiles: C3D 9.19 / 9.10 / 9.09, ICEP 10.40 / 10.31 / 10.74,
      // ICEP + reconstruction 11.30 / 11.22 / 11.62).  (The rule depends on the shapes only, never
      // on whether a side stream is used: results are bit-identical with and without one.)
      const int64_t rows = g.p1 - g.p0;
      int best = 1;
      if (part != nullptr && grid * static_cast<unsigned>(bm / 64) < (beside ? 1024u : 2048u)) {
        double best_cost = 1e30;
        for (int sp = 1; sp <= 8 && (sp == 1 || (rows / sp >= 256 && per_split * sp <= part_floats)); ++sp) {
          const double wgs = static_cast<double>(grid) * sp;
          const double per_cu = wgs / 256.0;
          const double busiest = static_cast<double>((static_cast<int64_t>(wgs) + 255) / 256);
          // time ~ the busiest CU's share of the work; co-residents below three leave latency exposed
          const double fill = per_cu >= resident ? 1.0 : (per_cu >= 2.0 ? 1.15 : 1.6);
          const double cost = busiest / sp * fill * (1.0 + 0.03 * (sp - 1));
          if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
        }
      }
      g.seg = 0;
      g.part = nullptr;
      g.part_stride = 0;
      unsigned splits = 1;
      if (best > 1) {
        g.seg = static_cast<int32_t>(((rows + best - 1) / best + kBK - 1) / kBK * kBK);
        splits = static_cast<unsigned>((rows + g.seg - 1) / g.seg);
        g.part = part;
        g.part_stride = per_split;
      }
      if (bm == kTnRowsBmTall)
        hipLaunchKernelGGL(gemm_tn_rows_kernel<3>, dim3(grid, splits), dim3(kThreads), smem, st, g);
      else
        hipLaunchKernelGGL(gemm_tn_rows_kernel<2>, dim3(grid, splits), dim3(kThreads), smem, st, g);
      if (splits > 1)
        for (int k = 0; k < g.n; ++k) {
          TnRowsReduce r;
          r.part = part; r.part_stride = per_split; r.part_off = g.part_off[k];
          r.c = g.q[k].c; r.ldc = g.q[k].ldc; r.bias = g.q[k].bias; r.M = g.q[k].M; r.N = g.q[k].N;
          r.splits = static_cast<int32_t>(splits); r.accumulate = g.accumulate;
          const int64_t total = static_cast<int64_t>(r.M) * r.N + r.M;
          const unsigned blocks = static_cast<unsigned>((total + kThreads * 4 - 1) / (kThreads * 4));
          hipLaunchKernelGGL(tn_rows_reduce_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(kThreads),
                             0, st, r);
        }
    }
  }
};

// Scratch of the two-launch BPTT step: `splits` partial [m_pad, H] tiles, splits x row tiles <= 256
// workgroups per 128 columns (rec_plan), i.e. at most 256 / (H / 128) x 32 rows of H floats.
static size_t rec_part_floats(int S, int H) {
  if (H % 4 != 0) return 0;
  const int n_tiles = (H + kRecBN - 1) / kRecBN;
  size_t rows = static_cast<size_t>(256 / n_tiles + 1 + (S + 31) / 32) * 32;
  return rows * H;
}

// Scratch of the weight-gradient row split (TnRowsLaunch): up to 4 parts of [dW_ih | db_ih | dW_hh |
// db_hh] (or 8 of the smaller dW_lin); none for a batch too short to be split at all.
static size_t wg_part_floats(int64_t sum_T, int I, int H) {
  if (sum_T < 512) return 0;
  const size_t per_split = static_cast<size_t>(3) * H * (I + H + 2);
  const size_t want = sum_T / 256 < 4 ? static_cast<size_t>(sum_T / 256) : 4;
  return per_split * want;
}

struct BwdWs {
  size_t dgx, dgh, dpool, carry, whh_t, wih_t, wlin_t, du, de, xaddr, hpaddr, dxaddr, hsaddr, p_t, zero_row,
      tail_sync, colsum, dx_part, wg_part, rec_part, total;
};

static BwdWs bwd_ws_layout(int32_t S, int64_t sum_T, int32_t I, int32_t H, int32_t mode) {
  BwdWs L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += ws_align(bytes); return o; };
  L.dgx = take(static_cast<size_t>(sum_T) * 3 * H * 4);
  L.dgh = take(static_cast<size_t>(sum_T) * 3 * H * 4);
  L.dpool = take(static_cast<size_t>(sum_T) * H * 4);
  L.carry = take(static_cast<size_t>(S) * H * 4);
  L.whh_t = take(static_cast<size_t>(3) * H * H * 4);
  L.wih_t = take(static_cast<size_t>(3) * H * I * 4);
  L.wlin_t = take(mode == CMHSE_POOL_ATTN ? static_cast<size_t>(H) * H * 4 : 0);
  L.du = take(mode == CMHSE_POOL_ATTN ? static_cast<size_t>(sum_T) * H * 4 : 0);
  L.de = take(mode == CMHSE_POOL_ATTN ? static_cast<size_t>(sum_T) * 4 : 0);
  L.xaddr = take(static_cast<size_t>(sum_T) * 8);
  L.hpaddr = take(static_cast<size_t>(sum_T) * 8);
  L.dxaddr = take(static_cast<size_t>(sum_T) * 8);
  L.hsaddr = take(mode == CMHSE_POOL_ATTN ? static_cast<size_t>(sum_T) * 8 : 0);
  L.p_t = take(static_cast<size_t>(sum_T) * 4);
  L.zero_row = take(static_cast<size_t>(H > I ? H : I) * 4);
  L.tail_sync = take(256);      // (right behind zero_row: one memset clears both)
  L.colsum = take(static_cast<size_t>((sum_T + kColsumRows - 1) / kColsumRows) * 3 * H * 4);
  L.dx_part = take(det_split_scratch_bytes(sum_T, I > H ? I : H));   // (also the attention backward's dpool product: N = H)
  L.wg_part = take(wg_part_floats(sum_T, I, H) * sizeof(float));
  L.rec_part = take(rec_part_floats(S, H) * sizeof(float));
  L.total = off;
  return L;
}

}  // namespace cmhse

using namespace cmhse;

extern "C" size_t cmhse_gru_pool_bwd_workspace(int32_t S, int32_t Tmax, int64_t sum_T, int32_t I,
                                               int32_t H, int32_t pool_mode) {
  (void)Tmax;
  if (S <= 0 || sum_T <= 0 || I <= 0 || H <= 0) return 0;
  return bwd_ws_layout(S, sum_T, I, H, pool_mode & kModeMask).total;
}

namespace {

// One validated cmhse_gru_pool_bwd request with its workspace carved up.
struct BwdJob {
  const cmhse_seq_batch* b;
  const cmhse_gru_weights* w;
  const cmhse_gru_grads* g;
  const float* dout;
  const uint64_t* dx_rows;
  float* d_emb_table;
  float* dh0;
  const char* fws;
  char* ws;
  GruWs F;
  BwdWs L;
  int64_t sum_T, off;     // off: running step offset of the BPTT walk (starts at sum_T)
  int32_t pool_mode;
  BwdStepParams sp;
  // weight gradients beside the chain: the packed rows [off, chunk_hi) have been produced by the
  // BPTT steps launched so far and not yet been contracted; `side` is the stream the products of
  // a closed chunk are launched on (== the main stream when the caller gave none)
  hipStream_t side;
  int64_t chunk_hi;
  bool chunk_first;
  hipStream_t st;         // the stream of this request's chain (its own, or the call's)
  int tail_lo;            // steps >= tail_lo run inside ONE resident kernel (gru_bwd_tail_kernel); -1 = none
};

int bwd_prepare(const cmhse_seq_batch* b, const cmhse_gru_weights* w, int32_t pool_mode,
                const float* dout, const void* fwd_workspace, const cmhse_gru_grads* g,
                const uint64_t* dx_rows, float* d_emb_table, float* dh0, void* workspace,
                size_t workspace_bytes, BwdJob* job) {
  if (!b || !w || !dout || !fwd_workspace || !g || !workspace) return CMHSE_ERR_ARG;
  pool_mode &= kModeMask;
  if (pool_mode != CMHSE_POOL_LAST && pool_mode != CMHSE_POOL_ATTN && pool_mode != CMHSE_POOL_MAX &&
      pool_mode != CMHSE_POOL_ALL)
    return CMHSE_ERR_ARG;
  if (b->S <= 0 || b->Tmax <= 0 || b->I <= 0 || b->H <= 0 || !b->step_count_host) return CMHSE_ERR_ARG;
  if (!g->dw_ih || !g->dw_hh || !g->db_ih || !g->db_hh) return CMHSE_ERR_ARG;
  if (pool_mode == CMHSE_POOL_ATTN && (!g->dw_lin || !g->db_lin || !g->dw_att || !w->w_lin || !w->w_att))
    return CMHSE_ERR_ARG;
  if (dx_rows && d_emb_table) return CMHSE_ERR_ARG;
  if (d_emb_table && !b->tok_rows) return CMHSE_ERR_ARG;
  if (dh0 && !b->h0_rows) return CMHSE_ERR_ARG;
  const int S = b->S, Tmax = b->Tmax, I = b->I, H = b->H;
  int64_t sum_T = 0;
  for (int t = 0; t < Tmax; ++t) sum_T += b->step_count_host[t];
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_gru_pool_bwd_workspace(S, Tmax, sum_T, I, H, pool_mode))
    return CMHSE_ERR_WORKSPACE;
  job->b = b; job->w = w; job->g = g; job->dout = dout; job->dx_rows = dx_rows;
  job->d_emb_table = d_emb_table; job->dh0 = dh0;
  job->fws = static_cast<const char*>(fwd_workspace);
  job->ws = static_cast<char*>(workspace);
  job->F = gru_ws_layout(S, sum_T, H, pool_mode | CMHSE_SAVE_FOR_BACKWARD);
  job->L = bwd_ws_layout(S, sum_T, I, H, pool_mode);
  job->sum_T = sum_T;
  job->off = sum_T;
  job->pool_mode = pool_mode;
  return CMHSE_OK;
}

// Phase 1 (pooling backward -> dpool) and the W_hh transpose the BPTT steps read, on `st`; what
// the chain does not wait for (the row-address tables of the weight-gradient products, dW_lin,
// db_lin, d att_w) goes to the job's side stream.
void bwd_begin(BwdJob& j, hipStream_t st) {
  const cmhse_seq_batch* b = j.b;
  const cmhse_gru_weights* w = j.w;
  const cmhse_gru_grads* g = j.g;
  const int S = b->S, I = b->I, H = b->H;
  const int64_t sum_T = j.sum_T;
  const GruWs& F = j.F;
  const BwdWs& L = j.L;
  const char* fws = j.fws;
  char* ws = j.ws;
  const float* hs = reinterpret_cast<const float*>(fws + F.hs);
  float* dpool = reinterpret_cast<float*>(ws + L.dpool);
  float* carry = reinterpret_cast<float*>(ws + L.carry);
  float* whh_t = reinterpret_cast<float*>(ws + L.whh_t);
  float* zero_row = reinterpret_cast<float*>(ws + L.zero_row);
  float* cs_scratch = reinterpret_cast<float*>(ws + L.colsum);
  const float* dout = j.dout;
  const int pool_mode = j.pool_mode;
  const bool beside = j.side != st;

  (void)hipMemsetAsync(zero_row, 0, L.tail_sync + 256 - L.zero_row, st);   // zero_row and the tail kernel's barrier counter
  (void)hipMemsetAsync(carry, 0, static_cast<size_t>(S) * H * 4, st);
  if (beside) stream_after(j.side, st);     // fork: the caller's inputs (and zero_row) are ready
  // W_hh^T for the chain depends on the weights only: with a side stream it is the FIRST thing
  // there, beside the pooling backward on `st` (it stood 50 us in front of the chain's first
  // launch), and the chain's stream waits for just that launch's event further down
  hipEvent_t whh_ev = nullptr;
  if (beside) {
    launch_transpose(w->w_hh, whh_t, 3 * H, H, j.side);
    whh_ev = event_get(false);
    if (whh_ev != nullptr && hipEventRecord(whh_ev, j.side) != hipSuccess) {
      (void)hipGetLastError();
      event_put(whh_ev, false);
      whh_ev = nullptr;
      (void)hipStreamSynchronize(j.side);      // no event: host-side ordering, then nothing to wait for
    }
    if (whh_ev == nullptr) (void)hipStreamSynchronize(j.side);
  }
  // per packed row: address of x_{t,s} and of h_{t-1,s} — the B operands of dW_ih / dW_hh
  RowAddrParams rp;
  rp.x_rows = b->x_rows; rp.tok_rows = b->tok_rows; rp.emb = b->emb_table; rp.h0_rows = b->h0_rows;
  rp.step_off = b->step_off; rp.hs = hs; rp.zero_row = zero_row;
  rp.xaddr = reinterpret_cast<uint64_t*>(ws + L.xaddr);
  rp.hpaddr = reinterpret_cast<uint64_t*>(ws + L.hpaddr);
  rp.p_t = reinterpret_cast<int32_t*>(ws + L.p_t);
  rp.hsaddr = (pool_mode == CMHSE_POOL_ATTN) ? reinterpret_cast<uint64_t*>(ws + L.hsaddr) : nullptr;
  rp.Tmax = b->Tmax; rp.I = I; rp.H = H; rp.vocab = b->vocab; rp.x_step = b->x_step_floats;
  rp.sum_T = sum_T;
  hipLaunchKernelGGL(row_addr_kernel, dim3(static_cast<unsigned>((sum_T + 255) / 256)), dim3(256),
                     0, j.side, rp);
  // ---- 1. pooling backward -> dpool ----
  if (pool_mode == CMHSE_POOL_ATTN) {
    float* du = reinterpret_cast<float*>(ws + L.du);
    float* de = reinterpret_cast<float*>(ws + L.de);
    float* wlin_t = reinterpret_cast<float*>(ws + L.wlin_t);
    const float* v = reinterpret_cast<const float*>(fws + F.v);
    AttnBwdParams ap;
    ap.dout = dout; ap.hs = hs; ap.e_part = reinterpret_cast<const float*>(fws + F.e_part);
    ap.lens = b->lens; ap.out_row = b->out_row; ap.step_off = b->step_off;
    ap.dpool = dpool; ap.de = de; ap.rows = sum_T; ap.H = H; ap.n_tiles = (H + kAttBN - 1) / kAttBN;
    hipLaunchKernelGGL(attn_pool_bwd_kernel, dim3(S), dim3(kPoolBwdThreads), 0, st, ap);
    hipLaunchKernelGGL(attn_du_kernel, dim3(static_cast<unsigned>(sum_T)), dim3(kThreads), 0, st,
                       de, v, w->w_att, du, sum_T, H);
    // the chain needs dpool += du . W_lin  (NT on W_lin^T) ...  (all of it, here: with only the rows
    // of the chain's first launches in front and the rest in chunks on the side stream, the BPTT
    // launches waiting for their chunk's event, the training step was 0.15 ms SLOWER)
    launch_transpose(w->w_lin, wlin_t, H, H, st);
    launch_nt_rows_once(du, H, wlin_t, H, nullptr, static_cast<int>(sum_T), H, H,
                        reinterpret_cast<float*>(ws + L.dx_part), st, dpool, H);
    // ... but not d W_lin[n][k] = sum_p du[p][n] hs[p][k], d b_lin = sum_p du[p], d att_w = sum_p de_p v_p
    if (beside) stream_after(j.side, st);
    TnRowsLaunch tl;
    tl.add(du, H, reinterpret_cast<const uint64_t*>(ws + L.hsaddr), g->dw_lin, H, g->db_lin, H, H);
    tl.launch(0, sum_T, false, j.side, reinterpret_cast<float*>(ws + L.wg_part),
              static_cast<int64_t>(wg_part_floats(sum_T, I, H)));
    launch_colsum(v, de, g->dw_att, cs_scratch, sum_T, H, H, j.side);
  } else {
    if (pool_mode != CMHSE_POOL_ALL)
      (void)hipMemsetAsync(dpool, 0, static_cast<size_t>(sum_T) * H * 4, st);
    PoolBwdParams pp;
    pp.dout = dout; pp.lens = b->lens; pp.out_row = b->out_row; pp.step_off = b->step_off;
    pp.argmax = reinterpret_cast<const int32_t*>(fws + F.argmax);
    pp.dpool = dpool; pp.S = S; pp.H = H; pp.mode = pool_mode;
    hipLaunchKernelGGL(pool_scatter_bwd_kernel, dim3(S), dim3(kThreads), 0, st, pp);
  }

  // W_hh^T for the chain: depends on the weights only, so with a side stream it runs there beside
  // the pooling backward (it stood 50 us in front of the chain's first launch) and the chain's
  // stream waits for it
  if (whh_ev != nullptr) {
    (void)hipStreamWaitEvent(st, whh_ev, 0);   // (the wait captures the record: the event may be reused)
    event_put(whh_ev, false);
  } else if (!beside) {
    launch_transpose(w->w_hh, whh_t, 3 * H, H, st);
  }
  if (j.dx_rows || j.d_emb_table)   // d(input) = dGx . W_ih runs on W_ih^T (NT)
    launch_transpose(w->w_ih, reinterpret_cast<float*>(ws + L.wih_t), 3 * H, I, j.side);
  if (j.dx_rows || j.d_emb_table) {
    // the output row of every packed row: d x_{t,s} at dx_rows[s] + t * x_step floats, or the
    // row of token (t, s) inside d_emb_table (same offset as inside the table)
    RowAddrParams rq = rp;
    rq.xaddr = reinterpret_cast<uint64_t*>(ws + L.dxaddr);
    rq.hpaddr = nullptr;
    rq.hsaddr = nullptr;
    rq.p_t = nullptr;
    if (j.dx_rows) { rq.x_rows = j.dx_rows; rq.tok_rows = nullptr; }
    else rq.emb = j.d_emb_table;
    hipLaunchKernelGGL(row_addr_kernel, dim3(static_cast<unsigned>((sum_T + 255) / 256)), dim3(256),
                       0, j.side, rq);
  }
  BwdStepParams& sp = j.sp;
  sp.whh_t = whh_t; sp.dpool = dpool;
  sp.gates = reinterpret_cast<const float*>(fws + F.gates);
  sp.hs = hs; sp.h0_rows = b->h0_rows;
  sp.out_row = b->out_row; sp.carry = carry;
  sp.dgx = reinterpret_cast<float*>(ws + L.dgx);
  sp.dgh = reinterpret_cast<float*>(ws + L.dgh);
  sp.dh0 = j.dh0; sp.H = H;
  j.chunk_hi = sum_T;
  j.chunk_first = true;
}

// Packed rows a weight-gradient chunk should at least span (K of its products): Tunables::
// bwd_chunk_rows, 2048.  Long enough that a tile's fill / drain and the read-modify-write of its C
// tile are amortised; what bounds it from above is only the LAST chunk, which nothing hides.
// Measured on the training step (ms, ICEP / C3D): 512 rows 9.85 / 8.50, 1024 9.62 / 8.39, 2048
// 9.42 / 8.17, 4096 9.48 / 8.12, 8192 9.73 / 8.14, no chunking at all (every product after the chain)
// 9.47 / 8.26 — with the two-launch step and the resident tail the chain and the products are
// bound by the same L2 -> CU fabric, and the step time is the SUM of their stand-alone times
// whether they run side by side or one after the other.

// The rows [j.off, j.chunk_hi) are final (the BPTT step that wrote j.off .. has been launched on
// `st`): contract them into dW_ih / db_ih, dW_hh / db_hh and scatter their d(input), on the side
// stream, ordered behind that step by an event.
void bwd_chunk(BwdJob& j, hipStream_t st) {
  const int64_t p0 = j.off, p1 = j.chunk_hi;
  if (p1 <= p0) return;
  const cmhse_seq_batch* b = j.b;
  const cmhse_gru_grads* g = j.g;
  const int I = b->I, H = b->H;
  char* ws = j.ws;
  const BwdWs& L = j.L;
  float* dgx = reinterpret_cast<float*>(ws + L.dgx);
  float* dgh = reinterpret_cast<float*>(ws + L.dgh);
  if (j.side != st) stream_after(j.side, st);
  TnRowsLaunch tl;
  tl.add(dgx, 3 * H, reinterpret_cast<const uint64_t*>(ws + L.xaddr), g->dw_ih, I, g->db_ih, 3 * H, I);
  tl.add(dgh, 3 * H, reinterpret_cast<const uint64_t*>(ws + L.hpaddr), g->dw_hh, H, g->db_hh, 3 * H, H);
  tl.launch(p0, p1, !j.chunk_first, j.side, reinterpret_cast<float*>(ws + L.wg_part),
            static_cast<int64_t>(wg_part_floats(j.sum_T, I, H)), p0 > 0);
  if (j.dx_rows || j.d_emb_table) {
    // d(input): dx_p = dGx_p . W_ih, rows scattered through the output-row table.  A
    // time-constant input and the embedding table receive SUMS over packed rows (rows repeat:
    // float atomics into the caller's zeroed storage, K split when the tiles are few); ordinary
    // rows are written exactly once, reproducibly.
    const uint64_t* dxaddr = reinterpret_cast<const uint64_t*>(ws + L.dxaddr) + p0;
    const float* wih_t = reinterpret_cast<const float*>(ws + L.wih_t);
    const bool sums = j.d_emb_table != nullptr || b->x_step_floats == 0;
    if (sums)
      launch_nt_out(dgx + p0 * 3 * H, 3 * H, wih_t, 3 * H, nullptr, 0, dxaddr,
                    static_cast<int>(p1 - p0), I, 3 * H, 2, j.side, 2);
    else
      launch_nt_rows_once(dgx + p0 * 3 * H, 3 * H, wih_t, 3 * H, dxaddr, static_cast<int>(p1 - p0), I,
                          3 * H, reinterpret_cast<float*>(ws + L.dx_part), j.side);
  }
  j.chunk_hi = p0;
  j.chunk_first = false;
}

// Active sequences at or below which a BPTT step runs on gru_bwd_step_mid_kernel
// (Tunables::bwd_mid_max_seqs; 0 = never).  Its unit tile is 16: unlike the forward step (K = H),
// narrower tiles LOSE here — the 32 dgh rows of K = 3H floats (384 KB) every workgroup pulls
// dominate, and 128-256 workgroups of them cost more L2 bandwidth than the spread gains (train_emb
// step, C3D: 16 units 12.8 ms, 8 units 16.0, 4 units 15.9; later, at 11.3 ms, 32 units — two column
// blocks per wave — 11.9; round 3, at 9.7 ms: 4 / 8 units only for steps with <= 2 / 4 ... 8 / 16
// active sequences, the long tail of the sentence chain: 9.74-9.78 against 9.68).
static int bwd_mid_max_seqs() { return tunables().bwd_mid_max_seqs.load(std::memory_order_relaxed); }
constexpr int kBwdMidUnits = 16;
// active sequences at or below which the 32 x 32-tile BPTT step (unaligned shapes, large batches)
// splits K over 8 waves instead of 4: a pure latency chain on an under-filled chip
constexpr int kBwdNw8Max = 256;

// The steps t >= tail_lo >= 1 with at most kTailMaxSeqs (32) active sequences, when there are at least
// bwd_tail_min_steps of them (Tunables; 0 = never), run inside gru_bwd_tail_kernel on the chain's
// stream; bwd_steps skips their launches and keeps its bookkeeping.
static void bwd_tail(BwdJob& j) {
  j.tail_lo = -1;
  const cmhse_seq_batch* b = j.b;
  const int H = b->H, Tmax = b->Tmax;
  const int min_steps = multi_step_knob(tunables().bwd_tail_min_steps);
  if (min_steps <= 0 || H % 16 != 0 || H > 1024 || !b->step_off || !resident_fits(H / 16)) return;
  int lo = Tmax;
  while (lo - 1 >= 1 && b->step_count_host[lo - 1] <= kTailMaxSeqs) --lo;
  if (Tmax - lo < min_steps) return;
  BwdTailParams q;
  const BwdStepParams& sp = j.sp;
  q.whh_t = sp.whh_t; q.dpool = sp.dpool; q.gates = sp.gates; q.hs = sp.hs;
  q.step_off = b->step_off;
  q.carry = sp.carry; q.dgx = sp.dgx; q.dgh = sp.dgh;
  unsigned* words = reinterpret_cast<unsigned*>(j.ws + j.L.tail_sync);
  q.sync = make_grid_sync(words, words + 63);
  q.H = H; q.t_hi = Tmax - 1; q.t_lo = lo;
  const int kb = 2 * ((3 * H / 16 + 15) / 16);   // 16-k blocks per wave, whole pairs (mid_phase's ownership)
  const dim3 grid(static_cast<unsigned>(H / 16)), block(512);
  const bool two = b->step_count_host[lo] > 16;    // 17 ... 32 sequences at the tail's widest step
#define BWD_TAIL_(KB)                                                                       \
  do {                                                                                      \
    if (two) hipLaunchKernelGGL((gru_bwd_tail_kernel<KB, 2>), grid, block, 0, j.st, q);     \
    else hipLaunchKernelGGL((gru_bwd_tail_kernel<KB, 1>), grid, block, 0, j.st, q);         \
  } while (0)
  if (kb <= 2) BWD_TAIL_(2);
  else if (kb <= 4) BWD_TAIL_(4);
  else if (kb <= 6) BWD_TAIL_(6);
  else if (kb <= 12) BWD_TAIL_(12);
  else BWD_TAIL_(24);
#undef BWD_TAIL_
  j.tail_lo = lo;
}

// Phase 2: BPTT of all jobs, last step first.  Launch i serves step Tmax_k - 1 - i of every job k
// that still has one (and the extra t = -1 launch of a job that wants d h0); jobs of equal block
// size share the launch.
void bwd_steps(BwdJob* jobs, int n) {
  constexpr int nw8_max = kBwdNw8Max;
  int longest = 0;
  for (int k = 0; k < n; ++k) longest = jobs[k].b->Tmax > longest ? jobs[k].b->Tmax : longest;
  for (int i = 0; i <= longest; ++i) {
    int kind[CMHSE_MAX_JOBS];      // 0 = not in this launch, 1 = 4 waves, 2 = 8 waves, +4 = scalar loads; 8 | shape bits = gru_bwd_step_mid_kernel
    unsigned grid_k[CMHSE_MAX_JOBS];
    for (int k = 0; k < n; ++k) {
      BwdJob& j = jobs[k];
      const cmhse_seq_batch* b = j.b;
      const int t = b->Tmax - 1 - i;
      kind[k] = 0;
      if (t < -1 || (t < 0 && !j.dh0)) continue;
      const int S_t = (t >= 0) ? b->step_count_host[t] : b->S;
      const int S_next = (t + 1 < b->Tmax) ? b->step_count_host[t + 1] : 0;
      const int64_t off_next = j.off;          // step_off[t+1]
      if (t >= 0) j.off -= S_t;                // step_off[t]
      BwdStepParams& sp = j.sp;
      sp.t = t; sp.S_t = S_t; sp.S_next = S_next;
      sp.dgh_next = sp.dgh + off_next * 3 * b->H;
      sp.off_cur = j.off;
      sp.off_prev = (t > 0) ? j.off - b->step_count_host[t - 1] : 0;
      if (i == 0) bwd_tail(j);                 // the chain's few-sequence tail, if it has one: one kernel
      if (j.tail_lo >= 0 && t >= j.tail_lo) continue;   // (kind 0) the resident kernel does this step
      grid_k[k] = static_cast<unsigned>((b->H + 31) / 32) * ((S_t + 31) / 32);
      // few active sequences: a pure latency chain on an under-filled chip -> 8 waves split K
      kind[k] = ((S_t <= nw8_max) ? 2 : 1) | ((b->H % 4 == 0) ? 0 : 4);
      if (b->H % 4 == 0 && S_t <= bwd_mid_max_seqs()) {   // the 16 x 16 x 4 tile shapes
        const int bm = (S_t <= 16) ? 16 : 32;
        const int bu = kBwdMidUnits;
        kind[k] = 8 | (bm == 16 ? 16 : 0);
        grid_k[k] = static_cast<unsigned>((b->H + bu - 1) / bu) * ((S_t + bm - 1) / bm);
        const int split_min = tunables().bwd_split_min_seqs.load(std::memory_order_relaxed);
        if (split_min > 0 && S_t >= split_min) kind[k] = 32;   // two launches, a third of the bytes
      }
    }
    for (int k = 0; k < n; ++k) {
      if (kind[k] != 32) continue;
      BwdJob& j = jobs[k];
      const BwdStepParams& sp = j.sp;
      const int H = sp.H, K = 3 * H;
      GatesBwdParams gp;
      gp.s = sp; gp.part = nullptr; gp.splits = 0; gp.m_pad = 0;
      if (sp.S_next > 0) {
        RecPartParams rp;
        rp.a = sp.dgh_next; rp.b = sp.whh_t;
        rp.part = reinterpret_cast<float*>(j.ws + j.L.rec_part);
        rp.S_next = sp.S_next; rp.H = H; rp.K = K;
        rp.n_tiles = (H + kRecBN - 1) / kRecBN;
        const int m_tiles = (sp.S_next + kRecBM - 1) / kRecBM, tiles = rp.n_tiles * m_tiles;
        int splits = 256 / tiles;       // (512 / 768 / 1024 workgroups measured 4 / 9 / 14 % slower: more partials)
        if (splits > K / 64) splits = K / 64;
        if (splits < 1) splits = 1;
        rp.k_slice = ((K + splits - 1) / splits + kRecBK - 1) / kRecBK * kRecBK;
        splits = (K + rp.k_slice - 1) / rp.k_slice;
        rp.m_pad = m_tiles * kRecBM;
        hipLaunchKernelGGL(bwd_rec_part_kernel, dim3(tiles, splits), dim3(kRecThreads), 0, j.st, rp);
        gp.part = rp.part; gp.splits = splits; gp.m_pad = rp.m_pad;
      }
      const int64_t elems = static_cast<int64_t>(sp.S_t) * (H / 4);
      hipLaunchKernelGGL(bwd_gates_kernel, dim3(static_cast<unsigned>((elems + kThreads - 1) / kThreads)),
                         dim3(kThreads), 0, j.st, gp);
      kind[k] = 0;
    }
    for (int k = 0; k < n; ++k) {
      if (kind[k] == 0) continue;
      BwdStepGroup g;
      g.n = 0;
      unsigned grid = 0;
      const int kd = kind[k];
      hipStream_t st = jobs[k].st;
      for (int m = k; m < n; ++m) {
        if (kind[m] != kd || jobs[m].st != st) continue;   // same kernel, same stream: one launch
        g.j[g.n] = jobs[m].sp;
        g.start[g.n] = grid;
        grid += grid_k[m];
        ++g.n;
        kind[m] = 0;
      }
      for (int m = g.n; m < CMHSE_MAX_JOBS; ++m) g.start[m] = 0xffffffffu;
      const bool vec = (kd & 4) == 0;
      if ((kd & 8) != 0) {
        if ((kd & 16) != 0)
          hipLaunchKernelGGL((gru_bwd_step_mid_kernel<1, kBwdMidUnits>), dim3(grid), dim3(64 * kBwdMidNW), 0, st, g);
        else
          hipLaunchKernelGGL((gru_bwd_step_mid_kernel<2, kBwdMidUnits>), dim3(grid), dim3(64 * kBwdMidNW), 0, st, g);
      } else if ((kd & 3) == 2) {
        if (vec)
          hipLaunchKernelGGL((gru_bwd_step_kernel<true, 8>), dim3(grid), dim3(512), 0, st, g);
        else
          hipLaunchKernelGGL((gru_bwd_step_kernel<false, 8>), dim3(grid), dim3(512), 0, st, g);
      } else if (vec) {
        hipLaunchKernelGGL((gru_bwd_step_kernel<true, 4>), dim3(grid), dim3(kThreads), 0, st, g);
      } else {
        hipLaunchKernelGGL((gru_bwd_step_kernel<false, 4>), dim3(grid), dim3(kThreads), 0, st, g);
      }
    }
    // weight gradients of the rows this launch completed, beside the rest of the chain: a chunk
    // closes when it spans bwd_chunk_rows rows, and at step 0
    for (int k = 0; k < n; ++k) {
      BwdJob& j = jobs[k];
      const int t = j.b->Tmax - 1 - i;
      if (t < 0) continue;
      if (t == 0 || j.chunk_hi - j.off >= tunables().bwd_chunk_rows.load(std::memory_order_relaxed)) {
        bwd_chunk(j, j.st);
      }
    }
  }
}

// Join: everything the side stream did for job j is ordered in front of what follows on `st`.
void bwd_finish(BwdJob& j, hipStream_t st) {
  if (j.off != 0 || j.chunk_hi != 0) {   // (defensive: the t = 0 step always closes the last chunk)
    j.off = 0;
    bwd_chunk(j, st);
  }
  if (j.side != st) stream_after(st, j.side);
}

}  // namespace

extern "C" int cmhse_gru_pool_bwd_multi(const cmhse_gru_bwd_job* reqs, int32_t n_jobs,
                                        void* stream_) {
  if (!reqs || n_jobs <= 0 || n_jobs > CMHSE_MAX_JOBS) return CMHSE_ERR_ARG;
  if (resident_check() != CMHSE_OK) return CMHSE_ERR_TIMEOUT;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  BwdJob jobs[CMHSE_MAX_JOBS];
  for (int k = 0; k < n_jobs; ++k) {
    const cmhse_gru_bwd_job& r = reqs[k];
    const int rc = bwd_prepare(r.seqs, r.weights, r.pool_mode, r.dout, r.fwd_workspace, r.grads,
                               r.dx_rows, r.d_emb_table, r.dh0, r.workspace, r.workspace_bytes,
                               &jobs[k]);
    if (rc != CMHSE_OK) return rc;
    jobs[k].st = r.stream ? static_cast<hipStream_t>(r.stream) : st;
    jobs[k].side = r.side_stream ? static_cast<hipStream_t>(r.side_stream) : jobs[k].st;
  }
  auto first_use = [&](int k) {   // fork / join every own stream once
    if (jobs[k].st == st) return false;
    for (int m = 0; m < k; ++m)
      if (jobs[m].st == jobs[k].st) return false;
    return true;
  };
  for (int k = 0; k < n_jobs; ++k)
    if (first_use(k)) stream_after(jobs[k].st, st);
  for (int k = 0; k < n_jobs; ++k) bwd_begin(jobs[k], jobs[k].st);
  bwd_steps(jobs, n_jobs);
  for (int k = 0; k < n_jobs; ++k) bwd_finish(jobs[k], jobs[k].st);
  for (int k = 0; k < n_jobs; ++k)
    if (first_use(k) && !(reqs[k].pool_mode & CMHSE_NO_JOIN)) stream_after(st, jobs[k].st);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_gru_pool_bwd(const cmhse_seq_batch* b, const cmhse_gru_weights* w,
                                  int32_t pool_mode, const float* dout, const void* fwd_workspace,
                                  const cmhse_gru_grads* g, const uint64_t* dx_rows,
                                  float* d_emb_table, float* dh0, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
  cmhse_gru_bwd_job r;
  r.seqs = b; r.weights = w; r.pool_mode = pool_mode; r.dout = dout;
  r.fwd_workspace = fwd_workspace; r.grads = g; r.dx_rows = dx_rows; r.d_emb_table = d_emb_table;
  r.dh0 = dh0; r.workspace = workspace; r.workspace_bytes = workspace_bytes;
  r.stream = nullptr;
  r.side_stream = nullptr;
  return cmhse_gru_pool_bwd_multi(&r, 1, stream_);
}

extern "C" int cmhse_l2norm_rows_bwd(const float* x, const float* g, float* dx, int32_t rows,
                                     int32_t cols, void* stream_) {
  if (!x || !g || !dx || rows < 0 || cols <= 0) return CMHSE_ERR_ARG;
  if (rows == 0) return CMHSE_OK;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(rows), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream_), x, g, dx, cols);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" size_t cmhse_contrastive_bwd_workspace(int32_t n) {
  if (n <= 0) return 0;
  return 2 * ws_align(static_cast<size_t>(n) * n * 4) + 4 * ws_align(static_cast<size_t>(n) * 4);
}

extern "C" int cmhse_contrastive_bwd(const float* im, const float* s, const float* scores,
                                     int32_t n, int32_t D, float margin, int32_t max_violation,
                                     int32_t norm, const float* grad_out, float* d_im, float* d_s,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
  if (!im || !s || !scores || !grad_out || !d_im || !d_s || !workspace || n <= 0 || D <= 0)
    return CMHSE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_contrastive_bwd_workspace(n))
    return CMHSE_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  const size_t mat = ws_align(static_cast<size_t>(n) * n * 4), vecb = ws_align(static_cast<size_t>(n) * 4);
  LossBwdParams q;
  q.scores = scores; q.gout = grad_out; q.n = n; q.max_violation = max_violation; q.norm = norm;
  q.margin = margin;
  q.G = reinterpret_cast<float*>(ws);
  q.GT = reinterpret_cast<float*>(ws + mat);
  q.row_arg = reinterpret_cast<int32_t*>(ws + 2 * mat);
  q.col_arg = reinterpret_cast<int32_t*>(ws + 2 * mat + vecb);
  q.row_cnt = reinterpret_cast<float*>(ws + 2 * mat + 2 * vecb);
  q.col_cnt = reinterpret_cast<float*>(ws + 2 * mat + 3 * vecb);
  q.blk_off = nullptr; q.blk_stride = 0; q.score_stride = 0; q.vec_stride = 0;
  hipLaunchKernelGGL(loss_bwd_stats_kernel, dim3((2 * n + 3) / 4), dim3(kThreads), 0, st, q);
  const int64_t elems = static_cast<int64_t>(n) * n;
  hipLaunchKernelGGL(loss_bwd_build_kernel, dim3(static_cast<unsigned>((elems + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, st, q);
  const bool vec = (n % 4 == 0) && (D % 4 == 0);
  // d im[i][d] = sum_j G[i][j] s[j][d]  = TN(A = G^T, B = s);  d s[j][d] = sum_i G[i][j] im[i][d]
  launch_tn(q.GT, n, s, D, nullptr, d_im, D, n, D, n, nullptr, vec, st);
  launch_tn(q.G, n, im, D, nullptr, d_s, D, n, D, n, nullptr, vec, st);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" size_t cmhse_contrastive_blocks_bwd_workspace(int32_t n_blocks, int32_t max_n) {
  if (n_blocks <= 0 || max_n <= 0) return 0;
  return static_cast<size_t>(n_blocks) *
         (2 * ws_align(static_cast<size_t>(max_n) * max_n * 4) + 4 * ws_align(static_cast<size_t>(max_n) * 4));
}

extern "C" int cmhse_contrastive_blocks_bwd(const float* im, const float* s, const float* scores,
                                            const int32_t* blk_off, int32_t n_blocks, int32_t max_n,
                                            int32_t D, float margin, int32_t max_violation,
                                            int32_t norm, const float* grad_out, float* d_im,
                                            float* d_s, void* workspace, size_t workspace_bytes,
                                            void* stream_) {
  if (!im || !s || !scores || !blk_off || !grad_out || !d_im || !d_s || !workspace || n_blocks <= 0 ||
      max_n <= 0 || D <= 0)
    return CMHSE_ERR_ARG;
  if (n_blocks > 65535) return CMHSE_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_contrastive_blocks_bwd_workspace(n_blocks, max_n))
    return CMHSE_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  const size_t mat = ws_align(static_cast<size_t>(max_n) * max_n * 4);
  const size_t vecb = ws_align(static_cast<size_t>(max_n) * 4);
  LossBwdParams q;
  q.scores = scores; q.gout = grad_out; q.n = max_n; q.max_violation = max_violation; q.norm = norm;
  q.margin = margin;
  q.G = reinterpret_cast<float*>(ws);
  q.GT = reinterpret_cast<float*>(ws + n_blocks * mat);
  char* vecs = ws + 2 * n_blocks * mat;
  q.row_arg = reinterpret_cast<int32_t*>(vecs);
  q.col_arg = reinterpret_cast<int32_t*>(vecs + n_blocks * vecb);
  q.row_cnt = reinterpret_cast<float*>(vecs + 2 * n_blocks * vecb);
  q.col_cnt = reinterpret_cast<float*>(vecs + 3 * n_blocks * vecb);
  q.blk_off = blk_off;
  q.blk_stride = static_cast<int64_t>(mat / 4);           // floats between blocks of G / GT
  q.vec_stride = static_cast<int32_t>(vecb / 4);
  // the stored scores of the forward pass: block b at scores + b * max_n * max_n
  // (cmhse_contrastive_blocks_fwd's workspace), leading dimension = the block's own n
  q.score_stride = static_cast<int64_t>(max_n) * max_n;
  hipLaunchKernelGGL(loss_bwd_stats_kernel, dim3((2 * max_n + 3) / 4, n_blocks), dim3(kThreads), 0, st, q);
  const int64_t elems = static_cast<int64_t>(max_n) * max_n;
  hipLaunchKernelGGL(loss_bwd_build_kernel,
                     dim3(static_cast<unsigned>((elems + kThreads - 1) / kThreads), n_blocks),
                     dim3(kThreads), 0, st, q);
  // d im[i][d] = sum_j G[i][j] s[j][d] = TN(A = G^T, B = s);  d s[j][d] = sum_i G[i][j] im[i][d]
  TnParams t;
  t.lda = max_n; t.ldb = D; t.b_addr = nullptr; t.ldc = D; t.M = max_n; t.N = D; t.K = max_n;
  t.n_tiles = (D + 127) / 128; t.scale = nullptr; t.blk_off = blk_off; t.blk_stride = q.blk_stride;
  const unsigned grid = static_cast<unsigned>(t.n_tiles) * ((max_n + 127) / 128);
  const size_t smem = TnSmem<128, 128>::kBytes;
  t.a = q.GT; t.b = s; t.c = d_im;
  hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(grid, n_blocks), dim3(kThreads), smem, st, t);
  t.a = q.G; t.b = im; t.c = d_s;
  hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(grid, n_blocks), dim3(kThreads), smem, st, t);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// cmhse_step_losses_bwd: per-term upstream gradients, the batched ContrastiveLoss backward, then
// F.normalize's backward over the sum of every use of each encoder output.
// ---------------------------------------------------------------------------------------------
struct StepNormBwdParams {
  const float* x[CMHSE_STEP_LOSS_MAX];
  float* dx[CMHSE_STEP_LOSS_MAX];
  int32_t rows[CMHSE_STEP_LOSS_MAX];
  int32_t term_a[CMHSE_STEP_LOSS_MAX], term_b[CMHSE_STEP_LOSS_MAX], term_off[CMHSE_STEP_LOSS_MAX];
  float weight[CMHSE_STEP_LOSS_MAX];
  int32_t n_emb, n_terms, D;
  const float* d_im;  // [R, D]
  const float* d_s;   // [R, D]
};

__global__ void step_gout_kernel(const StepNormBwdParams p, const float* grad_total, float* gout) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float g = *grad_total;
#pragma unroll
    for (int k = 0; k < CMHSE_STEP_LOSS_MAX; ++k)
      if (k < p.n_terms) gout[k] = p.weight[k] * g;
  }
}

// one workgroup per encoder-output row: g = sum over the terms that use the row (term order, left
// operand before right), then l2norm_bwd_kernel's arithmetic
__global__ __launch_bounds__(kThreads) void step_norm_bwd_kernel(const StepNormBwdParams p) {
  int e = -1, i = 0, off = 0;
  const float* xr = nullptr;
  float* dxr = nullptr;
#pragma unroll
  for (int k = 0; k < CMHSE_STEP_LOSS_MAX; ++k) {
    if (k < p.n_emb) {
      const int r = static_cast<int>(blockIdx.x) - off;
      if (r >= 0 && r < p.rows[k]) {
        e = k;
        i = r;
        xr = p.x[k] + static_cast<int64_t>(r) * p.D;
        dxr = p.dx[k] + static_cast<int64_t>(r) * p.D;
      }
      off += p.rows[k];
    }
  }
  if (e < 0) return;
  const float* use[2 * CMHSE_STEP_LOSS_MAX];
#pragma unroll
  for (int k = 0; k < CMHSE_STEP_LOSS_MAX; ++k) {
    const bool live = k < p.n_terms;
    const int64_t row = static_cast<int64_t>(p.term_off[k] + i) * p.D;
    use[2 * k] = (live && p.term_a[k] == e) ? p.d_im + row : nullptr;
    use[2 * k + 1] = (live && p.term_b[k] == e) ? p.d_s + row : nullptr;
  }
  auto grad = [&](int c) {
    float g = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * CMHSE_STEP_LOSS_MAX; ++u)
      if (use[u] != nullptr) g += use[u][c];
    return g;
  };
  float ss = 0.f, sg = 0.f;
  for (int c = threadIdx.x; c < p.D; c += kThreads) {
    ss += xr[c] * xr[c];
    sg += xr[c] * grad(c);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    ss += __shfl_xor(ss, o, 64);
    sg += __shfl_xor(sg, o, 64);
  }
  __shared__ float p1[4], p2[4];
  __shared__ float s_inv, s_dot;
  if ((threadIdx.x & 63) == 0) {
    p1[threadIdx.x >> 6] = ss;
    p2[threadIdx.x >> 6] = sg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n2 = p1[0] + p1[1] + p1[2] + p1[3];
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    s_inv = inv;
    s_dot = (p2[0] + p2[1] + p2[2] + p2[3]) * inv * inv;
  }
  __syncthreads();
  const float inv = s_inv, d = s_dot;
  for (int c = threadIdx.x; c < p.D; c += kThreads) dxr[c] = (grad(c) - xr[c] * d) * inv;
}

extern "C" int cmhse_step_losses_bwd(const cmhse_step_losses* d, const float* grad_total,
                                     float* const* dx, void* workspace, size_t workspace_bytes,
                                     void* stream_) {
  StepLossLayout L;
  if (!grad_total || !dx || !workspace || !step_loss_layout(d, &L)) return CMHSE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 || workspace_bytes < L.bytes)
    return CMHSE_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  StepNormBwdParams p = {};
  int64_t total_rows = 0;
  for (int e = 0; e < d->n_emb; ++e) {
    if (!d->x[e] || !dx[e]) return CMHSE_ERR_ARG;
    p.x[e] = d->x[e];
    p.dx[e] = dx[e];
    p.rows[e] = d->rows[e];
    total_rows += d->rows[e];
  }
  if (total_rows > 0x7fffffffLL) return CMHSE_ERR_UNSUPPORTED;
  int32_t off = 0;
  for (int k = 0; k < d->n_terms; ++k) {
    p.term_a[k] = d->term_a[k];
    p.term_b[k] = d->term_b[k];
    p.term_off[k] = off;
    p.weight[k] = d->weight[k];
    off += d->rows[d->term_a[k]];
  }
  p.n_emb = d->n_emb;
  p.n_terms = d->n_terms;
  p.D = d->D;
  float* d_im = reinterpret_cast<float*>(ws + L.d_im);
  float* d_s = reinterpret_cast<float*>(ws + L.d_s);
  p.d_im = d_im;
  p.d_s = d_s;
  float* gout = reinterpret_cast<float*>(ws + L.gout);
  hipLaunchKernelGGL(step_gout_kernel, dim3(1), dim3(64), 0, st, p, grad_total, gout);
  const int rc = cmhse_contrastive_blocks_bwd(
      reinterpret_cast<const float*>(ws + L.y_im), reinterpret_cast<const float*>(ws + L.y_s),
      reinterpret_cast<const float*>(ws + L.fwd_ws), reinterpret_cast<const int32_t*>(ws + L.blk_off),
      d->n_terms, L.max_n, d->D, d->margin, d->max_violation, d->norm, gout, d_im, d_s,
      ws + L.bwd_ws, cmhse_contrastive_blocks_bwd_workspace(d->n_terms, L.max_n), stream_);
  if (rc != CMHSE_OK) return rc;
  hipLaunchKernelGGL(step_norm_bwd_kernel, dim3(static_cast<unsigned>(total_rows)), dim3(kThreads), 0,
                     st, p);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_euclid_fwd(const float* a, const float* b, const uint64_t* b_rows, int32_t rows,
                                int32_t cols, int32_t norm, float* loss, float* row_dist,
                                void* stream_) {
  if (!a || (!b && !b_rows) || !loss || !row_dist || rows <= 0 || cols <= 0) return CMHSE_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  EuclidParams q;
  q.a = a; q.b = b; q.b_rows = b_rows; q.dist = row_dist; q.d_a = nullptr; q.gout = nullptr;
  q.loss = loss; q.rows = rows; q.cols = cols; q.norm = norm;
  hipLaunchKernelGGL(euclid_rows_kernel, dim3(rows), dim3(kThreads), 0, st, q, 0);
  hipLaunchKernelGGL(euclid_final_kernel, dim3(1), dim3(kThreads), 0, st, q);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_euclid_bwd(const float* a, const float* b, const uint64_t* b_rows, int32_t rows,
                                int32_t cols, int32_t norm, const float* grad_out, float* d_a,
                                void* stream_) {
  if (!a || (!b && !b_rows) || !grad_out || !d_a || rows <= 0 || cols <= 0) return CMHSE_ERR_ARG;
  EuclidParams q;
  q.a = a; q.b = b; q.b_rows = b_rows; q.dist = nullptr; q.d_a = d_a; q.gout = grad_out;
  q.loss = nullptr; q.rows = rows; q.cols = cols; q.norm = norm;
  hipLaunchKernelGGL(euclid_rows_kernel, dim3(rows), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream_), q, 1);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" size_t cmhse_groupwise_bwd_workspace(int32_t n, int32_t B) {
  if (n <= 0 || B <= 0) return 0;
  return 2 * ws_align(static_cast<size_t>(n) * n * 4) + cmhse_contrastive_bwd_workspace(B);
}

extern "C" int cmhse_groupwise_bwd(const float* im, const float* s, int32_t n, int32_t D,
                                   const int32_t* row_off, const int32_t* col_off, int32_t B,
                                   float margin, int32_t max_violation, int32_t norm,
                                   const float* reduced, const int32_t* arg, const float* grad_out,
                                   float* d_im, float* d_s, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
  if (!im || !s || !row_off || !col_off || !reduced || !arg || !grad_out || !d_im || !d_s ||
      !workspace || n <= 0 || D <= 0 || B <= 0 || B > 65535)
    return CMHSE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_groupwise_bwd_workspace(n, B))
    return CMHSE_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  const size_t big = ws_align(static_cast<size_t>(n) * n * 4);
  float* dS = reinterpret_cast<float*>(ws);
  float* dST = reinterpret_cast<float*>(ws + big);
  char* lws = ws + 2 * big;
  const size_t mat = ws_align(static_cast<size_t>(B) * B * 4), vecb = ws_align(static_cast<size_t>(B) * 4);
  LossBwdParams q;
  q.scores = reduced; q.gout = grad_out; q.n = B; q.max_violation = max_violation; q.norm = norm;
  q.margin = margin;
  q.G = reinterpret_cast<float*>(lws);
  q.GT = reinterpret_cast<float*>(lws + mat);
  q.row_arg = reinterpret_cast<int32_t*>(lws + 2 * mat);
  q.col_arg = reinterpret_cast<int32_t*>(lws + 2 * mat + vecb);
  q.row_cnt = reinterpret_cast<float*>(lws + 2 * mat + 2 * vecb);
  q.col_cnt = reinterpret_cast<float*>(lws + 2 * mat + 3 * vecb);
  q.blk_off = nullptr; q.blk_stride = 0; q.score_stride = 0; q.vec_stride = 0;
  hipLaunchKernelGGL(loss_bwd_stats_kernel, dim3((2 * B + 3) / 4), dim3(kThreads), 0, st, q);
  const int64_t elems = static_cast<int64_t>(B) * B;
  hipLaunchKernelGGL(loss_bwd_build_kernel, dim3(static_cast<unsigned>((elems + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, st, q);
  ExpandParams ep;
  ep.g_red = q.G; ep.arg = arg; ep.row_off = row_off; ep.col_off = col_off; ep.dS = dS;
  ep.dST = dST; ep.n = n; ep.B = B; ep.use_max = max_violation;
  hipLaunchKernelGGL(groupwise_expand_kernel, dim3(B, B), dim3(kThreads), 0, st, ep);
  const bool vec = (n % 4 == 0) && (D % 4 == 0);
  launch_tn(dST, n, s, D, nullptr, d_im, D, n, D, n, nullptr, vec, st);
  launch_tn(dS, n, im, D, nullptr, d_s, D, n, D, n, nullptr, vec, st);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}
