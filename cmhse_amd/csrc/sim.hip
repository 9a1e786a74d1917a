// sim.hip — N x M cosine-similarity contraction with fused ranking / loss epilogues.
//
//   cmhse_sim_rank        evaluation.i2t / t2i  (/root/reference/evaluation.py:160-213):
//                         numpy.dot + per-row numpy.argsort become one exact-fp32 MFMA GEMM whose
//                         epilogue counts, per row, the columns that beat the diagonal and tracks
//                         the arg-max column — the N x M matrix never reaches HBM.
//   cmhse_cosine_sim      loss.cosine_sim       (/root/reference/loss.py:12-13)
//   cmhse_contrastive_fwd loss.ContrastiveLoss.forward (/root/reference/loss.py:86-117)
//
// Bit-exactness of the ranks: every d[i][j] is produced by the same MFMA k-chain (nt_core.hpp:
// the k order per output element does not depend on the tile position), so the diagonal values
// written by the diagonal pass are bit-identical to what the counting pass recomputes, and
// `d[i][j] > d[i][i]` is evaluated on consistently rounded fp32 values, like the reference's
// comparison inside numpy.argsort.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/cmhse_hip.h"
#include "gru_ws.hpp"
#include "nt_core.hpp"
#include "step_loss.hpp"

namespace cmhse {

constexpr int kSimBM = 128;
constexpr int kSimBN = 128;
constexpr int kSimSmallMax = 512;   // stored-score matrices up to this size use 64 x 64 tiles

enum { kSimDiag = 0, kSimRank = 1, kSimStore = 2 };

struct SimParams {
  const float* A;  // [N, D]
  const float* B;  // [M, D]
  int32_t N, M, D, row0, nrows, n_tiles;
  int32_t m_tiles8;             // counting pass: row tiles per rasterisation group (see sim_kernel), 0 = plain
  float* diag;                  // [nrows]
  int32_t* rank;                // [nrows]
  unsigned long long* top1key;  // [nrows]
  float* scores;                // [nrows, M] (kSimStore)
  // batched square blocks (blockIdx.y = block): rows/cols [blk_off[b], blk_off[b+1]) of A and B,
  // scores of block b at scores + b * blk_stride, leading dimension = the block's size
  const int32_t* blk_off;
  int64_t blk_stride;
};

__device__ __forceinline__ bool aligned16s(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// monotone map float -> uint32 (total order of non-NaN floats)
__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// TS = 32 x 32 blocks per wave in each direction: 2 = the 128 x 128 tile; 1 = a 64 x 64 tile for the
// small matrices of the training losses (n = 32 ... ~130: ONE 128 x 128 workgroup there is a
// 62 us MFMA chain per wave, four to nine 64 x 64 workgroups run it in a quarter of that).  Same k
// order per output: bit-identical.
#ifdef TILE_TRACE_BUILD
// Timing-only debug build (tools/tile_trace.py --sim): per-workgroup stamps of the counting pass —
// [0] first instruction, [1] K loop start, [3] K loop end, [4] epilogue done (s_memrealtime, 10 ns);
// [5] / [7] s_memtime at [1] / [3]; [6] HW_ID | XCC_ID << 32.
__device__ uint64_t* g_sim_trace = nullptr;
#define SIM_MARK(i)                                                                                     \
  do {                                                                                                  \
    if (MODE == kSimRank && threadIdx.x == 0 && g_sim_trace)                                            \
      g_sim_trace[static_cast<size_t>(blockIdx.x) * 8 + (i)] = wall_clock64();                          \
  } while (0)
#define SIM_CLOCK(i)                                                                                    \
  do {                                                                                                  \
    if (MODE == kSimRank && threadIdx.x == 0 && g_sim_trace)                                            \
      g_sim_trace[static_cast<size_t>(blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime();            \
  } while (0)
#else
#define SIM_MARK(i) do {} while (0)
#define SIM_CLOCK(i) do {} while (0)
#endif

template <int MODE, bool VEC, int TS = 2>
__global__ __launch_bounds__(kThreads) void sim_kernel(const SimParams p_) {
  constexpr int BM = 64 * TS, BN = 64 * TS;
  SIM_MARK(0);
#ifdef TILE_TRACE_BUILD
  if (MODE == kSimRank && threadIdx.x == 0 && g_sim_trace) {
    g_sim_trace[static_cast<size_t>(blockIdx.x) * 8 + 6] =
        static_cast<uint64_t>(__builtin_amdgcn_s_getreg((31 << 11) | 4)) |
        (static_cast<uint64_t>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32);
  }
#endif
  SimParams p = p_;
  if (p.blk_off != nullptr) {
    const int off = p.blk_off[blockIdx.y];
    const int n = p.blk_off[blockIdx.y + 1] - off;
    p.A += static_cast<int64_t>(off) * p.D;
    p.B += static_cast<int64_t>(off) * p.D;
    p.N = p.M = p.nrows = n;
    p.scores += static_cast<int64_t>(blockIdx.y) * p.blk_stride;
    const int jt = blockIdx.x % p.n_tiles, it = blockIdx.x / p.n_tiles;
    if (jt * BN >= n || it * BM >= n) return;  // tile outside this (smaller) block
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int srow = tid >> 2;
  int i0, j0;  // tile origin: local stripe row, global column
  if (MODE == kSimDiag) {
    i0 = blockIdx.x * BM;
    j0 = p.row0 + i0;  // the column block that holds this row block's diagonal
  } else if (MODE == kSimRank && p.m_tiles8 > 0) {
    // Counting pass, XCD-aware: blocks b, b + 8 share an XCD (round-robin dispatch), whose 4 MB L2
    // holds neither operand (2 x 20 MB at N = 4917).  The tiles are dealt in GROUPS — G consecutive
    // row tiles x one column tile — group Q to XCD label Q % 8, its G workgroups back to back in
    // that XCD's order: they run together and pull their B tile through the fabric once instead of
    // G times, and the G A tiles of a row group stay L2-resident while the group's column tiles
    // stream by (931 -> 424 MB per launch at N = 4917, rocprofv3 FETCH_SIZE).  G is the largest of
    // 4, 3, 2 that keeps every XCD within the rounds of resident workgroups a perfectly even deal
    // needs: a launch is two rounds at N = 4917, and one XCD with a handful of workgroups in a
    // third round costs a third of the launch (0.70 ms with a deal by row tile, 0.49 with G = 4,
    // against 0.45 ms).
    const int G = p.m_tiles8;      // row tiles per group (2..4), chosen by the launcher
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int Q = (q / G) * 8 + x;
    const int it = G * (Q / p.n_tiles) + (q % G);
    j0 = (Q % p.n_tiles) * BN;
    i0 = it * BM;
    if (i0 >= p.nrows) return;     // padding: groups beyond the last row group, rows of a partial group
  } else {
    j0 = (blockIdx.x % p.n_tiles) * BN;
    i0 = (blockIdx.x / p.n_tiles) * BM;
  }
  rowaddr_t ar[BM / 64];
  rowaddr_t br[BN / 64];
  bool av[BM / 64], bv[BN / 64];
#pragma unroll
  for (int i = 0; i < BM / 64; ++i) {
    const int li = i0 + srow + 64 * i;
    av[i] = li < p.nrows;
    ar[i] = row_addr(p.A + static_cast<int64_t>(p.row0 + (av[i] ? li : 0)) * p.D);
  }
#pragma unroll
  for (int i = 0; i < BN / 64; ++i) {
    const int j = j0 + srow + 64 * i;
    bv[i] = j < p.M;
    br[i] = row_addr(p.B + static_cast<int64_t>(bv[i] ? j : 0) * p.D);
  }
  f32x16 acc[TS][TS];
#pragma unroll
  for (int ms = 0; ms < TS; ++ms)
#pragma unroll
    for (int ns = 0; ns < TS; ++ns) acc[ms][ns] = zero16();
  int b_row0[TS];
#pragma unroll
  for (int ns = 0; ns < TS; ++ns) b_row0[ns] = wn * 32 * TS + 32 * ns;
  // counting pass: the diagonal scores of the tile's rows, ONE coalesced request in front of the K
  // loop (the epilogue used to load diag[row] inside its 32-iteration loop: 32 dependent global
  // round trips per lane, ~50 of the 62 us the epilogue took — profiles/r04_sim_kernel_trace.txt);
  // they go through LDS once the tile buffers are free
  float my_diag = 0.f;
  if (MODE == kSimRank && tid < BM) my_diag = (i0 + tid < p.nrows) ? p.diag[i0 + tid] : 0.f;
  SIM_MARK(1);
  SIM_CLOCK(5);
  nt_phase<BM, BN, TS, TS, TS, TS - 1, VEC>(smem, ar, av, br, bv, p.D, wm * 32 * TS, b_row0, acc);
  SIM_MARK(3);
  SIM_CLOCK(7);
  if (MODE == kSimRank) {
    // (what is left of the epilogue is ~12 us of vector / scalar work on its own and 48 beside two
    // workgroups inside their MFMA loops on the same SIMDs; raising its wave priority changes
    // nothing: 404 / 407 us per launch)
    __syncthreads();                 // every wave has read its last fragments
    if (tid < BM) smem[tid] = my_diag;
    __syncthreads();
  }

#pragma unroll
  for (int ms = 0; ms < TS; ++ms) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int li = i0 + wm * 32 * TS + ms * 32 + acc_row(r, lane);  // local stripe row
      const int gi = p.row0 + li;                                  // its diagonal column
      const bool rok = li < p.nrows;
      if (MODE == kSimDiag) {
#pragma unroll
        for (int ns = 0; ns < TS; ++ns) {
          const int j = j0 + b_row0[ns] + acc_col(lane);
          if (rok && j == gi) p.diag[li] = acc[ms][ns][r];
        }
      } else if (MODE == kSimStore) {
#pragma unroll
        for (int ns = 0; ns < TS; ++ns) {
          const int j = j0 + b_row0[ns] + acc_col(lane);
          if (rok && j < p.M) p.scores[static_cast<int64_t>(li) * p.M + j] = acc[ms][ns][r];
        }
      } else {
        // The 32 lanes of a half-wave hold the 32 columns of ONE row (lanes 0-31: row acc_row(r, 0),
        // lanes 32-63: the row four below), so the row's count is a population count of the compare
        // mask (v_cmp writes the mask; scalar s_bcnt1) and its arg-max one 32-bit max reduction plus
        // an equality mask whose lowest set lane is the smallest column among the maxima — instead
        // of five shuffle steps on (count, 64-bit key) per element: the epilogue was 27 % of a
        // tile's time, VALU work beside the other workgroups' MFMA loops
        // (profiles/r04_sim_kernel_trace.txt).  Same count, same key (ordered bits of the score,
        // then the smaller column), same atomics.
        const float dii = smem[li - i0];          // (0 for rows past the stripe: never counted)
        const bool hi = lane >= 32;
        int cnt = 0;
        unsigned ob[TS];
        unsigned obmax = 0u;
#pragma unroll
        for (int ns = 0; ns < TS; ++ns) {
          const int j = j0 + b_row0[ns] + acc_col(lane);
          const float v = acc[ms][ns][r];
          const bool valid = rok && j < p.M;
          const unsigned long long gt = __ballot(valid && j != gi && v > dii);
          cnt += __popc(hi ? static_cast<unsigned>(gt >> 32) : static_cast<unsigned>(gt));
          ob[ns] = valid ? ordered_bits(v) : 0u;
          obmax = ob[ns] > obmax ? ob[ns] : obmax;
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {      // (xor masks < 32 stay inside a half-wave)
          const unsigned o = __shfl_xor(obmax, d, 64);
          obmax = o > obmax ? o : obmax;
        }
        int jbest = -1;
#pragma unroll
        for (int ns = 0; ns < TS; ++ns) {        // ns = 0 holds the smaller columns
          const unsigned long long eq = __ballot(obmax != 0u && ob[ns] == obmax);
          const unsigned m = hi ? static_cast<unsigned>(eq >> 32) : static_cast<unsigned>(eq);
          if (jbest < 0 && m != 0u) jbest = j0 + b_row0[ns] + (__ffs(m) - 1);
        }
        if (rok && (lane & 31) == 0) {
          if (cnt) atomicAdd(&p.rank[li], cnt);
          if (jbest >= 0)
            atomicMax(&p.top1key[li], (static_cast<unsigned long long>(obmax) << 32) |
                                          static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<unsigned>(jbest)));
        }
      }
    }
  }
#ifdef TILE_TRACE_BUILD
  __builtin_amdgcn_s_waitcnt(0);
  SIM_MARK(4);
#endif
}

__global__ void top1_finalize_kernel(const unsigned long long* key, int32_t* top1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) top1[i] = static_cast<int32_t>(0xFFFFFFFFu - static_cast<unsigned>(key[i] & 0xFFFFFFFFull));
}

// ---------------------------------------------------------------------------------------------
// Contrastive loss reduction over a stored n x n score matrix (loss.py:89-117).
// Workgroup b < nrb reduces rows [64b, 64b+64) (cost_s); workgroup nrb + c reduces columns
// [64c, 64c+64) (cost_im).  fp64 partial sums, fixed order -> bitwise reproducible.
// ---------------------------------------------------------------------------------------------
struct LossParams {
  const float* scores;  // [n, n]
  int32_t n, nrb;
  float margin;
  int32_t max_violation, norm;
  double* partial;  // [2 * nrb]
  float* loss;
  const int32_t* blk_off;  // batched blocks (blockIdx.y): see SimParams
  int64_t blk_stride;
};

__device__ __forceinline__ LossParams loss_block(const LossParams& q) {
  LossParams p = q;
  if (p.blk_off != nullptr) {
    p.n = p.blk_off[blockIdx.y + 1] - p.blk_off[blockIdx.y];
    p.scores += static_cast<int64_t>(blockIdx.y) * p.blk_stride;
    p.partial += static_cast<int64_t>(blockIdx.y) * 2 * p.nrb;
    p.loss += blockIdx.y;
  }
  return p;
}

__global__ __launch_bounds__(kThreads) void contrastive_partial_kernel(const LossParams p_) {
  const LossParams p = loss_block(p_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n;
  const float* S = p.scores;
  __shared__ double s_part[kThreads];
  double total = 0.0;
  if (static_cast<int>(blockIdx.x) < p.nrb) {
    // rows: one wave per row, lanes stride the columns
    const int r0 = blockIdx.x * 64;
    for (int i = r0 + wave; i < r0 + 64 && i < n; i += kThreads / 64) {
      const float dii = S[static_cast<int64_t>(i) * n + i];
      double sum = 0.0;
      float mx = 0.f;
      for (int j = lane; j < n; j += 64) {
        float c = fmaxf(p.margin + S[static_cast<int64_t>(i) * n + j] - dii, 0.f);
        if (j == i) c = 0.f;
        sum += c;
        mx = fmaxf(mx, c);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d, 64);
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
      }
      total += p.max_violation ? static_cast<double>(mx) : sum;
    }
    s_part[tid] = (lane == 0) ? total : 0.0;
  } else {
    // columns: thread c < 64 owns column j (coalesced across the wave), 4 waves split the rows
    const int j = (blockIdx.x - p.nrb) * 64 + lane;
    double sum = 0.0;
    float mx = 0.f;
    if (j < n) {
      const float djj = S[static_cast<int64_t>(j) * n + j];
      for (int i = wave; i < n; i += kThreads / 64) {
        float c = fmaxf(p.margin + S[static_cast<int64_t>(i) * n + j] - djj, 0.f);
        if (i == j) c = 0.f;
        sum += c;
        mx = fmaxf(mx, c);
      }
    }
    // combine the 4 waves' shares of each column, then the 64 columns
    __shared__ double c_sum[kThreads];
    __shared__ float c_max[kThreads];
    c_sum[tid] = sum;
    c_max[tid] = mx;
    __syncthreads();
    double v = 0.0;
    if (wave == 0) {
      const double cs = c_sum[lane] + c_sum[64 + lane] + c_sum[128 + lane] + c_sum[192 + lane];
      const float cm = fmaxf(fmaxf(c_max[lane], c_max[64 + lane]),
                             fmaxf(c_max[128 + lane], c_max[192 + lane]));
      v = p.max_violation ? static_cast<double>(cm) : cs;
    }
    s_part[tid] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int i = 0; i < kThreads; ++i) t += s_part[i];
    p.partial[blockIdx.x] = t;
  }
}

__global__ void contrastive_final_kernel(const LossParams p_) {
  const LossParams p = loss_block(p_);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double cs = 0.0, ci = 0.0;
    for (int b = 0; b < p.nrb; ++b) cs += p.partial[b];
    for (int b = 0; b < p.nrb; ++b) ci += p.partial[p.nrb + b];
    // loss.py:113-117: cost_s.sum() + cost_im.sum(), each an fp32 tensor sum
    float loss = static_cast<float>(cs) + static_cast<float>(ci);
    if (p.norm) loss = loss / static_cast<float>(static_cast<int64_t>(p.n) * p.n);
    *p.loss = loss;
  }
}

static inline size_t align_up_(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int launch_sim_store(const float* A, const float* B, int n, int m, int D, float* scores,
                            hipStream_t stream) {
  SimParams p;
  p.A = A;
  p.B = B;
  p.N = n;
  p.M = m;
  p.D = D;
  p.row0 = 0;
  p.nrows = n;
  const bool small = n <= kSimSmallMax && m <= kSimSmallMax;   // training-loss sizes: 64 x 64 tiles
  const int bt = small ? 64 : kSimBN;
  p.n_tiles = (m + bt - 1) / bt;
  p.m_tiles8 = 0;
  p.diag = nullptr;
  p.rank = nullptr;
  p.top1key = nullptr;
  p.scores = scores;
  p.blk_off = nullptr;
  p.blk_stride = 0;
  const int64_t blocks = static_cast<int64_t>(p.n_tiles) * ((n + bt - 1) / bt);
  if (blocks > 0x7fffffffLL) return CMHSE_ERR_UNSUPPORTED;
  const size_t smem = small ? TileSmem<64, 64>::kBytes : TileSmem<kSimBM, kSimBN>::kBytes;
  const dim3 grid(static_cast<unsigned>(blocks));
  if (small && D % 4 == 0)
    hipLaunchKernelGGL((sim_kernel<kSimStore, true, 1>), grid, dim3(kThreads), smem, stream, p);
  else if (small)
    hipLaunchKernelGGL((sim_kernel<kSimStore, false, 1>), grid, dim3(kThreads), smem, stream, p);
  else if (D % 4 == 0)
    hipLaunchKernelGGL((sim_kernel<kSimStore, true>), grid, dim3(kThreads), smem, stream, p);
  else
    hipLaunchKernelGGL((sim_kernel<kSimStore, false>), grid, dim3(kThreads), smem, stream, p);
  return CMHSE_OK;
}

}  // namespace cmhse

using namespace cmhse;

extern "C" size_t cmhse_sim_rank_workspace(int32_t nrows) {
  if (nrows <= 0) return 0;
  return align_up_(static_cast<size_t>(nrows) * sizeof(float), 256) +
         align_up_(static_cast<size_t>(nrows) * sizeof(unsigned long long), 256);
}

extern "C" int cmhse_sim_rank_ex(const float* A, const float* B, int32_t N, int32_t M, int32_t D,
                                 int32_t row0, int32_t nrows, int32_t* rank, int32_t* top1,
                                 void* workspace, size_t workspace_bytes, void* stream_,
                                 void* timer_) {
  if (!A || !B || !rank || !top1 || !workspace) return CMHSE_ERR_ARG;
  if (N <= 0 || M <= 0 || D <= 0 || row0 < 0 || nrows < 0 || row0 + nrows > N) return CMHSE_ERR_ARG;
  if (row0 + nrows > M) return CMHSE_ERR_ARG;  // diagonal d[i][i] must exist
  if (nrows == 0) return CMHSE_OK;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_sim_rank_workspace(nrows))
    return CMHSE_ERR_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SimParams p;
  p.A = A;
  p.B = B;
  p.N = N;
  p.M = M;
  p.D = D;
  p.row0 = row0;
  p.nrows = nrows;
  p.n_tiles = (M + kSimBN - 1) / kSimBN;
  p.diag = static_cast<float*>(workspace);
  p.top1key = reinterpret_cast<unsigned long long*>(
      static_cast<char*>(workspace) + align_up_(static_cast<size_t>(nrows) * sizeof(float), 256));
  p.rank = rank;
  p.scores = nullptr;
  p.blk_off = nullptr;
  p.blk_stride = 0;
  const int m_tiles = (nrows + kSimBM - 1) / kSimBM;
  // groups of G row tiles x 1 column tile dealt round-robin to the 8 XCD labels (see sim_kernel);
  // 96 = workgroups resident per XCD (3 per CU at 41 KB of LDS)
  const int64_t tiles = static_cast<int64_t>(m_tiles) * p.n_tiles;
  const int64_t ideal_rounds = (tiles + 767) / 768;
  int G = 0;
  for (int g = 4; g >= 2 && G == 0; --g) {
    const int64_t groups = static_cast<int64_t>((m_tiles + g - 1) / g) * p.n_tiles;
    if (((groups + 7) / 8 * g + 95) / 96 <= ideal_rounds) G = g;
  }
  p.m_tiles8 = G;
  int64_t blocks = tiles;
  if (G > 0) blocks = (static_cast<int64_t>((m_tiles + G - 1) / G) * p.n_tiles + 7) / 8 * 8 * G;
  if (blocks > 0x7fffffffLL) return CMHSE_ERR_UNSUPPORTED;
  if (hipMemsetAsync(rank, 0, sizeof(int32_t) * nrows, stream) != hipSuccess) return CMHSE_ERR_LAUNCH;
  if (hipMemsetAsync(p.top1key, 0, sizeof(unsigned long long) * nrows, stream) != hipSuccess)
    return CMHSE_ERR_LAUNCH;
  const size_t smem = TileSmem<kSimBM, kSimBN>::kBytes;
  Timer* timer = static_cast<Timer*>(timer_);   // brackets the counting pass alone
  if (D % 4 == 0) {
    hipLaunchKernelGGL((sim_kernel<kSimDiag, true>), dim3(m_tiles), dim3(kThreads), smem, stream, p);
    if (timer) (void)hipEventRecord(timer->start, stream);
    hipLaunchKernelGGL((sim_kernel<kSimRank, true>), dim3(static_cast<unsigned>(blocks)),
                       dim3(kThreads), smem, stream, p);
  } else {
    hipLaunchKernelGGL((sim_kernel<kSimDiag, false>), dim3(m_tiles), dim3(kThreads), smem, stream, p);
    if (timer) (void)hipEventRecord(timer->start, stream);
    hipLaunchKernelGGL((sim_kernel<kSimRank, false>), dim3(static_cast<unsigned>(blocks)),
                       dim3(kThreads), smem, stream, p);
  }
  if (timer) {
    (void)hipEventRecord(timer->stop, stream);
    timer->launches = 1;
  }
  hipLaunchKernelGGL(top1_finalize_kernel, dim3((nrows + 255) / 256), dim3(256), 0, stream,
                     p.top1key, top1, nrows);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" int cmhse_sim_rank(const float* A, const float* B, int32_t N, int32_t M, int32_t D,
                              int32_t row0, int32_t nrows, int32_t* rank, int32_t* top1,
                              void* workspace, size_t workspace_bytes, void* stream_) {
  return cmhse_sim_rank_ex(A, B, N, M, D, row0, nrows, rank, top1, workspace, workspace_bytes,
                           stream_, nullptr);
}

extern "C" int cmhse_cosine_sim(const float* im, const float* s, int32_t n, int32_t m, int32_t D,
                                float* scores, void* stream_) {
  if (!im || !s || !scores || n <= 0 || m <= 0 || D <= 0) return CMHSE_ERR_ARG;
  const int rc = launch_sim_store(im, s, n, m, D, scores, static_cast<hipStream_t>(stream_));
  if (rc != CMHSE_OK) return rc;
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" size_t cmhse_contrastive_workspace(int32_t n) {
  if (n <= 0) return 0;
  const size_t nrb = (n + 63) / 64;
  return align_up_(static_cast<size_t>(n) * n * sizeof(float), 256) +
         align_up_(2 * nrb * sizeof(double), 256);
}

extern "C" int cmhse_contrastive_fwd(const float* im, const float* s, int32_t n, int32_t D,
                                     float margin, int32_t max_violation, int32_t norm,
                                     float* loss, float* scores_out, void* workspace,
                                     size_t workspace_bytes, void* stream_) {
  if (!im || !s || !loss || !workspace || n <= 0 || D <= 0) return CMHSE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_contrastive_workspace(n))
    return CMHSE_ERR_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* scores = scores_out ? scores_out : static_cast<float*>(workspace);
  const int rc = launch_sim_store(im, s, n, n, D, scores, stream);
  if (rc != CMHSE_OK) return rc;
  LossParams lp;
  lp.scores = scores;
  lp.n = n;
  lp.nrb = (n + 63) / 64;
  lp.margin = margin;
  lp.max_violation = max_violation;
  lp.norm = norm;
  lp.partial = reinterpret_cast<double*>(static_cast<char*>(workspace) +
                                         align_up_(static_cast<size_t>(n) * n * sizeof(float), 256));
  lp.loss = loss;
  lp.blk_off = nullptr;
  lp.blk_stride = 0;
  hipLaunchKernelGGL(contrastive_partial_kernel, dim3(2 * lp.nrb), dim3(kThreads), 0, stream, lp);
  hipLaunchKernelGGL(contrastive_final_kernel, dim3(1), dim3(64), 0, stream, lp);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" size_t cmhse_contrastive_blocks_workspace(int32_t n_blocks, int32_t max_n) {
  if (n_blocks <= 0 || max_n <= 0) return 0;
  const size_t nrb = (max_n + 63) / 64;
  return align_up_(static_cast<size_t>(n_blocks) * max_n * max_n * sizeof(float), 256) +
         align_up_(static_cast<size_t>(n_blocks) * 2 * nrb * sizeof(double), 256);
}

extern "C" int cmhse_contrastive_blocks_fwd(const float* im, const float* s,
                                            const int32_t* blk_off, int32_t n_blocks,
                                            int32_t max_n, int32_t D, float margin,
                                            int32_t max_violation, int32_t norm, float* losses,
                                            void* workspace, size_t workspace_bytes,
                                            void* stream_) {
  if (!im || !s || !blk_off || !losses || !workspace || n_blocks <= 0 || max_n <= 0 || D <= 0)
    return CMHSE_ERR_ARG;
  if (n_blocks > 65535) return CMHSE_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_contrastive_blocks_workspace(n_blocks, max_n))
    return CMHSE_ERR_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  SimParams p;
  p.A = im;
  p.B = s;
  p.N = p.M = p.nrows = max_n;
  p.D = D;
  p.row0 = 0;
  const bool small = max_n <= kSimSmallMax;
  const int bt = small ? 64 : kSimBN;
  p.n_tiles = (max_n + bt - 1) / bt;
  p.m_tiles8 = 0;
  p.diag = nullptr;
  p.rank = nullptr;
  p.top1key = nullptr;
  p.scores = static_cast<float*>(workspace);
  p.blk_off = blk_off;
  p.blk_stride = static_cast<int64_t>(max_n) * max_n;
  const unsigned tiles = static_cast<unsigned>(p.n_tiles) * ((max_n + bt - 1) / bt);
  const size_t smem = small ? TileSmem<64, 64>::kBytes : TileSmem<kSimBM, kSimBN>::kBytes;
  if (small && D % 4 == 0)
    hipLaunchKernelGGL((sim_kernel<kSimStore, true, 1>), dim3(tiles, n_blocks), dim3(kThreads), smem,
                       stream, p);
  else if (small)
    hipLaunchKernelGGL((sim_kernel<kSimStore, false, 1>), dim3(tiles, n_blocks), dim3(kThreads), smem,
                       stream, p);
  else if (D % 4 == 0)
    hipLaunchKernelGGL((sim_kernel<kSimStore, true>), dim3(tiles, n_blocks), dim3(kThreads), smem,
                       stream, p);
  else
    hipLaunchKernelGGL((sim_kernel<kSimStore, false>), dim3(tiles, n_blocks), dim3(kThreads), smem,
                       stream, p);
  LossParams lp;
  lp.scores = p.scores;
  lp.n = max_n;
  lp.nrb = (max_n + 63) / 64;
  lp.margin = margin;
  lp.max_violation = max_violation;
  lp.norm = norm;
  lp.partial = reinterpret_cast<double*>(
      static_cast<char*>(workspace) +
      align_up_(static_cast<size_t>(n_blocks) * max_n * max_n * sizeof(float), 256));
  lp.loss = losses;
  lp.blk_off = blk_off;
  lp.blk_stride = p.blk_stride;
  hipLaunchKernelGGL(contrastive_partial_kernel, dim3(2 * lp.nrb, n_blocks), dim3(kThreads), 0,
                     stream, lp);
  hipLaunchKernelGGL(contrastive_final_kernel, dim3(1, n_blocks), dim3(64), 0, stream, lp);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// cmhse_step_losses_fwd: F.normalize of the step's encoder outputs written straight into the
// row-blocked operands of the batched ContrastiveLoss, the losses, and their weighted total.
// ---------------------------------------------------------------------------------------------
struct StepNormParams {
  const float* src[2][CMHSE_STEP_LOSS_MAX];  // [side][term]: the embedding that is the term's a / b
  int32_t n[CMHSE_STEP_LOSS_MAX];            // term sizes
  float weight[CMHSE_STEP_LOSS_MAX];
  int32_t n_terms, D, R;
  float* y[2];        // [R, D] each
  int32_t* blk_off;   // [n_terms + 1]
};

// one workgroup per operand row (2R of them); the arithmetic of l2norm_rows_kernel, order included
__global__ __launch_bounds__(kThreads) void step_norm_kernel(const StepNormParams p) {
  const int side = (static_cast<int>(blockIdx.x) >= p.R) ? 1 : 0;
  const int r = static_cast<int>(blockIdx.x) - side * p.R;
  const float* xr = nullptr;
  int off = 0;
#pragma unroll
  for (int k = 0; k < CMHSE_STEP_LOSS_MAX; ++k) {
    if (k < p.n_terms) {
      if (r >= off && r < off + p.n[k])
        xr = (side ? p.src[1][k] : p.src[0][k]) + static_cast<int64_t>(r - off) * p.D;
      off += p.n[k];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int o = 0;
    p.blk_off[0] = 0;
#pragma unroll
    for (int k = 0; k < CMHSE_STEP_LOSS_MAX; ++k)
      if (k < p.n_terms) {
        o += p.n[k];
        p.blk_off[k + 1] = o;
      }
  }
  float* yr = p.y[side] + static_cast<int64_t>(r) * p.D;
  float ss = 0.f;
  for (int c = threadIdx.x; c < p.D; c += kThreads) {
    const float v = xr[c];
    ss += v * v;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d, 64);
  __shared__ float s_part[kThreads / 64];
  __shared__ float s_inv;
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kThreads / 64; ++i) t += s_part[i];
    s_inv = 1.0f / fmaxf(sqrtf(t), 1e-12f);
  }
  __syncthreads();
  const float inv = s_inv;
  for (int c = threadIdx.x; c < p.D; c += kThreads) yr[c] = xr[c] * inv;
}

__global__ void step_total_kernel(const StepNormParams p, const float* values, float* total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < CMHSE_STEP_LOSS_MAX; ++k)
      if (k < p.n_terms) t += p.weight[k] * values[k];
    *total = t;
  }
}

// GroupWiseContrastiveLoss (loss.py:26-38): block (video i's clips x video j's captions) max or
// mean of the clip x caption score matrix.  One workgroup per block; the arg-max (row-major first
// maximum) is kept for the backward pass.
struct BlockReduceParams {
  const float* scores;  // [n, n]
  const int32_t* row_off;
  const int32_t* col_off;
  float* reduced;   // [B, B]
  int32_t* arg;     // [B, B] flat index r*n + c of the block maximum
  int32_t n, B, use_max;
};

__global__ __launch_bounds__(kThreads) void block_reduce_kernel(const BlockReduceParams q) {
  const int bi = blockIdx.y, bj = blockIdx.x, tid = threadIdx.x;
  const int r0 = q.row_off[bi], r1 = q.row_off[bi + 1], c0 = q.col_off[bj], c1 = q.col_off[bj + 1];
  const int w = c1 - c0, cnt = (r1 - r0) * w;
  double sum = 0.0;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  for (int e = tid; e < cnt; e += kThreads) {
    const int r = r0 + e / w, c = c0 + e % w;
    const float v = q.scores[static_cast<int64_t>(r) * q.n + c];
    sum += v;
    const int flat = r * q.n + c;
    if (v > best || (v == best && flat < arg)) {
      best = v;
      arg = flat;
    }
  }
  __shared__ double s_sum[kThreads];
  __shared__ float s_best[kThreads];
  __shared__ int s_arg[kThreads];
  s_sum[tid] = sum;
  s_best[tid] = best;
  s_arg[tid] = arg;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    float b = -INFINITY;
    int a = 0x7fffffff;
    for (int i = 0; i < kThreads; ++i) {
      t += s_sum[i];
      if (s_best[i] > b || (s_best[i] == b && s_arg[i] < a)) {
        b = s_best[i];
        a = s_arg[i];
      }
    }
    q.reduced[bi * q.B + bj] = q.use_max ? b : static_cast<float>(t / cnt);
    q.arg[bi * q.B + bj] = a;
  }
}

extern "C" size_t cmhse_step_losses_workspace(const cmhse_step_losses* d) {
  StepLossLayout L;
  return step_loss_layout(d, &L) ? L.bytes : 0;
}

extern "C" int cmhse_step_losses_fwd(const cmhse_step_losses* d, float* values, float* total,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
  StepLossLayout L;
  if (!values || !total || !workspace || !step_loss_layout(d, &L)) return CMHSE_ERR_ARG;
  for (int e = 0; e < d->n_emb; ++e)
    if (!d->x[e]) return CMHSE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 || workspace_bytes < L.bytes)
    return CMHSE_ERR_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  StepNormParams p = {};
  for (int k = 0; k < d->n_terms; ++k) {
    p.src[0][k] = d->x[d->term_a[k]];
    p.src[1][k] = d->x[d->term_b[k]];
    p.n[k] = d->rows[d->term_a[k]];
    p.weight[k] = d->weight[k];
  }
  p.n_terms = d->n_terms;
  p.D = d->D;
  p.R = L.R;
  p.y[0] = reinterpret_cast<float*>(ws + L.y_im);
  p.y[1] = reinterpret_cast<float*>(ws + L.y_s);
  p.blk_off = reinterpret_cast<int32_t*>(ws + L.blk_off);
  hipLaunchKernelGGL(step_norm_kernel, dim3(2u * L.R), dim3(kThreads), 0, stream, p);
  const int rc = cmhse_contrastive_blocks_fwd(
      p.y[0], p.y[1], p.blk_off, d->n_terms, L.max_n, d->D, d->margin, d->max_violation, d->norm,
      values, ws + L.fwd_ws, cmhse_contrastive_blocks_workspace(d->n_terms, L.max_n), stream_);
  if (rc != CMHSE_OK) return rc;
  hipLaunchKernelGGL(step_total_kernel, dim3(1), dim3(64), 0, stream, p, values, total);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

extern "C" size_t cmhse_groupwise_workspace(int32_t n, int32_t B) {
  if (n <= 0 || B <= 0) return 0;
  const size_t nrb = (B + 63) / 64;
  return align_up_(static_cast<size_t>(n) * n * sizeof(float), 256) +
         align_up_(2 * nrb * sizeof(double), 256);
}

extern "C" int cmhse_groupwise_fwd(const float* im, const float* s, int32_t n, int32_t D,
                                   const int32_t* row_off, const int32_t* col_off, int32_t B,
                                   float margin, int32_t max_violation, int32_t norm, float* loss,
                                   float* reduced, int32_t* arg, float* scores_out,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
  if (!im || !s || !row_off || !col_off || !loss || !reduced || !arg || !workspace || n <= 0 ||
      D <= 0 || B <= 0 || B > 65535)
    return CMHSE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0 ||
      workspace_bytes < cmhse_groupwise_workspace(n, B))
    return CMHSE_ERR_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* scores = scores_out ? scores_out : static_cast<float*>(workspace);
  const int rc = launch_sim_store(im, s, n, n, D, scores, stream);
  if (rc != CMHSE_OK) return rc;
  BlockReduceParams bp;
  bp.scores = scores; bp.row_off = row_off; bp.col_off = col_off; bp.reduced = reduced;
  bp.arg = arg; bp.n = n; bp.B = B; bp.use_max = max_violation;
  hipLaunchKernelGGL(block_reduce_kernel, dim3(B, B), dim3(kThreads), 0, stream, bp);
  LossParams lp;
  lp.scores = reduced;
  lp.n = B;
  lp.nrb = (B + 63) / 64;
  lp.margin = margin;
  lp.max_violation = max_violation;
  lp.norm = norm;
  lp.partial = reinterpret_cast<double*>(static_cast<char*>(workspace) +
                                         align_up_(static_cast<size_t>(n) * n * sizeof(float), 256));
  lp.loss = loss;
  lp.blk_off = nullptr;
  lp.blk_stride = 0;
  hipLaunchKernelGGL(contrastive_partial_kernel, dim3(2 * lp.nrb), dim3(kThreads), 0, stream, lp);
  hipLaunchKernelGGL(contrastive_final_kernel, dim3(1), dim3(64), 0, stream, lp);
  return (hipGetLastError() == hipSuccess) ? CMHSE_OK : CMHSE_ERR_LAUNCH;
}

#ifdef TILE_TRACE_BUILD
extern "C" int cmhse_debug_set_sim_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(cmhse::g_sim_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" const char* cmhse_strerror(int code) {
  switch (code) {
    case CMHSE_OK: return "ok";
    case CMHSE_ERR_ARG: return "invalid argument";
    case CMHSE_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
    case CMHSE_ERR_LAUNCH: return "HIP launch/runtime error";
    case CMHSE_ERR_UNSUPPORTED: return "shape not supported";
    case CMHSE_ERR_TIMEOUT:
      return "a resident chain kernel gave up at a grid barrier (its workgroups were not all on the chip: "
             "shared GPU, CU mask?); results of that call are invalid; clear with cmhse_async_status(1) and "
             "set the *_tail_min_steps / *_chain_min_steps tunables to 0 on such a GPU";
    default: return "unknown error";
  }
}

extern "C" const char* cmhse_version(void) { return "cmhse_hip 0.6.0 gfx950"; }
